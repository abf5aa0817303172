"""Thin tensor-level wrappers over the C ABI (``include/frmap_hip.h``).

PyTorch is used here only for device memory (``torch.empty``) and the current HIP stream; every
function validates that its operands live on the GPU and hands raw pointers to the library.
Activations are NHWC tensors in the compute dtype (bf16 / fp16); heads and matching are fp32.
"""
from __future__ import annotations

import functools
from typing import Optional, Tuple

import torch

from . import _lib

BF16, F16 = 0, 1
_DT = {torch.bfloat16: BF16, torch.float16: F16}


def _stream() -> int:
    # the CURRENT device's current stream: every public wrapper runs under `_on_operand_device`, which makes the
    # operands' device current first (HIP's current device is per thread, SURVEY.md §8b threading)
    return torch.cuda.current_stream().cuda_stream


def _on_operand_device(fn):
    """Launch on the device the operands live on, whatever the calling thread's current device is: a model on
    ``cuda:1`` works from a thread whose current device is ``cuda:0`` (`src/app.py:43` moves the input to
    ``next(model.parameters()).device``; `:331-335` calls the model from a daemon thread).  Operands on two
    different GPUs are rejected."""
    @functools.wraps(fn)
    def guarded(*args, **kwargs):
        dev = None
        for v in args:
            if isinstance(v, torch.Tensor) and v.is_cuda:
                if dev is None:
                    dev = v.device
                elif v.device != dev:
                    raise ValueError(f"{fn.__name__}: operands live on different devices ({dev} and {v.device})")
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return guarded


def _dev(t: torch.Tensor, what: str, dtype=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{what}: expected a tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; the HIP path needs GPU tensors (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{what}: expected dtype {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def dt_code(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f"compute dtype must be torch.bfloat16 or torch.float16, got {dtype}") from None


def set_batch_invariant(on: Optional[bool]) -> None:
    """``True``: kernel / tile-layout choices depend on the per-image geometry only, so a face's result is bit-identical in
    any batch, shard or rank (`frmap_set_batch_invariant`); ``False``: default planning (layouts follow the tile count: faster
    at small batches, results of different batch sizes agree to rounding); ``None``: the environment (FRMAP_BATCH_INVARIANT)."""
    _lib.check(_lib.load().frmap_set_batch_invariant(-1 if on is None else int(bool(on))), "set_batch_invariant")


def pack_input(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """fp32 NCHW B×3×H×W -> NHWC4 (zero 4th channel) in ``dtype``."""
    x = _dev(x, "pack_input.x", torch.float32)
    if x.dim() != 4 or x.shape[1] != 3:
        raise ValueError(f"expected B×3×H×W input, got {tuple(x.shape)}")
    B, _, H, W = x.shape
    out = torch.empty((B, H, W, 4), dtype=dtype, device=x.device)
    _lib.check(_lib.load().frmap_pack_input_nchw_f32(x.data_ptr(), out.data_ptr(), B, H, W, dt_code(dtype), _stream()),
               "pack_input")
    return out


def pack_conv_weight(w_folded: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    w = _dev(w_folded, "pack_conv_weight.w", torch.float32)
    Cout, Cin, KH, KW = w.shape
    out = torch.empty((Cout * Cin * KH * KW,), dtype=dtype, device=w.device)
    _lib.check(_lib.load().frmap_pack_conv_weight(w.data_ptr(), out.data_ptr(), Cout, Cin, KH, KW, dt_code(dtype),
                                                  _stream()), "pack_conv_weight")
    return out


def pack_conv_weight_c3(w_folded: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    w = _dev(w_folded, "pack_conv_weight_c3.w", torch.float32)
    Cout, Cin, KH, KW = w.shape
    if Cin != 3:
        raise ValueError("pack_conv_weight_c3: Cin must be 3")
    lib = _lib.load()
    out = torch.empty((Cout * lib.frmap_small_cin_kpad(KH, KW),), dtype=dtype, device=w.device)
    _lib.check(lib.frmap_pack_conv_weight_c3(w.data_ptr(), out.data_ptr(), Cout, KH, KW, dt_code(dtype), _stream()),
               "pack_conv_weight_c3")
    return out


def conv_small_cin(x4: torch.Tensor, wpk: torch.Tensor, shift: torch.Tensor, Cout: int, k: int, stride: int,
                   pad: int, relu: bool) -> torch.Tensor:
    x4 = _dev(x4, "conv_small_cin.x")
    B, H, W, C = x4.shape
    if C != 4:
        raise ValueError("conv_small_cin: input must be NHWC4")
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((B, Ho, Wo, Cout), dtype=x4.dtype, device=x4.device)
    _lib.check(_lib.load().frmap_conv_small_cin(x4.data_ptr(), _dev(wpk, "wpk").data_ptr(),
                                                _dev(shift, "shift", torch.float32).data_ptr(), out.data_ptr(),
                                                B, H, W, Cout, k, k, stride, pad, int(relu), dt_code(x4.dtype),
                                                _stream()), "conv_small_cin")
    return out


def conv_small_cin_pool2(x4: torch.Tensor, wpk: torch.Tensor, shift: torch.Tensor, Cout: int, relu: bool) -> torch.Tensor:
    """3x3 s1 p1 conv (Cin = 3 as NHWC4) + shift (+ReLU) + MaxPool2d(2, 2) in one launch (`face_models.py:38`)."""
    x4 = _dev(x4, "conv_small_cin_pool2.x")
    B, H, W, C = x4.shape
    if C != 4:
        raise ValueError("conv_small_cin_pool2: input must be NHWC4")
    if H % 2 or W % 2:
        raise ValueError("conv_small_cin_pool2: H and W must be even")
    out = torch.empty((B, H // 2, W // 2, Cout), dtype=x4.dtype, device=x4.device)
    _lib.check(_lib.load().frmap_conv_small_cin_pool2(x4.data_ptr(), _dev(wpk, "wpk").data_ptr(),
                                                      _dev(shift, "shift", torch.float32).data_ptr(), out.data_ptr(),
                                                      B, H, W, Cout, int(relu), dt_code(x4.dtype), _stream()),
               "conv_small_cin_pool2")
    return out


def stem_pool_dims(H: int, W: int):
    Hc, Wc = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    return (Hc + 2 - 3) // 2 + 1, (Wc + 2 - 3) // 2 + 1


def stem7x7_maxpool(x: torch.Tensor, wpk: torch.Tensor, shift: torch.Tensor, dtype: torch.dtype,
                    pool3: bool = True) -> torch.Tensor:
    """fp32 NCHW -> conv7x7 s2 + shift + ReLU -> maxpool (3x3 s2 p1 if ``pool3`` else 2x2 s2) -> NHWC
    B×Hq×Wq×64 (``dtype``)."""
    x = _dev(x, "stem7x7_maxpool.x", torch.float32)
    B, C, H, W = x.shape
    if C != 3:
        raise ValueError("stem7x7_maxpool: expected 3 input channels")
    if pool3:
        Hq, Wq = stem_pool_dims(H, W)
    else:
        Hq, Wq = ((H + 6 - 7) // 2 + 1) // 2, ((W + 6 - 7) // 2 + 1) // 2
    out = torch.empty((B, Hq, Wq, 64), dtype=dtype, device=x.device)
    fn = _lib.load().frmap_stem7x7_maxpool if pool3 else _lib.load().frmap_stem7x7_maxpool2
    _lib.check(fn(x.data_ptr(), _dev(wpk, "wpk", dtype).data_ptr(),
                                                 _dev(shift, "shift", torch.float32).data_ptr(), out.data_ptr(),
                                                 B, H, W, dt_code(dtype), _stream()), "stem7x7_maxpool")
    return out


def stem7x7_maxpool_u8(x_u8: torch.Tensor, wpk: torch.Tensor, shift: torch.Tensor, mean, std, dtype: torch.dtype,
                       pool3: bool = True) -> torch.Tensor:
    """uint8 HWC B×H×W×3 (RGB) -> ToTensor + Normalize(mean, std) -> conv7x7 s2 + shift + ReLU -> maxpool -> NHWC
    B×Hq×Wq×64 (``dtype``): the fused stem fed by the image bytes themselves (needs W % 4 == 0)."""
    import ctypes as C
    if not isinstance(x_u8, torch.Tensor) or x_u8.dtype != torch.uint8 or x_u8.dim() != 4 or x_u8.shape[3] != 3:
        raise TypeError("stem7x7_maxpool_u8: expected a uint8 tensor of shape [B, H, W, 3]")
    x_u8 = _dev(x_u8, "stem7x7_maxpool_u8.x")
    B, H, W, _ = x_u8.shape
    if pool3:
        Hq, Wq = stem_pool_dims(H, W)
    else:
        Hq, Wq = ((H + 6 - 7) // 2 + 1) // 2, ((W + 6 - 7) // 2 + 1) // 2
    out = torch.empty((B, Hq, Wq, 64), dtype=dtype, device=x_u8.device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    sd = (C.c_float * 3)(*[float(v) for v in std])
    _lib.check(_lib.load().frmap_stem7x7_maxpool_u8(x_u8.data_ptr(), m, sd, _dev(wpk, "wpk", dtype).data_ptr(),
                                                    _dev(shift, "shift", torch.float32).data_ptr(), out.data_ptr(),
                                                    B, H, W, int(bool(pool3)), dt_code(dtype), _stream()), "stem7x7_maxpool_u8")
    return out


def conv_igemm(x: torch.Tensor, wpk: torch.Tensor, shift: torch.Tensor, Cout: int, k: int, stride: int, pad: int,
               relu, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``relu``: False/0 none, True/1 ReLU, 2 exact GELU."""
    x = _dev(x, "conv_igemm.x")
    B, H, W, Cin = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((B, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
    res_ptr = 0
    if residual is not None:
        residual = _dev(residual, "conv_igemm.residual", x.dtype)
        if tuple(residual.shape) != tuple(out.shape):
            raise ValueError(f"conv_igemm: residual shape {tuple(residual.shape)} != output {tuple(out.shape)}")
        res_ptr = residual.data_ptr()
    if wpk.numel() != Cout * Cin * k * k or wpk.dtype != x.dtype:
        raise ValueError("conv_igemm: packed weight does not match Cout*Cin*k*k / dtype")
    if shift.numel() != Cout:
        raise ValueError("conv_igemm: shift must have Cout elements")
    _lib.check(_lib.load().frmap_conv_igemm(x.data_ptr(), _dev(wpk, "wpk").data_ptr(),
                                            _dev(shift, "shift", torch.float32).data_ptr(), res_ptr, out.data_ptr(),
                                            B, H, W, Cin, Cout, k, stride, pad, int(relu), dt_code(x.dtype),
                                            _stream()), "conv_igemm")
    return out


_pool2_cache: dict = {}


def conv_pool2_supported(B: int, H: int, W: int, Cin: int, Cout: int) -> bool:
    """True when fusing the 2x2 max-pool into the conv is expected to win (see `frmap_conv_igemm_pool2_supported`)."""
    key = (B, H, W, Cin, Cout)
    if key not in _pool2_cache:
        _pool2_cache[key] = bool(_lib.load().frmap_conv_igemm_pool2_supported(B, H, W, Cin, Cout))
    return _pool2_cache[key]


def conv_pool2_form(B: int, H: int, W: int, Cin: int, Cout: int) -> int:
    """3 = ping-pong kernel, 2 = wave kernel, 1 = generic kernel, 0 = `conv_igemm_pool2` rejects the shape."""
    return int(_lib.load().frmap_conv_igemm_pool2_form(B, H, W, Cin, Cout))


def conv_igemm_pool2(x: torch.Tensor, wpk: torch.Tensor, shift: torch.Tensor, Cout: int, relu) -> torch.Tensor:
    """conv 3x3 s1 p1 + shift (+ReLU) + MaxPool2d(2, 2) in one launch (`face_models.py:39-40,121-141`); the conv map
    never reaches HBM.  Raises ``ValueError`` for shapes `conv_pool2_supported` rejects."""
    x = _dev(x, "conv_igemm_pool2.x")
    B, H, W, Cin = x.shape
    if wpk.numel() != Cout * Cin * 9 or wpk.dtype != x.dtype:
        raise ValueError("conv_igemm_pool2: packed weight does not match Cout*Cin*9 / dtype")
    if shift.numel() != Cout:
        raise ValueError("conv_igemm_pool2: shift must have Cout elements")
    out = torch.empty((B, H // 2, W // 2, Cout), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().frmap_conv_igemm_pool2(x.data_ptr(), _dev(wpk, "wpk").data_ptr(),
                                                  _dev(shift, "shift", torch.float32).data_ptr(), out.data_ptr(),
                                                  B, H, W, Cin, Cout, int(relu), dt_code(x.dtype), _stream()),
               "conv_igemm_pool2")
    return out


_ds_ok_cache: dict = {}


def conv_ds_supported(B: int, H: int, W: int, Cin: int, Cout: int, dsH: int, dsW: int, dsCin: int, ds_stride: int) -> bool:
    key = (B, H, W, Cin, Cout, dsH, dsW, dsCin, ds_stride)
    v = _ds_ok_cache.get(key)
    if v is None:
        v = _ds_ok_cache[key] = bool(_lib.load().frmap_conv_igemm_ds_supported(*key))
    return v


def conv_igemm_ds(x: torch.Tensor, wpk: torch.Tensor, shift: torch.Tensor, Cout: int, x_ds: torch.Tensor, wpk_ds: torch.Tensor,
                  ds_stride: int, relu) -> torch.Tensor:
    """``act(conv3x3_s1_p1(x) + conv1x1_stride(x_ds) + shift)`` — a BasicBlock's second conv with its projection
    shortcut folded in (``shift`` = both folded BatchNorm shifts added).  Check ``conv_ds_supported`` first."""
    x = _dev(x, "conv_igemm_ds.x")
    x_ds = _dev(x_ds, "conv_igemm_ds.x_ds", x.dtype)
    B, H, W, Cin = x.shape
    Bd, Hd, Wd, Cd = x_ds.shape
    if Bd != B or wpk.numel() != Cout * Cin * 9 or wpk_ds.numel() != Cout * Cd or wpk.dtype != x.dtype or wpk_ds.dtype != x.dtype:
        raise ValueError("conv_igemm_ds: operand shapes / dtypes do not match")
    if shift.numel() != Cout:
        raise ValueError("conv_igemm_ds: shift must have Cout elements")
    out = torch.empty((B, H, W, Cout), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().frmap_conv_igemm_ds(x.data_ptr(), _dev(wpk, "wpk").data_ptr(),
                                               _dev(shift, "shift", torch.float32).data_ptr(), x_ds.data_ptr(),
                                               _dev(wpk_ds, "wpk_ds").data_ptr(), out.data_ptr(), B, H, W, Cin, Cout,
                                               Hd, Wd, Cd, ds_stride, int(relu), dt_code(x.dtype), _stream()), "conv_igemm_ds")
    return out


def linear_mfma(x2d: torch.Tensor, wpk: torch.Tensor, shift: torch.Tensor, N: int, act=0,
                residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``act(x · Wᵀ + shift [+ residual])`` on the MFMA conv kernel (split-K when the output is small)."""
    x2d = _dev(x2d, "linear_mfma.x")
    M, K = x2d.shape
    out = torch.empty((M, N), dtype=x2d.dtype, device=x2d.device)
    lib = _lib.load()
    nbytes = lib.frmap_linear_mfma_workspace_bytes(M, K, N)
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=x2d.device) if nbytes else None
    rp = 0
    if residual is not None:
        residual = _dev(residual, "linear_mfma.residual", x2d.dtype)
        if tuple(residual.shape) != (M, N):
            raise ValueError("linear_mfma: residual must be [M, N]")
        rp = residual.data_ptr()
    if wpk.numel() != N * K or wpk.dtype != x2d.dtype:
        raise ValueError("linear_mfma: packed weight does not match N*K / dtype")
    _lib.check(lib.frmap_linear_mfma(x2d.data_ptr(), _dev(wpk, "wpk").data_ptr(), _dev(shift, "shift", torch.float32).data_ptr(),
                                     rp, out.data_ptr(), ws.data_ptr() if ws is not None else 0, M, K, N, int(act),
                                     dt_code(x2d.dtype), _stream()), "linear_mfma")
    return out


def maxpool(x: torch.Tensor, k: int, stride: int, pad: int) -> torch.Tensor:
    x = _dev(x, "maxpool.x")
    B, H, W, Cc = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((B, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().frmap_maxpool(x.data_ptr(), out.data_ptr(), B, H, W, Cc, k, stride, pad, dt_code(x.dtype),
                                         _stream()), "maxpool")
    return out


def avgpool_global(x: torch.Tensor) -> torch.Tensor:
    x = _dev(x, "avgpool_global.x")
    B, H, W, Cc = x.shape
    out = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().frmap_avgpool_global(x.data_ptr(), out.data_ptr(), B, H * W, Cc, dt_code(x.dtype),
                                                _stream()), "avgpool_global")
    return out


def avgpool_adaptive(x: torch.Tensor, OH: int, OW: int) -> torch.Tensor:
    x = _dev(x, "avgpool_adaptive.x")
    B, H, W, Cc = x.shape
    out = torch.empty((B, OH, OW, Cc), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().frmap_avgpool_adaptive(x.data_ptr(), out.data_ptr(), B, H, W, Cc, OH, OW,
                                                  dt_code(x.dtype), _stream()), "avgpool_adaptive")
    return out


def linear_f32(x: torch.Tensor, w: torch.Tensor, scale: Optional[torch.Tensor] = None,
               shift: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    x = _dev(x, "linear_f32.x", torch.float32)
    w = _dev(w, "linear_f32.w", torch.float32)
    B, K = x.shape
    N, K2 = w.shape
    if K != K2:
        raise ValueError(f"linear_f32: x is B×{K} but w is N×{K2}")
    out = torch.empty((B, N), dtype=torch.float32, device=x.device)
    sp = _dev(scale, "scale", torch.float32).data_ptr() if scale is not None else 0
    hp = _dev(shift, "shift", torch.float32).data_ptr() if shift is not None else 0
    _lib.check(_lib.load().frmap_linear_f32(x.data_ptr(), w.data_ptr(), sp, hp, out.data_ptr(), B, K, N, int(relu),
                                            _stream()), "linear_f32")
    return out


def l2_normalize(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    x = _dev(x, "l2_normalize.x", torch.float32)
    B, D = x.shape
    out = torch.empty_like(x)
    _lib.check(_lib.load().frmap_l2_normalize_f32(x.data_ptr(), out.data_ptr(), B, D, float(eps), _stream()),
               "l2_normalize")
    return out


def cast_to_f32(x: torch.Tensor) -> torch.Tensor:
    x = _dev(x, "cast_to_f32.x")
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().frmap_cast_to_f32(x.data_ptr(), out.data_ptr(), x.numel(), dt_code(x.dtype), _stream()),
               "cast_to_f32")
    return out


def cast_from_f32(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    x = _dev(x, "cast_from_f32.x", torch.float32)
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _lib.check(_lib.load().frmap_cast_from_f32(x.data_ptr(), out.data_ptr(), x.numel(), dt_code(dtype), _stream()),
               "cast_from_f32")
    return out


def add_pos_layernorm(x: torch.Tensor, pos: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                      eps: float = 1e-5, want_sum: bool = False):
    """x: [B, L, D] tokens.  Returns (t, y): t = x + pos (None unless want_sum), y = LayerNorm(t)."""
    x = _dev(x, "add_pos_layernorm.x")
    B, L, D = x.shape
    y = torch.empty_like(x)
    t = torch.empty_like(x) if want_sum else None
    _lib.check(_lib.load().frmap_add_pos_layernorm(
        x.data_ptr(), _dev(pos, "pos", torch.float32).data_ptr() if pos is not None else 0,
        _dev(gamma, "gamma", torch.float32).data_ptr(), _dev(beta, "beta", torch.float32).data_ptr(),
        t.data_ptr() if t is not None else 0, y.data_ptr(), B, L, D, float(eps), dt_code(x.dtype), _stream()),
        "add_pos_layernorm")
    return t, y


def mha_tokens(qkv: torch.Tensor, H: int) -> torch.Tensor:
    qkv = _dev(qkv, "mha_tokens.qkv")
    B, L, D3 = qkv.shape
    D = D3 // 3
    out = torch.empty((B, L, D), dtype=qkv.dtype, device=qkv.device)
    _lib.check(_lib.load().frmap_mha_tokens(qkv.data_ptr(), out.data_ptr(), B, L, D, H, dt_code(qkv.dtype), _stream()),
               "mha_tokens")
    return out


def mean_layernorm(t: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    t = _dev(t, "mean_layernorm.t")
    B, L, D = t.shape
    out = torch.empty((B, D), dtype=torch.float32, device=t.device)
    _lib.check(_lib.load().frmap_mean_layernorm(t.data_ptr(), _dev(gamma, "gamma", torch.float32).data_ptr(),
                                                _dev(beta, "beta", torch.float32).data_ptr(), out.data_ptr(), B, L, D,
                                                float(eps), dt_code(t.dtype), _stream()), "mean_layernorm")
    return out


def cnn_attention(qkv: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, spatial_w: torch.Tensor, spatial_b: torch.Tensor,
                  Cq: int, want_map: bool = False, want_pool: bool = True):
    """AttentionNet's attention block (`face_models.py:194-262`) on an NHWC trunk map ``x`` [B,H,W,C] given the packed
    q|k|v projection ``qkv`` [B,H,W,2*Cq+C].  Returns (map [B,H,W,C] | None, pooled fp32 [B,C] | None)."""
    x = _dev(x, "cnn_attention.x")
    qkv = _dev(qkv, "cnn_attention.qkv", x.dtype)
    B, H, W, C = x.shape
    if tuple(qkv.shape) != (B, H, W, 2 * Cq + C):
        raise ValueError(f"cnn_attention: qkv shape {tuple(qkv.shape)} != {(B, H, W, 2 * Cq + C)}")
    sw = _dev(spatial_w, "spatial_w", torch.float32)
    KS = sw.shape[-1]
    if sw.numel() != 2 * KS * KS:
        raise ValueError("cnn_attention: spatial_w must be [1][2][KS][KS]")
    om = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device) if want_map else None
    op = torch.empty((B, C), dtype=torch.float32, device=x.device) if want_pool else None
    _lib.check(_lib.load().frmap_cnn_attention(qkv.data_ptr(), x.data_ptr(), _dev(gamma, "gamma", torch.float32).data_ptr(),
                                               sw.data_ptr(), _dev(spatial_b, "spatial_b", torch.float32).data_ptr(),
                                               om.data_ptr() if om is not None else 0, op.data_ptr() if op is not None else 0,
                                               B, H, W, Cq, C, KS, dt_code(x.dtype), _stream()), "cnn_attention")
    return om, op


def normalize_u8(img: torch.Tensor, mean, std, want_nchw: bool = True, nhwc4_dtype: Optional[torch.dtype] = None):
    """uint8 [B, H, W, 3] RGB on the GPU → ToTensor + Normalize.  Returns (fp32 NCHW | None, NHWC4 | None)."""
    import ctypes as C
    if not isinstance(img, torch.Tensor) or img.dtype != torch.uint8 or img.dim() != 4 or img.shape[3] != 3:
        raise TypeError("normalize_u8: expected a uint8 tensor of shape [B, H, W, 3]")
    img = _dev(img, "normalize_u8.img")
    B, H, W, _ = img.shape
    o1 = torch.empty((B, 3, H, W), dtype=torch.float32, device=img.device) if want_nchw else None
    o2 = torch.empty((B, H, W, 4), dtype=nhwc4_dtype, device=img.device) if nhwc4_dtype is not None else None
    m = (C.c_float * 3)(*[float(v) for v in mean])
    sd = (C.c_float * 3)(*[float(v) for v in std])
    _lib.check(_lib.load().frmap_normalize_u8_hwc(img.data_ptr(), o1.data_ptr() if o1 is not None else 0,
                                                  o2.data_ptr() if o2 is not None else 0, B, H, W, m, sd,
                                                  dt_code(nhwc4_dtype) if nhwc4_dtype is not None else BF16, _stream()),
               "normalize_u8")
    return o1, o2


def softmax_argmax(logits: torch.Tensor, want_probs: bool = True):
    logits = _dev(logits, "softmax_argmax.logits", torch.float32)
    B, Cc = logits.shape
    probs = torch.empty_like(logits) if want_probs else None
    pred = torch.empty((B,), dtype=torch.int32, device=logits.device)
    _lib.check(_lib.load().frmap_softmax_argmax(logits.data_ptr(), probs.data_ptr() if want_probs else 0, pred.data_ptr(),
                                                B, Cc, _stream()), "softmax_argmax")
    return probs, pred


def pairwise_distance(a: torch.Tensor, b: torch.Tensor, thresh: Optional[float] = None):
    a = _dev(a, "pairwise_distance.a", torch.float32)
    b = _dev(b, "pairwise_distance.b", torch.float32)
    if a.shape != b.shape or a.dim() != 2:
        raise ValueError("pairwise_distance: need two [B, D] tensors of the same shape")
    B, D = a.shape
    dist = torch.empty((B,), dtype=torch.float32, device=a.device)
    same = torch.empty((B,), dtype=torch.int32, device=a.device) if thresh is not None else None
    _lib.check(_lib.load().frmap_pairwise_distance(a.data_ptr(), b.data_ptr(), dist.data_ptr(),
                                                   same.data_ptr() if same is not None else 0,
                                                   float(thresh) if thresh is not None else 0.0, B, D, _stream()),
               "pairwise_distance")
    return dist, same


def _workspace(B: int, Cc: int, device) -> torch.Tensor:
    n = _lib.load().frmap_head_workspace_bytes(B, Cc)
    return torch.empty(((n + 15) // 16) * 2, dtype=torch.int64, device=device)


def _match_workspace(B: int, G: int, device) -> torch.Tensor:
    n = _lib.load().frmap_match_workspace_bytes(B, G)
    return torch.empty(((n + 15) // 16) * 2, dtype=torch.int64, device=device)


def _packed_buf(packed, B: int, device):
    """``packed``: False/None -> no record output; True -> a fresh int32 [B, 2]; a tensor -> written in place (a
    contiguous int32 [B, 2] view, e.g. one micro-batch's slice of the step's record buffer)."""
    if packed is None or packed is False:
        return None
    if packed is True:
        return torch.empty((B, 2), dtype=torch.int32, device=device)
    if not (isinstance(packed, torch.Tensor) and packed.is_cuda and packed.dtype == torch.int32 and tuple(packed.shape) == (B, 2)
            and packed.is_contiguous()):
        raise ValueError(f"packed output must be a contiguous int32 [{B}, 2] device tensor")
    return packed


class MatchPack:
    """A gallery prepared for the MFMA match path (`frmap_match_pack_gallery`): the fp32 rows split into fp16 (hi, lo)
    pairs in the GEMM kernel's operand order, plus the per-row statistics of the expanded distance.  ``capacity`` > G
    leaves room for `update_rows` (incremental enrolment: only the touched 64-row tiles are re-packed)."""
    __slots__ = ("packed", "stat_w", "G", "D", "capacity", "src_ptr", "src_version", "ready")

    def __init__(self, gallery: torch.Tensor, capacity: Optional[int] = None):
        gallery = _dev(gallery, "match_prepare.gallery", torch.float32)
        self.G, self.D = int(gallery.shape[0]), int(gallery.shape[1])
        self.capacity = max(int(capacity or 0), self.G)
        lib = _lib.load()
        self.packed = torch.empty((lib.frmap_match_gallery_pack_bytes(self.capacity, self.D),), dtype=torch.uint8, device=gallery.device)
        self.stat_w = torch.empty((self.capacity, 4), dtype=torch.float32, device=gallery.device)
        self.src_ptr, self.src_version = gallery.data_ptr(), gallery._version
        _lib.check(lib.frmap_match_pack_gallery(gallery.data_ptr(), self.packed.data_ptr(), self.stat_w.data_ptr(), self.G, self.D,
                                                _stream()), "match_pack_gallery")
        self.ready = torch.cuda.Event()
        self.ready.record()          # consumers on other streams wait for the pack kernels (`wait_ready`)

    def update_rows(self, gallery: torch.Tensor, row_lo: int, row_hi: int) -> None:
        """``gallery`` (same storage, now G rows) had rows [row_lo, row_hi) appended or edited: re-pack those rows only."""
        gallery = _dev(gallery, "match_update.gallery", torch.float32)
        G = int(gallery.shape[0])
        if gallery.data_ptr() != self.src_ptr or gallery.shape[1] != self.D or G > self.capacity:
            raise ValueError("MatchPack.update_rows: not the gallery storage this pack was built from (or beyond its capacity)")
        torch.cuda.current_stream().wait_event(self.ready)
        _lib.check(_lib.load().frmap_match_pack_gallery_rows(gallery.data_ptr(), self.packed.data_ptr(), self.stat_w.data_ptr(),
                                                             int(row_lo), int(row_hi), G, self.D, _stream()), "match_pack_gallery_rows")
        self.G, self.src_version = G, gallery._version
        self.ready = torch.cuda.Event()
        self.ready.record()

    def wait_ready(self) -> None:
        torch.cuda.current_stream().wait_event(self.ready)

    def matches(self, gallery: torch.Tensor) -> bool:
        return (gallery.data_ptr() == self.src_ptr and gallery._version == self.src_version and
                tuple(gallery.shape) == (self.G, self.D) and gallery.device == self.packed.device)


MATCH_MFMA_MIN_G = 512   # galleries at least this large take the MFMA path when a MatchPack is supplied


@_on_operand_device
def match_prepare(gallery: torch.Tensor) -> MatchPack:
    return MatchPack(gallery)


def match_top1(emb: torch.Tensor, gallery: torch.Tensor, thresh: Optional[float] = None, packed: bool = False,
               prepared: Optional[MatchPack] = None):
    """First arg-min over gallery rows of ``||e - g + 1e-6||_2`` and that distance (int32[B], fp32[B]).
    With ``thresh`` a third tensor is returned: idx where dist <= thresh else -1 ("Unknown");
    with ``packed`` a fourth: int32[B, 2] = (id-or-unknown, bits of dist), the all-gather record.
    ``prepared`` (`match_prepare(gallery)`): run the gallery scan on the fp16 MFMA pipe (same contract)."""
    emb = _dev(emb, "match_top1.emb", torch.float32)
    B, D = emb.shape
    G = int(gallery.shape[0]) if gallery is not None else 0
    if prepared is not None and G >= MATCH_MFMA_MIN_G and D % 32 == 0:
        gallery = _dev(gallery, "match_top1.gallery", torch.float32)
        if not prepared.matches(gallery):
            raise ValueError("match_top1: `prepared` was built from a different (or since modified) gallery")
        idx = torch.empty((B,), dtype=torch.int32, device=emb.device)
        dist = torch.empty((B,), dtype=torch.float32, device=emb.device)
        if D != prepared.D:
            raise ValueError(f"match_top1: embedding dim {D} != prepared gallery dim {prepared.D}")
        prepared.wait_ready()
        ws = _match_workspace(B, G, emb.device)       # candidate records + per-probe statistics
        split = torch.empty((B, 3 * D), dtype=torch.float16, device=emb.device)
        ids = torch.empty((B,), dtype=torch.int32, device=emb.device) if thresh is not None else None
        pk = _packed_buf(packed, B, emb.device)
        _lib.check(_lib.load().frmap_match_top1_packed(emb.data_ptr(), gallery.data_ptr(), prepared.packed.data_ptr(),
                                                       prepared.stat_w.data_ptr(), idx.data_ptr(), dist.data_ptr(),
                                                       ids.data_ptr() if ids is not None else 0,
                                                       pk.data_ptr() if pk is not None else 0,
                                                       float(thresh) if thresh is not None else float("inf"), ws.data_ptr(),
                                                       split.data_ptr(), B, G, D, _stream()), "match_top1_packed")
        if pk is not None:
            return idx, dist, ids, pk
        return (idx, dist) if thresh is None else (idx, dist, ids)
    gptr = 0
    if G > 0:
        gallery = _dev(gallery, "match_top1.gallery", torch.float32)
        if gallery.shape[1] != D:
            raise ValueError(f"match_top1: embedding dim {D} != gallery dim {gallery.shape[1]}")
        gptr = gallery.data_ptr()
    idx = torch.empty((B,), dtype=torch.int32, device=emb.device)
    dist = torch.empty((B,), dtype=torch.float32, device=emb.device)
    ws = _match_workspace(B, G, emb.device)
    ids = torch.empty((B,), dtype=torch.int32, device=emb.device) if thresh is not None else None
    pk = _packed_buf(packed, B, emb.device)
    _lib.check(_lib.load().frmap_match_top1(emb.data_ptr(), gptr, idx.data_ptr(), dist.data_ptr(),
                                            ids.data_ptr() if ids is not None else 0,
                                            pk.data_ptr() if pk is not None else 0,
                                            float(thresh) if thresh is not None else float("inf"), ws.data_ptr(),
                                            B, G, D, _stream()), "match_top1")
    if pk is not None:
        return idx, dist, ids, pk
    return (idx, dist) if thresh is None else (idx, dist, ids)


def gap_linear_norm(fmap: torch.Tensor, wt: torch.Tensor, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor],
                    eps: float = 1e-12, want_pre: bool = False, relu: bool = False):
    """ArcFaceNet head in one launch (`face_models.py:573-590`): global average pool of the NHWC trunk map [B,H,W,K] ->
    Linear (``wt`` = weight transposed, fp32 [K][N]) -> folded BatchNorm1d -> L2-normalise.  Returns ``(emb, pre | None)``."""
    fmap = _dev(fmap, "gap_linear_norm.map")
    B, H, W, K = fmap.shape
    wt = _dev(wt, "gap_linear_norm.wt", torch.float32)
    if wt.dim() != 2 or wt.shape[0] != K or not wt.is_contiguous():
        raise ValueError(f"gap_linear_norm: wt must be a contiguous [{K}][N] matrix")
    N = int(wt.shape[1])
    emb = torch.empty((B, N), dtype=torch.float32, device=fmap.device)
    pre = torch.empty((B, N), dtype=torch.float32, device=fmap.device) if want_pre else None
    _lib.check(_lib.load().frmap_gap_linear_norm(fmap.data_ptr(), wt.data_ptr(),
                                                 _dev(scale, "scale", torch.float32).data_ptr() if scale is not None else 0,
                                                 _dev(shift, "shift", torch.float32).data_ptr() if shift is not None else 0,
                                                 pre.data_ptr() if pre is not None else 0, emb.data_ptr(), float(eps),
                                                 B, H * W, K, N, int(bool(relu)), dt_code(fmap.dtype), _stream()), "gap_linear_norm")
    return emb, pre


def gap_norm_match(fmap: torch.Tensor, gallery: torch.Tensor, thresh: Optional[float] = None, normalize: bool = False,
                   eps: float = 1e-12, want_emb: bool = False, packed: bool = False):
    """Global-average-pool an NHWC trunk map [B,H,W,C], optionally L2-normalise, and match against a small gallery
    (<= 64 rows) in one launch.  Returns (idx, dist, ids | None, packed | None, emb | None)."""
    fmap = _dev(fmap, "gap_norm_match.map")
    B, H, W, Cc = fmap.shape
    G = int(gallery.shape[0]) if gallery is not None else 0
    gptr = 0
    if G > 0:
        gallery = _dev(gallery, "gap_norm_match.gallery", torch.float32)
        if gallery.shape[1] != Cc:
            raise ValueError(f"gap_norm_match: map channels {Cc} != gallery dim {gallery.shape[1]}")
        gptr = gallery.data_ptr()
    idx = torch.empty((B,), dtype=torch.int32, device=fmap.device)
    dist = torch.empty((B,), dtype=torch.float32, device=fmap.device)
    ids = torch.empty((B,), dtype=torch.int32, device=fmap.device) if thresh is not None else None
    pk = _packed_buf(packed, B, fmap.device)
    emb = torch.empty((B, Cc), dtype=torch.float32, device=fmap.device) if want_emb else None
    _lib.check(_lib.load().frmap_gap_norm_match(fmap.data_ptr(), gptr, emb.data_ptr() if emb is not None else 0, idx.data_ptr(),
                                                dist.data_ptr(), ids.data_ptr() if ids is not None else 0,
                                                pk.data_ptr() if pk is not None else 0,
                                                float(thresh) if thresh is not None else float("inf"), int(bool(normalize)),
                                                float(eps), B, H * W, Cc, G, dt_code(fmap.dtype), _stream()), "gap_norm_match")
    return idx, dist, ids, pk, emb


def cosine_logits(x: torch.Tensor, w: torch.Tensor, s: float = 1.0, want_logits: bool = True,
                  want_argmax: bool = True):
    x = _dev(x, "cosine_logits.x", torch.float32)
    w = _dev(w, "cosine_logits.w", torch.float32)
    B, D = x.shape
    Cc = w.shape[0]
    logits = torch.empty((B, Cc), dtype=torch.float32, device=x.device) if want_logits else None
    arg = torch.empty((B,), dtype=torch.int32, device=x.device) if want_argmax else None
    ws = _workspace(B, Cc, x.device)
    _lib.check(_lib.load().frmap_cosine_logits(x.data_ptr(), w.data_ptr(), logits.data_ptr() if want_logits else 0,
                                               arg.data_ptr() if want_argmax else 0, ws.data_ptr(), B, Cc, D,
                                               float(s), _stream()), "cosine_logits")
    return logits, arg


def arcmargin_eval(x: torch.Tensor, w: torch.Tensor, label: torch.Tensor, s: float, m: float,
                   easy_margin: bool = False, want_minmax: bool = False):
    x = _dev(x, "arcmargin_eval.x", torch.float32)
    w = _dev(w, "arcmargin_eval.w", torch.float32)
    label = _dev(label, "arcmargin_eval.label", torch.int64)
    B, D = x.shape
    Cc = w.shape[0]
    if label.numel() != B:
        raise ValueError("arcmargin_eval: one label per row required")
    out = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
    mm = torch.empty((2,), dtype=torch.float32, device=x.device) if want_minmax else None
    ws = _workspace(B, Cc, x.device)
    _lib.check(_lib.load().frmap_arcmargin_eval(x.data_ptr(), w.data_ptr(), label.data_ptr(), out.data_ptr(),
                                                mm.data_ptr() if want_minmax else 0, ws.data_ptr(), B, Cc, D,
                                                float(s), float(m), int(easy_margin), _stream()), "arcmargin_eval")
    return out, mm


# every tensor-taking wrapper launches on its operands' device (see _on_operand_device)
for _name in ("pack_input", "pack_conv_weight", "pack_conv_weight_c3", "conv_small_cin", "stem7x7_maxpool", "stem7x7_maxpool_u8", "conv_igemm",
              "conv_igemm_ds", "linear_mfma", "maxpool", "avgpool_global", "avgpool_adaptive", "linear_f32", "l2_normalize",
              "cast_to_f32", "cast_from_f32", "add_pos_layernorm", "mha_tokens", "mean_layernorm", "cnn_attention",
              "normalize_u8", "softmax_argmax", "pairwise_distance", "match_top1", "gap_norm_match", "cosine_logits",
              "arcmargin_eval", "conv_small_cin_pool2", "conv_igemm_pool2", "gap_linear_norm"):
    globals()[_name] = _on_operand_device(globals()[_name])
del _name


# ------------------------------------------------------------------------------------------------
# model handles (`frmap_model_*`): the per-layer plan, BatchNorm folding and weight packing live in the library
# ------------------------------------------------------------------------------------------------
IN_F32_NCHW, IN_U8_HWC = 0, 1
OUT_TRUNK_MAP, OUT_POOLED, OUT_EMBEDDING, OUT_LOGITS = 0, 1, 2, 3


class ModelHandle:
    """A `frmap_model` built from a module's ``state_dict`` (the reference's key names, device tensors).  Immutable after
    construction; forwards allocate only their outputs and scratch (torch's caching allocator, capturable into a HIP graph)."""

    def __init__(self, model_type: str, state: dict, num_classes: int, dtype: torch.dtype, mean=None, std=None):
        import ctypes as C
        lib = _lib.load()
        self._lib, self._h = lib, C.c_void_p()
        self.dtype, self.model_type, self.num_classes = dtype, model_type, int(num_classes)
        _lib.check(lib.frmap_model_create(C.byref(self._h), model_type.encode(), int(num_classes), dt_code(dtype)), "model_create")
        try:
            dev = None
            for key, t in state.items():
                if not (isinstance(t, torch.Tensor) and t.is_floating_point()):
                    continue
                if not t.is_cuda:
                    raise RuntimeError(f"{key} is on {t.device}; move the module to the GPU (no CPU fallback)")
                dev = t.device if dev is None else dev
                tt = t.detach().to(torch.float32).contiguous()
                with torch.cuda.device(tt.device):
                    # (1 = a key the inference path does not use, e.g. the trunk's own 1000-way fc: ignored)
                    _lib.check(min(lib.frmap_model_load_tensor(self._h, key.encode(), tt.data_ptr(), tt.numel(), 1), 0), f"model_load_tensor({key})")
            self.device = dev
            if mean is not None:
                m3, s3 = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
                _lib.check(lib.frmap_model_set_input_normalization(self._h, m3, s3), "model_set_input_normalization")
            with torch.cuda.device(dev):
                _lib.check(lib.frmap_model_finalize(self._h, _stream()), "model_finalize")
            self.embedding_dim = int(lib.frmap_model_embedding_dim(self._h))
        except Exception:
            lib.frmap_model_destroy(self._h)
            self._h = None
            raise

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        try:
            if h:
                self._lib.frmap_model_destroy(h)
        except Exception:   # interpreter shutdown: the library object may already be gone
            pass

    @staticmethod
    def _geometry(x):
        if x.dtype == torch.uint8:
            return IN_U8_HWC, x.shape[0], x.shape[1], x.shape[2]
        return IN_F32_NCHW, x.shape[0], x.shape[2], x.shape[3]

    def forward(self, x: torch.Tensor, what: int) -> torch.Tensor:
        x = _dev(x, "model_forward.x")
        kind, B, H, W = self._geometry(x)
        with torch.cuda.device(x.device):
            if what == OUT_TRUNK_MAP:
                hq, wq = stem_pool_dims(H, W)
                for _ in range(3):
                    hq, wq = (hq - 1) // 2 + 1, (wq - 1) // 2 + 1
                out = torch.empty((B, hq, wq, 512), dtype=self.dtype, device=x.device)
            else:
                out = torch.empty((B, self.num_classes if what == OUT_LOGITS else self.embedding_dim), dtype=torch.float32, device=x.device)
            ws = torch.empty((self._lib.frmap_model_workspace_bytes(self._h, B, H, W),), dtype=torch.uint8, device=x.device)
            _lib.check(self._lib.frmap_model_forward(self._h, x.data_ptr(), kind, B, H, W, what, out.data_ptr(), ws.data_ptr(), _stream()),
                       "model_forward")
        return out

    def embed_and_match(self, x: torch.Tensor, gallery: Optional[torch.Tensor], prepared, thresh: float, normalize: bool,
                        packed=False, want_emb: bool = False):
        """One C call: forward + top-1 match.  Returns (idx, dist, ids, packed | None, emb | None)."""
        x = _dev(x, "model_embed_and_match.x")
        kind, B, H, W = self._geometry(x)
        G = int(gallery.shape[0]) if gallery is not None else 0
        with torch.cuda.device(x.device):
            gptr = ppk = pst = 0
            if G:
                gallery = _dev(gallery, "model_embed_and_match.gallery", torch.float32)
                if gallery.shape[1] != self.embedding_dim:
                    raise ValueError(f"embed_and_match: embedding dim {self.embedding_dim} != gallery dim {gallery.shape[1]}")
                gptr = gallery.data_ptr()
                if prepared is not None and G >= MATCH_MFMA_MIN_G:
                    if not prepared.matches(gallery):
                        raise ValueError("embed_and_match: `prepared` was built from a different (or since modified) gallery")
                    prepared.wait_ready()
                    ppk, pst = prepared.packed.data_ptr(), prepared.stat_w.data_ptr()
            idx = torch.empty((B,), dtype=torch.int32, device=x.device)
            dist = torch.empty((B,), dtype=torch.float32, device=x.device)
            ids = torch.empty((B,), dtype=torch.int32, device=x.device)
            pk = _packed_buf(packed, B, x.device)
            emb = torch.empty((B, self.embedding_dim), dtype=torch.float32, device=x.device) if want_emb else None
            ws = torch.empty((self._lib.frmap_model_match_workspace_bytes(self._h, B, H, W, G),), dtype=torch.uint8, device=x.device)
            _lib.check(self._lib.frmap_model_embed_and_match(self._h, x.data_ptr(), kind, B, H, W, gptr, ppk, pst, G, float(thresh),
                                                             int(bool(normalize)), idx.data_ptr(), dist.data_ptr(), ids.data_ptr(),
                                                             pk.data_ptr() if pk is not None else 0,
                                                             emb.data_ptr() if emb is not None else 0, ws.data_ptr(), _stream()),
                       "model_embed_and_match")
        return idx, dist, ids, pk, emb

    def trace(self, enable: bool) -> None:
        _lib.check(self._lib.frmap_model_trace(self._h, int(bool(enable))), "model_trace")

    def trace_read(self, max_records: int = 4096):
        """[(kernel label, algorithmic FLOPs, algorithmic bytes, microseconds)] of the forwards since the last read."""
        import ctypes as C

        class Rec(C.Structure):
            _fields_ = [("kernel", C.c_char * 64), ("flop", C.c_double), ("bytes", C.c_double), ("us", C.c_float)]
        buf = (Rec * max_records)()
        n = self._lib.frmap_model_trace_read(self._h, C.cast(buf, C.c_void_p), max_records)
        _lib.check(min(n, 0), "model_trace_read")
        return [(buf[i].kernel.decode(), buf[i].flop, buf[i].bytes, buf[i].us) for i in range(n)]
