"""ctypes binding of ``libfrmap_hip.so`` (C ABI declared in ``include/frmap_hip.h``).

There is deliberately no fallback: if the shared library has not been built
(``python -c "import __graft_entry__ as g; g.build()"`` or ``csrc/build.sh``) every compute entry
point raises ``RuntimeError`` — the product path never routes through PyTorch eager or the CPU
oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# FRMAP_LIB: load another build of the library (A/B variants made by tools; must have the same ABI) instead of the in-tree one
LIB_PATH = os.environ.get("FRMAP_LIB") or os.path.join(_PKG_DIR, "libfrmap_hip.so")
ABI_VERSION = 9

_lock = threading.Lock()
_lib = None

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/frmap_hip.h one to one
PROTOTYPES = {
    "frmap_abi_version": (_i, []),
    "frmap_set_batch_invariant": (_i, [_i]),
    "frmap_last_error": (C.c_char_p, []),
    "frmap_pack_input_nchw_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "frmap_pack_conv_weight": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "frmap_small_cin_kpad": (_i, [_i, _i]),
    "frmap_pack_conv_weight_c3": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "frmap_conv_small_cin": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "frmap_match_gallery_pack_bytes": (_sz, [_i, _i]),
    "frmap_match_pack_gallery": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "frmap_match_pack_gallery_rows": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "frmap_match_top1_packed": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _vp]),
    "frmap_resize_bilinear_u8": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "frmap_gap_linear_norm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _i, _i, _vp]),
    "frmap_conv_small_cin_pool2": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "frmap_conv_igemm_pool2_supported": (_i, [_i, _i, _i, _i, _i]),
    "frmap_conv_igemm_pool2_form": (_i, [_i, _i, _i, _i, _i]),
    "frmap_conv3x3_pp_pool_layout": (_i, [_i, _i, _i, _i, _i]),
    "frmap_conv_igemm_pool2": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "frmap_stem7x7_maxpool": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "frmap_stem7x7_maxpool2": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "frmap_stem7x7_maxpool_u8": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "frmap_conv_igemm": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "frmap_conv_pp_tuning": (_i, [_i, _i, _i]),
    "frmap_conv_pp_ri": (_i, [_i]),
    "frmap_conv_pp_pitch": (_i, [_i]),
    "frmap_conv_pp_ds": (_i, [_i]),
    "frmap_conv_pp_im": (_i, [_i]),
    "frmap_conv1x1_pp_layout": (_i, [_i, _i, _i, _i, _i, _i]),
    "frmap_conv3x3_pp_layout": (_i, [_i, _i, _i, _i, _i]),
    "frmap_conv3x3s2_pp_layout": (_i, [_i, _i, _i, _i, _i]),
    "frmap_conv3x3_pp_ds_layout": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "frmap_conv_igemm_ds_supported": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "frmap_conv_igemm_ds": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "frmap_linear_mfma_workspace_bytes": (_sz, [_i, _i, _i]),
    "frmap_linear_mfma": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "frmap_maxpool": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "frmap_avgpool_global": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "frmap_avgpool_adaptive": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "frmap_linear_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "frmap_l2_normalize_f32": (_i, [_vp, _vp, _i, _i, _f, _vp]),
    "frmap_cast_to_f32": (_i, [_vp, _vp, _sz, _i, _vp]),
    "frmap_cast_from_f32": (_i, [_vp, _vp, _sz, _i, _vp]),
    "frmap_add_pos_layernorm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "frmap_mha_tokens": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "frmap_mean_layernorm": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "frmap_cnn_attention": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "frmap_normalize_u8_hwc": (_i, [_vp, _vp, _vp, _i, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _i, _vp]),
    "frmap_softmax_argmax": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "frmap_pairwise_distance": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "frmap_head_workspace_bytes": (_sz, [_i, _i]),
    "frmap_match_workspace_bytes": (_sz, [_i, _i]),
    "frmap_match_top1": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _vp]),
    "frmap_gap_norm_match": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _f, _i, _i, _i, _i, _i, _vp]),
    "frmap_cosine_logits": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "frmap_arcmargin_eval": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp]),
    # model handles
    "frmap_model_create": (_i, [C.POINTER(_vp), C.c_char_p, _i, _i]),
    "frmap_model_load_tensor": (_i, [_vp, C.c_char_p, _vp, _sz, _i]),
    "frmap_model_set_input_normalization": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "frmap_model_finalize": (_i, [_vp, _vp]),
    "frmap_model_embedding_dim": (_i, [_vp]),
    "frmap_model_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "frmap_model_match_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "frmap_model_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "frmap_model_embed_and_match": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "frmap_model_trace": (_i, [_vp, _i]),
    "frmap_model_trace_read": (_i, [_vp, _vp, _i]),
    "frmap_model_destroy": (None, [_vp]),
}


def lib_available() -> bool:
    return os.path.isfile(LIB_PATH)


def load() -> C.CDLL:
    """Load (once) and return the library; raise loudly if it is missing or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not lib_available():
            raise RuntimeError(
                f"HIP extension not built: {LIB_PATH} is missing. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU / PyTorch fallback for this path.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if lib.frmap_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libfrmap_hip.so ABI {lib.frmap_abi_version()} != expected {ABI_VERSION}; rebuild")
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().frmap_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise RuntimeError(f"{what}: {msg} (rc={rc})")
