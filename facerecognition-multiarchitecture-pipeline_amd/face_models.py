"""Host-side mirror of the reference model zoo's inference surface, computing on the HIP kernels.

Same names, argument meaning, ``state_dict`` keys and error behaviour as
``/root/reference/src/face_models.py`` for the hot path:

* ``get_model(model_type, num_classes=18, input_size=(224, 224))``  (`face_models.py:785-813`)
* ``BaselineNet`` (`:16-60`), ``ResNetTransfer`` (`:62-102`), ``SiameseNet`` (`:104-192`),
  ``ArcMarginProduct`` eval (`:297-445`), ``ArcFaceNet`` eval (`:447-613`), ``HybridNet`` (`:650-721`)

The modules are ordinary ``nn.Module`` parameter containers (so ``.to()``, ``.eval()``,
``.state_dict()`` and ``load_state_dict()`` of a reference ``best_model.pth`` work), but
``forward`` / ``get_embedding`` do not call ``torch.nn`` ops: they fold BatchNorm into the
neighbouring conv / linear (fp32), pack the weights once into the kernels' layout, and launch the
hand-written gfx950 kernels through the C ABI (``ops.py``).  Inputs must be fp32 NCHW tensors on
the GPU and the module must be on the GPU and in ``eval()`` mode; anything else raises — there is
no eager / CPU fallback.  Training (`face_models.py:527-572`, losses `:725-782`) is out of scope.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops

MODEL_TYPES = ['baseline', 'cnn', 'siamese', 'attention', 'arcface', 'hybrid', 'ensemble']

_DEFAULT_DTYPE = torch.bfloat16
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)   # `src/testing.py:102-103`


def set_default_compute_dtype(dtype: torch.dtype) -> None:
    """bf16 (default; BASELINE.json's benchmark precision) or fp16 (the precision at which the
    ≤1e-3 embedding-cosine bound of the north star is stated)."""
    global _DEFAULT_DTYPE
    ops.dt_code(dtype)
    _DEFAULT_DTYPE = dtype


# --------------------------------------------------------------------------------------------
# folding helpers (fp32, one-off per weight version)
# --------------------------------------------------------------------------------------------
def _bn_scale_shift(bn: nn.Module, bias: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Eval BatchNorm as y = x*scale + shift; a preceding conv/linear bias is absorbed in shift."""
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    if bias is not None:
        shift = shift + bias.detach().float() * scale
    return scale.contiguous(), shift.contiguous()


_PY_PLAN = os.environ.get("FRMAP_PY_PLAN", "0") == "1"      # A/B switch: 1 = plan the ResNet-18 families in Python (per-op C ABI calls)
_HEAD_FUSE = os.environ.get("FRMAP_HEAD_FUSE", "1") != "0"   # A/B switch: 0 = ArcFace head as avgpool + linear + normalize launches
_POOL_FUSE = os.environ.get("FRMAP_POOL_FUSE", "1") != "0"   # A/B switch: 0 = conv and 2x2 max-pool as two launches


class _PackedConv:
    __slots__ = ("wpk", "shift", "cout", "k", "stride", "pad", "small")

    def __init__(self, conv: nn.Conv2d, bn: Optional[nn.Module], dtype: torch.dtype):
        w = conv.weight.detach().float()
        if bn is not None:
            scale, shift = _bn_scale_shift(bn, conv.bias)
            w = w * scale.view(-1, 1, 1, 1)
        else:
            shift = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
        self.cout, cin, kh, kw = w.shape
        self.k, self.stride, self.pad = kh, conv.stride[0], conv.padding[0]
        self.small = cin == 3
        self.wpk = ops.pack_conv_weight_c3(w.contiguous(), dtype) if self.small else ops.pack_conv_weight(w.contiguous(), dtype)
        self.shift = shift.contiguous()

    def __call__(self, x: torch.Tensor, relu: bool, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.small:
            return ops.conv_small_cin(x, self.wpk, self.shift, self.cout, self.k, self.stride, self.pad, relu)
        return ops.conv_igemm(x, self.wpk, self.shift, self.cout, self.k, self.stride, self.pad, relu, residual)

    def pooled(self, x: torch.Tensor, relu: bool = True) -> torch.Tensor:
        """conv + shift (+ReLU) + MaxPool2d(2, 2) (`face_models.py:38-40,121-141`): one launch when the kernels take the
        shape (the conv map never reaches HBM), conv + pool kernels otherwise."""
        B, H, W, C = x.shape
        fusable = self.k == 3 and self.stride == 1 and self.pad == 1 and H % 2 == 0 and W % 2 == 0 and _POOL_FUSE
        if fusable and self.small and self.cout == 32:
            return ops.conv_small_cin_pool2(x, self.wpk, self.shift, self.cout, relu)
        if fusable and not self.small and ops.conv_pool2_supported(B, H, W, C, self.cout):
            return ops.conv_igemm_pool2(x, self.wpk, self.shift, self.cout, relu)
        return ops.maxpool(self(x, relu), 2, 2, 0)


class _PackedLinearAsConv:
    """Wide nn.Linear (+BatchNorm1d) run on the MFMA conv kernel as a 1x1 conv over H=W=1."""
    __slots__ = ("wpk", "shift", "cout")

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], bn: Optional[nn.Module], dtype: torch.dtype):
        w = weight.detach().float()
        if bn is not None:
            scale, shift = _bn_scale_shift(bn, bias)
            w = w * scale.view(-1, 1)
        else:
            shift = bias.detach().float() if bias is not None else torch.zeros(w.shape[0], device=w.device)
        self.cout = w.shape[0]
        self.wpk = ops.pack_conv_weight(w.reshape(w.shape[0], w.shape[1], 1, 1).contiguous(), dtype)
        self.shift = shift.contiguous()

    def __call__(self, x2d: torch.Tensor, relu, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        return ops.linear_mfma(x2d, self.wpk, self.shift, self.cout, relu, residual)


# --------------------------------------------------------------------------------------------
# base class: plan cache + input checks
# --------------------------------------------------------------------------------------------
class _HipModule(nn.Module):
    """nn.Module whose inference runs on the HIP kernels.  The packed-weight "plan" is rebuilt
    whenever a parameter/buffer is replaced, moved or modified in place (tensor version counters)."""

    def __init__(self):
        super().__init__()
        self._plan = None
        self._plan_sig = None
        self.compute_dtype = _DEFAULT_DTYPE
        # uint8 HWC inputs (B×H×W×3 RGB, what `transforms.Resize` leaves, `src/testing.py:99-100`) are normalised on the
        # device: ToTensor + Normalize(mean, std) of `src/testing.py:101-104` (ImageNet statistics by default)
        self.input_mean, self.input_std = IMAGENET_MEAN, IMAGENET_STD

    supports_u8_input = True

    def set_input_normalization(self, mean, std):
        self.input_mean, self.input_std = tuple(float(v) for v in mean), tuple(float(v) for v in std)
        return self

    def set_compute_dtype(self, dtype: torch.dtype):
        ops.dt_code(dtype)
        self.compute_dtype = dtype
        self._plan = None
        return self

    def _signature(self):
        sig = [self.compute_dtype, self.input_mean, self.input_std]
        for t in list(self.parameters()) + list(self.buffers()):
            sig.append((t.data_ptr(), t._version))
        return tuple(sig)

    def _get_plan(self):
        sig = self._signature()
        if self._plan is None or sig != self._plan_sig:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError(f"{type(self).__name__} is on {dev}; move it to the GPU (.to('cuda')) — "
                                   "the HIP path has no CPU fallback")
            with torch.no_grad():
                self._plan = self._build_plan(self.compute_dtype)
            self._plan_sig = self._signature()
        return self._plan

    def _build_plan(self, dtype):  # pragma: no cover - abstract
        raise NotImplementedError

    def _check_input(self, x: torch.Tensor) -> torch.Tensor:
        """fp32 NCHW B×3×H×W (the reference's tensor) or uint8 HWC B×H×W×3 (the image bytes; normalised on the device)."""
        u8 = isinstance(x, torch.Tensor) and x.dtype == torch.uint8
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or (x.shape[3] != 3 if u8 else x.shape[1] != 3):
            raise ValueError(f"expected a B×3×H×W float tensor or a B×H×W×3 uint8 tensor, got "
                             f"{tuple(x.shape) if isinstance(x, torch.Tensor) else type(x)}")
        if not x.is_cuda:
            raise RuntimeError("input is on the CPU; the HIP path needs GPU tensors (no CPU fallback)")
        if self.training:
            raise NotImplementedError(f"{type(self).__name__}: only eval-mode inference is implemented on the HIP path; "
                                      "call .eval() (training is out of scope, SURVEY.md §2.1)")
        if u8:
            return x
        return x.float() if x.dtype != torch.float32 else x

    def _as_nhwc4(self, x: torch.Tensor) -> torch.Tensor:
        """First-layer operand of the unfused paths: NHWC4 in the compute dtype from either input kind."""
        if x.dtype == torch.uint8:
            return ops.normalize_u8(x, self.input_mean, self.input_std, want_nchw=False, nhwc4_dtype=self.compute_dtype)[1]
        return ops.pack_input(x, self.compute_dtype)


# --------------------------------------------------------------------------------------------
# ResNet-18 parameter container with torchvision's attribute names / order
# (torchvision is un-vendored; call sites face_models.py:67,269,463,658)
# --------------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: Optional[nn.Module] = None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError("BasicBlock is a parameter container; inference runs through the owning HIP module")


class ResNet18(nn.Module):
    """Parameter container: ``conv1 bn1 relu maxpool layer1..4 avgpool fc`` (children order matters —
    the reference slices ``children()[:-1]`` / ``[:-2]``, `face_models.py:100,464,660`)."""

    def __init__(self, num_classes: int = 1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inpl = 64
        for i, (planes, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2)), start=1):
            ds = None
            if stride != 1 or inpl != planes:
                ds = nn.Sequential(nn.Conv2d(inpl, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
            setattr(self, f"layer{i}", nn.Sequential(BasicBlock(inpl, planes, stride, ds), BasicBlock(planes, planes)))
            inpl = planes
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)

    def forward(self, x):
        raise RuntimeError("ResNet18 is a parameter container; inference runs through the owning HIP module")


class _TrunkPlan:
    """ResNet-18 trunk (and, for 'cnn' / 'arcface', the heads) on a MODEL HANDLE of the C ABI (`frmap_model_*`,
    csrc/model_api.cpp): BatchNorm folding, weight packing and the per-layer kernel plan live in the library; this class only
    hands over the tensors under the reference's ``state_dict`` key names and asks for outputs."""

    def __init__(self, model_type: str, state: Dict[str, torch.Tensor], num_classes: int, dtype: torch.dtype,
                 mean=IMAGENET_MEAN, std=IMAGENET_STD):
        self.dtype = dtype
        self.handle = ops.ModelHandle(model_type, state, num_classes, dtype, mean, std)

    def features(self, x: torch.Tensor) -> torch.Tensor:
        """fp32 NCHW (or uint8 HWC) → NHWC B×7×7×512 (for 224² input) in the compute dtype."""
        return self.handle.forward(x, ops.OUT_TRUNK_MAP)

    def pooled(self, x: torch.Tensor) -> torch.Tensor:
        return self.handle.forward(x, ops.OUT_POOLED)  # fp32 B×512


def _trunk_plan(rn: "ResNet18", dtype: torch.dtype, mean, std):
    """The bare trunk of HybridNet / AttentionNet: a 'resnet18_trunk' handle, or the Python planner under FRMAP_PY_PLAN=1."""
    if _PY_PLAN:
        return _PyTrunkPlan(rn, dtype, mean, std)
    return _TrunkPlan("resnet18_trunk", dict(rn.state_dict()), 0, dtype, mean, std)


class _PyTrunkPlan:
    """The same trunk planned in Python, one C-ABI op per call (`FRMAP_PY_PLAN=1`): the A/B twin of the model handle - same
    launches on the same folded weights, bit-identical outputs (tests/test_model_cabi_gpu.py) - and the path per-op tools wrap."""

    def __init__(self, rn: ResNet18, dtype: torch.dtype, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        self.dtype = dtype
        self.mean, self.std = mean, std
        self.stem = _PackedConv(rn.conv1, rn.bn1, dtype)
        self.blocks = []
        for li in range(1, 5):
            for blk in getattr(rn, f"layer{li}"):
                ds = _PackedConv(blk.downsample[0], blk.downsample[1], dtype) if blk.downsample is not None else None
                c2 = _PackedConv(blk.conv2, blk.bn2, dtype)
                # (shift of the fused conv2 + shortcut launch: both folded BatchNorm shifts)
                fshift = (c2.shift + ds.shift).contiguous() if ds is not None else None
                self.blocks.append((_PackedConv(blk.conv1, blk.bn1, dtype), c2, ds, fshift))

    def features(self, x: torch.Tensor) -> torch.Tensor:
        """fp32 NCHW (or uint8 HWC) → NHWC B×7×7×512 (for 224² input) in the compute dtype."""
        if x.dtype == torch.uint8:
            H, W = x.shape[1], x.shape[2]
            if ops.stem_pool_dims(H, W)[1] <= 56 and W % 4 == 0:
                # the fused stem reads the image bytes; ToTensor + Normalize happen while its rows are staged
                x = ops.stem7x7_maxpool_u8(x, self.stem.wpk, self.stem.shift, self.mean, self.std, self.dtype)
            else:
                x4 = ops.normalize_u8(x, self.mean, self.std, want_nchw=False, nhwc4_dtype=self.dtype)[1]
                x = ops.maxpool(self.stem(x4, relu=True), 3, 2, 1)
        elif ops.stem_pool_dims(x.shape[2], x.shape[3])[1] <= 56:
            x = ops.stem7x7_maxpool(x, self.stem.wpk, self.stem.shift, self.dtype)   # fused stem, one kernel
        else:  # wider than the fused kernel's 8 column strips
            x = ops.maxpool(self.stem(ops.pack_input(x, self.dtype), relu=True), 3, 2, 1)
        for c1, c2, ds, fshift in self.blocks:
            if ds is None:
                x = c2(c1(x, relu=True), relu=True, residual=x)
                continue
            h = c1(x, relu=True)
            B, Hh, Wh, Ch = h.shape
            if ds.k == 1 and ops.conv_ds_supported(B, Hh, Wh, Ch, c2.cout, x.shape[1], x.shape[2], x.shape[3], ds.stride):
                # conv2 + bn2 + projection shortcut + add + ReLU in one launch (the shortcut as extra K stages)
                x = ops.conv_igemm_ds(h, c2.wpk, fshift, c2.cout, x, ds.wpk, ds.stride, True)
            else:
                x = c2(h, relu=True, residual=ds(x, relu=False))
        return x

    def pooled(self, x: torch.Tensor) -> torch.Tensor:
        return ops.avgpool_global(self.features(x))  # fp32 B×512


# --------------------------------------------------------------------------------------------
# a2  BaselineNet
# --------------------------------------------------------------------------------------------
class BaselineNet(_HipModule):
    """`face_models.py:16-60`."""

    def __init__(self, num_classes: int = 18, input_size: Tuple[int, int] = (224, 224)):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 32, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(32)
        self.conv2 = nn.Conv2d(32, 64, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(64)
        self.conv3 = nn.Conv2d(64, 128, 3, padding=1)
        self.bn3 = nn.BatchNorm2d(128)
        self.pool = nn.MaxPool2d(2, 2)
        self.adaptive_pool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Linear(128, 512)
        self.fc2 = nn.Linear(512, num_classes)
        self.dropout = nn.Dropout(0.5)

    def _build_plan(self, dtype):
        if not _PY_PLAN and _HEAD_FUSE and _POOL_FUSE:
            return _TrunkPlan("baseline", dict(self.state_dict()), self.fc2.out_features, dtype, self.input_mean, self.input_std)
        return {"c1": _PackedConv(self.conv1, self.bn1, dtype), "c2": _PackedConv(self.conv2, self.bn2, dtype),
                "c3": _PackedConv(self.conv3, self.bn3, dtype), "fc1_t": self.fc1.weight.detach().float().t().contiguous()}

    def model_handle(self):
        p = self._get_plan()
        return p.handle if isinstance(p, _TrunkPlan) else None

    def _embed(self, x):
        """-> (embedding as the reference returns it, its unit-norm copy or None)."""
        x = self._check_input(x)
        p = self._get_plan()
        if isinstance(p, _TrunkPlan):
            return p.handle.forward(x, ops.OUT_EMBEDDING), None
        x = self._as_nhwc4(x)
        x = p["c1"].pooled(x)      # self.pool(F.relu(self.bn1(self.conv1(x)))), `face_models.py:38`
        x = p["c2"].pooled(x)
        x = p["c3"].pooled(x)
        if _HEAD_FUSE:   # adaptive_pool + fc1 + ReLU (`face_models.py:41-46`) in one launch, the unit-norm copy for the matcher with it
            emb, pre = ops.gap_linear_norm(x, p["fc1_t"], None, self.fc1.bias.detach(), 1e-12, want_pre=True, relu=True)
            return pre, emb
        f = ops.avgpool_global(x)
        return ops.linear_f32(f, self.fc1.weight.detach(), None, self.fc1.bias.detach(), relu=True), None

    def get_embedding(self, x):
        return self._embed(x)[0]

    def unit_embedding(self, x):
        """``F.normalize(get_embedding(x))`` (what `embed_and_match(normalize=True)` matches on) without a second launch."""
        pre, emb = self._embed(x)
        return emb if emb is not None else ops.l2_normalize(pre, 1e-12)

    def forward(self, x):
        h = self.model_handle()
        if h is not None:
            return h.forward(self._check_input(x), ops.OUT_LOGITS)
        e = self.get_embedding(x)
        return ops.linear_f32(e, self.fc2.weight.detach(), None, self.fc2.bias.detach())


# --------------------------------------------------------------------------------------------
# a3  ResNetTransfer ('cnn')
# --------------------------------------------------------------------------------------------
class ResNetTransfer(_HipModule):
    """`face_models.py:62-102`.  No pretrained-weight fetch (offline); load a checkpoint instead."""

    def __init__(self, num_classes: int = 18, freeze_backbone: bool = False):
        super().__init__()
        self.resnet = ResNet18()
        in_feats = self.resnet.fc.in_features
        self.dropout = nn.Dropout(0.1)
        self.resnet.fc = nn.Sequential(self.dropout, nn.Linear(in_feats, num_classes))
        if freeze_backbone:
            self._freeze_backbone()

    def _freeze_backbone(self):
        for name, param in self.resnet.named_parameters():
            if "fc" not in name:
                param.requires_grad = False

    def unfreeze_backbone(self):
        for param in self.resnet.parameters():
            param.requires_grad = True

    def _build_plan(self, dtype):
        if _PY_PLAN:
            return _PyTrunkPlan(self.resnet, dtype, self.input_mean, self.input_std)
        fc = self.resnet.fc[1]
        return _TrunkPlan("cnn", {f"resnet.{k}": v for k, v in self.resnet.state_dict().items()}, fc.out_features, dtype,
                          self.input_mean, self.input_std)

    def model_handle(self):
        """The `frmap_model` behind this module (None under FRMAP_PY_PLAN=1): `matching.embed_and_match` runs on it in one call."""
        p = self._get_plan()
        return p.handle if isinstance(p, _TrunkPlan) else None

    def forward(self, x):
        x = self._check_input(x)
        p = self._get_plan()
        if isinstance(p, _TrunkPlan):
            return p.handle.forward(x, ops.OUT_LOGITS)
        fc = self.resnet.fc[1]
        return ops.linear_f32(p.pooled(x), fc.weight.detach(), None, fc.bias.detach())

    def trunk_map(self, x):
        """The NHWC trunk output whose global average pool is ``get_embedding`` (`face_models.py:98-102`); lets
        ``matching.embed_and_match`` pool, normalise and match in one kernel."""
        x = self._check_input(x)
        return self._get_plan().features(x)

    def get_embedding(self, x):
        x = self._check_input(x)
        return self._get_plan().pooled(x).squeeze()  # `.squeeze()` as the reference (`:102`): (512,) at B == 1


# --------------------------------------------------------------------------------------------
# a6  SiameseNet
# --------------------------------------------------------------------------------------------
class SiameseNet(_HipModule):
    """`face_models.py:104-192`."""

    def __init__(self):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.MaxPool2d(2, stride=2),
            nn.Conv2d(64, 128, kernel_size=3, padding=1), nn.BatchNorm2d(128), nn.ReLU(inplace=True),
            nn.Conv2d(128, 128, kernel_size=3, padding=1), nn.BatchNorm2d(128), nn.ReLU(inplace=True),
            nn.MaxPool2d(2, stride=2),
            nn.Conv2d(128, 256, kernel_size=3, padding=1), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
            nn.Conv2d(256, 256, kernel_size=3, padding=1), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
            nn.MaxPool2d(2, stride=2),
            nn.Conv2d(256, 512, kernel_size=3, padding=1), nn.BatchNorm2d(512), nn.ReLU(inplace=True),
            nn.AdaptiveAvgPool2d((6, 6)),
        )
        self.fc = nn.Sequential(
            nn.Dropout(0.3), nn.Linear(512 * 6 * 6, 1024), nn.BatchNorm1d(1024), nn.ReLU(inplace=True),
            nn.Dropout(0.2), nn.Linear(1024, 512), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
            nn.Linear(512, 256),
        )
        self.debug_shapes = {}

    def _build_plan(self, dtype):
        if not _PY_PLAN and _POOL_FUSE:
            return _TrunkPlan("siamese", dict(self.state_dict()), 0, dtype, self.input_mean, self.input_std)
        c = self.conv
        # fc.1 consumes the NCHW flatten (index c*36 + s, `face_models.py:171`); our pooled tensor is
        # NHWC (index s*512 + c): permute the weight's input axis once here.
        w1 = self.fc[1].weight.detach().float().view(1024, 512, 36).permute(0, 2, 1).reshape(1024, 36 * 512)
        return {
            "convs": [(_PackedConv(c[0], c[1], dtype), True), (_PackedConv(c[4], c[5], dtype), False),
                      (_PackedConv(c[7], c[8], dtype), True), (_PackedConv(c[11], c[12], dtype), False),
                      (_PackedConv(c[14], c[15], dtype), True), (_PackedConv(c[18], c[19], dtype), False)],
            "fc1": _PackedLinearAsConv(w1, self.fc[1].bias, self.fc[2], dtype),
            "fc2": _PackedLinearAsConv(self.fc[5].weight, self.fc[5].bias, self.fc[6], dtype),
            "fc3": _PackedLinearAsConv(self.fc[8].weight, self.fc[8].bias, None, dtype),
        }

    def model_handle(self):
        p = self._get_plan()
        return p.handle if isinstance(p, _TrunkPlan) else None

    def forward_one(self, x):
        x = self._check_input(x)
        p = self._get_plan()
        batch_size = x.size(0)
        self.debug_shapes["input"] = x.shape
        if isinstance(p, _TrunkPlan):   # the whole tower is one call on the model handle; the reference's shape log is static
            self.debug_shapes.update(after_conv=torch.Size((batch_size, 512, 6, 6)), flattened=torch.Size((batch_size, 512 * 6 * 6)),
                                     before_norm=torch.Size((batch_size, 256)))
            return p.handle.forward(x, ops.OUT_EMBEDDING)
        convs = p["convs"]
        u8 = x.dtype == torch.uint8
        Wi = x.shape[2] if u8 else x.shape[3]
        if ((Wi + 6 - 7) // 2 + 1) // 2 <= 64 and (not u8 or Wi % 4 == 0):   # fused conv.0-3: 7x7 conv + bias + BN + ReLU + MaxPool2d(2,2)
            if u8:
                x = ops.stem7x7_maxpool_u8(x, convs[0][0].wpk, convs[0][0].shift, self.input_mean, self.input_std,
                                           self.compute_dtype, pool3=False)
            else:
                x = ops.stem7x7_maxpool(x, convs[0][0].wpk, convs[0][0].shift, self.compute_dtype, pool3=False)
            convs = convs[1:]
        else:
            x = self._as_nhwc4(x)
        for conv, pool in convs:     # conv -> BN -> ReLU (-> MaxPool2d(2)), `face_models.py:121-146`
            x = conv.pooled(x) if pool else conv(x, relu=True)
        x = ops.avgpool_adaptive(x, 6, 6)  # NHWC B×6×6×512
        self.debug_shapes["after_conv"] = torch.Size((batch_size, 512, 6, 6))
        feats = x.view(batch_size, -1)
        self.debug_shapes["flattened"] = feats.shape
        feats = p["fc3"](p["fc2"](p["fc1"](feats, relu=True), relu=True), relu=False)
        self.debug_shapes["before_norm"] = feats.shape
        return ops.l2_normalize(ops.cast_to_f32(feats), 1e-12)

    def forward(self, x1, x2):
        return self.forward_one(x1), self.forward_one(x2)

    def get_embedding(self, x):
        return self.forward_one(x)

    def get_debug_info(self):
        return self.debug_shapes


# --------------------------------------------------------------------------------------------
# a5  ArcMarginProduct (eval)      a4  ArcFaceNet (eval)
# --------------------------------------------------------------------------------------------
class ArcMarginProduct(nn.Module):
    """`face_models.py:297-445`; only the eval-mode forward is implemented (margin = m, scale =
    min(s, 24), `:369,401-404`)."""

    def __init__(self, in_feats, out_feats, s=32.0, m=0.5, use_warm_up=True, easy_margin=False):
        super().__init__()
        self.in_feats, self.out_feats = in_feats, out_feats
        self.s, self.m, self.easy_margin = s, m, easy_margin
        self.use_warm_up = use_warm_up
        self.warm_up_epochs = 10
        self.margin_factor = 0.0
        self.scale_factor = 0.3
        self.current_epoch = 0
        self.weight = nn.Parameter(torch.empty(out_feats, in_feats))
        nn.init.xavier_normal_(self.weight, gain=math.sqrt(2))
        self.register_buffer('u', torch.zeros(1))
        self.max_cos_theta = 0.0
        self.min_cos_theta = 0.0
        self.easy_margin_used = False
        self._minmax = None

    def forward(self, input, label):
        if self.training:
            raise NotImplementedError("ArcMarginProduct: the training-mode margin schedule (face_models.py:336-348,"
                                      "404-420) is out of scope; call .eval()")
        if self.easy_margin:
            self.easy_margin_used = True
        out, mm = ops.arcmargin_eval(input.float(), self.weight.detach(), label, self.s, self.m, self.easy_margin,
                                     want_minmax=True)
        self._minmax = mm  # fetched lazily: the reference's two .item() syncs (`:358-360`) are not forced here
        return out

    def _sync_minmax(self):
        if self._minmax is not None:
            mx, mn = self._minmax.tolist()
            self.max_cos_theta, self.min_cos_theta = mx, mn
            self._minmax = None

    def update_epoch(self, epoch):
        self.current_epoch = epoch

    def get_margin_stats(self):
        self._sync_minmax()
        return {'margin_factor': self.margin_factor, 'scale_factor': self.scale_factor,
                'effective_margin': self.m * self.margin_factor, 'effective_scale': self.s * self.scale_factor,
                'max_cos_theta': self.max_cos_theta, 'min_cos_theta': self.min_cos_theta,
                'easy_margin_used': self.easy_margin_used if self.easy_margin else False}


class ArcFaceNet(_HipModule):
    """`face_models.py:447-613` (eval branch `:573-582`, ``get_embedding`` `:584-590`)."""

    def __init__(self, num_classes=18, dropout_rate=0.2, s=32.0, m=0.5, easy_margin=False):
        super().__init__()
        self.backbone = ResNet18()
        self.features = nn.Sequential(*list(self.backbone.children())[:-1])
        self.embedding = nn.Linear(512, 512, bias=False)
        self.bn = nn.BatchNorm1d(512, eps=1e-5)
        self.dropout = nn.Dropout(p=dropout_rate)
        self.arcface = ArcMarginProduct(512, num_classes, s=s, m=m, use_warm_up=True, easy_margin=easy_margin)
        self.last_grad_norm = 0.0
        self.max_grad_norm = 1.0
        self.current_epoch = 0
        self.phase = 1
        self.backbone_frozen = False
        self.val_classifier = nn.Linear(512, num_classes)
        nn.init.xavier_normal_(self.val_classifier.weight, gain=math.sqrt(2))

    def freeze_backbone(self):
        self.backbone_frozen = True
        self.phase = 1
        for param_name, param in self.named_parameters():
            if 'backbone' in param_name or 'features' in param_name:
                param.requires_grad = False

    def unfreeze_backbone(self):
        self.backbone_frozen = False
        self.phase = 2
        for param in self.parameters():
            param.requires_grad = True

    def set_max_grad_norm(self, max_norm):
        self.max_grad_norm = max_norm

    def _build_plan(self, dtype):
        scale, shift = _bn_scale_shift(self.bn)
        if _PY_PLAN or not _HEAD_FUSE:
            trunk = _PyTrunkPlan(self.backbone, dtype, self.input_mean, self.input_std) if _PY_PLAN else \
                _trunk_plan(self.backbone, dtype, self.input_mean, self.input_std)
        else:
            state = {f"backbone.{k}": v for k, v in self.backbone.state_dict().items()}
            state.update({"embedding.weight": self.embedding.weight, "val_classifier.weight": self.val_classifier.weight,
                          "val_classifier.bias": self.val_classifier.bias})
            state.update({f"bn.{k}": v for k, v in self.bn.state_dict().items()})
            trunk = _TrunkPlan("arcface", state, self.val_classifier.out_features, dtype, self.input_mean, self.input_std)
        return {"trunk": trunk, "bn_scale": scale, "bn_shift": shift,
                "wt": self.embedding.weight.detach().float().t().contiguous()}   # [K][N] for the fused head

    def model_handle(self):
        t = self._get_plan()["trunk"]
        return t.handle if isinstance(t, _TrunkPlan) and t.handle.model_type == "arcface" else None

    def _pre_norm(self, x):
        p = self._get_plan()
        f = p["trunk"].pooled(x)
        return ops.linear_f32(f, self.embedding.weight.detach(), p["bn_scale"], p["bn_shift"])

    def get_embedding(self, x):
        x = self._check_input(x)
        p = self._get_plan()
        h = self.model_handle()
        if h is not None:
            # avgpool + embedding + bn + F.normalize (`face_models.py:584-590`): the handle's head, one launch after the trunk
            return h.forward(x, ops.OUT_EMBEDDING)
        if _HEAD_FUSE and p["wt"].shape[1] in (256, 512):
            # avgpool + embedding + bn + F.normalize (`face_models.py:584-590`) in one launch
            return ops.gap_linear_norm(p["trunk"].features(x), p["wt"], p["bn_scale"], p["bn_shift"], 1e-12)[0]
        return ops.l2_normalize(self._pre_norm(x), 1e-12)

    def forward(self, x, labels=None):
        if self.training:
            if labels is None:
                raise ValueError("Labels must be provided during training")
            raise NotImplementedError("ArcFaceNet: the training branch (face_models.py:527-572) is out of scope")
        emb = self.get_embedding(x)
        # `face_models.py:576`: the classifier rows are re-normalised in place on every eval call
        if self.val_classifier.weight.is_cuda:
            self.val_classifier.weight.data.copy_(ops.l2_normalize(self.val_classifier.weight.data, 1e-12))
            if self._plan is not None:
                self._plan_sig = self._signature()   # the plan does not depend on the classifier rows: no rebuild for this write
        if labels is not None:
            return ops.linear_f32(emb, self.val_classifier.weight.detach(), None, self.val_classifier.bias.detach())
        return emb

    def update_epoch(self, epoch):
        self.current_epoch = epoch
        self.arcface.update_epoch(epoch)

    def get_arcface_stats(self):
        stats = self.arcface.get_margin_stats()
        stats.update(grad_norm=self.last_grad_norm, max_grad_norm=self.max_grad_norm, phase=self.phase,
                     backbone_frozen=self.backbone_frozen)
        return stats

    def get_training_phase(self):
        return {'phase': self.phase, 'backbone_frozen': self.backbone_frozen, 'epoch': self.current_epoch}


# --------------------------------------------------------------------------------------------
# a7  HybridNet (+ TransformerBlock)
# --------------------------------------------------------------------------------------------
class TransformerBlock(nn.Module):
    """`face_models.py:618-648` (parameter container)."""

    def __init__(self, embed_dim, num_heads=4, ff_dim=2048, dropout=0.1):
        super().__init__()
        self.attention = nn.MultiheadAttention(embed_dim, num_heads, dropout=dropout)
        self.norm1 = nn.LayerNorm(embed_dim)
        self.norm2 = nn.LayerNorm(embed_dim)
        self.ff = nn.Sequential(nn.Linear(embed_dim, ff_dim), nn.GELU(), nn.Dropout(dropout),
                                nn.Linear(ff_dim, embed_dim), nn.Dropout(dropout))

    def forward(self, x):
        raise RuntimeError("TransformerBlock is a parameter container; inference runs through HybridNet")


class HybridNet(_HipModule):
    """`face_models.py:650-721`."""

    def __init__(self, num_classes=18):
        super().__init__()
        self.cnn = ResNet18()
        self.features = nn.Sequential(*list(self.cnn.children())[:-2])
        self.fdim = 512
        self.seq_len = 49
        self.pos_encoding = nn.Parameter(torch.zeros(self.seq_len, 1, self.fdim))
        nn.init.normal_(self.pos_encoding, mean=0, std=0.02)
        self.transformer = TransformerBlock(self.fdim)
        self.dropout = nn.Dropout(0.1)
        self.norm = nn.LayerNorm(self.fdim)
        self.fc = nn.Linear(self.fdim, num_classes)

    def _build_plan(self, dtype):
        if not _PY_PLAN:
            return _TrunkPlan("hybrid", dict(self.state_dict()), self.fc.out_features, dtype, self.input_mean, self.input_std)
        tr = self.transformer
        f32 = lambda t: t.detach().float().contiguous()
        return {
            "trunk": _trunk_plan(self.cnn, dtype, self.input_mean, self.input_std),
            "pos": f32(self.pos_encoding).view(self.seq_len, self.fdim),
            "n1": (f32(tr.norm1.weight), f32(tr.norm1.bias)), "n2": (f32(tr.norm2.weight), f32(tr.norm2.bias)),
            "nf": (f32(self.norm.weight), f32(self.norm.bias)),
            "qkv": _PackedLinearAsConv(tr.attention.in_proj_weight, tr.attention.in_proj_bias, None, dtype),
            "proj": _PackedLinearAsConv(tr.attention.out_proj.weight, tr.attention.out_proj.bias, None, dtype),
            "ff1": _PackedLinearAsConv(tr.ff[0].weight, tr.ff[0].bias, None, dtype),
            "ff2": _PackedLinearAsConv(tr.ff[3].weight, tr.ff[3].bias, None, dtype),
        }

    def get_embedding(self, x):
        """`face_models.py:705-721`: trunk (no pool) → +pos → pre-LN transformer block → token mean → LN."""
        x = self._check_input(x)
        p = self._get_plan()
        if isinstance(p, _TrunkPlan):
            try:
                return p.handle.forward(x, ops.OUT_EMBEDDING)
            except ValueError as e:   # (the handle rejects maps that are not 49 tokens with the same message)
                raise ValueError(str(e).split(": ", 1)[-1]) from None
        f = p["trunk"].features(x)                       # NHWC B×7×7×512 == tokens [B][49][512]
        B, Hh, Ww, D = f.shape
        L = Hh * Ww
        if L != self.seq_len:
            raise ValueError(f"HybridNet expects a {self.seq_len}-token feature map (224×224 input), got {L}")
        t, n1 = ops.add_pos_layernorm(f.view(B, L, D), p["pos"], *p["n1"], want_sum=True)
        qkv = p["qkv"](n1.view(B * L, D), relu=0)
        att = ops.mha_tokens(qkv.view(B, L, 3 * D), self.transformer.attention.num_heads)
        t2 = p["proj"](att.view(B * L, D), relu=0, residual=t.view(B * L, D))          # x + attn_out
        _, n2 = ops.add_pos_layernorm(t2.view(B, L, D), None, *p["n2"])
        hdn = p["ff1"](n2.view(B * L, D), relu=2)                                       # Linear → GELU
        t3 = p["ff2"](hdn, relu=0, residual=t2)                                          # x + ff_out
        return ops.mean_layernorm(t3.view(B, L, D), *p["nf"])

    def model_handle(self):
        p = self._get_plan()
        return p.handle if isinstance(p, _TrunkPlan) else None

    def forward(self, x):
        h = self.model_handle()
        if h is not None:
            return h.forward(self._check_input(x), ops.OUT_LOGITS)
        e = self.get_embedding(x)
        return ops.linear_f32(e, self.fc.weight.detach(), None, self.fc.bias.detach())


# --------------------------------------------------------------------------------------------
# §8(f) AttentionNet (face_models.py:194-295) and EnsembleModel (face_models.py:843-959)
# --------------------------------------------------------------------------------------------
class SpatialAttention(nn.Module):
    """Parameter container, `face_models.py:194-211` (the gate itself runs inside ``frmap_cnn_attention``)."""

    def __init__(self, kernel_size=7):
        super().__init__()
        self.conv = nn.Conv2d(2, 1, kernel_size=kernel_size, padding=kernel_size // 2)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        raise RuntimeError("SpatialAttention is a parameter container here; run AttentionNet on the GPU")


class AttentionModule(nn.Module):
    """Parameter container, `face_models.py:213-262`."""

    def __init__(self, in_channels, reduction_ratio=8):
        super().__init__()
        self.query = nn.Conv2d(in_channels, in_channels // reduction_ratio, kernel_size=1)
        self.key = nn.Conv2d(in_channels, in_channels // reduction_ratio, kernel_size=1)
        self.value = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.gamma_value = 0.0
        self.num_heads = 2
        self.head_dim = in_channels // (reduction_ratio * self.num_heads)
        self.spatial_attention = SpatialAttention()

    def forward(self, x):
        raise RuntimeError("AttentionModule is a parameter container here; run AttentionNet on the GPU")


class AttentionNet(_HipModule):
    """`face_models.py:264-295`: ResNet-18 trunk (no pool) → AttentionModule → global average pool → fc.
    q/k/v are ONE 1x1 conv (the three weight matrices stacked along Cout, biases as its shift); everything after it
    up to and including the pooling is ``frmap_cnn_attention``."""

    def __init__(self, num_classes=18, dropout_rate=0.25):
        super().__init__()
        self.backbone = ResNet18()
        self.features = nn.Sequential(*list(self.backbone.children())[:-2])
        self.attention = AttentionModule(512)
        self.gap = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512, num_classes)

    def _build_plan(self, dtype):
        a = self.attention
        w = torch.cat([a.query.weight, a.key.weight, a.value.weight], dim=0).detach().float()
        bias = torch.cat([a.query.bias, a.key.bias, a.value.bias], dim=0).detach().float()
        return {"trunk": _trunk_plan(self.backbone, dtype, self.input_mean, self.input_std), "qkv": ops.pack_conv_weight(w.contiguous(), dtype),
                "qkv_bias": bias.contiguous(), "cq": a.query.weight.shape[0], "cqkv": w.shape[0],
                "gamma": a.gamma.detach().float().contiguous(),
                "sw": a.spatial_attention.conv.weight.detach().float().contiguous(),
                "sb": a.spatial_attention.conv.bias.detach().float().contiguous()}

    def _attend(self, x, want_map):
        x = self._check_input(x)
        p = self._get_plan()
        f = p["trunk"].features(x)                                                   # NHWC B×H×W×512
        qkv = ops.conv_igemm(f, p["qkv"], p["qkv_bias"], p["cqkv"], 1, 1, 0, 0)        # q | k | v  (+bias)
        return ops.cnn_attention(qkv, f, p["gamma"], p["sw"], p["sb"], p["cq"], want_map=want_map, want_pool=True)

    def get_embedding(self, x):
        """`face_models.py:284-288`."""
        return self._attend(x, False)[1]

    def forward(self, x):
        """`face_models.py:276-282`."""
        return ops.linear_f32(self.get_embedding(x), self.fc.weight.detach(), None, self.fc.bias.detach())

    def attention_map(self, x):
        """The attended map the module hands to the pooling, NCHW-logical (stored NHWC): B×H×W×512."""
        return self._attend(x, True)[0]

    def get_attention_params(self):
        self.attention.gamma_value = float(self.attention.gamma.detach().cpu().item())
        return {"gamma": self.attention.gamma_value}


class EnsembleModel(nn.Module):
    """`face_models.py:843-941`: runs every member on the GPU and merges their logits on the device.
    ArcFace members contribute cosine logits against their class centres (`:889-893`), Siamese members are
    skipped (`:894-897`); 'average' / 'weighted' / 'max' as in the reference, anything else raises ValueError
    at call time (`:919-920` — the constructor accepts 'attention' but ``forward`` has no branch for it)."""

    def __init__(self, models, ensemble_method: str = 'weighted'):
        super().__init__()
        self.models = nn.ModuleList(models)
        self.ensemble_method = ensemble_method
        self.weights = nn.Parameter(torch.ones(len(models)) / len(models),
                                    requires_grad=(ensemble_method in ['weighted', 'attention']))
        if ensemble_method == 'attention':
            self.attention_net = nn.Sequential(nn.Linear(len(models), 64), nn.ReLU(inplace=True),
                                               nn.Linear(64, len(models)), nn.Softmax(dim=0))

    def forward(self, x):
        outputs = []
        for model in self.models:
            if hasattr(model, 'training') and model.training:
                model.eval()
            if isinstance(model, ArcFaceNet):
                emb = model(x)                                                            # unit-norm embedding
                outputs.append(ops.cosine_logits(emb, model.arcface.weight.detach().float(), want_argmax=False)[0])
            elif isinstance(model, SiameseNet):
                continue
            else:
                outputs.append(model(x))
        if len(outputs) == 1:
            return outputs[0]
        if self.ensemble_method == 'average':
            return torch.mean(torch.stack(outputs), dim=0)
        elif self.ensemble_method == 'weighted':
            w = torch.softmax(self.weights.detach().float(), dim=0)
            return torch.sum(torch.stack([w[i] * outputs[i] for i in range(len(outputs))]), dim=0)
        elif self.ensemble_method == 'max':
            probs = [torch.softmax(o, dim=1) for o in outputs]
            return torch.log(torch.max(torch.stack(probs), dim=0)[0])
        raise ValueError(f"Unknown ensemble method: {self.ensemble_method}")

    def get_embedding(self, x):
        """`face_models.py:922-941`: the members' embeddings concatenated along the feature axis."""
        embs = [m.get_embedding(x) for m in self.models if hasattr(m, 'get_embedding')]
        embs = [e.unsqueeze(0) if e.dim() == 1 else e for e in embs]
        if len(embs) > 1:
            return torch.cat(embs, dim=1)
        return embs[0] if embs else None


def create_ensemble(model_types, num_classes: int, ensemble_method: str = 'average') -> EnsembleModel:
    """`face_models.py:943-959`."""
    return EnsembleModel([get_model(t, num_classes=num_classes) for t in model_types], ensemble_method=ensemble_method)


# --------------------------------------------------------------------------------------------
# a1  factory
# --------------------------------------------------------------------------------------------
def get_model(model_type: str, num_classes: int = 18, input_size: Tuple[int, int] = (224, 224)) -> nn.Module:
    """`face_models.py:785-813`: same type strings; unknown → ``ValueError``.  Returns a train-mode
    module on the CPU with fp32 parameters, as the reference does; ``.to('cuda').eval()`` before
    inference."""
    if model_type == 'baseline':
        return BaselineNet(num_classes=num_classes, input_size=input_size)
    elif model_type == 'cnn':
        return ResNetTransfer(num_classes=num_classes, freeze_backbone=False)
    elif model_type == 'siamese':
        return SiameseNet()
    elif model_type == 'arcface':
        return ArcFaceNet(num_classes=num_classes, dropout_rate=0.2)
    elif model_type == 'hybrid':
        return HybridNet(num_classes=num_classes)
    elif model_type == 'attention':
        return AttentionNet(num_classes=num_classes, dropout_rate=0.25)
    elif model_type == 'ensemble':
        return create_ensemble(['cnn', 'attention', 'arcface'], num_classes=num_classes)
    elif isinstance(model_type, list):
        return create_ensemble(model_type, num_classes=num_classes)
    else:
        raise ValueError(f"Invalid model type: {model_type}")
