"""Device-side `transforms.Resize(size)` for PIL-style 8-bit RGB images (`src/testing.py:99-100`, `src/training.py:305-310`,
`src/app.py:39`): bit-exact with Pillow's bilinear resampler.

Pillow resizes in two passes of small integer FIR filters (libImaging/Resample.c).  The filter tables depend only on the
(input size, output size) pair; they are built here on the host, in float64 with Pillow's operation order, cached per pair
and shipped with the batch; the integer arithmetic over the pixels (`frmap_resize_bilinear_u8`) runs on the GPU, so the
decoded images cross PCIe once, at their native size, and everything after the decode is on the device.
"""
from __future__ import annotations

import functools
import math
from typing import Sequence, Tuple

import numpy as np
import torch

from . import _lib

PRECISION_BITS = 32 - 8 - 2
ITEM_DTYPE = np.dtype([("src_off", "<u8"), ("H", "<i4"), ("W", "<i4"), ("bx_off", "<i4"), ("kx_off", "<i4"), ("ksx", "<i4"),
                       ("by_off", "<i4"), ("ky_off", "<i4"), ("ksy", "<i4")])   # FrmapResizeItem (csrc/resize.hip), 40 bytes


@functools.lru_cache(maxsize=512)
def bilinear_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Resample.c `precompute_coeffs` (bilinear, support 1, box = the whole axis) + `normalize_coeffs_8bpc`:
    ``(bounds int32 [out, 2] = (first input sample, taps), coeffs int32 [out, ksize])``.  float64, same operation order
    as the C code (the weights are summed left to right, divided by the sum, scaled by 2^22, rounded half away from zero)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    xx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    xmin = (center - support + 0.5).astype(np.int64)          # C cast: truncation toward zero
    xmin = np.maximum(xmin, 0)
    xmax = (center + support + 0.5).astype(np.int64)
    xmax = np.minimum(xmax, in_size) - xmin
    w = np.zeros((out_size, ksize), np.float64)
    ww = np.zeros(out_size, np.float64)
    for x in range(ksize):                                    # left-to-right accumulation, as the C loop
        v = np.abs((x + xmin - center + 0.5) * ss)
        wx = np.where(v < 1.0, 1.0 - v, 0.0)
        wx = np.where(x < xmax, wx, 0.0)
        w[:, x] = wx
        ww = np.where(x < xmax, ww + wx, ww)
    nz = ww != 0.0
    w[nz] = w[nz] / ww[nz, None]
    p = w * float(1 << PRECISION_BITS)
    kk = np.where(w < 0, (-0.5 + p), (0.5 + p)).astype(np.int64).astype(np.int32)   # (int) cast truncates
    kk[np.arange(ksize)[None, :] >= xmax[:, None]] = 0
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    bounds.setflags(write=False)
    kk.setflags(write=False)
    return bounds, kk


def resize_bilinear_u8(images: Sequence, size: Tuple[int, int] = (224, 224), device="cuda", _one_call: bool = False) -> torch.Tensor:
    """HWC uint8 RGB arrays / tensors of any sizes -> one uint8 ``[B, size[0], size[1], 3]`` tensor on ``device``,
    bit-identical to ``PIL.Image.fromarray(a).resize(size[::-1], Image.BILINEAR)`` per image."""
    out_h, out_w = int(size[0]), int(size[1])
    if len(images) == 0:
        return torch.empty((0, out_h, out_w, 3), dtype=torch.uint8, device=device)
    arrs = []
    for im in images:
        a = im.detach().cpu().numpy() if isinstance(im, torch.Tensor) else np.asarray(im)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3 or a.shape[0] < 1 or a.shape[1] < 1:
            raise ValueError("resize_bilinear_u8: expected H×W×3 uint8 RGB images")
        if not _one_call and a.shape[0] > a.shape[1] * 100 and out_h < a.shape[0]:
            # Pillow (Image.py, >= 11) resizes an image more than 100x taller than wide in height first, then in width:
            # two single-axis passes here too (each rounds to 8 bits, so the order is visible in the last bit)
            a = resize_bilinear_u8([a], (out_h, a.shape[1]), device, _one_call=True)[0].cpu().numpy()
        arrs.append(np.ascontiguousarray(a))
    items = np.zeros(len(arrs), ITEM_DTYPE)
    tables, table_off, off = [], {}, 0

    def table(in_size, out_size):
        nonlocal off
        key = (in_size, out_size)
        if key not in table_off:
            b, k = bilinear_coeffs(in_size, out_size)
            table_off[key] = (off, off + b.size, k.shape[1])
            tables.extend([b.reshape(-1), k.reshape(-1)])
            off += b.size + k.size
        return table_off[key]

    src_off, lds_rows, rows_per_block = 0, 1, 8
    needs = []
    for i, a in enumerate(arrs):
        H, W, _ = a.shape
        it = items[i]
        it["src_off"], it["H"], it["W"] = src_off, H, W
        src_off += a.size
        if W != out_w:
            it["bx_off"], it["kx_off"], it["ksx"] = table(W, out_w)
        if H != out_h:
            it["by_off"], it["ky_off"], it["ksy"] = table(H, out_h)
            needs.append(bilinear_coeffs(H, out_h)[0])
    # rows per workgroup: as many as keep every block's input-row window within 64 KB of LDS
    def window(rpb):
        m = rpb
        for b in needs:
            first = b[0::rpb, 0]
            last_idx = np.minimum(np.arange(0, out_h, rpb) + rpb, out_h) - 1
            m = max(m, int((b[last_idx, 0] + b[last_idx, 1] - first).max()))
        return m
    while rows_per_block > 1 and window(rows_per_block) * out_w * 4 > 64 * 1024:
        rows_per_block //= 2
    lds_rows = window(rows_per_block)
    pool = torch.from_numpy(np.concatenate([a.reshape(-1) for a in arrs])).to(device, non_blocking=True)
    items_d = torch.from_numpy(items.view(np.uint8).copy()).to(device, non_blocking=True)
    tab = np.concatenate(tables).astype(np.int32) if tables else np.zeros(1, np.int32)
    tab_d = torch.from_numpy(tab).to(device, non_blocking=True)
    out = torch.empty((len(arrs), out_h, out_w, 3), dtype=torch.uint8, device=pool.device)
    with torch.cuda.device(pool.device):
        _lib.check(_lib.load().frmap_resize_bilinear_u8(pool.data_ptr(), items_d.data_ptr(), tab_d.data_ptr(), out.data_ptr(),
                                                        len(arrs), out_h, out_w, rows_per_block, lds_rows,
                                                        torch.cuda.current_stream().cuda_stream), "resize_bilinear_u8")
    return out
