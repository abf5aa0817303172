"""TEST INFRASTRUCTURE (CPU oracle, never imported by the product): numpy restatement of Pillow's two-pass bilinear resize
for 8-bit RGB images - the arithmetic `torchvision.transforms.Resize((224, 224))` runs on a PIL image
(`/root/reference/src/testing.py:99-100`, `src/training.py:305-310`; torchvision's PIL backend calls
`Image.resize(size[::-1], BILINEAR)`).  Pillow is a third-party dependency of the reference (`requirements.txt`: Pillow,
unpinned), not part of /root/reference; this file restates its published algorithm (libImaging/Resample.c:
precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc) and is pinned against the
installed Pillow itself on random images (`tests/test_oracle_golden.py::test_pil_resize_restatement_equals_pillow`)."""
import math
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs with the bilinear filter (support 1.0) over the whole axis (box = (0, in_size)), then
    normalize_coeffs_8bpc.  Returns (bounds int32 [out, 2] = (xmin, xmax), coeffs int32 [out, ksize])."""
    in0, in1 = 0.0, float(in_size)
    scale = filterscale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.empty(xmax, np.float64)
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            if v < 0.0:
                v = -v
            w[x] = 1.0 - v if v < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
            p = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + p) if w[x] < 0 else int(0.5 + p)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v: np.ndarray) -> np.ndarray:
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """`Image.resize((out_w, out_h), BILINEAR)`: one ImagingResample call - except (Image.py, Pillow >= 11) for images more
    than 100 times taller than wide whose height shrinks, which are resized in height first, then in width."""
    H, W, _ = img.shape
    if H > W * 100 and out_h < H:
        return _resample(_resample(img, out_h, W), out_h, out_w)
    return _resample(img, out_h, out_w)


def _resample(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """HWC uint8 RGB -> [out_h, out_w, 3] as ImagingResample does: horizontal pass (only when the width changes) over the
    rows the vertical pass needs, then the vertical pass (only when the height changes); int32 accumulation from
    1 << (PRECISION_BITS - 1), arithmetic shift, clip to [0, 255]."""
    img = np.ascontiguousarray(img, np.uint8)
    H, W, _ = img.shape
    need_h, need_v = out_w != W, out_h != H
    cur = img
    if need_v:
        by, ky = precompute_coeffs(H, out_h)
        ybox_first, ybox_last = int(by[0, 0]), int(by[-1, 0] + by[-1, 1])
    if need_h:
        bx, kx = precompute_coeffs(W, out_w)
        rows = cur[ybox_first:ybox_last] if need_v else cur
        tmp = np.empty((rows.shape[0], out_w, 3), np.uint8)
        r32 = rows.astype(np.int64)
        for xx in range(out_w):
            xmin, xmax = bx[xx]
            acc = (r32[:, xmin:xmin + xmax, :] * kx[xx, :xmax].astype(np.int64)[None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            tmp[:, xx, :] = _clip8(acc)
        cur = tmp
        if need_v:
            by = by.copy()
            by[:, 0] -= ybox_first
    if need_v:
        out = np.empty((out_h, cur.shape[1], 3), np.uint8)
        c32 = cur.astype(np.int64)
        for yy in range(out_h):
            ymin, ymax = by[yy]
            acc = (c32[ymin:ymin + ymax] * ky[yy, :ymax].astype(np.int64)[:, None, None]).sum(0) + (1 << (PRECISION_BITS - 1))
            out[yy] = _clip8(acc)
        cur = out
    return cur
