#!/usr/bin/env python3
"""Micro-benchmark of the plain 3x3 stride-1 layer shapes on the second-generation kernel (conv_pp.hip)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frmap_amd import ops, _lib

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=256); ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--pp", type=int, default=1); ap.add_argument("--tile-px", type=int, default=-1); ap.add_argument("--bn", type=int, default=-1)
    ap.add_argument("--relu-data", type=int, default=1, help="1: non-negative half-sparse activations (post-ReLU statistics), 0: N(0,1)")
    ap.add_argument("--res", type=int, default=0); ap.add_argument("--variants", default="0,-1:0")
    ap.add_argument("--ds", type=int, default=0, help="1: the conv2 + projection-shortcut launches (frmap_conv_igemm_ds) instead of the plain layers")
    a = ap.parse_args()
    lib = _lib.load()
    dev, dt = "cuda", torch.bfloat16
    # settings to compare, "bn[:im]": bn = -1 heuristic, 128, 256, 1282 (split-K), 0 = first generation; im = 1 / 0: DMA issued in
    # the MFMA segments (default) / in the LOAD segments
    def _f(v, i, d=-1):
        parts = v.split(":")
        return int(parts[i]) if len(parts) > i else d
    # "bn[:im[:pitch[:ri]]]"; ri = 1: fragment reads interleaved with the MFMAs (conv3x3_pp_kernel<..., RI = true>)
    variants = [(_f(v, 0), _f(v, 1), _f(v, 2), _f(v, 3, 0)) for v in a.variants.split(",")]
    import ctypes
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name, H, C in (("l2 28x28 128", 28, 128), ("l3 14x14 256", 14, 256), ("l4 7x7 512", 7, 512)):
        x = torch.randn(a.batch, H, H, C, device=dev)
        if a.relu_data: x = torch.relu(x)
        x = x.to(dt)
        w = ops.pack_conv_weight(torch.randn(C, C, 3, 3, device=dev) * (2.0 / (C * 9)) ** 0.5, dt)
        sh = torch.zeros(C, device=dev)
        r = torch.relu(torch.randn(a.batch, H, H, C, device=dev)).to(dt) if a.res else None
        fl = 2.0 * a.batch * H * H * C * C * 9
        if a.ds:
            xd = torch.relu(torch.randn(a.batch, 2 * H, 2 * H, C // 2, device=dev)).to(dt)
            wd = ops.pack_conv_weight(torch.randn(C, C // 2, 1, 1, device=dev) * (2.0 / C) ** 0.5, dt)
            fl += 2.0 * a.batch * H * H * C * (C // 2)
            run = lambda: ops.conv_igemm_ds(x, w, sh, C, xd, wd, 2, True)
        else:
            run = lambda: ops.conv_igemm(x, w, sh, C, 3, 1, 1, True, r)
        res = {v: [] for v in variants}
        for rnd in range(4):           # interleaved rounds in one process (variants compared on the same device / clocks)
            for v in variants:
                lib.frmap_conv_pp_tuning(a.pp if v[0] != 0 else 0, a.tile_px, v[0])
                raw.frmap_conv_pp_im(v[1])
                raw.frmap_conv_pp_pitch(v[2])
                raw.frmap_conv_pp_ri(v[3])
                for _ in range(3): run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps): run()
                e1.record(); torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1) / a.reps * 1e3)
        print(f"{name:16s} " + "  ".join(f"bn{v[0]}/im{v[1]}/pitch{v[2]}/ri{v[3]}: {min(t):6.1f} us ({fl / min(t) / 1e6:5.0f} TF)" for v, t in res.items()), flush=True)

if __name__ == "__main__":
    main()
