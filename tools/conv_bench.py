#!/usr/bin/env python3
"""Micro-benchmark of frmap_conv_igemm on the ResNet-18 layer shapes (HIP-event timed)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frmap_amd import ops

SHAPES = [  # name, H, Cin, Cout, k, stride, residual
    ("l1 56x56 64->64", 56, 64, 64, 3, 1, False), ("l1 +res", 56, 64, 64, 3, 1, True),
    ("l2 28x28 128->128", 28, 128, 128, 3, 1, False), ("l3 14x14 256->256", 14, 256, 256, 3, 1, False),
    ("l4 7x7 512->512", 7, 512, 512, 3, 1, False),
    ("l2.0 s2 56->28 64->128", 56, 64, 128, 3, 2, False), ("l3.0 s2 128->256", 28, 128, 256, 3, 2, False),
    ("l4.0 s2 256->512", 14, 256, 512, 3, 2, False), ("ds 1x1 s2 64->128", 56, 64, 128, 1, 2, False),
]

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=256); ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--pp", type=int, default=-1, help="second-generation 3x3 s1 kernel: 1 force on (every Cin), 0 off, -1 default policy")
    ap.add_argument("--tile-px", type=int, default=-1); ap.add_argument("--bn", type=int, default=-1)
    a = ap.parse_args()
    from frmap_amd import _lib
    _lib.load().frmap_conv_pp_tuning(a.pp, a.tile_px, a.bn)
    dev = "cuda"; dt = torch.bfloat16
    for name, H, Cin, Cout, k, s, res in SHAPES:
        pad = 1 if k == 3 else 0
        x = torch.randn(a.batch, H, H, Cin, device=dev).to(dt)
        w = ops.pack_conv_weight(torch.randn(Cout, Cin, k, k, device=dev) * (2.0 / (Cin * k * k)) ** 0.5, dt)
        sh = torch.zeros(Cout, device=dev)
        Ho = (H + 2 * pad - k) // s + 1
        r = torch.randn(a.batch, Ho, Ho, Cout, device=dev).to(dt) if res else None
        for _ in range(3): ops.conv_igemm(x, w, sh, Cout, k, s, pad, True, r)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps): ops.conv_igemm(x, w, sh, Cout, k, s, pad, True, r)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.reps * 1e3
        fl = 2.0 * a.batch * Ho * Ho * Cout * Cin * k * k
        print(f"{name:28s} {us:8.1f} us  {fl / us / 1e6:8.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    main()
