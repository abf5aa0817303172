#!/usr/bin/env python3
"""Does layer 1 (HBM-bound 64-channel 56x56 convs) gain from keeping a BasicBlock's intermediate map in the 256 MB Infinity
Cache?  Runs conv1 -> conv2(+residual) over the whole batch, and over batch slices (conv1(s), conv2(s) per slice), one
process, HIP-event timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frmap_amd import ops

def main():
    B, dt, dev = int(sys.argv[1]) if len(sys.argv) > 1 else 256, torch.bfloat16, "cuda"
    x = torch.relu(torch.randn(B, 56, 56, 64, device=dev)).to(dt)
    w1 = ops.pack_conv_weight(torch.randn(64, 64, 3, 3, device=dev) * (2.0 / 576) ** 0.5, dt)
    w2 = ops.pack_conv_weight(torch.randn(64, 64, 3, 3, device=dev) * (2.0 / 576) ** 0.5, dt)
    sh = torch.zeros(64, device=dev)
    def block(xs):
        h = ops.conv_igemm(xs, w1, sh, 64, 3, 1, 1, True)
        return ops.conv_igemm(h, w2, sh, 64, 3, 1, 1, True, xs)
    def run(nsl):
        outs = [block(xs) for xs in x.chunk(nsl)]
        return outs
    res = {}
    for rnd in range(4):
        for nsl in (1, 2, 4, 8):
            for _ in range(3): run(nsl)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run(nsl)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(nsl, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    for nsl, t in res.items():
        print(f"B={B} slices={nsl}: block (conv1 + conv2+res) {min(t):7.1f} us")

if __name__ == "__main__":
    main()
