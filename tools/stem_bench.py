#!/usr/bin/env python3
"""Micro-benchmark of the fused stem (fp32 NCHW -> conv7x7 s2 + BN + ReLU + maxpool -> NHWC)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frmap_amd import ops

def main():
    dev, dt = "cuda", torch.bfloat16
    for B, H, pool3 in [(256, 224, True), (256, 200, False), (1024, 224, True)]:
        x = torch.randn(B, 3, H, H, device=dev)
        wpk = ops.pack_conv_weight_c3(torch.randn(64, 3, 7, 7, device=dev) * 0.1, dt)
        sh = torch.zeros(64, device=dev)
        for _ in range(3): ops.stem7x7_maxpool(x, wpk, sh, dt, pool3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): y = ops.stem7x7_maxpool(x, wpk, sh, dt, pool3)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        byt = x.numel() * 4 + y.numel() * 2
        print(f"stem B={B} {H}x{H} pool3={pool3}: {us:8.1f} us  {byt / us / 1e6:6.2f} TB/s (algorithmic bytes)", flush=True)
        if H % 4 == 0:   # the uint8-HWC variant (ToTensor + Normalize inside the stem)
            x8 = torch.randint(0, 256, (B, H, H, 3), device=dev, dtype=torch.uint8)
            mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
            for _ in range(3): ops.stem7x7_maxpool_u8(x8, wpk, sh, mean, std, dt, pool3)
            e0.record()
            for _ in range(20): y = ops.stem7x7_maxpool_u8(x8, wpk, sh, mean, std, dt, pool3)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            byt = x8.numel() + y.numel() * 2
            print(f"  uint8 HWC input            : {us:8.1f} us  {byt / us / 1e6:6.2f} TB/s (algorithmic bytes)", flush=True)

if __name__ == "__main__":
    main()
