#!/usr/bin/env python3
"""Micro-benchmark of frmap_linear_mfma (Hybrid token GEMMs, Siamese fc) — HIP-event timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frmap_amd import ops
dt = torch.bfloat16
for name, M, K, N, act in [("hyb qkv", 12544, 512, 1536, 0), ("hyb proj", 12544, 512, 512, 0), ("hyb ff1", 12544, 512, 2048, 2),
                           ("hyb ff2", 12544, 2048, 512, 0), ("sia fc1", 256, 18432, 1024, 1), ("sia fc2", 256, 1024, 512, 1)]:
    x = torch.randn(M, K, device="cuda").to(dt)
    w = ops.pack_conv_weight((torch.randn(N, K, device="cuda") / K ** 0.5).view(N, K, 1, 1), dt)
    sh = torch.zeros(N, device="cuda")
    for _ in range(3): ops.linear_mfma(x, w, sh, N, act)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.linear_mfma(x, w, sh, N, act)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:10s} M={M} K={K} N={N}: {us:8.1f} us  {2.0 * M * K * N / us / 1e6:8.1f} TFLOP/s", flush=True)
