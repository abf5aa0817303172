#!/usr/bin/env python3
"""In-process A/B of the two 1x1 / Linear kernels (conv1x1_pp_kernel vs conv1x1_kernel) on HybridNet's token-block shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frmap_amd import ops, _lib
lib = _lib.load()
dev, dt = "cuda", torch.bfloat16
for name, M, K, N in (("qkv 512->1536", 12544, 512, 1536), ("out 512->512", 12544, 512, 512), ("mlp 512->2048", 12544, 512, 2048),
                      ("mlp 2048->512", 12544, 2048, 512), ("attn-qkv conv 49px", 12544, 512, 640), ("tokens@1024 faces qkv", 50176, 512, 1536)):
    x = torch.randn(M, 1, 1, K, device=dev).to(dt)
    w = ops.pack_conv_weight(torch.randn(N, K, 1, 1, device=dev) * K ** -0.5, dt)
    sh = torch.zeros(N, device=dev)
    res = {}
    for rnd in range(3):
        for on in (1, 0):
            lib.frmap_conv_pp_tuning(on, -1, -1)
            for _ in range(3): ops.conv_igemm(x, w, sh, N, 1, 1, 0, 0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.conv_igemm(x, w, sh, N, 1, 1, 0, 0)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(on, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    lib.frmap_conv_pp_tuning(-1, -1, -1)
    fl = 2.0 * M * K * N
    print(f"{name:26s} layout {lib.frmap_conv1x1_pp_layout(M, 1, 1, K, N, 1)}  pp {min(res[1]):6.1f} us ({fl / min(res[1]) / 1e6:5.0f} TF)   gen1 {min(res[0]):6.1f} us ({fl / min(res[0]) / 1e6:5.0f} TF)", flush=True)
