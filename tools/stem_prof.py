#!/usr/bin/env python3
"""One-shape driver for profiling the fused stem under rocprofv3 (B x 3 x 224 x 224 -> B x 56 x 56 x 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frmap_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(B, 3, 224, 224, device="cuda")
wpk = ops.pack_conv_weight_c3(torch.randn(64, 3, 7, 7, device="cuda") * 0.1, torch.bfloat16)
sh = torch.zeros(64, device="cuda")
for _ in range(12):
    ops.stem7x7_maxpool(x, wpk, sh, torch.bfloat16, True)
torch.cuda.synchronize()
