import os, sys
sys.path.insert(0, "/root/repo")
import torch
from frmap_amd import ops
dev, dt = "cuda", torch.bfloat16
wpk = ops.pack_conv_weight_c3(torch.randn(64, 3, 7, 7, device=dev) * 0.1, dt)
sh = torch.zeros(64, device=dev)
for B in (64, 128, 256, 512, 1024):
    x = torch.randn(B, 3, 224, 224, device=dev)
    for _ in range(5): ops.stem7x7_maxpool(x, wpk, sh, dt, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): y = ops.stem7x7_maxpool(x, wpk, sh, dt, True)
    e1.record(); torch.cuda.synchronize()
    print(f"B={B}: {e0.elapsed_time(e1)/30*1e3:8.1f} us", flush=True)
