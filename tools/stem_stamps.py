#!/usr/bin/env python3
"""In-kernel s_memtime breakdown of the fused stem (instrumented variant library: tools/variants/libfrmap_stamps.so, built from a
copy of csrc with stamps around the MFMA phase, the pooling epilogue, the ring commit and the barrier; FRMAP_LIB selects it)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FRMAP_LIB"] = os.path.join(ROOT, "tools", "variants", "libfrmap_stamps.so")
sys.path.insert(0, ROOT)
import torch
from frmap_amd import ops, _lib
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(B, 3, 224, 224, device="cuda")
wpk = ops.pack_conv_weight_c3(torch.randn(64, 3, 7, 7, device="cuda") * 0.1, torch.bfloat16)
sh = torch.zeros(64, device="cuda")
st = torch.zeros((512 * 4, 8), dtype=torch.int64, device="cuda")
f = lib.frmap_dbg_set_stamps; f.argtypes = [ctypes.c_void_p]
for _ in range(5): ops.stem7x7_maxpool(x, wpk, sh, torch.bfloat16, True)
f(st.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.stem7x7_maxpool(x, wpk, sh, torch.bfloat16, True); e1.record(); torch.cuda.synchronize()
s = st.cpu().double()
s = s[s[:, 5] > 0]
names = ["mfma", "epilogue", "commit", "barrier1", "barrier2", "total"]
print(f"B={B}: kernel {e0.elapsed_time(e1)*1e3:.1f} us; per-wave s_memtime ticks (100 MHz), median over {s.shape[0]} waves:")
for i, n in enumerate(names):
    print(f"  {n:9s} median {s[:, i].median():9.0f}  p10 {s[:, i].quantile(0.1):9.0f}  p90 {s[:, i].quantile(0.9):9.0f}")
