#!/usr/bin/env python3
"""Micro-benchmark of the fp32 heads: ArcMargin eval / cosine top-1 / distance top-1 (config 3 shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frmap_amd import ops
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, C in ((1024, 1000), (1024, 10000), (256, 36)):
    x = torch.randn(B, 512, device="cuda"); w = torch.randn(C, 512, device="cuda")
    lab = torch.randint(0, C, (B,), device="cuda")
    g = torch.nn.functional.normalize(w, dim=1); e = torch.nn.functional.normalize(x, dim=1)
    a = t(lambda: ops.arcmargin_eval(x, w, lab, 30.0, 0.5))
    c = t(lambda: ops.cosine_logits(x, w, 1.0, want_logits=False))
    m = t(lambda: ops.match_top1(e, g))
    prep = ops.match_prepare(g) if C >= ops.MATCH_MFMA_MIN_G else None
    mp = t(lambda: ops.match_top1(e, g, prepared=prep)) if prep is not None else float("nan")
    fl = 2.0 * B * C * 512
    print(f"B={B} C={C}: arcmargin_eval {a:7.1f} us ({fl/a/1e6:6.1f} TF)  cosine top-1 {c:7.1f} us  match_top1 {m:7.1f} us ({fl/m/1e6:6.1f} TF)  packed (fp16x3 MFMA) {mp:7.1f} us", flush=True)
