#!/usr/bin/env python3
"""A/B: two micro-batches on two streams in lockstep (the graph mode's schedule) vs STAGGERED (micro-batch B starts when A has finished
stem + layer 1, so B's memory-heavy front runs beside A's MFMA-heavy layers 2-4).  Per-op Python planner, eager launches."""
import os, sys
os.environ["FRMAP_PY_PLAN"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import frmap_amd
from frmap_amd import ops, synth
import frmap_amd.face_models as fm

dev = "cuda"
m = frmap_amd.get_model("cnn", 36)
m.load_state_dict(synth.calibrated_state_dict("cnn", synth.shapes_of(m), 1002))
m = m.to(dev).eval().set_compute_dtype(torch.bfloat16)
plan = m._get_plan()
gal = synth.unit_rows(3002, 36, 512).to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(B, 3, 224, 224, device=dev)
xa, xb = x[: B // 2], x[B // 2:]

def front(xx):
    h = ops.stem7x7_maxpool(xx, plan.stem.wpk, plan.stem.shift, plan.dtype)
    for c1, c2, ds, fs in plan.blocks[:2]:
        h = c2(c1(h, relu=True), relu=True, residual=h)
    return h

def back(h):
    for c1, c2, ds, fs in plan.blocks[2:]:
        if ds is None:
            h = c2(c1(h, relu=True), relu=True, residual=h)
        else:
            t = c1(h, relu=True)
            h = ops.conv_igemm_ds(t, c2.wpk, fs, c2.cout, h, ds.wpk, ds.stride, True)
    return ops.gap_norm_match(h, gal, 1.0, normalize=True)

sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def lockstep():
    main = torch.cuda.current_stream()
    sA.wait_stream(main); sB.wait_stream(main)
    with torch.cuda.stream(sA): back(front(xa))
    with torch.cuda.stream(sB): back(front(xb))
    main.wait_stream(sA); main.wait_stream(sB)
def staggered():
    main = torch.cuda.current_stream()
    sA.wait_stream(main); sB.wait_stream(main)
    with torch.cuda.stream(sA):
        h = front(xa)
        ev = torch.cuda.Event(); ev.record(sA)
    with torch.cuda.stream(sB):
        sB.wait_event(ev)
        hb = front(xb)
    with torch.cuda.stream(sA): back(h)
    with torch.cuda.stream(sB): back(hb)
    main.wait_stream(sA); main.wait_stream(sB)
def single():
    back(front(x))
def timeit(fn, n=60):
    with torch.no_grad():
        for _ in range(20): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(2):
    print(f"B={B}: single stream {timeit(single):7.1f} us | lockstep {timeit(lockstep):7.1f} us | staggered {timeit(staggered):7.1f} us", flush=True)
