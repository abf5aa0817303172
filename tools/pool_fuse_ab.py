#!/usr/bin/env python3
"""In-process A/B of the fused conv + MaxPool2d(2, 2) epilogue: BaselineNet and SiameseNet forward
at one batch size with `face_models._POOL_FUSE` on / off, HIP-event timed, plus the HBM bytes the conv + pool stages move
per face in each form (algorithmic: reads + writes of every map).

usage: python3 tools/pool_fuse_ab.py [--batch 256] [--models baseline,siamese] [--iters 30]
"""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frmap_amd import face_models as fm, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--models", default="baseline,siamese")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    x = synth.randn(5, (a.batch, 3, 224, 224), "x").cuda()
    for name in a.models.split(","):
        m = fm.get_model(name, 36)
        m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), 1001))
        m = m.cuda().eval()
        m.set_compute_dtype(dt)
        res = {}
        for fuse in (True, False, True, False):
            fm._POOL_FUSE = fuse
            f = (lambda: m.get_embedding(x))
            for _ in range(5):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(fuse, []).append(e0.elapsed_time(e1) / a.iters * 1e3)
        fm._POOL_FUSE = True
        on, off = min(res[True]), min(res[False])
        print(f"{name:9s} B={a.batch} {a.dtype}: fused {on:8.1f} us ({a.batch / on * 1e6:9.0f} faces/s)   two-launch {off:8.1f} us "
              f"({a.batch / off * 1e6:9.0f} faces/s)   x{off / on:.2f}", flush=True)


if __name__ == "__main__":
    main()
