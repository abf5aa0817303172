#!/usr/bin/env python3
"""Headline benchmark: faces/s for 224×224 embed + match (BASELINE.json), MI355X.

A "step" = one pass of the hot path over one batch of synthetic faces already resident in HBM
(default: the step is captured once into a HIP graph and replayed, the batch split into two
micro-batches on concurrent streams — `frmap_amd.GraphedEmbedMatch`; `--graph 0 --streams 1` runs
the same kernels as plain eager launches on one stream):
fp32 NCHW batch → ResNet-18 embedding (HIP kernels, bf16 MFMA) → L2-normalise → top-1 match against
a 36-ID gallery (configs[1]: "ResNet18 ('cnn') bf16 embed+match, batch 256, 1×MI355X, 36-ID
gallery").  With N GPUs every rank runs the same per-GPU batch (weak scaling: faces shard
embarrassingly) and one RCCL all-gather collates the (id, distance) pairs each step.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     — the dominant kernel (conv3x3_fast_kernel<BF16>: the 3×3 stride-1 implicit-GEMM
                 convolutions, 13 launches per full-batch forward): algorithmic FLOPs per launch ÷ its
                 average launch duration, measured with HIP events on the launch stream in a separate
                 instrumented pass after the timed region — standalone eager launches at the full per-GPU
                 batch on one stream (per-kernel durations are not defined for kernels that share the GPU
                 with another stream's, and rocprofv3 serialises the streams); the committed rocprofv3
                 summary is of `bench.py --graph 0 --streams 1`, the same launches.  peak = 2.5 PFLOP/s.
  cpu_baseline — the CPU oracle (oracle/face_oracle.py, fp32 PyTorch restatement of the reference's
                 forward) timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/fp16, MI355X_MICROARCH.md
DOMINANT = "conv3x3_fast_kernel<BF16, false>"


def _pmc_traffic(kernel, args, dtype):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_hbm_traffic_pmc.json: separate FETCH_SIZE / WRITE_SIZE runs of this same command,
    FETCH_SIZE doubled per the gfx950 correction).  PMC counters cannot be read from inside the
    process, so this is the stored measurement; None when the configuration differs from the profiled one."""
    if args.model != "cnn" or args.batch != 256 or dtype != torch.bfloat16:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic_pmc.json")) as f:
            return json.load(f)["kernels"][kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle-steps", type=int, default=150, dest="settle_steps",
                    help="untimed steady-state settling steps before the warm-up steps (clock / power-state ramp)")
    ap.add_argument("--batch", type=int, default=256, help="faces per GPU per step")
    ap.add_argument("--gallery", type=int, default=36)
    ap.add_argument("--model", default="cnn", choices=["cnn", "arcface", "baseline", "siamese", "hybrid", "attention"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--streams", type=int, default=2, help="split the per-GPU batch over this many concurrent HIP streams")
    ap.add_argument("--graph", type=int, default=1, help="1: replay the step from a captured HIP graph (0: eager launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # one rank per GPU; FRMAP_BENCH_BACKEND=gloo lets the N>1 control flow be rehearsed on a 1-GPU box
    backend = os.environ.get("FRMAP_BENCH_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import frmap_amd
    from frmap_amd import dist as fdist
    from frmap_amd import ops, synth

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    B, G = args.batch, args.gallery
    model = frmap_amd.get_model(args.model, 36)
    sd = synth.synth_state_dict(synth.shapes_of(model), 1002)
    model.load_state_dict(sd)
    model = model.to(dev).eval().set_compute_dtype(dtype)
    gallery = frmap_amd.Gallery([f"id{i}" for i in range(G)], synth.unit_rows(3002, G, 512 if args.model != "siamese" else 256), dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(2002 + rank)
    x = torch.randn((B, 3, 224, 224), device=dev, dtype=torch.float32, generator=gen)  # resident in HBM
    need_norm = args.model in ("cnn", "baseline", "hybrid", "attention")
    total = B * world

    graphed, graph_note = None, ""
    if args.graph:
        try:
            graphed = frmap_amd.GraphedEmbedMatch(model, gallery, x, 1.0, normalize=need_norm, streams=args.streams)
        except Exception as e:  # capture refused (e.g. another thread touched the device): same kernels, eager launches
            graph_note = f" (graph capture failed: {type(e).__name__}; eager launches)"
            torch.cuda.synchronize()
    side = [torch.cuda.Stream(device=dev) for _ in range(max(args.streams - 1, 0))]
    xs = list(x.chunk(args.streams)) if args.streams > 1 else [x]

    def local_step():
        if graphed is not None:
            graphed()
            return graphed.ids(), graphed.dists()
        if args.streams == 1:
            return frmap_amd.embed_and_match(model, x, gallery, 1.0, normalize=need_norm)
        # micro-batches on concurrent streams: one stream's tail waves run beside the other's full waves
        main = torch.cuda.current_stream()
        outs = [None] * args.streams
        for i, st in enumerate(side):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                outs[i + 1] = frmap_amd.embed_and_match(model, xs[i + 1], gallery, 1.0, normalize=need_norm)
        outs[0] = frmap_amd.embed_and_match(model, xs[0], gallery, 1.0, normalize=need_norm)
        for st in side:
            main.wait_stream(st)
        return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])

    # N > 1: the all-gather of step i (8 B/face, pure latency) runs on its own stream under step i+1's kernels;
    # the records are first copied out of the step's buffer (the next graph replay overwrites it)
    overlap = world > 1 and backend == "nccl" and os.environ.get("FRMAP_BENCH_OVERLAP_GATHER", "1") == "1"
    if overlap:
        comm = torch.cuda.Stream(device=dev)
        staging = [torch.empty((B, 2), dtype=torch.int32, device=dev) for _ in range(2)]
        gathered = [torch.empty((world * B, 2), dtype=torch.int32, device=dev) for _ in range(2)]
        ready_ev = [torch.cuda.Event() for _ in range(2)]
        copied_ev = [torch.cuda.Event() for _ in range(2)]
    state = {"n": 0}

    def step():
        if world == 1:
            return local_step()
        # the match kernel emits the 8-byte (id, distance) records; one all-gather collates them
        if not overlap:
            rec = graphed() if graphed is not None else frmap_amd.embed_and_match(model, x, gallery, 1.0, normalize=need_norm, packed=True)
            return fdist.gather_packed(rec)
        k = state["n"] & 1
        main = torch.cuda.current_stream()
        if state["n"] > 0:
            main.wait_event(copied_ev[k ^ 1])  # the previous step's records have left the buffer this step overwrites
        rec = graphed() if graphed is not None else frmap_amd.embed_and_match(model, x, gallery, 1.0, normalize=need_norm, packed=True)
        ready_ev[k].record(main)
        with torch.cuda.stream(comm):
            comm.wait_event(ready_ev[k])
            staging[k].copy_(rec)
            copied_ev[k].record(comm)
            dist.all_gather_into_tensor(gathered[k], staging[k])
        state["n"] += 1
        return gathered[k][:, 0], gathered[k].view(torch.float32)[:, 1]

    if overlap:
        # self-check on the live system: one overlapped step must return exactly what the plain gather returns
        with torch.no_grad():
            try:
                ids_o, d_o = step()
                torch.cuda.synchronize()
                rec0 = graphed() if graphed is not None else frmap_amd.embed_and_match(model, x, gallery, 1.0, normalize=need_norm, packed=True)
                ids_p, d_p = fdist.gather_packed(rec0)
                torch.cuda.synchronize()
                same = bool(torch.equal(ids_o, ids_p)) and bool(torch.equal(d_o, d_p))
            except Exception:
                same = False
            flag = torch.tensor([1 if same else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                overlap = False  # (every rank takes the same decision)
    with torch.no_grad():
        # settle: the first ~50 steps after start-up run 5-12 % slower (clocks / power state ramping up: 218 k faces/s
        # over steps 6-10, 248 k in steady state); run untimed settling steps before the W warm-up steps
        # (a fixed COUNT, not a wall time: every rank must issue the same number of collectives)
        for _ in range(max(args.settle_steps, 0)):
            step()
        torch.cuda.synchronize()
        for _ in range(max(args.warmup, 1) if args.warmup > 0 else 0):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    faces_per_s = total * args.steps / elapsed

    # ---------------- roofline of the dominant kernel (instrumented pass, not part of `value`) ----
    roofline = None
    if rank == 0 and not args.no_roofline:
        records = []
        orig = ops.conv_igemm

        def timed(x_, wpk, shift, Cout, k, stride, pad, relu, residual=None):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = orig(x_, wpk, shift, Cout, k, stride, pad, relu, residual)
            e1.record()
            Bn, H, W, Cin = x_.shape
            records.append((e0, e1, k, stride, 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * Cout * Cin * k * k, Cin))
            return y

        import frmap_amd.face_models as fm
        fm.ops.conv_igemm = timed
        try:
            with torch.no_grad():
                for _ in range(5):  # rank-local, eager, one stream, full batch
                    frmap_amd.embed_and_match(model, x, gallery, 1.0, normalize=need_norm)
            torch.cuda.synchronize()
        finally:
            fm.ops.conv_igemm = orig
        # the dominant kernel = conv3x3_fast_kernel: the 3x3 stride-1 convs with Cin >= 128 (Cin = 64 runs conv3x3_c64_wave_kernel)
        dom = [(e0.elapsed_time(e1) * 1e-3, fl) for e0, e1, k, s, fl, cin in records if k == 3 and s == 1 and cin >= 128]
        allc = [(e0.elapsed_time(e1) * 1e-3, fl) for e0, e1, k, s, fl, cin in records]
        if dom:
            tsum, fsum = sum(t for t, _ in dom), sum(f for _, f in dom)
            achieved = fsum / tsum / 1e12
            roofline = {"bound": "mfma", "kernel": DOMINANT if dtype == torch.bfloat16 else DOMINANT.replace("BF16", "F16"),
                        "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": _pmc_traffic(DOMINANT, args, dtype),
                        "measured": "standalone eager launches at the full per-GPU batch, one stream (instrumented pass)",
                        "launches_per_step": len(dom) // 5, "avg_launch_us": round(tsum / len(dom) * 1e6, 2),
                        "flop_per_launch": fsum / len(dom),
                        "all_conv_igemm_tflops": round(sum(f for _, f in allc) / sum(t for t, _ in allc) / 1e12, 2),
                        "conv_igemm_ms_per_step": round(sum(t for t, _ in allc) / 5 * 1e3, 3)}

    # ---------------- CPU baseline: the oracle on the host cores (bounded sample) -----------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import face_oracle as fo
        # the GPU box exposes 256 logical CPUs but a 1-GPU job owns a 16-core share; more threads
        # than that only oversubscribe the quota (measured: 256 threads -> 3.8 faces/s)
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        cores = max(1, min(avail, int(os.environ.get("FRMAP_CPU_THREADS", "16"))))
        torch.set_num_threads(cores)
        nb = 64
        xc = x[:nb].cpu()
        gal = gallery.matrix.cpu()
        emb_fn = {"cnn": fo.cnn_embedding, "arcface": fo.arcface_embedding, "baseline": fo.baseline_embedding,
                  "siamese": fo.siamese_forward_one, "hybrid": fo.hybrid_embedding,
                  "attention": fo.attention_embedding}[args.model]
        best = float("inf")
        with torch.no_grad():
            for rep in range(4):
                c0 = time.perf_counter()
                e = emb_fn(sd, xc)
                e = torch.nn.functional.normalize(e, dim=1) if need_norm else e
                fo.match_top1(e, gal)
                c1 = time.perf_counter()
                if rep > 0:
                    best = min(best, c1 - c0)
        cpu_baseline = {"value": round(nb / best, 1), "unit": "faces/s", "cores": cores, "kind": "port",
                        "sample": f"{nb} faces (same synthetic batch), fp32 torch-CPU oracle forward + L2-normalise + "
                                  f"{G}-ID match, best of 3 after 1 warm-up"}

    if rank == 0:
        line = {
            "metric": "faces/sec (224x224 embed+match)", "value": round(faces_per_s, 1), "unit": "faces/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"ResNet18 ('{args.model}') embed + L2-normalise + top-1 match, batch {B}/GPU, "
                                   f"{G}-ID gallery, 224x224x3 fp32 NCHW inputs resident in HBM, random-init weights",
                       "global_batch": total, "parallelism": f"dp{world} (faces sharded, 1 all-gather of 8 B/face" + (", overlapped with the next step" if overlap else "") + ")",
                       "execution": (f"HIP graph replay, {args.streams} concurrent micro-batch streams" if graphed is not None
                                     else f"eager launches, {args.streams} stream(s)" + graph_note)},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
