"""CPU: bench.py's multi-rank launcher (driven with a stand-in worker at world size 2 over gloo), its argument /
environment checks, and the committed BatchNorm statistics the benchmark's weights are built from."""
import json
import os
import subprocess
import sys
import textwrap

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import bench  # noqa: E402  (repo root is on sys.path via conftest)


WORKER = textwrap.dedent("""
    import json, os, sys, time
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
    mode = sys.argv[1]
    if mode == "fail" and rank == 1:
        sys.exit(3)
    if mode == "fail":
        time.sleep(120)          # must be terminated by the launcher, not waited for
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([rank + 1])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"n_gpus": dist.get_world_size(), "sum": int(t), "argv": sys.argv[1:]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
""")


def test_launcher_spawns_ranks_and_relays_rank0(tmp_path, capfd):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    rc = bench.launch_ranks(2, ["ok", "--gpus", "2"], worker_cmd=[sys.executable, str(w)], timeout=240)
    out = capfd.readouterr().out
    assert rc == 0, out
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                     # exactly one JSON line, from rank 0
    doc = json.loads(lines[0])
    assert doc == {"n_gpus": 2, "sum": 3, "argv": ["ok", "--gpus", "2"]}


def test_launcher_propagates_failure_and_stops_survivors(tmp_path, capfd):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    import time
    t0 = time.time()
    rc = bench.launch_ranks(2, ["fail"], worker_cmd=[sys.executable, str(w)], timeout=240)
    assert rc == 3
    assert time.time() - t0 < 60                               # rank 0's sleep(120) was cut short
    assert "rank 1 exited with 3" in capfd.readouterr().err


def test_gpus_flag_must_match_world_size(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit, match="--gpus 2 but WORLD_SIZE=4"):
        bench.worker(bench.parse(["--gpus", "2"]))


def test_launcher_parent_never_imports_torch():
    """`python bench.py --gpus N` (no WORLD_SIZE) takes the launcher branch before torch is imported: the parent
    cannot have initialised the GPU.  Checked by making the spawn itself the observable: a stub `launch_ranks`."""
    code = ("import sys, os; os.environ.pop('WORLD_SIZE', None); import bench; "
            "bench.launch_ranks = lambda n, argv, **kw: (print('SPAWN', n, argv, 'torch' in sys.modules) or 0); "
            "sys.exit(bench.main(['--gpus', '2', '--steps', '3']))")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "SPAWN 2 ['--gpus', '2', '--steps', '3'] False" in r.stdout


def test_step_statistics_and_scaling_reference_are_reported():
    st = bench.step_stats([1.0, 1.1, 1.2, 5.0, 1.05])
    assert st == {"n": 5, "median": 1.1, "p10": 1.02, "p90": 3.48, "min": 1.0, "max": 5.0} and bench.step_stats([]) is None
    src = open(os.path.join(ROOT, "bench.py")).read()
    # the N = 1 line carries the multi-GPU workload's per-GPU shape as the scaling denominator, every line its per-step spread,
    # and the N > 1 path refuses to time anything unless identical inputs give bit-identical records on every rank
    assert 'line["scale_ref"] = scale_ref' in src and '"step_ms": step_stats(per_step_ms)' in src
    assert "fdist.replicated_shards_identical(ids_all, d_all, world)" in src and 'line["shard_check"] = shard_check' in src


def test_default_workloads_follow_baseline_configs():
    a = bench.parse([])
    assert a.gpus == 1 and a.model is None and a.batch is None and a.gallery is None   # resolved per world size in worker()
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"arcface" if cfg_multi else "cnn"' in src and "1024 if cfg_multi else 256" in src and "10000 if cfg_multi else 36" in src


@pytest.mark.parametrize("mt", ["cnn", "arcface"])
def test_committed_bn_statistics_equal_the_test_weights(mt, calibrated_sd):
    """bench.py's weights (seeded init + committed BatchNorm statistics) ARE the parity tests' calibrated weights."""
    import frmap_amd
    from frmap_amd import synth
    from oracle import weights
    sd_test = calibrated_sd(mt)
    sd_bench = synth.calibrated_state_dict(mt, synth.shapes_of(frmap_amd.get_model(mt, 36)), weights.SEEDS[mt][0])
    assert set(sd_test) == set(sd_bench)
    for k in sd_test:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert torch.allclose(sd_test[k], sd_bench[k], rtol=2e-4, atol=2e-5), k   # fp32 round-off of the re-run calibration
        else:
            assert torch.equal(sd_test[k], sd_bench[k]), k
    with pytest.raises(FileNotFoundError):
        synth.calibrated_state_dict("hybrid", {}, 1)
