"""CPU, world_size 2 over gloo: the shard → match → all-gather driver (frmap_amd.dist)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from frmap_amd import dist as fdist


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _fake_match(xs):
    # stand-in for embed+match: id = first pixel value, dist = second (deterministic per face)
    return xs[:, 0, 0, 0].to(torch.int32), xs[:, 0, 0, 1].to(torch.float32)


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = torch.zeros(total, 3, 2, 2)
        x[:, 0, 0, 0] = torch.arange(total) % 37 - 1          # includes the -1 "Unknown" id
        x[:, 0, 0, 1] = torch.arange(total) * 0.125
        ids, d = fdist.sharded_embed_and_match(_fake_match, x, total)
        lo, hi = fdist.shard_bounds(total, rank, world)
        ids2, d2 = fdist.sharded_embed_and_match(_fake_match, x[lo:hi], total, already_sharded=True)
        same = ids2.tolist() == ids.tolist() and d2.tolist() == d.tolist()
        if total % world == 0:   # equal shards: the packed fast path must agree too
            mi, md = _fake_match(x[lo:hi])
            rec = torch.stack([mi, md.view(torch.int32)], dim=1)
            i3, d3 = fdist.gather_packed(rec)
            same = same and i3.tolist() == ids.tolist() and d3.tolist() == d.tolist()
            # bench.py's N > 1 self-check: identical faces on every rank -> `world` bit-identical segments; rank-specific
            # faces (the shards above) are not
            mi, md = _fake_match(x[:total // world])
            i4, d4 = fdist.gather_packed(torch.stack([mi, md.view(torch.int32)], dim=1))
            same = same and fdist.replicated_shards_identical(i4, d4, world) and not fdist.replicated_shards_identical(i3, d3, world)
        q.put((rank, ids.tolist(), d.tolist(), same))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [16, 13])
def test_two_rank_gather_equals_single_rank(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(60) for p in procs]
    exp_ids = [(i % 37) - 1 for i in range(total)]
    exp_d = [i * 0.125 for i in range(total)]
    for rank, ids, d, same in res:
        assert ids == exp_ids and d == exp_d and same, rank
