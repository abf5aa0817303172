"""Multi-GPU driver: faces shard across ranks, one all-gather collates the top-1 results.

The reference is single-device (`/root/reference/src/testing.py:28`, `src/app.py:279`; no
``torch.distributed`` anywhere).  Per-face independence (eval-mode BatchNorm uses
running stats) makes the path embarrassingly parallel: one process per GPU, weights and gallery
replicated, rank r embeds + matches faces ``[r*B/R, (r+1)*B/R)``, and a single
``all_gather`` of 8 bytes per face — (int32 id, fp32 distance) packed in one int64 — over
RCCL/xGMI returns every face's result to every rank.  At 8 KiB per rank the collective is
latency-bound; it is issued once, in place, on the compute stream (no bucketing, no ring tuning).

The sharding and the pack/unpack are backend-agnostic, so the N>1 logic is tested on CPU with
``gloo`` (``tests/test_dist_cpu.py``) using a stand-in matcher; on GPUs the backend is ``nccl``
(= RCCL on ROCm).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split; the first ``total % world`` ranks take one extra face."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_results(ids: torch.Tensor, dists: torch.Tensor) -> torch.Tensor:
    """(int32 id, fp32 dist) → one int64 per face: high word = id, low word = the float's bits."""
    lo = dists.to(torch.float32).contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    return (ids.to(torch.int64) << 32) | lo


def unpack_results(packed: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    ids = (packed >> 32).to(torch.int32)
    lo = (packed & 0xFFFFFFFF).to(torch.int64)
    lo = torch.where(lo >= 2 ** 31, lo - 2 ** 32, lo).to(torch.int32)
    return ids, lo.contiguous().view(torch.float32)


def gather_results(ids: torch.Tensor, dists: torch.Tensor, total: int,
                   group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """All ranks contribute their shard's (ids, dists); every rank receives all ``total`` results in
    face order.  Shards may be ragged (sizes from ``shard_bounds``); each rank pads to the largest
    shard so a single fixed-size ``all_gather_into_tensor`` suffices."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return ids, dists
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = -(-total // world)
    packed = pack_results(ids, dists)
    out_dev = packed.device
    if dist.get_backend(group) == "gloo" and packed.is_cuda:
        packed = packed.cpu()  # gloo collectives run on host memory (CPU rehearsal of the N>1 path)
    send = torch.full((per,), -1, dtype=torch.int64, device=packed.device)
    send[: packed.numel()] = packed
    recv = torch.empty((world * per,), dtype=torch.int64, device=packed.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.to(out_dev)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(total, r, world)
        parts.append(recv[r * per: r * per + (hi - lo)])
    return unpack_results(torch.cat(parts))


def gather_packed(records: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Equal-shard fast path: ``records`` = this rank's int32 ``[B, 2]`` (id, bits(dist)) straight from
    the match kernel; ONE ``all_gather_into_tensor`` and two zero-copy views, no pack/unpack kernels.
    Returns (ids int32[world*B], dists fp32[world*B]) — strided views of the gathered buffer."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        recv = records
    else:
        world = dist.get_world_size(group)
        send = records
        if dist.get_backend(group) == "gloo" and send.is_cuda:
            send = send.cpu()
        recv = torch.empty((world * send.shape[0], 2), dtype=torch.int32, device=send.device)
        dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
        recv = recv.to(records.device)
    return recv[:, 0], recv.view(torch.float32)[:, 1]


def replicated_shards_identical(ids: torch.Tensor, dists: torch.Tensor, world: int) -> bool:
    """True iff a gathered result (``world`` equal segments, in rank order) holds the SAME records in every segment -
    what the all-gather returns when every rank was given identical faces.  Distances are compared by their bits."""
    n = ids.shape[0]
    if world <= 0 or n % world:
        return False
    per = n // world
    i0, d0 = ids[:per], dists[:per].contiguous().view(torch.int32)
    for r in range(1, world):
        if not torch.equal(ids[r * per:(r + 1) * per], i0):
            return False
        if not torch.equal(dists[r * per:(r + 1) * per].contiguous().view(torch.int32), d0):
            return False
    return True


def sharded_embed_and_match(match_fn: Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor]],
                            x_global_or_shard: torch.Tensor, total: int, already_sharded: bool = False,
                            group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Run ``match_fn`` (faces → (ids, dists)) on this rank's shard and all-gather the results.

    ``x_global_or_shard``: either the full B×3×H×W batch (every rank slices its own part) or, with
    ``already_sharded=True``, just this rank's faces (the usual case: each rank loads its own)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if already_sharded:
        xs = x_global_or_shard
    else:
        lo, hi = shard_bounds(total, rank, world)
        xs = x_global_or_shard[lo:hi]
    ids, dists = match_fn(xs)
    return gather_results(ids, dists, total, group)
