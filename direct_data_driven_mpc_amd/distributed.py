"""Multi-GPU sharding of a batch of independent controller instances.

The QP path has no exchange step: instances are independent (the reference
already treats its controllers independently,
utilities/reproduction/paper_reproduction.py:251-268).  So the batch is split
into contiguous blocks, one per rank (one process per GPU), every rank solves
its block with no data-path collective, and ONE all-gather at the end collects
optimal_u / cost / status (RCCL over xGMI on GPUs, gloo on CPU in the tests).
"""
from __future__ import annotations

from typing import Tuple


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `total` instances owned by `rank`; the first
    `total % world` ranks get one extra instance."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_results(u_opt, cost, status, total: int):
    """All-gather the per-rank result blocks into full [total, ...] tensors.

    Shards may differ by one row (see shard_bounds); blocks are padded to the
    largest shard for the collective and trimmed afterwards.  Works on any
    initialised torch.distributed backend ("nccl" = RCCL on ROCm, or "gloo")."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    per = -(-total // world)
    lo, hi = shard_bounds(total, rank, world)
    assert u_opt.shape[0] == hi - lo

    # one collective instead of three: [optimal_u | cost | status] packed per instance (status values are small
    # integers, exact in float64)
    nu = u_opt.shape[1]
    pad = torch.zeros((per, nu + 2), dtype=torch.float64, device=u_opt.device)
    pad[: hi - lo, :nu] = u_opt
    pad[: hi - lo, nu] = cost
    pad[: hi - lo, nu + 1] = status.to(torch.float64)
    out = torch.empty((world * per, nu + 2), dtype=torch.float64, device=u_opt.device)
    dist.all_gather_into_tensor(out, pad)
    if total == world * per:
        full = out
    else:
        parts = []
        for r in range(world):
            a, b = shard_bounds(total, r, world)
            parts.append(out[r * per: r * per + (b - a)])
        full = torch.cat(parts, dim=0)
    return full[:, :nu].contiguous(), full[:, nu].contiguous(), full[:, nu + 1].to(status.dtype)
