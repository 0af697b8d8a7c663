"""Sharding of a batch of independent solves over the ranks of a process group.

Trajectories never interact (control/iLQRController.py has no coupling between
calls), so a batch is cut into contiguous slices, one per rank / GPU, each rank
solves its slice with no communication, and the only exchange is an all-gather
of the small per-trajectory results at the end.  Works with any
torch.distributed backend ("nccl" = RCCL on the GPUs, "gloo" on CPU).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(B: int, world: int, rank: int) -> Tuple[int, int]:
    """[lo, hi) of rank's slice: contiguous, sizes differ by at most one, the first B % world ranks take the extra."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("need 0 <= rank < world")
    q, r = divmod(int(B), world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_results(local: Dict[str, torch.Tensor], B: int, group=None) -> Dict[str, torch.Tensor]:
    """All-gather per-trajectory tensors (first dimension = this rank's slice) into whole-batch tensors that every
    rank holds, in batch order.  Slices may differ in length by one: they are padded to the longest for the
    collective and trimmed afterwards."""
    world = dist.get_world_size(group)
    sizes = [shard_bounds(B, world, r)[1] - shard_bounds(B, world, r)[0] for r in range(world)]
    longest = max(sizes)
    out = {}
    for k, t in local.items():
        pad = t
        if t.shape[0] < longest:
            pad = torch.cat([t, t.new_zeros((longest - t.shape[0],) + tuple(t.shape[1:]))])
        buf = t.new_empty((world * longest,) + tuple(t.shape[1:]))
        dist.all_gather_into_tensor(buf, pad.contiguous(), group=group)
        out[k] = torch.cat([buf[r * longest: r * longest + sizes[r]] for r in range(world)])
    return out


def solve_sharded(solve_fn: Callable[[torch.Tensor, torch.Tensor], Dict[str, torch.Tensor]], x0: torch.Tensor,
                  u: torch.Tensor, group=None, gather: Optional[Tuple[str, ...]] = ("J", "iters", "status")):
    """Every rank holds (or can index) the whole batch x0 [B, nx], u [B, N-1, nu]; rank r solves rows
    shard_bounds(B, world, r) with `solve_fn(x0_slice, u_slice) -> dict of per-trajectory tensors` and the keys in
    `gather` are all-gathered.  Returns (local results, gathered whole-batch results, (lo, hi))."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = x0.shape[0]
    lo, hi = shard_bounds(B, world, rank)
    local = solve_fn(x0[lo:hi], u[lo:hi])
    whole = gather_results({k: local[k] for k in (gather or ())}, B, group) if gather else {}
    return local, whole, (lo, hi)
