"""Batch sharding across the GPUs of one node.

The instance batch is embarrassingly parallel: rank r solves a contiguous block of instances with
no collective on the data path; the only exchange is one all-gather of the control outputs
(front, rear) -- `2 * n/G * sizeof(T)` bytes per rank -- over RCCL/xGMI (backend "nccl") or, in the
CPU tests, gloo.  One process per GPU, launched by torch.distributed.run.
"""
from __future__ import annotations

from typing import Callable, Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """(first, count) of rank's contiguous block; blocks differ by at most one instance."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def shard_slice(n_total: int, rank: int, world: int, split: str = "block") -> slice:
    """The instances rank owns, as a slice of the batch: one contiguous block ("block"), or every world-th instance from
    `rank` on ("interleaved": SURVEY.md section 8e names it first -- the iteration count of an instance is a function of
    its speed, so a batch that arrives sorted by speed would give one rank of a block split all the long instances)."""
    if split == "block":
        first, count = shard_range(n_total, rank, world)
        return slice(first, first + count)
    if split != "interleaved":
        raise ValueError("split is 'block' or 'interleaved'")
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return slice(rank, n_total, world)


def solve_sharded(solve_fn: Callable, v, delta_y, delta_phi, group=None, split: str = "block"):
    """Solve the FULL batch (every rank holds the same full input tensors) by sharding it over the
    ranks of `group`, then all-gather the outputs so every rank returns the full (front, rear) in instance order.

    solve_fn(v, dy, dphi) -> (front, rear) on tensors of one shard (e.g. MpcSolver.solve_batch_compact).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = v.numel()
    sl = shard_slice(n, rank, world, split)
    f, r = solve_fn(v[sl].contiguous(), delta_y[sl].contiguous(), delta_phi[sl].contiguous())
    if world == 1:
        return f, r
    count = f.numel()
    # ragged shards: pad to the largest one so one all_gather_into_tensor moves everything
    cap = (n + world - 1) // world
    mine = torch.zeros((2, cap), dtype=f.dtype, device=f.device)
    mine[0, :count] = f
    mine[1, :count] = r
    allo = torch.empty((world, 2, cap), dtype=f.dtype, device=f.device)
    dist.all_gather_into_tensor(allo.view(-1), mine.view(-1), group=group)
    front = torch.empty(n, dtype=f.dtype, device=f.device)
    rear = torch.empty(n, dtype=f.dtype, device=f.device)
    for q in range(world):
        qs = shard_slice(n, q, world, split)
        qc = len(range(*qs.indices(n)))
        front[qs] = allo[q, 0, :qc]
        rear[qs] = allo[q, 1, :qc]
    return front, rear
