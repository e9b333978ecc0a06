"""Scenario sharding over the GPUs of a node.  Scenarios are independent closed loops, so the data path has no
collective; the only exchange is one all-reduce of the statistics vectors (sums and maxima) per report, over
RCCL ("nccl" backend on ROCm) for device tensors or gloo for host tensors."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of `total` scenarios owned by `rank`; block sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank outside 0..world-1")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_stats(sums: torch.Tensor, maxima: torch.Tensor, group=None) -> tuple[torch.Tensor, torch.Tensor]:
    """In-place SUM / MAX all-reduce of the two statistics vectors; a no-op outside a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(maxima, op=dist.ReduceOp.MAX, group=group)
    return sums, maxima


def allreduce_solver_stats(stats: dict, device=None, group=None) -> dict:
    """The dictionary a controller's stats() returns (the slots of mpc_get_stats: solver counters, and the cost and
    constraint-violation statistics of the last closed-loop run) combined over the ranks: keys starting with "max_" take
    the maximum, every other key adds.  One SUM and one MAX all-reduce; the identity outside a process group."""
    keys = sorted(stats)
    mk = [k for k in keys if k.startswith("max_")]
    sk = [k for k in keys if not k.startswith("max_")]
    sv = torch.tensor([float(stats[k]) for k in sk], dtype=torch.float64, device=device)
    mv = torch.tensor([float(stats[k]) for k in mk], dtype=torch.float64, device=device)
    allreduce_stats(sv, mv, group)
    out = dict(zip(sk, sv.tolist()))
    out.update(zip(mk, mv.tolist()))
    return out
