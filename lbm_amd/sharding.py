"""Multi-GPU layout of the tracking workload: independent video streams are partitioned over one process per GPU.

The step has no cross-stream (and no cross-point) interaction, so there is no collective in the data
path; the only communication is the gather of the finished tracks on one rank — `torch.distributed` backend "nccl"
(= RCCL over xGMI on ROCm) on GPUs, "gloo" in the CPU tests.  The reference's only multi-GPU inference pattern is
the same idea with Ray actors over contiguous dataset chunks (tools/eval_pttrack_mp.py:28-41, 215-228).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def split_streams(n_streams: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of streams owned by `rank`; blocks differ in size by at most one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside [0, {world})")
    base, extra = divmod(n_streams, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_tracks(local: torch.Tensor, n_streams: int, dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """local: (S_local, ...) results of this rank's streams (same trailing shape on every rank).
    Returns the (n_streams, ...) tensor in global stream order on rank `dst`, None elsewhere.
    One collective per call: call it once per run or per K frames, never per kernel (payload is ~16 B per point-frame)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        if local.shape[0] != n_streams:
            raise ValueError("single process must own every stream")
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = split_streams(n_streams, world, rank)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} owns streams [{lo},{hi}) but passed {local.shape[0]} result rows")
    cap = -(-n_streams // world)                                  # pad to the largest block so one gather suffices
    buf = local.new_zeros((cap,) + tuple(local.shape[1:]))
    buf[: hi - lo] = local
    out: Optional[List[torch.Tensor]] = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst, group=group)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        a, b = split_streams(n_streams, world, r)
        parts.append(out[r][: b - a])
    return torch.cat(parts, 0)


def max_over_ranks(value: float, device, group=None) -> float:
    """Slowest rank's time (the benchmark clock of a weak-scaling run)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
