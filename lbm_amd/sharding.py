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


# ---- the result record and its file (results wire format) --------------------------------------------
# One record per point-frame: four fp32 [x, y, visible (0 / 1), rho], 16 bytes -- the payload of the RCCL gather above
# (one collective for coordinates, visibility and uncertainty together) and, appended frame after frame behind a
# 40-byte header, the track file a trajectory consumer reads back in the layout of LBM.inference's result
# (lbm.py:276-280: 'points' (B,T,N,2), 'visibility' (B,T,N)).

TRACK_MAGIC = b"LBMTRK1\n"
_TRACK_HEADER = "<8sIIIIff8x"        # magic, streams, points, frames, floats per record, frame height, frame width


def pack_tracks(coords: torch.Tensor, visible: torch.Tensor, rho: torch.Tensor) -> torch.Tensor:
    """coords (..., N, 2), visible (..., N) bool, rho (..., N) -> records (..., N, 4) fp32."""
    return torch.cat([coords.float(), visible.float().unsqueeze(-1), rho.float().unsqueeze(-1)], dim=-1).contiguous()


def unpack_tracks(records: torch.Tensor):
    """records (..., N, 4) -> coords (..., N, 2), visible (..., N) bool, rho (..., N)."""
    return records[..., :2], records[..., 2] > 0.5, records[..., 3]


class TrackWriter:
    """Appends one (streams, points, 4) record block per frame; the frame count in the header is kept current, so a file
    cut short by a crash still reads back every complete frame."""

    def __init__(self, path: str, n_streams: int, n_points: int, frame_hw):
        import struct
        self._struct = struct
        self.shape = (int(n_streams), int(n_points), 4)
        self.hw = (float(frame_hw[0]), float(frame_hw[1]))
        self.frames = 0
        self._f = open(path, "wb")
        self._header()

    def _header(self) -> None:
        self._f.seek(0)
        self._f.write(self._struct.pack(_TRACK_HEADER, TRACK_MAGIC, self.shape[0], self.shape[1], self.frames, 4, *self.hw))
        self._f.seek(0, 2)

    def append(self, records: torch.Tensor) -> None:
        if tuple(records.shape) != self.shape:
            raise ValueError(f"expected a {self.shape} record block, got {tuple(records.shape)}")
        self._f.write(records.detach().to("cpu", torch.float32).contiguous().numpy().tobytes())
        self.frames += 1
        self._header()

    def close(self) -> None:
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_tracks(path: str) -> dict:
    """-> {'points': (S,T,N,2), 'visibility': (S,T,N) bool, 'rho': (S,T,N), 'frame_hw': (H, W)}; only complete frames."""
    import os
    import struct

    import numpy as np
    hsize = struct.calcsize(_TRACK_HEADER)
    with open(path, "rb") as f:
        head = f.read(hsize)
    if len(head) < hsize or head[:8] != TRACK_MAGIC:
        raise ValueError(f"{path}: not a track file")
    _, S, N, T, rec, H, W = struct.unpack(_TRACK_HEADER, head)
    if rec != 4:
        raise ValueError(f"{path}: {rec} floats per record, this build reads 4")
    T = min(T, (os.path.getsize(path) - hsize) // (S * N * 16)) if S * N else 0
    data = np.fromfile(path, dtype="<f4", count=T * S * N * 4, offset=hsize).reshape(T, S, N, 4)
    r = torch.from_numpy(data.copy()).permute(1, 0, 2, 3)
    pts, vis, rho = unpack_tracks(r)
    return {"points": pts.contiguous(), "visibility": vis.contiguous(), "rho": rho.contiguous(), "frame_hw": (H, W)}
