"""Deterministic synthetic checkpoints, videos and query grids (SURVEY §8 d).

``lbm.pt`` is a HuggingFace download that does not exist offline, so parity and benchmarks are pinned on seeded
synthetic weights in the exact ``lbm.pt`` key layout.  Generation uses numpy's PCG64 (stable across machines and
library versions) keyed by (seed, tensor name), so the same weights materialise here and on the GPU box without
shipping 70 MB.
"""
from __future__ import annotations

import hashlib
import math
from typing import Dict, Tuple

import numpy as np
import torch


def _rng(seed: int, name: str) -> np.random.Generator:
    digest = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.random.Generator(np.random.PCG64(int.from_bytes(digest[:8], "little")))


def _smooth_field(g: np.random.Generator, shape: Tuple[int, ...], passes: int) -> np.ndarray:
    """White noise over the last two axes low-passed by `passes` rounds of the separable [1, 2, 1] / 4 filter (edges clamped) and
    rescaled to unit standard deviation: neighbouring cells are alike, cells a few steps apart are not.  Plain numpy adds in a
    fixed order, so the field is the same on every machine."""
    v = g.normal(0.0, 1.0, shape)
    for _ in range(passes):
        for ax in (-2, -1):
            lo = np.concatenate([np.take(v, [0], axis=ax), np.take(v, range(v.shape[ax] - 1), axis=ax)], axis=ax)
            hi = np.concatenate([np.take(v, range(1, v.shape[ax]), axis=ax), np.take(v, [v.shape[ax] - 1], axis=ax)], axis=ax)
            v = 0.25 * lo + 0.5 * v + 0.25 * hi
    return v / v.std()


PRESETS = ("flat", "peaked")


def synthetic_state_dict(model: torch.nn.Module, seed: int = 42, preset: str = "flat") -> Dict[str, torch.Tensor]:
    """A full state dict for ``model`` (any module tree with lbm.pt-style names) with non-degenerate values:
    Kaiming-scale convolutions, Xavier-scale linears, N(0, 0.02) embeddings, BatchNorm statistics around
    (0, 1) +- 10 %, and **non-zero** deformable-attention offset / weight projections (the default init zeroes
    them, msdeformattn.py:76-79, which would leave the gather path untested).

    preset "flat" (the default, what every golden fixture and bench figure was made with): encoder content (token norm ~80,
    most of it one direction shared by every cell) swamps the N(0, 0.02) positional embedding, so every correlation row is
    nearly flat -- best cell 1.64-1.65 against a row median of 1.57, median gap between the k-th and (k+1)-th value 1e-3 --
    and the arg-max (transformer.py:306-313), the thing that moves a track by >= 4 px, is decided by near-ties.
    preset "peaked": what a trained tracker's rows look like -- one smooth bump around the point's cell.  The positional
    embedding is a low-passed random field of unit variance (neighbouring cells alike, cells three apart unrelated) and the
    fusion BatchNorm gain is cut to 5 %, so a token is its cell's signature plus a little frame content: the query's own
    cell wins by ~0.1 and the k best cells are its neighbourhood, ranked by distance from the query's sub-cell position."""
    if preset not in PRESETS:
        raise ValueError(f"preset must be one of {PRESETS}")
    out: Dict[str, torch.Tensor] = {}
    for name, t in model.state_dict().items():
        g = _rng(seed, name)
        shape = tuple(t.shape)
        leaf = name.rsplit(".", 1)[-1]
        if t.dtype in (torch.int64, torch.int32):
            out[name] = t.detach().clone().cpu()          # spatial_shapes / start_levels / num_batches_tracked
            continue
        if preset == "peaked" and "frame_pos_embedding" in name:
            v = _smooth_field(g, shape, passes=2)
        elif preset == "peaked" and name.endswith(("fusion.1.weight", "fusion.1.bias")):
            v = 0.05 * (g.normal(1.0, 0.1, shape) if leaf == "weight" else g.normal(0.0, 0.05, shape))
        elif leaf == "running_mean":
            v = g.normal(0.0, 0.1, shape)
        elif leaf == "running_var":
            v = g.uniform(0.9, 1.1, shape)
        elif "time_emb" in name or "pos_embedding" in name:
            v = g.normal(0.0, 0.02, shape)
        elif "sampling_offsets" in name:
            v = g.normal(0.0, 0.02, shape) if leaf == "weight" else g.uniform(-2.0, 2.0, shape)
        elif "attention_weights" in name:
            v = g.normal(0.0, 0.05, shape) if leaf == "weight" else g.normal(0.0, 0.5, shape)
        elif "corr_conv" in name:
            v = g.normal(0.25, 0.1, shape) if leaf == "weight" else g.normal(0.0, 0.05, shape)
        elif len(shape) == 4:                              # convolutions
            fan_in = shape[1] * shape[2] * shape[3]
            v = g.normal(0.0, math.sqrt(2.0 / fan_in), shape)
        elif len(shape) == 2:                              # linear / in_proj
            v = g.normal(0.0, math.sqrt(2.0 / (shape[0] + shape[1])), shape)
        elif len(shape) == 1:
            is_norm_gain = leaf == "weight"                # LayerNorm / BatchNorm gains
            v = g.normal(1.0, 0.1, shape) if is_norm_gain else g.normal(0.0, 0.05, shape)
        else:
            v = g.normal(0.0, 0.02, shape)
        out[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).reshape(shape)
    return out


def synthetic_video(T: int, H: int, W: int, seed: int = 42, blobs: int = 16) -> torch.Tensor:
    """(1, T, 3, H, W) float32, integer-valued in [0, 255]: low-frequency sinusoid texture + translating Gaussian
    blobs, so correlation peaks are well separated (pure noise would make top-1 ties meaningless)."""
    g = _rng(seed, f"video:{T}:{H}:{W}")
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    base = np.zeros((3, H, W), dtype=np.float32)
    for c in range(3):
        for _ in range(6):
            fx, fy = g.uniform(0.002, 0.03, 2)
            ph = g.uniform(0, 2 * np.pi)
            base[c] += g.uniform(10, 30) * np.sin(2 * np.pi * (fx * xx + fy * yy) + ph)
    pos = g.uniform([0.1 * W, 0.1 * H], [0.9 * W, 0.9 * H], (blobs, 2))
    vel = g.uniform(-3.0, 3.0, (blobs, 2))
    col = g.uniform(-90, 90, (blobs, 3))
    sig = g.uniform(8, 30, blobs)
    frames = np.empty((T, 3, H, W), dtype=np.float32)
    for t in range(T):
        f = base + 128.0
        for b in range(blobs):
            cx, cy = pos[b] + vel[b] * t
            blob = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sig[b] ** 2))
            f = f + col[b][:, None, None] * blob[None]
        frames[t] = np.clip(np.rint(f), 0, 255)
    return torch.from_numpy(frames).unsqueeze(0)


def grid_queries(H: int, W: int, grid_size: int) -> torch.Tensor:
    """The ``load_video(grid_query=True)`` rule (demo_utils.py:97-107): x = int(i W/(g+1)), y = int(j H/(g+1))."""
    sx, sy = W / (grid_size + 1), H / (grid_size + 1)
    pts = [(int(i * sx), int(j * sy)) for i in range(1, grid_size + 1) for j in range(1, grid_size + 1)]
    return torch.tensor(pts, dtype=torch.float32)


def random_queries(n: int, H: int, W: int, seed: int) -> torch.Tensor:
    g = _rng(seed, f"queries:{n}:{H}:{W}")
    xy = g.uniform([0.1 * W, 0.1 * H], [0.9 * W, 0.9 * H], (n, 2))
    return torch.from_numpy(xy.astype(np.float32))
