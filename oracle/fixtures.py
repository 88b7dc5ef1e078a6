"""Seeded inputs shared by the golden-vector generator and the parity tests.  TEST INFRASTRUCTURE ONLY.

Weights and inputs are regenerated from (seed, name) on both sides, so the committed fixtures under
``tests/golden/`` hold only the *reference's outputs* (small), never weights or feature maps.
"""
from __future__ import annotations

import os
import sys
from typing import Dict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from lbm_amd.synth import _rng, synthetic_state_dict  # noqa: E402
from oracle.lbm_ref_cpu import Geometry  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# geometry of the committed fixtures
GEOMS = {
    # reduced channels: oracle-vs-reference only (the HIP kernels are specialised for C = 256)
    "tiny64": dict(size=(96, 128), embed_dim=64, memory_size=4, N=16, B=2),
    # full channel count, reduced spatial size: oracle and HIP parity
    "small256": dict(size=(96, 128), embed_dim=256, memory_size=4, N=16, B=2),
    # the published geometry (lbm/configs/demo.yaml)
    "full256": dict(size=(384, 512), embed_dim=256, memory_size=12, N=8, B=1),
}


def geometry(name: str) -> Geometry:
    g = GEOMS[name]
    return Geometry(size=tuple(g["size"]), embed_dim=g["embed_dim"], memory_size=g["memory_size"])


def transformer_state_dict(g: Geometry, seed: int = 42) -> Dict[str, torch.Tensor]:
    """Synthetic weights for the tracker transformer, keys without the 'transformer.' prefix."""
    from lbm_amd.transformer import LBMTransformer
    tr = LBMTransformer(size=list(g.size), stride=g.stride, n_layer=g.num_layers, hidden_dim=g.embed_dim, n_head=g.num_heads,
                        n_level=g.num_levels, n_neighbor=g.num_neighbors, n_mem=g.memory_size, n_topk=g.top_k_regions)
    return synthetic_state_dict(tr, seed)


def _t(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a.astype(np.float32)))


def step_inputs(g: Geometry, B: int, N: int, seed: int = 7) -> Dict[str, torch.Tensor]:
    """query / feat / memories / masks for one LBMTransformer.forward call.

    feat is smooth-ish (random field + a few coarse bumps) so the top-k margins are not razor thin; the masks
    exercise all three row kinds: fully ignored rows, partially visible rows and fully attended rows; one point
    is 'not queried yet'."""
    C, M, h, w = g.embed_dim, g.memory_size, g.h, g.w
    r = _rng(seed, f"step:{B}:{N}:{C}:{M}:{h}:{w}")
    feat = r.normal(0, 1, (B, C, h, w))
    coarse = r.normal(0, 1, (B, C, h // 8, w // 8))
    feat = feat + np.repeat(np.repeat(coarse, 8, axis=2), 8, axis=3)
    query = r.normal(0, 1, (B, N, C))
    stream = r.normal(0, 1, (B, N, M, C))
    coll = r.normal(0, 1, (B, N, M, C))
    vis = r.uniform(0, 1, (B, N, M)) < 0.35
    mem = np.zeros((B, N, M), dtype=bool)
    for b in range(B):
        mem[b, 0] = True                       # never written: fully ignored row
        mem[b, 1, : M // 2] = True             # half-filled memory
        vis[b, 2] = True                       # always occluded: fully ignored via vis_mask
        vis[b, 3] = False                      # fully attended row
    queried = np.ones((B, N), dtype=bool)
    queried[:, 4] = False
    return {"query": _t(query), "feat": _t(feat), "stream": _t(stream), "collision": _t(coll),
            "vis_mask": torch.from_numpy(vis), "mem_mask": torch.from_numpy(mem), "queried": torch.from_numpy(queried)}


def sequence_feats(g: Geometry, B: int, T: int, seed: int = 11) -> torch.Tensor:
    """(T, B, C, h, w): a base field that drifts slowly so memories matter."""
    C, h, w = g.embed_dim, g.h, g.w
    r = _rng(seed, f"seq:{B}:{T}:{C}:{h}:{w}")
    base = r.normal(0, 1, (B, C, h, w))
    out = []
    for t in range(T):
        out.append(base + 0.3 * r.normal(0, 1, (B, C, h, w)))
    return _t(np.stack(out, 0))


def msda_inputs(g: Geometry, B: int, Q: int, seed: int = 13) -> Dict[str, torch.Tensor]:
    C = g.embed_dim
    r = _rng(seed, f"msda:{B}:{Q}:{C}:{g.h}:{g.w}")
    L = g.num_tokens
    ref = r.uniform(-0.05, 1.05, (B, Q, 1, 2)).clip(0, 1)          # some on the border -> zero-padded taps
    return {"query": _t(r.normal(0, 1, (B, Q, C))), "value": _t(r.normal(0, 1, (B, L, C))),
            "reference_points": _t(np.repeat(ref, g.num_levels, axis=2))}
