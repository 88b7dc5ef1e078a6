"""Tracker transformer with the reference's module tree; the arithmetic runs in ``liblbm_hip.so``.

Mirrors ``lbm/models/lbm/transformer.py`` (LBMTransformer :10-151, LBMTransformerLayer :154-340,
MLP :343-355): identical constructor signatures, parameter names and ``forward`` return structure, so a
``lbm.pt`` state dict loads with ``strict=True``.  The nn.Modules below are parameter holders — the
forward pass is ONE call into ``lbm_hip_track_step`` (see lbm_amd/engine.py), never PyTorch ops.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from . import engine, hip
from .config import LBMConfig
from .msdeformattn import MultiScaleDeformableAttention


class MLP(nn.Module):
    def __init__(self, in_dim, out_dim, expansion_factor=4, dropout=0.1):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(in_dim, in_dim * expansion_factor), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(in_dim * expansion_factor, out_dim), nn.Dropout(dropout))

    def forward(self, x):
        shp = x.shape
        y = hip.linear(x.reshape(-1, shp[-1]).contiguous(), self.net[0].weight.detach(), self.net[0].bias.detach(), act=1)
        y = hip.linear(y, self.net[3].weight.detach(), self.net[3].bias.detach())
        return y.view(*shp[:-1], -1)


class LBMTransformerLayer(nn.Module):
    def __init__(self, size, stride, hidden_dim, n_head, n_level, n_neighbor, n_mem, dropout=0.1):
        super().__init__()
        self.size, self.stride, self.hidden_dim = size, stride, hidden_dim
        for kind in ("collision", "stream"):
            setattr(self, f"{kind}_attention", nn.MultiheadAttention(hidden_dim, n_head, dropout, batch_first=True))
            setattr(self, f"{kind}_dropout", nn.Dropout(dropout))
            setattr(self, f"{kind}_norm", nn.LayerNorm(hidden_dim))
            setattr(self, f"{kind}_ffn", MLP(hidden_dim, hidden_dim, dropout=dropout))
            setattr(self, f"{kind}_norm2", nn.LayerNorm(hidden_dim))
            emb = nn.Parameter(torch.zeros(1, n_mem + 1, hidden_dim))
            nn.init.trunc_normal_(emb, std=0.02)
            setattr(self, f"{kind}_time_emb", emb)
        self.corr_conv = nn.Conv2d(n_level, 1, kernel_size=1, stride=1, padding=0)
        self.lattice_attention = MultiScaleDeformableAttention(hidden_dim, n_head, n_level, n_neighbor, dropout=dropout,
                                                               batch_first=True)
        self.lattice_dropout = nn.Dropout(dropout)
        self.lattice_norm = nn.LayerNorm(hidden_dim)
        self.lattice_ffn = MLP(hidden_dim, hidden_dim, dropout=dropout)
        self.lattice_norm2 = nn.LayerNorm(hidden_dim)
        self.lattice_update = nn.MultiheadAttention(hidden_dim, n_head, dropout, batch_first=True)
        self.lattice_update_dropout = nn.Dropout(dropout)
        self.lattice_update_norm = nn.LayerNorm(hidden_dim)
        self.lattice_update_ffn = MLP(hidden_dim, hidden_dim, dropout=dropout)
        self.lattice_update_norm2 = nn.LayerNorm(hidden_dim)
        self.reference_rho = nn.Linear(hidden_dim, 1)

    def forward(self, *a, **k):
        raise NotImplementedError("layers are executed by LBMTransformer.forward as one fused HIP step")


class LBMTransformer(nn.Module):
    def __init__(self, size, stride, n_layer, hidden_dim, n_head, n_level, n_neighbor, n_mem, n_topk, dropout=0.1,
                 random_mask_ratio=0):
        super().__init__()
        self.size = list(size)
        self.stride = stride
        self.n_neighbor = n_neighbor
        self.random_mask_ratio = random_mask_ratio
        self.layers_n_topk = [n_topk] + [4] * (n_layer - 2) + [1]
        H, W = size
        h, w = H // stride, W // stride
        self.register_buffer("spatial_shapes", torch.tensor([[h, w], [h // 2, w // 2], [h // 4, w // 4], [h // 8, w // 8]],
                                                            dtype=torch.long))
        self.register_buffer("start_levels", torch.tensor([0, h * w, h * w + h * w // 4, h * w + h * w // 4 + h * w // 16],
                                                          dtype=torch.long))
        self.layers = nn.ModuleList([LBMTransformerLayer(size, stride, hidden_dim, n_head, n_level, n_neighbor, n_mem, dropout)
                                     for _ in range(n_layer)])
        for name, pts in (("offset", n_neighbor), ("vis", n_neighbor), ("collision_mem", 4)):
            setattr(self, f"{name}_layer", MultiScaleDeformableAttention(hidden_dim, n_head, n_level, pts, dropout=dropout,
                                                                         batch_first=True))
            setattr(self, f"{name}_dropout", nn.Dropout(dropout))
            setattr(self, f"{name}_norm", nn.LayerNorm(hidden_dim))
            setattr(self, f"{name}_ffn", MLP(hidden_dim, hidden_dim, dropout=dropout))
            setattr(self, f"{name}_norm2", nn.LayerNorm(hidden_dim))
        self.offset_head = MLP(hidden_dim, out_dim=2, expansion_factor=1, dropout=0)
        self.offset_act = nn.Tanh()
        self.vis_head = MLP(hidden_dim, out_dim=2, expansion_factor=1, dropout=0)
        self.return_corr = False      # the (B,N,h*w) correlation maps are only needed by the training loss
        self.cfg = LBMConfig(input_size=(H, W), stride=stride, num_layers=n_layer, embed_dim=hidden_dim, num_heads=n_head,
                             num_levels=n_level, num_neighbors=n_neighbor, top_k_regions=n_topk, memory_size=n_mem)
        self._engine: Optional[engine.TrackEngine] = None
        self.precision: Optional[str] = None   # None -> LBM_AMD_PRECISION or "split"; "f32" = exact fp32 MFMA path

    def set_precision(self, precision: Optional[str]) -> None:
        self.precision = precision
        self._engine = None

    # -- engine / packed weights ----------------------------------------------------------------
    def invalidate(self) -> None:
        self._engine = None

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def get_engine(self) -> engine.TrackEngine:
        if self._engine is None:
            if self.random_mask_ratio:
                raise hip.HipError("random_mask_ratio > 0 is a training-time augmentation; the HIP step implements inference")
            self.cfg.memory_size = self.layers[0].collision_time_emb.shape[1] - 1
            self._engine = engine.TrackEngine(self.cfg, engine.TrackerWeights(self, self.cfg), self.precision)
        return self._engine

    # -- reference API --------------------------------------------------------------------------
    def tokens_from_feat(self, feat: torch.Tensor) -> torch.Tensor:
        """(B,C,h,w) feature map -> (B,L,C) pyramid token buffer (transformer.py:51-71, 106)."""
        B, Cc, h, w = feat.shape
        P0 = h * w
        L = P0 + P0 // 4 + P0 // 16 + P0 // 64
        tok = torch.empty(B, L, Cc, device=feat.device, dtype=torch.float32)
        tok[:, :P0] = feat.permute(0, 2, 3, 1).reshape(B, P0, Cc)
        hip.pyramid(tok, h, w)
        return tok

    def get_multiscale_feats(self, feat: torch.Tensor) -> List[torch.Tensor]:
        tok = self.tokens_from_feat(feat)
        out, st = [], 0
        for hl, wl in self.cfg.level_shapes:
            out.append(tok[:, st:st + hl * wl])
            st += hl * wl
        return out

    def forward(self, query, feat, stream_dist, collision_dist, vis_mask, mem_mask, queried_now_or_before, tokens=None):
        """transformer.py:74-151.  ``tokens`` (optional) is an already built (B,L,C) pyramid buffer that makes
        ``feat`` unnecessary (the encoder produces it directly)."""
        if tokens is None:
            tokens = self.tokens_from_feat(feat.float())
        eng = self.get_engine()
        mem_dt = engine.state_dtype(self.precision)
        out = eng.step(tokens, query.contiguous().float(), stream_dist.contiguous().to(mem_dt), collision_dist.contiguous().to(mem_dt),
                       vis_mask.contiguous(), mem_mask.contiguous(), ring_head=0,
                       queried=queried_now_or_before.contiguous(), update_state=False, want_memory=True, want_aux=True,
                       want_corr=self.return_corr)
        outs = []
        for i in range(len(self.layers)):
            outs.append({"corr": out.get(f"corr{i}"), "reference_points": out[f"ref{i}"], "reference_rho": out[f"rho{i}"],
                         "topk_indices": out[f"idx{i}"].long()})
        memory = {"stream": out["stream"], "collision": out["collision"]}
        return outs, memory, out["offset"], out["vis"], out["rho"]
