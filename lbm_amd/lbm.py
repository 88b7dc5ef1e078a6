"""LBM model container (reference: lbm/models/lbm/lbm.py) for the inference path.

Same constructor contract (an ``args`` namespace as produced by ``load_config``), same sub-module names
(``backbone``, ``transformer``) so ``load_state_dict(torch.load('lbm.pt')['model'], strict=True)`` works, same
``set_memory_size`` / ``inference`` behaviour.  Training (``forward`` + losses, lbm.py:111-209) is out of scope.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import hip
from .backbone import Backbone
from .config import LBMConfig
from .transformer import LBMTransformer


class LBM(nn.Module):
    def __init__(self, args):
        super().__init__()
        cfg = LBMConfig.from_args(args)
        self.cfg = cfg
        self.N = cfg.N
        self.T = cfg.T
        self.visibility_treshold = 0.8          # (sic) attribute name of the reference, lbm.py:17
        self.size = list(cfg.input_size)
        self.stride = cfg.stride
        self.num_layers = cfg.num_layers
        self.embed_dim = cfg.embed_dim
        self.num_heads = cfg.num_heads
        self.num_levels = cfg.num_levels
        self.num_neighbors = cfg.num_neighbors
        self.memory_size = cfg.memory_size
        self.top_k_regions = cfg.top_k_regions
        self.dropout = cfg.dropout
        for name in ("lambda_cls", "lambda_vis", "lambda_reg", "lambda_unc", "lambda_ref", "loss_after_query"):
            setattr(self, name, getattr(cfg, name))
        self.h = self.size[0] // self.stride
        self.w = self.size[1] // self.stride
        self.extend_queries = False
        self.backbone = Backbone(size=self.size, stride=self.stride, embed_dim=self.embed_dim)
        self.transformer = LBMTransformer(size=self.size, stride=self.stride, n_layer=self.num_layers,
                                          hidden_dim=self.embed_dim, n_head=self.num_heads, n_level=self.num_levels,
                                          n_neighbor=self.num_neighbors, n_mem=self.memory_size, n_topk=self.top_k_regions,
                                          dropout=self.dropout)

    def set_precision(self, precision) -> None:
        """"split" (default): contractions on the f16 matrix cores with split operands, fp32-grade; "f32": exact fp32 MFMA;
        "f16": reduced precision (plain f16 operands, fp32 accumulate; 1e-2 on logits), ~20 % faster;
        "f16s": f16 STORAGE of the memories and value maps, single-plane products, but an fp32-grade encoder and correlation +
        top-k (the arg-max is the fp32-grade one): the reduced-precision mode to prefer."""
        self.backbone.set_precision(precision)
        self.transformer.set_precision(precision)

    def save_packed(self, path: str) -> int:
        """Writes the packed kernel weights of this model as one arena file (lbm_amd/packed.py)."""
        from .packed import save_packed
        return save_packed(self, path)

    def load_packed(self, path: str, expect_digest=None) -> dict:
        """Installs packed kernel weights from an arena file: one read, one host-to-device copy, no folding."""
        from .packed import load_packed
        return load_packed(self, path, expect_digest)

    # lbm.py:70-87 — resize the past time embeddings by linear interpolation (align_corners=True), keep "now".
    def set_memory_size(self, memory_size: int) -> None:
        for layer in self.transformer.layers:
            for kind in ("collision", "stream"):
                emb = getattr(layer, f"{kind}_time_emb").detach()
                past, now = emb[0, :-1], emb[0, -1:]
                M0 = past.shape[0]
                if memory_size == 1 or M0 == 1:
                    pos = torch.zeros(memory_size, device=emb.device, dtype=emb.dtype)
                else:
                    pos = torch.arange(memory_size, device=emb.device, dtype=emb.dtype) * ((M0 - 1) / (memory_size - 1))
                i0 = pos.floor().long().clamp(max=M0 - 1)
                i1 = (i0 + 1).clamp(max=M0 - 1)
                lam = (pos - i0.to(emb.dtype)).unsqueeze(1)
                new_past = past[i0] * (1 - lam) + past[i1] * lam
                setattr(layer, f"{kind}_time_emb", nn.Parameter(torch.cat([new_past, now], 0).unsqueeze(0)))
        self.memory_size = memory_size
        self.transformer.cfg.memory_size = memory_size
        self.transformer.invalidate()

    def forward(self, video, queries, gt_tracks=None, gt_visibility=None):
        raise NotImplementedError("training (lbm.py:111-209) is out of scope; use inference() / LBM_online")

    @torch.no_grad()
    def inference(self, video: torch.Tensor, queries: torch.Tensor, K: int = 20):
        """lbm.py:211-280 — video (B,T,3,H,W) in [0,255]; queries (B,N,3) = (t, x, y) in video pixels
        (scaled in place, as the reference does).  Returns {'points': (B,T,N,2), 'visibility': (B,T,N) bool}."""
        B, T, _, H, W = video.shape
        N = queries.shape[1]
        dev = video.device
        queries[:, :, 2] = (queries[:, :, 2] / H) * self.size[0]
        queries[:, :, 1] = (queries[:, :, 1] / W) * self.size[1]
        times = queries[:, :, 0].long()
        eng = self.transformer.get_engine()
        M = self.memory_size
        # Queries are sampled on their own frame (backbone.py:130-172).  The frames on which some query starts are
        # encoded first and their token buffers kept (a handful of frames: one host read of the distinct start times);
        # the tracking loop then encodes each remaining frame once into one reused buffer -- O(1) token memory in T
        # (the reference keeps nothing and re-runs its backbone on every query frame).
        q_init = torch.zeros(B, N, self.embed_dim, device=dev)
        starts = [int(v) for v in torch.unique(times).tolist() if 0 <= int(v) < T]
        keep_tokens = len(starts) <= 16
        cache = {}
        for t in starts:
            tok = self.backbone.encode_tokens(video[:, t])
            sel = times == t
            for b in range(B):
                idx = torch.nonzero(sel[b]).flatten()
                if idx.numel():
                    q_init[b, idx] = self.backbone.sample_points(tok[b:b + 1], queries[b:b + 1, idx, 1:].contiguous())[0]
            if keep_tokens:
                ts = self.backbone.last_tokens_split
                cache[t] = (tok, None if ts is None else ts.clone())
        from .engine import state_dtype
        mem_dt = state_dtype(self.transformer.precision)
        stream = torch.zeros(B, N, M, self.embed_dim, device=dev, dtype=mem_dt)
        coll = torch.zeros(B, N, M, self.embed_dim, device=dev, dtype=mem_dt)
        vis_mask = torch.ones(B, N, M, device=dev, dtype=torch.uint8)
        mem_mask = torch.ones(B, N, M, device=dev, dtype=torch.uint8)
        pts, vis = [], []
        buf = None
        for t in range(T):
            if t in cache:
                tok, tok_s = cache.pop(t)
            else:
                tok = buf = self.backbone.encode_tokens(video[:, t], buf)
                tok_s = self.backbone.last_tokens_split
            queried = (times <= t).to(torch.uint8).contiguous()
            out = eng.step(tok, q_init, stream, coll, vis_mask, mem_mask, ring_head=t % M, queried=queried,
                           update_state=True, img_hw=(H, W), vis_threshold=self.visibility_treshold, tokens_split=tok_s)
            pts.append(out["coords"])
            vis.append(out["vis"])
        pts = torch.stack(pts, dim=1)
        vis = torch.stack(vis, dim=1)
        if self.extend_queries:
            pts, vis = pts[:, :, :N], vis[:, :, :N]
        return {"points": pts, "visibility": torch.sigmoid(vis) > self.visibility_treshold}
