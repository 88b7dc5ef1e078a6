"""CPU restatement of the reference LBM per-frame tracking step.  TEST INFRASTRUCTURE ONLY.

This file is the *oracle*: a plain-PyTorch (fp32, CPU) restatement of the algorithm the
reference runs for one frame step.  It is written functionally (a flat ``{name: tensor}``
state dict in the ``lbm.pt`` layout plus a :class:`Geometry`), not as nn.Modules, and every
function cites the reference file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it; the product path (``lbm_amd``) never does.

Pinning: ``oracle/make_golden.py`` runs this restatement against the importable reference
(`/root/reference/lbm/models/lbm/transformer.py` with ``mmcv.ops`` stubbed by the reference's own
``msdeformattn.py``) and commits the reference's outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` re-checks the restatement against those outputs on every run.
The ResNet-18 body follows the standard torchvision architecture (torchvision is not installed
here, the reference has no tests for it): that sub-block is **parity unpinned**.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


@dataclass
class Geometry:
    """Model constants (reference: lbm/configs/demo.yaml:2-16, lbm.py:15-38)."""
    size: Tuple[int, int] = (384, 512)      # (H, W) network input
    stride: int = 4
    embed_dim: int = 256
    num_heads: int = 4
    num_levels: int = 4
    num_neighbors: int = 9                  # MSDA points of lattice/offset/vis heads
    memory_size: int = 12
    top_k_regions: int = 9
    num_layers: int = 3
    collision_mem_points: int = 4           # transformer.py:44
    vis_threshold: float = 0.8              # lbm.py:17

    @property
    def h(self) -> int:
        return self.size[0] // self.stride

    @property
    def w(self) -> int:
        return self.size[1] // self.stride

    @property
    def topk_per_layer(self) -> List[int]:
        # transformer.py:17
        return [self.top_k_regions] + [4] * (self.num_layers - 2) + [1]

    @property
    def level_shapes(self) -> List[Tuple[int, int]]:
        # transformer.py:21
        return [(self.h >> l, self.w >> l) for l in range(self.num_levels)]

    @property
    def level_starts(self) -> List[int]:
        # transformer.py:22
        out, acc = [], 0
        for (a, b) in self.level_shapes:
            out.append(acc)
            acc += a * b
        return out

    @property
    def num_tokens(self) -> int:
        return sum(a * b for a, b in self.level_shapes)


# ----------------------------------------------------------------------------------------------
# small building blocks
# ----------------------------------------------------------------------------------------------

def linear(sd: SD, name: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def layer_norm(sd: SD, name: str, x: Tensor) -> Tensor:
    # nn.LayerNorm default eps 1e-5 (transformer.py:31,163,...)
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def mlp(sd: SD, name: str, x: Tensor) -> Tensor:
    """transformer.py:343-355 — Linear, exact (erf) GELU, Linear; dropouts are identity in eval."""
    y = linear(sd, name + ".net.0", x)
    y = F.gelu(y)
    return linear(sd, name + ".net.3", y)


def mha(sd: SD, name: str, q: Tensor, k: Tensor, v: Tensor, n_head: int,
        key_padding_mask: Optional[Tensor] = None) -> Tensor:
    """nn.MultiheadAttention(batch_first=True) forward, restated with explicit matmuls.

    q (R, Lq, C), k/v (R, Lk, C), key_padding_mask (R, Lk) bool (True = ignore).
    Call sites: transformer.py:272, 294, 333.
    """
    C = q.shape[-1]
    d = C // n_head
    w, b = sd[name + ".in_proj_weight"], sd[name + ".in_proj_bias"]
    Q = F.linear(q, w[:C], b[:C])
    K = F.linear(k, w[C:2 * C], b[C:2 * C])
    V = F.linear(v, w[2 * C:], b[2 * C:])
    R, Lq, _ = Q.shape
    Lk = K.shape[1]
    Q = Q.view(R, Lq, n_head, d).transpose(1, 2)
    K = K.view(R, Lk, n_head, d).transpose(1, 2)
    V = V.view(R, Lk, n_head, d).transpose(1, 2)
    s = torch.matmul(Q, K.transpose(-1, -2)) / math.sqrt(d)          # R h Lq Lk
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, V).transpose(1, 2).reshape(R, Lq, C)
    return F.linear(o, sd[name + ".out_proj.weight"], sd[name + ".out_proj.bias"])


# ----------------------------------------------------------------------------------------------
# multi-scale deformable attention (the operator seam)
# ----------------------------------------------------------------------------------------------

def msda_core(value: Tensor, shapes: Sequence[Tuple[int, int]], loc: Tensor, attw: Tensor) -> Tensor:
    """msdeformattn.py:188-235 — bilinear, zero-padded, align_corners=False sampling per level.

    value (B, L, Hh, D); loc (B, Q, Hh, Lv, P, 2) in [0,1] (x, y); attw (B, Q, Hh, Lv, P).
    Returns (B, Q, Hh*D).
    """
    B, _, Hh, D = value.shape
    _, Q, _, Lv, P, _ = loc.shape
    grids = 2.0 * loc - 1.0
    start = 0
    sampled = []
    for l, (hl, wl) in enumerate(shapes):
        v = value[:, start:start + hl * wl]                          # B, hw, Hh, D
        start += hl * wl
        v = v.permute(0, 2, 3, 1).reshape(B * Hh, D, hl, wl)
        g = grids[:, :, :, l].permute(0, 2, 1, 3, 4).reshape(B * Hh, Q, P, 2)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    s = torch.stack(sampled, dim=-2).flatten(-2)                     # B*Hh, D, Q, Lv*P
    a = attw.permute(0, 2, 1, 3, 4).reshape(B * Hh, 1, Q, Lv * P)
    out = (s * a).sum(-1).view(B, Hh * D, Q)
    return out.transpose(1, 2).contiguous()


def msda_core_gather(value: Tensor, shapes: Sequence[Tuple[int, int]], loc: Tensor, attw: Tensor) -> Tensor:
    """Same contract as :func:`msda_core`, written as an explicit 4-tap gather (no grid_sample).

    Independent second statement used to cross-check the grid_sample convention
    (pixel = loc * size - 0.5, taps outside the map contribute zero).
    """
    B, _, Hh, D = value.shape
    _, Q, _, Lv, P, _ = loc.shape
    out = torch.zeros(B, Q, Hh, D, dtype=value.dtype)
    start = 0
    bidx = torch.arange(B).view(B, 1, 1, 1)
    hidx = torch.arange(Hh).view(1, 1, Hh, 1)
    for l, (hl, wl) in enumerate(shapes):
        v = value[:, start:start + hl * wl].reshape(B, hl, wl, Hh, D)
        start += hl * wl
        x = loc[:, :, :, l, :, 0] * wl - 0.5                        # B Q Hh P
        y = loc[:, :, :, l, :, 1] * hl - 0.5
        x0 = torch.floor(x)
        y0 = torch.floor(y)
        fx = x - x0
        fy = y - y0
        for dy in (0, 1):
            for dx in (0, 1):
                xi = (x0 + dx).long()
                yi = (y0 + dy).long()
                wt = (fx if dx else 1 - fx) * (fy if dy else 1 - fy)
                ok = (xi >= 0) & (xi < wl) & (yi >= 0) & (yi < hl)
                tap = v[bidx, yi.clamp(0, hl - 1), xi.clamp(0, wl - 1), hidx]   # B Q Hh P D
                wt = torch.where(ok, wt, torch.zeros_like(wt)) * attw[:, :, :, l]
                out += (tap * wt.unsqueeze(-1)).sum(3)
    return out.reshape(B, Q, Hh * D)


def msda(sd: SD, name: str, query: Tensor, value: Tensor, ref_pts: Tensor,
         shapes: Sequence[Tuple[int, int]], n_head: int, n_level: int, n_point: int) -> Tensor:
    """msdeformattn.py:85-185 with batch_first=True, identity=query, 2-d reference points.

    query (B, Q, C); value (B, L, C); ref_pts (B, Q, n_level, 2) normalised (x, y).
    Returns output_proj(sample(value_proj(value))) + query.
    """
    B, Q, C = query.shape
    v = linear(sd, name + ".value_proj", value).view(B, -1, n_head, C // n_head)
    off = linear(sd, name + ".sampling_offsets", query).view(B, Q, n_head, n_level, n_point, 2)
    aw = linear(sd, name + ".attention_weights", query).view(B, Q, n_head, n_level * n_point)
    aw = aw.softmax(-1).view(B, Q, n_head, n_level, n_point)
    norm = torch.tensor([[wl, hl] for (hl, wl) in shapes], dtype=query.dtype)   # (x, y) normaliser
    loc = ref_pts[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    o = msda_core(v, shapes, loc, aw)
    return linear(sd, name + ".output_proj", o) + query


# ----------------------------------------------------------------------------------------------
# tracker transformer
# ----------------------------------------------------------------------------------------------

def pyramid(feat: Tensor) -> List[Tensor]:
    """transformer.py:51-71 — feat (B, C, h, w) → 4 token maps (B, hw/4^l, C)."""
    B, C, _, _ = feat.shape
    maps = [feat] + [F.avg_pool2d(feat, kernel_size=s, stride=s) for s in (2, 4, 8)]
    return [m.reshape(B, C, -1).permute(0, 2, 1) for m in maps]


def indices_to_cell_centres(idx: Tensor, g: Geometry) -> Tensor:
    """coord_utils.py:4-29 — flat cell index → (x, y) pixel centre in network-input pixels."""
    cols = idx % g.w
    rows = idx // g.w
    return torch.stack((cols * g.stride + 0.5 * g.stride, rows * g.stride + 0.5 * g.stride), dim=-1)


def correlation(sd: SD, lname: str, q: Tensor, feats: List[Tensor], g: Geometry) -> Tensor:
    """transformer.py:193-229 — cosine correlation on 4 levels, upsample, 1x1 conv + level-0 residual."""
    B, N, _ = q.shape
    qn = F.normalize(q, p=2, dim=-1)
    maps = []
    for l, f in enumerate(feats):
        hl, wl = g.level_shapes[l]
        c = torch.einsum("bnc,bpc->bnp", qn, F.normalize(f, p=2, dim=-1)).reshape(B * N, 1, hl, wl)
        maps.append(F.interpolate(c, size=(g.h, g.w), mode="bilinear", align_corners=False))
    stack = torch.cat(maps, dim=1)
    out = F.conv2d(stack, sd[lname + ".corr_conv.weight"], sd[lname + ".corr_conv.bias"]) + stack[:, 0:1]
    return out.view(B, N, g.h * g.w)


def norm_ref_points(pts: Tensor, g: Geometry) -> Tensor:
    """transformer.py:309-313 / 107-111 — pixel (x, y) → [0,1], replicated over the 4 levels."""
    B = pts.shape[0]
    n = (pts / torch.tensor([g.size[1], g.size[0]], dtype=pts.dtype)).clamp(0, 1)
    return n.unsqueeze(3).expand(-1, -1, -1, g.num_levels, -1).reshape(B, -1, g.num_levels, 2)


def memory_attention(sd: SD, lname: str, kind: str, x: Tensor, mem: Tensor, ignore: Tensor,
                     valid: Tensor, g: Geometry) -> Tensor:
    """transformer.py:258-302 — one of the two memory read-outs (kind = 'collision' | 'stream').

    x (R, C) point tokens, mem (R, M, C), ignore (R, M) bool, valid (R,) bool.
    Rows with ``valid == False`` are returned unchanged (transformer.py:263, 279, 301).
    """
    temb = sd[f"{lname}.{kind}_time_emb"]                              # 1, M+1, C
    out = x.clone()
    if not bool(valid.any()):
        return out
    xv = x[valid].unsqueeze(1) + temb[:, -1:]                          # time-emb "now" enters the residual
    kv = mem[valid]
    a = mha(sd, f"{lname}.{kind}_attention", xv, kv + temb[:, :-1], kv, g.num_heads, ignore[valid])
    xv = layer_norm(sd, f"{lname}.{kind}_norm", xv + a)
    xv = layer_norm(sd, f"{lname}.{kind}_norm2", xv + mlp(sd, f"{lname}.{kind}_ffn", xv))
    out[valid] = xv.squeeze(1)
    return out


def transformer_layer(sd: SD, lname: str, query: Tensor, feats: List[Tensor], stream: Tensor,
                      collision: Tensor, vis_mask: Tensor, mem_mask: Tensor, k: int, g: Geometry):
    """transformer.py:232-340 (random_mask_ratio = 0)."""
    B, N, C = query.shape
    M = stream.shape[2]
    ignore = (mem_mask | vis_mask).reshape(B * N, M)
    valid = ~ignore.all(dim=-1)
    x = query.reshape(B * N, C)
    x = memory_attention(sd, lname, "collision", x, collision.reshape(B * N, M, C), ignore, valid, g)
    x = memory_attention(sd, lname, "stream", x, stream.reshape(B * N, M, C), ignore, valid, g)
    q_stream = x.view(B, N, C)

    corr = correlation(sd, lname, q_stream, feats, g)
    top_idx = torch.topk(corr, k, dim=-1)[1]
    ref_px = indices_to_cell_centres(top_idx, g)                       # B N k 2
    ref_ms = norm_ref_points(ref_px, g)                                # B N*k 4 2

    feats_cat = torch.cat(feats, dim=1)
    ql = q_stream.unsqueeze(2).expand(-1, -1, k, -1).reshape(B, N * k, C)
    a = msda(sd, lname + ".lattice_attention", ql, feats_cat, ref_ms, g.level_shapes,
             g.num_heads, g.num_levels, g.num_neighbors)
    ql = layer_norm(sd, lname + ".lattice_norm", ql + a)              # = LN(2*q + out_proj(.))
    ql = layer_norm(sd, lname + ".lattice_norm2", ql + mlp(sd, lname + ".lattice_ffn", ql))
    ql = ql.view(B, N, k, C)
    ref_rho = linear(sd, lname + ".reference_rho", ql).squeeze(-1)

    qs = q_stream.reshape(B * N, 1, C)
    lat = ql.reshape(B * N, k, C)
    qs = layer_norm(sd, lname + ".lattice_update_norm", qs + mha(sd, lname + ".lattice_update", qs, lat, lat, g.num_heads))
    qs = layer_norm(sd, lname + ".lattice_update_norm2", qs + mlp(sd, lname + ".lattice_update_ffn", qs))
    return corr, ref_px, ref_rho, qs.view(B, N, C)


def _head_block(sd: SD, name: str, n_point: int, q: Tensor, feats_cat: Tensor, ref_ms: Tensor, g: Geometry) -> Tensor:
    a = msda(sd, f"{name}_layer", q, feats_cat, ref_ms, g.level_shapes, g.num_heads, g.num_levels, n_point)
    x = layer_norm(sd, f"{name}_norm", q + a)
    return layer_norm(sd, f"{name}_norm2", x + mlp(sd, f"{name}_ffn", x))


def transformer_forward(sd: SD, g: Geometry, query: Tensor, feat: Tensor, stream: Tensor, collision: Tensor,
                        vis_mask: Tensor, mem_mask: Tensor, queried: Tensor, prefix: str = "transformer."):
    """transformer.py:74-151.  Returns (out list, memory dict, offset, vis, rho)."""
    sd = _sub(sd, prefix)
    query_init = query.clone()
    feats = pyramid(feat)
    out = []
    ref_px = None
    for i, k in enumerate(g.topk_per_layer):
        corr, ref_px, ref_rho, query = transformer_layer(
            sd, f"layers.{i}", query, feats, stream, collision, vis_mask, mem_mask, k, g)
        out.append({"corr": corr, "reference_points": ref_px, "reference_rho": ref_rho})
    memory = {"stream": query.clone()}

    feats_cat = torch.cat(feats, dim=1)
    ref_ms = norm_ref_points(ref_px, g)
    off_tok = _head_block(sd, "offset", g.num_neighbors, query, feats_cat, ref_ms, g)
    offset = torch.tanh(mlp(sd, "offset_head", off_tok)) * g.stride     # transformer.py:119-120

    pred_ms = norm_ref_points(ref_px + offset.unsqueeze(2), g)          # transformer.py:123-127
    coll = _head_block(sd, "collision_mem", g.collision_mem_points, query, feats_cat, pred_ms, g)
    memory["collision"] = torch.where(queried.unsqueeze(-1), coll, query_init)

    vis_tok = _head_block(sd, "vis", g.num_neighbors, query, feats_cat, ref_ms, g)
    vr = mlp(sd, "vis_head", vis_tok)
    return out, memory, offset, vr[..., 0], vr[..., 1]


def _sub(sd: SD, prefix: str) -> SD:
    if not prefix:
        return sd
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


# ----------------------------------------------------------------------------------------------
# feature encoder (parity unpinned for the ResNet-18 body: torchvision absent, see module docstring)
# ----------------------------------------------------------------------------------------------

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _bn(sd: SD, name: str, x: Tensor) -> Tensor:
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], False, 0.0, 1e-5)


def _basic_block(sd: SD, name: str, x: Tensor, stride: int) -> Tensor:
    y = F.relu(_bn(sd, name + ".bn1", F.conv2d(x, sd[name + ".conv1.weight"], None, stride, 1)))
    y = _bn(sd, name + ".bn2", F.conv2d(y, sd[name + ".conv2.weight"], None, 1, 1))
    if (name + ".downsample.0.weight") in sd:
        x = _bn(sd, name + ".downsample.1", F.conv2d(x, sd[name + ".downsample.0.weight"], None, stride, 0))
    return F.relu(y + x)


def resnet_body(sd: SD, x: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """backbone.py:20-26, 49-51 — torchvision resnet18 conv1..layer3 (eval-mode BN)."""
    y = F.relu(_bn(sd, "layer1.1", F.conv2d(x, sd["layer1.0.weight"], None, 2, 3)))
    y = F.max_pool2d(y, 3, 2, 1)
    y = _basic_block(sd, "layer1.4.0", y, 1)
    c1 = _basic_block(sd, "layer1.4.1", y, 1)
    y = _basic_block(sd, "layer2.0", c1, 2)
    c2 = _basic_block(sd, "layer2.1", y, 1)
    y = _basic_block(sd, "layer3.0", c2, 2)
    c3 = _basic_block(sd, "layer3.1", y, 1)
    return c1, c2, c3


def preprocess(frames: Tensor, size: Tuple[int, int]) -> Tensor:
    """backbone.py:181-183 — /255, bilinear resize (no antialias), ImageNet normalise."""
    x = F.interpolate(frames / 255.0, size=tuple(size), mode="bilinear", align_corners=False)
    mean = torch.tensor(IMAGENET_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=x.dtype).view(1, 3, 1, 1)
    return (x - mean) / std


def encoder_epilogue(sd: SD, c1: Tensor, c2: Tensor, c3: Tensor) -> Tensor:
    """backbone.py:53-65, 187 — 1x1 projections, upsample, concat, fusion conv+BN+GELU, + pos-emb."""
    p1 = F.conv2d(c1, sd["proj1.weight"], sd["proj1.bias"])
    p2 = F.conv2d(c2, sd["proj2.weight"], sd["proj2.bias"])
    p3 = F.conv2d(c3, sd["proj3.weight"], sd["proj3.bias"])
    hw = p1.shape[-2:]
    p2 = F.interpolate(p2, size=hw, mode="bilinear", align_corners=False)
    p3 = F.interpolate(p3, size=hw, mode="bilinear", align_corners=False)
    f = F.conv2d(torch.cat([p1, p2, p3], dim=1), sd["fusion.0.weight"], sd["fusion.0.bias"])
    f = F.gelu(_bn(sd, "fusion.1", f))
    return f + sd["frame_pos_embedding"]


def encode_frames(sd: SD, g: Geometry, frames: Tensor, prefix: str = "backbone.") -> Tensor:
    """backbone.py:175-193 — frames (B,3,H,W) in [0,255] → feature map (B, C, h, w)."""
    sd = _sub(sd, prefix)
    x = preprocess(frames, g.size)
    return encoder_epilogue(sd, *resnet_body(sd, x))


def sample_queries(feat: Tensor, pts_net_px: Tensor, g: Geometry) -> Tensor:
    """lbm_online.py:72-81 / backbone.py:88-92 — bilinear, *border*-padded sample of feat (1,C,h,w)
    at points given in network-input pixels (N, 2) → (N, C)."""
    gx = pts_net_px[:, 0] / g.size[1] * 2 - 1
    gy = pts_net_px[:, 1] / g.size[0] * 2 - 1
    grid = torch.stack([gx, gy], dim=-1).view(1, -1, 1, 2)
    s = F.grid_sample(feat, grid, mode="bilinear", padding_mode="border", align_corners=False)
    return s[0, :, :, 0].t().contiguous()


def resize_time_embeddings(sd: SD, g: Geometry, memory_size: int, prefix: str = "transformer.") -> None:
    """lbm.py:70-87 — linear (align_corners=True) interpolation of the past rows, 'now' row kept."""
    for i in range(g.num_layers):
        for kind in ("collision", "stream"):
            key = f"{prefix}layers.{i}.{kind}_time_emb"
            e = sd[key]
            past = F.interpolate(e[:, :-1].permute(0, 2, 1), size=memory_size, mode="linear", align_corners=True)
            sd[key] = torch.cat([past.permute(0, 2, 1), e[:, -1:]], dim=1)
    g.memory_size = memory_size


# ----------------------------------------------------------------------------------------------
# stateful online wrapper and batched inference loop
# ----------------------------------------------------------------------------------------------

@dataclass
class OnlineState:
    query: Tensor            # (B, N, C)
    collision: Tensor        # (B, N, M, C)
    stream: Tensor           # (B, N, M, C)
    vis_mask: Tensor         # (B, N, M) bool
    mem_mask: Tensor         # (B, N, M) bool
    t: int = 0


def fresh_state(query: Tensor, g: Geometry) -> OnlineState:
    """lbm_online.py:84-88 / lbm.py:233-236 — zero memories, all-True masks."""
    B, N, C = query.shape
    M = g.memory_size
    return OnlineState(query, torch.zeros(B, N, M, C), torch.zeros(B, N, M, C),
                       torch.ones(B, N, M, dtype=torch.bool), torch.ones(B, N, M, dtype=torch.bool))


def roll_state(st: OnlineState, memory: Dict[str, Tensor], vis: Tensor, queried: Tensor, g: Geometry) -> None:
    """lbm_online.py:143-148 / lbm.py:256-259."""
    st.collision = torch.cat([st.collision[:, :, 1:], memory["collision"].unsqueeze(2)], dim=2)
    st.stream = torch.cat([st.stream[:, :, 1:], memory["stream"].unsqueeze(2)], dim=2)
    st.mem_mask = torch.cat([st.mem_mask[:, :, 1:], ~queried.unsqueeze(-1)], dim=2)
    st.vis_mask = torch.cat([st.vis_mask[:, :, 1:], (torch.sigmoid(vis) < g.vis_threshold).unsqueeze(-1)], dim=2)
    st.t += 1


def track_step(sd: SD, g: Geometry, st: OnlineState, feat: Tensor, queried: Optional[Tensor] = None):
    """One transformer step + roll on an already-encoded feature map (B,C,h,w).

    Returns coords in network-input pixels (B,N,2), vis logit (B,N), rho logit (B,N).
    """
    if queried is None:
        queried = torch.ones(st.query.shape[:2], dtype=torch.bool)
    out, memory, offset, vis, rho = transformer_forward(
        sd, g, st.query.clone(), feat, st.stream, st.collision, st.vis_mask, st.mem_mask, queried)
    roll_state(st, memory, vis, queried, g)
    coords = out[-1]["reference_points"].squeeze(2) + offset
    return coords, vis, rho, out


class RefOnline:
    """lbm_online.py:10-158 restated for B = 1 (init / online_forward)."""

    def __init__(self, sd: SD, g: Geometry):
        self.sd, self.g = sd, g
        self.state: Optional[OnlineState] = None
        self.img_size = None

    def init(self, query_points: Tensor, first_frame: Tensor) -> None:
        g = self.g
        self.img_size = first_frame.shape[-2:]
        feat = encode_frames(self.sd, g, first_frame)
        H, W = self.img_size
        pts = query_points.clone().float()
        pts[:, 0] = pts[:, 0] / W * g.size[1]
        pts[:, 1] = pts[:, 1] / H * g.size[0]
        self.state = fresh_state(sample_queries(feat, pts, g).unsqueeze(0), g)

    def online_forward(self, frame: Tensor):
        g = self.g
        H, W = frame.shape[-2:]
        feat = encode_frames(self.sd, g, frame)
        coords, vis, rho, _ = track_step(self.sd, g, self.state, feat)
        c = coords[0].clone()
        c[:, 0] = c[:, 0] / g.size[1] * W
        c[:, 1] = c[:, 1] / g.size[0] * H
        return c, torch.sigmoid(vis[0]) > g.vis_threshold, torch.sigmoid(rho[0]), feat, vis[0], rho[0]

    def update(self, query_points: Tensor, feat: Tensor, update_mask: Tensor, remove_mask: Tensor) -> None:
        """lbm_online.py:48-112 — rows that survive ``remove_mask`` move to the front with their memories; rows flagged
        in ``update_mask`` get a query sampled at their point on ``feat`` (their memories are whatever the row holds:
        fresh for appended rows, the old ones for a re-planted point of a surviving row, lbm_online.py:93-98)."""
        g = self.g
        N = query_points.shape[0]
        n_old = 0 if self.state is None else self.state.query.shape[1]
        if N > 0:
            remove_mask = remove_mask[:n_old]
            n_prev = int((~remove_mask).sum())
            H, W = self.img_size
            pts = query_points[update_mask].clone().float()
            pts[:, 0] = pts[:, 0] / W * g.size[1]
            pts[:, 1] = pts[:, 1] / H * g.size[0]
            q_new = sample_queries(feat, pts, g)
            st = fresh_state(torch.zeros(1, N, g.embed_dim), g)
            if self.state is not None and n_prev > 0:
                old = self.state
                st.query[:, :n_prev] = old.query[:, ~remove_mask]
                st.query[:, update_mask] = q_new
                for name in ("collision", "stream", "vis_mask", "mem_mask"):
                    getattr(st, name)[:, :n_prev] = getattr(old, name)[:, ~remove_mask]
            else:
                st.query = q_new.unsqueeze(0)
            st.t = 0 if self.state is None else self.state.t
            self.state = st
        self.n_points = N

    def online_forward_obj(self, frame: Tensor, bboxes, scores, labels, tracker):
        """lbm_online.py:160-200 — `tracker` is the association object (its own parity is pinned separately against the
        reference's LBMObjTracker outputs, tests/golden/objtrack_*.npz)."""
        coords = vis = feat = inst = None
        n = getattr(self, "n_points", 0 if self.state is None else self.state.query.shape[1])
        if n > 0:
            coords, vis, _, feat, _, _ = self.online_forward(frame)
        if bboxes is not None:
            inst = tracker.track(0 if self.state is None else self.state.t, bboxes, scores, labels, coords, vis)
            memo = tracker.memo
            if n == 0:
                self.img_size = frame.shape[-2:]
                feat = encode_frames(self.sd, self.g, frame)
            self.update(memo[4].reshape(-1, 2), feat, memo[5].flatten(), tracker.points_remove_mask.flatten())
        return coords, vis, inst


def inference(sd: SD, g: Geometry, video: Tensor, queries: Tensor):
    """lbm.py:211-280 — batched loop with per-point query times; queries (B,N,3) = (t, x, y) in
    original pixels.  Returns points (B,T,N,2) in original pixels and visibility logits (B,T,N)."""
    B, T, _, H, W = video.shape
    q = queries.clone().float()
    q[:, :, 2] = q[:, :, 2] / H * g.size[0]
    q[:, :, 1] = q[:, :, 1] / W * g.size[1]
    times = q[:, :, 0].long()
    N = q.shape[1]
    feats = [encode_frames(sd, g, video[:, t]) for t in range(T)]
    q_init = torch.zeros(B, N, g.embed_dim)
    for b in range(B):
        for t in range(T):
            sel = times[b] == t
            if bool(sel.any()):
                q_init[b, sel] = sample_queries(feats[t][b:b + 1], q[b, sel, 1:], g)
    st = fresh_state(q_init, g)
    pts, vis_all = [], []
    for t in range(T):
        coords, vis, _, _ = track_step(sd, g, st, feats[t], times <= t)
        pts.append(coords)
        vis_all.append(vis)
    pts = torch.stack(pts, dim=1)
    pts[..., 0] = pts[..., 0] / g.size[1] * W
    pts[..., 1] = pts[..., 1] / g.size[0] * H
    return pts, torch.stack(vis_all, dim=1)
