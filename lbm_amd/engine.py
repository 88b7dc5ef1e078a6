"""Host side of the tracking step: weight folding/packing and the call into ``lbm_hip_track_step``.

* :class:`EncoderWeights` — BatchNorm folded into the ResNet-18 convolutions (backbone.py:20-26),
  tap-major NHWC weight packing, the fused 1x1 fusion conv (backbone.py:33-37, 53-65).
* :class:`TrackerWeights` — the LBM_W_* table of ``include/lbm_hip.h``: for every nn.MultiheadAttention
  of the reference (transformer.py:161,169,185) the query/key projections are folded into one
  ``[1024,256]`` matrix per block (per head ``Wk_hᵀ Wq_h / sqrt(64)``) and the value/output projections into
  one ``[256,1024]`` matrix, so the kernels never project the M memory slots; the "now" time embedding
  (transformer.py:268,290) is folded into the two biases.
* :class:`TrackEngine` — owns the workspace and fills the step descriptor.

Folding is done once, in float64 on the host, and rounded to fp32.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import torch

from . import hip
from .config import LBMConfig


def _f64(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to("cpu", torch.float64)


# ----------------------------------------------------------------------------------------------
# encoder
# ----------------------------------------------------------------------------------------------

def _fold_bn(conv_w: torch.Tensor, bn) -> (torch.Tensor, torch.Tensor):
    """conv (no bias) followed by eval-mode BatchNorm -> (w', b')."""
    scale = _f64(bn.weight) / torch.sqrt(_f64(bn.running_var) + bn.eps)
    w = _f64(conv_w) * scale.view(-1, 1, 1, 1)
    b = _f64(bn.bias) - _f64(bn.running_mean) * scale
    return w, b


def _pack_conv(w: torch.Tensor, cin_pad: int = 0) -> torch.Tensor:
    """(Cout, Cin, kh, kw) -> (Cout, roundup32(kh*kw*Cin')) with columns ordered (ky, kx, ci)."""
    Cout, Cin, kh, kw = w.shape
    if cin_pad and cin_pad > Cin:
        w = torch.cat([w, torch.zeros(Cout, cin_pad - Cin, kh, kw, dtype=w.dtype)], dim=1)
        Cin = cin_pad
    flat = w.permute(0, 2, 3, 1).reshape(Cout, kh * kw * Cin)
    K = flat.shape[1]
    Kp = (K + 31) // 32 * 32
    if Kp != K:
        flat = torch.cat([flat, torch.zeros(Cout, Kp - K, dtype=w.dtype)], dim=1)
    return flat.contiguous()


# "split": fp32 operands as f16 hi/lo planes, three MFMAs per block (fp32-grade, the default); "f32": exact fp32 MFMA;
# "f16": hi planes only, one MFMA per block (the reduced-precision mode: 1e-2 on logits)
# "f16s": the f16-STORAGE mode -- both memories and the six value maps are f16 in HBM and every product of the tracker except the
#         correlation runs on the hi planes; the encoder and the correlation product + top-k stay fp32-grade (split), so the
#         arg-max is decided on fp32-grade scores (step precision 3; 1e-2 on logits, 0.05 px on coordinates)
PRECISIONS = ("split", "f32", "f16", "f16s")
STEP_PRECISION = {"f32": 0, "split": 1, "f16": 2, "f16s": 3}


def state_dtype(precision: Optional[str]) -> torch.dtype:
    """dtype of the two ring-buffer memories for a tracker precision (None = LBM_AMD_PRECISION or "split")."""
    import os
    precision = precision or os.environ.get("LBM_AMD_PRECISION", "split")
    return torch.float16 if precision == "f16s" else torch.float32


class _Conv:
    def __init__(self, w, b, kh, stride, pad, dev):
        self.w = w.to(dev, torch.float32).contiguous()
        self.b = b.to(dev, torch.float32).contiguous()
        self.ws = hip.SplitWeight(self.w)           # f16 hi/lo planes for the split-precision path
        self.kh, self.stride, self.pad = kh, stride, pad

    def __call__(self, x, relu, res=None):
        """fp32 NHWC in -> fp32 NHWC out on the exact fp32 MFMA."""
        return hip.conv2d_nhwc(x, self.w, self.b, self.kh, self.kh, self.stride, self.pad, relu, res)

    def loader_split(self, x, relu, res=None):
        """fp32 NHWC in (split in the loader) -> fp32 NHWC out on the f16 matrix cores (the 4-channel RGB stem)."""
        return hip.conv2d_nhwc(x, self.ws, self.b, self.kh, self.kh, self.stride, self.pad, relu, res)

    def split(self, xs, relu, res=None, want_f32=True, want_split=True, scratch=None, planes=2):
        """split NHWC in -> (fp32, split) out; im2col rows gathered by LDS-DMA."""
        return hip.conv2d_nhwc_ss(xs, self.ws, self.b, self.kh, self.stride, self.pad, relu, res, want_f32, want_split, scratch,
                                  planes)


class EncoderWeights:
    def __init__(self, bb, precision: Optional[str] = None):
        import os
        precision = precision or os.environ.get("LBM_AMD_PRECISION", "split")
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {PRECISIONS}")
        if precision == "f16s":          # the encoder stays fp32-grade in the f16-storage mode (the correlation reads its tokens)
            precision = "split"
        self.precision = precision
        self.planes = 1 if precision == "f16" else 2
        dev = bb.frame_pos_embedding.device
        if dev.type != "cuda":
            raise hip.HipError("the encoder runs on the GPU only: move the model with .cuda() first")
        self.size = tuple(bb.size)
        self.h, self.w = bb.H_prime, bb.W_prime
        self.P0 = self.h * self.w
        self.L = self.P0 + self.P0 // 4 + self.P0 // 16 + self.P0 // 64
        self.dev = dev
        w, b = _fold_bn(bb.layer1[0].weight, bb.layer1[1])
        self.stem = _Conv(_pack_conv(w, 4), b, 7, 2, 3, dev)
        self.blocks = []
        for stage in (bb.layer1[4], bb.layer2, bb.layer3):
            for blk in stage:
                w1, b1 = _fold_bn(blk.conv1.weight, blk.bn1)
                w2, b2 = _fold_bn(blk.conv2.weight, blk.bn2)
                down = None
                if blk.downsample is not None:
                    wd, bd = _fold_bn(blk.downsample[0].weight, blk.downsample[1])
                    down = _Conv(_pack_conv(wd), bd, 1, blk.stride, 0, dev)
                self.blocks.append((_Conv(_pack_conv(w1), b1, 3, blk.stride, 1, dev), _Conv(_pack_conv(w2), b2, 3, 1, 1, dev), down))
        f32 = lambda t: t.to(dev, torch.float32).contiguous()
        # backbone.py:53-63: proj_i (1x1) -> bilinear up-sampling -> concat -> fusion 1x1 -> BatchNorm -> GELU.  Everything
        # before the GELU is linear and the bilinear weights sum to one, so each level is multiplied by
        # (BN o fusion)[:, 64 i : 64 i + 64] . proj_i at its OWN resolution (folded in float64) and the three 256-channel
        # maps are summed after up-sampling (lbm_hip_encoder_tail): no 64-channel intermediates, no 192-channel concat,
        # 12288 x 256 x 192 of fusion flops become 12288 x 256 x 64 + two small products.
        fw, fb = _fold_bn(bb.fusion[0].weight, bb.fusion[1])
        scale = _f64(bb.fusion[1].weight) / torch.sqrt(_f64(bb.fusion[1].running_var) + bb.fusion[1].eps)
        fb = fb + _f64(bb.fusion[0].bias) * scale
        fw = _f64(fw).flatten(1)                                            # (256, 192)
        self.lvl_w, bias = [], _f64(fb).clone()
        for i, p in enumerate((bb.proj1, bb.proj2, bb.proj3)):
            blk = fw[:, 64 * i: 64 * (i + 1)]                               # (256, 64)
            self.lvl_w.append(f32(blk @ _f64(p.weight).flatten(1)))        # (256, C_i)
            bias = bias + blk @ _f64(p.bias)
        self.lvl_b = f32(bias)
        self.lvl_ws = [hip.SplitWeight(wt) for wt in self.lvl_w]
        self._scratch = None
        self.tokens_split: Optional[hip.SplitTensor] = None     # split twin of the last encode() output
        self._own_split: Optional[hip.SplitTensor] = None
        self._split_target: Optional[hip.SplitTensor] = None
        # positional embedding, token-major (P0, 256)
        self.pos = f32(_f64(bb.frame_pos_embedding)[0].permute(1, 2, 0).reshape(self.P0, -1))

    @property
    def _fused_stem(self) -> bool:
        """conv1 + bn1 + relu + maxpool as ONE launch over split input planes (stem_rs.hip); LBM_AMD_FUSED_STEM=0: the two kernels."""
        import os
        return (self.precision == "split" and self.size[0] % 4 == 0 and self.size[1] % 4 == 0 and
                os.environ.get("LBM_AMD_FUSED_STEM", "1") != "0")

    def encode(self, frames: torch.Tensor, out: Optional[torch.Tensor] = None,
               split_out: Optional["hip.SplitTensor"] = None, u8_layout: str = "chw") -> torch.Tensor:
        """split_out: caller-owned split twin to fill instead of the encoder's own (double-buffered pipelines).
        u8_layout: how a uint8 frame is laid out ("chw" (B,3,H,W); "hwc_rgb" / "hwc_bgr" (B,H,W,3), a video decoder's bytes as
        they are); float frames are always (B,3,H,W)."""
        self._split_target = split_out
        if frames.dtype != torch.uint8:
            u8_layout = "chw"
        if self._fused_stem:
            return self._body_split(hip.preprocess_split(frames, self.size[0], self.size[1], u8_layout), True, out)
        x = hip.preprocess(frames, self.size[0], self.size[1], layout=4, u8_layout=u8_layout)
        return self._body(x, True, out)

    def encode_normalised(self, x_nchw: torch.Tensor, add_pos: bool = True) -> torch.Tensor:
        B = x_nchw.shape[0]
        x = torch.zeros(B, self.size[0], self.size[1], 4, device=self.dev)
        x[..., :3] = x_nchw.permute(0, 2, 3, 1)
        return self._body(x, add_pos, None)

    def _body(self, x: torch.Tensor, add_pos: bool, out: Optional[torch.Tensor]) -> torch.Tensor:
        if self.precision in ("split", "f16"):
            return self._body_split(x, add_pos, out)
        self.tokens_split = None
        B = x.shape[0]
        y = self.stem(x, True)
        y = hip.maxpool3x3s2_nhwc(y)
        feats = []
        for i, (c1, c2, down) in enumerate(self.blocks):
            idn = y if down is None else down(y, False)
            t = c1(y, True)
            y = c2(t, True, idn)
            if i % 2 == 1:
                feats.append(y)
        qs = [hip.linear(f.view(-1, f.shape[-1]), wt, self.lvl_b if i == 0 else None) for i, (f, wt) in enumerate(zip(feats, self.lvl_w))]
        if out is None:
            out = torch.empty(B, self.L, 256, device=self.dev, dtype=torch.float32)
        hip.encoder_tail(qs[0], qs[1], qs[2], self.pos if add_pos else None, out, None, self.h, self.w)
        return out


    def _body_split(self, x: torch.Tensor, add_pos: bool, out: Optional[torch.Tensor]) -> torch.Tensor:
        """Same network with every contraction on the f16 matrix cores (split operands, fp32-grade)."""
        B = x.shape[0]
        if self._scratch is None:
            self._scratch = torch.empty(32 << 20, device=self.dev, dtype=torch.uint8)    # K-split slabs of small convolutions
        if self._fused_stem:
            y, ys = hip.stem_pool(x if isinstance(x, hip.SplitTensor) else hip.split_tensor(x), self.stem.ws, self.stem.b)
        else:
            y = self.stem.loader_split(x, True)
            y, ys = hip.maxpool3x3s2_nhwc_s(y)
        feats = []
        for i, (c1, c2, down) in enumerate(self.blocks):
            idn = y if down is None else down.split(ys, False, want_split=False, scratch=self._scratch, planes=self.planes)[0]
            _, ts = c1.split(ys, True, want_f32=False, scratch=self._scratch, planes=self.planes)
            # the fp32 copy of a block's output is only read as the next block's identity residual
            nxt = self.blocks[i + 1] if i + 1 < len(self.blocks) else None
            y, ys = c2.split(ts, True, idn, want_f32=nxt is not None and nxt[2] is None, scratch=self._scratch, planes=self.planes)
            if i % 2 == 1:
                feats.append(ys)
        fused_q1 = self.planes == 2 and os.environ.get("LBM_AMD_FUSED_Q1", "1") != "0"     # level-0 product inside the tail kernel
        qs = [None if (i == 0 and fused_q1) else
              hip.linear_ss(f.reshape(-1, f.shape[-1]), wt, self.lvl_b if i == 0 else None, planes=self.planes)[0]
              for i, (f, wt) in enumerate(zip(feats, self.lvl_ws))]
        if out is None:
            out = torch.empty(B, self.L, 256, device=self.dev, dtype=torch.float32)
        target, self._split_target = self._split_target, None
        if target is None:      # the encoder's own twin; a caller-owned one (pipelined slots) is never reused for another frame
            if self._own_split is None or self._own_split.shape != (B, self.L, 256):
                self._own_split = hip.SplitTensor((B, self.L, 256), self.dev)
            target = self._own_split
        self.tokens_split = target
        if fused_q1:
            # the tail also leaves F.normalize's factor of every token row beside the split twin (engine.step hands it to the step)
            if target.inv_norm is None or target.inv_norm.shape != (B, self.L):
                target.inv_norm = torch.empty(B, self.L, device=self.dev, dtype=torch.float32)
            hip.encoder_tail_fused(feats[0], self.lvl_ws[0], self.lvl_b, qs[1], qs[2], self.pos if add_pos else None, out, target,
                                   self.h, self.w, target.inv_norm)
        else:
            target.inv_norm = None
            hip.encoder_tail(qs[0], qs[1], qs[2], self.pos if add_pos else None, out, target, self.h, self.w)
        return out


# ----------------------------------------------------------------------------------------------
# tracker transformer
# ----------------------------------------------------------------------------------------------

def fold_attention(mha, e_now: Optional[torch.Tensor], n_head: int = 4):
    """nn.MultiheadAttention parameters -> (G [4C,C], gb [4C], Hm [C,4C], hb [C]) in float64.

    u_h = G_h x + gb_h reproduces Wk_hᵀ(Wq_h (x + e_now) + bq_h)/sqrt(d); out = Hm · concat_h(m̄_h) + hb
    reproduces out_proj(concat_h(Wv_h m̄_h + bv_h)) + e_now (the residual carries x + e_now)."""
    Wi, bi = _f64(mha.in_proj_weight), _f64(mha.in_proj_bias)
    Wo, bo = _f64(mha.out_proj.weight), _f64(mha.out_proj.bias)
    Cc = Wo.shape[0]
    d = Cc // n_head
    Wq, Wk, Wv = Wi[:Cc], Wi[Cc:2 * Cc], Wi[2 * Cc:]
    bq, bv = bi[:Cc], bi[2 * Cc:]
    e = torch.zeros(Cc, dtype=torch.float64) if e_now is None else _f64(e_now).reshape(Cc)
    G, gb, Hm = [], [], []
    for h in range(n_head):
        sl = slice(h * d, (h + 1) * d)
        Gh = Wk[sl].t() @ Wq[sl] / (d ** 0.5)
        G.append(Gh)
        gb.append(Gh @ e + Wk[sl].t() @ bq[sl] / (d ** 0.5))
        Hm.append(Wo[:, sl] @ Wv[sl])
    hb = Wo @ bv + bo + e
    return torch.cat(G, 0), torch.cat(gb, 0), torch.cat(Hm, 1), hb


class TrackerWeights:
    """Device-resident LBM_W_* pointer table for one LBMTransformer (keeps the packed tensors alive)."""

    def __init__(self, tr, cfg: LBMConfig):
        cfg.validate_for_hip()
        self.cfg = cfg
        k = hip.header_constants()
        dev = tr.spatial_shapes.device
        if dev.type != "cuda":
            raise hip.HipError("the tracker runs on the GPU only: move the model with .cuda() first")
        n_layers = len(tr.layers)
        n = n_layers * k["LBM_W_LAYER_COUNT"] + k["LBM_W_HEAD_COUNT"]
        self.tensors: List[Optional[torch.Tensor]] = [None] * n
        self.split: List[Optional[hip.SplitWeight]] = [None] * n      # f16 hi/lo planes of the matrices (precision 1)
        self.memory_size = tr.layers[0].collision_time_emb.shape[1] - 1
        keep_f32 = {"LBM_W_ATTN_TEMB", "LBM_W_LAYER_RHO_W", "LBM_W_HEAD_OFFH_W3", "LBM_W_HEAD_VISH_W3"}
        f32_idx = set()

        def put(idx, t):
            self.tensors[idx] = t.detach().to(dev, torch.float32).contiguous()

        def put_attn(base, mha, temb, ln1, ffn, ln2):
            e_now = None if temb is None else temb[0, -1]
            G, gb, Hm, hb = fold_attention(mha, e_now, cfg.num_heads)
            put(base + k["LBM_W_ATTN_G"], G)
            put(base + k["LBM_W_ATTN_GB"], gb)
            if temb is not None:
                put(base + k["LBM_W_ATTN_TEMB"], temb[0, :-1])
            put(base + k["LBM_W_ATTN_HM"], Hm)
            put(base + k["LBM_W_ATTN_HB"], hb)
            put(base + k["LBM_W_ATTN_HMSUM"], Hm.double().view(Hm.shape[0], cfg.num_heads, -1).sum(1).float())
            put(base + k["LBM_W_ATTN_LN1G"], ln1.weight)
            put(base + k["LBM_W_ATTN_LN1B"], ln1.bias)
            put(base + k["LBM_W_ATTN_F1W"], ffn.net[0].weight)
            put(base + k["LBM_W_ATTN_F1B"], ffn.net[0].bias)
            put(base + k["LBM_W_ATTN_F2W"], ffn.net[3].weight)
            put(base + k["LBM_W_ATTN_F2B"], ffn.net[3].bias)
            put(base + k["LBM_W_ATTN_LN2G"], ln2.weight)
            put(base + k["LBM_W_ATTN_LN2B"], ln2.bias)

        def put_msda(base, att, ln1, ffn, ln2):
            put(base + k["LBM_W_MSDA_OLW"], torch.cat([att.sampling_offsets.weight, att.attention_weights.weight], 0))
            put(base + k["LBM_W_MSDA_OLB"], torch.cat([att.sampling_offsets.bias, att.attention_weights.bias], 0))
            put(base + k["LBM_W_MSDA_OUTW"], att.output_proj.weight)
            put(base + k["LBM_W_MSDA_OUTB"], att.output_proj.bias)
            put(base + k["LBM_W_MSDA_LN1G"], ln1.weight)
            put(base + k["LBM_W_MSDA_LN1B"], ln1.bias)
            put(base + k["LBM_W_MSDA_F1W"], ffn.net[0].weight)
            put(base + k["LBM_W_MSDA_F1B"], ffn.net[0].bias)
            put(base + k["LBM_W_MSDA_F2W"], ffn.net[3].weight)
            put(base + k["LBM_W_MSDA_F2B"], ffn.net[3].bias)
            put(base + k["LBM_W_MSDA_LN2G"], ln2.weight)
            put(base + k["LBM_W_MSDA_LN2B"], ln2.bias)

        self.corr_mix: List[float] = []
        self.corr_bias: List[float] = []
        vw, vb = [], []
        for i, ly in enumerate(tr.layers):
            base = i * k["LBM_W_LAYER_COUNT"]
            put_attn(base + k["LBM_W_LAYER_COLL"], ly.collision_attention, ly.collision_time_emb, ly.collision_norm,
                     ly.collision_ffn, ly.collision_norm2)
            put_attn(base + k["LBM_W_LAYER_STREAM"], ly.stream_attention, ly.stream_time_emb, ly.stream_norm,
                     ly.stream_ffn, ly.stream_norm2)
            put_msda(base + k["LBM_W_LAYER_LATTICE"], ly.lattice_attention, ly.lattice_norm, ly.lattice_ffn, ly.lattice_norm2)
            put_attn(base + k["LBM_W_LAYER_UPDATE"], ly.lattice_update, None, ly.lattice_update_norm,
                     ly.lattice_update_ffn, ly.lattice_update_norm2)
            put(base + k["LBM_W_LAYER_RHO_W"], ly.reference_rho.weight)
            put(base + k["LBM_W_LAYER_RHO_B"], ly.reference_rho.bias)
            # products that share their input are merged: lattice offsets/weights and the update attention's queries
            lat, upd = base + k["LBM_W_LAYER_LATTICE"], base + k["LBM_W_LAYER_UPDATE"]
            put(base + k["LBM_W_LAYER_OLG_W"], torch.cat([self.tensors[lat + k["LBM_W_MSDA_OLW"]], self.tensors[upd + k["LBM_W_ATTN_G"]]], 0))
            put(base + k["LBM_W_LAYER_OLG_B"], torch.cat([self.tensors[lat + k["LBM_W_MSDA_OLB"]], self.tensors[upd + k["LBM_W_ATTN_GB"]]], 0))
            self.corr_mix += [float(v) for v in ly.corr_conv.weight.detach().flatten().tolist()]
            self.corr_bias.append(float(ly.corr_conv.bias.detach().item()))
            vw.append(ly.lattice_attention.value_proj.weight)
            vb.append(ly.lattice_attention.value_proj.bias)
        hb_ = n_layers * k["LBM_W_LAYER_COUNT"]
        put_msda(hb_ + k["LBM_W_HEAD_OFFSET"], tr.offset_layer, tr.offset_norm, tr.offset_ffn, tr.offset_norm2)
        put_msda(hb_ + k["LBM_W_HEAD_CMEM"], tr.collision_mem_layer, tr.collision_mem_norm, tr.collision_mem_ffn,
                 tr.collision_mem_norm2)
        put_msda(hb_ + k["LBM_W_HEAD_VIS"], tr.vis_layer, tr.vis_norm, tr.vis_ffn, tr.vis_norm2)
        for name, head in (("OFFH", tr.offset_head), ("VISH", tr.vis_head)):
            put(hb_ + k[f"LBM_W_HEAD_{name}_W0"], head.net[0].weight)
            put(hb_ + k[f"LBM_W_HEAD_{name}_B0"], head.net[0].bias)
            put(hb_ + k[f"LBM_W_HEAD_{name}_W3"], head.net[3].weight)
            put(hb_ + k[f"LBM_W_HEAD_{name}_B3"], head.net[3].bias)
        for att in (tr.offset_layer, tr.collision_mem_layer, tr.vis_layer):
            vw.append(att.value_proj.weight)
            vb.append(att.value_proj.bias)
        heads3 = [hb_ + k[n] for n in ("LBM_W_HEAD_OFFSET", "LBM_W_HEAD_CMEM", "LBM_W_HEAD_VIS")]
        put(hb_ + k["LBM_W_HEAD_OL3_W"], torch.cat([self.tensors[h + k["LBM_W_MSDA_OLW"]] for h in heads3], 0))
        put(hb_ + k["LBM_W_HEAD_OL3_B"], torch.cat([self.tensors[h + k["LBM_W_MSDA_OLB"]] for h in heads3], 0))
        put(hb_ + k["LBM_W_HEAD_VPROJ_W"], torch.cat(vw, 0))
        put(hb_ + k["LBM_W_HEAD_VPROJ_B"], torch.cat(vb, 0))
        # which table entries are matrices of a dense contraction (they get split planes)
        for i in range(n_layers):
            base = i * k["LBM_W_LAYER_COUNT"]
            f32_idx.add(base + k["LBM_W_LAYER_RHO_W"])
            for blk in ("LBM_W_LAYER_COLL", "LBM_W_LAYER_STREAM", "LBM_W_LAYER_UPDATE"):
                f32_idx.add(base + k[blk] + k["LBM_W_ATTN_TEMB"])
        f32_idx.add(hb_ + k["LBM_W_HEAD_OFFH_W3"])
        f32_idx.add(hb_ + k["LBM_W_HEAD_VISH_W3"])
        for i, t in enumerate(self.tensors):
            if t is not None and t.dim() == 2 and i not in f32_idx:
                self.split[i] = hip.SplitWeight(t)
        # fragment-major copies of the feed-forward matrices for the fused few-rows kernel (lbm_hip_ffn_slabs)
        blocks = [(i * k["LBM_W_LAYER_COUNT"] + k[b], "ATTN") for i in range(n_layers) for b in ("LBM_W_LAYER_COLL", "LBM_W_LAYER_STREAM", "LBM_W_LAYER_UPDATE")]
        blocks += [(i * k["LBM_W_LAYER_COUNT"] + k["LBM_W_LAYER_LATTICE"], "MSDA") for i in range(n_layers)]
        blocks += [(hb_ + k[b], "MSDA") for b in ("LBM_W_HEAD_OFFSET", "LBM_W_HEAD_CMEM", "LBM_W_HEAD_VIS")]
        for base, kind in blocks:
            for src, dst in (("F1W", "F1FM"), ("F2W", "F2FM")):
                sw = self.split[base + k[f"LBM_W_{kind}_{src}"]]
                if sw is not None and sw.shape in ((1024, 256), (256, 1024)):
                    self.tensors[base + k[f"LBM_W_{kind}_{dst}"]] = hip.fragment_major(sw)
        self.n = n
        self.n_layers = n_layers
        self.device = dev
        self.build_tables()

    def build_tables(self) -> None:
        """The three pointer tables of the step descriptor (fp32, hi planes, lo planes) over self.tensors / self.split."""
        n = self.n
        self.table = (C.c_void_p * n)(*[None if t is None else t.data_ptr() for t in self.tensors])
        self.table_hi = (C.c_void_p * n)(*[(None if t is None else t.data_ptr()) if self.split[i] is None else self.split[i].hi
                                           for i, t in enumerate(self.tensors)])
        self.table_lo = (C.c_void_p * n)(*[None if sw is None else sw.lo for sw in self.split])


class TrackEngine:
    """Fills ``lbm_hip_step_desc`` and enqueues one tracking step on the current stream."""

    def __init__(self, cfg: LBMConfig, weights: TrackerWeights, precision: Optional[str] = None):
        """precision: "split" (default; f16 matrix cores with split operands, fp32-grade) or "f32" (exact fp32 MFMA);
        the environment variable LBM_AMD_PRECISION overrides the default."""
        import os
        precision = precision or os.environ.get("LBM_AMD_PRECISION", "split")
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {PRECISIONS}")
        self.precision = precision
        # optional point-parallel sub-streams (B == 1): the per-point chain of up to 1 + n_aux row ranges is enqueued on
        # concurrent HIP streams.  Off by default: measured on MI355X it does not shorten the step (host-bound eagerly,
        # serialised inside a graph replay); kept as a knob (LBM_AMD_AUX_STREAMS=1..3) and covered by a parity test.
        self.n_aux = max(0, min(3, int(os.environ.get("LBM_AMD_AUX_STREAMS", "0"))))
        self._aux: List[torch.cuda.Stream] = []
        self.cfg = cfg
        self.w = weights
        self.dev = weights.device
        self._qc = None                               # what is cached for the last query tensor seen (see step())
        self._cache_u0 = os.environ.get("LBM_AMD_CACHE_U0", "1") != "0"
        self._ws: Dict[int, torch.Tensor] = {}       # workspace per slot (a pipelined caller alternates two)
        self._ws_key: Dict[int, tuple] = {}

    def _desc(self, B: int, N: int, M: int) -> hip.StepDesc:
        cfg = self.cfg
        d = hip.StepDesc()
        d.B, d.N, d.M, d.n_layers = B, N, M, self.w.n_layers
        for i, kk in enumerate(cfg.topk_per_layer):
            d.topk[i] = kk
        d.net_h, d.net_w, d.stride = cfg.input_size[0], cfg.input_size[1], cfg.stride
        d.n_points, d.n_points_cmem = cfg.num_neighbors, 4
        d.vis_threshold = 0.8
        for i, v in enumerate(self.w.corr_mix):
            d.corr_mix[i] = v
        for i, v in enumerate(self.w.corr_bias):
            d.corr_bias[i] = v
        if self.precision in ("split", "f16", "f16s"):
            d.precision = STEP_PRECISION[self.precision]
            d.weights = C.cast(self.w.table_hi, C.POINTER(C.c_void_p))
            d.weights_lo = C.cast(self.w.table_lo, C.POINTER(C.c_void_p))
        else:
            d.precision = 0
            d.weights = C.cast(self.w.table, C.POINTER(C.c_void_p))
        d.n_weights = self.w.n
        return d

    def workspace(self, d: hip.StepDesc, slot: int = 0) -> torch.Tensor:
        key = (d.B, d.N, d.M, d.precision, bool(d.tokens_split))
        if self._ws.get(slot) is None or self._ws_key.get(slot) != key:
            nbytes = int(hip.load().lbm_hip_step_workspace_bytes(C.byref(d)))
            if nbytes == 0:
                raise hip.HipError("step workspace: " + hip.load().lbm_hip_last_error().decode())
            self._ws[slot] = None
            self._ws[slot] = torch.empty(nbytes, device=self.dev, dtype=torch.uint8)
            if os.environ.get("LBM_AMD_WS_FILL"):      # debugging aid: a byte pattern (255 = NaNs) exposes reads of unwritten scratch
                self._ws[slot].fill_(int(os.environ["LBM_AMD_WS_FILL"]))
            self._ws_key[slot] = key
        self._last_slot = slot
        return self._ws[slot]

    def step(self, tokens: torch.Tensor, query: torch.Tensor, stream_mem: torch.Tensor, coll_mem: torch.Tensor,
             vis_mask: torch.Tensor, mem_mask: torch.Tensor, ring_head: int = 0, queried: Optional[torch.Tensor] = None,
             update_state: bool = True, img_hw: Optional[Sequence[int]] = None, want_memory: bool = False,
             want_aux: bool = False, want_corr: bool = False, vis_threshold: float = 0.8,
             tokens_split: Optional["hip.SplitTensor"] = None, phase: int = 0, rows: Optional[Sequence[int]] = None,
             sub_index: int = 0, out: Optional[Dict[str, torch.Tensor]] = None, ws_slot: int = 0) -> Dict[str, torch.Tensor]:
        """tokens (B,L,256) pyramid; query (B,N,256); memories (B,N,M,256) in ring order; masks (B,N,M) bool/uint8.

        phase 0: the whole step.  phase 1: only the point-independent part (value maps, correlation maps) into the
        workspace of ``ws_slot``; phase 2: only the per-point part, reading that workspace.  A pipelined caller runs
        phase 1 of frame t+1 on a second stream (and second slot) while phase 2 of frame t is in flight."""
        hip._require_gpu()
        if self.w.memory_size != stream_mem.shape[2]:
            raise hip.HipError(f"memory has {stream_mem.shape[2]} slots but the time embeddings were built for "
                               f"{self.w.memory_size}; call set_memory_size first")
        B, N, M = stream_mem.shape[:3]
        cfg = self.cfg
        mem_dt = state_dtype(self.precision)
        if stream_mem.dtype != mem_dt or coll_mem.dtype != mem_dt:
            raise hip.HipError(f"precision {self.precision!r} keeps the two memories as {mem_dt}, got {stream_mem.dtype} / {coll_mem.dtype}")
        d = self._desc(B, N, M)
        d.vis_threshold = vis_threshold
        d.ring_head, d.update_state = ring_head, 1 if update_state else 0
        ih, iw = img_hw if img_hw is not None else cfg.input_size
        d.img_h, d.img_w = float(ih), float(iw)
        d.inv_norm = None
        if tokens_split is not None and d.precision >= 1:
            if tokens_split.shape != tuple(tokens.shape):
                raise hip.HipError("tokens_split does not match tokens")
            d.tokens_split = tokens_split.hi
            inv = getattr(tokens_split, "inv_norm", None)
            d.inv_norm = inv.data_ptr() if inv is not None and os.environ.get("LBM_AMD_TAIL_NORMS", "1") != "0" else None
        ws = self.workspace(d, ws_slot)
        dev = self.dev
        d.phase, d.sub_index = phase, sub_index
        if phase == 2 and rows is None:
            rows = (0, B * N)
        if rows is not None:
            d.row_lo, d.row_count = int(rows[0]), int(rows[1])
        if out is None:
            out = {"coords": torch.empty(B, N, 2, device=dev), "vis": torch.empty(B, N, device=dev),
                   "rho": torch.empty(B, N, device=dev), "offset": torch.empty(B, N, 2, device=dev),
                   "vis_flag": torch.empty(B, N, device=dev, dtype=torch.bool), "rho_prob": torch.empty(B, N, device=dev)}
        if "vis_flag" in out:
            d.vis_flag, d.rho_prob = out["vis_flag"].data_ptr(), out["rho_prob"].data_ptr()
        if N > 0 and phase != 1:
            # The query only changes at point-set updates: what depends on it alone is kept across frames -- its split twin and
            # the first layer's collision-attention queries G . query (filled by the first step that sees this query, read by
            # the following ones).  The cache holds the tensor itself, so "the same query" means the same object at the same
            # version: its memory cannot have been handed to another tensor in between.
            qc = self._qc
            qver = hip.tensor_version(query)
            if qc is None or qc["ref"] is not query or qc["ver"] != qver or qc["prec"] != d.precision:
                qc = self._qc = {"ref": query, "ver": qver, "prec": d.precision, "split": None, "u0": None, "u0_valid": 0}
            if d.precision >= 1:
                if qc["split"] is None:
                    qs = hip.SplitTensor(tuple(query.shape), dev)
                    hip._check(hip.load().lbm_hip_split_planes(hip._p(query), qs.hi, qs.planes[1].data_ptr(), query.numel(), hip._stream()),
                               "lbm_hip_split_planes")
                    qc["split"] = qs
                d.query_split = qc["split"].hi
            if self._cache_u0 and (rows is None or (int(rows[0]) == 0 and int(rows[1]) == B * N)):   # whole point set only
                if qc["u0"] is None:
                    qc["u0"] = torch.empty(B, N, 1024, device=dev)
                d.query_u0, d.query_u0_valid = qc["u0"].data_ptr(), qc["u0_valid"]
                qc["u0_valid"] = 1
        if want_memory:
            out["stream"] = torch.empty(B, N, 256, device=dev)
            out["collision"] = torch.empty(B, N, 256, device=dev)
            d.stream_out, d.coll_out = out["stream"].data_ptr(), out["collision"].data_ptr()
        P0 = cfg.num_cells
        for i, kk in enumerate(cfg.topk_per_layer):
            last = i == len(cfg.topk_per_layer) - 1
            if want_aux or last:
                if f"ref{i}" not in out:
                    out[f"ref{i}"] = torch.empty(B, N, kk, 2, device=dev)
                d.ref_points[i] = out[f"ref{i}"].data_ptr()
            if want_aux:
                out[f"idx{i}"] = torch.empty(B, N, kk, device=dev, dtype=torch.int32)
                out[f"rho{i}"] = torch.empty(B, N, kk, device=dev)
                d.topk_idx[i] = out[f"idx{i}"].data_ptr()
                d.ref_rho[i] = out[f"rho{i}"].data_ptr()
            if want_corr:
                out[f"corr{i}"] = torch.empty(B, N, P0, device=dev)
                d.corr[i] = out[f"corr{i}"].data_ptr()
        d.tokens, d.query = hip._p(tokens), hip._p(query)
        d.stream_mem, d.coll_mem = hip._p(stream_mem, mem_dt), hip._p(coll_mem, mem_dt)
        d.vis_mask, d.mem_mask = hip._p(vis_mask, torch.uint8), hip._p(mem_mask, torch.uint8)
        d.queried = hip._p(queried, torch.uint8)
        d.coords, d.vis, d.rho, d.offset = (out[n].data_ptr() for n in ("coords", "vis", "rho", "offset"))
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        if self.n_aux and B == 1 and phase == 0:
            while len(self._aux) < self.n_aux:
                self._aux.append(torch.cuda.Stream(device=dev))
            for i in range(self.n_aux):
                d.aux_streams[i] = self._aux[i].cuda_stream
            d.n_aux_streams = self.n_aux
        hip._check(hip.load().lbm_hip_track_step(C.byref(d), hip._stream()), "lbm_hip_track_step")
        if want_corr:   # the kernels rank without the (constant) bias; the returned map carries it
            for i in range(len(cfg.topk_per_layer)):
                out[f"corr{i}"] += self.w.corr_bias[i]
        self._last_desc = d
        self._last_out = out
        return out

    def debug_buffer(self, name: str, shape, dtype=torch.float32) -> torch.Tensor:
        """View of a named workspace intermediate after a step (parity debugging)."""
        off = int(hip.load().lbm_hip_step_workspace_offset(C.byref(self._last_desc), name.encode()))
        if off < 0:
            raise KeyError(name)
        n = int(torch.tensor(shape).prod()) * torch.empty((), dtype=dtype).element_size()
        return self._ws[self._last_slot][off:off + n].view(dtype).view(*shape)
