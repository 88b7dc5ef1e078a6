"""Feature encoder with the reference's parameter layout, computed on the GPU by HIP kernels.

Mirrors ``lbm/models/lbm/backbone.py`` (Backbone: :10-45 construction, :48-65 ``forward_backbone``,
:130-172 ``sample_queries_online``, :175-193 ``encode_frames_online``).  Differences on purpose:

* no torchvision and **no weight fetch at construction** (reference backbone.py:20 downloads ImageNet
  weights); the ResNet-18 stem/layer1..3 is rebuilt here with identical state-dict keys so ``lbm.pt``
  loads with ``strict=True``;
* the internal feature layout is token-major ``(B, L, C)`` (level 0 in rows ``[0, h*w)``), which is what
  every downstream kernel reads; the ``(B, C, h, w)`` tensors the reference API exposes are zero-copy
  permuted views of it.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import hip
from . import engine


class _BasicBlock(nn.Module):
    """torchvision BasicBlock parameter holder (conv3x3-BN-ReLU-conv3x3-BN + skip)."""

    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, 0, bias=False), nn.BatchNorm2d(cout))
        self.stride = stride


def _stage(cin: int, cout: int, stride: int) -> nn.Sequential:
    return nn.Sequential(_BasicBlock(cin, cout, stride), _BasicBlock(cout, cout, 1))


class Backbone(nn.Module):
    def __init__(self, size, stride, embed_dim):
        super().__init__()
        proj_dim, out_dim = 64, 256
        self.size = list(size)
        self.stride = stride
        self.embedding_dim = embed_dim
        # keys: layer1.0 (conv1) / layer1.1 (bn1) / layer1.4.{0,1} (resnet layer1)  — backbone.py:21-24
        self.layer1 = nn.Sequential(
            nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.MaxPool2d(3, 2, 1), _stage(64, 64, 1))
        self.layer2 = _stage(64, 128, 2)
        self.layer3 = _stage(128, 256, 2)
        self.proj1 = nn.Conv2d(64, proj_dim, 1)
        self.proj2 = nn.Conv2d(128, proj_dim, 1)
        self.proj3 = nn.Conv2d(256, proj_dim, 1)
        self.fusion = nn.Sequential(nn.Conv2d(proj_dim * 3, out_dim, 1), nn.BatchNorm2d(out_dim), nn.GELU())
        self.H_prime = self.size[0] // stride
        self.W_prime = self.size[1] // stride
        self.P = self.H_prime * self.W_prime
        self.frame_pos_embedding = nn.Parameter(torch.zeros(1, embed_dim, self.H_prime, self.W_prime))
        nn.init.trunc_normal_(self.frame_pos_embedding, std=0.02)
        self._packed = None          # engine.EncoderWeights, rebuilt when parameters change
        self.precision = None        # None -> LBM_AMD_PRECISION or "split"; "f32" = exact fp32 MFMA path
        # layout of uint8 frames handed to encode_tokens: "chw" (B,3,H,W) as the reference's float frames, or a video decoder's
        # interleaved bytes "hwc_rgb" / "hwc_bgr" (B,H,W,3) read as they are (demo_utils.py:59-66 converts on the host instead)
        self.frame_layout = "chw"

    # -- weight packing -------------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    def packed(self) -> "engine.EncoderWeights":
        if self._packed is None:
            self._packed = engine.EncoderWeights(self, self.precision)
        return self._packed

    def set_precision(self, precision) -> None:
        self.precision = precision
        self._packed = None

    @property
    def last_tokens_split(self):
        """Split-plane twin of the token buffer produced by the most recent encode_tokens() (None on the fp32 path)."""
        return None if self._packed is None else self._packed.tokens_split

    # -- reference API --------------------------------------------------------------------------
    def encode_tokens(self, frames: torch.Tensor, out: torch.Tensor = None, split_out=None) -> torch.Tensor:
        """frames (B,3,H,W) in [0,255] → pyramid token buffer (B, L, C), level 0 = rows [0, h*w).

        Levels 1..3 (rows h*w..L) are filled too (average pooling, transformer.py:58-69)."""
        return self.packed().encode(frames, out, split_out, u8_layout=self.frame_layout)

    def encode_frames_online(self, frames: torch.Tensor) -> torch.Tensor:
        """backbone.py:175-193 — returns (B, h*w, C) tokens."""
        tok = self.encode_tokens(frames)
        return tok[:, : self.P]

    def forward_backbone(self, x: torch.Tensor) -> torch.Tensor:
        """backbone.py:48-65 — x is the already normalised (B,3,384,512) image; returns (B,C,h,w)
        *without* the positional embedding (as in the reference)."""
        tok = self.packed().encode_normalised(x, add_pos=False)
        B = x.shape[0]
        return tok[:, : self.P].reshape(B, self.H_prime, self.W_prime, -1).permute(0, 3, 1, 2)

    def sample_points(self, tokens: torch.Tensor, pts_net_px: torch.Tensor) -> torch.Tensor:
        """Bilinear, border-padded sample of level-0 tokens (B, L|P, C) at (B, N, 2) network pixels
        (lbm_online.py:72-81; backbone.py:88-92) → (B, N, C)."""
        return hip.sample_queries(tokens, pts_net_px.float().contiguous(), self.H_prime, self.W_prime,
                                  self.size[0], self.size[1])

    def sample_queries_online(self, video: torch.Tensor, queries: torch.Tensor) -> torch.Tensor:
        """backbone.py:130-172 — queries (B,N,3) = (t, x, y) in network pixels."""
        B, T = video.shape[:2]
        N = queries.shape[1]
        out = torch.zeros(B, N, self.embedding_dim, device=queries.device)
        times = queries[:, :, 0]
        for t in range(T):
            sel = times == t
            if not bool(sel.any()):
                continue
            tok = self.encode_tokens(video[:, t])
            n_sel = int(sel[0].sum())
            pts = queries[sel].view(B, n_sel, 3)[:, :, 1:]
            out[sel] = self.sample_points(tok, pts).reshape(-1, self.embedding_dim)
        return out

    def forward(self, video, queries):
        raise NotImplementedError("training forward is out of scope (inference step only)")
