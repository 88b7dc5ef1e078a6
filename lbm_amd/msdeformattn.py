"""Multi-scale deformable attention — the reference's operator seam, served by HIP kernels.

Mirrors ``mmcv.ops.MultiScaleDeformableAttention`` / the reference's in-tree pure-PyTorch variant
(reference: lbm/models/lbm/msdeformattn.py:9-185, 188-235): same constructor arguments, same
parameter names (``sampling_offsets``, ``attention_weights``, ``value_proj``, ``output_proj``),
same ``forward`` keyword arguments and the same ``output_proj(sample(value_proj(value))) + identity``
result.  All arithmetic runs in ``liblbm_hip.so``; there is no PyTorch fallback.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import hip


def multi_scale_deformable_attn(value: torch.Tensor, value_spatial_shapes: torch.Tensor,
                                sampling_locations: torch.Tensor,
                                attention_weights: torch.Tensor) -> torch.Tensor:
    """Drop-in for ``multi_scale_deformable_attn_pytorch`` (msdeformattn.py:188-235).

    value (B, L, 4, 64) fp32; sampling_locations (B, Q, 4, n_levels, P, 2) in [0, 1];
    attention_weights (B, Q, 4, n_levels, P) (already soft-maxed).  Returns (B, Q, 256).
    """
    B, L, Hh, D = value.shape
    _, Q, _, Lv, P, _ = sampling_locations.shape
    shapes = [(int(a), int(b)) for a, b in value_spatial_shapes.tolist()]
    out = torch.empty(B, Q, Hh * D, device=value.device, dtype=torch.float32)
    hip.msda_sample_loc(value.contiguous(), shapes, sampling_locations.contiguous(),
                        attention_weights.contiguous(), out)
    return out


# alias under the reference's function name
multi_scale_deformable_attn_pytorch = multi_scale_deformable_attn


class MultiScaleDeformableAttention(nn.Module):
    def __init__(self, embed_dims: int = 256, num_heads: int = 8, num_levels: int = 4,
                 num_points: int = 4, dropout: float = 0.1, batch_first: bool = False,
                 value_proj_ratio: float = 1.0):
        super().__init__()
        if embed_dims % num_heads != 0:
            raise ValueError(f"embed_dims must be divisible by num_heads, but got {embed_dims} and {num_heads}")
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.num_levels = num_levels
        self.num_points = num_points
        self.batch_first = batch_first
        self.dropout = nn.Dropout(dropout)
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        vdim = int(embed_dims * value_proj_ratio)
        self.value_proj = nn.Linear(embed_dims, vdim)
        self.output_proj = nn.Linear(vdim, embed_dims)
        self._reset_parameters()

    def _reset_parameters(self) -> None:
        # msdeformattn.py:74-83 (the in-tree variant zero-initialises the offset/weight projections)
        for lin in (self.sampling_offsets, self.attention_weights):
            nn.init.zeros_(lin.weight)
            nn.init.zeros_(lin.bias)
        for lin in (self.value_proj, self.output_proj):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.zeros_(lin.bias)

    def forward(self, query: torch.Tensor, key: Optional[torch.Tensor] = None,
                value: Optional[torch.Tensor] = None, identity: Optional[torch.Tensor] = None,
                query_pos: Optional[torch.Tensor] = None, key_padding_mask: Optional[torch.Tensor] = None,
                reference_points: Optional[torch.Tensor] = None, spatial_shapes: Optional[torch.Tensor] = None,
                level_start_index: Optional[torch.Tensor] = None, **kwargs) -> torch.Tensor:
        if self.training and self.dropout.p > 0:
            raise RuntimeError("lbm_amd implements the inference step only (call .eval())")
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query = query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
            identity = identity.permute(1, 0, 2)
        if key_padding_mask is not None:
            raise NotImplementedError("key_padding_mask is never used on the LBM path (transformer.py:114,130,142,320)")
        if reference_points is None or reference_points.shape[-1] != 2:
            raise ValueError("reference_points must have last dim 2 on the LBM path")
        B, Q, C = query.shape
        shapes = [(int(a), int(b)) for a, b in spatial_shapes.tolist()]
        if sum(a * b for a, b in shapes) != value.shape[1]:
            raise ValueError("spatial_shapes do not add up to the number of value tokens")
        out = hip.msda_forward(self, query.contiguous().float(), value.contiguous().float(),
                               reference_points.contiguous().float(), shapes,
                               identity.contiguous().float())
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        return out
