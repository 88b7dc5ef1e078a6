"""Model constants for the LBM tracking step.

Mirrors the fields ``LBM.__init__`` reads from its ``args`` namespace
(reference: lbm/models/lbm/lbm.py:15-38, lbm/configs/demo.yaml:2-22).
"""
from __future__ import annotations

import argparse
from dataclasses import dataclass, asdict
from typing import List, Tuple


@dataclass
class LBMConfig:
    input_size: Tuple[int, int] = (384, 512)
    stride: int = 4
    num_layers: int = 3
    embed_dim: int = 256
    num_heads: int = 4
    num_levels: int = 4
    num_neighbors: int = 9
    top_k_regions: int = 9
    memory_size: int = 12
    dropout: float = 0.1
    N: int = 256
    T: int = 24
    lambda_cls: float = 3.0
    lambda_vis: float = 1.0
    lambda_reg: float = 1.0
    lambda_unc: float = 1.0
    lambda_ref: float = 1.0
    loss_after_query: bool = False

    # ---- derived geometry (reference: transformer.py:17-22) ----
    @property
    def h(self) -> int:
        return self.input_size[0] // self.stride

    @property
    def w(self) -> int:
        return self.input_size[1] // self.stride

    @property
    def num_cells(self) -> int:
        return self.h * self.w

    @property
    def level_shapes(self) -> List[Tuple[int, int]]:
        return [(self.h >> l, self.w >> l) for l in range(self.num_levels)]

    @property
    def level_starts(self) -> List[int]:
        out, acc = [], 0
        for a, b in self.level_shapes:
            out.append(acc)
            acc += a * b
        return out

    @property
    def num_tokens(self) -> int:
        return sum(a * b for a, b in self.level_shapes)

    @property
    def topk_per_layer(self) -> List[int]:
        return [self.top_k_regions] + [4] * (self.num_layers - 2) + [1]

    def to_namespace(self) -> argparse.Namespace:
        d = asdict(self)
        d["input_size"] = list(self.input_size)
        return argparse.Namespace(**d)

    @classmethod
    def from_args(cls, args) -> "LBMConfig":
        """Accepts an argparse.Namespace / any attribute bag as produced by ``load_config``."""
        kw = {}
        for f in cls.__dataclass_fields__:
            if hasattr(args, f):
                kw[f] = getattr(args, f)
        if "input_size" in kw:
            kw["input_size"] = tuple(int(v) for v in kw["input_size"])
        return cls(**kw)

    def validate_for_hip(self) -> None:
        """The HIP kernels are specialised for the published model shape; say so loudly otherwise."""
        errs = []
        if self.embed_dim != 256:
            errs.append(f"embed_dim must be 256 (got {self.embed_dim})")
        if self.num_heads != 4:
            errs.append(f"num_heads must be 4 (got {self.num_heads})")
        if self.num_levels != 4:
            errs.append(f"num_levels must be 4 (got {self.num_levels})")
        if self.stride != 4:
            errs.append(f"stride must be 4 (got {self.stride})")
        if self.h % 8 or self.w % 8:
            errs.append(f"feature grid {self.h}x{self.w} must be a multiple of 8")
        if not (1 <= self.num_neighbors <= 16):
            errs.append("num_neighbors must be in [1,16]")
        if not (1 <= self.top_k_regions <= 16):
            errs.append("top_k_regions must be in [1,16]")
        if not (1 <= self.memory_size <= 128):
            errs.append("memory_size must be in [1,128]")
        if errs:
            raise ValueError("configuration not supported by the gfx950 kernels: " + "; ".join(errs))
