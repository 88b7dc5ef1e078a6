"""Reference import path lbm.models.lbm.msdeformattn (msdeformattn.py:9,188) -> the MI355X operator."""
from lbm_amd.msdeformattn import MultiScaleDeformableAttention, multi_scale_deformable_attn_pytorch  # noqa: F401
