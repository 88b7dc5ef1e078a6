"""Reference import path lbm.models.lbm.transformer (transformer.py:10,154,343) -> the MI355X implementation."""
from lbm_amd.transformer import LBMTransformer, LBMTransformerLayer, MLP  # noqa: F401
