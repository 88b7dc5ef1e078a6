"""Reference import path lbm.models.lbm.lbm (lbm.py:11) -> the MI355X implementation."""
from lbm_amd.lbm import LBM  # noqa: F401
