"""Reference import path lbm.models.lbm_online (lbm_online.py:10, :203) -> the MI355X implementation."""
from lbm_amd.lbm_online import LBM_online, LBM_export  # noqa: F401
