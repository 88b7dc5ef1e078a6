"""Reference import path lbm.models.lbm.object_tracker (object_tracker.py:12) -> the host-side association layer."""
from lbm_amd.object_tracker import LBMObjTracker  # noqa: F401
