"""Reference import path lbm.models.lbm.backbone (backbone.py:9) -> the MI355X implementation."""
from lbm_amd.backbone import Backbone  # noqa: F401
