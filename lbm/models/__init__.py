"""Reference import path ``from lbm.models import LBM`` (tools/eval_pttrack.py:8 — unresolvable in the reference
itself, which ships no __init__.py)."""
from lbm_amd.lbm import LBM  # noqa: F401
from lbm_amd.lbm_online import LBM_online, LBM_export  # noqa: F401
