"""lbm_amd — MI355X-native implementation of the LBM online point-tracking step.

The public surface mirrors the reference (``LBM``, ``LBM_online``, ``LBMTransformer``, ``Backbone``,
``MultiScaleDeformableAttention``); the ``lbm`` package next to this one re-exports it under the reference's
import paths so ``tools/*.py`` run unchanged.  All arithmetic is in ``liblbm_hip.so`` (see ``include/lbm_hip.h``).
"""
from .config import LBMConfig  # noqa: F401

__all__ = ["LBMConfig", "LBM", "LBM_online", "LBMTransformer", "Backbone", "MultiScaleDeformableAttention"]


def __getattr__(name):
    if name == "LBM":
        from .lbm import LBM
        return LBM
    if name == "LBM_online":
        from .lbm_online import LBM_online
        return LBM_online
    if name == "LBMTransformer":
        from .transformer import LBMTransformer
        return LBMTransformer
    if name == "Backbone":
        from .backbone import Backbone
        return Backbone
    if name == "MultiScaleDeformableAttention":
        from .msdeformattn import MultiScaleDeformableAttention
        return MultiScaleDeformableAttention
    raise AttributeError(name)
