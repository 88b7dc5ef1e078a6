"""Delegation of the import-path twin ``lbm/utils/*.py`` to the reference checkout.

The reference ships no ``__init__.py`` (PEP-420 namespace packages), so with this repository first on ``PYTHONPATH`` and the
reference checkout second, ``lbm.utils`` spans both trees and a module name resolves to THIS repository's file when it has
one.  A twin that only provides what the tracking path needs would then hide everything else of the reference's module
(metric code, dataset helpers, ...) from the reference's own tools.  ``delegate`` therefore executes the reference's file --
the next ``<dir>/<module>.py`` along the namespace path that is not ours -- inside the twin's namespace, so every name of the
reference module exists unchanged; the twin then overrides only what must differ.  Without a reference checkout on the path
(this repository alone), or when the reference module cannot be imported because one of ITS third-party dependencies is
missing, the twin falls back to its own minimal definitions and says why in ``__lbm_amd_delegate_error__``.
"""
from __future__ import annotations

import importlib
import os
from typing import Optional

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_file(module_name: str) -> Optional[str]:
    """Path of the reference checkout's source file for ``module_name`` (e.g. 'lbm.utils.eval_utils'), or None."""
    pkg_name, _, leaf = module_name.rpartition(".")
    try:
        pkg = importlib.import_module(pkg_name)
    except ImportError:
        return None
    for d in list(getattr(pkg, "__path__", [])):
        d = os.path.abspath(d)
        if d == _REPO or d.startswith(_REPO + os.sep):
            continue
        cand = os.path.join(d, leaf + ".py")
        if os.path.isfile(cand):
            return cand
    return None


def delegate(module_globals: dict) -> bool:
    """Run the reference's file for the calling twin module in ``module_globals``.  True when it ran to the end."""
    name = module_globals["__name__"]
    module_globals["__lbm_amd_delegate_error__"] = None
    module_globals["__lbm_amd_reference_file__"] = None
    path = reference_file(name)
    if path is None:
        module_globals["__lbm_amd_delegate_error__"] = "no reference checkout on the import path"
        return False
    keep = dict(module_globals)
    try:
        with open(path, "r") as f:
            code = compile(f.read(), path, "exec")
        exec(code, module_globals)
    except ImportError as e:            # a third-party dependency of the REFERENCE module is missing: fall back
        for k in list(module_globals):
            if k not in keep:
                del module_globals[k]
        module_globals.update(keep)
        module_globals["__lbm_amd_delegate_error__"] = f"{path}: {e}"
        return False
    module_globals["__lbm_amd_reference_file__"] = path
    return True


def unavailable(name: str, why: str):
    """A placeholder for a reference function that could not be provided; calling it raises with the reason."""
    def fn(*a, **k):
        raise ImportError(f"{name} comes from the reference checkout and is not available here: {why}")
    fn.__name__ = name
    fn.__lbm_amd_stub__ = True
    return fn
