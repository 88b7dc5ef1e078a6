"""Import-path twin of lbm/utils/eval_utils.py.  With the reference checkout on the path its own file is executed here
(lbm_amd/_refshim.py), so ``Evaluator``, ``compute_tapvid_metrics``, ``evaluate_results``, ``bboxes_resize``, ``get_paths``,
``print_args`` ... are the reference's (tools/eval_pttrack.py:8-10, tools/eval_objtrack.py:14-19, main.py:19-21).  Alone, the
twin provides ``load_config`` with the reference's merge rule (eval_utils.py:21-32): YAML first, then every CLI argument that
is not None overrides it; the result is a flat argparse.Namespace consumed by LBM.__init__."""
from lbm_amd._refshim import delegate as _delegate, unavailable as _unavailable

if not _delegate(globals()):
    import argparse
    import os

    import yaml

    def load_config(args):
        path = getattr(args, "config_path", None)
        if not path or not os.path.exists(path):
            print(f"Warning: Config file not found at {path}. Using default CLI args.")
            return {}                       # the reference returns an empty dict in this case (eval_utils.py:24-25)
        with open(path, "r") as f:
            merged = yaml.safe_load(f) or {}
        merged.update({k: v for k, v in vars(args).items() if v is not None})
        return argparse.Namespace(**merged)

    for _n in ("Evaluator", "compute_tapvid_metrics", "evaluate_results", "bboxes_resize", "get_paths", "print_args",
               "set_random_seed", "compute_teta_on_ovsetup", "default_configs"):
        globals()[_n] = _unavailable(_n, __lbm_amd_delegate_error__)  # noqa: F821
