"""Import-path twin of lbm/utils/train_utils.py.  With the reference checkout (and the dataset dependencies its file imports
at the top, train_utils.py:11-14) on the path, the reference's own file is executed here, so ``get_dataloaders``,
``init_distributed_mode``, ``get_scheduler``, ``restart_from_checkpoint``, ``save_on_master`` ... are the reference's
(tools/eval_pttrack.py:9, main.py:15-17).  Otherwise the twin provides ``fix_random_seeds`` (train_utils.py:243-250), which is
all the demo / benchmark tools take from this module (tools/demo_grid.py:10, tools/benchmark_torch.py:11)."""
from lbm_amd._refshim import delegate as _delegate, unavailable as _unavailable

if not _delegate(globals()):
    import os
    import random

    import numpy as np
    import torch

    def fix_random_seeds(seed):
        os.environ["PYTHONHASHSEED"] = str(seed)
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)

    for _n in ("get_dataloaders", "get_scheduler", "restart_from_checkpoint", "restart_from_checkpoint_not_dist",
               "is_dist_avail_and_initialized", "get_world_size", "get_rank", "is_main_process", "save_on_master",
               "init_distributed_mode", "setup_for_distributed"):
        globals()[_n] = _unavailable(_n, __lbm_amd_delegate_error__)  # noqa: F821
