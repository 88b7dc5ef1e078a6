"""fix_random_seeds as the tools use it (lbm/utils/train_utils.py:243-250).  The dataset / DDP helpers of the
reference file belong to training (out of scope)."""
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


def get_dataloaders(args):
    raise NotImplementedError("datasets (TAP-Vid, Kubric, ...) are host-side IO outside the hot path; "
                              "use the reference's lbm/utils/train_utils.py and lbm/datasets for them")
