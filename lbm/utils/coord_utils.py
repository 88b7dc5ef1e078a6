"""Import-path twin of lbm/utils/coord_utils.py.  With the reference checkout on the path its own file is executed here
(``coords_to_indices``, ``get_queries``, ``get_points_on_a_grid`` for main.py:20 and the datasets).  Alone, the twin provides
``indices_to_coords`` (coord_utils.py:4-29) without the three host-synchronising asserts; on the hot path the same
arithmetic is fused into the top-k kernel (lbm_hip_topk)."""
from lbm_amd._refshim import delegate as _delegate

if not _delegate(globals()):
    import torch

    def indices_to_coords(indices, size, ps):
        """flat cell index (B,T,N) -> cell-centre (x, y) pixels (B,T,N,2) on a grid of size[1] // ps columns."""
        cols_per_row = size[1] // ps
        x = (indices % cols_per_row) * ps + 0.5 * ps
        y = torch.div(indices, cols_per_row, rounding_mode="floor") * ps + 0.5 * ps
        return torch.stack((x, y), dim=-1)
