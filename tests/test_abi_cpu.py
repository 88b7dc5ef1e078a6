"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/lbm_hip.h declares,
and the ctypes structures agree with the header.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

from lbm_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    with open(os.path.join(ROOT, "include", "lbm_hip.h")) as f:
        return f.read()


def test_library_exports_every_declared_symbol():
    lib = hip.load()
    text = re.sub(r"/\*.*?\*/", "", _header(), flags=re.S)
    declared = set(re.findall(r"\b(lbm_hip_\w+)\s*\(", text))
    declared -= {n for n in declared if n.endswith("_desc")}
    assert declared, "no declarations found"
    assert declared == set(hip.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.lbm_hip_version() == hip.header_constants()["LBM_HIP_ABI_VERSION"]


def test_weight_table_constants_are_consistent():
    k = hip.header_constants()
    assert k["LBM_W_LAYER_STREAM"] == k["LBM_W_ATTN_COUNT"]
    assert k["LBM_W_LAYER_LATTICE"] == 2 * k["LBM_W_ATTN_COUNT"]
    assert k["LBM_W_LAYER_UPDATE"] == k["LBM_W_LAYER_LATTICE"] + k["LBM_W_MSDA_COUNT"]
    assert k["LBM_W_LAYER_RHO_W"] == k["LBM_W_LAYER_UPDATE"] + k["LBM_W_ATTN_COUNT"]
    assert k["LBM_W_LAYER_OLG_W"] == k["LBM_W_LAYER_RHO_B"] + 1
    assert k["LBM_W_LAYER_COUNT"] == k["LBM_W_LAYER_OLG_B"] + 1
    assert k["LBM_W_HEAD_CMEM"] == k["LBM_W_MSDA_COUNT"] and k["LBM_W_HEAD_VIS"] == 2 * k["LBM_W_MSDA_COUNT"]
    assert k["LBM_W_HEAD_OFFH_W0"] == 3 * k["LBM_W_MSDA_COUNT"]
    assert k["LBM_W_HEAD_OL3_W"] == k["LBM_W_HEAD_VPROJ_B"] + 1
    assert k["LBM_W_HEAD_COUNT"] == k["LBM_W_HEAD_OL3_B"] + 1


def test_host_side_validation_needs_no_gpu():
    """Argument validation happens before any HIP call, so bad descriptors are rejected on a GPU-less host too."""
    lib = hip.load()
    d = hip.StepDesc()
    d.B, d.N, d.M, d.n_layers = 1, 8, 0, 3
    assert lib.lbm_hip_step_workspace_bytes(C.byref(d)) == 0
    assert b"bad B/N/M" in lib.lbm_hip_last_error()
    d.M = 12
    d.topk[0], d.topk[1], d.topk[2] = 9, 4, 1
    d.net_h, d.net_w, d.stride = 384, 512, 4
    n = lib.lbm_hip_step_workspace_bytes(C.byref(d))
    assert n > 6 * 16320 * 256 * 4      # at least the six projected value maps
    assert lib.lbm_hip_step_workspace_offset(C.byref(d), b"value") == 0
    assert lib.lbm_hip_step_workspace_offset(C.byref(d), b"nope") == -1
    d.topk[2] = 2
    assert lib.lbm_hip_step_workspace_bytes(C.byref(d)) == 0
    assert b"last layer" in lib.lbm_hip_last_error()


def test_product_path_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hip.HipError):
        hip.l2norm(torch.zeros(4, 256))
