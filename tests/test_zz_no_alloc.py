"""lbm_hip_update_points_static (the entry point that allocates nothing: scan scratch in a static device array) against
lbm_hip_update_points, through the C ABI, bit for bit, on two calls in a row -- the second with a smaller input -- and the
size limit.  Collected last (test_zz_*): the round's end runs pytest -x."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(N_old, N_new, M, seed):
    g = torch.Generator().manual_seed(seed)
    dev = "cuda"
    old = (torch.randn(1, N_old, 256, generator=g).to(dev), torch.randn(1, N_old, M, 256, generator=g).to(dev),
           torch.randn(1, N_old, M, 256, generator=g).to(dev),
           (torch.rand(1, N_old, M, generator=g) < 0.5).to(torch.uint8).to(dev),
           (torch.rand(1, N_old, M, generator=g) < 0.3).to(torch.uint8).to(dev))
    remove = (torch.rand(N_old, generator=g) < 0.4).to(torch.uint8).to(dev)
    update = (torch.rand(N_new, generator=g) < 0.5).to(torch.uint8).to(dev)
    new_q = torch.randn(1, N_new, 256, generator=g).to(dev)
    return old, remove, update, new_q


def _outputs(N_new, M):
    dev = "cuda"
    return (torch.full((1, N_new, 256), float("nan"), device=dev), torch.full((1, N_new, M, 256), float("nan"), device=dev),
            torch.full((1, N_new, M, 256), float("nan"), device=dev), torch.full((1, N_new, M), 7, device=dev, dtype=torch.uint8),
            torch.full((1, N_new, M), 7, device=dev, dtype=torch.uint8))


def _call(lib, static, old, remove, update, new_q, out, M, stream):
    N_old, N_new = old[0].shape[1], update.shape[0]
    args = [remove.data_ptr(), N_old, update.data_ptr(), N_new, M] + [t.data_ptr() for t in old] + [new_q.data_ptr()] + \
           [t.data_ptr() for t in out]
    if static:
        return lib.lbm_hip_update_points_static(*args, stream)
    scratch = torch.empty(N_old + 1, device="cuda", dtype=torch.int32)
    rc = lib.lbm_hip_update_points(*args, scratch.data_ptr(), stream)
    torch.cuda.synchronize()
    return rc


def test_static_scratch_entry_point_matches_the_allocating_one():
    from lbm_amd import hip
    lib = hip.load()
    stream = torch.cuda.current_stream().cuda_stream
    M = 5
    # two calls in a row on the same GPU (they share the static array), the second smaller than the first
    for N_old, N_new, seed in ((3000, 3300, 1), (700, 650, 2), (1, 4, 3)):
        old, remove, update, new_q = _case(N_old, N_new, M, seed)
        want, got = _outputs(N_new, M), _outputs(N_new, M)
        assert _call(lib, False, old, remove, update, new_q, want, M, stream) == 0, lib.lbm_hip_last_error()
        before = torch.cuda.memory_allocated()
        assert _call(lib, True, old, remove, update, new_q, got, M, stream) == 0, lib.lbm_hip_last_error()
        assert torch.cuda.memory_allocated() == before
        torch.cuda.synchronize()
        for w, g in zip(want, got):
            assert torch.equal(w.view(torch.uint8) if w.dtype == torch.uint8 else w.view(torch.int32),
                               g.view(torch.uint8) if g.dtype == torch.uint8 else g.view(torch.int32))
        assert not bool(torch.isnan(got[0]).any())


def test_static_scratch_entry_point_refuses_more_points_than_it_holds():
    from lbm_amd import hip
    lib = hip.load()
    limit = hip.header_constants()["LBM_UPDATE_STATIC_MAX_POINTS"]
    dummy = torch.zeros(16, device="cuda")
    p = dummy.data_ptr()
    rc = lib.lbm_hip_update_points_static(p, limit + 1, p, 4, 2, p, p, p, p, p, p, p, p, p, p, p, torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b"exceeds the static scratch" in lib.lbm_hip_last_error()
