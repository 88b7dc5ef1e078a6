"""Layer-2 / layer-3 convolutions of 64 frames on the tiled kernel with the LDS-DMA switched off after the prologue (tuning build,
debug key 2 bit 1; results wrong): what the DMA costs.  usage: python scripts/conv_abl64.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LBM_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lbm_amd", "liblbm_hip_tuning.so"))
import numpy as np, torch
from lbm_amd import hip
lib = hip.load()
r = np.random.default_rng(0)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, H, W, Cin, Cout, k, st) in [(64, 48, 64, 128, 128, 3, 1), (64, 24, 32, 256, 256, 3, 1)]:
    x = hip.SplitTensor((B, H, W, Cin), "cuda")
    xf = torch.from_numpy(r.normal(size=(B, H, W, Cin)).astype(np.float32)).cuda()
    lib.lbm_hip_split_planes(xf.data_ptr(), x.hi, x.planes[1].data_ptr(), xf.numel(), None)
    w = hip.SplitWeight(torch.from_numpy((r.normal(size=(Cout, k * k * Cin)) * 0.05).astype(np.float32)).cuda())
    b = torch.zeros(Cout, device="cuda")
    for rep in range(2):
        row = []
        for dbg in (0, 1, 4, 5, 8):
            assert lib.lbm_hip_debug_set(2, dbg) == 0
            row.append(timeit(lambda: hip.conv2d_nhwc_ss(x, w, b, k, st, k // 2, True)))
        lib.lbm_hip_debug_set(2, 0)
        print((B, H, W, Cin, Cout), "everything %.1f | no DMA %.1f | no stores %.1f | no DMA, no stores %.1f | skeleton %.1f us" % tuple(row), flush=True)
