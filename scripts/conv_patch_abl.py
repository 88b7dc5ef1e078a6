"""conv_patch at 64 frames (layer-2 and layer-3 shapes): time against the tiled kernel, and its timing-only ablation (tuning build,
debug key 26: 1 no patch DMA after the first, 2 no weight DMA after the prologue, 4 no MFMAs, 8 no epilogue).
usage: python scripts/conv_patch_abl.py"""
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
    res = torch.randn(B, H, W, Cout, device="cuda")
    for rep in range(2):
        lib.lbm_hip_debug_set(4, 21)
        t_tiled = timeit(lambda: hip.conv2d_nhwc_ss(x, w, b, k, st, k // 2, True))
        lib.lbm_hip_debug_set(4, 22)
        row = []
        for dbg in (0, 1, 2, 3, 4, 8, 3 + 8, 4 + 8, 15, 16, 32, 48, 3 + 8 + 16):
            assert lib.lbm_hip_debug_set(26, dbg) == 0
            row.append(timeit(lambda: hip.conv2d_nhwc_ss(x, w, b, k, st, k // 2, True)))
        lib.lbm_hip_debug_set(26, 0)
        lib.lbm_hip_debug_set(4, 0)
        print((B, H, W, Cin, Cout), "tiled %.1f | patch %.1f | no patch DMA %.1f | no weight DMA %.1f | no DMA %.1f | no MFMA %.1f | no epilogue %.1f | "
              "no DMA, no epilogue %.1f | DMA only %.1f | skeleton %.1f | no barrier %.1f | no DMA waits %.1f | neither %.1f | no DMA, epilogue, barrier %.1f us" % tuple([t_tiled] + row), flush=True)
