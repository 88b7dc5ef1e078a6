import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lbm_amd import hip
from conv_bench import timeit, lib, r, scratch
for (B, H, W, Cin, Cout, k, st) in [(1, 96, 128, 64, 64, 3, 1), (1, 24, 32, 256, 256, 3, 1)]:
    x = hip.SplitTensor((B, H, W, Cin), "cuda")
    xf = torch.from_numpy(r.normal(size=(B, H, W, Cin)).astype(np.float32)).cuda()
    lib.lbm_hip_split_planes(xf.data_ptr(), x.hi, x.planes[1].data_ptr(), xf.numel(), None)
    w = hip.SplitWeight(torch.from_numpy((r.normal(size=(Cout, k * k * Cin)) * 0.05).astype(np.float32)).cuda())
    b = torch.zeros(Cout, device="cuda")
    for mode in (0, 2, 6):
        lib.lbm_hip_debug_set(4, mode)
        row = []
        for dbg in (0, 1, 2, 4, 3, 6, 7, 8):
            lib.lbm_hip_debug_set(2, dbg)
            row.append(timeit(lambda: hip.conv2d_nhwc_ss(x, w, b, k, st, k // 2, True, scratch=scratch))[0])
        lib.lbm_hip_debug_set(2, 0)
        print((H, W, Cin, Cout), "mode", mode, "full %.1f | noDMA %.1f | noMFMA %.1f | noStore %.1f | noDMA+noMFMA %.1f | noMFMA+noStore %.1f | none %.1f | return %.1f" % tuple(row), flush=True)
lib.lbm_hip_debug_set(4, 0)
