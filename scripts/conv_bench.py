"""Encoder convolution shapes (one 384x512 frame) under the kernel variants of conv2d_nhwc_ss (GPU only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lbm_amd import hip

lib = hip.load()
r = np.random.default_rng(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    hip.profile_begin()
    for _ in range(n):
        fn()
    prof = hip.profile_end()
    return sum(p["total_ms"] for p in prof) / n * 1e3, "+".join(sorted(p["name"] for p in prof))


SHAPES = [  # B, H, W, Cin, Cout, k, stride
    (1, 96, 128, 64, 64, 3, 1), (1, 96, 128, 64, 128, 3, 2), (1, 48, 64, 128, 128, 3, 1), (1, 96, 128, 64, 128, 1, 2),
    (1, 48, 64, 128, 256, 3, 2), (1, 24, 32, 256, 256, 3, 1), (1, 48, 64, 128, 256, 1, 2),
]
B_ = int(sys.argv[1]) if len(sys.argv) > 1 else 1
MODES = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(0, 7))
scratch = torch.empty(32 << 20, device="cuda", dtype=torch.uint8) if B_ == 1 else None
for (B, H, W, Cin, Cout, k, st) in SHAPES:
    B = B_
    x = hip.SplitTensor((B, H, W, Cin), "cuda")
    xf = torch.from_numpy(r.normal(size=(B, H, W, Cin)).astype(np.float32)).cuda()
    lib.lbm_hip_split_planes(xf.data_ptr(), x.hi, x.planes[1].data_ptr(), xf.numel(), None)
    w = hip.SplitWeight(torch.from_numpy((r.normal(size=(Cout, k * k * Cin)) * 0.05).astype(np.float32)).cuda())
    b = torch.zeros(Cout, device="cuda")
    ref = None
    row = []
    for mode in MODES:
        lib.lbm_hip_debug_set(4, mode)
        y, _ = hip.conv2d_nhwc_ss(x, w, b, k, st, k // 2, True, scratch=scratch)
        if ref is None:
            ref = y.clone()
        err = float((y - ref).abs().max())
        t, names = timeit(lambda: hip.conv2d_nhwc_ss(x, w, b, k, st, k // 2, True, scratch=scratch))
        row.append("%5.1f%s" % (t, "" if err < 1e-3 else "!ERR%.2g" % err))
        if mode == MODES[0]:
            n0 = names
    lib.lbm_hip_debug_set(4, 0)
    print((B, H, W, Cin, Cout, k, st), " modes %s (0 default | 1 64x64s3 | 2 64x64s4 | 3 32x64s4 | 4 32x64s6 | 5 64x64s6 | 6 32x32s6 | 7 128x64 | 8 256x64/8w | 9 256x64/4w | 10 256x128 | 11 128x256 | 12 128x128/8w | 13 128x128/4w | 14 64x64s2):" % MODES, " | ".join(row), "  [", n0, "]", flush=True)
