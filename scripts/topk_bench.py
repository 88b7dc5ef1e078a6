"""top-k kernel timing on noise-like and smooth 1024 x 12288 correlation maps (GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lbm_amd import hip

lib = hip.load()
r = np.random.default_rng(0)
R, P = 1024, 12288


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    hip.profile_begin()
    for _ in range(n):
        fn()
    prof = hip.profile_end()
    return sum(p["total_ms"] for p in prof) / n * 1e3


noise = torch.from_numpy(r.normal(size=(R, P)).astype(np.float32)).cuda()
yy, xx = np.meshgrid(np.arange(96), np.arange(128), indexing="ij")
cy, cx = r.uniform(0, 96, size=(R, 1, 1)), r.uniform(0, 128, size=(R, 1, 1))
smooth = torch.from_numpy(np.exp(-((yy[None] - cy) ** 2 + (xx[None] - cx) ** 2) / (2 * 20.0 ** 2)).reshape(R, P).astype(np.float32)).cuda()
mixed = (smooth + 0.05 * noise).contiguous()
for name, c in (("noise", noise), ("smooth bump (sigma 20 cells)", smooth), ("bump + noise", mixed)):
    for k in (1, 4, 9):
        row = []
        for mode in (0, 1):
            lib.lbm_hip_debug_set(5, mode)
            row.append(timeit(lambda: hip.topk(c, k, grid_w=128)))
        lib.lbm_hip_debug_set(5, 0)
        print(f"{name:30s} k={k}: register-resident {row[0]:6.1f} us   per-lane lists {row[1]:6.1f} us", flush=True)
