"""Ablation of the LDS-DMA split GEMM: which part of the kernel the time goes to (GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lbm_amd import hip
from gemm_bench import timeit

lib = hip.load()
r = np.random.default_rng(0)
for (M, N, K) in [(1024, 1024, 256), (1024, 256, 256), (1024, 1024, 64), (16384, 1024, 256)]:
    x = hip.SplitWeight(torch.from_numpy(r.normal(size=(M, K)).astype(np.float32)).cuda())
    w = hip.SplitWeight(torch.from_numpy(r.normal(size=(N, K)).astype(np.float32)).cuda())
    for tile in (0, 2):
        lib.lbm_hip_debug_set(1, tile)
        row = []
        for dbg in (0, 1, 2, 4, 3, 5, 6, 7, 8):
            lib.lbm_hip_debug_set(2, dbg)
            row.append(timeit(lambda: hip.linear_ss(x, w)))
        lib.lbm_hip_debug_set(2, 0)
        print(M, N, K, "tile", tile, " full %.1f | noDMA %.1f | noMFMA %.1f | noStore %.1f | noDMA+noMFMA %.1f | noDMA+noStore %.1f | noMFMA+noStore %.1f | none %.1f | return %.1f" % tuple(row))
lib.lbm_hip_debug_set(1, 0)
