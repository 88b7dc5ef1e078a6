"""Large split-precision products of the throughput regime (C3 / C5 shapes): tilings and knobs side by side (GPU only).
usage: gemm_big.py "tiles" "dbg flags"   e.g. gemm_big.py 3,1,12 0,32"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lbm_amd import hip
from gemm_bench import timeit

SHAPES = [(16384, 1024, 256, 1), (16384, 256, 1024, 0), (16384, 256, 256, 0), (16384, 1456, 256, 0), (10000, 1024, 256, 1), (10000, 256, 1024, 0),
          (90000, 256, 256, 0), (147456, 1024, 256, 1), (1044480, 1536, 256, 0)]
tiles = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3").split(",")]
flags = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
lib = hip.load()
r = np.random.default_rng(0)
for M, N, K, act in SHAPES:
    if M * N * 4 > 8e9:
        continue
    xs = hip.SplitTensor((M, K), "cuda")
    xs.planes.copy_(torch.randn(2, M, K, device="cuda").half())
    ws = hip.SplitWeight(torch.from_numpy((r.normal(size=(N, K)) * 0.1).astype(np.float32)).cuda())
    b = torch.zeros(N, device="cuda")
    row = []
    for t in tiles:
        for f in flags:
            lib.lbm_hip_debug_set(1, t)
            lib.lbm_hip_debug_set(2, f)
            us = timeit(lambda: hip.linear_ss(xs, ws, b, act=act, want_f32=False, want_split=True), n=10)
            row.append(f"t{t}/f{f}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF")
    lib.lbm_hip_debug_set(1, 0)
    lib.lbm_hip_debug_set(2, 0)
    print(f"{M:8d} {N:5d} {K:5d} act{act} | " + " | ".join(row), flush=True)
