"""One split-GEMM shape, a few launches (for rocprofv3 --pmc passes): gemm_one.py M N K [tile]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lbm_amd import hip
M, N, K = (int(v) for v in sys.argv[1:4])
tile = int(sys.argv[4]) if len(sys.argv) > 4 else 0
r = np.random.default_rng(0)
x = hip.SplitWeight(torch.from_numpy(r.normal(size=(M, K)).astype(np.float32)).cuda())
w = hip.SplitWeight(torch.from_numpy(r.normal(size=(N, K)).astype(np.float32)).cuda())
hip.load().lbm_hip_debug_set(1, tile)
for _ in range(5):
    y = hip.linear_ss(x, w)
torch.cuda.synchronize()
