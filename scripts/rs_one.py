"""One big split product a few times (for rocprofv3 --pmc passes).  usage: rs_one.py M N flag [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lbm_amd import hip

M, N, flag = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
lib = hip.load()
xs = hip.SplitTensor((M, 256), "cuda")
xs.planes.copy_(torch.randn(2, M, 256, device="cuda").half())
ws = hip.SplitWeight(torch.from_numpy((np.random.default_rng(0).normal(size=(N, 256)) * 0.1).astype(np.float32)).cuda())
b = torch.zeros(N, device="cuda")
lib.lbm_hip_debug_set(2, flag)
for _ in range(reps):
    hip.linear_ss(xs, ws, b, want_f32=True, want_split=False)
torch.cuda.synchronize()
