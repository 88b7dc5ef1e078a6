"""Ablation of stem_pool_kernel (debug key 20 bits) at 64 frames of 384 x 512."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import hip

lib = hip.load()
g = torch.Generator(device="cpu").manual_seed(0)
w = torch.cat([torch.randn(64, 196, generator=g) * 0.05, torch.zeros(64, 28)], 1).contiguous()
ws = hip.SplitWeight(w.cuda())
b = torch.randn(64, generator=g).cuda()
B = 64
xs = hip.SplitTensor((B, 384, 512, 4), "cuda")
xs.planes.copy_(torch.randn(2, B, 384, 512, 4, generator=g).half())
for name, bits in [("full", 0), ("full again", 0), ("no dma", 1), ("no pooling/stores", 2), ("no mfma", 4), ("no hand-over writes", 8), ("mfma only", 1 | 2 | 8), ("dma only", 2 | 4 | 8),
                   ("pool only", 1 | 4 | 8), ("nothing", 15)]:
    lib.lbm_hip_debug_set(20, bits)
    for _ in range(10):
        hip.stem_pool(xs, ws, b)
    torch.cuda.synchronize()
    hip.profile_begin()
    for _ in range(8):
        hip.stem_pool(xs, ws, b)
    p = hip.profile_end()
    print(f"{name:22s} {1e3 * p[0]['total_ms'] / p[0]['launches']:7.1f} us", flush=True)
lib.lbm_hip_debug_set(20, 0)
