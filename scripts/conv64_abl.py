"""Ablation of conv64_rows_kernel (debug key 18 bits) at 64 x 96 x 128 pixels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import hip
from gemm_bench import timeit

lib = hip.load()
B, H, W = 64, 96, 128
g = torch.Generator(device="cpu").manual_seed(0)
xs = hip.SplitTensor((B, H, W, 64), "cuda")
xs.planes.copy_(torch.randn(2, B, H, W, 64, generator=g).half())
w = hip.SplitWeight((torch.randn(64, 576, generator=g) * 0.05).cuda())
b = torch.zeros(64, device="cuda")
for f32 in (False, True):
    for name, bits in [("full", 0), ("no dma", 1), ("no epilogue", 2), ("no producer mfma", 4), ("no consumer mfma", 8), ("no mfma", 12), ("mfma only", 3), ("nothing", 15)]:
        lib.lbm_hip_debug_set(18, bits)
        lib.lbm_hip_debug_set(4, 16)
        t = timeit(lambda: hip.conv2d_nhwc_ss(xs, w, b, 3, 1, 1, True, None, f32, True), n=10)
        print(f"f32={f32} {name:18s} {t:7.1f} us", flush=True)
lib.lbm_hip_debug_set(18, 0)
lib.lbm_hip_debug_set(4, 0)
