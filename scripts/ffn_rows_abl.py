"""Ablation of ffn_rows_kernel (debug key 16 bits) at one row count.  usage: ffn_rows_abl.py [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import hip
from gemm_bench import timeit

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
lib = hip.load()
g = torch.Generator(device="cpu").manual_seed(0)
s1 = hip.SplitWeight((torch.randn(1024, 256, generator=g) / 16).cuda())
s2 = hip.SplitWeight((torch.randn(256, 1024, generator=g) / 32).cuda())
b1 = (torch.randn(1024, generator=g) * 0.1).cuda()
f1, f2 = hip.fragment_major(s1), hip.fragment_major(s2)
xs = hip.SplitTensor((M, 256), "cuda")
xs.planes.copy_(torch.randn(2, M, 256).half())
lib.lbm_hip_debug_set(15, 2)
for name, bits in [("full", 0), ("no gelu", 1), ("no stores", 2), ("no H writes", 4), ("no gelu/H/stores", 7), ("no producer mfma", 8), ("no consumer mfma", 16),
                   ("no mfma at all", 24), ("only mfma", 7), ("nothing", 31)]:
    lib.lbm_hip_debug_set(16, bits)
    t = timeit(lambda: hip.ffn_slabs(xs, f1, b1, f2), n=10)
    print(f"M {M}: {name:18s} {t:7.1f} us", flush=True)
lib.lbm_hip_debug_set(16, 0)
lib.lbm_hip_debug_set(15, 0)
