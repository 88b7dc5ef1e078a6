"""Feed-forward 256 -> 1024 -> 256 for many rows: the two products, the few-rows fused kernel and the many-rows kernel (GPU only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import hip
from gemm_bench import timeit

lib = hip.load()
g = torch.Generator(device="cpu").manual_seed(0)
W1 = (torch.randn(1024, 256, generator=g) / 16).cuda()
b1 = (torch.randn(1024, generator=g) * 0.1).cuda()
W2 = (torch.randn(256, 1024, generator=g) / 32).cuda()
s1, s2 = hip.SplitWeight(W1), hip.SplitWeight(W2)
f1, f2 = hip.fragment_major(s1), hip.fragment_major(s2)
b2 = torch.zeros(256, device="cuda")
for M in (1024, 2048, 4096, 9216, 10000, 16384):
    xs = hip.SplitTensor((M, 256), "cuda")
    xs.planes.copy_(torch.randn(2, M, 256).half())

    def two():
        _, hs = hip.linear_ss(xs, s1, b1, act=1, want_f32=False, want_split=True)
        h = hip.SplitTensor((M, 1024), "cuda")
        h.planes = hs
        hip.linear_ss(h, s2, b2)
    t2 = timeit(two, n=10)
    lib.lbm_hip_debug_set(15, 1)
    tf = timeit(lambda: hip.ffn_slabs(xs, f1, b1, f2), n=10)
    lib.lbm_hip_debug_set(15, 2)
    tr = timeit(lambda: hip.ffn_slabs(xs, f1, b1, f2), n=10)
    lib.lbm_hip_debug_set(15, 0)
    print(f"M {M:6d}: two products {t2:7.1f} us | few-rows fused {tf:7.1f} us | many-rows {tr:7.1f} us ({4.0 * M * 256 * 1024 / tr / 1e6:6.1f} TF)", flush=True)
