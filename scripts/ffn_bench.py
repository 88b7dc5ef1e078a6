"""Fused feed-forward (lbm_hip_ffn_slabs) against two linear_ss products: correctness vs fp64 torch and timing."""
import sys
import time

import torch

sys.path.insert(0, ".")
from lbm_amd import hip  # noqa: E402


ACT_GELU = hip.header_constants().get("LBM_ACT_GELU", 1)


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def main():
    torch.manual_seed(0)
    dev = "cuda"
    W1 = torch.randn(1024, 256, device=dev) / 16
    b1 = torch.randn(1024, device=dev) * 0.1
    W2 = torch.randn(256, 1024, device=dev) / 32
    s1, s2 = hip.SplitWeight(W1), hip.SplitWeight(W2)
    f1, f2 = hip.fragment_major(s1), hip.fragment_major(s2)
    for M in [int(a) for a in (sys.argv[1:] or ["1024", "1000", "256", "4096", "9216"])]:
        x = torch.randn(M, 256, device=dev)
        xs = hip.SplitTensor((M, 256), dev)
        hip._check(hip.load().lbm_hip_split_planes(x.data_ptr(), xs.hi, xs.planes[1].data_ptr(), x.numel(), hip._stream()), "split")
        want = (torch.nn.functional.gelu(x.double() @ W1.double().t() + b1.double()) @ W2.double().t())
        for planes in (2, 1):
            got = hip.ffn_slabs(xs, f1, b1, f2, planes).sum(0)
            err = (got.double() - want).abs().max().item()
            t_f = timeit(lambda: hip.ffn_slabs(xs, f1, b1, f2, planes))

            def two():
                _, hs = hip.linear_ss(xs, s1, b1, act=ACT_GELU, want_f32=False, want_split=True, planes=planes)
                t = hip.SplitTensor.__new__(hip.SplitTensor)
                t.planes, t.shape = hs, tuple(hs.shape[1:])
                hip.linear_ss(t, s2, None, planes=planes)
            try:
                t_2 = timeit(two)
            except Exception as e:      # signature differences: report and go on
                t_2 = float("nan")
                print("two-product path not timed:", e)
            print(f"M={M} planes={planes}: max abs err {err:.2e} (|want| max {want.abs().max().item():.2f}); fused {t_f:.2f} us, two products {t_2:.2f} us", flush=True)


main()
