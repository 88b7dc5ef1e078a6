"""Micro-benchmark + accuracy check of the two GEMM paths at the shapes of the tracking step (GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lbm_amd import hip

SHAPES = [(1024, 1024, 256), (1024, 256, 1024), (1024, 432, 256), (1024, 256, 256), (9216, 256, 256), (9216, 1024, 256),
          (9216, 256, 1024), (16320, 1536, 256), (1024, 12288, 256), (12288, 256, 192), (16384, 1024, 256), (16384, 256, 1024)]


if os.environ.get("SHAPES"):          # e.g. SHAPES=16384x256x1024,147456x256x1024
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(",")]
TILES = [int(v) for v in os.environ.get("TILES", "1,2,3,4,5,6,7,8,9").split(",")]


def timeit(fn, n=20):
    """GPU time per launch in us from the library's per-launch HIP events (no CPU launch gaps)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    hip.profile_begin()
    for _ in range(n):
        fn()
    prof = [p for p in hip.profile_end() if p["name"] != "split_planes"]
    return sum(p["total_ms"] for p in prof) / n * 1e3


def main():
    r = np.random.default_rng(0)
    print(f"{'M':>6} {'N':>6} {'K':>5} | fp32 us   TF  err      | split us   TF  err")
    for M, N, K in SHAPES:
        x = torch.from_numpy(r.normal(size=(M, K)).astype(np.float32)).cuda()
        w = torch.from_numpy((r.normal(size=(N, K)) * 0.1).astype(np.float32)).cuda()
        b = torch.from_numpy(r.normal(size=(N,)).astype(np.float32)).cuda()
        ws = hip.SplitWeight(w)
        want = (x.double() @ w.double().t() + b.double())
        y32 = torch.empty(M, N, device="cuda")
        y16 = torch.empty(M, N, device="cuda")
        t32 = timeit(lambda: hip.linear(x, w, b, out=y32))
        t16 = timeit(lambda: hip.linear(x, ws, b, out=y16))
        alt = []
        xs = hip.SplitWeight(x)
        for tile in TILES:
            hip.load().lbm_hip_debug_set(1, tile)
            yss, _ = hip.linear_ss(xs, ws, b)
            ess = (yss.double() - want).abs().max().item()
            alt.append(timeit(lambda: hip.linear_ss(xs, ws, b)))
            if ess > 1e-4:
                alt[-1] = -ess
        hip.load().lbm_hip_debug_set(1, 0)
        e32 = (y32.double() - want).abs().max().item()
        e16 = (y16.double() - want).abs().max().item()
        fl = 2.0 * M * N * K
        print(f"{M:6d} {N:6d} {K:5d} | {t32:7.1f} {fl / t32 / 1e6:5.1f} {e32:.1e} | {t16:7.1f} {fl / t16 / 1e6:6.1f} {e16:.1e} | " + " ".join(f"{t:7.1f}" for t in alt))


if __name__ == "__main__":
    main()
