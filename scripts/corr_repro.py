"""Block correlation-map kernel (debug key 9 = 1) alone on one stream, a chain of small products on another: does its output
ever differ from its own serial output?  Tokens alternate between two fixed pyramids, so a stale norm or row would show.
usage: corr_repro.py [iterations=300] [variant=1] [busy=1] [fresh=0]
fresh = 1: the pyramid is copied into ONE buffer on the same stream right before every call (a producer kernel just before)."""
import sys

import torch

sys.path.insert(0, ".")
from lbm_amd import hip  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 1
busy = int(sys.argv[3]) if len(sys.argv) > 3 else 1
fresh = int(sys.argv[4]) if len(sys.argv) > 4 else 0
h, w = 96, 128
P0 = h * w
L = P0 + P0 // 4 + P0 // 16 + P0 // 64
g = torch.Generator(device="cuda").manual_seed(3)
toks = []
for i in range(2):
    t = torch.zeros(1, L, 256, device="cuda")
    t[:, :P0] = torch.randn(1, P0, 256, device="cuda", generator=g) * (1.0 + i)
    hip.pyramid(t, h, w)
    toks.append(t)
mix = [0.3, -0.2, 0.1, 0.05, 0.2, 0.1, -0.1, 0.3, 0.1, 0.2, 0.3, -0.4]
lib = hip.load()
lib.lbm_hip_debug_set(9, variant)
ref = [hip.corr_maps(t, mix, 3, h, w).clone() for t in toks]
torch.cuda.synchronize()
# the other queue: a long chain of small products (the per-point chain's dominant kernel)
A, B = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.randn(1024, 256, device="cuda")
wt = torch.randn(1024, 256, device="cuda") * 0.05
bias = torch.zeros(1024, device="cuda")
bad = 0
worst = 0.0
T = torch.empty_like(toks[0])
for it in range(iters):
    if busy:
        with torch.cuda.stream(A):
            for _ in range(24):
                hip.linear(x, wt, bias)
    with torch.cuda.stream(B):
        if fresh:
            T.copy_(toks[it & 1])
        F = hip.corr_maps(T if fresh else toks[it & 1], mix, 3, h, w)
        d = (F - ref[it & 1]).abs().amax()
    B.synchronize()
    if float(d) != 0.0:
        bad += 1
        worst = max(worst, float(d))
torch.cuda.synchronize()
lib.lbm_hip_debug_set(9, 0)
print(f"variant {variant}, busy {busy}, fresh {fresh}: {bad} of {iters} outputs differ from the serial output (max abs diff {worst:.3g})")
