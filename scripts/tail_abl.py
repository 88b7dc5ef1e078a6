"""Timing-only ablation of the fused encoder tail at 64 frames (tuning build: LBM_AMD_LIB=lbm_amd/liblbm_hip_tuning.so; debug key 24,
bits 16 = no level-0 product, 32 = no q2 / q3 / pos loads, 64 = no stores, 128 = no GELU).  usage: python scripts/tail_abl.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LBM_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lbm_amd", "liblbm_hip_tuning.so"))
import torch
from lbm_amd import hip
lib = hip.load()
B, h, w = 64, 96, 128
P0 = h * w
L = P0 + P0 // 4 + P0 // 16 + P0 // 64
g = torch.Generator(device="cpu").manual_seed(0)
f1 = hip.split_tensor(torch.randn(B, h, w, 64, generator=g).cuda())
w1 = hip.SplitWeight((torch.randn(256, 64, generator=g) * 0.2).cuda())
b1 = torch.randn(256, generator=g).cuda()
q2 = torch.randn(B, P0 // 4, 256, generator=g).cuda()
q3 = torch.randn(B, P0 // 16, 256, generator=g).cuda()
pos = torch.randn(P0, 256, generator=g).cuda()
tok = torch.zeros(B, L, 256, device="cuda")
toks = hip.SplitTensor((B, L, 256), "cuda")
inv = torch.zeros(B, L, device="cuda")


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


names = {0: "everything", 16: "no level-0 product", 32: "no q2/q3/pos loads", 64: "no stores", 128: "no GELU", 16 + 32: "no product, no loads",
         16 + 32 + 128: "stores only", 16 + 32 + 64 + 128: "skeleton", 64 + 128: "loads + product only", 32 + 64 + 128: "product only", 16 + 64 + 128: "tap loads only"}
for bits, nm in names.items():
    assert lib.lbm_hip_debug_set(24, 1 | bits) == 0
    t = timeit(lambda: hip.encoder_tail_fused(f1, w1, b1, q2, q3, pos, tok, toks, h, w, inv))
    print(f"{nm:28s} {t:8.1f} us", flush=True)
lib.lbm_hip_debug_set(24, 1)
