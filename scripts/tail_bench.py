"""Encoder tail (lbm_hip_encoder_tail_fused) and correlation maps at 64 streams: time per launch for debug-key variants.
usage: python scripts/tail_bench.py [key:value,...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


ref = None
for variant in sys.argv[1:] or ["24:0", "24:1"]:
    k, v = (int(x) for x in variant.split(":"))
    lib.lbm_hip_debug_set(k, v)
    t = timeit(lambda: hip.encoder_tail_fused(f1, w1, b1, q2, q3, pos, tok, toks, h, w, inv))
    torch.cuda.synchronize()
    fp = (tok.clone(), toks.planes.clone(), inv.clone())
    same = ref is None or all(torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a.view(torch.int16), b.view(torch.int32) if b.dtype == torch.float32 else b.view(torch.int16)) for a, b in zip(ref, fp))
    ref = ref or fp
    print(f"encoder_tail_fused key {k} = {v}: {t:.1f} us  bit-identical to the first variant: {same}", flush=True)
    lib.lbm_hip_debug_set(k, 1 if k in (24,) else 0)
