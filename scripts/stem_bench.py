"""stem_pool alone at a batch of frames (in-library HIP events).  usage: stem_bench.py [B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import hip

g = torch.Generator(device="cpu").manual_seed(0)
w = torch.cat([torch.randn(64, 196, generator=g) * 0.05, torch.zeros(64, 28)], 1).contiguous()
ws = hip.SplitWeight(w.cuda())
b = torch.randn(64, generator=g).cuda()
for B in [int(v) for v in sys.argv[1:]] or [1, 8, 64]:
    xs = hip.SplitTensor((B, 384, 512, 4), "cuda")
    xs.planes.copy_(torch.randn(2, B, 384, 512, 4, generator=g).half())
    for _ in range(3):
        hip.stem_pool(xs, ws, b)
    torch.cuda.synchronize()
    hip.profile_begin()
    for _ in range(10):
        hip.stem_pool(xs, ws, b)
    p = hip.profile_end()
    print(f"B={B}: " + ", ".join(f"{q['name']} {1e3 * q['total_ms'] / q['launches']:.1f} us" for q in p), flush=True)
