import sys, os
sys.path.insert(0, os.getcwd())
import torch
from lbm_amd import hip
lib = hip.load()
B, h, w = 8, 96, 128
P0 = h * w
L = P0 + P0 // 4 + P0 // 16 + P0 // 64
g = torch.Generator(device="cpu").manual_seed(0)
f1 = hip.split_tensor(torch.randn(B, h, w, 64, generator=g).cuda())
w1 = hip.SplitWeight((torch.randn(256, 64, generator=g) * 0.2).cuda())
b1 = torch.randn(256, generator=g).cuda()
q2 = torch.randn(B, P0 // 4, 256, generator=g).cuda()
q3 = torch.randn(B, P0 // 16, 256, generator=g).cuda()
pos = torch.randn(P0, 256, generator=g).cuda()
outs = []
for v in (3, 1):
    lib.lbm_hip_debug_set(24, v)
    tok = torch.zeros(B, L, 256, device="cuda")
    toks = hip.SplitTensor((B, L, 256), "cuda")
    inv = torch.zeros(B, L, device="cuda")
    hip.encoder_tail_fused(f1, w1, b1, q2, q3, pos, tok, toks, h, w, inv)
    torch.cuda.synchronize()
    outs.append(tok.clone())
a, b = outs
d = (a.view(torch.int32) != b.view(torch.int32))
print("mismatching elements", int(d.sum()), "of", d.numel(), "max abs diff", float((a - b).abs().max()))
rows = d.any(-1)[0, :P0].view(h, w)
print("level-0 rows differing (stream 0):", int(rows.sum()), "of", P0)
ys, xs = torch.nonzero(rows, as_tuple=True)
print("y range", ys.min().item() if len(ys) else None, ys.max().item() if len(ys) else None, "x", xs.min().item() if len(xs) else None, xs.max().item() if len(xs) else None)
interior = rows[2:-2, 2:-2].sum().item()
print("interior rows differing:", interior)
print("torch.equal:", torch.equal(a, b))
