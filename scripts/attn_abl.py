import sys, os
sys.path.insert(0, '/root/repo')
import torch
from lbm_amd import hip
lib = hip.load()
R, S = 10000, 48
g = torch.Generator(device="cpu").manual_seed(0)
U = (torch.randn(R, 4, 256, generator=g) * 0.15).cuda()
seq = torch.randn(R, S, 256, generator=g).cuda()
temb = (torch.randn(S, 256, generator=g) * 0.5).cuda()
ma = (torch.rand(R, S, generator=g) < 0.2).cuda().to(torch.uint8)
def t(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
print("temb+mask", t(lambda: hip.attn_readout(U, seq, S, ring_head=5, temb=temb, mask_a=ma, mask_b=None, want_valid=True)))
print("mask only", t(lambda: hip.attn_readout(U, seq, S, ring_head=5, temb=None, mask_a=ma, mask_b=None, want_valid=True)))
print("temb only", t(lambda: hip.attn_readout(U, seq, S, ring_head=5, temb=temb)))
print("plain", t(lambda: hip.attn_readout(U, seq, S, ring_head=5)))
x = torch.empty(R * S * 256, device="cuda"); y = torch.empty_like(x)
print("torch copy of the same bytes (read+write)", t(lambda: y.copy_(x)))
print("torch sum (read only)", t(lambda: x.sum()))
