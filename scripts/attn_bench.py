"""attn_readout at the sizes of C5 (10 000 points, M = 48) and C3 (16 384 points, M = 12): time and GB/s of the two forms of the
soft-max step (debug key 3: + 16 = every lane evaluates every exponential).  usage: python scripts/attn_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import hip
lib = hip.load()
for R, S in ((10000, 48), (16384, 12), (10000, 96), (1024, 12)):
    g = torch.Generator(device="cpu").manual_seed(0)
    U = (torch.randn(R, 4, 256, generator=g) * 0.15).cuda()
    seq = torch.randn(R, S, 256, generator=g).cuda()
    temb = (torch.randn(S, 256, generator=g) * 0.5).cuda()
    ma = (torch.rand(R, S, generator=g) < 0.2).cuda().to(torch.uint8)
    mb = torch.zeros(R, S, dtype=torch.uint8).cuda()
    res = {}
    for mode in (16, 0, 4, 1, 2):
        lib.lbm_hip_debug_set(3, mode)
        f = lambda: hip.attn_readout(U, seq, S, ring_head=5 % S, temb=temb, mask_a=ma, mask_b=mb, want_valid=True)
        for _ in range(3):
            out = f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        res[mode] = (e0.elapsed_time(e1) / 10 * 1e3, out[0])
    lib.lbm_hip_debug_set(3, 0)
    nbytes = R * (S + 8) * 1024 + 2 * R * S
    same = torch.equal(res[0][1].view(torch.int32), res[16][1].view(torch.int32))
    print(f"   forced 4 waves per point {res[4][0]:.1f} us, forced 1 wave per point {res[1][0]:.1f} us, forced 2 {res[2][0]:.1f} us")
    print(f"R={R} S={S}: first form {res[16][0]:.1f} us ({nbytes / res[16][0] / 1e6:.2f} TB/s)  cross-lane {res[0][0]:.1f} us ({nbytes / res[0][0] / 1e6:.2f} TB/s)  bit-identical={same}", flush=True)
