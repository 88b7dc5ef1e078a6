"""Time of one msda_sample launch, direct kernel (debug key 22 = 0) against the LDS-staged window kernel (1), at the sizes of the
lattice launches of C3 (64 streams x 256 points) and C5 (10 000 points), k = 9 / 4, for reference cells that form a
neighbourhood ("block": a peaked correlation row) or lie anywhere ("scattered": a flat row).  usage: python scripts/msda_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import hip

lib = hip.load()
shapes = [(96, 128), (48, 64), (24, 32), (12, 16)]
L = sum(h * w for h, w in shapes)


H16 = len(sys.argv) > 1 and sys.argv[1] == "f16"       # value maps stored as f16 (the f16-storage mode)


def case(B, N, k, spread, off_scale=2.0, reps=5):
    g = torch.Generator(device="cpu").manual_seed(1)
    R = B * N
    value = torch.randn(B, L, 256, generator=g).cuda()
    if H16:
        value = value.half()
    offlog = torch.randn(R, 48 * 9, generator=g)
    offlog[:, :32 * 9] = (torch.rand(R, 32 * 9, generator=g) * 2 - 1) * off_scale
    offlog = offlog.cuda()
    centre = torch.rand(R, 1, 2, generator=g) * torch.tensor([512.0, 384.0])
    side = 3 if k == 9 else 2 if k == 4 else 1
    cells = torch.stack([torch.tensor([float(i % side), float(i // side)]) for i in range(k)], 0) * 4.0
    ref = ((centre // 4.0) * 4.0 + 2.0 + cells.unsqueeze(0)) if spread == "block" else torch.rand(R, k, 2, generator=g) * torch.tensor([512.0, 384.0])
    ref = ref.cuda().contiguous()
    res = {}
    outs = {}
    for mode in (0, 3):
        lib.lbm_hip_debug_set(22, mode)
        for _ in range(2):
            outs[mode] = hip.msda_sample(value, shapes, offlog, ref, B, N, k, 9, 512.0, 384.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            hip.msda_sample(value, shapes, offlog, ref, B, N, k, 9, 512.0, 384.0)
        e1.record()
        torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / reps * 1e3
    lib.lbm_hip_debug_set(22, 3)
    same = torch.equal(outs[0].view(torch.int32), outs[3].view(torch.int32))
    print(f"B={B:3d} N={N:5d} k={k} {spread:9s} offsets +-{off_scale}: direct {res[0]:8.1f} us  window {res[3]:8.1f} us  x{res[0] / res[3]:.2f}  bit-identical={same}", flush=True)


for B, N in ((64, 256), (1, 10000), (1, 1024)):
    for k in (9, 4):
        for spread in ("block", "scattered"):
            case(B, N, k, spread)
case(64, 256, 9, "block", off_scale=5.0)
for B, N in ((64, 256), (1, 10000), (1, 1024)):
    case(B, N, 1, "scattered")
