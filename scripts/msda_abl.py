"""Ablation of msda_window_kernel (tuning build: make -C lbm_amd/csrc tuning; LBM_AMD_LIB=lbm_amd/liblbm_hip_tuning.so): time of the
k = 9 launch at C3 size with parts switched off (debug key 23: 1 no tap table, 2 no staging, 4 no accumulation)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import hip
lib = hip.load()
assert lib.lbm_hip_tuning_build(), "needs the tuning build"
shapes = [(96, 128), (48, 64), (24, 32), (12, 16)]
L = sum(h * w for h, w in shapes)
B, N = 64, 256
g = torch.Generator(device="cpu").manual_seed(1)
R = B * N
value = torch.randn(B, L, 256, generator=g).cuda()
for k in (9, 4):
    offlog = torch.randn(R, 48 * 9, generator=g)
    offlog[:, :32 * 9] = (torch.rand(R, 32 * 9, generator=g) * 2 - 1) * 2.0
    offlog = offlog.cuda()
    centre = torch.rand(R, 1, 2, generator=g) * torch.tensor([512.0, 384.0])
    side = 3 if k == 9 else 2
    cells = torch.stack([torch.tensor([float(i % side), float(i // side)]) for i in range(k)], 0) * 4.0
    ref = ((centre // 4.0) * 4.0 + 2.0 + cells.unsqueeze(0)).cuda().contiguous()
    for bits in (0, 1, 2, 4, 3, 5, 6, 7):
        lib.lbm_hip_debug_set(23, bits)
        for _ in range(2):
            hip.msda_sample(value, shapes, offlog, ref, B, N, k, 9, 512.0, 384.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            hip.msda_sample(value, shapes, offlog, ref, B, N, k, 9, 512.0, 384.0)
        e1.record()
        torch.cuda.synchronize()
        print(f"k={k} off: {'tap-table ' if bits & 1 else ''}{'staging ' if bits & 2 else ''}{'accumulate' if bits & 4 else ''} -> {e0.elapsed_time(e1) / 5 * 1e3:.1f} us", flush=True)
    lib.lbm_hip_debug_set(23, 0)
