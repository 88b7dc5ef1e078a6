"""conv64_rows against the tiled kernel at few pixels (in-library HIP events).  usage: conv64_small.py [B ...]"""
import os
import sys

os.environ["LBM_PROF_SHAPES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import hip

lib = hip.load()
H, W = 96, 128
g = torch.Generator(device="cpu").manual_seed(0)
for B in [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8]:
    xs = hip.SplitTensor((B, H, W, 64), "cuda")
    xs.planes.copy_(torch.randn(2, B, H, W, 64, generator=g).half())
    w = hip.SplitWeight((torch.randn(64, 576, generator=g) * 0.05).cuda())
    b = torch.zeros(64, device="cuda")
    res = torch.randn(B, H, W, 64, device="cuda")
    for mode, spg in ((15, 0), (16, 0), (16, 1), (16, 3), (16, 4), (16, 6)):
        lib.lbm_hip_debug_set(4, mode)
        lib.lbm_hip_debug_set(18, spg << 8)
        for use_res, f32 in ((False, False), (True, True)):
            for _ in range(3):
                hip.conv2d_nhwc_ss(xs, w, b, 3, 1, 1, True, res if use_res else None, f32, True)
            torch.cuda.synchronize()
            hip.profile_begin()
            for _ in range(20):
                hip.conv2d_nhwc_ss(xs, w, b, 3, 1, 1, True, res if use_res else None, f32, True)
            p = hip.profile_end()
            print(f"B={B} mode={mode} spg={spg} res={use_res} f32={f32}: " + ", ".join(f"{q['name']} {1e3 * q['total_ms'] / q['launches']:.1f} us" for q in p), flush=True)
lib.lbm_hip_debug_set(4, 0)
lib.lbm_hip_debug_set(18, 0)
