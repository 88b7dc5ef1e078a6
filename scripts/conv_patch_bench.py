"""conv_patch against the tiled implicit GEMM on the layer-2 / layer-3 shapes of a batch of frames: time and agreement.
usage: python scripts/conv_patch_bench.py [frames = 64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lbm_amd import hip
lib = hip.load()
r = np.random.default_rng(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (H, W, Cin, Cout) in [(48, 64, 128, 128), (24, 32, 256, 256)]:
    x = hip.SplitTensor((B, H, W, Cin), "cuda")
    xf = torch.from_numpy(r.normal(size=(B, H, W, Cin)).astype(np.float32)).cuda()
    lib.lbm_hip_split_planes(xf.data_ptr(), x.hi, x.planes[1].data_ptr(), xf.numel(), None)
    w = hip.SplitWeight(torch.from_numpy((r.normal(size=(Cout, 9 * Cin)) * 0.03).astype(np.float32)).cuda())
    b = torch.from_numpy(r.normal(size=(Cout,)).astype(np.float32)).cuda()
    res = torch.from_numpy(r.normal(size=(B, H, W, Cout)).astype(np.float32)).cuda()
    for outs, kw in (("split only", dict(want_f32=False)), ("fp32 + split + residual", dict(res=res))):
        ref = None
        for rep in range(2):
            row = []
            for name, k4, k26 in (("tiled", 21, 0), ("patch auto", 22, 0), ("patch 192", 22, 192 << 8), ("patch 256", 22, 256 << 8)):
                lib.lbm_hip_debug_set(4, k4)
                lib.lbm_hip_debug_set(26, k26)
                y, ys = hip.conv2d_nhwc_ss(x, w, b, 3, 1, 1, True, **kw)
                got = ys.float() if y is None else y
                if ref is None:
                    ref = got.clone()
                err = float((got - ref).abs().max())
                t = timeit(lambda: hip.conv2d_nhwc_ss(x, w, b, 3, 1, 1, True, **kw))
                row.append(f"{name} {t:6.1f} us (max diff {err:.1e})")
            print((B, H, W, Cin, Cout), outs, "|", " | ".join(row), flush=True)
lib.lbm_hip_debug_set(4, 0)
lib.lbm_hip_debug_set(26, 0)
