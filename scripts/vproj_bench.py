"""The value projection of 64 streams (1 044 480 x 1536 x 256, fp32 output) by the three forms of the split product:
debug key 2 = 256 tiled, 64 weights-stationary row streaming, 1024 rows-stationary.  usage: python scripts/vproj_bench.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import hip
lib = hip.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 16320
N = 1536
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(M, 256, generator=g).cuda()
xs = hip.SplitTensor((M, 256), "cuda")
lib.lbm_hip_split_planes(x.data_ptr(), xs.hi, xs.planes[1].data_ptr(), x.numel(), None)
del x
ws = hip.SplitWeight((torch.randn(N, 256, generator=g) * 0.1).cuda())
b = torch.randn(N, generator=g).cuda()
y = torch.empty(M, N, device="cuda")
ref = None
for flag, name in ((256, "tiled"), (64, "weights stationary"), (1024, "rows stationary")):
    lib.lbm_hip_debug_set(2, flag)
    f = lambda: hip.linear_ss_raw(A=xs.hi, a_plane=xs.plane, B=ws.hi, b_plane=N * 256, bias=(b.data_ptr() if len(sys.argv) < 3 else None), Y=y.data_ptr(), M=M, N=N, K=256,
                                  lda=256, ldb=256, ldy=N, batch=1, planes=2, res_scale=1.0)
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    chk = y[::997].clone()
    same = ref is None or torch.equal(ref.view(torch.int32), chk.view(torch.int32))
    ref = chk if ref is None else ref
    print(f"{name:20s} {t:7.3f} ms  {2.0 * M * N * 256 / t / 1e9:6.1f} algorithmic TF  {(M * 1024 + M * N * 4) / t / 1e9:5.2f} TB/s algorithmic  same bits as tiled: {same}", flush=True)
lib.lbm_hip_debug_set(2, 0)
