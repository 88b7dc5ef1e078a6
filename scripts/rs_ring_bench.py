"""gemm_rs with a 3-stage (steps fetched two ahead) against a 4-stage LDS ring (three ahead), debug key 25: time and bit-identity on
the shapes of C3 / C5 -- the value projection, the two many-row products of the lattice feed-forward, the fused correlation + top-k.
usage: python scripts/rs_ring_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import hip
lib = hip.load()


def timeit(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def split(x):
    s = hip.SplitTensor(tuple(x.shape), "cuda")
    lib.lbm_hip_split_planes(x.data_ptr(), s.hi, s.planes[1].data_ptr(), x.numel(), None)
    torch.cuda.synchronize()
    return s


g = torch.Generator(device="cpu").manual_seed(0)
for M, N, split_out, act, tag in ((64 * 16320, 1536, False, 0, "value projection C3"), (16320, 1536, False, 0, "value projection, 1 stream"),
                                  (147456, 256, False, 0, "lattice out-proj C3"), (147456, 1024, True, 1, "lattice FFN-1 C3 (GELU, split out)"),
                                  (90000, 1024, True, 1, "lattice FFN-1 C5"), (40000, 256, False, 0, "k=4 out-proj C5")):
    xs = split(torch.randn(M, 256, generator=g).cuda())
    ws = hip.SplitWeight((torch.randn(N, 256, generator=g) * 0.1).cuda())
    b = torch.randn(N, generator=g).cuda()
    y = None if split_out else torch.empty(M, N, device="cuda")
    ys = hip.SplitTensor((M, N), "cuda") if split_out else None
    res = {}
    for ring in (3, 4, 3, 4):
        lib.lbm_hip_debug_set(25, ring)
        f = lambda: hip.linear_ss_raw(A=xs.hi, a_plane=xs.plane, B=ws.hi, b_plane=N * 256, bias=b.data_ptr(), Y=(y.data_ptr() if y is not None else None),
                                      Ys=(ys.hi if ys is not None else None), ys_plane=(ys.plane if ys is not None else 0), ldys=N, act=act,
                                      M=M, N=N, K=256, lda=256, ldb=256, ldy=N, batch=1, planes=2, res_scale=1.0)
        t = timeit(f)
        out = (y if y is not None else ys.planes[0])[::499].clone()
        res.setdefault(ring, []).append((t, out))
    same = torch.equal(res[3][0][1].view(torch.int16 if split_out else torch.int32), res[4][0][1].view(torch.int16 if split_out else torch.int32))
    print(f"{tag:38s} {M:8d} x {N:5d}: ring 3 {res[3][0][0]:8.1f} / {res[3][1][0]:8.1f} us   ring 4 {res[4][0][0]:8.1f} / {res[4][1][0]:8.1f} us   same bits: {same}", flush=True)
    del xs, y, ys

P0, GW = 12288, 128
for B, N, tag in ((64, 256, "C3"), (1, 10000, "C5")):
    q = torch.nn.functional.normalize(torch.randn(B, N, 256, generator=g), dim=-1)
    F = torch.nn.functional.normalize(torch.randn(B, P0, 256, generator=g), dim=-1)
    qs, Fs = split(q.cuda()), split(F.cuda())
    for k in (9, 4, 1):
        res = {}
        for ring in (3, 4, 3, 4):
            lib.lbm_hip_debug_set(25, ring)
            t = timeit(lambda: hip.corr_topk(qs, Fs, k, GW), 10)
            idx, _, val = hip.corr_topk(qs, Fs, k, GW)
            res.setdefault(ring, []).append((t, idx.clone(), val.clone()))
        same = torch.equal(res[3][0][1], res[4][0][1]) and torch.equal(res[3][0][2], res[4][0][2])
        print(f"corr + top-{k} {tag}: ring 3 {res[3][0][0]:7.1f} / {res[3][1][0]:7.1f} us   ring 4 {res[4][0][0]:7.1f} / {res[4][1][0]:7.1f} us   same: {same}", flush=True)
lib.lbm_hip_debug_set(25, 3)
