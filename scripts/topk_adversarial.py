"""Fused correlation + top-k (lbm_hip_corr_topk = gemm_rs EPI 1, the kernel C3 / C5 run) on three kinds of correlation rows:

  random    rows that look like the flat synthetic checkpoint's (a new maximum is rare after the first few cells)
  peaked    one smooth bump per row (what the 'peaked' checkpoint and a trained tracker produce)
  ascending the adversarial case: every point's score INCREASES with the cell index, so every one of the 12288 cells beats the
            current k-th best of every lane and the k-deep sorted insertion runs for every score

The rows are produced by the product itself (q . F^T), so the maps are crafted through F: ascending = F[c] = (c / P0) u, q = u.
usage: python scripts/topk_adversarial.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lbm_amd import hip

lib = hip.load()
P0, GW = 12288, 128


def split(x):
    s = hip.SplitTensor(tuple(x.shape), "cuda")
    lib.lbm_hip_split_planes(x.data_ptr(), s.hi, s.planes[1].data_ptr(), x.numel(), None)
    torch.cuda.synchronize()
    return s


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


for B, N, tag in ((64, 256, "C3"), (1, 10000, "C5")):
    g = torch.Generator(device="cpu").manual_seed(0)
    u = torch.nn.functional.normalize(torch.randn(256, generator=g), dim=0)
    maps = {}
    q_rand = torch.nn.functional.normalize(torch.randn(B, N, 256, generator=g), dim=-1)
    maps["random"] = (q_rand, torch.nn.functional.normalize(torch.randn(B, P0, 256, generator=g), dim=-1))
    # peaked: F[c] = smooth random field over the grid (neighbouring cells alike), q = the field at the point's cell
    fld = torch.randn(B, 256, 24, 32, generator=g)
    fld = torch.nn.functional.interpolate(fld, size=(96, 128), mode="bilinear", align_corners=False).permute(0, 2, 3, 1).reshape(B, P0, 256)
    fld = torch.nn.functional.normalize(fld, dim=-1)
    cells = torch.randint(0, P0, (B, N), generator=g)
    maps["peaked"] = (torch.gather(fld, 1, cells.unsqueeze(-1).expand(B, N, 256)).contiguous(), fld.contiguous())
    ramp = (torch.arange(P0, dtype=torch.float32) / P0).view(1, P0, 1) * u.view(1, 1, 256)
    maps["ascending"] = (u.view(1, 1, 256).expand(B, N, 256).contiguous(), ramp.expand(B, P0, 256).contiguous())
    for name, (q, F) in maps.items():
        qs, Fs = split(q.cuda()), split(F.cuda())
        row = []
        for k in (9, 4, 1):
            t = timeit(lambda: hip.corr_topk(qs, Fs, k, GW))
            row.append(f"k={k}: {t:7.1f} us")
        idx, _, val = hip.corr_topk(qs, Fs, 9, GW)
        if name == "ascending":       # the answer is known: the last nine cells, best first
            assert torch.equal(idx[0, 0].cpu(), torch.arange(P0 - 1, P0 - 10, -1, dtype=torch.int32)), idx[0, 0]
        print(f"{tag} ({B} x {N} points) {name:9s} " + "   ".join(row), flush=True)
