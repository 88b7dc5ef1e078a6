"""Row-streaming product (gemm_rs.hip) against the tiled one (gemm16.hip): outputs must be bit-identical; then timings side by side.
usage: gemm_rs_check.py [check|time|both]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lbm_amd import hip
from gemm_bench import timeit

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
lib = hip.load()
r = np.random.default_rng(0)


def operands(M, N, K=256):
    xs = hip.SplitTensor((M, K), "cuda")
    xs.planes.copy_(torch.randn(2, M, K, device="cuda").half())
    ws = hip.SplitWeight(torch.from_numpy((r.normal(size=(N, K)) * 0.1).astype(np.float32)).cuda())
    b = torch.from_numpy(r.normal(size=(N,)).astype(np.float32)).cuda()
    return xs, ws, b


def run(flag, xs, ws, b, act, res, f32, split):
    lib.lbm_hip_debug_set(2, flag)
    y, ys = hip.linear_ss(xs, ws, b, act=act, res=res, res_scale=0.5, want_f32=f32, want_split=split)
    torch.cuda.synchronize()
    lib.lbm_hip_debug_set(2, 0)
    return y, ys


if mode in ("check", "both"):
    bad = 0
    for M, N in [(4096, 1024), (4100, 256), (33, 1456), (5000, 432), (16384, 1056), (1, 256), (31, 8), (70000, 1536), (257, 260)]:
        xs, ws, b = operands(M, N)
        res = torch.randn(M, N, device="cuda")
        for act, use_res, f32, split in [(0, False, True, False), (1, False, False, True), (2, True, True, True), (0, True, True, False)]:
            y0, s0 = run(256, xs, ws, b, act, res if use_res else None, f32, split)
            y1, s1 = run(512, xs, ws, b, act, res if use_res else None, f32, split)
            ok = (y0 is None or torch.equal(y0.view(torch.int32), y1.view(torch.int32))) and \
                 (s0 is None or torch.equal(s0.view(torch.int16), s1.view(torch.int16)))
            if not ok:
                bad += 1
                d = (y0 - y1).abs().max().item() if y0 is not None else (s0.float() - s1.float()).abs().max().item()
                nb = int((y0 != y1).sum()) if y0 is not None else int((s0 != s1).sum())
                print(f"MISMATCH M={M} N={N} act={act} res={use_res} f32={f32} split={split}: max diff {d:.3e}, {nb} elements", flush=True)
        print(f"checked {M} x {N}", flush=True)
    print("bit-identical" if bad == 0 else f"{bad} MISMATCHES", flush=True)

FLAGS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [256, 512, 512 + 4096]
if mode in ("time", "both"):
    print("flags (tiled = 256, rs = 512, rs dbg bits << 12):", FLAGS, flush=True)
    for M, N, act, f32, split in [(16384, 1024, 1, False, True), (16384, 1024, 0, True, False), (16384, 256, 0, True, False), (16384, 1456, 0, True, False),
                                  (10000, 1024, 1, False, True), (10000, 256, 0, True, False), (4096, 1024, 1, False, True), (4096, 256, 0, True, False),
                                  (2048, 1024, 1, False, True), (1024, 1024, 1, False, True), (1024, 256, 0, True, False),
                                  (147456, 1024, 1, False, True), (1044480, 1536, 0, True, False)]:
        xs, ws, b = operands(M, N)
        row = []
        for flag in FLAGS:
            lib.lbm_hip_debug_set(2, flag)
            us = timeit(lambda: hip.linear_ss(xs, ws, b, act=act, want_f32=f32, want_split=split), n=10)
            row.append(f"{us:9.1f} us {2.0 * M * N * 256 / us / 1e6:6.1f} TF")
        lib.lbm_hip_debug_set(2, 0)
        print(f"{M:8d} {N:5d} act{act} {'f32' if f32 else '   '} {'split' if split else '     '} | " + " | ".join(row), flush=True)
        del xs, ws
