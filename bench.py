"""Benchmark of the LBM online tracking step on MI355X — the driver's contract.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c5] [--no-cpu-baseline]

One "step" = one full ``online_forward`` (feature encoder + tracker transformer + state update) on one new frame
for every stream the rank owns; frames are resident in HBM before the timed region.  Workloads (BASELINE.json
``configs``; weights = seeded synthetic checkpoint in the lbm.pt layout, videos = seeded synthetic):

  c2 (default, the configuration the metric is quoted on):  1 stream, 480x854, dense 32x32 grid = 1024 points, M = 12
  c3:  64 streams x 256 points, 480x854, batched in one step
  c5:  1 stream, 1080x1920, dense 100x100 grid = 10 000 points, memory M = 48

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank runs the same workload on its own streams
(weak scaling; streams are independent, no collective in the data path); the only collective is the RCCL gather of
the results at the end.  value = points x streams x ranks x K / max-over-ranks wall time.
"""
from __future__ import annotations

import argparse
import json
import pathlib
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOADS = {
    "c2": dict(desc="1 stream 480x854, dense 32x32 grid (1024 points), memory 12", B=1, grid=32, H=480, W=854, M=12, frames=8),
    "c3": dict(desc="64 streams 480x854 x 256 random points, memory 12", B=64, npts=256, H=480, W=854, M=12, frames=2),
    "c5": dict(desc="1 stream 1080x1920, dense 100x100 grid (10000 points), memory 48", B=1, grid=100, H=1080, W=1920, M=48, frames=4),
}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
F32_MFMA_PEAK_TF = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact fp32
F16_MFMA_PEAK_TF = 2500.0    # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (the split-precision kernels issue 3 MFMA
                             # flops per algorithmic flop, so their ceiling in algorithmic TFLOP/s is a third of it)


PRECISION_LABEL = {
    "split": "fp32 operands as split f16 hi/lo planes on the f16 matrix cores, fp32 accumulate (fp32-grade, passes the 1e-4 / 0.05 px parity bars)",
    "f32": "exact fp32 MFMA",
    "f16": "REDUCED precision: plain f16 operands on the matrix cores, fp32 accumulate (1e-2 logit bar); not the headline mode",
}


def build_model(wl, device):
    from lbm_amd.config import LBMConfig
    from lbm_amd.lbm_online import LBM_online
    from lbm_amd.synth import synthetic_state_dict
    cfg = LBMConfig(input_size=(384, 512), memory_size=12)
    model = LBM_online(cfg.to_namespace())
    sd = synthetic_state_dict(model, seed=42)
    model.load_state_dict(sd, strict=True)
    model = model.to(device).eval()
    if wl["M"] != 12:
        model.set_memory_size(wl["M"])
    return model, sd


def make_inputs(wl, rank):
    from lbm_amd.synth import synthetic_video, grid_queries, random_queries
    H, W, B = wl["H"], wl["W"], wl["B"]
    vids, pts = [], []
    base = synthetic_video(wl["frames"] + 1, H, W, seed=42 + rank)[0]            # (T,3,H,W)
    for s in range(B):
        # streams of a batch: the same clip shifted by s pixels (cheap to generate, different content per stream)
        vids.append(torch.roll(base, shifts=(3 * s) % W, dims=-1) if s else base)
        pts.append(grid_queries(H, W, wl["grid"]) if "grid" in wl else random_queries(wl["npts"], H, W, 42 + rank * 1000 + s))
    video = torch.stack(vids, 0)                                                 # (B,T,3,H,W)
    return video, torch.stack(pts, 0)                                            # (B,N,2)


def cpu_baseline(wl, sd, video, pts, budget_s=20.0):
    """The CPU oracle (oracle/lbm_ref_cpu.py, a port checked against the reference's outputs) on the host cores."""
    from oracle import lbm_ref_cpu as ref
    threads = min(16, os.cpu_count() or 1)          # the GPU box's CPU share for one GPU
    torch.set_num_threads(threads)
    g = ref.Geometry(size=(384, 512), memory_size=12)
    sd = {k: v.clone() for k, v in sd.items()}
    if wl["M"] != 12:
        ref.resize_time_embeddings(sd, g, wl["M"])
    o = ref.RefOnline(sd, g)
    n_streams = 1                                     # bounded sample: the first stream only
    with torch.no_grad():
        o.init(pts[0], video[0, :1])
        o.online_forward(video[0, 1:2])               # warm-up frame (fills one memory slot)
        t0 = time.perf_counter()
        frames = 0
        while frames < 3 or (time.perf_counter() - t0 < budget_s and frames < 30):
            o.online_forward(video[0, 1 + frames % wl["frames"]].unsqueeze(0))
            frames += 1
        dt = time.perf_counter() - t0
    N = pts.shape[1]
    return {"value": N * frames * n_streams / dt, "unit": "point-frames/s", "cores": threads, "kind": "port",
            "sample": f"{frames} frames of stream 0 ({N} points) of workload after 1 warm-up frame, fp32 PyTorch CPU oracle, "
                      f"{dt:.1f} s; {1000 * dt / frames:.0f} ms/frame"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--pipeline", action="store_true",
                    help="force the frame pipeline on (LBM_online.prefetch: frame t+1's encoder on a second HIP stream beside "
                         "frame t's per-point phase); default: decided by a short probe during warm-up")
    ap.add_argument("--no-pipeline", action="store_true", help="force strictly serial frames")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="gloo")

    from lbm_amd import hip
    for kv in os.environ.get("LBM_DEBUG_SET", "").split(","):        # tuning knobs, e.g. LBM_DEBUG_SET=3:4
        if kv:
            hip.load().lbm_hip_debug_set(int(kv.split(":")[0]), int(kv.split(":")[1]))
    model, sd = build_model(wl, device)
    video_cpu, pts_cpu = make_inputs(wl, rank)
    video = video_cpu.to(device)                       # (B,T,3,H,W) resident in HBM
    pts = pts_cpu.to(device)
    B, N = pts.shape[:2]
    nfr = wl["frames"]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        model.init_streams(pts, video[:, 0].contiguous())
        frames = [video[:, 1 + i].contiguous() for i in range(nfr)]
        tok = None
        # Frame pipeline (LBM_online.prefetch): encoder + value/correlation maps of frame t+1 on a second HIP stream beside
        # the per-point phase of frame t; bit-identical tracks (tests/test_hip_parity.py).  Whether two queues of one process
        # overlap or get in each other's way differs from box to box (measured both: -11 % and +30 % step time), so unless
        # forced (--pipeline / --no-pipeline, LBM_AMD_PIPELINE=0|1) both forms are timed on a few warm-up frames and the
        # faster one runs the timed steps.
        forced = os.environ.get("LBM_AMD_PIPELINE")
        if args.pipeline:
            forced = "1"
        if args.no_pipeline:
            forced = "0"
        pipeline = forced == "1"

        def run(n):
            # n complete frames: n encoder passes and n tracker passes, all inside the caller's barriers.  Pipelined, the
            # encoder of frame i+1 is enqueued on a second HIP stream before the per-point phase of frame i (the first
            # frame of a run is encoded inline, the last one prefetches nothing).
            nonlocal tok
            out = None
            for i in range(n):
                if pipeline:
                    if i + 1 < n:
                        model.prefetch(frames[(i + 1) % nfr])
                    out = model.online_forward_batch(frames[i % nfr])
                else:
                    out = model.online_forward_batch(frames[i % nfr], tokens=tok)
                    tok = out[3]
            return out

        probe = {}
        if forced is None:
            run(3)
            for mode in (False, True, False, True):
                pipeline, tok = mode, None
                barrier()
                tp = time.perf_counter()
                run(12)
                torch.cuda.synchronize()
                probe[mode] = min(probe.get(mode, 1e9), (time.perf_counter() - tp) / 12)
            pipeline = probe[True] < 0.97 * probe[False]
            if dist is not None:      # one decision for the whole job (rank 0's)
                flag = torch.tensor([1 if pipeline else 0], device=device if args.backend == "nccl" else "cpu")
                dist.broadcast(flag, 0)
                pipeline = bool(flag.item())
            tok = None

        run(args.warmup)
        barrier()
        t0 = time.perf_counter()
        coords, vis, rho, tok = run(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        if pipeline:
            tok = None

        # p50 per-frame latency: each frame synchronised on its own
        lat = []
        for i in range(min(args.steps, 50)):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            model.online_forward_batch(frames[i % nfr], tokens=tok)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t1) * 1e3)

        # per-kernel HIP-event timing of the same step (separate pass so the events do not perturb `elapsed`)
        prof = []
        if not args.no_profile and rank == 0:
            nprof = min(args.steps, 20)
            model.use_graph = False            # the per-launch events live in the eager path, not in graph replays
            hip.profile_begin()
            for i in range(nprof):
                model.online_forward_batch(frames[i % nfr], tokens=tok)
            prof = hip.profile_end()
            for p in prof:
                p["avg_us"] = 1e3 * p["total_ms"] / max(p["launches"], 1)
                p["per_step_us"] = 1e3 * p["total_ms"] / nprof

    # results travel over RCCL only here (gather of the last frame's tracks), as the north star asks
    from lbm_amd.sharding import gather_tracks, max_over_ranks, pack_tracks
    res = pack_tracks(coords, vis, rho)                                                          # (B, N, 4) records
    if args.backend == "gloo":
        res = res.cpu()
    gathered = gather_tracks(res, B * world, dst=0)
    if rank == 0:
        assert gathered.shape == (B * world, N, 4) and bool(torch.isfinite(gathered).all())
    elapsed = max_over_ranks(elapsed, device if args.backend == "nccl" else torch.device("cpu"))

    if rank == 0:
        total_pf = float(B) * N * world * args.steps
        line = {
            "metric": "point-frames/sec (whole node) dense-grid tracking; p50 per-frame latency",
            "value": total_pf / elapsed, "unit": "point-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if os.environ.get("LBM_AMD_PRECISION", "split") == "f16" else "f32", "data": "synthetic (seeded video, seeded lbm.pt-layout weights)",
            "config": {"workload": f"{args.workload}: {wl['desc']}", "streams_per_gpu": B, "points_per_stream": N,
                       "memory_size": wl["M"], "frame": [wl["H"], wl["W"]],
                       "frame_pipeline": ("encoder + value/correlation maps of frame t+1 on a second HIP stream, overlapping the "
                                          "per-point phase of frame t" if pipeline else "none (serial frames)") +
                                         (f"; chosen by warm-up probe: serial {1e3 * probe[False]:.3f} ms/frame, pipelined "
                                          f"{1e3 * probe[True]:.3f} ms/frame" if probe else "; forced"),
                       "precision": PRECISION_LABEL[os.environ.get("LBM_AMD_PRECISION", "split")], "parallelism": f"streams x{world} (no collective in the step)"},
            "p50_latency_ms": statistics.median(lat),
        }
        if prof:
            dom = max(prof, key=lambda p: p["total_ms"])
            is_mfma = dom["name"].startswith(("gemm_", "conv_"))
            sec = dom["total_ms"] / 1e3
            if is_mfma:
                ach = dom["flops"] / sec / 1e12            # ALGORITHMIC flops (2 M N K) per second
                f32 = dom["name"].startswith(("gemm_f32", "conv_f32"))
                peak = F32_MFMA_PEAK_TF if f32 else F16_MFMA_PEAK_TF
                line["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                                    "frac": ach / peak, "traffic": None,
                                    "mfma_flops_per_algorithmic_flop": 1 if (f32 or dom["name"].startswith(("gemm_sh", "conv_sh"))) else 3}
            else:
                ach = dom["bytes"] / sec / 1e9
                line["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": ach / HBM_PEAK_GBS, "traffic": None}
            # HBM-side bytes per launch of that kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate
            # runs, gfx950 corrections applied by scripts/pmc_traffic.py); counters cannot be read from inside this process
            pmc = sorted((pathlib.Path(__file__).parent / "profiles").glob(f"r*_{args.workload}_pmc.json"))
            if pmc:
                ent = json.loads(pmc[-1].read_text())["kernels"].get(dom["name"])
                if ent:
                    line["roofline"]["traffic"] = ent["traffic_bytes_per_launch"]
                    line["roofline"]["traffic_source"] = f"profiles/{pmc[-1].name}"
                    line["roofline"]["algorithmic_bytes_per_launch"] = dom["bytes"] / max(dom["launches"], 1)
            line["roofline"]["kernel"] = dom["name"]
            line["roofline"]["avg_launch_us"] = dom["avg_us"]
            line["roofline"]["launches_per_step"] = dom["launches"] / max(1, min(args.steps, 20))
            step_us = sum(p["per_step_us"] for p in prof)
            line["kernels"] = [{"name": p["name"], "launches_per_step": p["launches"] / min(args.steps, 20),
                                "avg_us": round(p["avg_us"], 2), "share": round(p["per_step_us"] / step_us, 4),
                                "TFLOP/s": round(p["flops"] / (p["total_ms"] / 1e3) / 1e12, 2) if p["flops"] else None,
                                "GB/s": round(p["bytes"] / (p["total_ms"] / 1e3) / 1e9, 1)}
                               for p in sorted(prof, key=lambda p: -p["total_ms"])[:12]]
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(wl, sd, video_cpu, pts_cpu)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
