"""Benchmark of the LBM online tracking step on MI355X — the driver's contract.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c5] [--no-cpu-baseline]

One "step" = one full ``online_forward`` (feature encoder + tracker transformer + state update) on one new frame
for every stream the rank owns; frames are resident in HBM before the timed region.  Workloads (BASELINE.json
``configs``; weights = seeded synthetic checkpoint in the lbm.pt layout, videos = seeded synthetic):

  c2 (default at N = 1, the configuration the metric is quoted on):  1 stream, 480x854, dense 32x32 grid = 1024 points, M = 12
  c3 (default at N > 1: BASELINE config 4 = 64 streams per GPU):     64 streams x 256 points, 480x854, batched in one step
  c5:  1 stream, 1080x1920, dense 100x100 grid = 10 000 points, memory M = 48
  c1:  1 stream, 480x854, 8 fixed "clicked" points (BASELINE config 1, the reference's own CPU-runnable case; demo_click.py:40-48):
       reported under ``workloads.c1`` as the CPU oracle's ms/frame and the GPU's p50 latency, eager and graph-replayed, plus the
       10-point figure the reference publishes its own numbers at (assets/tensorrt.md:106-107)

Two frame loops are timed on the same K frames from the same state: strictly serial frames, and the frame pipeline
(LBM_online.prefetch: encoder of frame t+1 on a second HIP stream).  ``value`` is the pipelined figure only when its final
tracks and logits of EVERY timed frame (a whole-run log the step writes into, no extra launches) and its final memories are
bit-identical to the serial loop's in this very run (``pipeline_bit_identical``) and it is faster;
otherwise the serial figure.  Both are in the line (``serial_frames``, ``frame_pipeline``).
At N = 1 the line also carries ``workloads``: the throughput regime (c3, c5) measured in the same process.

N > 1: ``python bench.py --gpus N`` starts N ranks itself (child ``torch.distributed.run`` before anything touches the
GPU) unless a launcher already did (WORLD_SIZE set).  Every rank runs the workload on its own streams (weak scaling; streams
are independent, no collective in the data path); the only collective is the RCCL gather of the results at the end.
value = points x streams x ranks x K / max-over-ranks wall time.
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
    "c1": dict(desc="1 stream 480x854, 8 fixed clicked points, memory 12", B=1, clicks=8, H=480, W=854, M=12, frames=8),
    "c1_10": dict(desc="1 stream 480x854, 10 fixed clicked points, memory 12", B=1, clicks=10, H=480, W=854, M=12, frames=8),
    "c2": dict(desc="1 stream 480x854, dense 32x32 grid (1024 points), memory 12", B=1, grid=32, H=480, W=854, M=12, frames=8),
    "c3": dict(desc="64 streams 480x854 x 256 random points, memory 12", B=64, npts=256, H=480, W=854, M=12, frames=2),
    "c5": dict(desc="1 stream 1080x1920, dense 100x100 grid (10000 points), memory 48", B=1, grid=100, H=1080, W=1920, M=48, frames=4),
}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
L2_GATHER_PEAK_GBS = 18800.0   # MI355X_MICROARCH.md: L2 shared-row gather 16.8-18.8 TB/s; the upper figure (C5 measures 17.5 TB/s)
F32_MFMA_PEAK_TF = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact fp32
F16_MFMA_PEAK_TF = 2500.0    # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (the split-precision kernels issue 3 MFMA
                             # flops per algorithmic flop, so their ceiling in algorithmic TFLOP/s is a third of it)


PRESET_LABEL = {
    "flat": "seeded synthetic lbm.pt-layout checkpoint, preset 'flat' (nearly flat correlation rows: the k best cells of a point lie anywhere on the map)",
    "peaked": "seeded synthetic lbm.pt-layout checkpoint, preset 'peaked' (one smooth bump per correlation row, as a trained tracker's: the k best "
              "cells are a neighbourhood of the point)",
}
PRECISION_LABEL = {
    "split": "fp32 operands as split f16 hi/lo planes on the f16 matrix cores, fp32 accumulate (fp32-grade, passes the 1e-4 / 0.05 px parity bars)",
    "f32": "exact fp32 MFMA",
    "f16": "REDUCED precision: plain f16 operands on the matrix cores, fp32 accumulate (1e-2 logit bar); not the headline mode",
    "f16s": "REDUCED precision, f16 STORAGE: both memories and the six value maps f16 in HBM, tracker products on plain f16 operands, but the "
            "encoder and the correlation product + top-k fp32-grade (split) so the arg-max is decided on fp32-grade scores (1e-2 logit bar, "
            "0.05 px); a secondary line, never the headline",
}


def build_model(wl, device, preset="flat", precision=None):
    from lbm_amd.config import LBMConfig
    from lbm_amd.lbm_online import LBM_online
    from lbm_amd.synth import synthetic_state_dict
    cfg = LBMConfig(input_size=(384, 512), memory_size=12)
    model = LBM_online(cfg.to_namespace())
    sd = synthetic_state_dict(model, seed=42, preset=preset)
    model.load_state_dict(sd, strict=True)
    if precision is not None:
        model.set_precision(precision)
    model = model.to(device).eval()
    if wl["M"] != 12:
        model.set_memory_size(wl["M"])
    return model, sd


def clicked_points(n, H, W, seed=42):
    """BASELINE config 1's "clicked" query points: a GUI click is impossible headless (demo_utils.py:73-93), so n fixed seeded
    points, uniform in the central 80 % of the frame, integer pixels like a click."""
    import numpy as np
    g = np.random.Generator(np.random.PCG64(seed * 7919 + n))
    xy = np.rint(g.uniform([0.1 * W, 0.1 * H], [0.9 * W, 0.9 * H], (n, 2))).astype("float32")
    return torch.from_numpy(xy)


def make_inputs(wl, rank):
    from lbm_amd.synth import synthetic_video, grid_queries, random_queries
    H, W, B = wl["H"], wl["W"], wl["B"]
    vids, pts = [], []
    base = synthetic_video(wl["frames"] + 1, H, W, seed=42 + rank)[0]            # (T,3,H,W)
    for s in range(B):
        # streams of a batch: the same clip shifted by s pixels (cheap to generate, different content per stream)
        vids.append(torch.roll(base, shifts=(3 * s) % W, dims=-1) if s else base)
        pts.append(clicked_points(wl["clicks"], H, W) if "clicks" in wl else grid_queries(H, W, wl["grid"]) if "grid" in wl else
                   random_queries(wl["npts"], H, W, 42 + rank * 1000 + s))
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


def step_flops(wl, N, B):
    """ALGORITHMIC flops of one step: 84.7 MFLOP per point-frame at M = 12 (+1.573 per extra memory slot)
    and 25.2 GFLOP per stream-frame (encoder + the six value projections)."""
    return B * N * (84.7e6 + 1.573e6 * (wl["M"] - 12)) + B * 25.2e9


def roofline_of(prof, steps_profiled):
    """Roofline entry of the dominant kernel of a HIP-event profile (None without one)."""
    if not prof:
        return None, []
    dom = max(prof, key=lambda p: p["total_ms"])
    sec = dom["total_ms"] / 1e3
    if dom["name"].startswith(("gemm_", "conv_", "ffn_")):
        ach = dom["flops"] / sec / 1e12            # ALGORITHMIC flops (2 M N K) per second
        f32 = dom["name"].startswith(("gemm_f32", "conv_f32"))
        peak = F32_MFMA_PEAK_TF if f32 else F16_MFMA_PEAK_TF
        roof = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                "mfma_flops_per_algorithmic_flop": 1 if (f32 or dom["name"].startswith(("gemm_sh", "gemm_rh", "conv_sh", "ffn_fused_sh"))) else 3}
    elif dom["name"].startswith(("msda_sample", "msda_window")):
        # a gather out of a 16.7 MB map that every workgroup shares: served by the L2s, not by HBM -- priced against the
        # guide's L2 shared-row rate, reported under bound "hbm" would exceed 1
        ach = dom["bytes"] / sec / 1e9
        roof = {"bound": "l2-gather", "achieved": ach, "peak": L2_GATHER_PEAK_GBS, "unit": "GB/s", "frac": ach / L2_GATHER_PEAK_GBS,
                "traffic": None}
    else:
        ach = dom["bytes"] / sec / 1e9
        roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None}
    roof["kernel"] = dom["name"]
    roof["avg_launch_us"] = dom["avg_us"]
    roof["launches_per_step"] = dom["launches"] / max(1, steps_profiled)
    step_us = sum(p["per_step_us"] for p in prof)
    kernels = [{"name": p["name"], "launches_per_step": p["launches"] / steps_profiled, "avg_us": round(p["avg_us"], 2),
                "share": round(p["per_step_us"] / step_us, 4),
                "TFLOP/s": round(p["flops"] / (p["total_ms"] / 1e3) / 1e12, 2) if p["flops"] else None,
                "GB/s": round(p["bytes"] / (p["total_ms"] / 1e3) / 1e9, 1)}
               for p in sorted(prof, key=lambda p: -p["total_ms"])[:12]]
    return roof, kernels


class Runner:
    """One workload on this rank's GPU: model, resident frames, the serial / pipelined frame loops and a state snapshot."""

    def __init__(self, wl, rank, device, preset="flat", precision=None):
        self.wl = wl
        self.model, self.sd = build_model(wl, device, preset, precision)
        self.video_cpu, self.pts_cpu = make_inputs(wl, rank)
        video = self.video_cpu.to(device)                  # (B,T,3,H,W) resident in HBM
        self.pts = self.pts_cpu.to(device)
        self.B, self.N = self.pts.shape[:2]
        self.nfr = wl["frames"]
        self.model.init_streams(self.pts, video[:, 0].contiguous())
        self.frames = [video[:, 1 + i].contiguous() for i in range(self.nfr)]
        self.tok = None

    def new_log(self, n):
        """Result tensors of n frames, which the step writes in place (LBM_online.online_forward_batch(out=...)): the tracks and
        logits of EVERY timed frame are kept and compared between the two frame loops at no cost inside the timed region."""
        dev, B, N = self.pts.device, self.B, self.N
        f = lambda *sh: torch.zeros(n, B, N, *sh, device=dev)
        return {"coords": f(2), "vis": f(), "rho": f(), "offset": f(2), "rho_prob": f(),
                "vis_flag": torch.zeros(n, B, N, device=dev, dtype=torch.bool)}

    def run(self, n, pipeline=False, first=0, log=None):
        """n complete frames (n encoder passes, n tracker passes).  Pipelined, the encoder of frame i+1 is enqueued on a second
        HIP stream before the per-point phase of frame i (the first frame is encoded inline, the last prefetches nothing)."""
        m, fr, nfr = self.model, self.frames, self.nfr
        out = None
        for i in range(first, first + n):
            dst = None if log is None else {k: v[i - first] for k, v in log.items()}
            if pipeline:
                if i + 1 < first + n:
                    m.prefetch(fr[(i + 1) % nfr])
                out = m.online_forward_batch(fr[i % nfr], out=dst)
            else:
                out = m.online_forward_batch(fr[i % nfr], tokens=self.tok, out=dst)
                self.tok = out[3]
        if pipeline:
            self.tok = None
        return out

    def snapshot(self):
        m = self.model
        return (m._stream.clone(), m._coll.clone(), m._vis.clone(), m._mem.clone(), m._head, m.t)

    def restore(self, snap):
        m = self.model
        m._stream.copy_(snap[0]); m._coll.copy_(snap[1]); m._vis.copy_(snap[2]); m._mem.copy_(snap[3])
        m._head, m.t = snap[4], snap[5]

    def fingerprint(self, log):
        """Everything a frame loop leaves behind: the tracks, logits and flags of every frame, and both memories with their masks."""
        m = self.model
        return [log[k] for k in sorted(log)] + [m._stream.clone(), m._coll.clone(), m._vis.clone(), m._mem.clone()]

    def time_loops(self, steps, warmup, barrier, want_pipeline=True):
        """Both frame loops on the same `steps` frames from the same state, each bracketed by `barrier` (synchronize, and the
        process-group barrier at N > 1).  -> (seconds serial, seconds pipelined or None, bit-identical?, last frame's outputs)."""
        self.run(warmup)
        torch.cuda.synchronize()
        snap = self.snapshot()
        log_s = self.new_log(steps)
        barrier()
        t0 = time.perf_counter()
        out = self.run(steps, log=log_s)
        barrier()
        el_serial = time.perf_counter() - t0
        fp_serial = self.fingerprint(log_s)
        el_pipe, same = None, False
        if want_pipeline:
            self.restore(snap)
            self.run(2, pipeline=True)                            # (its first use allocates the second slot)
            self.restore(snap)
            log_p = self.new_log(steps)
            barrier()
            t1 = time.perf_counter()
            self.run(steps, pipeline=True, log=log_p)
            barrier()
            el_pipe = time.perf_counter() - t1
            same = all(torch.equal(a, b) for a, b in zip(fp_serial, self.fingerprint(log_p)))
        return el_serial, el_pipe, same, (out[0], out[1], out[2])

    def p50(self, n, graph=False):
        """Median per-frame latency, every (serial) frame synchronised on its own; graph=True: the frame step replayed as a HIP
        graph (LBM_online.use_graph, one graph per ring position).  None if the capture was not possible."""
        m = self.model
        if graph and getattr(m, "use_graph", None) is None:
            return None
        try:
            if graph:
                m.use_graph = True
                self.run(self.wl["M"] + 2)                    # first use of a ring position captures its graph
            lat = []
            for i in range(n):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                self.run(1, first=i)
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t1) * 1e3)
            if graph and not m.use_graph:                     # (a failed capture switches the model back to eager launches)
                return None
            return statistics.median(lat)
        finally:
            if graph:
                m.use_graph = False

    def profile(self, nprof):
        from lbm_amd import hip
        self.model.use_graph = False            # the per-launch events live in the eager path, not in graph replays
        hip.profile_begin()
        self.run(nprof)
        prof = hip.profile_end()
        for p in prof:
            p["avg_us"] = 1e3 * p["total_ms"] / max(p["launches"], 1)
            p["per_step_us"] = 1e3 * p["total_ms"] / nprof
        return prof


def attach_traffic(roof, prof, wname):
    """(prof may be None: then only the counter figure is attached)"""
    return _attach_traffic(roof, prof, wname)


def _attach_traffic(roof, prof, wname):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate
    runs, gfx950 corrections applied by scripts/pmc_traffic.py); counters cannot be read from inside this process."""
    pmc = sorted((pathlib.Path(__file__).parent / "profiles").glob(f"r*_{wname}_pmc.json"))
    if not pmc:
        return
    ent = json.loads(pmc[-1].read_text())["kernels"].get(roof["kernel"])
    if ent:
        roof["traffic"] = ent["traffic_bytes_per_launch"]
        roof["traffic_source"] = f"profiles/{pmc[-1].name}"
        dom = next((p for p in (prof or []) if p["name"] == roof["kernel"]), None)
        if dom is not None:
            roof["algorithmic_bytes_per_launch"] = dom["bytes"] / max(dom["launches"], 1)


def measure_throughput(name, steps, warmup, rank, device, sync, profile=True, precision=None, preset="flat"):
    """One entry of `workloads`: the same Runner.time_loops as the headline (so `--workload c3` and `workloads.c3` are the same
    measurement up to the step count), plus the roofline entry of its dominant kernel."""
    w2 = WORKLOADS[name]
    r2 = Runner(w2, rank, device, preset, precision)
    e_s, e_p, same2, _ = r2.time_loops(steps, warmup, sync)
    prof2 = r2.profile(4) if profile else []
    roof2, kern2 = roofline_of(prof2, 4)
    if roof2 is not None and precision is None:
        attach_traffic(roof2, prof2, name)
    fl = step_flops(w2, r2.N, r2.B)
    ent = {"workload": w2["desc"], "weights": PRESET_LABEL[preset], "steps": steps, "ms_per_step": 1e3 * e_s / steps,
           "value": float(r2.B) * r2.N * steps / e_s, "unit": "point-frames/s",
           "dtype": "f16" if precision in ("f16", "f16s") else "f32",
           "step_algorithmic_TFLOPs": fl / (e_s / steps) / 1e12,
           "step_frac_of_f16_mfma_peak": fl / (e_s / steps) / 1e12 / F16_MFMA_PEAK_TF,
           "frame_pipeline": {"ms_per_step": 1e3 * e_p / steps, "value": float(r2.B) * r2.N * steps / e_p, "bit_identical": same2},
           "roofline": roof2, "kernels": kern2[:6]}
    if precision is not None:
        ent["precision"] = PRECISION_LABEL[precision]
    del r2
    torch.cuda.empty_cache()
    return ent


def measure_clicks(name, rank, device, cpu=False):
    """BASELINE config 1 (8 clicked points, one 480x854 stream; the 10-point variant is what assets/tensorrt.md:104-109 quotes):
    a latency regime -- p50 of single synchronised frames, eager and graph-replayed -- and, for c1 itself, the CPU oracle."""
    w = WORKLOADS[name]
    r = Runner(w, rank, device)
    r.run(10)                                         # warm-up as benchmark_torch.py:61-62
    lat = r.p50(50)
    lat_g = r.p50(50, graph=True)
    ent = {"workload": w["desc"], "p50_latency_ms": lat, "p50_latency_graph_replay_ms": lat_g, "frames_timed": 50, "dtype": "f32",
           "value": r.N / (min(x for x in (lat, lat_g) if x is not None) / 1e3), "unit": "point-frames/s"}
    if name == "c1_10":
        ent["reference_published"] = "assets/tensorrt.md:106-107: 20.50 ms PyTorch / 3.84 ms TensorRT-FP16 at 10 points on an RTX 4090 (other hardware, real weights)"
    if cpu:
        ent["cpu_baseline"] = cpu_baseline(w, r.sd, r.video_cpu, r.pts_cpu, budget_s=6.0)
    del r
    torch.cuda.empty_cache()
    return ent


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child torch.distributed.run (this process has not
    touched the GPU and never will) and pass its output and exit code on."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c2 on every GPU (at --gpus > 1 the line also carries c3 per rank = BASELINE config 4 as workloads.c3_per_rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extra-workloads", action="store_true", help="N = 1: skip the c3 / c5 entries of `workloads`")
    ap.add_argument("--pipeline", action="store_true", help="(default behaviour; kept for compatibility)")
    ap.add_argument("--no-pipeline", action="store_true", help="serial frames only: the frame pipeline is neither timed nor used")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--preset", default="flat", choices=["flat", "peaked"], help="synthetic checkpoint preset (lbm_amd/synth.py); the headline "
                    "figures of every round were taken on 'flat'")
    ap.add_argument("--allow-knobs", action="store_true", help="tuning run with lbm_hip_debug_set knobs off their defaults (LBM_DEBUG_SET) "
                    "or a -DLBM_TUNING library: the line carries headline_valid = false")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    # the same workload per GPU at every N (weak scaling: the N = 1 headline, BASELINE config 2, on every rank); BASELINE config 4
    # (64 streams per GPU) rides along as workloads.c3_per_rank
    wname = args.workload or "c2"
    wl = WORKLOADS[wname]
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
            hip._check(hip.load().lbm_hip_debug_set(int(kv.split(":")[0]), int(kv.split(":")[1])), "LBM_DEBUG_SET")
    # A figure measured with a knob off its default is a tuning experiment, not a headline: the knobs are printed in the line
    # (`config.knobs`) and the run is refused unless --allow-knobs says it is meant (then `headline_valid` is false).
    knobs = hip.debug_active()
    env_knobs = {k: os.environ[k] for k in sorted(os.environ) if k.startswith("LBM_AMD_") and k not in ("LBM_AMD_LIB",)}
    tuning_lib = bool(hip.load().lbm_hip_tuning_build())
    if (knobs or tuning_lib) and not args.allow_knobs:
        raise SystemExit(f"bench.py: debug knobs {knobs} are set / tuning build = {tuning_lib}; a headline needs the product library at its "
                         f"defaults (pass --allow-knobs for a tuning run: the line then says headline_valid = false)")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def local_sync():
        torch.cuda.synchronize()

    # Both frame loops are timed on the same K frames from the same starting state, each bracketed by barrier + synchronize.
    # The headline is the frame pipeline ONLY IF the tracks, logits and flags of every timed frame and the final memories are
    # bit-identical to the serial run's on every rank (pipeline_bit_identical); otherwise -- or with --no-pipeline -- it is the
    # serial figure.
    want_pipeline = not args.no_pipeline and os.environ.get("LBM_AMD_PIPELINE") != "0"
    from lbm_amd.sharding import gather_tracks, max_over_ranks, pack_tracks
    cdev = device if args.backend == "nccl" else torch.device("cpu")
    with torch.no_grad():
        r = Runner(wl, rank, device, args.preset)
        B, N = r.B, r.N
        el_serial, el_pipe, same, (coords, vis, rho) = r.time_loops(args.steps, args.warmup, barrier, want_pipeline)
        my_serial, my_pipe = el_serial, el_pipe
        if dist is not None and want_pipeline:                 # one verdict for the whole job: identical on EVERY rank
            flag = torch.tensor([1 if same else 0], device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            same = bool(flag.item())

        # results travel over RCCL only here (gather of the last frame's tracks), as the north star asks -- straight after the
        # timed loops on every rank, nothing (no latency or profiling pass) in between
        res = pack_tracks(coords, vis, rho)                                                          # (B, N, 4) records
        if args.backend == "gloo":
            res = res.cpu()
        gathered = gather_tracks(res, B * world, dst=0)
        if rank == 0:
            assert gathered.shape == (B * world, N, 4) and bool(torch.isfinite(gathered).all())
        el_serial = max_over_ranks(el_serial, cdev)
        if el_pipe is not None:
            el_pipe = max_over_ranks(el_pipe, cdev)
        ranks = None
        if dist is not None:
            mine = {"rank": rank, "device": torch.cuda.get_device_name(device), "device_index": local,
                    "serial_ms_per_step": 1e3 * my_serial / args.steps,
                    "pipelined_ms_per_step": None if my_pipe is None else 1e3 * my_pipe / args.steps}
            ranks = [None] * world
            dist.all_gather_object(ranks, mine)

        lat = lat_graph = None
        prof, extra = [], {}
        if dist is not None and args.workload is None and not args.no_extra_workloads:
            # BASELINE config 4 (64 streams x 256 points per GPU: 512 streams at N = 8) on every rank as well, bracketed by the same
            # process-group barrier and reduced with MAX over ranks: its scaling ratio is formed against the N = 1 line's
            # `workloads.c3` (c3 / c3), the headline's against that line's `value` (c2 / c2: the same per-GPU work at every N)
            def per_rank_entry(name, steps):
                e2 = measure_throughput(name, steps, min(args.warmup, 4), rank, device, barrier, profile=False, preset=args.preset)
                pf2 = e2["value"] * e2["ms_per_step"] / 1e3 * world            # point-frames per step, all ranks
                e2["ms_per_step"] = 1e3 * max_over_ranks(e2["ms_per_step"] / 1e3, cdev)
                e2["frame_pipeline"]["ms_per_step"] = 1e3 * max_over_ranks(e2["frame_pipeline"]["ms_per_step"] / 1e3, cdev)
                e2["value"] = pf2 / (e2["ms_per_step"] / 1e3)
                e2["frame_pipeline"]["value"] = pf2 / (e2["frame_pipeline"]["ms_per_step"] / 1e3)
                e2["frame_pipeline"]["bit_identical"] = max_over_ranks(0.0 if e2["frame_pipeline"]["bit_identical"] else 1.0, cdev) == 0.0   # on EVERY rank
                e2["serial_frames"] = {"ms_per_step": e2["ms_per_step"], "value": e2["value"]}
                if e2["frame_pipeline"]["bit_identical"] and e2["frame_pipeline"]["value"] > e2["value"]:      # the headline rule of the N = 1 line
                    e2["ms_per_step"], e2["value"], e2["headline_mode"] = e2["frame_pipeline"]["ms_per_step"], e2["frame_pipeline"]["value"], "frame_pipeline"
                else:
                    e2["headline_mode"] = "serial_frames"
                e2["aggregate_over_ranks"] = world
                for key in ("roofline", "kernels", "step_algorithmic_TFLOPs", "step_frac_of_f16_mfma_peak"):
                    e2.pop(key, None)
                return e2
            extra["c3_per_rank"] = per_rank_entry("c3", min(args.steps, 20))
        nprof = min(args.steps, 20)
        if world > 1 and rank == 0 and not args.no_profile:
            # the line's `roofline` at N > 1 as well: rank 0's per-kernel HIP-event pass over its own (identical) workload, after every
            # collective of the run; the other ranks wait in the closing barrier meanwhile
            prof = r.profile(nprof)
        if world == 1:
            # p50 per-frame latency: each (serial) frame synchronised on its own; and the same with the frame step replayed as a
            # HIP graph: what a latency-bound caller would run; reported beside the eager figure, never instead of it
            lat = r.p50(min(args.steps, 50))
            if not args.no_profile:
                lat_graph = r.p50(min(args.steps, 50), graph=True)
                # per-kernel HIP-event timing of the same step (separate pass so the events do not perturb `elapsed`)
                prof = r.profile(nprof)
        sd, video_cpu, pts_cpu = r.sd, r.video_cpu, r.pts_cpu

        # N = 1: the throughput regime (BASELINE configs 3 and 5) and the click regime (config 1) in the same line
        if world == 1 and not args.no_extra_workloads and args.workload is None:
            del r
            torch.cuda.empty_cache()
            for name, steps in (("c3", 12), ("c5", 30)):
                extra[name] = measure_throughput(name, steps, 4, rank, device, local_sync, not args.no_profile, preset=args.preset)
                # the same workload on the other checkpoint preset: where the reference cells of a point lie decides how much of
                # the deformable sampling the LDS window catches (and how often the fused top-k inserts)
                other = "peaked" if args.preset == "flat" else "flat"
                extra[f"{name}_{other}"] = measure_throughput(name, steps, 4, rank, device, local_sync, not args.no_profile, preset=other)
                # the reduced-precision line (f16 storage; fp32-grade encoder and correlation + top-k): secondary, dtype stated per entry
                extra[f"{name}_f16s"] = measure_throughput(name, steps, 4, rank, device, local_sync, not args.no_profile, precision="f16s",
                                                           preset=args.preset)
            for name in ("c1", "c1_10"):
                extra[name] = measure_clicks(name, rank, device, cpu=(name == "c1" and not args.no_cpu_baseline))
    headline_pipelined = el_pipe is not None and same and el_pipe < el_serial
    elapsed = el_pipe if headline_pipelined else el_serial

    if rank == 0:
        total_pf = float(B) * N * world * args.steps
        mode_txt = ("encoder + value/correlation maps of frame t+1 on a second HIP stream, overlapping the per-point phase of frame t; "
                    "every timed frame verified bit-identical to the serial frame loop in this run" if headline_pipelined else
                    "none in the headline (strictly serial frames)" + ("" if el_pipe is None else
                                                                      ": the pipelined loop was " + ("not bit-identical" if not same else "not faster")))
        prec = os.environ.get("LBM_AMD_PRECISION", "split")
        line = {
            "metric": "point-frames/sec (whole node) dense-grid tracking; p50 per-frame latency",
            "value": total_pf / elapsed, "unit": "point-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if prec in ("f16", "f16s") else "f32", "data": f"synthetic (seeded video, seeded lbm.pt-layout weights, preset '{args.preset}')",
            "config": {"workload": f"{wname}: {wl['desc']}", "streams_per_gpu": B, "points_per_stream": N,
                       "memory_size": wl["M"], "frame": [wl["H"], wl["W"]], "frame_pipeline": mode_txt,
                       "precision": PRECISION_LABEL[prec], "parallelism": f"streams x{world} (no collective in the step)",
                       "knobs": {"debug_set": {str(k): v for k, v in knobs.items()}, "env": env_knobs, "tuning_build": tuning_lib}},
            "headline_valid": not knobs and not tuning_lib,
            "p50_latency_ms": lat,
            "p50_latency_graph_replay_ms": lat_graph,
            "step_algorithmic_TFLOPs": step_flops(wl, N, B) / (elapsed / args.steps) / 1e12,
        }
        per = lambda el: {"ms_per_step": 1e3 * el / args.steps, "value": total_pf / el, "unit": "point-frames/s"}
        line["serial_frames"] = per(el_serial)
        if el_pipe is not None:
            line["frame_pipeline"] = per(el_pipe)
            line["pipeline_bit_identical"] = same
            line["pipeline_compared"] = f"coords, offsets, vis / rho logits, flags and probabilities of all {args.steps} timed frames + both memories and masks"
        line["headline_mode"] = "frame_pipeline" if headline_pipelined else "serial_frames"
        if world > 1:
            line["backend"] = dist.get_backend()
            line["world_size"] = dist.get_world_size()
            line["ranks"] = ranks
            line["gathered_records"] = list(gathered.shape)
            line["scaling_base"] = ("per-GPU work is fixed (weak scaling): divide `value` by N x the N = 1 line's `value` (c2 on every GPU), and "
                                    "workloads.c3_per_rank.value (BASELINE config 4: 64 streams per GPU) by N x that line's workloads.c3 figure of the same headline_mode")
        # a scaling curve needs real GPUs behind every rank: a single-device / gloo rehearsal is not one
        line["scaling_measured"] = bool(world > 1 and args.backend == "nccl" and not args.single_device)
        if not line["scaling_measured"]:
            line["scaling_note"] = ("no 1/2/4/8-GPU curve has been measured on hardware in this repository yet (the driver had no 8-GPU node in "
                                    "rounds 1-2); a run with --gpus N > 1 over RCCL is one point of it")
        roof, kernels = roofline_of(prof, nprof)
        if roof is not None:
            attach_traffic(roof, prof, wname)
            line["roofline"] = roof
            line["kernels"] = kernels
        if extra:
            line["workloads"] = extra
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(wl, sd, video_cpu, pts_cpu)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
