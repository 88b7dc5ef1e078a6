"""Run N pipelined (or serial) C2 frames and nothing else: the tail of a rocprofv3 --kernel-trace of this is steady state."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "pipe"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    wl = bench.WORKLOADS["c2"]
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(wl, dev)
    video, pts = bench.make_inputs(wl, 0)
    video, pts = video.to(dev), pts.to(dev)
    nfr = wl["frames"]
    with torch.no_grad():
        model.init_streams(pts, video[:, 0].contiguous())
        frames = [video[:, 1 + i].contiguous() for i in range(nfr)]
        for rep in range(2):
            prof = None
            if rep == 1 and os.environ.get("PIPE_CPROFILE"):
                import cProfile
                prof = cProfile.Profile()
                prof.enable()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tp = 0.0
            for i in range(n):
                if mode == "pipe" and i + 1 < n:
                    ta = time.perf_counter()
                    model.prefetch(frames[(i + 1) % nfr])
                    tp += time.perf_counter() - ta
                model.online_forward_batch(frames[i % nfr])
            t_host = time.perf_counter() - t0
            print(f"   prefetch host {1e3 * tp / n:.3f} ms/frame", flush=True)
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
            if prof is not None:
                prof.disable()
                import pstats
                pstats.Stats(prof).sort_stats("tottime").print_stats(18)
            print(f"{mode} rep {rep}: {1e3 * t / n:.3f} ms/frame (host enqueue {1e3 * t_host / n:.3f} ms/frame)", flush=True)


if __name__ == "__main__":
    main()
