"""Where do the encoder chain (stream E) and the per-point chain (main stream) of pipelined frames sit in time?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    wl = bench.WORKLOADS["c2"]
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(wl, dev)
    video, pts = bench.make_inputs(wl, 0)
    video, pts = video.to(dev), pts.to(dev)
    nfr = wl["frames"]
    ev = lambda: torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        model.init_streams(pts, video[:, 0].contiguous())
        frames = [video[:, 1 + i].contiguous() for i in range(nfr)]
        for rep in range(2):
            torch.cuda.synchronize()
            base = ev(); base.record()
            rec = []
            for i in range(n):
                e = [ev() for _ in range(4)]
                if i + 1 < n:
                    if model._pipe is not None:
                        e[0].record(model._pipe["stream"])
                    model.prefetch(frames[(i + 1) % nfr])
                    e[1].record(model._pipe["stream"])
                e[2].record()
                model.online_forward_batch(frames[i % nfr])
                e[3].record()
                rec.append(e)
            torch.cuda.synchronize()
            if rep == 1:
                for i, e in enumerate(rec[5:17]):
                    t = [base.elapsed_time(x) if x.query() else float("nan") for x in e]
                    try:
                        print(f"frame {i+5}: enc(next) {t[0]:8.3f} -> {t[1]:8.3f} ({t[1]-t[0]:.3f})   points {t[2]:8.3f} -> {t[3]:8.3f} ({t[3]-t[2]:.3f})")
                    except Exception as ex:
                        print(ex)


if __name__ == "__main__":
    main()
