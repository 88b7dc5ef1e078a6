"""Two independent trackers on two HIP streams of ONE process (no cross-stream dependency at all): does the GPU overlap them?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

wl = bench.WORKLOADS["c2"]
dev = torch.device("cuda", 0)
models = []
for i in range(2):
    m, _ = bench.build_model(wl, dev)
    models.append(m)
video, pts = bench.make_inputs(wl, 0)
video, pts = video.to(dev), pts.to(dev)
nfr = wl["frames"]
frames = [video[:, 1 + i].contiguous() for i in range(nfr)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
with torch.no_grad():
    for m, s in zip(models, streams):
        with torch.cuda.stream(s):
            m.init_streams(pts, video[:, 0].contiguous())
    torch.cuda.synchronize()
    for mode in ("one stream, one model", "two streams, two models", "one stream, two models"):
        n = 40
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(n):
                if mode == "one stream, one model":
                    with torch.cuda.stream(streams[0]):
                        models[0].online_forward_batch(frames[t % nfr])
                elif mode == "two streams, two models":
                    for m, s in zip(models, streams):
                        with torch.cuda.stream(s):
                            m.online_forward_batch(frames[t % nfr])
                else:
                    for m in models:
                        with torch.cuda.stream(streams[0]):
                            m.online_forward_batch(frames[t % nfr])
            th = time.perf_counter() - t0
            torch.cuda.synchronize()
            tt = time.perf_counter() - t0
        k = 1 if mode == "one stream, one model" else 2
        print(f"{mode:28s}: {1e3 * tt / n:.3f} ms per round of {k} frame(s) (host {1e3 * th / n:.3f}) -> {1e3 * tt / n / k:.3f} ms per frame", flush=True)
