"""Host-side enqueue cost of one step (no synchronisation inside the timed region) vs GPU time (GPU only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

wl = bench.WORKLOADS[os.environ.get("WL", "c2")]
dev = torch.device("cuda", 0)
model, sd = bench.build_model(wl, dev)
video, pts = bench.make_inputs(wl, 0)
video, pts = video.to(dev), pts.to(dev)
with torch.no_grad():
    model.init_streams(pts, video[:, 0].contiguous())
    frames = [video[:, 1 + i].contiguous() for i in range(wl["frames"])]
    tok = None
    for i in range(10):
        _, _, _, tok = model.online_forward_batch(frames[i % len(frames)], tokens=tok)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for i in range(n):
        model.online_forward_batch(frames[i % len(frames)], tokens=tok)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue per step: %.3f ms; total per step incl. GPU drain: %.3f ms" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
    # encoder alone / tracker alone (host)
    t0 = time.perf_counter()
    for i in range(n):
        model.backbone.encode_tokens(frames[i % len(frames)], tok)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("encoder: host %.3f ms, with drain %.3f ms" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
