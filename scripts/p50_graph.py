"""p50 per-frame latency at C2, eager launches against HIP-graph replay (LBM_online.use_graph), each frame synchronised on its own."""
import statistics
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402
import bench  # noqa: E402

with torch.no_grad():
    r = bench.Runner(bench.WORKLOADS["c2"], 0, torch.device("cuda", 0))
    r.run(4)
    for graph in (False, True, False, True):
        r.model.use_graph = graph
        r.run(30)                      # one capture per ring position on first use
        lat = []
        for i in range(60):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            out = r.run(1, first=i)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t1) * 1e3)
        print(f"graph={graph}: p50 {statistics.median(lat):.3f} ms, p90 {sorted(lat)[54]:.3f} ms", flush=True)
