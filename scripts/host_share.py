"""Is the host the limit?  Enqueue time of n frames (python loop until the last launch is queued) against their total time.
usage: host_share.py [c2] [frames]"""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402
import bench  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
with torch.no_grad():
    r = bench.Runner(wl, 0, torch.device("cuda", 0))
    for pipe in (False, True, False, True):
        r.run(10, pipeline=pipe)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.run(n, pipeline=pipe)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"pipeline={pipe}: enqueue {1e3 * (t1 - t0) / n:.3f} ms/frame, total {1e3 * (t2 - t0) / n:.3f} ms/frame", flush=True)
