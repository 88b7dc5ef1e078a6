"""Step time of a workload under values of one debug knob (serial frames, no profiler).  usage: knob_bench.py c2 KEY v1,v2,... [frames]"""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402
from lbm_amd import hip  # noqa: E402
import bench  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1]]
key = int(sys.argv[2])
vals = [int(v) for v in sys.argv[3].split(",")]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 40
lib = hip.load()
with torch.no_grad():
    r = bench.Runner(wl, 0, torch.device("cuda", 0))
    snap = r.snapshot()
    for rep in range(2):
        for v in vals:
            lib.lbm_hip_debug_set(key, v)
            r.restore(snap)
            r.run(4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r.run(n)
            torch.cuda.synchronize()
            print(f"{sys.argv[1]} key {key} = {v}: {1e3 * (time.perf_counter() - t0) / n:.3f} ms/step", flush=True)
