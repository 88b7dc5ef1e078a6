"""Every launch of one step of a workload, by problem shape (in-library HIP events): name, launches per step, average us, us per
step, algorithmic TFLOP/s and GB/s.   usage: c2_launches.py [c2|c3|c5] [steps]"""
import os
import sys

os.environ["LBM_PROF_SHAPES"] = "1"
sys.path.insert(0, ".")
import torch  # noqa: E402
from lbm_amd import hip  # noqa: E402
import bench  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
with torch.no_grad():
    r = bench.Runner(wl, 0, torch.device("cuda", 0))
    r.run(3)
    torch.cuda.synchronize()
    hip.profile_begin()
    r.run(n)
    prof = hip.profile_end()
tot = 0
for p in sorted(prof, key=lambda p: -p["total_ms"]):
    us = 1e3 * p["total_ms"] / n
    tf = p["flops"] / (p["total_ms"] / 1e3) / 1e12 if p.get("flops") else 0.0
    gb = p["bytes"] / (p["total_ms"] / 1e3) / 1e9 if p.get("bytes") else 0.0
    print("%-72s x%5.1f %9.2f us  = %9.1f  %6.1f TF %7.0f GB/s" % (p["name"], p["launches"] / n, 1e3 * p["total_ms"] / p["launches"], us, tf, gb))
    tot += us
print("launches/step", sum(p["launches"] for p in prof) / n, "sum us", tot)
