import sys, os
os.environ["LBM_PROF_SHAPES"]="1"
sys.path.insert(0,'.')
import torch
from lbm_amd import hip
import bench
wl=bench.WORKLOADS["c2"]
with torch.no_grad():
    r=bench.Runner(wl,0,torch.device("cuda",0))
    r.run(5)
    torch.cuda.synchronize()
    hip.profile_begin()
    r.run(10)
    prof=hip.profile_end()
tot=0
for p in prof:
    print("%-70s x%4.1f %7.2f us  = %7.1f" % (p["name"], p["launches"]/10, 1e3*p["total_ms"]/p["launches"], 1e2*p["total_ms"]))
    tot+=1e2*p["total_ms"]
print("launches/step", sum(p["launches"] for p in prof)/10, "sum us", tot)
