"""One step's launches in order from a rocprofv3 --kernel-trace CSV: name, grid, duration, gap to the previous kernel.
usage: step_trace.py <kernel_trace.csv> [marker kernel substring = preprocess_px]"""
import csv
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_traffic import short

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "preprocess_px"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
prev_end = None
tot = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    prev_end = max(e, prev_end or e)
    tot += (e - s) / 1e3
    print(f"{short(r['Kernel_Name'])[:44]:44s} grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):7d} x{r['Grid_Size_Y']:>4s} x{r['Grid_Size_Z']:>3s}  wg {r['Workgroup_Size_X']:>4s}  {(e - s) / 1e3:9.1f} us  gap {gap:6.1f}")
print(f"sum of kernel time {tot / 1e3:.3f} ms, wall {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e6:.3f} ms, {b - a} launches")
