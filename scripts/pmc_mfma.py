"""MFMA utilisation per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; CSV output).

SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over every SIMD (32 per v_mfma_f32_32x32x16_f16, MI355X_MICROARCH.md
'PMC units'); GRBM_GUI_ACTIVE is the dispatch's active cycles summed over the 8 XCDs.  Utilisation of the chip's 1024 matrix pipes
(256 CUs x 4 SIMDs) = busy / (1024 x GUI_ACTIVE / 8).  The split-precision kernels issue three MFMAs per algorithmic block, so this is
the fraction of the f16 MFMA peak AT THE CLOCK THE KERNEL RAN AT (DVFS lowers it under matrix load), not of 2.5 PFLOP/s.

usage: pmc_mfma.py <counter_collection.csv> <out.json>"""
import csv
import json
import sys
from collections import defaultdict

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_traffic import short


def main(path, out):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                n[k] += 1
    res = {}
    for k, c in acc.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if gui <= 0:
            continue
        res[k] = {"launches": n[k], "mfma_busy_cycles_per_launch": busy / max(n[k], 1), "active_cycles_per_launch": gui / 8 / max(n[k], 1),
                  "mfma_util": busy / (1024.0 * gui / 8.0)}
    json.dump({"note": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 matrix pipes x GRBM_GUI_ACTIVE / 8 XCDs), per kernel over all its launches",
               "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["active_cycles_per_launch"] * kv[1]["launches"])[:16]:
        print(f"{k:40s} n={v['launches']:5d}  active {v['active_cycles_per_launch'] / 1e3:9.1f} kcycles/launch  MFMA util {100 * v['mfma_util']:5.1f} %")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
