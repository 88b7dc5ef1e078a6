"""Where the waves' cycles go, per kernel, from one rocprofv3 --pmc pass of SQ counters (CSV output).

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over all waves (MI355X_MICROARCH.md 'PMC slots'):
WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES.  Printed as fractions of
WAVE_CYCLES, plus VALU instructions per wave and the kernel's active cycles per launch (GRBM_GUI_ACTIVE / 8 XCDs).

usage: pmc_sq.py <counter_collection.csv> <out.json>"""
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
            if row["Counter_Name"] == "SQ_WAVE_CYCLES":
                n[k] += 1
    res = {}
    for k, c in acc.items():
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        e = {"launches": n[k]}
        for name, v in c.items():
            e[name + "_per_launch"] = v / max(n[k], 1)
            if name.startswith("SQ_WAIT") or name.startswith("SQ_ACTIVE"):
                e[name + "_frac"] = v / wc
        if c.get("SQ_WAVES"):
            e["valu_insts_per_wave"] = c.get("SQ_INSTS_VALU", 0.0) / c["SQ_WAVES"]
            e["quad_cycles_per_wave"] = wc / c["SQ_WAVES"]
        res[k] = e
    json.dump({"note": "fractions of SQ_WAVE_CYCLES (quad-cycles summed over waves), per kernel over all its launches", "kernels": res},
              open(out, "w"), indent=1)
    keys = sorted(res, key=lambda k: -res[k].get("GRBM_GUI_ACTIVE_per_launch", res[k]["SQ_WAVE_CYCLES_per_launch"]) * res[k]["launches"])[:22]
    for k in keys:
        v = res[k]
        fr = "  ".join(f"{nm[3:-5].lower()} {100 * val:5.1f}%" for nm, val in sorted(v.items()) if nm.endswith("_frac"))
        print(f"{k[:38]:38s} n={v['launches']:4d} active {v.get('GRBM_GUI_ACTIVE_per_launch', 0) / 8e3:8.1f} kcyc  valu/wave {v.get('valu_insts_per_wave', 0):8.0f}  {fr}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
