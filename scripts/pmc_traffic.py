"""HBM-side traffic per launch and kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; CSV output).

Units and gfx950 corrections as in MI355X_MICROARCH.md (HBM section): both counters are in KiB; FETCH_SIZE reports half
of the bytes of wide coalesced reads on gfx950 (128-B requests tallied at 64 B), so the read side is doubled; WRITE_SIZE
is exact for 16-B-per-lane stores.  Infinity-Cache hits are counted by these fabric-side counters, so `traffic` is what
left the L2, an upper bound of what reached HBM.

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    """rocprof kernel name -> the launcher's ProfScope name used by bench.py."""
    m = re.search(r"gemm_ss_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false), (true|false)(?:, (\d+))?(?:, (\d+))?>", name)
    if m:
        bm, bn, wm, wn, st, conv, full, topk, npl = m.groups()
        sfx = "h" if npl == "1" else "s"
        base = "conv_s" + sfx if conv == "true" else ("gemm_ss_topk" + topk if topk not in (None, "0") else "gemm_s" + sfx)
        return f"{base}{'_full' if full == 'true' else ''}<{bm},{bn},{wm},{wn},{st}>"
    m = re.search(r"ffn_fused_kernel<\d+, (\d+)", name)
    if m:
        return "ffn_fused_ss" if m.group(1) == "2" else "ffn_fused_sh"
    m = re.search(r"lbm::(\w+?)_kernel", name)
    return m.group(1) if m else name[:60]


def collect(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            a = acc[short(row["Kernel_Name"])]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc


def main(fetch_csv, write_csv, out):
    fe, wr = collect(fetch_csv, "FETCH_SIZE"), collect(write_csv, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fe) | set(wr)):
        nf, f = fe.get(k, [0, 0.0])
        nw, w = wr.get(k, [0, 0.0])
        rd = 2.0 * 1024.0 * f / max(nf, 1)
        wb = 1024.0 * w / max(nw, 1)
        res[k] = {"launches": max(nf, nw), "read_bytes_per_launch": rd, "write_bytes_per_launch": wb,
                  "traffic_bytes_per_launch": rd + wb}
    json.dump({"note": "FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024, per launch (gfx950 corrections, MI355X_MICROARCH.md HBM section)",
               "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:14]:
        print(f"{k:36s} n={v['launches']:5d} read {v['read_bytes_per_launch']/1e6:8.2f} MB  write {v['write_bytes_per_launch']/1e6:8.2f} MB")


if __name__ == "__main__":
    main(*sys.argv[1:4])
