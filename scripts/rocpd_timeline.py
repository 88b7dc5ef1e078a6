"""Timeline facts from a rocpd .db (rocprofv3 --kernel-trace): busy time, overlap between queues, gaps, and a dump of a window."""
import sqlite3
import sys


def main(db, dump_from=0.5, dump_n=80):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    print("columns:", cols)
    qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
    scol = "stream_id" if "stream_id" in cols else None
    sel = "name, start, end" + (f", {qcol}" if qcol else ", 0") + (f", {scol}" if scol else ", 0")
    rows = list(c.execute(f"select {sel} from kernels order by start"))
    n = len(rows)
    print("kernels:", n)
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    # restrict to the second half (steady state)
    mid = t0 + (t1 - t0) * dump_from
    rs = [r for r in rows if r[1] >= mid]
    span = max(r[2] for r in rs) - rs[0][1]
    summed = sum(r[2] - r[1] for r in rs)
    # union of intervals
    busy, cur_s, cur_e = 0, None, None
    for r in rs:
        if cur_e is None or r[1] > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = r[1], r[2]
        else:
            cur_e = max(cur_e, r[2])
    busy += cur_e - cur_s
    print(f"window span {span/1e3:.1f} us, sum of kernel durations {summed/1e3:.1f} us, union busy {busy/1e3:.1f} us, "
          f"overlap {(summed-busy)/1e3:.1f} us, idle {(span-busy)/1e3:.1f} us")
    qs = {}
    for r in rs:
        qs.setdefault((r[3], r[4]), [0, 0])
        qs[(r[3], r[4])][0] += 1
        qs[(r[3], r[4])][1] += r[2] - r[1]
    for k, v in qs.items():
        print("queue/stream", k, "kernels", v[0], "sum us", round(v[1] / 1e3, 1))
    base = rs[0][1]
    for r in rs[:dump_n]:
        print(f"{(r[1]-base)/1e3:9.1f} {(r[2]-base)/1e3:9.1f} {(r[2]-r[1])/1e3:7.1f} q{r[3]} s{r[4]} {r[0][:60]}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5, int(sys.argv[3]) if len(sys.argv) > 3 else 80)


def gaps(db, frac=0.5):
    """Per queue: kernel time, and the gaps between consecutive kernels of that queue (start - previous end)."""
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else "0"
    scol = "stream_id" if "stream_id" in cols else "0"
    rows = list(c.execute(f"select name, start, end, {qcol}, {scol} from kernels order by start"))
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    rs = [r for r in rows if r[1] >= t0 + (t1 - t0) * frac]
    span = max(r[2] for r in rs) - rs[0][1]
    print(f"window {span / 1e3:.0f} us, {len(rs)} kernels")
    by = {}
    for r in rs:
        by.setdefault((r[3], r[4]), []).append(r)
    for k, v in by.items():
        dur = sum(r[2] - r[1] for r in v)
        g = sorted((v[i + 1][1] - v[i][2]) / 1e3 for i in range(len(v) - 1))
        small = [x for x in g if x < 30]
        print(f"queue {k}: {len(v)} kernels, busy {dur / 1e3:.0f} us ({100 * dur / span:.0f} % of window), avg kernel {dur / len(v) / 1e3:.2f} us; "
              f"gaps: median {g[len(g) // 2]:.2f} us, mean of gaps < 30 us {sum(small) / max(len(small), 1):.2f} us, sum of gaps < 30 us {sum(small):.0f} us, "
              f"gaps >= 30 us: {len(g) - len(small)} totalling {sum(x for x in g if x >= 30):.0f} us")


if __name__ == "__main__" and len(sys.argv) > 4 and sys.argv[4] == "gaps":
    gaps(sys.argv[1], float(sys.argv[2]))
