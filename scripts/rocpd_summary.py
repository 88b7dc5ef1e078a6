"""Dump the per-kernel summary (the `top_kernels` view = rocprofv3 --kernel-trace --stats) of a rocpd .db to CSV."""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for name, calls, tot, avg, pct in cur:
            w.writerow([name[:160], calls, round(tot, 3), round(avg, 3), round(pct, 3)])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
