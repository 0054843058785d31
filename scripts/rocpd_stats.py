#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max duration) from a rocprofv3 rocpd sqlite file."""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"{'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for n, k, s, a, lo, hi in rows:
        print(f"{n[:90]:90s} {k:6d} {s/1e3:12.1f} {a/1e3:10.2f} {lo/1e3:10.2f} {hi/1e3:10.2f} {100*s/tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
