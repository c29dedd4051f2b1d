#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max duration, share) of a rocprofv3 rocpd database -- the same table
`rocprofv3 --stats` prints, written as CSV so it can be committed under profiles/.   usage: rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                     f"from kernels group by {name} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    out.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
    for n, k, tot, avg, mn, mx in rows:
        out.write(f'"{n}",{k},{tot},{avg:.1f},{100.0 * tot / total:.4f},{mn},{mx}\n')


if __name__ == "__main__":
    main()
