#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace: per-kernel calls / total / average / min / max (us).
usage: rocpd_summary.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
rows = cur.execute("select * from kernels").fetchall()
ni, si, ei = cols.index("name"), cols.index("start"), cols.index("end")
by = {}
for r in rows:
    by.setdefault(r[ni].split("(")[0], []).append((r[si], r[ei]))
tot = 0
out = []
for k, v in by.items():
    d = [(e - s) / 1e3 for s, e in sorted(v)]
    out.append((sum(d), k, len(d), d))
    tot += sum(d)
print("%-28s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
for s, k, n, d in sorted(out, reverse=True):
    print("%-28s %8d %12.1f %10.2f %10.2f %10.2f %6.2f%%" % (k, n, s, s / n, min(d), max(d), 100 * s / tot))
