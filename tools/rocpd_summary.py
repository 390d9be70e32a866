#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace: per-kernel calls / total / average / min / max (us),
plus the GPU busy time (union of all kernel intervals) of the trace.
usage: rocpd_summary.py <results.db> [--tail FRACTION]   (--tail 0.7: only dispatches in the last 70 % of the trace span,
i.e. skip process start-up / warm-up)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tail = float(sys.argv[sys.argv.index("--tail") + 1]) if "--tail" in sys.argv else 1.0
cur = db.cursor()
cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
rows = cur.execute("select * from kernels").fetchall()
ni, si, ei = cols.index("name"), cols.index("start"), cols.index("end")
t_lo, t_hi = min(r[si] for r in rows), max(r[ei] for r in rows)
cut = t_hi - tail * (t_hi - t_lo)
rows = [r for r in rows if r[si] >= cut]
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
iv = sorted((r[si], r[ei]) for r in rows)
busy, cs, ce = 0, None, None
for s, e in iv:
    if cs is None:
        cs, ce = s, e
    elif s <= ce:
        ce = max(ce, e)
    else:
        busy += ce - cs
        cs, ce = s, e
if cs is not None:
    busy += ce - cs
span = (max(e for _, e in iv) - min(s for s, _ in iv)) if iv else 0
print("\nspan of the selected dispatches: %.3f ms; GPU busy (union of kernel intervals): %.3f ms = %.1f %%; sum of kernel durations: %.3f ms"
      % (span / 1e6, busy / 1e6, 100.0 * busy / max(1, span), tot / 1e3))
