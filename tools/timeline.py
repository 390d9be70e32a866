#!/usr/bin/env python3
"""Concurrency analysis of a rocprofv3 kernel trace (rocpd sqlite): for the last --tail fraction of the trace span, how many
kernels run at once, how long each kernel class is resident, and - for the turbo kernels - the workgroup-slot occupancy implied by
grid sizes (a launch of G workgroups that lasts T keeps at most min(G, slots) slots busy).
usage: timeline.py <results.db> [--tail 0.7]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tail = float(sys.argv[sys.argv.index("--tail") + 1]) if "--tail" in sys.argv else 0.7
cur = db.cursor()
cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
rows = [dict(zip(cols, r)) for r in cur.execute("select * from kernels")]
t_lo, t_hi = min(r["start"] for r in rows), max(r["end"] for r in rows)
cut = t_hi - tail * (t_hi - t_lo)
rows = [r for r in rows if r["start"] >= cut]
ev = []
for r in rows:
    n = r["name"].split("(")[0].replace("void ", "")
    ev.append((r["start"], 1, n))
    ev.append((r["end"], -1, n))
ev.sort()
hist = {}
res = {}
active = {}
last = ev[0][0]
for t, d, n in ev:
    dt = t - last
    if dt > 0:
        c = sum(active.values())
        hist[c] = hist.get(c, 0) + dt
        for k, v in active.items():
            if v > 0:
                res[k] = res.get(k, 0) + dt
        tk = sum(v for k, v in active.items() if k.startswith("k_turbo"))
        hist[("turbo", tk)] = hist.get(("turbo", tk), 0) + dt
    active[n] = active.get(n, 0) + d
    last = t
span = ev[-1][0] - ev[0][0]
print("span %.3f ms" % (span / 1e6))
print("kernels running at once (share of the span):")
for c in sorted(k for k in hist if isinstance(k, int)):
    print("  %2d : %5.1f %%" % (c, 100.0 * hist[c] / span))
print("turbo kernels running at once:")
for c in sorted(k[1] for k in hist if not isinstance(k, int)):
    print("  %2d : %5.1f %%" % (c, 100.0 * hist[("turbo", c)] / span))
print("share of the span with at least one instance resident:")
for k, v in sorted(res.items(), key=lambda kv: -kv[1]):
    print("  %-28s %5.1f %%" % (k, 100.0 * v / span))
gx = [c for c in cols if c.lower() in ("grid_size_x", "grid_x", "grid_size")]
wx = [c for c in cols if c.lower() in ("workgroup_size_x", "workgroup_x", "workgroup_size")]
if gx and wx:
    print("turbo launches: workgroups per launch and duration")
    for name in ("k_turbo<128>", "k_turbo<64>", "k_turbo_q"):
        sel = [r for r in rows if r["name"].split("(")[0].replace("void ", "") == name]
        if not sel:
            continue
        wg = [r[gx[0]] / max(1, r[wx[0]]) for r in sel]
        du = [(r["end"] - r["start"]) / 1e3 for r in sel]
        print("  %-14s launches %5d  workgroups/launch avg %8.1f max %8.0f  duration avg %8.1f us  sum(wg*1)/sum(dur) = %.1f wg/us"
              % (name, len(sel), sum(wg) / len(wg), max(wg), sum(du) / len(du), sum(wg) / sum(du)))
