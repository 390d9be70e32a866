#!/usr/bin/env python3
"""tools/kernel_trace_json.py <results.db> [--last-ofdm N] [--subframes S] [--out file.json]
Per-kernel figures of a rocprofv3 --kernel-trace run (rocpd sqlite) that bench.py's roofline block can be re-derived from:
  calls, avg / min / max / total launch duration, and the EXCLUSIVE time of each kernel name = the union of its launch intervals (launches
  of the same kernel overlap across the engine's streams; summing durations counts that time twice), plus the union over all kernels (GPU busy).
--last-ofdm N keeps only what starts at or after the N-th last k_ofdm launch (one k_ofdm launch per pipeline chunk: the timed region of
`bench.py --steps K` is the last K * step_sf / batch chunks); --subframes = subframes that region processed (per-subframe columns)."""
import json
import sqlite3
import sys


def arg(name, default=None, cast=str):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def union(iv):
    iv.sort()
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
rows = [dict(zip(cols, r)) for r in cur.execute("select * from kernels")]
for r in rows:
    r["k"] = r["name"].split("(")[0].replace("void ", "")
last_ofdm = arg("--last-ofdm", 0, int)
if last_ofdm:
    of = sorted(r["start"] for r in rows if r["k"] == "k_ofdm")
    if len(of) >= last_ofdm:
        cut = of[-last_ofdm]
        rows = [r for r in rows if r["start"] >= cut]
sub = arg("--subframes", 0, int)
out = {}
names = sorted(set(r["k"] for r in rows))
for n in names:
    sel = [r for r in rows if r["k"] == n]
    du = [(r["end"] - r["start"]) / 1e6 for r in sel]
    ex = union([(r["start"], r["end"]) for r in sel]) / 1e6
    out[n] = {"calls": len(sel), "avg_ms": round(sum(du) / len(du), 5), "min_ms": round(min(du), 5), "max_ms": round(max(du), 5), "total_ms": round(sum(du), 3),
              "exclusive_ms": round(ex, 3)}
    if sub:
        out[n]["exclusive_ms_per_subframe"] = round(ex / sub, 7)
        out[n]["total_ms_per_6400_subframes"] = round(sum(du) * 6400.0 / sub, 3)
t_lo, t_hi = min(r["start"] for r in rows), max(r["end"] for r in rows)
out["_wall_ms"] = round((t_hi - t_lo) / 1e6, 3)
out["_busy_ms"] = round(union([(r["start"], r["end"]) for r in rows]) / 1e6, 3)
out["_subframes"] = sub
out["_kernels_sum_ms"] = round(sum(v["total_ms"] for k, v in out.items() if not k.startswith("_")), 3)
if sub:
    out["_subframes_per_s_over_trace_span"] = round(sub / ((t_hi - t_lo) / 1e9), 1)
path = arg("--out")
if path:
    json.dump(out, open(path, "w"), indent=1)
print("%-28s %7s %10s %10s %10s %12s %12s" % ("kernel", "calls", "avg ms", "min ms", "max ms", "total ms", "exclusive ms"))
for n in sorted(names, key=lambda k: -out[k]["exclusive_ms"]):
    v = out[n]
    print("%-28s %7d %10.4f %10.4f %10.4f %12.3f %12.3f" % (n, v["calls"], v["avg_ms"], v["min_ms"], v["max_ms"], v["total_ms"], v["exclusive_ms"]))
print("trace span %.3f ms, GPU busy (union of all kernels) %.3f ms = %.1f %%, sum of durations %.3f ms%s" % (
    out["_wall_ms"], out["_busy_ms"], 100.0 * out["_busy_ms"] / out["_wall_ms"], out["_kernels_sum_ms"],
    (", %d subframes -> %.0f subframes/s over the span" % (sub, out["_subframes_per_s_over_trace_span"])) if sub else ""))
