#!/usr/bin/env python3
"""Per-kernel totals of every counter found in one or more rocprofv3 --pmc result databases (rocpd sqlite, ROCm 7.2).
usage: pmc_generic_summary.py <out.json> <results.db> [<results.db> ...] [--subframes N]   (N = subframes the profiled command
processed, stored as "_subframes" so that counters can be quoted per subframe)
Prints one row per (kernel, counter): launches, total, per launch; kernel durations come from the same pass (profiled
kernels run serialised, so durations are stand-alone durations, not the pipelined ones of a normal run)."""
import json
import sqlite3
import sys

out = {}
argv = list(sys.argv)
subframes = None
if "--subframes" in argv:
    i = argv.index("--subframes")
    subframes = int(argv[i + 1])
    del argv[i:i + 2]
for path in argv[2:]:
    cur = sqlite3.connect(path).cursor()
    cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
    dur = {}
    for r in cur.execute("select * from kernels"):
        d = dict(zip(cols, r))
        dur[d["dispatch_id"]] = d["end"] - d["start"]
    pc = [d[0] for d in cur.execute("select * from pmc_events limit 1").description]
    seen = {}
    for r in cur.execute("select * from pmc_events"):
        d = dict(zip(pc, r))
        k = d["name"].split("(")[0].replace("void ", "")
        e = out.setdefault(k, {})
        c = e.setdefault(d["counter_name"], {"total": 0.0, "launches": 0, "ns": 0})
        c["total"] += float(d["counter_value"])
        key = (k, d["counter_name"], d["dispatch_id"])
        if key not in seen:
            seen[key] = 1
            c["launches"] += 1
            c["ns"] += dur.get(d["dispatch_id"], 0)
if subframes:
    out["_subframes"] = subframes
json.dump(out, open(argv[1], "w"), indent=1)
print("%-24s %-26s %9s %16s %16s %12s" % ("kernel", "counter", "launches", "total", "per launch", "avg us"))
for k in sorted((k for k in out if not k.startswith("_")), key=lambda k: -max(c["ns"] for c in out[k].values())):
    for cn, c in sorted(out[k].items()):
        n = max(1, c["launches"])
        print("%-24s %-26s %9d %16.4e %16.1f %12.2f" % (k[:24], cn, c["launches"], c["total"], c["total"] / n, c["ns"] / n / 1e3))
