#!/usr/bin/env python3
"""Per-kernel VALU / LDS instruction rates from one rocprofv3 pass `--pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --kernel-trace`
(separate from the HBM passes).  SQ_INSTS_* count wave-level instructions.  The peak they are priced against is the one
measured on this chip by tools/ubench/valu_rate (profiles/*_valu_ubench.txt): 740 G wave-instructions/s for plain
32-bit integer VOP2 ops at >= 2 waves per SIMD (620 G with one wave per SIMD; VOP3 / packed-16 forms ~460 G).
usage: pmc_valu_summary.py <results.db> <out.json> [subframes processed by the profiled command]"""
import json, sqlite3, sys

PEAK_G = 740.0
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
dur = {}
cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
for r in cur.execute("select * from kernels"):
    d = dict(zip(cols, r))
    dur[d["dispatch_id"]] = (d["name"].split("(")[0].replace("void ", ""), d["end"] - d["start"])
acc = {}
pc = [d[0] for d in cur.execute("select * from pmc_events limit 1").description]
for r in cur.execute("select * from pmc_events"):
    d = dict(zip(pc, r))
    k = d["name"].split("(")[0].replace("void ", "")
    a = acc.setdefault(k, {"disp": set(), "SQ_INSTS_VALU": 0.0, "SQ_INSTS_LDS": 0.0, "SQ_WAVES": 0.0, "ns": 0})
    if d["counter_name"] in a:
        a[d["counter_name"]] += float(d["counter_value"])
    if d["dispatch_id"] not in a["disp"]:
        a["disp"].add(d["dispatch_id"])
        a["ns"] += dur.get(d["dispatch_id"], (k, 0))[1]
out = {}
print("%-22s %8s %14s %12s %12s %10s %9s" % ("kernel", "launches", "VALU inst/launch", "LDS inst/l.", "waves/l.", "G VALU/s", "of peak"))
for k, a in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"]):
    n = max(1, len(a["disp"]))
    g = a["SQ_INSTS_VALU"] / max(1, a["ns"])  # wave-instructions per ns = G/s, while the kernel is resident (it shares the chip)
    out[k] = {"dispatches": n, "valu_insts_per_launch": a["SQ_INSTS_VALU"] / n, "lds_insts_per_launch": a["SQ_INSTS_LDS"] / n,
              "waves_per_launch": a["SQ_WAVES"] / n, "avg_launch_ns": a["ns"] / n, "valu_G_per_s": g, "valu_frac_of_peak": g / PEAK_G,
              "valu_insts_total": a["SQ_INSTS_VALU"]}
    print("%-22s %8d %14.0f %12.0f %12.0f %10.1f %8.1f%%" % (k[:22], n, a["SQ_INSTS_VALU"] / n, a["SQ_INSTS_LDS"] / n, a["SQ_WAVES"] / n, g, 100 * g / PEAK_G))
tot = sum(a["SQ_INSTS_VALU"] for a in acc.values())
print("total VALU wave-instructions: %.3e" % tot)
out["_total_valu_insts"] = tot
out["_peak_G_wave_insts_per_s"] = PEAK_G
out["_subframes"] = int(sys.argv[3]) if len(sys.argv) > 3 else 3 * 6400  # subframes the profiled command processed (bench.py --steps 2 --warmup 1)
json.dump(out, open(sys.argv[2], "w"), indent=1)
