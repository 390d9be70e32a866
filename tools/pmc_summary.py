#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as the
MI355X guide prescribes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2).  Counter values are KiB per dispatch.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read
(requests tallied at 64 B instead of 128 B) -> 'fetch_corrected' doubles it; our narrow (2-byte-per-lane) gathers are
uncalibrated, so both the raw and the doubled figure are kept.  WRITE_SIZE is taken as reported.
usage: pmc_summary.py <fetch.db> <write.db> <out.json>"""
import json, sqlite3, sys


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    out = {}
    for name, val in cur.execute("select name, counter_value from pmc_events where counter_name = ?", (counter,)):
        k = name.split("(")[0].replace("void ", "")
        out.setdefault(k, []).append(float(val) * 1024.0)
    return out


f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
res = {}
for k in sorted(set(f) | set(w)):
    fv, wv = f.get(k, []), w.get(k, [])
    res[k] = {"dispatches": len(fv), "fetch_raw_bytes_per_launch": sum(fv) / max(1, len(fv)),
              "fetch_corrected_bytes_per_launch": 2.0 * sum(fv) / max(1, len(fv)),
              "write_bytes_per_launch": sum(wv) / max(1, len(wv)),
              "fetch_raw_total": sum(fv), "write_total": sum(wv)}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print("%-24s %8s %16s %16s %16s" % ("kernel", "launches", "fetch raw B/launch", "fetch x2 B/launch", "write B/launch"))
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["fetch_raw_total"] - kv[1]["write_total"]):
    print("%-24s %8d %16.0f %16.0f %16.0f" % (k, v["dispatches"], v["fetch_raw_bytes_per_launch"], v["fetch_corrected_bytes_per_launch"], v["write_bytes_per_launch"]))
