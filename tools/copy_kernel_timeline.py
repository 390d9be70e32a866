#!/usr/bin/env python3
"""tools/copy_kernel_timeline.py <results.db> [--min-copy-bytes N] [--kernels k_ofdm,k_viterbi,...] [--max-lines N]
Merged time line of memory copies and selected kernels of a rocprofv3 --kernel-trace --memory-copy-trace run (rocpd sqlite), times in ms
relative to the first large host -> device copy: shows whether the copies of the next blocks, stage A and the device -> host mirrors overlap."""
import sqlite3
import sys


def arg(name, default=None, cast=str):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
minb = arg("--min-copy-bytes", 1 << 20, int)
ks = arg("--kernels", "k_ofdm,k_viterbi,k_turbo<128>,k_turbo<64>,k_pdsch_demod,k_rm").split(",")
maxl = arg("--max-lines", 400, int)
ev = []
cols = [d[0] for d in cur.execute("select * from memory_copies limit 1").description]
for r in cur.execute("select * from memory_copies"):
    r = dict(zip(cols, r))
    if r["size"] >= minb:
        ev.append((r["start"], r["end"], "%s %.1f MB" % (r["name"].replace("MEMORY_COPY_", ""), r["size"] / 1e6), r.get("stream_id")))
cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
for r in cur.execute("select * from kernels"):
    r = dict(zip(cols, r))
    n = r["name"].split("(")[0].replace("void ", "")
    if n in ks:
        ev.append((r["start"], r["end"], n, r.get("stream_id")))
ev.sort()
big = [e for e in ev if "HOST_TO_DEVICE" in e[2] and float(e[2].split()[1]) > 100]
t0 = big[0][0] if big else ev[0][0]
n = 0
for s, e, name, st in ev:
    if s < t0:
        continue
    print("%10.3f %10.3f %8.3f ms  stream %-4s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, st, name))
    n += 1
    if n >= maxl:
        break
