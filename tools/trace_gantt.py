#!/usr/bin/env python3
"""Summarise an LSN_TRACE pipeline event log (thread event chunk ms): per stage, mean latency per chunk and the share of the
traced span each thread spends in it.
usage: trace_gantt.py <trace.txt> [--skip-ms 200]"""
import sys
from collections import defaultdict

skip = float(sys.argv[sys.argv.index("--skip-ms") + 1]) if "--skip-ms" in sys.argv else 0.0
ev = []
for line in open(sys.argv[1]):
    thr, name, chunk, t = line.split()
    ev.append((float(t), int(thr), name, int(chunk)))
ev.sort()
t_end = ev[-1][0]
if skip >= t_end - ev[0][0]:
    skip = 0.0
ev = [e for e in ev if e[0] >= skip]
span = t_end - ev[0][0]
pairs = [("front: wait slot", "acq_begin", "acq_end"), ("front: spec RAR", "stage_a_done", "spec_done"), ("search", "search_begin", "search_end"),
         ("decode: plan (host)", "dec_begin", "w1_launched"), ("decode: wave 1", "w1_launched", "w1_done"), ("decode: wave 2", "w1_done", "w2_done"),
         ("decode: wait turn", "w2_done", "commit_begin"), ("decode: commit", "commit_begin", "commit_end"),
         ("chunk: A launched -> A done", "acq_end", "stage_a_done")]
last = {}
acc = defaultdict(lambda: [0.0, 0])
per_thr = defaultdict(lambda: defaultdict(float))
for t, thr, name, chunk in ev:
    for label, b, e in pairs:
        if name == e and (thr, b, chunk) in last:
            d = t - last.pop((thr, b, chunk))
            acc[label][0] += d
            acc[label][1] += 1
            per_thr[thr][label] += d
    last[(thr, name, chunk)] = t
print("traced span %.1f ms" % span)
print("%-30s %8s %10s %12s" % ("interval", "count", "mean ms", "sum/span"))
for label, b, e in pairs:
    if label in acc:
        s, n = acc[label]
        print("%-30s %8d %10.3f %12.3f" % (label, n, s / n, s / span))
print("per thread share of the span (thread 0 = search/caller, 1 = front, 2.. = decode):")
for thr in sorted(per_thr):
    print("  thr %2d: " % thr + ", ".join("%s %.2f" % (k.split(": ")[-1], v / span) for k, v in per_thr[thr].items()))
