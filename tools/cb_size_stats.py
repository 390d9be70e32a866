#!/usr/bin/env python3
"""Code-block sizes of the metric's workload, from the ORACLE's decode sequence (test infrastructure, CPU only): how many turbo code blocks have at most
64 trellis windows (one working wavefront in k_turbo) and how much of the decoder's slot time (iterations x steps per window) they hold.
usage: cb_size_stats.py [subframes=60]   -> profiles/r05_cb_size_stats.txt is its output"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import ltesniffer_amd as la  # noqa: E402
from lsn_testlib import oracle_trace, scenario  # noqa: E402
from parity import gen_subframes, run_oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sc = scenario("cfg3", seed=3)
tti0, iq, _ = gen_subframes(sc, n)
run_oracle(sc, tti0, iq, update_meta_period=20, trace=True, taps=False)
cnt, slot, wave = collections.Counter(), collections.Counter(), collections.Counter()
ks = collections.Counter()
for o in oracle_trace():
    if o["is_ul"]:
        continue
    for c in o["cbs"]:
        K = c["K"]
        P = la.turbo_nwin(K)
        W, it = K // P, max(1, c["iters"])
        cls = "more than 64 windows (two wavefronts)" if P > 64 else ("at most 64 windows, K <= 2752 (pairable)" if K <= 2752 else "at most 64 windows, K > 2752 (alone)")
        cnt[cls] += 1
        slot[cls] += it * W
        wave[cls] += it * W * (2 if P > 64 else 1)
        ks[(K // 512) * 512] += 1
tot_s = sum(slot.values())
print("cfg3, seed 3, %d subframes, the oracle's decode sequence (every code block of every decode call)" % n)
for k in cnt:
    print("%-48s blocks %5d   slot time (iterations x steps) %7d = %4.1f %%   wavefront-steps %7d" % (k, cnt[k], slot[k], 100.0 * slot[k] / tot_s, wave[k]))
print("K histogram (bucket of 512):", sorted(ks.items()))
