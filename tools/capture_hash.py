#!/usr/bin/env python3
"""tools/capture_hash.py [nsf] - xxh3 of the first nsf subframes of the gated cfg3 capture (tools/make_cfg3_golden.py): run here and on the GPU box
to check that the synthetic transmitter renders bit-identical samples on both hosts (libm variants), which the cached oracle stream relies on."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
from make_cfg3_golden import capture_hash, cfg3_stream  # noqa: E402
from parity import gen_capture  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
sc, *_ = cfg3_stream()
t = time.time()
tti0, iq = gen_capture(sc, n)
h, parts = capture_hash(iq)
print("capture_hash nsf %d tti0 %d: %s parts %s (%.1f s, %d threads)" % (n, tti0, h, ",".join(parts), time.time() - t, len(os.sched_getaffinity(0))))
