"""CPU micro-benchmark of the product's FALCON search over oracle-decoded candidate tables (no GPU)."""
import sys, os, ctypes as C, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from lsn_testlib import *
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cache = "/tmp/search_bench_v2_%d.pkl" % n
h = hosttest()
sc = scenario("cfg3", seed=3)
cce = (C.c_uint32 * 3)(20, 54, 87)
hs = h.lsnh_search_new(100, 2, 1, cce, 5, 0.99, 0)
sizes = [h.lsnh_search_size(hs, k) for k in range(h.lsnh_search_nof_sizes(hs))]
if os.path.exists(cache):
    ttis, cfis, cands, pws = pickle.load(open(cache, "rb"))
else:
    tx = TxGen(**sc); ow = OracleWorker(100, 2, 1, 2)
    ttis, cfis, cands, pws = [], [], [], []
    for i in range(n):
        tti, iq, _ = tx.next(); ow.work(iq, tti)
        cand, pw = candidate_table(ow.llr(), 87, sizes, tti % 10)
        ttis.append(tti); cfis.append(ow.cfi()); cands.append(bytes(cand)); pws.append(pw.tobytes())
    pickle.dump((ttis, cfis, cands, pws), open(cache, "wb"))
T = (C.c_uint32 * n)(*ttis); F = (C.c_uint32 * n)(*cfis)
CB = C.create_string_buffer(b"".join(cands)); PB = C.create_string_buffer(b"".join(pws))
for reps in (5, 5, 5, 5):
    print("search: %.2f us/subframe" % h.lsnh_search_bench(hs, n, reps, T, F, CB, PB), "active", h.lsnh_search_nof_active(hs))
st = (C.c_uint32 * 7)(); h.lsnh_search_stats(hs, st)
print("decoded locations/sf %.1f, subframes %d" % (st[0] / st[3], st[3]))
