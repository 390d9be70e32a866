#!/usr/bin/env python3
"""Word length the max-log-MAP recursions of the turbo decoder need: largest |alpha|, |beta|, |beta + gamma|, |alpha + beta + gamma| and |gamma|
seen by the oracle's decoder (per-step normalised to state 0, the quantities a 16-bit-wrapped variant of k_turbo would compare with v_max_i16)
(a) on the cfg3 workload over an SNR sweep, (b) on adversarial code blocks (saturated and random soft values, all 188 block sizes sampled).
Builds oracle/_build/liblsn_oracle_stats.so (-DO_TURBO_STATS) - a separate library, the test / bench oracle is not instrumented.
usage: turbo_metric_ranges.py [subframes per SNR point = 60]  ->  profiles/r02_turbo_metric_ranges.txt"""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import lsn_testlib as T

so = os.path.join(ROOT, "oracle", "_build", "liblsn_oracle_stats.so")
subprocess.check_call("mkdir -p _build && gcc -std=gnu11 -O2 -fPIC -shared -ffp-contract=off -fno-fast-math -w -DO_TURBO_STATS -o _build/liblsn_oracle_stats.so o_*.c -lm",
                      shell=True, cwd=os.path.join(ROOT, "oracle"))
T.ORACLE_SO = so
lib = T.oracle()
stat = (C.c_longlong * 5).in_dll(lib, "o_turbo_stat_max")
names = ["|alpha|", "|beta|", "|beta+gamma|", "|alpha+beta+gamma|", "|gamma|"]


def take():
    v = [int(x) for x in stat]
    for i in range(5):
        stat[i] = 0
    return v


n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
out = ["# " + __doc__.strip().split("\n")[0], "# columns: " + "  ".join(names) + "   (int16 limit 32767; LSN_NEG_METRIC = -12000, |gamma| <= 3069 by construction)"]
t0 = time.time()
for snr in (6.0, 8.0, 10.0, 14.0, 20.0, 30.0):
    sc = T.scenario("cfg3", seed=91, snr_db=snr)
    tx = T.TxGen(**sc)
    ow = T.OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"], sc["phich_ng_x6"])
    take()
    for i in range(n):
        tti, iq, _ = tx.next()
        ow.work(iq, tti, update_meta=1 if i % 20 == 0 else 0)
    out.append("cfg3 %5.1f dB, %3d subframes: " % (snr, n) + "  ".join("%6d" % v for v in take()))
    print(out[-1], flush=True)
# adversarial code blocks straight into o_turbo_decode_cb: d3 = [3][K + 4] int16 soft values (sys, par1, par2 streams incl. tail)
lib.o_turbo_decode_cb.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.POINTER(C.c_int)]
rng = np.random.RandomState(7)
ks = [40, 48, 512, 1024, 2048, 3072, 4096, 5056, 6144]
for name, gen in (("all +511", lambda m: np.full(m, 511)), ("all -511", lambda m: np.full(m, -511)), ("random +-511", lambda m: rng.choice([-511, 511], size=m)),
                  ("uniform", lambda m: rng.randint(-511, 512, size=m)), ("sys +511 / parity -511", None), ("alternating blocks", None)):
    take()
    for K in ks:
        m = 3 * (K + 4)
        if name == "sys +511 / parity -511":
            d = np.concatenate([np.full(K + 4, 511), np.full(2 * (K + 4), -511)])
        elif name == "alternating blocks":
            d = np.where((np.arange(m) // 37) % 2 == 0, 511, -511)
        else:
            d = gen(m)
        d3 = np.ascontiguousarray(d, dtype=np.int16)
        bits = np.zeros(K, dtype=np.uint8)
        ok = C.c_int(0)
        lib.o_turbo_decode_cb(d3.ctypes.data, K, 12, 0x1864CFB, bits.ctypes.data, C.byref(ok))
    out.append("adversarial %-24s (K = %s): " % (name, ",".join(map(str, ks))) + "  ".join("%6d" % v for v in take()))
    print(out[-1], flush=True)
out.append("# %.0f s" % (time.time() - t0))
open(os.path.join(ROOT, "profiles", "r02_turbo_metric_ranges.txt"), "w").write("\n".join(out) + "\n")
