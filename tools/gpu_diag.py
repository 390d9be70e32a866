"""Diagnostic driver (GPU box): per-stage parity against the oracle + timing breakdown. Not a test."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import ltesniffer_amd as la
from lsn_testlib import scenario
from parity import compare_taps, gen_subframes, gpu_records, oracle_records, run_oracle

scn = sys.argv[1] if len(sys.argv) > 1 else "small"
nsf = int(sys.argv[2]) if len(sys.argv) > 2 else 20
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16
sc = scenario(scn, seed=3)
t0 = time.time()
tti0, iq, truth = gen_subframes(sc, nsf)
t1 = time.time()
ow, per_sf, orecs = run_oracle(sc, tti0, iq)
t2 = time.time()
print("gen %.2fs oracle %.2fs (%.1f ms/sf)" % (t1 - t0, t2 - t1, 1e3 * (t2 - t1) / nsf), flush=True)
phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=batch)
assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
allbad = []
for base in range(0, nsf, batch):
    n = min(batch, nsf - base)
    t = time.time()
    phy.process_host(iq[base:base + n], tti0 + base, 0)
    dt = time.time() - t
    p = phy.perf()
    print("batch@%d n=%d wall %.1f ms | A %.1f search %.1f C %.1f commit %.1f | tb %d cb %d iters %d ondemand %d pdus %d" % (
        base, n, dt * 1e3, p.ms_stage_a, p.ms_search, p.ms_stage_c, p.ms_commit, p.nof_tb_decodes, p.nof_cb_decodes,
        p.nof_turbo_iterations, p.nof_ondemand_decodes, p.nof_pdus), flush=True)
    print("   kernels: " + " ".join("%s=%.3f" % (la.KERNELS[k], p.kernel_ms[k]) for k in range(len(la.KERNELS))), flush=True)
    bad = compare_taps(phy, per_sf, sc, base, n)
    for b in bad[:6]:
        print("   MISMATCH", base, str(b)[:300], flush=True)
    allbad += bad
g, o = gpu_records(phy), oracle_records(orecs)
print("records gpu %d oracle %d equal %s; tap mismatches %d" % (len(g), len(o), g == o, len(allbad)))
if g != o:
    for i, (a, b) in enumerate(zip(g, o)):
        if a != b:
            print("first differing record", i, a[:19].hex(), len(a), b[:19].hex(), len(b))
            break
