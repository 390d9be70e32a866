"""Two-second GPU check of the p-a feedback loop (RRCConnectionSetup -> UE configuration -> PDSCH power offset): record stream and
learned configuration against the oracle, one 24-subframe chunk (plan-time p-a stale, decoded again at commit) and chunks of 4."""
import os, sys, time
t0 = time.time()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import ltesniffer_amd as la
from lsn_testlib import scenario
from parity import gen_subframes, gpu_records, oracle_records, run_oracle
os.makedirs("gpurun_out", exist_ok=True)
log = open("gpurun_out/pa.log", "w")
def P(*a):
    print(*a, file=log, flush=True); print(*a, flush=True)
sc = scenario("cfg2", seed=3, nof_prb=25, n_rnti=3, dl_min=3, dl_max=3, ul_min=0, ul_max=0, mcs_min=20, mcs_max=28, msg4_period=6, snr_db=36.0, msg4_p_a_idx=0)
tti0, iq, truth = gen_subframes(sc, 24)
ow, _, orecs = run_oracle(sc, tti0, iq, taps=False)
o = oracle_records(orecs)
P("oracle", len(o), round(time.time() - t0, 2))
phy = la.Phy(nof_rx_antennas=2, max_batch=24)
phy.setCell(25, 2, 1)
P("phy", round(time.time() - t0, 2))
phy.process_host(iq, tti0, 0)
g = gpu_records(phy)
P("gpu one chunk", len(g), g == o, round(time.time() - t0, 2))
phy2 = la.Phy(nof_rx_antennas=2, max_batch=4)
phy2.setCell(25, 2, 1)
for b in range(0, 24, 4):
    phy2.process_host(iq[b:b + 4], tti0 + b, 0)
g2 = gpu_records(phy2)
P("gpu chunks of 4", len(g2), g2 == o, round(time.time() - t0, 2))
P([(hex(r), phy.ue_config(r).has_ue_config, phy.ue_config(r).p_a_db, ow.ue_cfg(r)[:2]) for r in sorted({p["rnti"] for pd in truth for p in pd})])
