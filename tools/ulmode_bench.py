#!/usr/bin/env python3
"""UL_MODE throughput (BASELINE configs[3]-like: 20 MHz, antenna 0 = downlink with 64 RNTIs, antenna 1 = uplink PUSCH at n+4).
usage: ulmode_bench.py [gen=400] [reps=8]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import ltesniffer_amd as la
from lsn_testlib import scenario, gen_ul_mode_subframes

gen = int(sys.argv[1]) if len(sys.argv) > 1 else 400
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc = scenario("cfg2", seed=4, nof_rx=1, n_rnti=64, ul_min=2, ul_max=4, mcs_min=0, mcs_max=28, snr_db=28.0)
t0 = time.perf_counter()
tti0, iq, sent = gen_ul_mode_subframes(sc, gen, ul_snr_db=22.0)
print("generated %d UL_MODE subframes (%d PUSCH) in %.1f s" % (gen, len(sent), time.perf_counter() - t0), flush=True)
big = np.ascontiguousarray(np.tile(iq, (reps, 1, 1)))
phy = la.Phy(nof_rx_antennas=2, sniffer_mode=1, max_batch=200, pcapwriter=la.PcapWriter(None))
assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"]) and phy.setUlConfig(3, 5)
for r in range(3):
    phy.pcapwriter.reset()
    t0 = time.perf_counter()
    phy.process_host(big, tti0, 500)
    dt = time.perf_counter() - t0
    p = phy.perf()
    print("run %d: %d subframes in %.3f s = %.0f subframes/s; records %d; search %.0f ms stageC %.0f ms commit %.0f ms" % (
        r, len(big), dt, len(big) / dt, phy.pcapwriter.nof_records(), p.ms_search, p.ms_stage_c, p.ms_commit), flush=True)
phy.close()
