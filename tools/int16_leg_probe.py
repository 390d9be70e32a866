#!/usr/bin/env python3
"""tools/int16_leg_probe.py - chunk size of the host-buffer path on int16 samples (lsn_phy_process_host_int): on cf32 the leg is the link, on int16
it is the link plus the pipeline's latency behind the last copy, which grows with the chunk.  Ungated rate probe: the headline capture in pinned
memory, three consecutive passes per chunk size (cold, warm, warm).   usage: int16_leg_probe.py [nsf=20000] [batch ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import ltesniffer_amd as la  # noqa: E402
from make_cfg3_golden import cfg3_stream, sc16_capture  # noqa: E402
from parity import gen_capture  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
batches = [int(x) for x in sys.argv[2:]] or [200, 300, 400, 600, 800]
sc, *_ = cfg3_stream()
tti0, iq = gen_capture(sc, n)
q, lsb = sc16_capture(iq)
hq = torch.from_numpy(q).pin_memory()
hc = torch.from_numpy(iq).pin_memory()
del q, iq
for rnd in (1, 2):
    for b in batches:
        for name, run in (("int16", lambda ph, t: ph.process_host_int(hq.numpy(), t, 500, sample_scale=lsb)), ("cf32", lambda ph, t: ph.process_host(hc.numpy(), t, 500))):
            w = la.PcapWriter(None)
            w.set_store(False)
            phy = la.Phy(nof_rx_antennas=2, max_batch=b, pcapwriter=w)
            phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
            r = []
            for k in range(3):
                t0 = time.perf_counter()
                run(phy, (tti0 + k * n) % 10240)
                r.append(n / (time.perf_counter() - t0))
            phy.close()
            print("round %d  chunks of %4d  %-5s  %7.0f %7.0f %7.0f subframes/s (cold, warm, warm)" % (rnd, b, name, r[0], r[1], r[2]), flush=True)
