#!/usr/bin/env python3
"""tools/host_leg_probe.py - where does the host-buffer path (lsn_phy_process_host) lose time?  One engine, the cfg3 capture head in pinned
memory: (a) resident submit, (b) process_host, (c) resident submit while an unrelated stream copies host -> device at full rate.
Run under `rocprofv3 --kernel-trace --memory-copy-trace` for the copy / kernel overlap picture."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import ltesniffer_amd as la  # noqa: E402
from make_cfg3_golden import cfg3_stream  # noqa: E402
from parity import gen_capture  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 800
sc, *_ = cfg3_stream()
tti0, iq = gen_capture(sc, n)
host = torch.from_numpy(iq).pin_memory()
d = torch.from_numpy(iq.view(np.float32)).to("cuda:0")
w = la.PcapWriter(None)
w.set_store(False)
phy = la.Phy(nof_rx_antennas=2, max_batch=batch, pcapwriter=w)
phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
phy.process_device(d.data_ptr(), n, tti0, 500)          # learn the tables
t = 0


def timed(name, fn):
    global t
    t += n
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn((tti0 + t) % 10240)
    dt = time.perf_counter() - t0
    print("%-44s %8.0f subframes/s  %6.2f GB/s" % (name, n / dt, n * iq[0].nbytes / dt / 1e9), flush=True)


timed("resident (submit_device + wait)", lambda tt: (phy.submit_device(d.data_ptr(), n, tt, 500), phy.wait()))
timed("host pinned (process_host)", lambda tt: phy.process_host(host.numpy(), tt, 500))
side = torch.cuda.Stream()
dst = torch.empty_like(d)


def resident_with_copies(tt):
    phy.submit_device(d.data_ptr(), n, tt, 500)
    with torch.cuda.stream(side):
        for a in range(0, n, 800):
            dst[a:a + 800].copy_(torch.from_numpy(host.numpy()[a:a + 800].view(np.float32)).view(dst[a:a + 800].shape), non_blocking=True)
    phy.wait()
    side.synchronize()


timed("resident + unrelated H2D copies of the capture", resident_with_copies)
timed("host pinned again", lambda tt: phy.process_host(host.numpy(), tt, 500))
p = phy.perf()
print("ms_wait_slot %.1f ms_wait_front %.1f ms_stage_a %.1f ms_search %.1f ms_drain %.1f" % (p.ms_wait_slot, p.ms_wait_front, p.ms_stage_a, p.ms_search, p.ms_drain))
phy.close()
