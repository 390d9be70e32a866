#!/usr/bin/env python3
"""File-mode replay rate (host file -> pinned blocks -> PCIe -> k_file_unpack -> engine), the PCIe/IO-inclusive companion of
bench.py's resident-capture number.  usage: file_replay_bench.py [nsf=12800] [gen=800] [cf32|sc16]
(ungated: a rate probe for block-size / reader-count experiments; the gated figures are bench.py's file_replay / file_replay_sc16 legs)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import ltesniffer_amd as la
from lsn_testlib import scenario
from parity import gen_subframes

nsf = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
gen = int(sys.argv[2]) if len(sys.argv) > 2 else 800
fmt = sys.argv[3] if len(sys.argv) > 3 else "cf32"
sc = scenario("cfg3", seed=3)
tti0, iq, _ = gen_subframes(sc, gen)
path = "/dev/shm/lsn_capture.cf32" if os.path.isdir("/dev/shm") else "/tmp/lsn_capture.cf32"
blockdata = np.ascontiguousarray(np.transpose(iq, (0, 2, 1)))
kw = {}
if fmt == "sc16":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_cfg3_golden import sc16_capture
    q, lsb = sc16_capture(iq)
    blockdata = np.ascontiguousarray(np.transpose(q, (0, 2, 1, 3)))
    kw = dict(sample_format=la.FILE_SC16, sample_scale=lsb)
with open(path, "wb") as f:
    for _ in range(nsf // gen):
        blockdata.tofile(f)
size = os.path.getsize(path)
with open(path, "rb", buffering=0) as f:   # read once: the first read of freshly written page-cache pages is slow whoever reads them
    buf = bytearray(64 << 20)
    while f.readinto(buf):
        pass
phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=int(os.environ.get("LSN_BENCH_BATCH", "400")), pcapwriter=la.PcapWriter(None))
assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
phy.prepare_file()
for rep in range(3):
    phy.pcapwriter.reset()
    t0 = time.perf_counter()
    done = phy.process_file(path, start_tti=tti0, update_meta_period=500, **kw)
    dt = time.perf_counter() - t0
    print("replay %d: %d subframes in %.3f s = %.0f subframes/s (%.2f GB/s from the file, page cache)" % (rep, done, dt, done / dt, size / dt / 1e9), flush=True)
phy.close()
os.remove(path)
