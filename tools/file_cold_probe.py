#!/usr/bin/env python3
"""First call of the file source on a fresh engine whose block buffers were reserved (lsn_phy_prepare_file) against the following calls:
tools/file_cold_probe.py [nsf=20000] [gen=800]   (LSN_FILE_DEBUG=1: per-block read times; LSN_PROBE_PREREAD=1: the file is read once before the engine opens it)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import ltesniffer_amd as la
from lsn_testlib import scenario
from parity import gen_capture

nsf = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
gen = int(sys.argv[2]) if len(sys.argv) > 2 else 800
sc = scenario("cfg3", seed=3)
tti0, iq = gen_capture(sc, gen)
path = "/dev/shm/lsn_cold_probe_%d.cf32" % os.getpid()
blockdata = np.ascontiguousarray(np.transpose(iq, (0, 2, 1)))
with open(path, "wb") as f:
    for _ in range(nsf // gen):
        blockdata.tofile(f)
try:
    if os.environ.get("LSN_PROBE_PREREAD"):   # the freshly written tmpfs pages are read once by this (unpinned) thread before the engine sees the file
        buf = bytearray(64 << 20)
        t0 = time.perf_counter()
        with open(path, "rb", buffering=0) as f:
            n = 0
            while True:
                k = f.readinto(buf)
                if not k:
                    break
                n += k
        print("pre-read %.2f GB in %.3f s" % (n / 1e9, time.perf_counter() - t0), flush=True)
    w = la.PcapWriter(None)
    w.set_store(False)
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=400, pcapwriter=w)
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    t0 = time.perf_counter()
    phy.prepare_file()
    print("prepare_file %.3f s" % (time.perf_counter() - t0), flush=True)
    for rep in range(3):
        w.reset()
        t0 = time.perf_counter()
        done = phy.process_file(path, start_tti=tti0, update_meta_period=500)
        dt = time.perf_counter() - t0
        print("replay %d: %d subframes in %.3f s = %.0f subframes/s" % (rep, done, dt, done / dt), flush=True)
    phy.close()
finally:
    os.remove(path)
