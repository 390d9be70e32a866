#!/usr/bin/env python3
"""tools/bench_legs.py - the other BASELINE.json configurations bench.py times next to the headline (configs[1], configs[1] on four CRS
ports, configs[2] at 16 dB, configs[3] = UL_MODE), defined ONCE here, and the CPU oracle's record stream over each of them, cached per block
so that bench.py can gate every throughput figure it prints on the oracle (round-4 review: "no figure on the line without a gate").

A leg's stream: a capture of `nsf` distinct subframes replayed `passes` times with the TTI advancing and every sequential state carried
over (pass 1 runs from cold state and is not timed); meta formats update whenever the stream position is a multiple of 500
(LTESniffer_Core.cc:434).  The oracle (scalar C, one thread) walks passes x nsf subframes once; its records are hashed per block of 200
subframes by the product's pcap writer (the hash function only: lsn_pcap_set_digest_blocks) -> tests/golden/leg_<name>.json, keyed by the
xxh3 of the capture bytes.

  python tools/bench_legs.py <leg> [--out PATH]        (run in the background: the 16 dB leg walks 80 000 subframes)
  python tools/bench_legs.py --list"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]

BLOCK = 200
META_PERIOD = 500

# name -> definition.  kind "dl": scenario(preset, **over) rendered by tools/txgen; kind "ul": UL_MODE (antenna 0 = downlink of the scenario,
# antenna 1 = the uplink the UEs answer its DCI 0 with, lsn_testlib.gen_ul_mode_subframes), `gen` distinct subframes tiled to `nsf`
LEGS = {
    "cfg2_32_rnti_tm2_64qam": dict(kind="dl", preset="cfg2", over=dict(seed=2), nsf=3200, passes=5,
                                   what="BASELINE configs[1]: 20 MHz, 32 RNTIs, TM2 64QAM"),
    "cfg2_on_four_crs_ports": dict(kind="dl", preset="cfg2", over=dict(seed=2, nof_ports=4), nsf=3200, passes=5,
                                   what="configs[1] on a four-port cell (SFBC-FSTD on every channel)"),
    "cfg3_at_16_dB_snr": dict(kind="dl", preset="cfg3", over=dict(seed=16, snr_db=16.0), nsf=20000, passes=4,
                              what="BASELINE configs[2] at 16 dB instead of 30 dB: most code blocks need many iterations, many fail"),
    "cfg3_16_dB_harq_mode_1": dict(kind="dl", preset="cfg3", over=dict(seed=17, snr_db=16.0, pct_harq=50), nsf=3200, passes=3, harq_mode=1,
                                   what="configs[2] at 16 dB, half of the transport blocks sent again 8 subframes later, harq_mode = 1 (soft combining: retransmissions combined and decoded in batches ahead of the commit walk)"),
    "cfg3_on_eva70_fading": dict(kind="dl", preset="cfg3", over=dict(seed=18, snr_db=24.0, chan_model=2, doppler_hz=70.0, timing_offset_samples=0.37), nsf=1600, passes=3,
                                 what="configs[2] through the EVA channel of TS 36.101 B.2 at 70 Hz Doppler (independent Rayleigh taps per antenna and port), 24 dB, 0.37 samples of timing offset"),
    "cfg4_ul_mode_64_rnti": dict(kind="ul", preset="cfg2", over=dict(seed=4, nof_rx=1, n_rnti=64, ul_min=2, ul_max=4, mcs_min=0, mcs_max=28, snr_db=28.0),
                                 gen=200, nsf=3200, passes=3, ul_snr_db=22.0, cyclic_shift=3, delta_ss=5,
                                 what="BASELINE configs[3]: UL_MODE, 64 RNTIs, PUSCH at n + 4 with 16/64QAM turbo decodes"),
}


def golden_path(name):
    return os.path.join(ROOT, "tests", "golden", "leg_%s.json" % name)


def leg_scenario(name):
    from lsn_testlib import scenario
    d = LEGS[name]
    return scenario(d["preset"], **d["over"])


def leg_capture(name, threads=None):
    """-> (scenario dict, tti0, iq[nsf, antennas, sf_len] complex64)"""
    import numpy as np
    d = LEGS[name]
    sc = leg_scenario(name)
    if d["kind"] == "dl":
        from parity import gen_capture
        tti0, iq = gen_capture(sc, d["nsf"], threads=threads)
        return sc, tti0, iq
    from lsn_testlib import gen_ul_mode_subframes
    tti0, iq, _ = gen_ul_mode_subframes(sc, d["gen"], cyclic_shift=d["cyclic_shift"], delta_ss=d["delta_ss"], ul_snr_db=d["ul_snr_db"])
    return sc, tti0, np.ascontiguousarray(np.tile(iq, (d["nsf"] // d["gen"], 1, 1)))


def load_golden(name, iq, tti0):
    """the cached oracle stream of a leg when it belongs to THIS capture -> (golden dict | None, note | None)"""
    from make_cfg3_golden import capture_hash
    try:
        g = json.load(open(golden_path(name)))
    except Exception as ex:
        return None, "no cached oracle stream: %s" % str(ex)[:100]
    chash, _ = capture_hash(iq)
    if g["capture_xxh3_64"] != chash:
        return None, "the capture rendered on this host (xxh3 %s) is not the one the cached oracle stream was made from (%s)" % (chash, g["capture_xxh3_64"])
    if g["stream"]["block_subframes"] != BLOCK or g["stream"]["tti0"] != tti0 or g["stream"]["meta_period"] != META_PERIOD:
        return None, "cached oracle stream was made with another block / tti0 / meta period"
    return g, None


def check_blocks(golden, blocks, first_block=0):
    """product blocks [(digest, nrec)] against the oracle's from stream block `first_block` on -> (compared, mismatching, record-count difference)"""
    if golden is None:
        return 0, None, None
    ob = golden["blocks"]
    n = max(0, min(len(blocks), len(ob) - first_block))
    bad, rd = 0, 0
    for j in range(n):
        d, c = blocks[j]
        if "%016x" % d != ob[first_block + j][0] or c != ob[first_block + j][1]:
            bad += 1
            rd += abs(c - ob[first_block + j][1])
    return n, bad, rd


def make_golden(name, out=None, threads=None, limit=None):
    import ctypes as C
    import ltesniffer_amd as la
    from lsn_testlib import OracleWorker, OracleWorkerUl, parse_pcap
    from make_cfg3_golden import capture_hash, source_hash
    d = LEGS[name]
    out = out or golden_path(name)
    t = time.time()
    sc, tti0, iq = leg_capture(name, threads)
    chash, cparts = capture_hash(iq)
    nsf = d["nsf"]
    total = d["passes"] * nsf if not limit else min(limit, d["passes"] * nsf) // BLOCK * BLOCK
    print("%s: capture of %d subframes in %.0f s, xxh3 %s; oracle walk of %d" % (name, nsf, time.time() - t, chash, total), flush=True)
    if d["kind"] == "dl":
        ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"], sc["phich_ng_x6"])
        if d.get("harq_mode"):
            ow.set_harq(d["harq_mode"])
    else:
        ow = OracleWorkerUl(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], d["cyclic_shift"], d["delta_ss"])
    w = la.PcapWriter(None)  # the product's writer as the hash function over the ORACLE's records
    w.set_store(False)
    w.set_digest_blocks(BLOCK, tti0)
    lib = la.lib()

    def save(done):
        blocks = w.block_digests()[:done // BLOCK]
        blocks += [(0x9E3779B97F4A7C15, 0)] * (done // BLOCK - len(blocks))  # trailing blocks without a record
        o = {"stream": {"leg": name, "definition": {k: v for k, v in d.items()}, "distinct_subframes": nsf, "block_subframes": BLOCK, "meta_period": META_PERIOD,
                        "tti0": tti0, "scenario": sc},
             "capture_xxh3_64": chash, "source_hash": source_hash(), "oracle_subframes": done, "oracle_records": sum(c for _, c in blocks),
             "blocks": [["%016x" % dg, c] for dg, c in blocks]}
        json.dump(o, open(out + ".tmp", "w"))
        os.replace(out + ".tmp", out)

    t = time.time()
    for i in range(total):
        upd = 1 if i % META_PERIOD == 0 else 0
        x = iq[i % nsf]
        if d["kind"] == "dl":
            ow.work(x, tti0 + i, update_meta=upd)
        else:
            ow.work_ul(x[0], x[1], tti0 + i, update_meta=upd)
        if (i + 1) % BLOCK == 0:
            for r in parse_pcap(ow.pcap_bytes()):
                c = r["ctx"]
                fs = (c[10] << 8) | c[11]
                ctx = la.PduCtx((fs >> 4) * 10 + (fs & 15), (c[4] << 8) | c[5], c[1], c[2], c[13], 0, 0, 0)
                assert lib.lsn_pcap_write(w._h, C.byref(ctx), r["pdu"], len(r["pdu"])) == 0
            old = ow.pcap
            ow.pcap = ow.lib.o_pcap_open_mem()
            ow.lib.o_worker_set_pcap(ow.h, ow.pcap)
            ow.lib.o_pcap_close(old)
            if (i + 1) % (10 * BLOCK) == 0:
                print("%s: %d / %d subframes, %.1f sf/s, %d records" % (name, i + 1, total, (i + 1) / (time.time() - t), w.nof_records()), flush=True)
                save(i + 1)
    save(total)
    print("wrote", out, flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("leg", nargs="?")
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--threads", type=int, default=None)
    ap.add_argument("--limit", type=int, default=None, help="walk only this many subframes (development)")
    a = ap.parse_args()
    if a.list or not a.leg:
        for k, v in LEGS.items():
            print("%-28s %s  (%d x %d subframes)" % (k, v["what"], v["passes"], v["nsf"]))
    else:
        make_golden(a.leg, a.out, a.threads, a.limit)
