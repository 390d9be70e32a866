#!/usr/bin/env python3
"""tools/make_cfg3_golden.py - the CPU oracle's record stream over BASELINE.json configs[2] as SURVEY.md 8(d) specifies it, cached
piecewise so that bench.py can gate its TIMED stream on the oracle (not on a second pass of the product).

Stream definition (shared with bench.py, `cfg3_stream()` below is the single source):
  * capture: `scenario("cfg3", seed=3)`, NSF = 20 000 distinct subframes (a fresh RNTI by RAR every 200 subframes -> 100 new UEs on top of
    the 150 initial ones, crossing the 250-entry ageing logic of the MCS-tracking database), rendered by tools/txgen;
  * the capture is replayed cyclically while the TTI keeps advancing (20 000 is a multiple of 20: subframe index / SIB pattern stay
    consistent) and all sequential state (RNTI histograms, MCS tables, meta formats, database clocks) carries over;
  * meta-format update every 500 subframes of the stream (LTESniffer_Core.cc:434).
The oracle (scalar C, one thread - its state is a sequential scan) walks the first --subframes of that stream once; the records of every
BLOCK = 200 subframes are hashed on their own chain by the product's pcap writer (lsn_pcap_set_digest_blocks) fed with the ORACLE's
records.  Output: tests/golden/cfg3_stream_oracle.json = capture hash + per-block (digest, record count).  bench.py hashes the blocks of
what the HIP pipeline wrote in its timed region and compares block by block.

  python tools/make_cfg3_golden.py [--subframes 100000] [--out tests/golden/cfg3_stream_oracle.json]
About 40 ms per subframe on one core of this container: 100 000 subframes ~ 70 min (run it in the background)."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

NSF = 20000
BLOCK = 200
SEED = 3
META_PERIOD = 500


def cfg3_stream():
    """(scenario dict, distinct subframes, block length, meta-format period) of the gated stream"""
    from lsn_testlib import scenario
    return scenario("cfg3", seed=SEED), NSF, BLOCK, META_PERIOD


def capture_hash(iq, step=1000):
    """xxh3-64 of the capture bytes (fast enough for 10 GB) + one digest per `step` subframes to localise a difference"""
    import xxhash
    h = xxhash.xxh3_64()
    parts = []
    for a in range(0, iq.shape[0], step):
        b = memoryview(iq[a:a + step]).cast("B")
        h.update(b)
        parts.append(xxhash.xxh3_64_hexdigest(b))
    return h.hexdigest(), parts


def sc16_capture(iq, step=1000, gain=None):
    """the capture as a 16-bit recording (lsn_file_cfg_t.sample_format = LSN_FILE_SC16): int16 I/Q pairs [nsf, antennas, sf_len, 2] at the largest
    power-of-two gain that keeps the peak inside 98 % of full scale (or at `gain`) -> (integers, value of one LSB)"""
    import numpy as np
    if gain is None:
        peak = max(float(max(np.abs(iq[a:a + step].real).max(), np.abs(iq[a:a + step].imag).max())) for a in range(0, iq.shape[0], step))
        gain = 2.0 ** np.floor(np.log2(0.98 * 32767 / peak))
    q = np.empty(iq.shape + (2,), dtype=np.int16)
    for a in range(0, iq.shape[0], step):
        q[a:a + step] = np.rint(iq[a:a + step].view(np.float32).reshape(iq[a:a + step].shape + (2,)) * np.float32(gain)).astype(np.int16)
    return q, float(1.0 / gain)


def sc16_subframes(q, lsb):
    """what the file source makes of the recording: (float)integer * lsb, complex64 [n, antennas, sf_len]"""
    import numpy as np
    return np.ascontiguousarray(q.astype(np.float32) * np.float32(lsb)).view(np.complex64)[..., 0]


def source_hash():
    """what the cached stream depends on: the oracle, the tables, the transmitter (a change there means: run this tool again)"""
    h = hashlib.sha256()
    files = sorted(os.path.join("oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith((".c", ".h")))
    files += ["spec/lte_tables.h", "tools/txgen/txgen.cc"]
    for f in files:
        h.update(f.encode())
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--subframes", type=int, default=100000)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "cfg3_stream_oracle.json"))
    ap.add_argument("--threads", type=int, default=None)
    ap.add_argument("--sc16", action="store_true", help="the stream of the capture recorded as int16 pairs (bench.py's file_replay_sc16 leg); "
                    "--out tests/golden/cfg3_stream_sc16_oracle.json --subframes 60000")
    args = ap.parse_args()
    import ctypes as C
    import ltesniffer_amd as la
    from lsn_testlib import OracleWorker, parse_pcap
    from parity import gen_capture

    sc, nsf, blk, meta = cfg3_stream()
    total = (args.subframes // blk) * blk
    t = time.time()
    tti0, iq = gen_capture(sc, nsf, threads=args.threads)
    chash, cparts = capture_hash(iq)
    print("capture: %d subframes in %.0f s, xxh3 %s" % (nsf, time.time() - t, chash), flush=True)
    extra = {}
    if args.sc16:
        q, lsb = sc16_capture(iq)
        del iq
        qhash, cparts = capture_hash(q)
        extra = {"sample_format": "sc16", "lsb": lsb, "cf32_capture_xxh3_64": chash}
        chash = qhash
        print("as a 16-bit recording: one LSB = %g, xxh3 %s" % (lsb, chash), flush=True)

        class _Deq:   # the oracle walks what the file source delivers: (float)integer * lsb
            def __getitem__(self, i):
                return sc16_subframes(q[i:i + 1], lsb)[0]
        iq = _Deq()

    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"], sc["phich_ng_x6"])
    w = la.PcapWriter(None)           # the product's writer, used here as the hash function over the ORACLE's records
    w.set_store(False)
    w.set_digest_blocks(blk, tti0)
    lib = la.lib()
    t = time.time()
    for i in range(total):
        ow.work(iq[i % nsf], tti0 + i, update_meta=1 if i % meta == 0 else 0)
        if (i + 1) % blk == 0:
            for r in parse_pcap(ow.pcap_bytes()):
                c = r["ctx"]
                # MAC-LTE context (lsn_pcap.cc / PcapWriter.cc:97-111): [1]=direction [2]=rnti type [4:6]=rnti [10:12]=sfn<<4|sf [13]=crc
                fs = (c[10] << 8) | c[11]
                ctx = la.PduCtx((fs >> 4) * 10 + (fs & 15), (c[4] << 8) | c[5], c[1], c[2], c[13], 0, 0, 0)
                rc = lib.lsn_pcap_write(w._h, C.byref(ctx), r["pdu"], len(r["pdu"]))
                assert rc == 0
            old = ow.pcap
            ow.pcap = ow.lib.o_pcap_open_mem()
            ow.lib.o_worker_set_pcap(ow.h, ow.pcap)
            ow.lib.o_pcap_close(old)
            if (i + 1) % (10 * blk) == 0:
                dt = time.time() - t
                print("oracle: %d / %d subframes, %.1f sf/s, %d records" % (i + 1, total, (i + 1) / dt, w.nof_records()), flush=True)
                _save(args.out, sc, nsf, blk, meta, tti0, chash, cparts, w, i + 1, extra)
    _save(args.out, sc, nsf, blk, meta, tti0, chash, cparts, w, total, extra)
    print("wrote", args.out)


def _save(path, sc, nsf, blk, meta, tti0, chash, cparts, w, done, extra=None):
    blocks = w.block_digests()[:done // blk]
    out = {"stream": {"config": "cfg3", "seed": SEED, "distinct_subframes": nsf, "block_subframes": blk, "meta_period": meta, "tti0": tti0,
                      "scenario": sc},
           "capture_xxh3_64": chash, "capture_xxh3_64_per_1000": cparts, "source_hash": source_hash(),
           "oracle_subframes": done, "oracle_records": sum(c for _, c in blocks),
           "blocks": [["%016x" % d, c] for d, c in blocks]}
    out.update(extra or {})
    tmp = path + ".tmp"
    json.dump(out, open(tmp, "w"))
    os.replace(tmp, path)


if __name__ == "__main__":
    main()
