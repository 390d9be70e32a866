"""The gated cfg3 stream of bench.py (tools/make_cfg3_golden.py): the pieces that make a cached oracle stream trustworthy, without a GPU -
the capture renders identically in parallel and sequentially, the block digests cut the stream where they should, and the cached file still
describes today's sources (capture bytes, oracle records of the first block)."""
import json
import os
import sys

import numpy as np
import pytest

import ltesniffer_amd as la
from lsn_testlib import scenario
from parity import gen_capture, gen_subframes, run_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_cfg3_golden as mg  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "cfg3_stream_oracle.json")


def test_parallel_capture_equals_sequential_calls():
    sc = scenario("cfg3", seed=3)
    tti0, iq, _ = gen_subframes(sc, 60)
    for thr in (1, 3):
        t1, iq1 = gen_capture(sc, 60, threads=thr)
        assert t1 == tti0 and np.array_equal(iq.view(np.uint32), iq1.view(np.uint32))


def test_block_digests_follow_the_unwrapped_tti():
    w = la.PcapWriter(None)
    w.set_digest_blocks(200, 10100)
    ttis = (10100, 10100, 10239, 0, 60, 59, 61, 500)  # wraps at 10240, steps back by one subframe once (uplink records do), skips a block
    for t in ttis:
        w.write(dict(tti=t, rnti=70, direction=1, rnti_type=3), b"abc" * 30)
    b = w.block_digests()
    assert [c for _, c in b] == [5, 2, 0, 1]
    # a block's digest is the digest of exactly its records on a fresh chain
    w2 = la.PcapWriter(None)
    for t in (10100, 10100, 10239, 0, 59):
        w2.write(dict(tti=t, rnti=70, direction=1, rnti_type=3), b"abc" * 30)
    assert w2.digest()[0] == b[0][0]
    w.reset()
    assert w.block_digests() == []


@pytest.mark.skipif(not os.path.exists(GOLDEN), reason="no cached oracle stream (tools/make_cfg3_golden.py)")
def test_cached_oracle_stream_matches_todays_sources():
    """the cache is only as good as its agreement with TODAY's transmitter and oracle: the head of the capture renders to the cached bytes and
    the oracle, walked here from cold state, reproduces the first three cached blocks record for record (bench.py repeats this over 1 600
    subframes on the GPU box, `parity.live_oracle_reproduces_cached_blocks`).  `source_hash` names the sources the cache was made from."""
    g = json.load(open(GOLDEN))
    sc, nsf, blk, meta = mg.cfg3_stream()
    assert g["stream"]["distinct_subframes"] == nsf >= 20000 and g["stream"]["block_subframes"] == blk and g["stream"]["scenario"] == sc
    assert g["oracle_subframes"] >= 5 * nsf and len(g["blocks"]) == g["oracle_subframes"] // blk   # the driver's 25 steps of 4 000 subframes
    assert len(g["source_hash"]) == 16
    tti0, iq = gen_capture(sc, 1000)
    h, parts = mg.capture_hash(iq)
    assert tti0 == g["stream"]["tti0"] and parts[0] == g["capture_xxh3_64_per_1000"][0]
    nb = 3
    _, _, recs = run_oracle(sc, tti0, iq[:nb * blk], update_meta_period=meta, taps=False)
    w = la.PcapWriter(None)
    w.set_digest_blocks(blk, tti0)
    for r in recs:
        c = r["ctx"]
        fs = (c[10] << 8) | c[11]
        w.write(dict(tti=(fs >> 4) * 10 + (fs & 15), rnti=(c[4] << 8) | c[5], direction=c[1], rnti_type=c[2], crc_ok=c[13]), r["pdu"])
    got = [["%016x" % d, n] for d, n in w.block_digests()[:nb]]
    assert got == g["blocks"][:nb], "oracle / tables / transmitter changed the stream: run tools/make_cfg3_golden.py again"


GOLDEN_SC16 = os.path.join(ROOT, "tests", "golden", "cfg3_stream_sc16_oracle.json")


@pytest.mark.skipif(not os.path.exists(GOLDEN_SC16), reason="no cached oracle stream of the 16-bit recording (tools/make_cfg3_golden.py --sc16)")
def test_cached_sc16_oracle_stream_matches_todays_sources():
    """the stream behind bench.py's file_replay_sc16 leg: the same capture recorded as int16 pairs, the oracle on what the file source makes of
    them ((float)integer * lsb).  The head of the recording quantises to the cached bytes and the oracle reproduces the first two cached blocks."""
    g = json.load(open(GOLDEN_SC16))
    sc, nsf, blk, meta = mg.cfg3_stream()
    assert g["sample_format"] == "sc16" and g["stream"]["distinct_subframes"] == nsf and g["stream"]["block_subframes"] == blk and g["stream"]["scenario"] == sc
    assert g["oracle_subframes"] >= 3 * nsf and len(g["blocks"]) == g["oracle_subframes"] // blk   # the leg's three passes
    assert g["cf32_capture_xxh3_64"] == json.load(open(GOLDEN))["capture_xxh3_64"]               # made from the headline's capture
    lsb = g["lsb"]
    assert np.log2(lsb) == np.round(np.log2(lsb))                                                # a power of two: the conversion is exact
    tti0, iq = gen_capture(sc, 1000)
    q, lsb2 = mg.sc16_capture(iq, gain=1.0 / lsb)
    assert lsb2 == lsb and np.abs(q).max() <= int(0.98 * 32767) + 1
    assert mg.capture_hash(q)[1][0] == g["capture_xxh3_64_per_1000"][0]
    nb = 2
    sub = mg.sc16_subframes(q[:nb * blk], lsb)
    assert sub.dtype == np.complex64 and np.abs(sub - iq[:nb * blk]).max() <= 0.71 * lsb
    _, _, recs = run_oracle(sc, tti0, sub, update_meta_period=meta, taps=False)
    w = la.PcapWriter(None)
    w.set_digest_blocks(blk, tti0)
    for r in recs:
        c = r["ctx"]
        fs = (c[10] << 8) | c[11]
        w.write(dict(tti=(fs >> 4) * 10 + (fs & 15), rnti=(c[4] << 8) | c[5], direction=c[1], rnti_type=c[2], crc_ok=c[13]), r["pdu"])
    got = [["%016x" % d, n] for d, n in w.block_digests()[:nb]]
    assert got == g["blocks"][:nb], "oracle / tables / transmitter changed the stream: run tools/make_cfg3_golden.py --sc16 again"
