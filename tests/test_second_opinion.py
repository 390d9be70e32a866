"""Second-opinion decoders (oracle/o_second.c): a full-trellis, 16-bit-input max-log-MAP turbo decoder and a float tail-biting Viterbi
that share none of the production oracle's design parameters (trellis windows, 10-bit soft values, 3/4 extrinsic scaling, u8 Viterbi
symbols).  These tests check (a) that both opinions are decoders at all (loop-back at high SNR), and (b) that the production oracle's
verdicts agree with them except near the waterfall, where the count of differing verdicts is small and neither emits bytes that were not sent - the
full sweep lives in profiles/r02_second_opinion.txt (tools/second_opinion_sweep.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
from lsn_testlib import scenario  # noqa: E402
from second_opinion_sweep import compare  # noqa: E402


def test_high_snr_both_opinions_decode_the_same_stream():
    r = compare(scenario("cfg3", seed=5, snr_db=32.0, n_rnti=30), 16)
    assert r["prod"] > 100 and r["prod"] == r["second"] and r["only_prod"] == r["only_second"] == 0 and r["false_prod"] == r["false_second"] == 0
    assert r["dci_prod"] == r["dci_second"] > 100


def test_mid_snr_verdicts_differ_rarely_and_payloads_never():
    r = compare(scenario("cfg3", seed=6, snr_db=14.0, n_rnti=30), 16)
    assert r["prod"] > 40
    assert r["false_prod"] == r["false_second"] == 0       # neither opinion emits bytes that were not sent
    assert r["only_prod"] + r["only_second"] <= 0.08 * max(r["prod"], r["second"]), r
    assert abs(r["dci_prod"] - r["dci_second"]) <= 0.05 * r["dci_prod"], r


def test_small_cell_qpsk_agrees():
    r = compare(scenario("cfg1", seed=7, snr_db=8.0), 40)
    assert r["false_prod"] == r["false_second"] == 0 and r["only_prod"] + r["only_second"] <= max(2, 0.1 * max(r["prod"], r["second"])), r
