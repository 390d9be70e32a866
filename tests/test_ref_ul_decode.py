"""SURVEY 8 row a15 pinned on the REFERENCE'S OWN CODE: PUSCH_Decoder::decode / decode_run / investigate_valid_ul_grant (/root/reference/src/src/UL_Sniffer_PUSCH.cc:
248-310, 389-583, 894-918 - which scheduled grants are tried at all, with which uplink MCS table and modulation in which order for which tracked maximum modulation,
the UCI layout each attempt is configured with, what a CRC-ok block writes and teaches the tracking database, the SNR gate of the statistics that age it) compiled
verbatim into oracle/_ref/libref_falcon_ul_decode.so on top of the reference's own MCSTracking (oracle/Makefile.ref; stand-in srsRAN receiver types and message classes:
oracle/ref_shim_search/srsran/standin_ul.h).  The uplink receiver itself is srsRAN's and absent: every srsran_chest_ul_estimate_pusch + srsran_pusch_decode pair is
recorded and answered by a SCRIPTED decoder (tests/ref_ul_decode.py), the same one that answers the oracle's attempts.  Committed as digests
(tests/golden/ul_decode_ref.json, made by tests/golden/make_ul_decode_fixture.py): five lives, 23 000 subframes, 24 000 attempts (walked at twice the length when the harness was built: 43 000 subframes, no difference) - the oracle's restatement
(o_worker.c: decode_pusch) reproduces every one; the product's trial order and grant test (lsn_lte.cc: ulTrialPlan, ulGrantValid - what commitChunkUl walks) are held
to the trial table read off the reference's decoder, and its record streams to the oracle's on the GPU (tests/test_gpu_ul.py).

What building the harness found: no difference in the decisions.  Two places where the reference hands srsRAN a grant that the oracle and the product do not attempt -
a RAR entry whose conversion failed (size 0; :419 lets every RAR entry through) and an allocation of 0 PRB (valid_prb_ul[0] is true) - are refused inside srsRAN and
cannot come out of a successful grant conversion; the lives leave them out.  How many bits a higher-layer sub-band report has on a band without sub-bands (up to
7 PRB: N = 0) is srsran_cqi_size's decision, not this file's: the pin compares N."""
import ctypes as C
import json
import os

import pytest

import ref_ul_decode as U
from lsn_testlib import hosttest

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ul_decode_ref.json")))
HAVE_LIB = os.path.exists(U.REF_SO)
LIFE = {l[0]: l for l in U.LIVES}


def test_fixture_is_whole():
    assert set(FIX["lives"]) == set(LIFE) and all(FIX["oracle_equal_when_made"].values())
    tot = {k: sum(v[k] for v in FIX["lives"].values()) for k in ("subframes", "attempts", "attempts_with_csi", "passed", "records")}
    assert tot["subframes"] >= 27000 and tot["attempts"] > 20000 and tot["attempts_with_csi"] > 3000 and tot["passed"] == tot["records"] > 12000, tot
    assert all(sum(v["attempts_by_modulation"][q] for v in FIX["lives"].values()) > 1000 for q in range(4))          # QPSK ... 256QAM
    assert all(sum(v["final_tracked_by_modulation"][q] for v in FIX["lives"].values()) > 10 for q in range(4))       # unknown, 16 / 64 / 256QAM maximum
    assert FIX["lives"]["crowd_25prb_full_buffer"]["final_tracked_by_modulation"][4] > 50                            # RNTIs that found the database full
    ages = [a for v in FIX["lives"].values() for a in v["tracked_after_each_ageing"]]
    assert len(ages) >= 18 and any(b < a for v in FIX["lives"].values() for a, b in zip(v["tracked_after_each_ageing"], v["tracked_after_each_ageing"][1:]))  # the ageing dropped someone
    assert len(FIX["trial_table"]) > 500


@pytest.mark.parametrize("name", sorted(LIFE))
def test_oracle_decides_like_the_reference(name):
    assert U.digest(U.run(U.Oracle(), LIFE[name])) == FIX["lives"][name]["digest"]


def test_products_trial_order_is_the_references():
    """lsn_lte.cc: ulTrialPlan / ulGrantValid (the functions commitChunkUl plans its decode waves with) against the order read off the reference's decoder"""
    h = hosttest()
    h.lsn_host_ul_trial.argtypes = [C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_int32)]
    n_att = 0
    for mcs in range(32):
        mod = 2 if mcs < 11 else 4 if mcs < 21 else 6
        mod256 = 2 if mcs < 6 else 4 if mcs < 14 else 6 if mcs < 23 else 8
        tbs, tbs256 = (0, 0) if mcs >= 29 else (800, 1600)
        valid = h.lsn_host_ul_grant_valid(0x1234, 0, tbs, tbs256, 10)
        assert valid == (0 if mcs >= 29 else 1)
        for L256 in (10, 0, 110):
            for state in (1, 2, 3, 4):
                out = (C.c_int32 * 9)()
                n = h.lsn_host_ul_trial(mcs, mod, L256, mod256, state, out) if valid else 0
                got = [[out[3 * i], out[3 * i + 1]] for i in range(n)]
                assert got == FIX["trial_table"]["%d/%d/%d" % (mcs, state, L256)], (mcs, state, L256, got)
                n_att += n
                if state == 1:
                    for k in range(n):
                        want = FIX["trial_table"]["learn %d/%d/%d" % (mcs, L256, k)]
                        assert (out[3 * k + 2] or 1) == want, (mcs, L256, k, out[3 * k + 2], want)
    assert n_att > 300
    # the grant test: RNTI 0, a size of 0 in either table, PRB counts that are no product of 2, 3, 5, more than 100 PRB; RAR entries pass as they are
    assert [h.lsn_host_ul_grant_valid(*a) for a in ((0, 0, 800, 1600, 10), (5, 0, 0, 1600, 10), (5, 0, 800, 0, 10), (5, 0, 800, 1600, 7), (5, 0, 800, 1600, 108), (5, 1, 0, 0, 7), (0, 1, 800, 1600, 10))] == [0, 0, 0, 0, 0, 1, 0]


@pytest.mark.skipif(not HAVE_LIB, reason="oracle/_ref/libref_falcon_ul_decode.so not built (needs /root/reference: make -C oracle -f Makefile.ref)")
def test_reference_library_reproduces_the_committed_fixture():
    ref = U.Reference()
    for name in ("mix_50prb", "six_prb"):
        assert U.digest(U.run(ref, LIFE[name])) == FIX["lives"][name]["digest"], name
    assert U.trial_table(ref) == FIX["trial_table"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/src"), reason="/root/reference is not here")
def test_fixture_was_made_from_the_reference_sources_that_are_here():
    assert U.reference_sources_sha256() == FIX["reference_sources_sha256"]
