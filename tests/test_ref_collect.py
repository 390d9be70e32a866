"""SURVEY 8 rows a11 / a12 / a15 pinned on the REFERENCE'S OWN CODE: /root/reference/src/src/DCICollection.cc (addCandidate: the MCS table an accepted DCI is
collected under, the hopping configuration from SIB2, the RB collision maps, the size of a reserved MCS from the HARQ database, transport blocks without bits
disabled), lib/src/phy/falcon_phch/falcon_dci.c (srsran_dci_msg_to_trace_timestamp: which grant conversion runs under which table, RNTI 0 after a failed one;
the RAR grant -> DCI 0 -> PUSCH grant chain), dl_sniffer_pdsch.c INCLUDING the C-RNTI branch of dl_sniffer_ra_dl_dci_to_grant, ul_sniffer_pusch.c and
src/src/ULSchedule.cc, compiled verbatim into oracle/_ref/libref_falcon_collect.so next to MCSTracking.cc and HARQ.cc (oracle/Makefile.ref; stand-in srsRAN types;
collect_glue.cc binds the DCI bit unpacking and the TBS table to the oracle and states the resource-allocation functions of TS 36.213 7.1.6 / 7.1.7 and the
distributed-VRB interleaver of TS 36.211 6.2.3.2 a second time, independently of o_dci.c).  Its answers to the scripts of tests/ref_collect.py are committed as
digests (tests/golden/collect_ref.json, made by tests/golden/make_collect_fixture.py, which also walked the lives at ten times the length: 100 000 downlink and
18 000 uplink entries, no difference); the oracle's restatement (o_worker.c: add_candidate, o_dci.c) and the product's host code (lsn_search.cc: finishSubframe +
the commit-side helpers the engine uses) must give the same answers.  Where the library is present the reference itself runs again."""
import json
import os

import pytest

import ref_collect as R
from lsn_testlib import hosttest, oracle

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "collect_ref.json")))
HAVE_LIB = os.path.exists(R.REF_SO)
LIFE = {l[0]: l for l in R.LIVES}


def test_fixture_is_whole():
    assert set(FIX["lives"]) == set(LIFE) and all(FIX["oracle_equal_when_made"].values()) and all(FIX["product_equal_when_made"].values())
    tot = {k: sum(v[k] for v in FIX["lives"].values()) for k in ("dl_entries", "ul_entries", "dl_conversion_failed", "ul_conversion_failed", "ul_type1_hopping",
                                                                  "size_from_harq_database", "two_block_grants", "slot_hopping_allocations", "dl_collisions", "ul_collisions")}
    assert tot["dl_entries"] > 9000 and tot["ul_entries"] > 1500 and all(v > 0 for v in tot.values()), tot
    assert all(sum(v["dl_by_format"][f] for v in FIX["lives"].values()) > 100 for f in range(1, 9))          # every downlink format
    assert all(sum(v["dl_by_table"][t] for v in FIX["lives"].values()) > 500 for t in (0, 1, 2))             # 64QAM, 256QAM and unknown table
    assert FIX["rar"]["oracle_equal_when_made"] and FIX["rar"]["product_equal_when_made"] and 0 < FIX["rar"]["grants_converted"] < FIX["rar"]["cases"]
    lr = FIX["long_run"]
    assert sum(v["dl_entries"] for v in lr.values()) > 90_000 and all(v["oracle_differs_in"] == 0 and v.get("product_differs_in", 0) == 0 for v in lr.values())


@pytest.mark.parametrize("name", sorted(LIFE))
def test_oracle_collects_like_the_reference(name):
    assert R.digest(R.Oracle().run(LIFE[name])) == FIX["lives"][name]["digest"]


@pytest.mark.parametrize("name", sorted(n for n, l in LIFE.items() if not l[7]))
def test_product_collects_like_the_reference(name):
    p = R.Product()
    assert R.digest([p.view(x) for x in p.run(LIFE[name])]) == FIX["lives"][name]["digest_without_maps"]


def test_rar_grant_chain_of_oracle_and_product_is_the_references():
    o, h = oracle(), hosttest()
    assert R.digest_rows([tuple(v & 0xFFFFFFFF for v in R.rar_oracle(o, *a)) for a in R.rar_sweep()]) == FIX["rar"]["digest"]
    assert R.digest_rows([tuple(v & 0xFFFFFFFF for v in R.rar_product(h, *a)) for a in R.rar_sweep()]) == FIX["rar"]["digest"]


def test_ulschedule_is_a_map_by_grant_tti_and_a_ring_is_the_same_without_gaps():
    """ULSchedule.cc:11-138.  The ten-line model of tests/ref_collect.py IS the reference's behaviour on both scripts (fixture); oracle and product keep a ring of
    16 slots / a map bounded to 64 entries, which hands out the same lists on a stream without gaps (file replay).  With skipped subframes and restarts the
    reference's never-deleted lists come back 10 240 subframes later, appended to the new ones: counted, not imitated."""
    for name, gaps in (("gapless", False), ("with_gaps_and_restarts", True)):
        s = R.ulsche_script(gaps=gaps)
        m = R.ulsche_model(s)
        f = FIX["ulsche"][name]
        assert f["model_equal"] and R.digest_rows(m) == f["digest"] and len(m) == f["fetches"]
        assert sum(a != b for a, b in zip(m, R.ulsche_ring(s))) == f["fetches_where_a_16_slot_ring_differs"]
    assert FIX["ulsche"]["gapless"]["fetches_where_a_16_slot_ring_differs"] == 0 and FIX["ulsche"]["gapless"]["non_empty"] > 5000
    assert FIX["ulsche"]["with_gaps_and_restarts"]["fetches_where_a_16_slot_ring_differs"] > 0


@pytest.mark.skipif(not HAVE_LIB, reason="oracle/_ref/libref_falcon_collect.so not built (needs /root/reference: make -C oracle -f Makefile.ref)")
def test_reference_library_reproduces_the_committed_fixture():
    ref = R.Reference()
    for name in ("100prb_2port_harq", "50prb_4port_sib2", "75prb_2port_extcp", "50prb_ul_mode"):
        assert R.digest(ref.run(LIFE[name])) == FIX["lives"][name]["digest"], name
    assert R.digest_rows([tuple(v & 0xFFFFFFFF for v in R.rar_reference(ref, *a)) for a in R.rar_sweep()]) == FIX["rar"]["digest"]
    for name, gaps in (("gapless", False), ("with_gaps_and_restarts", True)):
        assert R.digest_rows(R.ulsche_reference(ref, R.ulsche_script(gaps=gaps))) == FIX["ulsche"][name]["digest"]
    h = ref.lib.ref_collect_new(100, 2, 1, 0, 1, 0, 0)
    assert ref.lib.ref_ulsche_ul_tti(h, 2, 0) == 10238 and ref.lib.ref_ulsche_ul_tti(h, 5, 1) == 10239 and ref.lib.ref_ulsche_ul_tti(h, 600, 0) == 596
    # one grant by hand: format 1A of an unknown C-RNTI on 100 PRB, localized, RIV = 100 * (4 - 1) + 10 -> PRBs 10..13 in both slots; I_MCS 9 -> QPSK, I_TBS 9;
    # subframe 1, CFI 2: 12 symbols x 12 - 3 CRS symbols x 4 REs (2 ports) = 132 REs per PRB
    from lsn_testlib import OCell
    import ctypes as C
    o = oracle()
    o.o_tbs_from_idx.restype = C.c_int
    n = R._sizes(o, 100, 2)[R.FMT1A]
    riv = 100 * (4 - 1) + 10
    bits = [1, 0] + [(riv >> (12 - i)) & 1 for i in range(13)] + [0, 1, 0, 0, 1] + [0, 1, 1] + [1] + [0, 0] + [0, 1]
    bits += [0] * (n - len(bits))
    ref.lib.ref_collect_begin(h, 7, 1, 2)
    ref.lib.ref_collect_add(h, 0x1234, R.FMT1A, 2, 8, 9, (C.c_uint8 * 128)(*bits), n)
    flags, dl, ul, md, mu = R._end(ref.lib.ref_collect_end, h, 100)
    assert flags == 0 and len(dl) == 1 and not ul and md[9:15] == (0, 0x1234, 0x1234, 0x1234, 0x1234, 0)
    r = dl[0]
    assert r[:13] == (0x1234, R.FMT1A, 0, 0x1234, 3, 0, 0, 9, 0, 1, 0, 0, 0) and r[14:17] == (4, 4 * 132, 1) and r[17:21] == r[21:25] == (0xF << 10, 0, 0, 0)
    assert r[25:32] == (1, 2, o.o_tbs_from_idx(9, 4), 4 * 132 * 2, 0, 9, 0) and r[39] == 0   # format 1A is collected under the 64QAM table only
    ref.lib.ref_collect_free(h)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/src"), reason="/root/reference is not here")
def test_fixture_was_made_from_the_reference_sources_that_are_here():
    assert R.reference_sources_sha256() == FIX["reference_sources_sha256"]
