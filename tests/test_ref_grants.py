"""SURVEY 8 rows a11 / a12 / a15 pinned on the REFERENCE'S OWN CODE: /root/reference/lib/src/phy/falcon_phch/ul_sniffer_pusch.c (DCI 0 -> PUSCH grant: resource
indication value -> PRBs, type-1 hopping with the SIB2 offset, Tables 8.6.1-1 and 8.6.1-3 of TS 36.213 with the reference's row 32A) and dl_sniffer_pdsch.c
(dl_sniffer_config_mimo, transport-block enabling, transport-block sizes of SI / P / RA-RNTI grants in formats 1A / 1C) compiled verbatim into
oracle/_ref/libref_falcon_grant.so (oracle/Makefile.ref; stand-in srsRAN types, the oracle's TBS table bound in: grant_glue.c).  Their answers to the sweeps of
tests/ref_grants.py are committed as digests (tests/golden/grants_ref.json, made by tests/golden/make_grant_fixture.py, which also walked the full uplink sweep of
2.5 million grants next to the oracle: no difference); the oracle's restatement (o_dci.c) must give the same answers.  The product's conversions are held to the
oracle's on random DCI payloads in tests/test_host_logic.py.  Where the library is present the reference itself runs again."""
import json
import os

import pytest

import ref_grants as G
from lsn_testlib import oracle

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "grants_ref.json")))
HAVE_LIB = os.path.exists(G.REF_SO)


def _thin(side):
    return [G.normalise_ul(side.ul(*a), a[6]) for i, a in enumerate(G.ul_sweep()) if i % G.SUITE_STRIDE == 0]


def test_fixture_is_whole():
    f = FIX["ul"]["full_sweep"]
    assert f["cases"] > 2_000_000 and f["oracle_differs_in"] == 0 and f["type_1_hopping_grants"] > 100_000 and 0 < f["accepted_by_the_reference"] < f["cases"]
    assert all(FIX["oracle_equal_when_made"].values()) and FIX["ul"]["suite_stride"] == G.SUITE_STRIDE
    assert FIX["mimo"]["ok"] > 0 and all(n > 0 for n in FIX["mimo"]["by_error"])      # every error class of dl_sniffer_config_mimo is reached
    assert {40, 1736} <= set(FIX["tb_common"]["sizes_seen"])                          # first and last entry of the reference's format-1C table


def test_oracle_uplink_grants_are_the_references():
    rows = _thin(G.Oracle())
    assert len(rows) == FIX["ul"]["suite_cases"] and sum(r is not None for r in rows) == FIX["ul"]["suite_accepted"]
    assert G.digest(rows) == FIX["ul"]["suite_digest"]


def test_oracle_mimo_configuration_is_the_references():
    o = oracle()
    assert G.digest([G.oracle_mimo(o, *a) for a in G.mimo_sweep()]) == FIX["mimo"]["digest"]


def test_oracle_common_rnti_transport_blocks_are_the_references():
    o = oracle()
    assert G.digest([G.oracle_tb_common(o, *a) for a in G.tb_common_sweep()]) == FIX["tb_common"]["digest"]


@pytest.mark.skipif(not HAVE_LIB, reason="oracle/_ref/libref_falcon_grant.so not built (needs /root/reference: make -C oracle -f Makefile.ref)")
def test_reference_library_reproduces_the_committed_fixture():
    r = G.Reference()
    assert G.digest(_thin(r)) == FIX["ul"]["suite_digest"]
    assert G.digest([r.mimo(*a) for a in G.mimo_sweep()]) == FIX["mimo"]["digest"]
    assert G.digest([r.tb_common(*a) for a in G.tb_common_sweep()]) == FIX["tb_common"]["digest"]
    # a hand-computed grant: 100 PRB, offset 10 -> n_rb_pusch 90; +N/2 hopping, 4 PRB at 12 -> slot 1 at (45 + 12) % 90 = 57; I_MCS 20 -> 16QAM, I_TBS 19
    o = oracle()
    o.o_tbs_from_idx.restype = int
    assert r.ul(100, 0, 10, 0, 100 * 3 + 12, 2, 20, 0) == (4, 12, 57, 1, 4, o.o_tbs_from_idx(19, 4), 0, 12 * 4 * 12)


def test_fixture_was_made_from_the_reference_sources_that_are_here():
    if not os.path.isdir("/root/reference/lib/src"):
        pytest.skip("no /root/reference on this host")
    assert G.reference_sources_sha256() == FIX["reference_sources_sha256"]
