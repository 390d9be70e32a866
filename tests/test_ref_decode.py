"""SURVEY 8 row a13 pinned on the REFERENCE'S OWN CODE: PDSCH_Decoder::decode_dl_mode (/root/reference/src/src/DL_Sniffer_PDSCH.cc:881-1291 - the gate, the known-table
decode or the 64QAM-then-256QAM trial, the HARQ verdicts per transport block, which blocks become pcap records, random-access responses into the RNTI manager and the
tracking database, RRCConnectionSetups into the UE-configuration database, the MCS-table learning and the statistics that age it) compiled verbatim into
oracle/_ref/libref_falcon_decode.so on top of the reference's own DCICollection / falcon_dci.c / MCSTracking / HARQ / RNTIManager (oracle/Makefile.ref; stand-in srsRAN
types and L2 / L3 classes whose byte parsers are bound to the oracle's: oracle/ref_shim_search/srsran/standin_l2.h).  The PDSCH decoder itself is srsRAN's and absent:
every call of srsran_ue_dl_decode_pdsch is recorded and answered by a SCRIPTED decoder (tests/ref_decode.py), the same one that answers the oracle's decode calls.
Committed as digests (tests/golden/decode_ref.json, made by tests/golden/make_decode_fixture.py, which also walked the lives at five times the length: 16 000
subframes, 27 000 decode calls, no difference): per subframe every decode call as the reference configured it and every record it handed to the pcap writer, at check
points the tracking database, the activation reasons and the UE configurations.  The oracle's restatement (o_worker.c: decode_dl_mode) must reproduce them; the
product's commit stage is held to the oracle's record streams on the GPU (tests/test_gpu_parity.py).  Where the library is present the reference itself runs again.

What building the harness found: (1) the unknown-table branch tries EVERY SDU of a decoded block as an RRCConnectionSetup, the known-table branch only those on
logical channel 0 (DL_Sniffer_PDSCH.cc:1140 against :1049) - oracle and product learnt from LCID 0 only in both, fixed; (2) the gate reads the 64QAM-table grant
of every entry (:887-888), also of a 256QAM-table entry for which the reference never computed it - uninitialised memory; the oracle gates on the grant it uses.
The two agree unless the first transport block has no size of its own (reserved MCS index or disabled): the lives leave those DCI out, and a test of its own shows
that they are where the runs part."""
import json
import os

import pytest

import ref_decode as D

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decode_ref.json")))
HAVE_LIB = os.path.exists(D.REF_SO)
LIFE = {l[0]: l for l in D.LIVES}


def _strip(results):
    return [r for r in results if r[0] != "harq"]


def test_fixture_is_whole():
    assert set(FIX["lives"]) == set(LIFE) and all(FIX["oracle_equal_when_made"].values())
    tot = {k: sum(v[k] for v in FIX["lives"].values()) for k in ("decode_calls", "two_block_calls", "blocks_passed", "records", "rar_activated_at_the_end", "ue_configs_at_the_end")}
    assert tot["decode_calls"] > 5000 and tot["records"] > 3000 and all(v > 0 for v in tot.values()), tot
    assert all(sum(v["calls_by_modulation"][q] for v in FIX["lives"].values()) > 100 for q in range(4))      # QPSK ... 256QAM
    assert all(sum(v["calls_by_tx_scheme"][t] for v in FIX["lives"].values()) > 100 for t in range(4))        # single port, diversity, spatial multiplexing, CDD
    assert all(sum(v["records_by_kind"][k] for v in FIX["lives"].values()) > 50 for k in range(4))            # C-RNTI, RA-RNTI, SI-RNTI, paging
    assert all(sum(v["tables_at_the_end"][t] for v in FIX["lives"].values()) > 0 for t in (0, 1, 2))          # UEs on the 64QAM table, the 256QAM table, undecided
    hq = FIX["lives"]["100prb_2rx_harq_64qam_table"]["harq_verdicts_new_retx_full_decoded_busy"]
    assert hq[0] > 300 and hq[1] > 50 and hq[3] > 50                                                          # new transmissions, retransmissions, already decoded
    assert len({p for v in FIX["lives"].values() for p in v["p_a_values"]}) >= 2                              # a connection setup changed some UE's p_a
    lr = FIX["long_run"]
    assert sum(v["decode_calls"] for v in lr.values()) > 20_000 and all(v["oracle_differs_in"] == 0 for v in lr.values())


@pytest.mark.parametrize("name", sorted(LIFE))
def test_oracle_decodes_like_the_reference(name):
    assert D.digest(_strip(D.run(D.Oracle(), LIFE[name]))) == FIX["lives"][name]["digest"]


@pytest.mark.skipif(not HAVE_LIB, reason="oracle/_ref/libref_falcon_decode.so not built (needs /root/reference: make -C oracle -f Makefile.ref)")
def test_reference_library_reproduces_the_committed_fixture():
    ref = D.Reference()
    for name in ("100prb_2rx", "100prb_2rx_harq_64qam_table", "50prb_1rx", "15prb_4port"):
        assert D.digest(D.run(ref, LIFE[name])) == FIX["lives"][name]["digest"], name


@pytest.mark.skipif(not HAVE_LIB, reason="oracle/_ref/libref_falcon_decode.so not built")
def test_where_the_gate_reads_a_grant_the_reference_did_not_compute():
    """With first blocks that have no size of their own in the script, reference (unwritten memory filled with 0x01: "a positive size") and oracle part exactly in
    subframes that hold a 256QAM-table entry of that kind: the reference decodes the second block of such a grant, the oracle skips the grant (as both do for a
    64QAM-table entry, where the reference reads the real size 0)."""
    life = D.LIVES[0]
    a, b = D.run(D.Reference(), life, True), _strip(D.run(D.Oracle(), life, True))
    diff = [i for i, (x, y) in enumerate(zip(a, b)) if x != y]
    assert diff and diff[0] == FIX["reserved_first_block"]["first_differing"] and len(diff) == FIX["reserved_first_block"]["differing"]
    x, y = a[diff[0]], b[diff[0]]
    extra = [c for c in x[0] if c not in y[0]]
    assert extra and all(c[0][8] == 0 and c[0][11] == 1 and c[0][13] > 0 for c in extra)   # first block without a size, second block enabled with one
    assert [c for c in y[0] if c not in x[0]] == []


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/src"), reason="/root/reference is not here")
def test_fixture_was_made_from_the_reference_sources_that_are_here():
    assert D.reference_sources_sha256() == FIX["reference_sources_sha256"]
