"""CPU tests of the four-CRS-port restatement (36.211 6.10.1.2 ports 2 / 3, 6.2.4 REGs of symbol 1, 6.3.4.3 SFBC-FSTD on PCFICH / PDCCH / PBCH /
PDSCH, 36.212 5.3.3.1 four-port DCI sizes): the synthetic transmitter (tools/txgen) -> oracle worker loop-back.  The reference hands any
srsran_cell_t to srsRAN (/root/reference/src/src/SubframeWorker.cc:102), whose four-port support is transmit diversity; grants that ask for
spatial multiplexing are found on the PDCCH and not decoded (dl_sniffer_pdsch.c:134-276 has no four-port branch of its own)."""
import ctypes as C

import numpy as np
import pytest

from lsn_testlib import OCell, oracle, scenario
from parity import gen_subframes, run_oracle


def _sent(truth):
    return [p for t in truth for p in t if not p["is_ul"]]


@pytest.mark.parametrize("scn,over", [("small", {}), ("small", dict(nof_prb=100, n_rnti=8, dl_min=3, dl_max=5, cfi=0, rar_period=20, paging_period=16)),
                                      ("cfg2", dict(nof_prb=50, n_rnti=6, cell_id=329, phich_ng_x6=6)), ("small", dict(nof_prb=6, cfi=0, n_rnti=2, dl_min=1, dl_max=1)),
                                      ("small", dict(nof_rx=1, cell_id=5))])
def test_four_port_loopback_decodes_what_was_sent(scn, over):
    sc = scenario(scn, seed=5, nof_ports=4, **over)
    tti0, iq, truth = gen_subframes(sc, 60)
    ow, per_sf, recs = run_oracle(sc, tti0, iq)
    sent = _sent(truth)
    got = set(r["pdu"] for r in recs)
    hit = sum(1 for p in sent if p["payload"] in got)
    assert len(sent) >= 40 and hit >= 0.9 * len(sent), (len(sent), hit, len(recs))  # the first subframes go by before the RNTI histogram accepts the UEs
    # every port is estimated: per-port noise far below the pilot power at 30 dB
    ch = per_sf[-1]["chest"]
    A = sc["nof_rx"]
    noise, cep = ch[0:8].reshape(2, 4)[:A], ch[16:24].reshape(2, 4)[:A]
    assert (cep > 0.2).all() and (noise < cep * 0.02).all(), (noise, cep)


def test_four_port_dci_sizes_and_reg_count():
    o = oracle()
    # 36.212 5.3.3.1.5 / 5.3.3.1.5A: precoding information 3 / 0 bits with two ports, 6 / 2 bits with four; written out by hand: type-0 header 1 +
    # RBG bitmap 25 / 17 / 13 + TPC 2 + HARQ 3 + swap 1 + 2 x (MCS 5 + NDI 1 + RV 2) + precoding; 40 bits is an ambiguous size (-> 41)
    for nprb, f2_2, f2_4, f2a_2, f2a_4 in ((100, 51, 54, 48, 50), (50, 43, 46, 41, 42), (25, 39, 42, 36, 38)):
        c2, c4 = OCell(nprb, 2, 1, 1), OCell(nprb, 4, 1, 1)
        assert (o.o_dci_format_sizeof(C.byref(c2), 6), o.o_dci_format_sizeof(C.byref(c4), 6)) == (f2_2, f2_4)
        assert (o.o_dci_format_sizeof(C.byref(c2), 7), o.o_dci_format_sizeof(C.byref(c4), 7)) == (f2a_2, f2a_4)


def test_spatial_multiplexing_grants_on_four_ports_are_found_and_not_decoded():
    sc = scenario("cfg3", seed=9, nof_ports=4, nof_prb=50, n_rnti=12, dl_min=4, dl_max=6, mix_tm3_pct=50, mix_tm4_pct=50, rar_period=0, paging_period=0)
    tti0, iq, truth = gen_subframes(sc, 80)
    ow, per_sf, recs = run_oracle(sc, tti0, iq)
    sent = _sent(truth)
    two_tb = {p["rnti"] for p in sent if p["format"] >= 6}
    assert len(two_tb) >= 8
    got_rnti = {r["rnti"] for r in recs}
    assert not (got_rnti & two_tb)                      # no record for a format 2 / 2A grant
    acc = [a for sf in per_sf for a in sf["accepted"]]    # (rnti, format, L, ncce, nof_bits, histogram value)
    sizes = {a[1]: a[4] for a in acc if a[0] in two_tb and a[1] >= 6}  # (the same UEs also get uplink grants, format 0)
    assert sizes == {6: 46, 7: 42}, sizes  # the search accepts them at their four-port sizes
