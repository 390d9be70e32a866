"""DL HARQ soft combining, SURVEY 8(f) row 4 (HARQ.cc:71-190, DL_Sniffer_PDSCH.cc:943-1020; off and unreachable in the reference's CLI,
ArgManager.cc:50,211-213): the oracle's restatement on a stream whose eNB repeats 60 % of its C-RNTI grants 8 subframes later (same HARQ process,
NDI not toggled, next redundancy version, same transport blocks) at an SNR where many first transmissions fail."""
import numpy as np

from lsn_testlib import OracleWorker, parse_pcap, scenario
from parity import gen_subframes


def _run(sc, tti0, iq, harq):
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"], sc["phich_ng_x6"], mcs_tracking_mode=0)
    if harq:
        ow.set_harq(1)
    for i in range(iq.shape[0]):
        ow.work(iq[i], tti0 + i)
    recs = [r for r in parse_pcap(ow.pcap_bytes()) if r["rnti_type"] == 3 and r["direction"] == 1]
    return recs, ow.harq_stats()


def test_retransmissions_are_combined_and_decoded_blocks_are_not_decoded_again():
    sc = scenario("small", seed=93, n_rnti=3, dl_min=2, dl_max=2, ul_min=0, ul_max=0, mcs_min=18, mcs_max=22, snr_db=11.0, pct_harq=60)
    tti0, iq, truth = gen_subframes(sc, 90)
    sent = {}
    for i, pdus in enumerate(truth):
        for p in pdus:
            if not p["is_ul"] and 0x000B <= p["rnti"] <= 0xFFF3:
                sent.setdefault((p["rnti"], p["payload"]), []).append((tti0 + i) % 10240)
    assert any(len(v) > 1 and v[1] - v[0] == 8 for v in sent.values()), "the transmitter must repeat transport blocks 8 subframes later"
    off, st_off = _run(sc, tti0, iq, 0)
    on, st_on = _run(sc, tti0, iq, 1)
    assert st_off == [0, 0, 0, 0, 0]
    new_tx, re_tx, full, decoded, busy = st_on
    assert new_tx > 20 and re_tx > 5 and decoded > 5 and full == 0 and busy == 0, st_on
    # every record is a transport block the eNB sent to that RNTI
    for r in on + off:
        assert (r["rnti"], r["pdu"]) in sent
    d_off, d_on = {(r["rnti"], r["pdu"]) for r in off}, {(r["rnti"], r["pdu"]) for r in on}
    # soft combining recovers blocks that no single transmission decodes ...
    assert len(d_on - d_off) >= 3, (len(d_on), len(d_off))
    # ... and a block that was decoded is not decoded (and written) again when its retransmission arrives: fewer duplicate records than without HARQ
    # (a retransmission whose first transmission was not seen - missed DCI - still counts as new)
    assert len(on) - len(d_on) < len(off) - len(d_off) and len(off) - len(d_off) >= 10, (len(on), len(d_on), len(off), len(d_off))
