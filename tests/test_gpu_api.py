"""GPU test of the security-API sink (lsn_phy_set_api_mode): identities reported for the decoded downlink blocks and the API pcap are the
oracle worker's (run_api_dl_mode, DL_Sniffer_PDSCH.cc:804-879), in record order, across chunk boundaries."""
import ctypes as C

import numpy as np
import pytest

import ltesniffer_amd as la
from lsn_testlib import OracleWorker, encode_paging, oracle_worker_api_events, oracle_worker_set_api, parse_pcap, scenario
from parity import gen_subframes, gpu_records, oracle_records

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("api_mode", [3, 2, 0])
def test_api_events_and_api_pcap_match_oracle(api_mode):
    paging = encode_paging([("imsi", "262019876543210"), ("tmsi", 0x21, 0xC0FFEE42), ("imsi", "001010123456")])
    sc = scenario("small", seed=8, paging_period=8, msg4_period=7, msg4_p_a_idx=4)
    n = 80
    tti0, iq, _ = gen_subframes(sc, n, paging_msg=paging)
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"], sc["phich_ng_x6"])
    oracle_worker_set_api(ow, api_mode)
    for i in range(n):
        ow.work(iq[i], tti0 + i)
    oev = oracle_worker_api_events(ow)
    nb = C.c_size_t()
    oapi = parse_pcap(C.string_at(ow.lib.o_pcap_mem(C.c_void_p(ow.api_pcap), C.byref(nb)), nb.value))
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=16, pcapwriter=la.PcapWriter(None))
    api_pcap = la.PcapWriter(None)
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"]) and phy.setApiMode(api_mode, api_pcap)
    phy.process_host(iq, tti0)
    assert gpu_records(phy) == oracle_records(parse_pcap(ow.pcap_bytes()))
    assert phy.api_events == oev
    assert oracle_records(parse_pcap(api_pcap.bytes())) == oracle_records(oapi)
    if api_mode == 3:
        assert {e[3] for e in oev} == {1, 5} and len(oapi) >= 6
    if api_mode == 2:
        assert {e[3] for e in oev} == {5}
    if api_mode == 0:
        assert {e[3] for e in oev} == {1}
    phy.close()


def test_api_events_do_not_need_a_pdu_sink():
    """round-2 advisor finding: lsn_phy_set_api_mode without a PDU sink / pcap writer silently produced nothing.  The identities and the API pcap
    must be the same with the sink removed"""
    paging = encode_paging([("imsi", "262019876543210"), ("tmsi", 0x21, 0xC0FFEE42)])
    sc = scenario("small", seed=9, paging_period=8, msg4_period=7, msg4_p_a_idx=4)
    n = 48
    tti0, iq, _ = gen_subframes(sc, n, paging_msg=paging)
    results = []
    for with_sink in (True, False):
        phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=16, pcapwriter=la.PcapWriter(None))
        api_pcap = la.PcapWriter(None)
        assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"]) and phy.setApiMode(3, api_pcap)
        if not with_sink:
            assert la.lib().lsn_phy_set_pcap_writer(phy._h, None) == 0   # neither pcap writer nor callback
        phy.process_host(iq, tti0)
        results.append((list(phy.api_events), api_pcap.bytes()))
        phy.close()
    assert results[0] == results[1] and len(results[0][0]) >= 4
