"""CPU tests of the security-API view of decoded downlink blocks (PDSCH_Decoder::run_api_dl_mode,
DL_Sniffer_PDSCH.cc:804-879, decode_imsi_tmsi_paging :84-127): (a) the connection-setup side against the reference's own
api_collector.pcap (tests/golden/pcap_records.json): every recorded contention-resolution message yields one event whose value is the
identity the matching uplink message 3 carries; (b) the PCCH walk (oracle o_rrc.c and the product's lsn_rrc.cc through tests/native)
against an independent UPER encoder, truncated and random input; (c) the oracle worker with -a 3 on a synthetic cell."""
import json
import os

import numpy as np
import pytest

from lsn_testlib import (OracleWorker, TxGen, encode_paging, host_api_events, host_api_ul_msg3, host_paging_decode, oracle_api_events, oracle_api_ul_msg3, oracle_paging_decode,
                         oracle_worker_api_events, oracle_worker_set_api, parse_pcap, scenario)

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pcap_records.json")))


def test_recorded_connection_setups_report_the_contention_resolution_identity():
    setups = FIX["api_collector.pcap"]["conn_setup"]
    assert len(setups) == 5
    for m in setups:
        pdu = bytes.fromhex(m["pdu"])
        assert pdu[0] == 0x3C  # contention resolution identity first
        ident = pdu[4:10].hex().lstrip("0")
        for api_mode in (0, 3):
            ev, keep = oracle_api_events(api_mode, "C", pdu, m["rnti"], 1234)
            assert keep and ev == [(1234, m["rnti"], 2, 1, ident[3:11])]
            assert host_api_events(api_mode, "C", pdu, m["rnti"], 1234) == (ev, keep)
        for api_mode in (-1, 1, 2):
            assert oracle_api_events(api_mode, "C", pdu, m["rnti"], 1) == ([], False) == host_api_events(api_mode, "C", pdu, m["rnti"], 1)
    # the reference wrote exactly one API record per connection setup into api_collector.pcap
    assert FIX["api_collector.pcap"]["pdu_lengths"]["1/3"] == [32]
    # an ordinary downlink block reports nothing
    assert oracle_api_events(3, "C", bytes([0x03]) + bytes(40), 70, 5) == ([], False) == host_api_events(3, "C", bytes([0x03]) + bytes(40), 70, 5)


def test_recorded_msg3_blocks_report_the_identity_the_connection_setup_echoes():
    """identity mapping of -a 0 / 3: the random value of the RRCConnectionRequest (uplink) and the contention resolution identity of the
    RRCConnectionSetup (downlink) are printed as the same eight hex characters - on the reference's own capture"""
    msg3 = FIX["api_collector.pcap"]["msg3"]
    setups = {m["rnti"]: bytes.fromhex(m["pdu"]) for m in FIX["api_collector.pcap"]["conn_setup"]}
    assert len(msg3) == 5
    for m in msg3:
        pdu = bytes.fromhex(m["pdu"])
        ev, keep = oracle_api_ul_msg3(3, pdu, m["rnti"], 77)
        assert keep and len(ev) == 1 and ev[0][:4] == (77, m["rnti"], 0, 0)  # ID_RAN_VAL, MSG_CON_REQ
        dl, _ = oracle_api_events(3, "C", setups[m["rnti"]], m["rnti"], 83)
        assert ev[0][4] == dl[0][4] == pdu[1:7].hex()[3:11]
        assert host_api_ul_msg3(3, pdu, m["rnti"], 77) == (ev, keep) == host_api_ul_msg3(0, pdu, m["rnti"], 77)
        assert oracle_api_ul_msg3(2, pdu, m["rnti"], 77) == ([], False) == host_api_ul_msg3(2, pdu, m["rnti"], 77)
    # an s-TMSI request reports the m-TMSI; truncated / random input: both walks agree
    req = bytes([0x00]) + int("0" "1" "0" "0" + "{:08b}".format(0x5A) + "{:032b}".format(0x00C0FFEE) + "011" "0", 2).to_bytes(6, "big")
    assert oracle_api_ul_msg3(0, req, 9, 1) == ([(1, 9, 1, 0, "c0ffee")], True) == host_api_ul_msg3(0, req, 9, 1)
    rng = np.random.RandomState(5)
    for i in range(3000):
        m = bytes(rng.randint(0, 256, size=int(rng.randint(1, 24))).astype(np.uint8))
        if i % 2:
            m = bytes([m[0] & 0x3F]) + m[1:]
        assert oracle_api_ul_msg3(3, m, 5, 2) == host_api_ul_msg3(3, m, 5, 2)


def _records(rng):
    n = int(rng.randint(1, 17))
    out = []
    for _ in range(n):
        if rng.randint(2):
            out.append(("imsi", "".join(str(int(d)) for d in rng.randint(0, 10, size=int(rng.randint(6, 22))))))
        else:
            out.append(("tmsi", int(rng.randint(256)), int(rng.randint(0, 1 << 32, dtype=np.uint64))))
    return out


def test_paging_round_trip_against_an_independent_encoder():
    rng = np.random.RandomState(11)
    assert oracle_paging_decode(encode_paging([])) == [] == host_paging_decode(encode_paging([], sys_info_mod=1))
    for i in range(300):
        recs = _records(rng)
        msg = encode_paging(recs, sys_info_mod=int(rng.randint(2)), etws=int(rng.randint(2)), ext_record=int(rng.randint(len(recs))) if i % 3 == 0 else None)
        assert oracle_paging_decode(msg) == recs
        assert host_paging_decode(msg) == recs
        ev, keep = oracle_api_events(2, "P", msg, 0xFFFE, 77)
        assert keep and len(ev) == len(recs)
        for e, r in zip(ev, recs):
            if r[0] == "imsi":
                assert e == (77, 65534, 3, 5, r[1][:15])
            else:
                assert e == (77, 65534, 1, 5, "%08x" % r[2])
        assert host_api_events(2, "P", msg, 0xFFFE, 77) == (ev, keep)
        assert oracle_api_events(0, "P", msg, 0xFFFE, 77) == ([], False)


def test_truncated_and_random_pcch_is_rejected_identically():
    rng = np.random.RandomState(12)
    for i in range(60):
        msg = encode_paging(_records(rng))
        for cut in range(len(msg)):
            assert oracle_paging_decode(msg[:cut]) == host_paging_decode(msg[:cut])
        b = bytearray(msg)
        for _ in range(3):
            k = int(rng.randint(8 * len(b)))
            b[k >> 3] ^= 0x80 >> (k & 7)
        assert oracle_paging_decode(bytes(b)) == host_paging_decode(bytes(b))
    ok = 0
    for i in range(3000):
        m = bytes(rng.randint(0, 256, size=int(rng.randint(1, 50))).astype(np.uint8))
        a, b = oracle_paging_decode(m), host_paging_decode(m)
        assert a == b
        ok += a is not None
        assert oracle_api_events(3, "P", m, 0xFFFE, 3) == host_api_events(3, "P", m, 0xFFFE, 3)
        assert oracle_api_events(3, "C", m, 0x46, 3) == host_api_events(3, "C", m, 0x46, 3)
    assert ok > 50


def test_oracle_worker_reports_paging_and_connection_setups():
    paging = encode_paging([("imsi", "262019876543210"), ("tmsi", 0x21, 0xC0FFEE42)])
    sc = scenario("small", seed=8, paging_period=8, msg4_period=7, msg4_p_a_idx=4)
    tx = TxGen(paging_msg=paging, **sc)
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"], sc["phich_ng_x6"])
    oracle_worker_set_api(ow, 3)
    n = 60
    for i in range(n):
        tti, iq, _ = tx.next()
        ow.work(iq, tti)
    ev = oracle_worker_api_events(ow)
    pag = [e for e in ev if e[3] == 5]
    con = [e for e in ev if e[3] == 1]
    assert len(pag) >= 8 and len(pag) % 2 == 0 and len(con) >= 3
    assert {e[4] for e in pag} == {"262019876543210", "c0ffee42"} and all(e[1] == 65534 for e in pag)
    assert all(e[2] == 2 and len(e[4]) == 8 for e in con)
    recs = parse_pcap(ow.pcap_bytes())
    import ctypes as C
    nb = C.c_size_t()
    api = parse_pcap(C.string_at(ow.lib.o_pcap_mem(C.c_void_p(ow.api_pcap), C.byref(nb)), nb.value))
    assert len(api) == len(pag) // 2 + len(con)
    main = [r["ctx"] + r["pdu"] for r in recs]
    assert all(r["ctx"] + r["pdu"] in main for r in api)
    assert {r["rnti_type"] for r in api} == {1, 3}
