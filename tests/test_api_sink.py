"""CPU tests of the security-API view of decoded downlink blocks (PDSCH_Decoder::run_api_dl_mode,
DL_Sniffer_PDSCH.cc:804-879, decode_imsi_tmsi_paging :84-127): (a) the connection-setup side against the reference's own
api_collector.pcap (tests/golden/pcap_records.json): every recorded contention-resolution message yields one event whose value is the
identity the matching uplink message 3 carries; (b) the PCCH walk (oracle o_rrc.c and the product's lsn_rrc.cc through tests/native)
against an independent UPER encoder, truncated and random input; (c) the oracle worker with -a 3 on a synthetic cell."""
import json
import os

import numpy as np
import pytest

from lsn_testlib import (OracleWorker, TxGen, encode_paging, host_api_events, host_api_ul_dcch, host_api_ul_msg3, host_paging_decode, oracle_api_events, oracle_api_ul_dcch, oracle_api_ul_msg3, oracle_paging_decode,
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


def _bits_to_bytes(bits):
    while len(bits) % 8:
        bits.append(0)
    return bytes(int("".join(map(str, bits[i:i + 8])), 2) for i in range(0, len(bits), 8))


def _mobile_id(type_, digits):
    """TS 24.008 10.5.1.4 / TS 24.301 9.9.3.12 value part for a digit string identity (1 IMSI, 2 / 3 IMEI(SV))"""
    d = [int(c) for c in digits]
    odd = len(d) % 2
    out = [(d[0] << 4) | (odd << 3) | type_]
    rest = d[1:] + ([15] if not odd else [])
    out += [rest[i] | (rest[i + 1] << 4) for i in range(0, len(rest), 2)]
    return bytes(out)


def _nas(msg_type, body, protected, rng):
    plain = bytes([0x07, msg_type]) + body
    return (bytes([0x17]) + bytes(rng.randint(0, 256, size=4).astype(np.uint8)) + bytes([int(rng.randint(256))]) + plain) if protected else plain


def _ul_dcch_block(rrc_bits, nas, rng, sn=None):
    """MAC UL-SCH PDU: short BSR + one SRB1 SDU (RLC AM data PDU, whole SDU) + padding; rrc_bits = UL-DCCH head up to the NAS container"""
    bits = list(rrc_bits)
    n = len(nas)
    bits += [int(c) for c in ("{:08b}".format(n) if n < 128 else "10" + "{:014b}".format(n))]
    for b in nas:
        bits += [int(c) for c in "{:08b}".format(b)]
    rrc = _bits_to_bytes(bits)
    sn = int(rng.randint(1024)) if sn is None else sn
    sdu = bytes([0x80 | (int(rng.randint(2)) << 5) | (sn >> 8), sn & 0xFF]) + bytes([int(rng.randint(32))]) + rrc
    hdr = bytes([0x3D, 0x21]) + (bytes([len(sdu)]) if len(sdu) < 128 else bytes([0x80 | (len(sdu) >> 8), len(sdu) & 0xFF])) + bytes([0x1F])
    return hdr + bytes([int(rng.randint(64))]) + sdu + bytes(int(rng.randint(0, 9)))


def _setup_complete_head(rng, mme):
    bits = [0, 0, 1, 0, 0] + [int(rng.randint(2)), int(rng.randint(2))] + [0] + [0, 0] + [mme, 0] + [int(c) for c in "{:03b}".format(int(rng.randint(6)))]
    if mme:
        plmn = int(rng.randint(2))
        bits += [plmn]
        if plmn:
            mcc = int(rng.randint(2))
            bits += [mcc]
            if mcc:
                for _ in range(3):
                    bits += [int(c) for c in "{:04b}".format(int(rng.randint(10)))]
            nd = int(rng.randint(2, 4))
            bits += [nd - 2]
            for _ in range(nd):
                bits += [int(c) for c in "{:04b}".format(int(rng.randint(10)))]
        bits += [int(rng.randint(2)) for _ in range(24)]
    return bits


def test_recorded_uplink_srb_blocks_are_what_the_reference_accepted():
    """api_collector.pcap holds the blocks whose API parse succeeded in the reference: five RRCConnectionSetupComplete + attach request
    (GUTI, consecutive m-TMSIs of one test network) and five UECapabilityInformation"""
    blocks = FIX["api_collector.pcap"]["ul_dcch"]
    assert sorted(len(b["pdu"]) // 2 for b in blocks) == [333] * 5 + [533] * 5
    tmsi = []
    for m in blocks:
        pdu = bytes.fromhex(m["pdu"])
        ev, keep = oracle_api_ul_dcch(3, pdu, m["rnti"], 9)
        assert keep and len(ev) == 1 and host_api_ul_dcch(3, pdu, m["rnti"], 9) == (ev, keep)
        if len(pdu) == 333:
            assert ev[0][:4] == (9, m["rnti"], 1, 2)                      # ID_TMSI, MSG_ATT_REQ
            tmsi.append(int(ev[0][4], 16))
            assert oracle_api_ul_dcch(2, pdu, m["rnti"], 9) == (ev, True) and oracle_api_ul_dcch(1, pdu, m["rnti"], 9) == ([], False)
        else:
            assert ev[0] == (9, m["rnti"], 0xFFFFFFFF, 4, "-")             # print_api(..., -1, "-", MSG_UE_CAP)
            assert oracle_api_ul_dcch(1, pdu, m["rnti"], 9) == (ev, True) and oracle_api_ul_dcch(2, pdu, m["rnti"], 9) == ([], False)
        assert oracle_api_ul_dcch(0, pdu, m["rnti"], 9) == ([], False) == host_api_ul_dcch(-1, pdu, m["rnti"], 9)
    assert tmsi == list(range(tmsi[0], tmsi[0] + 5))


def test_uplink_identities_round_trip():
    rng = np.random.RandomState(21)
    kinds = 0
    for i in range(400):
        imsi = "".join(str(int(d)) for d in rng.randint(0, 10, size=15))
        imei = "".join(str(int(d)) for d in rng.randint(0, 10, size=15))
        imeisv = "".join(str(int(d)) for d in rng.randint(0, 10, size=16))
        m_tmsi = int(rng.randint(1, 1 << 32, dtype=np.uint64))
        guti = bytes([0xF6]) + bytes(rng.randint(0, 256, size=6).astype(np.uint8)) + m_tmsi.to_bytes(4, "big")
        k = i % 7
        prot = int(rng.randint(2))
        if k < 4:  # attach request in an RRCConnectionSetupComplete
            ident, want = [(_mobile_id(1, imsi), (3, 2, imsi)), (guti, (1, 2, "%x" % m_tmsi)), (_mobile_id(3, imei), (4, 2, imei)),
                           (bytes([0xF0]) + bytes(4), None)][k]
            nas = _nas(0x41, bytes([0x71, len(ident)]) + ident + bytes(rng.randint(0, 256, size=int(rng.randint(0, 30))).astype(np.uint8)), prot, rng)
            head = _setup_complete_head(rng, int(rng.randint(2)))
        else:      # identity response in an ULInformationTransfer
            ident, want = [(_mobile_id(1, imsi), (3, 3, imsi)), (_mobile_id(2, imei), (4, 3, imei)), (_mobile_id(3, imeisv), (5, 3, imeisv))][k - 4]
            nas = _nas(0x56, bytes([len(ident)]) + ident, prot, rng)
            head = [0, 1, 0, 0, 1] + [0] + [0, 0] + [0] + [0, 0]
        pdu = _ul_dcch_block(head, nas, rng)
        ev, keep = oracle_api_ul_dcch(3, pdu, 61, 4)
        assert (ev, keep) == (([(4, 61, want[0], want[1], want[2])], True) if want else ([], False)), (k, prot, pdu.hex(), ev)
        assert host_api_ul_dcch(3, pdu, 61, 4) == (ev, keep) == host_api_ul_dcch(2, pdu, 61, 4)
        assert oracle_api_ul_dcch(1, pdu, 61, 4) == ([], False)
        kinds |= 1 << k
        # a ciphered NAS message, an RLC control PDU, a segment (FI != 0) and a re-segment report nothing
        ciphered = _ul_dcch_block(head, bytes([0x27]) + nas[1:], rng)
        assert oracle_api_ul_dcch(3, ciphered, 61, 4) == ([], False)
        for bad in (ciphered, pdu[:5] + bytes([pdu[5] & 0x7F]) + pdu[6:], pdu[:5] + bytes([pdu[5] | 0x08]) + pdu[6:], pdu[:5] + bytes([pdu[5] | 0x40]) + pdu[6:]):
            assert oracle_api_ul_dcch(3, bad, 61, 4) == host_api_ul_dcch(3, bad, 61, 4)
        assert oracle_api_ul_dcch(3, pdu[:5] + bytes([pdu[5] & 0x7F]) + pdu[6:], 61, 4) == ([], False)
        for cut in range(0, len(pdu), 3):
            assert oracle_api_ul_dcch(3, pdu[:cut], 61, 4) == host_api_ul_dcch(3, pdu[:cut], 61, 4)
    assert kinds == 127
    for i in range(3000):
        m = bytes(rng.randint(0, 256, size=int(rng.randint(1, 60))).astype(np.uint8))
        if i % 2:
            m = bytes([0x21, min(len(m), 50), 0x1F, 0x80, 0x00, 0x00]) + m
        assert oracle_api_ul_dcch(3, m, 5, 2) == host_api_ul_dcch(3, m, 5, 2)


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


# ---------------------------------------------------------------------------------------------------- RRCConnectionReconfiguration -> attach accept -> GUTI
def _put(bits, v, n):
    bits.extend((int(v) >> (n - 1 - i)) & 1 for i in range(n))


def _put_open(b, octets):
    """open type: general length determinant (one or two octets), then the octets"""
    n = len(octets)
    if n < 128:
        _put(b, n, 8)
    else:
        _put(b, 0x8000 | n, 16)
    for x in octets:
        _put(b, x, 8)


def _put_additions(b, adds):
    """extension additions of a SEQUENCE whose extension bit is set (X.691 19.7-19.9): adds = [bytes | None, ...], at least one present"""
    _put(b, 0, 1); _put(b, len(adds) - 1, 6)
    for x in adds:
        _put(b, int(x is not None), 1)
    for x in adds:
        if x is not None:
            _put_open(b, x)


def _put_irat_object(b, ob):
    """MeasObjectToAddMod with measObjectUTRA / GERAN / CDMA2000 (TS 36.331 6.3.5, release-8 components), written for this test"""
    rat, oid, c = ob[0], ob[1], ob[2]
    adds = ob[3] if len(ob) > 3 else None
    _put(b, oid - 1, 5); _put(b, 0, 1); _put(b, {"utra": 1, "geran": 2, "cdma2000": 3}[rat], 2); _put(b, int(bool(adds)), 1)
    if rat == "utra":
        _put(b, int(c.get("offset") is not None), 1); _put(b, int(bool(c.get("remove"))), 1); _put(b, int(bool(c.get("cells"))), 1); _put(b, int(c.get("cgi") is not None), 1)
        _put(b, c["arfcn"], 14)
        if c.get("offset") is not None:
            _put(b, c["offset"] + 15, 5)
        if c.get("remove"):
            _put(b, len(c["remove"]) - 1, 5)
            for v in c["remove"]:
                _put(b, v - 1, 5)
        if c.get("cells"):
            _put(b, int(c["tdd"]), 1); _put(b, len(c["cells"]) - 1, 5)
            for ci, pci in c["cells"]:
                _put(b, ci - 1, 5); _put(b, pci, 7 if c["tdd"] else 9)
        if c.get("cgi") is not None:
            _put(b, int(c["tdd"]), 1); _put(b, c["cgi"], 7 if c["tdd"] else 9)
    elif rat == "geran":
        _put(b, int(c.get("offset") is not None), 1); _put(b, int(c.get("ncc") is not None), 1); _put(b, int(c.get("cgi") is not None), 1)
        _put(b, c["arfcn"], 10); _put(b, c["band"], 1)
        f = c["following"]
        if f[0] == "list":
            _put(b, 0, 2); _put(b, len(f[1]), 5)
            for v in f[1]:
                _put(b, v, 10)
        elif f[0] == "spaced":
            _put(b, 1, 2); _put(b, f[1] - 1, 3); _put(b, f[2], 5)
        else:
            _put(b, 2, 2); _put(b, len(f[1]) - 1, 4)
            for v in f[1]:
                _put(b, v, 8)
        if c.get("offset") is not None:
            _put(b, c["offset"] + 15, 5)
        if c.get("ncc") is not None:
            _put(b, c["ncc"], 8)
        if c.get("cgi") is not None:
            _put(b, c["cgi"][0], 3); _put(b, c["cgi"][1], 3)
    else:
        _put(b, int(c.get("window") is not None), 1); _put(b, int(c.get("offset") is not None), 1); _put(b, int(bool(c.get("remove"))), 1)
        _put(b, int(bool(c.get("cells"))), 1); _put(b, int(c.get("cgi") is not None), 1)
        _put(b, c["type"], 1); _put(b, 0, 1); _put(b, c["band"], 5); _put(b, c["arfcn"], 11)
        if c.get("window") is not None:
            _put(b, c["window"], 4)
        if c.get("offset") is not None:
            _put(b, c["offset"] + 15, 5)
        if c.get("remove"):
            _put(b, len(c["remove"]) - 1, 5)
            for v in c["remove"]:
                _put(b, v - 1, 5)
        if c.get("cells"):
            _put(b, len(c["cells"]) - 1, 5)
            for ci, pci in c["cells"]:
                _put(b, ci - 1, 5); _put(b, pci, 9)
        if c.get("cgi") is not None:
            _put(b, c["cgi"], 9)
    if adds:
        _put_additions(b, adds)


def _encode_reconfig(nas, meas=None, mobility=False, rrcd=True):
    """DL-DCCH-Message { rrcConnectionReconfiguration-r8 } in unaligned PER, written for this test (TS 36.331 6.2.2 / 6.3.5).  meas: None or
    dict(objects=[(id, arfcn, offset or None, cells [(idx, pci, off)], additions or None)], reports=[("a3", offset, adds) | ("a1", rsrp, adds) |
    ("a6", octets, adds) | ("periodical", None, adds)], ids=[(m, o, r)], quantity=(rsrp_fc or None, rsrq_fc or None, adds) or None,
    gap=None | ("gp0", v) | ("gp1", v) | "release", s_measure=None | v, additions=None | [bytes | None, ...]); "additions" are the extension
    additions of later releases, which a release-8 reader has to step over"""
    b = []
    _put(b, 0, 1); _put(b, 4, 4); _put(b, 1, 2); _put(b, 0, 1); _put(b, 0, 3)
    _put(b, 1 if meas is not None else 0, 1); _put(b, int(mobility), 1); _put(b, 1 if nas is not None else 0, 1); _put(b, int(rrcd), 1); _put(b, 0, 1); _put(b, 0, 1)
    if meas is not None:
        _put(b, int(bool(meas.get("additions"))), 1)
        pres = [0, bool(meas.get("objects")), 0, bool(meas.get("reports")), 0, bool(meas.get("ids")), meas.get("quantity") is not None, meas.get("gap") is not None,
                meas.get("s_measure") is not None, meas.get("prereg") is not None, meas.get("speed") is not None]
        for x in pres:
            _put(b, int(bool(x)), 1)
        if meas.get("objects"):
            _put(b, len(meas["objects"]) - 1, 5)
            for ob in meas["objects"]:
                if isinstance(ob[0], str):   # another radio access technology: ("utra" | "geran" | "cdma2000", measObjectId, dict of components, additions)
                    _put_irat_object(b, ob)
                    continue
                oid, arfcn, off, cells = ob[:4]
                adds = ob[4] if len(ob) > 4 else None
                _put(b, oid - 1, 5); _put(b, 0, 1); _put(b, 0, 2); _put(b, int(bool(adds)), 1)
                _put(b, int(off is not None), 1); _put(b, 0, 1); _put(b, int(bool(cells)), 1); _put(b, 0, 3)
                _put(b, arfcn, 16); _put(b, 3, 3); _put(b, 1, 1); _put(b, 1, 2)
                if off is not None:
                    _put(b, off, 5)
                if cells:
                    _put(b, len(cells) - 1, 5)
                    for ci, pci, co in cells:
                        _put(b, ci - 1, 5); _put(b, pci, 9); _put(b, co, 5)
                if adds:
                    _put_additions(b, adds)
        if meas.get("reports"):
            _put(b, len(meas["reports"]) - 1, 5)
            for k, rc in enumerate(meas["reports"]):
                adds = rc[2] if len(rc) > 2 else None
                if rc[0] in ("b1", "b2", "periodical_irat"):   # ReportConfigInterRAT
                    _put(b, k, 5); _put(b, 1, 1); _put(b, int(bool(adds)), 1)
                    if rc[0] == "periodical_irat":
                        _put(b, 1, 1); _put(b, rc[1], 2)
                    else:
                        _put(b, 0, 1); _put(b, 0, 1); _put(b, 0 if rc[0] == "b1" else 1, 1)
                        th = rc[1]
                        if rc[0] == "b2":
                            _put(b, 0, 1); _put(b, th["eutra_rsrp"], 7)
                        _put(b, {"utra": 0, "geran": 1, "cdma2000": 2}[th["rat"]], 2)
                        if th["rat"] == "utra":
                            if "rscp" in th:
                                _put(b, 0, 1); _put(b, th["rscp"] + 5, 7)
                            else:
                                _put(b, 1, 1); _put(b, th["ecn0"], 6)
                        else:
                            _put(b, th["v"], 6)
                        _put(b, 3, 5); _put(b, 5, 4)
                    _put(b, 4, 3); _put(b, 6, 4); _put(b, 7, 3)
                    if adds:
                        _put_additions(b, adds)
                    continue
                _put(b, k, 5); _put(b, 0, 1); _put(b, int(bool(adds)), 1)
                if rc[0] == "periodical":
                    _put(b, 1, 1); _put(b, 0, 1)
                else:
                    _put(b, 0, 1)
                    if rc[0] == "a6":      # eventId is an extensible CHOICE: a6-r10 is extension alternative 0, carried as an open type
                        _put(b, 1, 1); _put(b, 0, 1); _put(b, 0, 6); _put_open(b, rc[1])
                    elif rc[0] == "a3":
                        _put(b, 0, 1); _put(b, 2, 3); _put(b, rc[1] + 30, 6); _put(b, 0, 1)
                    else:
                        _put(b, 0, 1); _put(b, 0, 3); _put(b, 0, 1); _put(b, rc[1], 7)
                    _put(b, 4, 5); _put(b, 8, 4)
                _put(b, 0, 1); _put(b, 1, 1); _put(b, 3, 3); _put(b, 6, 4); _put(b, 7, 3)
                if adds:
                    _put_additions(b, adds)
        if meas.get("ids"):
            _put(b, len(meas["ids"]) - 1, 5)
            for m, o, r in meas["ids"]:
                _put(b, m - 1, 5); _put(b, o - 1, 5); _put(b, r - 1, 5)
        if meas.get("quantity") is not None:
            q = meas["quantity"]
            adds = q[2] if len(q) > 2 else None
            other = q[3] if len(q) > 3 and q[3] else {}   # dict(utra=(fdd quantity, filter or None), geran=(filter or None,), cdma2000=quantity)
            _put(b, int(bool(adds)), 1); _put(b, 1, 1); _put(b, int("utra" in other), 1); _put(b, int("geran" in other), 1); _put(b, int("cdma2000" in other), 1)
            _put(b, int(q[0] is not None), 1); _put(b, int(q[1] is not None), 1)
            for v in q[:2]:
                if v is not None:
                    _put(b, 0, 1); _put(b, v, 4)
            if "utra" in other:
                fq, fc = other["utra"]
                _put(b, int(fc is not None), 1); _put(b, fq, 1)
                if fc is not None:
                    _put(b, 0, 1); _put(b, fc, 4)
            if "geran" in other:
                fc = other["geran"][0]
                _put(b, int(fc is not None), 1)
                if fc is not None:
                    _put(b, 0, 1); _put(b, fc, 4)
            if "cdma2000" in other:
                _put(b, other["cdma2000"], 1)
            if adds:
                _put_additions(b, adds)
        if meas.get("gap") is not None:
            g = meas["gap"]
            if g == "release":
                _put(b, 0, 1)
            else:
                _put(b, 1, 1); _put(b, 0, 1)  # setup; gapOffset is an extensible CHOICE (36.331 MeasGapConfig: gp0, gp1, ...): extension bit 0
                _put(b, 0 if g[0] == "gp0" else 1, 1); _put(b, g[1], 6 if g[0] == "gp0" else 7)
        if meas.get("s_measure") is not None:
            _put(b, meas["s_measure"], 7)
        if meas.get("prereg") is not None:   # (allowed, zone id or None, [secondary zone ids] or None)
            al, z, sec = meas["prereg"]
            _put(b, int(z is not None), 1); _put(b, int(bool(sec)), 1); _put(b, int(al), 1)
            if z is not None:
                _put(b, z, 8)
            if sec:
                _put(b, len(sec) - 1, 1)
                for v in sec:
                    _put(b, v, 8)
        if meas.get("speed") is not None:    # "release" | (t_eval, t_hyst, n_medium, n_high, sf_medium, sf_high)
            sp = meas["speed"]
            if sp == "release":
                _put(b, 0, 1)
            else:
                _put(b, 1, 1); _put(b, sp[0], 3); _put(b, sp[1], 3); _put(b, sp[2] - 1, 4); _put(b, sp[3] - 1, 4); _put(b, sp[4], 2); _put(b, sp[5], 2)
        if meas.get("additions"):
            _put_additions(b, meas["additions"])
    if nas is not None:
        _put(b, 0, 4)
        assert len(nas) < 128
        _put(b, len(nas), 8)
        for x in nas:
            _put(b, x, 8)
    if rrcd:
        _put(b, 0, 1); _put(b, 0, 6)   # RadioResourceConfigDedicated with nothing in it
    while len(b) % 8:
        b.append(0)
    return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))


def _attach_accept(m_tmsi, guti=True, protected=True, esm=b"\x52\x01\xc1\x01\x07"):
    plain = bytes([0x07, 0x42, 0x02, 0x3e, 0x06, 0x00, 0x09, 0xf1, 0x55, 0x00, 0x07]) + len(esm).to_bytes(2, "big") + esm
    if guti:
        plain += bytes([0x50, 0x0b, 0xf6, 0x09, 0xf1, 0x55, 0x00, 0x01, 0x1a]) + m_tmsi.to_bytes(4, "big")
    plain += bytes([0x13, 0x09, 0xf1, 0x55, 0x00, 0x01])
    return (bytes([0x27, 1, 2, 3, 4, 9]) + plain) if protected else plain


def _dcch_pdu(rrc):
    sdu = bytes([0xa0, 0x06, 0x06]) + rrc
    if len(sdu) < 128:
        return bytes([0x21, len(sdu), 0x1f]) + sdu + bytes(3)
    return bytes([0x21, 0x80 | (len(sdu) >> 8), len(sdu) & 0xFF, 0x1f]) + sdu + bytes(3)  # F = 1: 15-bit length


def test_recorded_reconfigurations_report_the_assigned_tmsi():
    """the reference's own captures hold the whole attach of one UE on SRB1; the RRCConnectionReconfiguration that carries the attach accept
    (measConfig + dedicatedInfoNASList + radioResourceConfigDedicated, integrity-protected NAS) yields the M-TMSI of the GUTI - the
    identity print_api_dl reports as (ID_TMSI, MSG_CON_RECONFIG), DL_Sniffer_PDSCH.cc:836-848; the other DL-DCCH messages yield nothing"""
    expect = {"ltesniffer_dl_mode.pcap": "cd5d47ec", "ltesniffer_ul_mode.pcap": None}
    for cap, want in expect.items():
        blocks = FIX[cap]["dl_dcch"]
        assert len(blocks) >= 7
        hits = []
        for m in blocks:
            pdu = bytes.fromhex(m["pdu"])
            for api_mode in (0, 3):
                ev, keep = oracle_api_events(api_mode, "C", pdu, m["rnti"], m["tti"])
                assert (ev, keep) == host_api_events(api_mode, "C", pdu, m["rnti"], m["tti"]) and not keep
            assert oracle_api_events(2, "C", pdu, m["rnti"], m["tti"]) == ([], False)
            if ev:
                hits.append(ev)
        assert len(hits) == 1 and len(hits[0]) == 1   # exactly one reconfiguration per attach carries the accept
        tti, rnti, id_type, msg_type, value = hits[0][0]
        assert (rnti, id_type, msg_type) == (70, 1, 6) and len(value) == 8 and int(value, 16) > 0
        if want:
            assert value == want   # hand-decoded in DESIGN / this round's notes: GUTI 09f155-0001-1a-cd5d47ec


def test_reconfiguration_walk_against_an_independent_encoder():
    rng = np.random.default_rng(5)
    n_hit = 0
    for trial in range(60):
        tmsi = int(rng.integers(1, 1 << 32))
        meas = None
        if trial % 3:
            meas = dict(objects=[(1 + i, int(rng.integers(0, 65536)), (int(rng.integers(0, 31)) if rng.integers(0, 2) else None),
                                  [(1 + j, int(rng.integers(0, 504)), int(rng.integers(0, 31))) for j in range(int(rng.integers(0, 4)))]) for i in range(int(rng.integers(1, 4)))],
                        reports=[("a3", int(rng.integers(-30, 31))), ("a1", int(rng.integers(0, 98))), ("periodical",)][:int(rng.integers(0, 4))],
                        ids=[(1, 1, 1), (2, 1, 2)][:int(rng.integers(0, 3))],
                        quantity=[None, (None, None), (4, None), (6, 9)][int(rng.integers(0, 4))],
                        gap=[None, "release", ("gp0", 17), ("gp1", 63)][int(rng.integers(0, 4))],
                        s_measure=[None, 70][int(rng.integers(0, 2))])
        guti, prot = bool(trial % 5), bool(trial % 2)
        pdu = _dcch_pdu(_encode_reconfig(_attach_accept(tmsi, guti=guti, protected=prot), meas=meas))
        ev, keep = oracle_api_events(3, "C", pdu, 4321, 77)
        assert (ev, keep) == host_api_events(3, "C", pdu, 4321, 77)
        if guti:
            assert ev == [(77, 4321, 1, 6, "%08x" % tmsi)], (trial, meas)
            n_hit += 1
        else:
            assert ev == []
    assert n_hit >= 40
    # not an attach accept / no NAS list / a handover command / truncated input: nothing, identically on both sides
    other = bytes([0x27, 1, 2, 3, 4, 9, 0x07, 0x44]) + bytes(20)
    for rrc in (_encode_reconfig(other), _encode_reconfig(None), _encode_reconfig(_attach_accept(5), mobility=True)):
        pdu = _dcch_pdu(rrc)
        assert oracle_api_events(0, "C", pdu, 9, 9) == ([], False) == host_api_events(0, "C", pdu, 9, 9)
    good = _dcch_pdu(_encode_reconfig(_attach_accept(0xCAFEF00D), meas=dict(objects=[(1, 3400, None, [])], quantity=(None, None))))
    for cut in range(6, len(good), 3):
        p = good[:cut]
        p = bytes([0x21, max(4, min(p[1], cut - 3)), 0x1f]) + p[3:]
        assert oracle_api_events(0, "C", p, 9, 9) == host_api_events(0, "C", p, 9, 9)
    for _ in range(200):
        p = bytes([0x21, 40, 0x1f, 0xa0, 0, 0, 0x20 | int(rng.integers(0, 8))]) + bytes(rng.integers(0, 256, 45, dtype=np.uint8))
        assert oracle_api_events(3, "C", p, 9, 9) == host_api_events(3, "C", p, 9, 9)


def test_reconfiguration_walk_steps_over_later_release_extension_additions():
    """measConfig of an LTE-A network: release 9-11 additions in MeasConfig, MeasObjectEUTRA, ReportConfigEUTRA and QuantityConfig and an event the
    release-8 CHOICE does not know (a6) travel as open types behind the root components; the NAS list behind them must still be found.  Both
    parsers (oracle o_rrc.c, product lsn_rrc.cc - two texts) against the encoder above, which knows nothing of either."""
    rng = np.random.default_rng(11)

    def adds(always=False):
        if not always and rng.integers(0, 2):
            return None
        n = int(rng.integers(1, 5))
        out = [bytes(rng.integers(0, 256, int(rng.integers(1, 5 if rng.integers(0, 8) else 140)), dtype=np.uint8)) if rng.integers(0, 3) else None for _ in range(n)]
        if all(x is None for x in out):
            out[0] = b"\x80"
        return out

    n_ext = 0
    for trial in range(80):
        tmsi = int(rng.integers(1, 1 << 32))
        objects = [(1 + i, int(rng.integers(0, 65536)), (int(rng.integers(0, 31)) if rng.integers(0, 2) else None),
                    [(1 + j, int(rng.integers(0, 504)), int(rng.integers(0, 31))) for j in range(int(rng.integers(0, 3)))], adds()) for i in range(int(rng.integers(1, 4)))]
        reports = [("a3", int(rng.integers(-30, 31)), adds()), ("a6", bytes(rng.integers(0, 256, int(rng.integers(1, 4)), dtype=np.uint8)), adds()),
                   ("a1", int(rng.integers(0, 98)), adds()), ("periodical", None, adds())][:int(rng.integers(1, 5))]
        meas = dict(objects=objects, reports=reports, ids=[(1, 1, 1), (2, 1, 2)][:int(rng.integers(0, 3))],
                    quantity=[None, (None, None, adds()), (4, None, adds(True)), (6, 9, adds())][int(rng.integers(0, 4))],
                    gap=[None, "release", ("gp0", 17), ("gp1", 63)][int(rng.integers(0, 4))], s_measure=[None, 70][int(rng.integers(0, 2))], additions=adds())
        n_ext += any(o[4] for o in objects) or any(r[2] for r in reports) or bool(meas["additions"])
        pdu = _dcch_pdu(_encode_reconfig(_attach_accept(tmsi, protected=bool(trial % 2)), meas=meas))
        ev, keep = oracle_api_events(3, "C", pdu, 4321, 77)
        assert (ev, keep) == host_api_events(3, "C", pdu, 4321, 77)
        assert ev == [(77, 4321, 1, 6, "%08x" % tmsi)], (trial, meas)
        # truncations inside the additions are rejected identically, never read past the end
        if pdu[1] < 128:
            for cut in range(8, len(pdu) - 4, 7):
                p = bytes([0x21, max(4, cut - 3), 0x1f]) + pdu[3:cut]
                assert oracle_api_events(3, "C", p, 9, 9) == host_api_events(3, "C", p, 9, 9)
    assert n_ext >= 40


def test_reconfiguration_walk_steps_over_inter_rat_measurement_configuration():
    """measConfig of a network with 3G / 2G / CDMA2000 neighbours (round-4 review, missing 4): MeasObjectUTRA / GERAN / CDMA2000, ReportConfigInterRAT (events b1, b2,
    periodical), the inter-RAT quantity configurations, HRPD pre-registration and speed-state parameters sit in FRONT of the NAS list - both parsers (oracle o_rrc.c,
    product lsn_rrc.cc: two texts) must step over all of them and still find the attach accept; the encoder above knows nothing of either"""
    rng = np.random.default_rng(23)

    def adds():
        if rng.integers(0, 3):
            return None
        return [bytes(rng.integers(0, 256, int(rng.integers(1, 6)), dtype=np.uint8)) if rng.integers(0, 2) else None for _ in range(int(rng.integers(1, 4)))] or None

    def opt(v):
        return v if rng.integers(0, 2) else None

    def irat_object(i):
        kind = ["utra", "geran", "cdma2000", "eutra"][int(rng.integers(0, 4))]
        if kind == "eutra":
            return (1 + i, int(rng.integers(0, 65536)), opt(int(rng.integers(0, 31))), [(1 + j, int(rng.integers(0, 504)), int(rng.integers(0, 31))) for j in range(int(rng.integers(0, 3)))], None)
        a = adds()
        if a is not None and all(x is None for x in a):
            a = None
        if kind == "utra":
            tdd = bool(rng.integers(0, 2))
            return ("utra", 1 + i, dict(arfcn=int(rng.integers(0, 16384)), offset=opt(int(rng.integers(-15, 16))), remove=opt([1 + int(v) for v in rng.integers(0, 32, int(rng.integers(1, 4)))]),
                                         tdd=tdd, cells=opt([(1 + j, int(rng.integers(0, 128 if tdd else 512))) for j in range(int(rng.integers(1, 5)))]),
                                         cgi=opt(int(rng.integers(0, 128 if tdd else 512)))), a)
        if kind == "geran":
            f = [("list", [int(v) for v in rng.integers(0, 1024, int(rng.integers(0, 6)))]), ("spaced", int(rng.integers(1, 9)), int(rng.integers(0, 32))),
                 ("bitmap", [int(v) for v in rng.integers(0, 256, int(rng.integers(1, 17)))])][int(rng.integers(0, 3))]
            return ("geran", 1 + i, dict(arfcn=int(rng.integers(0, 1024)), band=int(rng.integers(0, 2)), following=f, offset=opt(int(rng.integers(-15, 16))), ncc=opt(int(rng.integers(0, 256))),
                                          cgi=opt((int(rng.integers(0, 8)), int(rng.integers(0, 8))))), a)
        return ("cdma2000", 1 + i, dict(type=int(rng.integers(0, 2)), band=int(rng.integers(0, 18)), arfcn=int(rng.integers(0, 2048)), window=opt(int(rng.integers(0, 16))),
                                         offset=opt(int(rng.integers(-15, 16))), remove=opt([1 + int(v) for v in rng.integers(0, 32, int(rng.integers(1, 3)))]),
                                         cells=opt([(1 + j, int(rng.integers(0, 512))) for j in range(int(rng.integers(1, 4)))]), cgi=opt(int(rng.integers(0, 512)))), a)

    def threshold():
        rat = ["utra", "geran", "cdma2000"][int(rng.integers(0, 3))]
        if rat == "utra":
            return dict(rat=rat, rscp=int(rng.integers(-5, 92))) if rng.integers(0, 2) else dict(rat=rat, ecn0=int(rng.integers(0, 50)))
        return dict(rat=rat, v=int(rng.integers(0, 64)))

    def report():
        k = int(rng.integers(0, 5))
        if k == 0:
            return ("b1", threshold(), None)
        if k == 1:
            return ("b2", dict(threshold(), eutra_rsrp=int(rng.integers(0, 98))), None)
        if k == 2:
            return ("periodical_irat", int(rng.integers(0, 3)), None)
        if k == 3:
            return ("a3", int(rng.integers(-30, 31)), None)
        return ("periodical", None, None)

    n_irat = 0
    for trial in range(120):
        tmsi = int(rng.integers(1, 1 << 32))
        objects = [irat_object(i) for i in range(int(rng.integers(1, 5)))]
        reports = [report() for _ in range(int(rng.integers(0, 4)))]
        other = {}
        if rng.integers(0, 2):
            other["utra"] = (int(rng.integers(0, 2)), opt(int(rng.integers(0, 16))))
        if rng.integers(0, 2):
            other["geran"] = (opt(int(rng.integers(0, 16))),)
        if rng.integers(0, 2):
            other["cdma2000"] = int(rng.integers(0, 2))
        meas = dict(objects=objects, reports=reports, ids=[(1, 1, 1)][:int(rng.integers(0, 2))],
                    quantity=[None, (opt(4), opt(7), None, other)][int(rng.integers(0, 2))],
                    gap=[None, ("gp0", 17)][int(rng.integers(0, 2))], s_measure=opt(70),
                    prereg=[None, (True, None, None), (False, 17, [3]), (True, 200, [1, 255])][int(rng.integers(0, 4))],
                    speed=[None, "release", (3, 5, 4, 16, 1, 2)][int(rng.integers(0, 3))])
        n_irat += any(isinstance(o[0], str) for o in objects) or any(r[0] in ("b1", "b2", "periodical_irat") for r in reports)
        pdu = _dcch_pdu(_encode_reconfig(_attach_accept(tmsi, protected=bool(trial % 2)), meas=meas))
        ev, keep = oracle_api_events(3, "C", pdu, 4321, 77)
        assert (ev, keep) == host_api_events(3, "C", pdu, 4321, 77)
        assert ev == [(77, 4321, 1, 6, "%08x" % tmsi)], (trial, meas)
        if pdu[1] < 128:   # truncated messages: rejected identically, never read past the end
            for cut in range(8, len(pdu) - 4, 5):
                p = bytes([0x21, max(4, cut - 3), 0x1f]) + pdu[3:cut]
                assert oracle_api_events(3, "C", p, 9, 9) == host_api_events(3, "C", p, 9, 9)
    assert n_irat >= 80
