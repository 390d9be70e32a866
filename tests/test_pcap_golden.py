"""CPU tests: MAC-LTE pcap framing of (a) the product's native writer (C ABI, no GPU needed) and (b) the oracle's writer
against the record layout pinned by the reference's example captures (tests/golden/pcap_records.json, generated from
/root/reference/pcap_file_example by tests/golden/make_pcap_fixture.py)."""
import ctypes as C
import json
import os
import struct

import pytest

import ltesniffer_amd as la
from lsn_testlib import oracle, parse_pcap

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pcap_records.json")))


def _fields(rec):
    # 01 dd tt | 02 RR RR | 03 UU UU | 04 SS SS | 07 cc | 0a 00 | 0f 00 | 01 | PDU
    assert rec[0] == 1 and rec[3] == 2 and rec[6] == 3 and rec[9] == 4 and rec[12] == 7 and rec[14] == 0x0A and rec[16] == 0x0F and rec[18] == 1
    fs = (rec[10] << 8) | rec[11]
    return dict(direction=rec[1], rnti_type=rec[2], rnti=(rec[4] << 8) | rec[5], tti=(fs >> 4) * 10 + (fs & 15), crc_ok=rec[13])


@pytest.mark.parametrize("name", sorted(FIX))
def test_record_framing_matches_reference_captures(name):
    fx = FIX[name]
    assert bytes.fromhex(fx["global_header"]) == struct.pack("<IHHiIII", 0xA1B2C3D4, 2, 4, 0, 0, 65535, 147)
    w = la.PcapWriter(None)
    o = oracle()
    op = o.o_pcap_open_mem()
    for hexrec, full_len in zip(fx["records"], fx["record_lens"]):
        rec = bytes.fromhex(hexrec)
        f = _fields(rec)
        assert (f["tti"] % 10) < 10
        pdu = rec[19:] + bytes(full_len - len(rec))  # the fixture keeps the first bytes of long PDUs only
        w.write(f, pdu)
        o.o_pcap_write(op, pdu, len(pdu), f["tti"], f["rnti"], f["direction"], f["rnti_type"], f["crc_ok"], 0, 0)
    n = C.c_size_t()
    obytes = C.string_at(o.o_pcap_mem(op, C.byref(n)), n.value)
    assert w.bytes() == obytes, "product writer and oracle writer disagree"
    recs = parse_pcap(w.bytes())
    assert len(recs) == len(fx["records"]) == w.nof_records()
    for r, hexrec in zip(recs, fx["records"]):
        ref = bytes.fromhex(hexrec)
        assert r["ctx"] == ref[:19]
        assert r["pdu"][:len(ref) - 19] == ref[19:]
    assert w.bytes()[:24] == bytes.fromhex(fx["global_header"])
    w.close()
    o.o_pcap_close(op)


def test_python_framing_helper_equals_native_writer():
    ctx = dict(tti=4305, rnti=0xFFFF, direction=1, rnti_type=4, crc_ok=1)
    w = la.PcapWriter(None)
    w.write(ctx, b"\x00\x01\x40")
    assert parse_pcap(w.bytes())[0]["ctx"] + b"\x00\x01\x40" == la.mac_lte_record(ctx, b"\x00\x01\x40")
    # first record of the reference's ltesniffer_dl_mode.pcap
    assert la.mac_lte_record(ctx, b"")[:19].hex() == "01010402ffff030000041ae507010a000f0001"


def test_wall_clock_and_file_output(tmp_path):
    p = str(tmp_path / "x.pcap")
    w = la.PcapWriter(p)
    w.write(dict(tti=17, rnti=0x46, direction=1, rnti_type=3), b"\xAA" * 5)
    w.close()
    data = open(p, "rb").read()
    assert len(data) == 24 + 16 + 19 + 5
    ts = struct.unpack("<I", data[24:28])[0]
    assert ts > 1_600_000_000  # gettimeofday stamp, like the reference (excluded from every diff)


# ---- procedure-level golden data of the same captures: what the reference decoded around a random access ----
def _fixture():
    import json, os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pcap_records.json")))


@pytest.mark.parametrize("name", ["ltesniffer_dl_mode.pcap", "ltesniffer_ul_mode.pcap"])
def test_real_rar_pdu_gives_the_rnti_and_the_msg3_the_reference_then_decoded(name):
    """The reference's own captures contain a random-access response (RA-RNTI 2) followed by traffic of the UE it addressed:
    both RAR parsers (oracle and product) must read the temporary C-RNTI that the following records carry, and - UL_MODE capture -
    the 20-bit grant must describe the Msg3 that the reference decoded exactly 6 subframes later (ULSchedule's tti - 6 rule):
    3 PRBs, MCS 0 -> TBS 56 bits = the 7-byte uplink PDU; Msg4 echoes bytes 1..6 of that Msg3 as contention-resolution identity."""
    import ctypes as C
    from lsn_testlib import OCell, hosttest, oracle
    ra = _fixture()[name]["random_access"]
    assert len(ra) == 1
    pdu = bytes.fromhex(ra[0]["rar_pdu"])
    o, h = oracle(), hosttest()

    class ORar(C.Structure):
        _fields_ = [(n, C.c_uint32) for n in ("rapid", "ta", "hopping", "riv", "mcs", "tpc", "ul_delay", "csi_req")] + \
                   [("t_crnti", C.c_uint16), ("grant_ok", C.c_int)] + [(n, C.c_uint32) for n in ("L_prb", "n_prb", "mcs_idx")] + \
                   [("mod", C.c_int), ("tbs", C.c_int), ("rv", C.c_int), ("n_prb2", C.c_uint32), ("hop", C.c_uint32)]
    o.o_rar_parse.argtypes = [C.POINTER(OCell), C.c_char_p, C.c_int, C.POINTER(ORar), C.c_int]
    cell = OCell(100, 2, 1, 1)  # the capture is a 20 MHz cell: RIV 202 = 100 * (3 - 1) + 2
    r = (ORar * 8)()
    assert o.o_rar_parse(C.byref(cell), pdu, len(pdu), r, 8) == 1
    out = (C.c_uint32 * 64)()
    tbs = (C.c_int * 8)()
    h.lsnh_rar_parse.argtypes = [C.c_uint32, C.c_char_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.c_int]
    assert h.lsnh_rar_parse(100, pdu, len(pdu), out, tbs, 8) == 1
    crnti = [q["rnti"] for q in ra[0]["following"] if q["rnti_type"] == 3][0]
    assert r[0].t_crnti == out[0] == crnti == 70
    assert (r[0].rapid, r[0].ta, r[0].riv, r[0].mcs) == (out[1], out[2], out[3], out[4]) == (pdu[0] & 0x3F, 1, 202, 0)
    assert r[0].grant_ok == 1 and out[5] == 1 and (r[0].n_prb, r[0].L_prb, r[0].tbs) == (out[6], out[7], tbs[0]) == (2, 3, 56)
    if name == "ltesniffer_ul_mode.pcap":
        msg3, msg4 = ra[0]["following"][0], ra[0]["following"][1]
        assert msg3["direction"] == 0 and msg3["rnti"] == 70 and msg3["dtti"] == 6 and len(bytes.fromhex(msg3["pdu"])) * 8 == r[0].tbs
        assert msg4["direction"] == 1 and bytes.fromhex(msg3["pdu"])[1:7] in bytes.fromhex(msg4["pdu"])


def test_real_record_lengths_are_sizes_of_the_tbs_table():
    """every C-RNTI PDU of the reference's captures has a length that the transport-block-size table produces - downlink for some
    1..100 PRBs, uplink for a PRB count of the 2^a 3^b 5^c set (UL_Sniffer_PUSCH.cc:3-10)"""
    import ctypes as C
    from lsn_testlib import VALID_UL_PRB, hosttest
    h = hosttest()
    h.lsnh_tbs.argtypes = [C.c_int, C.c_uint32]
    dl = {h.lsnh_tbs(i, n) // 8 for i in range(34) for n in range(1, 101)}
    ul = {h.lsnh_tbs(i, n) // 8 for i in range(27) for n in VALID_UL_PRB}
    fx = _fixture()
    seen_dl = seen_ul = 0
    for name in ("ltesniffer_dl_mode.pcap", "ltesniffer_ul_mode.pcap", "api_collector.pcap"):
        for key, lens in fx[name]["pdu_lengths"].items():
            if key == "1/3":
                assert set(lens) <= dl, (name, sorted(set(lens) - dl))
                seen_dl += len(lens)
            elif key == "0/3":
                assert set(lens) <= ul, (name, sorted(set(lens) - ul))
                seen_ul += len(lens)
    assert seen_dl >= 40 and seen_ul >= 50
