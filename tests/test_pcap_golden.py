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
