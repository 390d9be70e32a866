"""CPU tests of what the decode loop learns from decoded C-RNTI transport blocks: MAC DL-SCH walk + RRCConnectionSetup (p-a, betaOffset
indices, aperiodic CQI mode).  (a) the oracle against the reference's own captures (tests/golden/pcap_records.json: real
contention-resolution messages, real DL-SCH headers); (b) the product's host code (tests/native glue, no GPU) against the oracle on
those, on messages of the synthetic eNB and on corrupted / random input; (c) the UE-configuration database semantics of MCSTracking;
(d) the loop through the oracle worker: a UE whose connection setup changes p-a keeps decoding."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from lsn_testlib import OracleWorker, TxGen, hosttest, oracle, parse_pcap, scenario

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pcap_records.json")))


class OSub(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("lcid", "is_sdu", "off", "len")]


class OUeCfg(C.Structure):
    _fields_ = [("has_ue_config", C.c_uint32), ("p_a", C.c_float), ("i_offset_ack", C.c_uint32), ("i_offset_cqi", C.c_uint32),
                ("i_offset_ri", C.c_uint32), ("cqi_type", C.c_uint32), ("bits_used", C.c_uint32)]


def apis():
    o, h = oracle(), hosttest()
    o.o_mac_dlsch_parse.argtypes = [C.c_char_p, C.c_int, C.POINTER(OSub), C.c_int]
    o.o_rrc_conn_setup_decode.argtypes = [C.c_char_p, C.c_int, C.POINTER(OUeCfg)]
    h.lsnh_mac_dlsch_parse.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int]
    h.lsnh_rrc_conn_setup.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    h.lsnh_mcs_new.restype = C.c_void_p
    h.lsnh_mcs_free.argtypes = [C.c_void_p]
    h.lsnh_mcs_learn.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_uint16]
    h.lsnh_mcs_touch.argtypes = [C.c_void_p, C.c_uint16]
    h.lsnh_mcs_get.argtypes = [C.c_void_p, C.c_uint16, C.c_void_p]
    return o, h


def walk_o(o, pdu):
    sub = (OSub * 24)()
    n = o.o_mac_dlsch_parse(pdu, len(pdu), sub, 20)
    return [(s.lcid, s.is_sdu, s.off, s.len) for s in sub[:n]]


def walk_h(h, pdu):
    out = np.zeros(4 * 24, dtype=np.uint32)
    n = h.lsnh_mac_dlsch_parse(pdu, len(pdu), out.ctypes.data, 20)
    return [tuple(int(v) for v in out[4 * i:4 * i + 4]) for i in range(n)]


def rrc_o(o, sdu):
    c = OUeCfg()
    r = o.o_rrc_conn_setup_decode(sdu, len(sdu), C.byref(c))
    return (np.float32(c.p_a).view(np.uint32).item(), c.i_offset_ack, c.i_offset_cqi, c.i_offset_ri, c.cqi_type) if r else None, c.bits_used


def rrc_h(h, sdu):
    out = np.zeros(8, dtype=np.uint32)
    return tuple(int(v) for v in out[:5]) if h.lsnh_rrc_conn_setup(sdu, len(sdu), out.ctypes.data) else None


def test_real_contention_resolution_messages_decode_completely():
    o, h = apis()
    n = 0
    for name, fx in FIX.items():
        for m in fx["conn_setup"]:
            pdu = bytes.fromhex(m["pdu"])
            subs = walk_o(o, pdu)
            assert [(s[0], s[3]) for s in subs] == [(28, 6), (0, 20), (31, 0)] and subs == walk_h(h, pdu)
            sdu = pdu[subs[1][2]:subs[1][2] + subs[1][3]]
            cfg, bits = rrc_o(o, sdu)
            # the eNB of the captures: p-a dB0, betaOffset-ACK/RI/CQI-Index 6/6/6, no aperiodic report mode (CQI type stays 0 = wideband);
            # every field of the message is read and the walk ends inside the last octet
            assert cfg == (np.float32(0.0).view(np.uint32).item(), 6, 6, 6, 0) and 8 * len(sdu) - 8 < bits <= 8 * len(sdu)
            assert rrc_h(h, sdu) == cfg
            n += 1
    assert n >= 7


def test_real_dlsch_headers_walk_like_an_independent_parser():
    o, h = apis()
    n = 0
    for name, fx in FIX.items():
        for w in fx["dl_crnti_walks"]:
            head = bytes.fromhex(w["pdu_head"])
            pdu = head + bytes(w["length"] - len(head))
            subs = walk_o(o, pdu)
            assert [[s[0], s[3]] for s in subs] == w["subheaders"], w
            assert subs == walk_h(h, pdu)
            used = sum(s[3] for s in subs) + subs[0][2]  # headers + payloads; what a trailing padding subheader leaves over is padding
            assert used == w["length"] if subs[-1][1] else used <= w["length"]
            n += 1
    assert n >= 80


def test_product_decoder_equals_oracle_on_synthetic_corrupted_and_random_messages():
    o, h = apis()
    sc = scenario("cfg2", seed=3, nof_prb=25, n_rnti=6, dl_min=2, dl_max=3, ul_min=0, ul_max=0, msg4_period=2, msg4_p_a_idx=8)
    tx = TxGen(**sc)
    msgs = []
    for _ in range(160):
        _, _, pdus = tx.next()
        msgs += [p["payload"] for p in pdus if not p["is_ul"] and p["payload"][:2] == b"\x3c\x20"]
    assert len(msgs) >= 60
    rng = np.random.default_rng(1)
    seen_pa, seen_type, ok = set(), set(), 0
    for m in msgs:
        subs = walk_o(o, m)
        assert subs == walk_h(h, m) and subs[1][0] == 0
        sdu = m[subs[1][2]:subs[1][2] + subs[1][3]]
        cfg, bits = rrc_o(o, sdu)
        assert cfg is not None and cfg == rrc_h(h, sdu) and 8 * len(sdu) - 8 < bits <= 8 * len(sdu)
        seen_pa.add(cfg[0]); seen_type.add(cfg[4]); ok += 1
        for _ in range(30):  # bit errors and truncation: both sides must take the same decision and, when they accept, report the same
            b = bytearray(sdu)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            b = bytes(b[:int(rng.integers(1, len(b) + 1))])
            assert rrc_o(o, b)[0] == rrc_h(h, b)
    assert len(seen_pa) == 8 and seen_type == {0, 1, 2}
    for _ in range(4000):  # random PDUs: the MAC walk and whatever a CCCH-looking SDU decodes to
        pdu = bytes(rng.integers(0, 256, int(rng.integers(1, 80)), dtype=np.uint8))
        subs = walk_o(o, pdu)
        assert subs == walk_h(h, pdu)
        sdu = bytes([0x60 | int(rng.integers(0, 32))]) + pdu
        assert rrc_o(o, sdu)[0] == rrc_h(h, sdu)


def test_ue_configuration_database_semantics():
    """MCSTracking.cc:1444-1540: the first connection setup ever seen becomes the default of RNTIs without entry; an entry keeps the
    default it was created with; a UE's own connection setup replaces its entry's configuration."""
    o, h = apis()
    sc = scenario("cfg2", seed=5, nof_prb=25, n_rnti=4, dl_min=2, dl_max=2, ul_min=0, ul_max=0, msg4_period=2, msg4_p_a_idx=8)
    tx = TxGen(**sc)
    msgs = []
    while len(msgs) < 3:
        _, _, pdus = tx.next()
        msgs += [p["payload"] for p in pdus if not p["is_ul"] and p["payload"][:2] == b"\x3c\x20"]

    def cfg_of(m):
        s = walk_o(o, m)[1]
        return rrc_o(o, m[s[2]:s[2] + s[3]])[0]

    def get(m, rnti):
        out = np.zeros(8, dtype=np.uint32)
        h.lsnh_mcs_get(m, rnti, out.ctypes.data)
        return tuple(int(v) for v in out[:5]), int(out[5])

    dflt = (np.float32(0.0).view(np.uint32).item(), 10, 8, 11, 2)  # set_default_of_default_config
    m = h.lsnh_mcs_new()
    assert get(m, 100) == (dflt, 0)
    h.lsnh_mcs_touch(m, 100)                     # entry created before any connection setup: keeps the initial default for good
    assert h.lsnh_mcs_learn(m, msgs[0], len(msgs[0]), 200) == 1
    assert get(m, 200) == (cfg_of(msgs[0]), 1)
    assert get(m, 300) == (cfg_of(msgs[0]), 0)   # no entry: the default, which is now the first connection setup
    assert get(m, 100) == (dflt, 0)
    assert h.lsnh_mcs_learn(m, msgs[1], len(msgs[1]), 400) == 1
    assert get(m, 400) == (cfg_of(msgs[1]), 1)
    assert get(m, 300) == (cfg_of(msgs[0]), 0)   # the default does not move again
    h.lsnh_mcs_touch(m, 300)
    assert get(m, 300) == (cfg_of(msgs[0]), 0)   # a new entry copies the default in force
    assert h.lsnh_mcs_learn(m, msgs[2], len(msgs[2]), 100) == 1
    assert get(m, 100) == (cfg_of(msgs[2]), 1)
    assert h.lsnh_mcs_learn(m, b"\x21\x02\x1f\x00\x04", 5, 500) == 0 and get(m, 500) == (cfg_of(msgs[0]), 0)
    h.lsnh_mcs_free(m)


@pytest.mark.parametrize("p_a_idx", [0, 2, 7])
def test_oracle_worker_follows_p_a_of_the_connection_setup(p_a_idx):
    """64QAM at high code rate only decodes when the receiver scales with the right power offset: after a UE's connection setup
    (p-a != 0 dB) its transport blocks keep decoding, and nothing the worker emits differs from what was sent."""
    sc = scenario("cfg2", seed=3, nof_prb=25, n_rnti=3, dl_min=3, dl_max=3, ul_min=0, ul_max=0, mcs_min=20, mcs_max=28, msg4_period=6,
                  msg4_p_a_idx=p_a_idx, snr_db=36.0)
    tx = TxGen(**sc)
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"], sc["phich_ng_x6"])
    sent, configured, after = {}, set(), {}
    for _ in range(70):
        tti, iq, pdus = tx.next()
        ow.work(iq, tti)
        for p in pdus:
            if p["is_ul"]:
                continue
            sent.setdefault((tti, p["rnti"]), []).append(p["payload"])
            if p["rnti"] in configured:
                after[(tti, p["rnti"], p["tb"])] = p["payload"]
        configured |= {p["rnti"] for p in pdus if not p["is_ul"] and p["payload"][:2] == b"\x3c\x20"}
    recs = parse_pcap(ow.pcap_bytes())
    for r in recs:
        assert r["pdu"] in sent.get((r["sfn"] * 10 + r["sf"], r["rnti"]), [])
    got = {(r["sfn"] * 10 + r["sf"], r["rnti"]) for r in recs}
    hit = sum((k[0], k[1]) in got for k in after)
    assert len(configured) == 3 and len(after) > 60 and hit >= 0.9 * len(after), (len(after), hit)
    p_a_db = [-6.0, -4.77, -3.0, -1.77, 0.0, 1.0, 2.0, 3.0][p_a_idx]
    for rnti in configured:
        c = ow.ue_cfg(rnti)
        assert c[0] == 1 and abs(c[1] - p_a_db) < 1e-6
    assert ow.ue_cfg(0x0BAD)[0] == 0 and abs(ow.ue_cfg(0x0BAD)[1] - p_a_db) < 1e-6  # no entry: the default = the first connection setup
