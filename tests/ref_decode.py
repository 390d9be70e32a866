"""Driver of oracle/_ref/libref_falcon_decode.so: the REFERENCE'S OWN downlink decode control flow - PDSCH_Decoder::decode_dl_mode (src/src/DL_Sniffer_PDSCH.cc:881-1291)
on top of its own DCICollection / falcon_dci.c / MCSTracking / HARQ / RNTIManager - compiled from /root/reference (oracle/Makefile.ref; stand-in srsRAN types and L2/L3
classes, oracle/ref_shim_search/decode_glue.cc).  The PDSCH decoder itself (srsRAN) is a SCRIPTED one: a seeded function of (tti, RNTI, block, size) hands back CRC
verdicts and payload bytes - random bytes, real RRCConnectionSetup PDUs (the reference's own captures), random-access responses - and the same function answers the
oracle's decode calls (o_worker_set_script_decoder).  Compared per subframe: every decode call as configured (grant of the table that was tried, MIMO configuration,
redundancy versions, p_a), every record handed to the pcap writer; at check points the tracking database, the RNTI manager's activation reasons, the UE
configurations.  Test infrastructure only."""
import ctypes as C
import hashlib
import json
import os
import random
import struct

import ref_collect as RC
from lsn_testlib import OCell, OUeCfg, OWorkerCfg, oracle, parse_pcap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_falcon_decode.so")
REF_SOURCES = ["src/src/DL_Sniffer_PDSCH.cc"] + RC.REF_SOURCES + ["lib/src/util/RNTIManager.cc", "lib/src/util/Histogram.cc", "lib/src/util/Interval.cc"]
SCRIPT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_float, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_int32))

# (name, nof_prb, nof_ports, cell id, cp, mcs_tracking_mode, harq_mode, rx antennas, subframes, seed, fill byte of memory the reference reads unwritten)
LIVES = [
    ("100prb_2rx", 100, 2, 1, 0, 1, 0, 2, 500, 31, 1),
    # harq_mode 1 (unreachable in the reference: ArgManager.cc:50,211-213) with the tracking database off = every entry on the 64QAM table: under the 256QAM table
    # DCICollection.cc:244-249 WRITES the HARQ size into the grant it did not compute and the gate then reads that - behaviour of uninitialised memory, not of a design
    ("100prb_2rx_harq_64qam_table", 100, 2, 3, 0, 0, 1, 2, 500, 32, 1),
    ("50prb_1rx", 50, 2, 7, 0, 1, 0, 1, 400, 33, 1),
    ("25prb_1port_2rx", 25, 1, 2, 0, 1, 0, 2, 400, 34, 1),
    ("75prb_extcp", 75, 2, 5, 1, 1, 0, 2, 300, 35, 1),
    ("15prb_4port", 15, 4, 9, 0, 1, 0, 2, 300, 36, 1),
    ("100prb_both_tables_mode", 100, 2, 6, 0, 2, 0, 2, 300, 37, 1),
    ("100prb_tracking_off", 100, 2, 8, 0, 0, 0, 2, 300, 38, 1),
]


_MSG4 = None


def _msg4_templates():
    """MAC PDUs that carry an RRCConnectionSetup: the real ones of the reference's own captures, and the synthetic transmitter's with every p-a value and every
    aperiodic CQI mode (tools/txgen: msg4_p_a_idx = 8 cycles through them)"""
    global _MSG4
    if _MSG4 is None:
        from lsn_testlib import TxGen, scenario
        fx = json.load(open(os.path.join(ROOT, "tests", "golden", "pcap_records.json")))
        out = []
        for name in fx:
            for m in fx[name].get("conn_setup", []):
                out.append(bytes.fromhex(m["pdu"]))
        assert out
        tx = TxGen(**scenario("cfg2", seed=3, nof_prb=25, n_rnti=6, dl_min=2, dl_max=3, ul_min=0, ul_max=0, msg4_period=2, msg4_p_a_idx=8))
        syn = []
        for _ in range(120):
            _, _, pdus = tx.next()
            syn += [p["payload"] for p in pdus if not p["is_ul"] and p["payload"][:2] == b"\x3c\x20"]
        assert len(syn) >= 40
        _MSG4 = out + syn[:40]
    return _MSG4


def _mix(*v):
    h = 0x9E3779B97F4A7C15
    for x in v:
        h = ((h ^ (x & 0xFFFFFFFFFFFF)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        h ^= h >> 29
    return h


class ScriptedDecoder:
    """CRC verdict and payload of a transport block as a pure function of (seed, tti, rnti, block, size): both sides get the same answers whenever they ask the same
    question, and the log says what each side asked"""

    def __init__(self, seed, ues):
        self.seed, self.ues, self.msg4 = seed, ues, _msg4_templates()
        self.log = []
        self.fn = SCRIPT_FN(self._cb)

    def payload(self, h, rnti, n):
        kind = (h >> 12) % 10
        if 1 <= rnti <= 10:  # a random-access response: E/T/RAPID sub-headers, then R | TA(11) | grant(20) | T-CRNTI(16) per RAPID
            if kind < 8 and n >= 7:
                k = 1 + (h >> 20) % 2 if n >= 14 else 1
                hdr = bytes([(0x80 if i + 1 < k else 0) | 0x40 | ((h >> (24 + 6 * i)) & 0x3F) for i in range(k)])
                body = b""
                for i in range(k):
                    g20 = (h >> (7 + 13 * i)) & 0xFFFFF
                    t = self.ues[(h >> (3 + 5 * i)) % len(self.ues)]
                    body += bytes([0x00, 0x10 | (g20 >> 16), (g20 >> 8) & 0xFF, g20 & 0xFF, t >> 8, t & 0xFF])
                return (hdr + body + bytes(n))[:n]
        elif rnti < 0xFFF4 and kind < 4:  # a MAC PDU with a CCCH SDU that is an RRCConnectionSetup (real ones), padded
            m = self.msg4[(h >> 20) % len(self.msg4)]
            if kind == 3 and m[1] == 0x20:  # ... the same bytes on logical channel 1: the unknown-table branch tries EVERY SDU as a connection setup (DL_Sniffer_PDSCH.cc:1140), the known-table branch only LCID 0 (:1049)
                m = m[:1] + b"\x21" + m[2:]
            if len(m) <= n:
                return m + bytes(n - len(m))
        return random.Random(h).randbytes(n)

    def _cb(self, user, call, p_a, pl0, pl1, crc):
        w = tuple(call[i] for i in range(16))
        verdict = [0, 0]
        for tb, pl in ((0, pl0), (1, pl1)):
            en, tbs = w[6 + 5 * tb], w[8 + 5 * tb]
            if en and tbs > 0:
                h = _mix(self.seed, w[0], w[1], tb, tbs)
                ok = (h % 100) < 55
                verdict[tb] = int(ok)
                crc[tb] = int(ok)
                if ok:
                    b = self.payload(h, w[1], tbs // 8)
                    C.memmove(pl, b, len(b))
        self.log.append((w, round(float(p_a), 3), tuple(verdict)))
        return 0


def _tb0_mcs(o, cell, fmt, rnti, bits):
    from ref_grants import ODciDl
    d = ODciDl()
    o.o_dci_unpack_dl.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint16, C.c_void_p]
    if o.o_dci_unpack_dl(C.byref(cell), (C.c_uint8 * 128)(*bits), len(bits), fmt, rnti, C.byref(d)) != 0:
        return 0
    if fmt >= RC.FMT2 and d.tb[0].mcs_idx == 0 and d.tb[0].rv == 1:  # first block disabled (36.212 5.3.3.1.5): no size either
        return 99
    return d.tb[0].mcs_idx


def script(life, reserved_first_block=False):
    """like ref_collect.script, with what the decoders need on top: RNTIs of a UE pool whose DCI come in the formats of their transmission mode, common RNTIs in
    formats 1A / 1C, and now and then an ageing pass of the tracking database.

    One kind of DCI is left out unless `reserved_first_block`: a user grant beyond format 1A whose FIRST block carries I_MCS >= 28 or is disabled.  Under the 256QAM table
    that block has no size of its own, and decode_dl_mode's gate (DL_Sniffer_PDSCH.cc:887) then reads tb[0].tbs of the 64QAM-table grant - which the reference did not compute
    for a 256QAM-table entry: uninitialised memory, i.e. whatever an earlier grant left in that heap block.  The oracle defines the gate on the grant that is used
    (o_worker.c header).  tests/test_ref_decode.py has a test of its own for that case."""
    name, nprb, ports, cid, cp, mode, harq, nrx, nsf, seed, fill = life
    rng = random.Random(seed)
    o = oracle()
    cell = OCell(nprb, ports, cid, 1, 0, cp)
    sizes = RC._sizes(o, nprb, ports)
    ues = [rng.randrange(0x0100, 0xFFF0) for _ in range(20)]
    fmt_of = {u: rng.choice((RC.FMT1, RC.FMT2, RC.FMT2A, RC.FMT2A, RC.FMT1B, RC.FMT2B, RC.FMT1D)) for u in ues}
    ev = []
    tti = rng.randrange(0, 10240)
    for k in range(nsf):
        tti = (tti + 1) % 10240
        if k and k % 97 == 0:
            ev.append(("age",))
        dcis = []
        for _k in range(rng.randrange(0, 7)):
            u = rng.random()
            if u < 0.16:
                rnti, fmt = rng.choice((0xFFFF, 0xFFFE, rng.randrange(1, 11), rng.randrange(2, 10))), rng.choice((RC.FMT1A, RC.FMT1A, RC.FMT1C))
            elif u < 0.19:
                rnti, fmt = rng.randrange(0x000B, 0xFFF4), rng.randrange(1, 9)
            else:
                rnti = rng.choice(ues)
                fmt = rng.choice((fmt_of[rnti], fmt_of[rnti], fmt_of[rnti], RC.FMT1A, RC.FMT0))
            n = sizes[fmt]
            while True:
                bits = [rng.randrange(2) for _ in range(n)]
                if fmt == RC.FMT0:
                    bits[0] = 0
                if fmt == RC.FMT1A:
                    bits[0] = 1
                if reserved_first_block or fmt in (RC.FMT0, RC.FMT1A, RC.FMT1C) or not (0x000B <= rnti <= 0xFFF3) or _tb0_mcs(o, cell, fmt, rnti, bits) < 28:
                    break
            L = rng.randrange(4)
            dcis.append((rnti, fmt, L, rng.randrange(0, 80) // (1 << L) * (1 << L), rng.randrange(0, 40), bits))
        if harq and len(ev) >= 8 and rng.random() < 0.6:  # retransmissions: a grant of 8 subframes ago again, bit for bit (same process, same NDI, same size)
            old = [e for e in ev if e[0] == "sf"][-8:]
            if len(old) == 8 and (old[0][1] * 10 + old[0][2] + 8) % 10240 == tti:
                for d in old[0][4]:
                    if d[1] != RC.FMT0 and 0x000B <= d[0] <= 0xFFF3 and rng.random() < 0.7 and all(x[0] != d[0] for x in dcis):
                        dcis.append(d)
        ev.append(("sf", tti // 10, tti % 10, rng.randrange(1, 4), dcis))
    return ues, ev


class Reference:
    name = "reference"

    def __init__(self):
        self.lib = C.CDLL(REF_SO)
        o = oracle()
        L = self.lib
        L.ref_collect_bind.argtypes = [C.c_void_p] * 3
        L.ref_collect_bind(C.cast(o.o_dci_unpack_dl, C.c_void_p), C.cast(o.o_dci_unpack_ul, C.c_void_p), C.cast(o.o_tbs_from_idx, C.c_void_p))
        L.ref_decode_bind.argtypes = [C.c_void_p] * 6
        L.ref_decode_bind(*[C.cast(f, C.c_void_p) for f in (o.o_mac_dlsch_parse, o.o_rrc_conn_setup_decode, o.o_rar_parse, o.o_paging_decode, o.o_rrc_reconfig_tmsi, o.o_sib2_decode)])
        L.ref_decode_set_script.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_decode_set_fill.argtypes = [C.c_int]
        L.ref_decode_new.restype = C.c_void_p
        L.ref_decode_new.argtypes = [C.c_uint32] * 4 + [C.c_int] * 4
        L.ref_decode_free.argtypes = [C.c_void_p]
        L.ref_decode_collect.restype = C.c_void_p
        L.ref_decode_collect.argtypes = [C.c_void_p]
        L.ref_collect_begin.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.ref_collect_add.argtypes = [C.c_void_p, C.c_uint16, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.ref_collect_set_now_ms.argtypes = [C.c_uint64]
        L.ref_decode_dl_mode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.ref_decode_records.restype = C.c_uint32
        L.ref_decode_records.argtypes = [C.c_void_p, C.c_uint32]
        L.ref_decode_table.argtypes = [C.c_void_p, C.c_uint16]
        L.ref_decode_rnti_reason.argtypes = [C.c_void_p, C.c_uint16]
        L.ref_decode_ue_config.argtypes = [C.c_void_p, C.c_uint16, C.c_void_p, C.c_void_p]
        L.ref_decode_nof_tracked.restype = C.c_uint32
        L.ref_decode_nof_tracked.argtypes = [C.c_void_p]
        L.ref_decode_update_database.argtypes = [C.c_void_p]

    def open(self, life, dec):
        name, nprb, ports, cid, cp, mode, harq, nrx, nsf, seed, fill = life
        self.lib.ref_decode_set_fill(fill)
        self.lib.ref_decode_set_script(C.cast(dec.fn, C.c_void_p), None)
        self.h = self.lib.ref_decode_new(nprb, ports, cid, cp, mode, harq, nrx, -1)
        self.c = self.lib.ref_decode_collect(self.h)

    def close(self):
        self.lib.ref_decode_free(self.h)

    def now(self, k):
        self.lib.ref_collect_set_now_ms(k)

    def age(self):
        self.lib.ref_decode_update_database(self.h)

    def subframe(self, sfn, sf_idx, cfi, dcis):
        self.lib.ref_collect_begin(self.c, sfn, sf_idx, cfi)
        for rnti, fmt, L, ncce, hv, bits in dcis:
            self.lib.ref_collect_add(self.c, rnti, fmt, L, ncce, hv, (C.c_uint8 * 128)(*bits), len(bits))
        self.lib.ref_decode_dl_mode(self.h, sfn, sf_idx)
        buf = (C.c_uint32 * (7 * 64))()
        n = self.lib.ref_decode_records(buf, 64)
        return [tuple(buf[7 * i:7 * i + 5]) + (buf[7 * i + 5] | (buf[7 * i + 6] << 32),) for i in range(min(n, 64))]

    def state(self, rntis):
        out = []
        for r in rntis:
            p_a, w = C.c_float(), (C.c_uint32 * 5)()
            self.lib.ref_decode_ue_config(self.h, r, C.byref(p_a), w)
            out.append((r, self.lib.ref_decode_table(self.h, r), self.lib.ref_decode_rnti_reason(self.h, r), round(p_a.value, 3)) + tuple(w))
        return (self.lib.ref_decode_nof_tracked(self.h), out)


def _fnv(b):
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


class Oracle:
    name = "oracle"
    KIND = {3: 1, 2: 2, 4: 3, 1: 4}  # MAC-LTE RNTI type (C, RA, SI, P) -> the recorder's kind

    def __init__(self):
        self.o = oracle()
        o = self.o
        RC.Oracle()  # (argtypes of the collect probes)
        o.o_worker_set_script_decoder.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        o.o_worker_collect_set_now.argtypes = [C.c_void_p, C.c_uint32]
        o.o_worker_collect_decode_dl_mode.argtypes = [C.c_void_p]
        o.o_worker_collect_find_table.argtypes = [C.c_void_p, C.c_uint16]
        o.o_worker_collect_update_database.argtypes = [C.c_void_p]
        o.o_worker_ue_cfg.argtypes = [C.c_void_p, C.c_uint16, C.c_void_p]
        o.o_worker_rntiman.restype = C.c_void_p
        o.o_worker_rntiman.argtypes = [C.c_void_p]
        o.o_rntiman_get_activation_reason.argtypes = [C.c_void_p, C.c_uint16]
        o.o_worker_nof_tracked.restype = C.c_uint32
        o.o_worker_nof_tracked.argtypes = [C.c_void_p]
        o.o_pcap_open_mem.restype = C.c_void_p
        o.o_worker_set_pcap.argtypes = [C.c_void_p, C.c_void_p]
        o.o_pcap_mem.restype = C.POINTER(C.c_uint8)
        o.o_pcap_mem.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]

    def open(self, life, dec):
        name, nprb, ports, cid, cp, mode, harq, nrx, nsf, seed, fill = life
        cfg = OWorkerCfg(OCell(nprb, ports, cid, 1, 0, cp), nrx, 5, 0.99, 0, mode, 12, 1)
        self.h = self.o.o_worker_new(C.byref(cfg))
        if harq:
            self.o.o_worker_set_harq(self.h, 1)
        self.pcap = self.o.o_pcap_open_mem()
        self.o.o_worker_set_pcap(self.h, self.pcap)
        self.o.o_worker_set_script_decoder(self.h, C.cast(dec.fn, C.c_void_p), None)
        self.off = 24

    def close(self):
        self.o.o_worker_free(self.h)

    def now(self, k):
        self.o.o_worker_collect_set_now(self.h, k)

    def age(self):
        self.o.o_worker_collect_update_database(self.h)

    def subframe(self, sfn, sf_idx, cfi, dcis):
        self.o.o_worker_collect_begin(self.h, sfn, sf_idx, cfi)
        for rnti, fmt, L, ncce, hv, bits in dcis:
            self.o.o_worker_collect_add(self.h, rnti, fmt, L, ncce, hv, (C.c_uint8 * 128)(*bits), len(bits))
        self.o.o_worker_collect_decode_dl_mode(self.h)
        n = C.c_size_t()
        p = self.o.o_pcap_mem(self.pcap, C.byref(n))
        data = C.string_at(p, n.value)
        new = b"\xd4\xc3\xb2\xa1" + bytes(20) + data[self.off:]
        self.off = n.value
        return [(self.KIND[r["rnti_type"]], r["sfn"] * 10 + r["sf"], r["rnti"], len(r["pdu"]), r["crc"], _fnv(r["pdu"])) for r in parse_pcap(new)]

    def harq_stats(self):
        st = (C.c_uint32 * 5)()
        self.o.o_worker_harq_stats.argtypes = [C.c_void_p, C.c_void_p]
        self.o.o_worker_harq_stats(self.h, st)
        return tuple(st)

    def state(self, rntis):
        out = []
        rm = self.o.o_worker_rntiman(self.h)
        for r in rntis:
            c = OUeCfg()
            self.o.o_worker_ue_cfg(self.h, r, C.byref(c))
            out.append((r, self.o.o_worker_collect_find_table(self.h, r), self.o.o_rntiman_get_activation_reason(rm, r), round(c.p_a, 3), c.has_ue_config, c.i_offset_ack, c.i_offset_cqi,
                        c.i_offset_ri, c.cqi_type))
        return (self.o.o_worker_nof_tracked(self.h), out)


def run(side, life, reserved_first_block=False):
    """-> list of per-subframe (decode calls [(call16, p_a, verdicts)], records) + check points ("state", ...)"""
    ues, ev = script(life, reserved_first_block)
    dec = ScriptedDecoder(life[9], ues)
    side.open(life, dec)
    out, k = [], 0
    probe = sorted(set(ues) | {0xFFFF, 0xFFFE, 3, 0x2345})
    for e in ev:
        if e[0] == "age":
            side.age()
            out.append(("state", side.state(probe)))
            continue
        k += 1
        side.now(k)
        dec.log = []
        recs = side.subframe(e[1], e[2], e[3], e[4])
        out.append((list(dec.log), recs))
    out.append(("state", side.state(probe)))
    if hasattr(side, "harq_stats"):
        out.append(("harq", side.harq_stats()))
    side.close()
    return out


def digest(results):
    h = hashlib.sha256()
    for r in results:
        h.update(repr(r).encode())
    return h.hexdigest()[:32]


def facts(results):
    results = [r for r in results if r[0] != "harq"]
    calls = [c for r in results if r[0] != "state" for c in r[0]]
    recs = [x for r in results if r[0] != "state" for x in r[1]]
    return {"subframes": sum(r[0] != "state" for r in results), "decode_calls": len(calls), "two_block_calls": sum(c[0][6] and c[0][11] for c in calls),
            "calls_by_modulation": [sum(c[0][7] == q or c[0][12] == q for c in calls) for q in (2, 4, 6, 8)],
            "calls_by_tx_scheme": [sum(c[0][3] == t for c in calls) for t in range(4)], "blocks_passed": sum(sum(c[2]) for c in calls),
            "records": len(recs), "records_by_kind": [sum(x[0] == k for x in recs) for k in (1, 2, 3, 4)],
            "p_a_values": sorted({c[1] for c in calls})}


def reference_sources_sha256(ref="/root/reference"):
    h = hashlib.sha256()
    for f in REF_SOURCES:
        h.update(open(os.path.join(ref, f), "rb").read())
    return h.hexdigest()
