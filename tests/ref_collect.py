"""Driver of oracle/_ref/libref_falcon_collect.so: the REFERENCE'S OWN DCI collection - src/src/DCICollection.cc (addCandidate), lib/src/phy/falcon_phch/
falcon_dci.c (srsran_dci_msg_to_trace_timestamp, the RAR-grant chain), dl_sniffer_pdsch.c INCLUDING the C-RNTI branch of dl_sniffer_ra_dl_dci_to_grant,
ul_sniffer_pusch.c, src/src/ULSchedule.cc, with MCSTracking.cc and HARQ.cc behind them - compiled from /root/reference (oracle/Makefile.ref; stand-in srsRAN
types; oracle/ref_shim_search/collect_glue.cc binds the DCI bit unpacking and the TBS table to the oracle and WRITES the resource-allocation functions of
TS 36.213 7.1.6 / 7.1.7 a second time, independently of the oracle's text).

`lives()` are seeded scripts: per subframe a handful of accepted DCI (random payload bits of the right size, every format, C-RNTIs with known and unknown
MCS tables, SI / P / RA-RNTIs), and between subframes what the decoders feed back (a learnt table, a HARQ record).  The same script goes through the
reference, the oracle's restatement (o_worker.c: add_candidate) and the product's host code (lsn_search.cc: finishSubframe).  Test infrastructure only."""
import ctypes as C
import hashlib
import os
import random

from lsn_testlib import OCell, OWorkerCfg, oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_falcon_collect.so")
REF_SOURCES = ["src/src/DCICollection.cc", "lib/src/phy/falcon_phch/falcon_dci.c", "lib/src/phy/falcon_phch/dl_sniffer_pdsch.c",
               "lib/src/phy/falcon_phch/ul_sniffer_pusch.c", "src/src/ULSchedule.cc", "src/src/MCSTracking.cc", "src/src/HARQ.cc"]
DL_WORDS, UL_WORDS = 64, 32
FMT0, FMT1, FMT1A, FMT1B, FMT1C, FMT1D, FMT2, FMT2A, FMT2B = range(9)

# (name, nof_prb, nof_ports, cell id, cp, mcs_tracking_mode, harq_mode, sniffer_mode, SIB2 hop offset or None, subframes, seed)
LIVES = [
    ("100prb_2port", 100, 2, 1, 0, 1, 0, 0, None, 400, 11),
    ("100prb_2port_harq", 100, 2, 3, 0, 1, 1, 0, None, 400, 12),
    ("50prb_4port_sib2", 50, 4, 7, 0, 1, 0, 0, 5, 300, 13),
    ("25prb_1port", 25, 1, 2, 0, 1, 0, 0, None, 300, 14),
    ("75prb_2port_extcp", 75, 2, 5, 1, 1, 0, 0, 8, 300, 15),
    ("15prb_2port", 15, 2, 9, 0, 1, 0, 0, 2, 300, 16),
    ("6prb_1port", 6, 1, 4, 0, 1, 0, 0, None, 300, 17),
    ("100prb_both_tables_mode", 100, 2, 6, 0, 2, 0, 0, None, 200, 18),
    ("100prb_tracking_off", 100, 2, 8, 0, 0, 0, 0, None, 200, 19),
    ("50prb_ul_mode", 50, 1, 10, 0, 1, 0, 1, 4, 300, 20),
]


def _sizes(o, nprb, ports):
    o.o_dci_format_sizeof.restype = C.c_uint32
    o.o_dci_format_sizeof.argtypes = [C.c_void_p, C.c_int]
    cell = OCell(nprb, ports, 0, 1, 0, 0)
    return [o.o_dci_format_sizeof(C.byref(cell), f) for f in range(9)]


def script(life):
    """-> list of events: ("sf", sfn, sf_idx, cfi, [dci, ...]) with dci = (rnti, format, L, ncce, histval, bits) | ("mcs", rnti, table) |
    ("harq", rnti, pid, tid, sfn, sf_idx, decoded, ndi, rv, tbs)"""
    name, nprb, ports, cid, cp, mode, harq, smode, hop, nsf, seed = life
    rng = random.Random(seed)
    sizes = _sizes(oracle(), nprb, ports)
    ues = [rng.randrange(0x0100, 0xFFF0) for _ in range(24)]
    ev = []
    tti = rng.randrange(0, 10240)
    for _ in range(nsf):
        tti = (tti + 1 + (rng.random() < 0.02) * rng.randrange(1, 50)) % 10240
        if rng.random() < 0.15:
            ev.append(("mcs", rng.choice(ues), rng.choice((0, 0, 1, 1, 2))))
        if harq and rng.random() < 0.5:
            ev.append(("harq", rng.choice(ues), rng.randrange(8), rng.randrange(2), tti // 10, tti % 10, rng.randrange(2), rng.randrange(2), rng.randrange(4),
                       rng.choice((0, 16, 328, 2216, 14112, 75376))))
        dcis = []
        for _k in range(rng.randrange(0, 9)):
            u = rng.random()
            if u < 0.10:
                rnti, fmt = rng.choice((0xFFFF, 0xFFFE, rng.randrange(1, 11))), rng.choice((FMT1A, FMT1A, FMT1C, FMT1))
            elif u < 0.13:
                rnti, fmt = rng.randrange(0x000B, 0xFFF4), rng.randrange(9)  # a UE nobody has heard of
            else:
                rnti, fmt = rng.choice(ues), rng.choice((FMT0, FMT0, FMT1, FMT1A, FMT1A, FMT1B, FMT1D, FMT2, FMT2, FMT2A, FMT2A, FMT2B))
            n = sizes[fmt]
            bits = [rng.randrange(2) for _ in range(n)]
            if fmt == FMT0:
                bits[0] = 0
            if fmt == FMT1A:
                bits[0] = 1
            if rng.random() < 0.3 and fmt in (FMT1, FMT2, FMT2A):  # full-band type-0 allocations are what real cells send
                na = (nprb + (1 if nprb <= 10 else 2 if nprb <= 26 else 3 if nprb <= 63 else 4) - 1) // (1 if nprb <= 10 else 2 if nprb <= 26 else 3 if nprb <= 63 else 4)
                o0 = 1 if nprb > 10 else 0
                if o0:
                    bits[0] = 0
                for i in range(na):
                    bits[o0 + i] = 1 if rng.random() < 0.8 else 0
            L = rng.randrange(4)
            dcis.append((rnti, fmt, L, rng.randrange(0, 80) // (1 << L) * (1 << L), rng.randrange(0, 40), bits))
        ev.append(("sf", tti // 10, tti % 10, rng.randrange(1, 4), dcis))
    return ev


class _Side:
    def view(self, res):
        return res

    def run(self, life):
        """-> list of per-subframe results (flags, [dl rows], [ul rows], dl map, ul map)"""
        name, nprb, ports, cid, cp, mode, harq, smode, hop, nsf, seed = life
        self.open(nprb, ports, cid, cp, mode, harq, smode, hop)
        out = []
        now = 0
        for e in script(life):
            if e[0] == "mcs":
                self.mcs(e[1], e[2])
            elif e[0] == "harq":
                self.harq(*e[1:])
            else:
                now += 1
                self.now(now)
                out.append(self.subframe(e[1], e[2], e[3], e[4], nprb))
        self.close()
        return out


def _rows(buf, n, w):
    return [tuple(buf[i * w:(i + 1) * w]) for i in range(n)]


class Reference(_Side):
    name = "reference"

    def __init__(self):
        self.lib = C.CDLL(REF_SO)
        o = oracle()
        self.lib.ref_collect_bind.argtypes = [C.c_void_p] * 3
        self.lib.ref_collect_bind(C.cast(o.o_dci_unpack_dl, C.c_void_p), C.cast(o.o_dci_unpack_ul, C.c_void_p), C.cast(o.o_tbs_from_idx, C.c_void_p))
        L = self.lib
        L.ref_collect_new.restype = C.c_void_p
        L.ref_collect_new.argtypes = [C.c_uint32] * 4 + [C.c_int] * 3
        L.ref_collect_free.argtypes = [C.c_void_p]
        L.ref_collect_set_sib2.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.ref_collect_mcs_update.argtypes = [C.c_void_p, C.c_uint16, C.c_int]
        L.ref_collect_harq_update.argtypes = [C.c_void_p, C.c_uint16, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_collect_begin.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.ref_collect_add.argtypes = [C.c_void_p, C.c_uint16, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.ref_collect_end.restype = C.c_uint32
        L.ref_collect_end.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_collect_set_now_ms.argtypes = [C.c_uint64]
        L.ref_collect_rar_grant.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.ref_ulsche_push.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int]
        L.ref_ulsche_get.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int]
        L.ref_ulsche_delete.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.ref_ulsche_ul_tti.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        self.h = None

    def open(self, nprb, ports, cid, cp, mode, harq, smode, hop):
        self.h = self.lib.ref_collect_new(nprb, ports, cid, cp, mode, harq, smode)
        if hop is not None:
            self.lib.ref_collect_set_sib2(self.h, hop, 1)

    def close(self):
        self.lib.ref_collect_free(self.h)
        self.h = None

    def now(self, ms):
        self.lib.ref_collect_set_now_ms(ms)

    def mcs(self, rnti, table):
        self.lib.ref_collect_mcs_update(self.h, rnti, table)

    def harq(self, *a):
        self.lib.ref_collect_harq_update(self.h, *a)

    def subframe(self, sfn, sf_idx, cfi, dcis, nprb):
        self.lib.ref_collect_begin(self.h, sfn, sf_idx, cfi)
        for rnti, fmt, L, ncce, hv, bits in dcis:
            b = (C.c_uint8 * 128)(*bits)
            self.lib.ref_collect_add(self.h, rnti, fmt, L, ncce, hv, b, len(bits))
        return _end(self.lib.ref_collect_end, self.h, nprb)


def _end(fn, h, nprb):
    dl, ul = (C.c_uint32 * (DL_WORDS * 64))(), (C.c_uint32 * (UL_WORDS * 64))()
    md, mu, cnt = (C.c_uint16 * 110)(), (C.c_uint16 * 110)(), (C.c_uint32 * 2)()
    flags = fn(h, dl, 64, ul, 64, md, mu, cnt)
    return (flags, _rows(dl, min(cnt[0], 64), DL_WORDS), _rows(ul, min(cnt[1], 64), UL_WORDS), tuple(md[:nprb]), tuple(mu[:nprb]))


class Oracle(_Side):
    name = "oracle"

    def __init__(self):
        self.o = oracle()
        o = self.o
        o.o_worker_new.restype = C.c_void_p
        o.o_worker_free.argtypes = [C.c_void_p]
        o.o_worker_set_harq.argtypes = [C.c_void_p, C.c_int]
        o.o_worker_set_ul_mode.argtypes = [C.c_void_p, C.c_void_p]
        o.o_worker_collect_begin.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        o.o_worker_collect_add.argtypes = [C.c_void_p, C.c_uint16, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        o.o_worker_collect_end.restype = C.c_uint32
        o.o_worker_collect_end.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        o.o_worker_collect_mcs_update.argtypes = [C.c_void_p, C.c_uint16, C.c_int]
        o.o_worker_collect_harq_update.argtypes = [C.c_void_p, C.c_uint16, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int]
        o.o_worker_collect_set_hop_offset.argtypes = [C.c_void_p, C.c_uint32]

    def open(self, nprb, ports, cid, cp, mode, harq, smode, hop):
        cfg = OWorkerCfg(OCell(nprb, ports, cid, 1, 0, cp), 1, 5, 0.99, 0, mode, 12, 1)
        self.h = self.o.o_worker_new(C.byref(cfg))
        if harq:
            self.o.o_worker_set_harq(self.h, 1)
        if smode:
            from lsn_testlib import OUlCfg
            self.ulcfg = OUlCfg()
            self.o.o_worker_set_ul_mode(self.h, C.byref(self.ulcfg))
        if hop is not None:
            self.o.o_worker_collect_set_hop_offset(self.h, hop)

    def close(self):
        self.o.o_worker_free(self.h)

    def now(self, ms):
        pass

    def mcs(self, rnti, table):
        self.o.o_worker_collect_mcs_update(self.h, rnti, table)

    def harq(self, *a):
        self.o.o_worker_collect_harq_update(self.h, *a)

    def subframe(self, sfn, sf_idx, cfi, dcis, nprb):
        self.o.o_worker_collect_begin(self.h, sfn, sf_idx, cfi)
        for rnti, fmt, L, ncce, hv, bits in dcis:
            b = (C.c_uint8 * 128)(*bits)
            self.o.o_worker_collect_add(self.h, rnti, fmt, L, ncce, hv, b, len(bits))
        return _end(self.o.o_worker_collect_end, self.h, nprb)


def normalise(res):
    """What is NOT compared, and why.  An uplink entry whose conversion failed carries RNTI 0 in its DCI (falcon_dci.c:207,224,229) and is skipped by the PUSCH
    decoder: the reference leaves the grants half-written, the oracle zeroed - only the entry's RNTIs and the RB-map source are kept.  A downlink entry whose grant
    conversion failed is skipped by decode_dl_mode's gate (DL_Sniffer_PDSCH.cc:887-889) unless it is a paging grant: for the others only the head of the row and
    the PRB masks (the RB map reads them) are kept."""
    flags, dl, ul, md, mu = res
    ndl = []
    for r in dl:
        if r[3] == 0 and r[0] != 0xFFFE:
            r = r[:13] + tuple(0 for _ in r[13:])
        if r[3] == 0:  # (`check` of a failed paging conversion: set by the reference on the half-written grant, read by nobody)
            r = r[:13] + (0,) + r[14:]
        if r[2] == 1:  # a 256QAM-table entry: DCICollection.cc:244-249 reads the 64QAM-table grant it did not compute for this entry; `check` is read by nobody
            r = r[:13] + (0,) + r[14:]
        ndl.append(r)
    nul = []
    for r in ul:
        if r[1] == 0:
            r = r[:2] + tuple(0 for _ in r[2:28]) + r[28:]
        elif r[8] >= 29:  # I_MCS 29-31: modulation and size of the previous transmission, none here - the reference's empty last_tb reads as its enum value 0 (tests/ref_grants.py)
            r = r[:14] + (0,) + r[15:23] + (0,) + r[24:]
        nul.append(r)
    return (flags, ndl, nul, md, mu)


def digest(results):
    h = hashlib.sha256()
    for r in results:
        h.update(repr(normalise(r)).encode())
    return h.hexdigest()[:32]


def reference_sources_sha256(ref="/root/reference"):
    h = hashlib.sha256()
    for f in REF_SOURCES:
        h.update(open(os.path.join(ref, f), "rb").read())
    return h.hexdigest()


class Product(_Side):
    """the product's host code: FalconSearch::finishSubframe + the commit-side helpers of lsn_search.h (tests/native/lsn_hosttest.cc: lsnh_collect_*).  It keeps no RB
    maps (only the collision statistics) and has a separate uplink mode: the maps are blanked in view(), LIVES with sniffer_mode = 1 are not run."""
    name = "product"

    def __init__(self):
        from lsn_testlib import hosttest
        self.lib = hosttest()
        L = self.lib
        L.lsnh_collect_new.restype = C.c_void_p
        L.lsnh_collect_new.argtypes = [C.c_uint32] * 4 + [C.c_int] * 2
        L.lsnh_collect_free.argtypes = [C.c_void_p]
        L.lsnh_collect_set_hop_offset.argtypes = [C.c_void_p, C.c_uint32]
        L.lsnh_collect_set_now.argtypes = [C.c_void_p, C.c_uint32]
        L.lsnh_collect_mcs_update.argtypes = [C.c_void_p, C.c_uint16, C.c_int]
        L.lsnh_collect_harq_update.argtypes = [C.c_void_p, C.c_uint16, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int]
        L.lsnh_collect_subframe.restype = C.c_uint32
        L.lsnh_collect_subframe.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]

    def view(self, res):
        flags, dl, ul, md, mu = res
        return (flags, dl, ul, (), ())

    def open(self, nprb, ports, cid, cp, mode, harq, smode, hop):
        assert not smode
        self.h = self.lib.lsnh_collect_new(nprb, ports, cid, cp, mode, harq)
        if hop is not None:
            self.lib.lsnh_collect_set_hop_offset(self.h, hop)

    def close(self):
        self.lib.lsnh_collect_free(self.h)

    def now(self, ms):
        self.lib.lsnh_collect_set_now(self.h, ms)

    def mcs(self, rnti, table):
        self.lib.lsnh_collect_mcs_update(self.h, rnti, table)

    def harq(self, *a):
        self.lib.lsnh_collect_harq_update(self.h, *a)

    def subframe(self, sfn, sf_idx, cfi, dcis, nprb):
        n = len(dcis)
        meta = (C.c_uint32 * (6 * max(n, 1)))()
        bits = (C.c_uint8 * (128 * max(n, 1)))()
        for i, (rnti, fmt, L, ncce, hv, b) in enumerate(dcis):
            meta[6 * i:6 * i + 6] = [rnti, fmt, L, ncce, hv, len(b)]
            bits[128 * i:128 * i + len(b)] = b
        dl, ul, cnt = (C.c_uint32 * (DL_WORDS * 64))(), (C.c_uint32 * (UL_WORDS * 64))(), (C.c_uint32 * 2)()
        flags = self.lib.lsnh_collect_subframe(self.h, sfn, sf_idx, cfi, n, meta, bits, dl, ul, cnt)
        return (flags, _rows(dl, min(cnt[0], 64), DL_WORDS), _rows(ul, min(cnt[1], 64), UL_WORDS), (), ())


# ---- the RAR grant chain (falcon_dci.c:636-683: ul_sniffer_dci_rar_unpack -> ul_sniffer_dci_rar_to_ul_dci -> ul_sniffer_ra_ul_dci_to_grant) ----
def rar_sweep():
    """(nof_prb, cp, n_rb_ho, 20-bit grant): every resource block assignment of six bandwidths on a thinned grid of the other fields, with and without the
    hopping flag"""
    for nprb in (6, 15, 25, 50, 75, 100):
        for ho in (0, 4) if nprb >= 25 else (0, 2):
            for hop in (0, 1):
                for rba in range(0, 1024, 1 if nprb >= 50 else 3):
                    rest = (rba * 7 + nprb) & 0x1FF  # mcs(4) tpc(3) delay(1) csi(1)
                    yield nprb, (rba >> 3) & 1, ho, (hop << 19) | (rba << 9) | rest


def rar_reference(ref, nprb, cp, ho, g20):
    bits = (C.c_uint8 * 20)(*[(g20 >> (19 - i)) & 1 for i in range(20)])
    out = (C.c_int32 * 19)()
    rc = ref.lib.ref_collect_rar_grant(nprb, cp, ho, bits, out)
    head = tuple(out[0:6])
    return head + ((1,) + tuple(out[6:15]) if rc == 0 else (0,) + (0,) * 9)


def rar_oracle(o, nprb, cp, ho, g20):
    class ORar(C.Structure):
        _fields_ = [(n, C.c_uint32) for n in ("rapid", "ta", "hopping", "riv", "mcs", "tpc", "ul_delay", "csi_req")] + \
                   [("t_crnti", C.c_uint16), ("grant_ok", C.c_int)] + [(n, C.c_uint32) for n in ("L_prb", "n_prb", "mcs_idx")] + \
                   [("mod", C.c_int), ("tbs", C.c_int), ("rv", C.c_int), ("n_prb2", C.c_uint32), ("hop", C.c_uint32)]
    cell = OCell(nprb, 1, 0, 1, ho, cp)
    pdu = bytes([0x41, 0x00, 0x10 | ((g20 >> 16) & 0xF), (g20 >> 8) & 0xFF, g20 & 0xFF, 0x12, 0x34])
    r = (ORar * 2)()
    o.o_rar_parse.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int]
    assert o.o_rar_parse(C.byref(cell), pdu, 7, r, 2) == 1 and r[0].t_crnti == 0x1234 and r[0].rapid == 1 and r[0].ta == 1
    e = r[0]
    head = (e.hopping, e.riv, e.mcs, e.tpc, e.ul_delay, e.csi_req)
    if not e.grant_ok:
        return head + (0,) + (0,) * 9
    nsym = 2 * ((6 if cp else 7) - 1)
    return head + (1, e.L_prb, e.n_prb, e.n_prb2 if e.hop == 1 else e.n_prb, e.hop, e.mod if e.L_prb else 0, e.tbs, e.rv, e.mcs_idx, e.L_prb * 12 * nsym)


def rar_product(h, nprb, cp, ho, g20):
    out = (C.c_uint32 * 16)()
    h.lsnh_rar_grant.argtypes = [C.c_uint32] * 4 + [C.c_void_p]
    assert h.lsnh_rar_grant(nprb, cp, ho, g20, out) == 0
    return tuple(out[0:6]) + ((1,) + tuple(out[7:16]) if out[6] else (0,) + (0,) * 9)


# ---- ULSchedule (ULSchedule.cc:11-138) ----
def ulsche_script(seed=5, n=3000, gaps=True):
    """the call pattern of SubframeWorker::run_ul_mode (SubframeWorker.cc:339-349) over a subframe counter that crosses the SFN wrap; with `gaps` it skips
    subframes now and then and restarts elsewhere: ("push", tti, rntis, rar) / ("get", tti, rar) / ("delete", tti, rar)"""
    rng = random.Random(seed)
    tti = 10200
    ev = []
    for _ in range(n):
        tti = (tti + 1 + (gaps and rng.random() < 0.03) * rng.randrange(1, 9)) % 10240
        if gaps and rng.random() < 0.02:  # the counter restarts somewhere else: entries that were never fetched stay behind
            tti = rng.randrange(0, 200)
        ev.append(("push", tti, [rng.randrange(0x100, 0xFFF0) for _ in range(rng.randrange(0, 5))], 0))
        ev.append(("push", tti, [rng.randrange(0x100, 0xFFF0) for _ in range(rng.randrange(0, 2))], 1))
        ev.append(("get", tti, 0))
        ev.append(("get", tti, 1))
        ev.append(("delete", tti, 0))
        ev.append(("delete", tti, 1))
    return ev


def ulsche_reference(ref, script):
    h = ref.lib.ref_collect_new(50, 1, 0, 0, 1, 0, 1)
    out = []
    for e in script:
        if e[0] == "push":
            a = (C.c_uint16 * max(len(e[2]), 1))(*e[2])
            ref.lib.ref_ulsche_push(h, e[1], a, len(e[2]), e[3])
        elif e[0] == "get":
            a = (C.c_uint16 * 64)()
            n = ref.lib.ref_ulsche_get(h, e[1], a, 64, e[2])
            out.append(None if n < 0 else tuple(a[:n]))
        else:
            ref.lib.ref_ulsche_delete(h, e[1], e[2])
    ref.lib.ref_collect_free(h)
    return out


def ulsche_model(script):
    """ULSchedule's semantics said in ten lines: two maps keyed by the tti of the grant; a DCI 0 list pushed for a tti that still has an (unfetched) entry is
    APPENDED to it, a RAR list is kept only if the tti has none (std::map::insert); a fetch at tti t reads (t - 4) mod 10240 / (t - 6) mod 10240"""
    db = {0: {}, 1: {}}
    out = []
    for e in script:
        if e[0] == "push":
            if e[3] == 0:
                db[0].setdefault(e[1], []).extend(e[2])
            else:
                db[1].setdefault(e[1], list(e[2]))
        elif e[0] == "get":
            k = (e[1] - (6 if e[2] else 4)) % 10240
            out.append(tuple(db[e[2]][k]) if k in db[e[2]] else None)
        else:
            db[e[2]].pop((e[1] - (6 if e[2] else 4)) % 10240, None)
    return out


def ulsche_ring(script):
    """what the oracle keeps (o_worker.c: ul_sched / rar_sched, 16 slots by tti % 16, a slot is overwritten when its tti comes round) - the product keeps a map
    bounded to 64 entries (lsn_ulmode.cc), the same thing on any stream without gaps"""
    ring = {0: [None] * 16, 1: [None] * 16}
    out = []
    for e in script:
        if e[0] == "push":
            ring[e[3]][e[1] % 16] = (e[1], list(e[2]))
        elif e[0] == "get":
            k = (e[1] - (6 if e[2] else 4)) % 10240
            slot = ring[e[2]][k % 16]
            out.append(tuple(slot[1]) if slot and slot[0] == k else None)
    return out


def digest_rows(rows):
    h = hashlib.sha256()
    for r in rows:
        h.update(repr(r).encode())
    return h.hexdigest()[:32]
