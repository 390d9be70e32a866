"""Driver of oracle/_ref/libref_falcon_search.so: the REFERENCE'S OWN DCISearch.cc / falcon_pdcch.c / MetaFormats.cc / RNTIManager.cc compiled from
/root/reference (oracle/Makefile.ref, stand-in srsRAN types under oracle/ref_shim_search/), run subframe by subframe on the ORACLE's PDCCH soft bits
with the oracle's two DSP primitives (DCI size of a format, de-rate-matching + tail-biting Viterbi + CRC remainder of one candidate) bound in.
What is compared is everything the search decides: the accepted DCI of every subframe - (RNTI, format, L, nCCE, bits, histogram value), in the
order DCICollection::addCandidate receives them - the search statistics, the primary / secondary format lists and probes of the RNTI manager.
Test infrastructure only (tests/test_ref_dci_search.py, tests/golden/make_dci_search_fixture.py)."""
import ctypes as C
import hashlib
import os

import numpy as np

from lsn_testlib import OCell, OracleWorker, oracle, parse_pcap, scenario
from parity import gen_capture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_falcon_search.so")
REF_SOURCES = ["src/src/DCISearch.cc", "lib/src/phy/falcon_phch/falcon_pdcch.c", "src/src/MetaFormats.cc", "src/src/SubframeInfo.cc",
               "lib/src/util/RNTIManager.cc", "lib/src/util/Histogram.cc", "lib/src/util/Interval.cc"]

# the streams: (name, scenario, subframes in the suite, subframes of the long run recorded in the fixture, meta-format period, worker options)
CASES = [
    ("cfg3_100prb_150rnti_rar", dict(name="cfg3", seed=3), 200, 20000, 500, {}),   # long run = the whole distinct capture of the gated bench stream (tools/make_cfg3_golden.py)
    ("cfg2_100prb_32rnti", dict(name="cfg2", seed=2), 150, 2000, 500, {}),
    ("cfg3_threshold_8_split_0.8", dict(name="cfg3", seed=5), 120, 1500, 200, dict(threshold=8, split_ratio=0.8)),
    ("cfg3_skip_secondary_no_shortcut", dict(name="cfg3", seed=6), 150, 1000, 100, dict(skip_secondary=1, enable_shortcut=0, split_ratio=0.6)),
    ("small_25prb_16dB", dict(name="small", seed=7, snr_db=16.0, rar_period=50), 400, 4000, 100, {}),
    ("cfg3_50prb_four_ports", dict(name="cfg3", seed=8, nof_prb=50, nof_ports=4, n_rnti=40, dl_min=3, dl_max=6, ul_min=1, ul_max=3), 200, 2000, 100, {}),
    ("cfg1_50prb_low_snr_gate", dict(name="cfg1", seed=9, snr_db=4.0), 200, 1000, 100, {}),
    ("cfg3_15prb_cfi_small_region", dict(name="cfg3", seed=10, nof_prb=15, n_rnti=12, dl_min=1, dl_max=3, ul_min=0, ul_max=2, rar_period=40), 500, 4000, 100, {}),
    ("small_6prb_cfi_varies", dict(name="small", seed=31, nof_prb=6, n_rnti=3, dl_min=1, dl_max=1, ul_min=0, ul_max=1, cfi=0, rar_period=30), 600, 4000, 100, {}),
    ("cfg3_75prb_cfi_varies", dict(name="cfg3", seed=32, nof_prb=75, n_rnti=60, dl_min=4, dl_max=8, ul_min=1, ul_max=4, cfi=0, rar_period=100), 120, 1500, 100, {}),
]
PRODUCT_SUBFRAMES = {100: 10 ** 6, 75: 10 ** 6, 50: 10 ** 6, 25: 10 ** 6, 15: 10 ** 6, 6: 10 ** 6}  # the product's host search walks every stream to its end (candidate tables built in C)


def rar_temp_crntis(pdu):
    """temporary C-RNTIs of a MAC RAR PDU (TS 36.321 6.1.5), walked like srsran::rar_pdu (DL_Sniffer_PDSCH.cc:782-797): a sub-header without
    RAPID (backoff indicator) activates RNTI 0, as the reference does"""
    is_rapid, pos = [], 0
    while pos < len(pdu) and len(is_rapid) < 32:
        b = pdu[pos]
        pos += 1
        is_rapid.append(bool(b & 0x40))
        if not (b & 0x80):
            break
    out = []
    for r in is_rapid:
        t = 0
        if r:
            if pos + 6 > len(pdu):
                break
            t = (pdu[pos + 4] << 8) | pdu[pos + 5]
            pos += 6
        out.append(t)
    return out


class RefSearch:
    def __init__(self, ow, threshold=5, split_ratio=0.99, skip_secondary=0, enable_shortcut=1):
        self.lib = C.CDLL(REF_SO)
        L = self.lib
        L.ref_search_new.restype = C.c_void_p
        L.ref_search_new.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_int, C.c_int]
        L.ref_search_free.argtypes = [C.c_void_p]
        L.ref_search_bind.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_search_subframe.restype = C.c_int
        L.ref_search_subframe.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_uint32]
        L.ref_search_activate_rar.argtypes = [C.c_void_p, C.c_uint16]
        L.ref_search_meta_formats.restype = C.c_uint32
        L.ref_search_meta_formats.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_search_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_search_rnti_frequency.restype = C.c_uint32
        L.ref_search_rnti_frequency.argtypes = [C.c_void_p, C.c_uint16, C.c_uint32]
        L.ref_search_rnti_reason.restype = C.c_int
        L.ref_search_rnti_reason.argtypes = [C.c_void_p, C.c_uint16]
        L.ref_search_validate_location.restype = C.c_uint32
        L.ref_search_validate_location.argtypes = [C.c_uint32] * 4 + [C.c_uint16]
        o = oracle()
        self._cell = OCell(ow.cfg.cell.nof_prb, ow.cfg.cell.nof_ports, ow.cfg.cell.id, ow.cfg.cell.phich_ng_x6, 0, ow.cfg.cell.cp)
        L.ref_search_bind(C.cast(o.o_dci_format_sizeof, C.c_void_p), C.cast(o.o_dci_decode, C.c_void_p), C.addressof(self._cell))
        self.h = L.ref_search_new(self._cell.nof_prb, self._cell.nof_ports, self._cell.id, threshold, split_ratio, skip_secondary, enable_shortcut)

    def close(self):
        if self.h:
            self.lib.ref_search_free(self.h)
            self.h = None

    def subframe(self, llr, cfi, tti, snr_db, update_meta):
        """-> None when the search did not run (SNR gate), else the accepted DCI [(rnti, format, L, ncce, bits, histval)]"""
        llr = np.ascontiguousarray(llr, dtype=np.float32)
        out = (C.c_uint32 * (64 * 6))()
        n = self.lib.ref_search_subframe(self.h, llr.ctypes.data, len(llr) // 72, cfi, tti % 10, (tti // 10) % 1024, float(snr_db), int(update_meta), out, None, 64)
        if n < 0:
            return None
        return [tuple(out[6 * i:6 * i + 6]) for i in range(min(n, 64))]

    def activate_rar(self, t_crnti):
        self.lib.ref_search_activate_rar(self.h, t_crnti)

    def meta_formats(self):
        buf = (C.c_uint32 * 18)()
        r = self.lib.ref_search_meta_formats(self.h, buf)
        return list(buf[:r & 0xFF]), list(buf[9:9 + (r >> 8)])

    def stats(self):
        """(nof_locations, nof_decoded_locations, nof_cce, nof_missed_cce, nof_subframes)"""
        buf = (C.c_uint32 * 5)()
        self.lib.ref_search_stats(self.h, buf)
        return tuple(buf)

    def frequency(self, rnti, f):
        return self.lib.ref_search_rnti_frequency(self.h, rnti, f)

    def reason(self, rnti):
        return self.lib.ref_search_rnti_reason(self.h, rnti)


def case_capture(case, nsf=None):
    name, sc_kw, nsf_suite, nsf_long, meta, okw = case
    kw = dict(sc_kw)
    sc = scenario(kw.pop("name"), **kw)
    tti0, iq = gen_capture(sc, nsf or nsf_suite, threads=4)
    return sc, tti0, iq


class ProductSearch:
    """the PRODUCT's host search (lsn_search.cc: FalconSearch, through tests/native/liblsn_hosttest.so) on candidate tables decoded by the oracle's
    candidate decoder - what k_viterbi + k_cce_power hand it on the GPU"""

    def __init__(self, sc, threshold=5, split_ratio=0.99, skip_secondary=0, enable_shortcut=1):
        from lsn_testlib import hosttest

        class Regs(C.Structure):
            _fields_ = [("nof_regs", C.c_uint32 * 3), ("nof_cce", C.c_uint32 * 3), ("k0", (C.c_uint16 * 800) * 3),
                        ("l", (C.c_uint8 * 800) * 3), ("pcfich_k0", C.c_uint16 * 4), ("ngroups_phich", C.c_uint32)]
        self.h = hosttest()
        regs = Regs()
        cell = OCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["phich_ng_x6"], 0, sc.get("cp", 0))
        oracle().o_regs_init.argtypes = [C.POINTER(OCell), C.c_void_p]
        oracle().o_regs_init(C.byref(cell), C.byref(regs))
        self.regs_cce = (C.c_uint32 * 3)(*regs.nof_cce)
        self.hs = self.h.lsnh_search_new(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], self.regs_cce, threshold, split_ratio, skip_secondary)
        self.sizes = [self.h.lsnh_search_size(self.hs, k) for k in range(self.h.lsnh_search_nof_sizes(self.hs))]
        self.h.lsnh_search_set_shortcut_discovery.argtypes = [C.c_void_p, C.c_int]
        self.h.lsnh_search_set_shortcut_discovery(self.hs, int(enable_shortcut))

    def table(self, llr, cfi, tti):
        """lsn_testlib.candidate_table() through its C twin in the host-test glue (tests/native/lsn_hosttest.cc: lsnh_candidate_table) -> (cand, ccepow)"""
        from lsn_testlib import CCE_STRIDE, MAX_LOC, MAX_SIZES, LsnCand
        o = oracle()
        cand = (LsnCand * (MAX_LOC * MAX_SIZES))()
        pw = np.zeros(CCE_STRIDE, dtype=np.float32)
        llr = np.ascontiguousarray(llr, dtype=np.float32)
        sizes = (C.c_uint32 * len(self.sizes))(*self.sizes)
        self.h.lsnh_candidate_table.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.h.lsnh_candidate_table(llr.ctypes.data, self.regs_cce[cfi - 1], sizes, len(self.sizes), tti % 10, C.cast(o.o_dci_decode, C.c_void_p),
                                    C.cast(o.o_validate_location, C.c_void_p), cand, pw.ctypes.data)
        return cand, pw

    def subframe(self, llr, cfi, tti, snr_db, update_meta):
        cand, pw = self.table(llr, cfi, tti)
        out = (C.c_uint32 * (64 * 6))()
        n = self.h.lsnh_search_run(self.hs, tti, cfi, float(snr_db), cand, pw.ctypes.data, int(update_meta), out, 64 * 6)
        if not snr_db > 6.0:
            return None
        return [tuple(out[6 * k:6 * k + 6]) for k in range(n)]

    def activate_rar(self, t):
        self.h.lsnh_search_activate_rar(self.hs, t)

    def close(self):
        self.h.lsnh_search_free(self.hs)


def sf_digest(i, a):
    return hashlib.sha256(("%d:%s;" % (i, "-" if a is None else ",".join("%d.%d.%d.%d.%d.%d" % t for t in a))).encode()).hexdigest()[:10]


def digest(per_sf):
    h = hashlib.sha256()
    for i, a in enumerate(per_sf):
        h.update(sf_digest(i, a).encode())
    return h.hexdigest()[:32]


def walk(case, nsf=None, with_reference=False, product_subframes=0):
    """the oracle worker over the case's stream, subframe by subframe; next to it, on the oracle's soft bits: the reference's own search
    (with_reference, needs the library) and the product's host search over the first product_subframes subframes.
    -> dict(per_sf: accepted list (None = SNR gate) per subframe for "oracle" / "reference" / "product", stats, digests, llr_sha256)"""
    name, sc_kw, nsf_suite, nsf_long, meta, okw = case
    nsf = nsf or nsf_suite
    sc, tti0, iq = case_capture(case, nsf)
    skw = {k: v for k, v in okw.items() if k in ("threshold", "split_ratio", "skip_secondary", "enable_shortcut")}
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"], sc["phich_ng_x6"], cp=sc.get("cp", 0), **okw)
    ref = RefSearch(ow, **skw) if with_reference else None
    prod = ProductSearch(sc, **skw) if product_subframes else None
    o_sf, r_sf, p_sf, llr_hash, searched = [], [], [], hashlib.sha256(), 0
    probes = dict(dci0_of_rar_rntis=0, dci0=0, formats=[0] * 9, levels=[0] * 4)  # of the reference's accepted DCI
    for i in range(nsf):
        tti = tti0 + i
        upd = 1 if i % meta == 0 else 0
        ow.work(iq[i], tti, update_meta=upd)
        llr, cfi, snr = ow.llr(), ow.cfi(), ow.chest().snr_db
        ran = snr > 6.0  # DCISearch.cc:566
        o_sf.append(ow.accepted() if ran else None)
        searched += 1 if ran else 0
        llr_hash.update(llr.tobytes())
        recs = parse_pcap(ow.pcap_bytes())
        old = ow.pcap
        ow.pcap = ow.lib.o_pcap_open_mem()
        ow.lib.o_worker_set_pcap(ow.h, ow.pcap)
        ow.lib.o_pcap_close(old)
        # random-access responses decoded in this subframe activate their temporary C-RNTIs - after the search (DL_Sniffer_PDSCH.cc:782-797)
        rars = [t for r in recs if r["rnti_type"] == 2 and r["direction"] == 1 for t in rar_temp_crntis(r["pdu"])]
        if ref:
            r_sf.append(ref.subframe(llr, cfi, tti, snr, upd))
            for a in r_sf[-1] or []:
                probes["formats"][a[1]] += 1
                probes["levels"][a[2]] += 1
                if a[1] == 0:
                    probes["dci0"] += 1
                    probes["dci0_of_rar_rntis"] += 1 if ref.reason(a[0]) == 2 else 0  # the temp_dci0 path of DCISearch.cc:139-158 / 422-432
            for t in rars:
                ref.activate_rar(t)
        if prod and i < product_subframes:
            p_sf.append(prod.subframe(llr, cfi, tti, snr, upd))
            for t in rars:
                prod.activate_rar(t)
    st = ow.stats()
    rm = ow.lib.o_worker_rntiman(ow.h)
    ow.lib.o_rntiman_get_activation_reason.argtypes = [C.c_void_p, C.c_uint16]
    reasons = [0] * 6
    for rnti in range(65536):
        reasons[ow.lib.o_rntiman_get_activation_reason(rm, rnti)] += 1
    res = dict(case=name, scenario=sc, subframes=nsf, searched=searched, llr_sha256=llr_hash.hexdigest()[:32],
               oracle=dict(per_sf=o_sf, digest=digest(o_sf), accepted=sum(len(a) for a in o_sf if a), reasons=reasons,
                           stats=[st.nof_locations, st.nof_decoded_locations, st.nof_cce, st.nof_missed_cce, st.nof_subframes]))
    if ref:
        rr = [0] * 6
        for rnti in range(65536):
            rr[ref.reason(rnti)] += 1
        res["reference"] = dict(per_sf=r_sf, digest=digest(r_sf), accepted=sum(len(a) for a in r_sf if a), stats=list(ref.stats()),
                                meta_final=[list(x) for x in ref.meta_formats()], reasons=rr, probes=probes)
        ref.close()
    if prod:
        res["product"] = dict(per_sf=p_sf)
        prod.close()
    return res


def first_difference(a_sf, b_sf):
    for i, (a, b) in enumerate(zip(a_sf, b_sf)):
        if a != b:
            return i, a, b
    return None


def reference_sources_sha256(ref="/root/reference"):
    h = hashlib.sha256()
    for f in REF_SOURCES:
        h.update(open(os.path.join(ref, f), "rb").read())
    return h.hexdigest()
