"""CPU tests (no GPU): the product's HIP-free host logic (DCI sizes/unpack, grants, search space, FALCON search, code
block segmentation, rate-matcher index arithmetic) against the oracle's independent restatement."""
import ctypes as C
import subprocess
import os

import numpy as np
import pytest

from lsn_testlib import (LsnCand, OCell, OGrant, OracleWorker, TxGen, candidate_table, hosttest, oracle, scenario, ROOT,
                         MAX_LOC, MAX_SIZES, CCE_STRIDE)


def test_rm_index_closed_form():
    hosttest()
    out = subprocess.check_output([os.path.join(ROOT, "tests", "native", "_build", "test_rm_index")]).decode()
    assert out.startswith("OK")


@pytest.mark.parametrize("nprb,ports", [(6, 1), (15, 2), (25, 1), (25, 2), (50, 1), (50, 2), (75, 1), (75, 2), (100, 1), (100, 2),
                                        (6, 4), (15, 4), (25, 4), (50, 4), (75, 4), (100, 4)])
def test_dci_sizes(nprb, ports):
    h, o = hosttest(), oracle()
    cell = OCell(nprb, ports, 1, 1)
    for f in range(9):
        assert h.lsnh_dci_format_sizeof(nprb, ports, f) == o.o_dci_format_sizeof(C.byref(cell), f), f


def test_search_space_closed_form_vs_enumeration():
    h, o = hosttest(), oracle()
    rng = np.random.default_rng(5)
    for cces in ([20, 54, 87], [10, 26, 43], [2, 5, 9], [4, 12, 21]):
        arr = (C.c_uint32 * 3)(*cces)
        rntis = list(rng.integers(0, 65536, 400)) + [0, 1, 2, 9, 10, 11, 0xFFF3, 0xFFF4, 0xFFFC, 0xFFFD, 0xFFFE, 0xFFFF]
        for cfi in (1, 2, 3):
            n = cces[cfi - 1]
            for rnti in rntis:
                for l in range(4):
                    L = 1 << l
                    for ncce in range(0, min(n, 84) - L + 1, L):
                        sf = int(rng.integers(0, 10))
                        a = h.lsnh_validate_location(arr, cfi, ncce, l, sf, int(rnti))
                        b = o.o_validate_location(n, ncce, l, sf, int(rnti))
                        assert a == b, (cces, cfi, ncce, l, sf, rnti, a, b)
                        assert a == h.lsnh_validate_location_enum(n, ncce, l, sf, int(rnti))


def _oracle_grant_api():
    o = oracle()
    o.o_dci_unpack_dl.argtypes = [C.POINTER(OCell), C.c_void_p, C.c_uint32, C.c_int, C.c_uint16, C.c_void_p]
    o.o_ra_dl_dci_to_grant.argtypes = [C.POINTER(OCell), C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.POINTER(OGrant)]
    o.o_config_mimo.argtypes = [C.POINTER(OCell), C.c_int, C.c_void_p, C.POINTER(OGrant)]
    o.o_dci_unpack_ul.argtypes = [C.POINTER(OCell), C.c_void_p, C.c_uint32, C.c_uint16, C.c_void_p]
    o.o_ra_ul_dci_to_grant.argtypes = [C.POINTER(OCell), C.c_void_p, C.c_void_p]
    return o


@pytest.mark.parametrize("nprb,ports", [(100, 2), (75, 2), (50, 1), (25, 2), (6, 1), (15, 2), (100, 4), (25, 4), (6, 4)])
def test_dl_grants_random_payloads(nprb, ports):
    """random DCI payloads of every DL format -> identical unpack verdict, PRB set, TBS/modulation, nof_re, MIMO config"""
    h, o = hosttest(), _oracle_grant_api()
    rng = np.random.default_rng(nprb * 10 + ports)
    cell = OCell(nprb, ports, 3, 1)
    dci = (C.c_uint8 * 512)()
    n_ok = 0
    for it in range(1500):
        fmt = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8]))
        nb = h.lsnh_dci_format_sizeof(nprb, ports, fmt)
        payload = rng.integers(0, 2, nb).astype(np.uint8)
        if fmt == 2:
            payload[0] = 1
        rnti = int(rng.choice([0xFFFF, 0xFFFE, 2, 5, 0x46, 0x1234, int(rng.integers(11, 0xFFF3))]))
        sf_idx, cfi, alt = int(rng.integers(0, 10)), int(rng.integers(1, 4)), int(rng.integers(0, 2))
        C.memset(dci, 0, 512)
        g_o, g_h = OGrant(), OGrant()
        dci_view = (C.c_uint32 * 4).from_buffer(dci)
        u_ok = o.o_dci_unpack_dl(C.byref(cell), payload.ctypes.data, nb, fmt, rnti, dci) == 0
        r_o = 0
        if u_ok:
            r_o = 1
            if o.o_ra_dl_dci_to_grant(C.byref(cell), sf_idx, cfi, alt, dci, C.byref(g_o)) == 0:
                r_o |= 2
                r_o |= o.o_config_mimo(C.byref(cell), fmt, dci, C.byref(g_o)) << 8
        r_h = h.lsnh_dl_grant(nprb, ports, 3, sf_idx, cfi, alt, payload.ctypes.data, nb, fmt, rnti, 1, C.byref(g_h))
        assert r_h == r_o, (fmt, rnti, r_h, r_o)
        if r_o & 2:
            n_ok += 1
            assert bytes(g_h) == bytes(g_o), (fmt, rnti, sf_idx, cfi, alt)
    assert n_ok > 300


def test_ul_grants_random_payloads():
    h, o = hosttest(), _oracle_grant_api()
    rng = np.random.default_rng(9)
    for nprb in (25, 50, 75, 100):
        cell = OCell(nprb, 2, 1, 1)
        nb = h.lsnh_dci_format_sizeof(nprb, 2, 0)
        for it in range(500):
            payload = rng.integers(0, 2, nb).astype(np.uint8)
            payload[0] = 0
            dci = (C.c_uint8 * 256)()
            og = (C.c_uint32 * 8)()
            hg = (C.c_uint32 * 6)()
            r_o = 0
            if o.o_dci_unpack_ul(C.byref(cell), payload.ctypes.data, nb, 0x100, dci) == 0:
                r_o = 1
                if o.o_ra_ul_dci_to_grant(C.byref(cell), dci, og) == 0:
                    r_o = 3
            r_h = h.lsnh_ul_grant(nprb, 2, payload.ctypes.data, nb, 0x100, hg)
            assert r_h == r_o
            if r_o == 3:  # o_pusch_grant_t {L_prb, n_prb, mcs_idx, mod, tbs, rv}
                assert list(hg) == list(og)[:6]


def test_ul_grants_with_frequency_hopping_offsets():
    """DCI 0 with the hopping flag: hop bits (36.213 Tables 8.4-1/2), type-1 second-slot position for several pusch-HoppingOffset values,
    type 2 flagged - product == oracle on random payloads, and a hand-computed case"""
    h, o = hosttest(), _oracle_grant_api()
    h.lsnh_ul_grant_hop.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint16, C.c_void_p]
    rng = np.random.default_rng(19)
    seen = set()
    for nprb in (25, 50, 100):
        nb = h.lsnh_dci_format_sizeof(nprb, 2, 0)
        for off in (0, 3, 8, 20):
            cell = OCell(nprb, 2, 1, 1, off)
            for it in range(400):
                payload = rng.integers(0, 2, nb).astype(np.uint8)
                payload[0] = 0
                payload[1] = 1  # hopping flag
                dci = (C.c_uint8 * 256)()
                og = (C.c_uint32 * 8)()
                hg = (C.c_uint32 * 8)()
                r_o = 0
                if o.o_dci_unpack_ul(C.byref(cell), payload.ctypes.data, nb, 0x100, dci) == 0:
                    r_o = 1
                    if o.o_ra_ul_dci_to_grant(C.byref(cell), dci, og) == 0:
                        r_o = 3
                assert h.lsnh_ul_grant_hop(nprb, 2, off, payload.ctypes.data, nb, 0x100, hg) == r_o
                if r_o == 3:
                    assert list(hg) == list(og)
                    seen.add((nprb, int(og[7])))
                    if og[7] == 1:
                        assert og[6] + og[0] <= nprb and og[6] != og[1] or og[0] == 0
    assert {(25, 1), (25, 2), (100, 1), (100, 2)} <= seen
    # 100 PRB, offset 10: n_rb_pusch = 90; hop bits 10 (= +N/2), L = 4 at PRB 12 -> slot 1 at (45 + 12) % 90 = 57
    nb = h.lsnh_dci_format_sizeof(100, 2, 0)
    riv = 100 * (4 - 1) + 12
    bits = [0, 1, 1, 0] + [int(b) for b in format(riv, "011b")] + [0] * (nb - 15)
    hg = (C.c_uint32 * 8)()
    pl = np.array(bits, dtype=np.uint8)
    assert h.lsnh_ul_grant_hop(100, 2, 10, pl.ctypes.data, nb, 0x100, hg) == 3 and (hg[0], hg[1], hg[6], hg[7]) == (4, 12, 57, 1)


def test_cbsegm_all_tbs():
    h, o = hosttest(), oracle()

    class Seg(C.Structure):
        _fields_ = [(n, C.c_int) for n in ("C", "Cp", "Cm", "Kp", "Km", "F", "tbs")]
    o.o_cbsegm.argtypes = [C.POINTER(Seg), C.c_int]
    o.o_tbs_from_idx.restype = C.c_int
    seen = set()
    for itbs in range(0, 34):
        for nprb in range(1, 111):
            seen.add(o.o_tbs_from_idx(itbs, nprb))
    seen.discard(-1)
    assert len(seen) > 150
    for tbs in sorted(seen):
        s = Seg()
        out = (C.c_int * 6)()
        ro, rh = o.o_cbsegm(C.byref(s), tbs), h.lsnh_cbsegm(tbs, out)
        assert (ro == 0) == (rh == 0)
        if ro == 0:
            assert list(out) == [s.C, s.Cp, s.Cm, s.Kp, s.Km, s.F], tbs


def _search_parity(scn, nsf, seed, update_meta_period=0, **over):
    """FALCON search of the product over oracle-decoded candidate tables == the oracle worker's own search."""
    h = hosttest()
    sc = scenario(scn, seed=seed, **over)
    tx = TxGen(**sc)
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"], sc["phich_ng_x6"])
    regs_cce = None
    hs = None
    total = 0
    for i in range(nsf):
        tti, iq, pdus = tx.next()
        upd = 1 if (update_meta_period and i % update_meta_period == 0) else 0
        ow.work(iq, tti, update_meta=upd)
        if hs is None:
            # nof_cce per CFI from the oracle's REG tables
            from lsn_testlib import oracle as _o

            class Regs(C.Structure):
                _fields_ = [("nof_regs", C.c_uint32 * 3), ("nof_cce", C.c_uint32 * 3), ("k0", (C.c_uint16 * 800) * 3),
                            ("l", (C.c_uint8 * 800) * 3), ("pcfich_k0", C.c_uint16 * 4), ("ngroups_phich", C.c_uint32)]
            regs = Regs()
            cell = OCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["phich_ng_x6"])
            _o().o_regs_init.argtypes = [C.POINTER(OCell), C.c_void_p]
            _o().o_regs_init(C.byref(cell), C.byref(regs))
            regs_cce = (C.c_uint32 * 3)(*regs.nof_cce)
            hs = h.lsnh_search_new(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], regs_cce, 5, 0.99, 0)
            sizes = [h.lsnh_search_size(hs, k) for k in range(h.lsnh_search_nof_sizes(hs))]
        cfi = ow.cfi()
        cand, pw = candidate_table(ow.llr(), regs_cce[cfi - 1], sizes, tti % 10)
        out = (C.c_uint32 * (64 * 6))()
        n = h.lsnh_search_run(hs, tti, cfi, float(ow.chest().snr_db), cand, pw.ctypes.data, upd, out, 64 * 6)
        got = [tuple(out[6 * k:6 * k + 6]) for k in range(n)]
        exp = ow.accepted()
        assert got == exp, (i, got, exp)
        # RAR feedback (decode results are the oracle's here): keep the RNTI managers in step
        total += n
        assert h.lsnh_search_nof_active(hs) <= ow.nof_active() + 64
    st = (C.c_uint32 * 7)()
    h.lsnh_search_stats(hs, st)
    os_ = ow.stats()
    assert list(st) == [os_.nof_decoded_locations, os_.nof_cce, os_.nof_missed_cce, os_.nof_subframes,
                        os_.nof_subframe_collisions_dw, os_.nof_subframe_collisions_up, os_.nof_locations]
    h.lsnh_search_free(hs)
    assert total > 0
    return total


def test_falcon_search_small_cell():
    _search_parity("small", 30, seed=4)


def test_falcon_search_cfg1():
    _search_parity("cfg1", 25, seed=1)


def test_falcon_search_15mhz_cell():
    _search_parity("cfg2", 12, seed=6, nof_prb=75, cell_id=77, n_rnti=20)


def test_falcon_search_four_port_cells():
    """the product's host search on four-port candidate tables (four-port DCI sizes of formats 2 / 2A, 76 CCEs at 20 MHz) == the oracle worker's"""
    _search_parity("cfg2", 14, seed=7, nof_ports=4, cfi=0, n_rnti=12)
    _search_parity("cfg3", 12, seed=8, nof_ports=4, nof_prb=50, n_rnti=20, rar_period=0)


def test_falcon_search_cfg3_meta_update():
    _search_parity("cfg3", 16, seed=3, update_meta_period=8, rar_period=0)
