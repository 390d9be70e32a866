"""CPU tests of the PSS / SSS cell-search restatement (oracle/o_sync.c): 36.211 6.11 known answers, then the synthetic eNB
(tools/txgen) -> search loop-back at unknown timing, cell id and carrier offset."""
import ctypes as C

import numpy as np
import pytest

from lsn_testlib import OSync, OSyncCfg, TxGen, oracle, oracle_sync_api, scenario, sync_capture


def test_sss_m0_m1_table_entries():
    # 36.211 Table 6.11.2.1-1 (first row, row ends, last row)
    o = oracle_sync_api()
    want = {0: (0, 1), 1: (1, 2), 29: (29, 30), 30: (0, 2), 58: (28, 30), 59: (0, 3), 86: (27, 30), 87: (0, 4), 113: (26, 30), 114: (0, 5), 164: (24, 30), 165: (0, 7), 166: (1, 8), 167: (2, 9)}
    for n1, (m0, m1) in want.items():
        a, b = C.c_uint32(), C.c_uint32()
        o.o_sss_m0m1(n1, C.byref(a), C.byref(b))
        assert (a.value, b.value) == (m0, m1), n1
    pairs = set()
    for n1 in range(168):
        a, b = C.c_uint32(), C.c_uint32()
        o.o_sss_m0m1(n1, C.byref(a), C.byref(b))
        assert a.value < b.value < 31
        pairs.add((a.value, b.value))
    assert len(pairs) == 168


def test_pss_is_a_zadoff_chu_sequence_with_the_three_roots():
    o = oracle_sync_api()
    for n2, u in enumerate((25, 29, 34)):
        d = np.zeros(62, dtype=np.complex64)
        o.o_pss_seq(n2, d.ctypes.data)
        n = np.arange(63)
        zc = np.exp(-1j * np.pi * u * n * (n + 1) / 63.0)
        assert np.allclose(d, np.delete(zc, 31), atol=1e-6)  # the length-63 sequence with the DC element punctured
    # roots 29 and 34 are complex conjugates of each other in time (u2 = 63 - u1)
    a, b = np.zeros(62, np.complex64), np.zeros(62, np.complex64)
    o.o_pss_seq(1, a.ctypes.data)
    o.o_pss_seq(2, b.ctypes.data)
    assert np.allclose(a, np.conj(b), atol=1e-6)
    # unit-energy time replica whose DFT has the sequence on carriers -31..-1, 1..31 and nothing elsewhere
    p = np.zeros(512, dtype=np.complex64)
    o.o_pss_time(0, 512, p.ctypes.data)
    assert abs(np.sum(np.abs(p) ** 2) - 1.0) < 1e-5
    P = np.fft.fft(p)
    d = np.zeros(62, dtype=np.complex64)
    o.o_pss_seq(0, d.ctypes.data)
    sc = np.sqrt(512.0 / 62.0)
    assert np.allclose(P[512 - 31:], d[:31] * sc, atol=1e-4) and np.allclose(P[1:32], d[31:] * sc, atol=1e-4)
    assert np.max(np.abs(P[32:512 - 31])) < 1e-4 and abs(P[0]) < 1e-4


def test_sss_sequences_are_binary_distinct_and_differ_between_the_half_frames():
    o = oracle_sync_api()
    seqs = np.zeros((3, 168, 2, 62), dtype=np.int8)
    for n2 in range(3):
        for n1 in range(168):
            for h in range(2):
                o.o_sss_seq(n1, n2, h, seqs[n2, n1, h].ctypes.data)
    assert set(np.unique(seqs)) == {-1, 1}
    flat = seqs.reshape(3 * 168 * 2, 62).astype(np.int32)
    g = flat @ flat.T
    np.fill_diagonal(g, 0)
    assert g.max() < 62  # no two of the 1008 sequences coincide
    # cell 0, subframe 0, first elements by hand: m0 = 0, m1 = 1: d(0) = s~(0) c~(0) = 1 * 1, d(1) = s~(1) c~(3) z~(0) = 1 * 1 * 1
    assert seqs[0, 0, 0, 0] == 1 and seqs[0, 0, 0, 1] == 1
    # the even elements of subframe 5 use s1 where subframe 0 uses s0
    assert not np.array_equal(seqs[0, 0, 0], seqs[0, 0, 1])


@pytest.mark.parametrize("scn,over,lead,cfo,periods,force", [
    ("small", dict(cell_id=301), 1234, 0.0, 1, -1),
    ("small", dict(cell_id=2, nof_ports=1, nof_rx=1, snr_db=5.0), 38000, 900.0, 2, -1),
    ("small", dict(cell_id=503, nof_prb=6), 77, -2500.0, 2, -1),
    ("cfg1", dict(cell_id=150), 60001, 0.0, 2, 0),  # the first 5 ms hold no PSS: found in the second period
])
def test_cell_search_finds_id_timing_and_cfo(scn, over, lead, cfo, periods, force):
    sc = scenario(scn, seed=5, start_tti=10 * 77 + 3, cfo_hz=cfo, **over)
    x, first_tti = sync_capture(sc, lead, periods)
    r, s, corr = oracle_cell_search(x, sc["nof_prb"], periods, force, 20.0)
    N = {6: 128, 15: 256, 25: 512, 50: 1024, 75: 1536, 100: 2048}[sc["nof_prb"]]
    sflen, w5 = 15 * N, 75 * N
    assert r == 1 and s.found and s.cell_id == sc["cell_id"] and s.n_id_2 == sc["cell_id"] % 3 and s.n_id_1 == sc["cell_id"] // 3
    # subframe first_tti starts at sample `lead`; the reported boundary is the first sf-0/5 start in the buffer
    k = next(k for k in range(10) if (first_tti + k) % 5 == 0)
    want_start = (lead + k * sflen) % w5
    want_sf = (first_tti + (want_start - lead) // sflen) % 10
    assert s.sf_start == want_start and s.sf_idx == want_sf, (s.sf_start, want_start, s.sf_idx, want_sf)
    assert abs(s.cfo_hz - cfo) < 150.0, s.cfo_hz
    assert s.sss_metric > 2.0 * s.sss_second  # sequences sharing m0 or m1 reach a quarter of the peak
    assert s.pss_p2avg >= 20.0 and corr[s.n_id_2, s.pss_pos] == s.pss_peak == corr.max()


def test_cell_search_rejects_noise_and_short_buffers():
    rng = np.random.default_rng(3)
    n = 2 * 75 * 128 + 128
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    r, s, _ = oracle_cell_search(x, 6, 1, -1, 20.0)
    assert r == 0 and not s.found and s.pss_p2avg < 20.0
    r, _, _ = oracle_cell_search(x[:-1], 6, 1, -1, 20.0)
    assert r == -1
    r, _, _ = oracle_cell_search(x, 70, 1, -1, 20.0)
    assert r == -1
    r, _, _ = oracle_cell_search(x, 6, 1, 3, 20.0)
    assert r == -1


def oracle_cell_search(x, nof_prb, periods, force, threshold):
    o = oracle_sync_api()
    N = {6: 128, 15: 256, 25: 512, 50: 1024, 75: 1536, 100: 2048}.get(nof_prb, 128)
    corr = np.zeros((3, 75 * N), dtype=np.float32)
    cfg = OSyncCfg(periods, force, threshold)
    s = OSync()
    x = np.ascontiguousarray(x, dtype=np.complex64)
    r = o.o_cell_search(x.ctypes.data, x.size, nof_prb, C.byref(cfg), C.byref(s), corr.ctypes.data)
    return r, s, corr


def test_cell_search_detects_the_cyclic_prefix():
    """the SSS symbol sits N + 144 (x N / 2048) samples in front of the PSS symbol with the normal CP, N + 512 with the extended one: the search tries both
    and reports which carried the better SSS, the subframe boundary follows from it"""
    for cp, nprb, cid, lead in ((1, 25, 301, 1234), (1, 6, 77, 5000), (0, 25, 301, 1234), (1, 50, 500, 99)):
        sc = scenario("small", seed=5, start_tti=10 * 77 + 3, cell_id=cid, nof_prb=nprb, cp=cp)
        x, first_tti = sync_capture(sc, lead, 2)
        r, s, _ = oracle_cell_search(x, nprb, 2, -1, 20.0)
        N = {6: 128, 25: 512, 50: 1024}[nprb]
        assert r == 1 and s.cell_id == cid and s.cp == cp, (cp, nprb, s.cell_id, s.cp)
        # the capture starts `lead` samples in front of subframe 3: the first subframe 0 / 5 boundary is subframe 5, two subframes on
        assert s.sf_idx == 5 and s.sf_start == lead + 2 * 15 * N, (s.sf_idx, s.sf_start)
