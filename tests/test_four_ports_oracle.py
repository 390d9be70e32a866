"""CPU tests of the four-CRS-port restatement (36.211 6.10.1.2 ports 2 / 3, 6.2.4 REGs of symbol 1, 6.3.4.3 SFBC-FSTD on PCFICH / PDCCH / PBCH /
PDSCH, 36.212 5.3.3.1 four-port DCI sizes): the synthetic transmitter (tools/txgen) -> oracle worker loop-back.  The reference hands any
srsran_cell_t to srsRAN (/root/reference/src/src/SubframeWorker.cc:102), whose four-port support is transmit diversity; grants that ask for
spatial multiplexing are found on the PDCCH and not decoded (dl_sniffer_pdsch.c:134-276 has no four-port branch of its own)."""
import ctypes as C

import numpy as np
import pytest

from lsn_testlib import OCell, oracle, scenario
from parity import gen_subframes, run_oracle


def _sent(truth):
    return [p for t in truth for p in t if not p["is_ul"]]


@pytest.mark.parametrize("scn,over", [("small", {}), ("small", dict(nof_prb=100, n_rnti=8, dl_min=3, dl_max=5, cfi=0, rar_period=20, paging_period=16)),
                                      ("cfg2", dict(nof_prb=50, n_rnti=6, cell_id=329, phich_ng_x6=6)), ("small", dict(nof_prb=6, cfi=0, n_rnti=2, dl_min=1, dl_max=1)),
                                      ("small", dict(nof_rx=1, cell_id=5))])
def test_four_port_loopback_decodes_what_was_sent(scn, over):
    sc = scenario(scn, seed=5, nof_ports=4, **over)
    tti0, iq, truth = gen_subframes(sc, 60)
    ow, per_sf, recs = run_oracle(sc, tti0, iq)
    sent = _sent(truth)
    got = set(r["pdu"] for r in recs)
    hit = sum(1 for p in sent if p["payload"] in got)
    assert len(sent) >= 40 and hit >= 0.9 * len(sent), (len(sent), hit, len(recs))  # the first subframes go by before the RNTI histogram accepts the UEs
    # every port is estimated: per-port noise far below the pilot power at 30 dB
    ch = per_sf[-1]["chest"]
    A = sc["nof_rx"]
    noise, cep = ch[0:8].reshape(2, 4)[:A], ch[16:24].reshape(2, 4)[:A]
    assert (cep > 0.2).all() and (noise < cep * 0.02).all(), (noise, cep)


def test_four_port_dci_sizes_and_reg_count():
    o = oracle()
    # 36.212 5.3.3.1.5 / 5.3.3.1.5A: precoding information 3 / 0 bits with two ports, 6 / 2 bits with four; written out by hand: type-0 header 1 +
    # RBG bitmap 25 / 17 / 13 + TPC 2 + HARQ 3 + swap 1 + 2 x (MCS 5 + NDI 1 + RV 2) + precoding; 40 bits is an ambiguous size (-> 41)
    for nprb, f2_2, f2_4, f2a_2, f2a_4 in ((100, 51, 54, 48, 50), (50, 43, 46, 41, 42), (25, 39, 42, 36, 38)):
        c2, c4 = OCell(nprb, 2, 1, 1), OCell(nprb, 4, 1, 1)
        assert (o.o_dci_format_sizeof(C.byref(c2), 6), o.o_dci_format_sizeof(C.byref(c4), 6)) == (f2_2, f2_4)
        assert (o.o_dci_format_sizeof(C.byref(c2), 7), o.o_dci_format_sizeof(C.byref(c4), 7)) == (f2a_2, f2a_4)


def test_spatial_multiplexing_grants_on_four_ports_are_found_and_not_decoded():
    sc = scenario("cfg3", seed=9, nof_ports=4, nof_prb=50, n_rnti=12, dl_min=4, dl_max=6, mix_tm3_pct=50, mix_tm4_pct=50, rar_period=0, paging_period=0)
    tti0, iq, truth = gen_subframes(sc, 80)
    ow, per_sf, recs = run_oracle(sc, tti0, iq)
    sent = _sent(truth)
    two_tb = {p["rnti"] for p in sent if p["format"] >= 6}
    assert len(two_tb) >= 8
    got_rnti = {r["rnti"] for r in recs}
    assert not (got_rnti & two_tb)                      # no record for a format 2 / 2A grant
    acc = [a for sf in per_sf for a in sf["accepted"]]    # (rnti, format, L, ncce, nof_bits, histogram value)
    sizes = {a[1]: a[4] for a in acc if a[0] in two_tb and a[1] >= 6}  # (the same UEs also get uplink grants, format 0)
    assert sizes == {6: 46, 7: 42}, sizes  # the search accepts them at their four-port sizes


class _ORegs(C.Structure):  # o_regs_t (lsn_oracle.h): only the leading counters are read here
    _fields_ = [("nof_regs", C.c_uint32 * 3), ("nof_cce", C.c_uint32 * 3), ("rest", C.c_uint8 * 65536)]


def test_control_region_sizes_against_hand_arithmetic():
    """36.211 6.2.4: symbol 0 holds 2 REGs per PRB whatever the port count; symbol 1 holds 3 per PRB with one or two ports and 2 with four (the CRS of
    ports 2 / 3); symbol 2 holds 3.  Minus 4 PCFICH REGs and 3 x ceil(Ng x N_RB / 8) PHICH REGs.  Written out by hand for Ng = 1/6:
    100 PRB -> PHICH 3 x 3 = 9; 50 PRB -> 3 x 2 = 6; 25 PRB -> 3 x 1 = 3."""
    o = oracle()
    o.o_regs_init.argtypes = [C.POINTER(OCell), C.POINTER(_ORegs)]
    for nprb, phich in ((100, 9), (50, 6), (25, 3)):
        for ports, per_prb in ((2, (2, 3, 3)), (4, (2, 2, 3))):
            r = _ORegs()
            o.o_regs_init(C.byref(OCell(nprb, ports, 1, 1)), C.byref(r))
            for cfi in (1, 2, 3):
                regs = nprb * sum(per_prb[:cfi]) - 4 - phich
                assert (r.nof_regs[cfi - 1], r.nof_cce[cfi - 1]) == (regs, regs // 9), (nprb, ports, cfi)
    # the numbers every LTE engineer knows: 20 MHz, two ports, CFI 3 -> 87 CCEs; four ports -> 76
    r2, r4 = _ORegs(), _ORegs()
    o.o_regs_init(C.byref(OCell(100, 2, 1, 1)), C.byref(r2))
    o.o_regs_init(C.byref(OCell(100, 4, 1, 1)), C.byref(r4))
    assert r2.nof_cce[2] == 87 and r4.nof_cce[2] == 76


def _gold(cinit, n):
    """36.211 7.2, restated here in numpy terms (not the oracle's o_gold)"""
    x1 = np.zeros(1600 + n + 31, dtype=np.uint8)
    x2 = np.zeros(1600 + n + 31, dtype=np.uint8)
    x1[0] = 1
    for i in range(31):
        x2[i] = (cinit >> i) & 1
    for i in range(1600 + n):
        x1[i + 31] = x1[i + 3] ^ x1[i]
        x2[i + 31] = x2[i + 3] ^ x2[i + 2] ^ x2[i + 1] ^ x2[i]
    return x1[1600:1600 + n] ^ x2[1600:1600 + n]


@pytest.mark.parametrize("cell_id,nof_prb", [(1, 25), (302, 6), (77, 50)])
def test_transmitted_crs_of_ports_2_and_3_against_an_independent_restatement(cell_id, nof_prb):
    """36.211 6.10.1: r_{l,ns}(m) = (1 - 2 c(2m)) / sqrt2 + j (1 - 2 c(2m + 1)) / sqrt2, c_init = 2^10 (7 (ns + 1) + l + 1)(2 N_ID + 1) + 2 N_ID + 1;
    ports 2 / 3 sit on symbol 1 of every slot at k = 6 m + (v + v_shift) mod 6 with v = 3 (ns mod 2) for port 2 and 3 + 3 (ns mod 2) for port 3.
    The transmitter's waveform (one rx antenna, no noise to speak of) is demodulated with a plain numpy FFT and the pilots of ports 2 / 3 are
    divided by this test's own sequence: every port must show ONE constant (its channel gain) over all its pilots, and no PDSCH / control symbol
    may sit on any port's pilot positions."""
    from lsn_testlib import TxGen
    sc = scenario("small", seed=3, nof_ports=4, nof_prb=nof_prb, cell_id=cell_id, nof_rx=1, snr_db=80.0, cfi=0 if nof_prb > 6 else 3,
                  n_rnti=3, dl_min=1, dl_max=2)
    tx = TxGen(**sc)
    N = {6: 128, 25: 512, 50: 1024}[nof_prb]
    nre = 12 * nof_prb
    for _ in range(6):
        tti, iq, _ = tx.next()
        sf = tti % 10
        x = iq[0]
        grid = np.zeros((14, nre), dtype=np.complex128)
        pos = 0
        for l in range(14):
            cp = (160 if l % 7 == 0 else 144) * N // 2048
            f = np.fft.fft(x[pos + cp:pos + cp + N].astype(np.complex128))
            grid[l, :nre // 2] = f[N - nre // 2:]
            grid[l, nre // 2:] = f[1:nre // 2 + 1]
            pos += cp + N
        for port in (2, 3):
            gains = []
            for slot, l in ((0, 1), (1, 8)):
                ns = 2 * sf + slot
                c = _gold(1024 * (7 * (ns + 1) + 1 + 1) * (2 * cell_id + 1) + 2 * cell_id + 1, 440)
                v = (3 * slot if port == 2 else 3 + 3 * slot)
                koff = (v + cell_id % 6) % 6
                m = np.arange(2 * nof_prb)
                mp = m + 110 - nof_prb
                r = ((1 - 2.0 * c[2 * mp]) + 1j * (1 - 2.0 * c[2 * mp + 1])) / np.sqrt(2.0)
                gains.append(grid[l, 6 * m + koff] / r)
            g = np.concatenate(gains)
            assert np.abs(g).mean() > 0.3 and np.abs(g - g.mean()).max() < 2e-3 * np.abs(g.mean()) + 2e-3, (port, sf, float(np.abs(g - g.mean()).max()))


def test_transmitted_pcfich_follows_sfbc_fstd_restated_in_numpy():
    """36.211 6.3.4.3 (four ports): of every four symbols x0..x3, REs 0 / 1 carry (x0, x1) on port 0 and (-x1*, x0*) on port 2, REs 2 / 3 carry
    (x2, x3) on port 1 and (-x3*, x2*) on port 3, all / sqrt2.  Checked on the PCFICH (6.7: 32 scrambled bits of the CFI code word, QPSK, four
    quadruplets at k = kbar + floor(i N_RB / 2) 6, the CRS positions of a REG left out) of the transmitter's waveform: per-port gains from this test's
    own CRS restatement, expected REs from this test's own precoding - equal to the received grid."""
    from lsn_testlib import TxGen
    cell_id, nof_prb, cfi = 77, 25, 2
    sc = scenario("small", seed=4, nof_ports=4, nof_prb=nof_prb, cell_id=cell_id, nof_rx=1, snr_db=80.0, cfi=cfi, n_rnti=2, dl_min=1, dl_max=1)
    tx = TxGen(**sc)
    N, nre = 512, 12 * nof_prb
    cw = {1: "01101101101101101101101101101101", 2: "10110110110110110110110110110110", 3: "11011011011011011011011011011011"}[cfi]
    for _ in range(4):
        tti, iq, _ = tx.next()
        sf = tti % 10
        grid = np.zeros((14, nre), dtype=np.complex128)
        pos = 0
        for l in range(14):
            cp = (160 if l % 7 == 0 else 144) * N // 2048
            f = np.fft.fft(iq[0][pos + cp:pos + cp + N].astype(np.complex128))
            grid[l, :nre // 2] = f[N - nre // 2:]
            grid[l, nre // 2:] = f[1:nre // 2 + 1]
            pos += cp + N
        # per-port channel gain (flat channel) from the pilots: ports 0 / 1 on symbol 0 (v = 0 / 3), ports 2 / 3 on symbol 1 (v = 0 / 3 in slot 0)
        h = []
        for port in range(4):
            l = 0 if port < 2 else 1
            c = _gold(1024 * (7 * (2 * sf + 1) + l + 1) * (2 * cell_id + 1) + 2 * cell_id + 1, 440)
            koff = ((3 if port in (1, 3) else 0) + cell_id % 6) % 6
            m = np.arange(2 * nof_prb)
            mp = m + 110 - nof_prb
            r = ((1 - 2.0 * c[2 * mp]) + 1j * (1 - 2.0 * c[2 * mp + 1])) / np.sqrt(2.0)
            h.append((grid[l, 6 * m + koff] / r).mean())
        scr = _gold((sf + 1) * (2 * cell_id + 1) * 512 + cell_id, 32)
        b = np.array([int(ch) for ch in cw], dtype=np.uint8) ^ scr
        d = ((1 - 2.0 * b[0::2]) + 1j * (1 - 2.0 * b[1::2])) / np.sqrt(2.0)
        kbar = 6 * (cell_id % (2 * nof_prb))
        for i in range(4):
            k0 = (kbar + (i * nof_prb // 2) * 6) % nre
            ks = [k for k in range(k0, k0 + 6) if k % 3 != cell_id % 3]
            x0, x1, x2, x3 = d[4 * i:4 * i + 4]
            want = np.array([h[0] * x0 - h[2] * np.conj(x1), h[0] * x1 + h[2] * np.conj(x0),
                             h[1] * x2 - h[3] * np.conj(x3), h[1] * x3 + h[3] * np.conj(x2)]) / np.sqrt(2.0)
            got = grid[0, ks]
            assert np.abs(got - want).max() < 5e-3, (sf, i, got, want)
