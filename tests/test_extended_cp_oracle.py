"""Extended cyclic prefix (srsran_cell_t.cp = 1) in the CPU oracle, checked without a GPU: the transmitter's 12-symbol subframes (CP of N / 4,
CRS on symbols 0 / 3 of each slot with N_CP = 0 in c_init, PSS / SSS on symbols 5 / 4, PBCH on symbols 6 - 9 with 216 symbols) decode through
the oracle's OFDM -> CRS estimate -> PCFICH / PDCCH -> FALCON search -> PDSCH chain; and a normal-CP receiver does NOT decode them (the two
grids really differ).  Independent structure checks: sample count 12 (N + N / 4) = 15 N, the PDSCH RE count of a full-band grant."""
import ctypes as C

import numpy as np

from lsn_testlib import OCell, oracle, scenario
from parity import gen_subframes, run_oracle


def _loopback(nsf=16, **over):
    sc = scenario("small", seed=41, cp=1, **over)
    tti0, iq, truth = gen_subframes(sc, nsf)
    _, per_sf, recs = run_oracle(sc, tti0, iq)
    sent = [p["payload"] for t in truth for p in t if not p["is_ul"]]
    got = {bytes(r["pdu"]) for r in recs}
    return sc, tti0, iq, per_sf, sent, got


def test_transmitter_to_oracle_loopback_extended_cp():
    for over in (dict(), dict(nof_prb=100, mix_tm3_pct=40, mix_tm4_pct=30, pct_256qam=50, n_rnti=12), dict(nof_prb=50, nof_ports=1, nof_rx=1),
                 dict(nof_ports=4), dict(nof_prb=6, cfi=3, dl_min=1, dl_max=1, n_rnti=2)):
        sc, tti0, iq, per_sf, sent, got = _loopback(**over)
        ok = sum(1 for p in sent if p in got)
        assert ok >= 0.6 * len(sent) and ok > 8, (over, ok, len(sent))   # (the rest: second-table attempts the tracker has not learnt yet, as with the normal CP)
        assert all(p["cfi"] == sc["cfi"] for p in per_sf)
        # rows 12, 13 of the 14-row grid do not exist in an extended-CP subframe
        assert not np.any(per_sf[0]["grid"][:, 12:, :]) and np.any(per_sf[0]["grid"][:, 11, :])


def test_normal_cp_receiver_does_not_decode_an_extended_cp_capture():
    sc = scenario("small", seed=42, cp=1)
    tti0, iq, truth = gen_subframes(sc, 10)
    _, _, recs = run_oracle(dict(sc, cp=0), tti0, iq, taps=False)
    assert len(recs) == 0


def test_structure():
    o = oracle()
    o.o_fft_size.argtypes = [C.c_uint32]
    for nprb in (6, 15, 25, 50, 75, 100):
        N = o.o_fft_size(nprb)
        assert 12 * (N + 512 * N // 2048) == 15 * N           # twelve symbols with a CP of N / 4 fill the subframe exactly
    # PDSCH-capable REs of one PRB pair in an ordinary subframe, CFI 2, two ports: 12 symbols - 2 control = 10, CRS in symbols 3, 6, 9 (4 REs each)
    cell = OCell(25, 2, 1, 1, 0, 1)
    o.o_pdsch_re_ok.argtypes = [C.POINTER(OCell), C.c_uint32, C.c_uint32, C.c_uint32]
    n = sum(o.o_pdsch_re_ok(C.byref(cell), 1, l, k) for l in range(2, 12) for k in range(12))
    assert n == 10 * 12 - 3 * 4
    # subframe 0: PSS / SSS / PBCH take the centre 72 carriers of symbols 4 - 9
    mid = 6 * 25 - 36
    assert all(o.o_pdsch_re_ok(C.byref(cell), 0, l, mid + 5) == 0 for l in range(4, 10))
    assert o.o_pdsch_re_ok(C.byref(cell), 0, 10, mid + 5) == 1


def test_pusch_loopback_extended_cp_with_control_information():
    """uplink with the extended CP: 12 SC-FDMA symbols (CP N / 4), reference signal on symbol 2 of each slot with n_PN over 8 * 6 bits per slot, a
    10-column channel interleaver with the rank indication on columns 0 3 5 8 and the HARQ-ACK on 1 2 6 7 (36.212 Tables 5.2.2.8-1 / -2), Q' from
    N_symb^PUSCH = 10 - transmitter -> oracle, every modulation, with and without multiplexed control information; a normal-CP receiver fails"""
    from lsn_testlib import OPuschGrant, OUci, OUlCfg, TxgUlCell, VALID_UL_PRB, oracle_ul_api, ul_make_subframe, ul_mcs_to_mod_tbs
    o = oracle_ul_api()
    ok = bad_n = 0
    for nprb, cell_id in ((25, 7), (100, 1)):
        rng = np.random.default_rng(nprb + 1)
        ocell, ncell = OCell(nprb, 1, cell_id, 1, 0, 1), OCell(nprb, 1, cell_id, 1, 0, 0)
        ucell, ucfg = TxgUlCell(nprb, cell_id, 3, 5, 0, 0, 1), OUlCfg(3, 5)
        cqi_bits = o.o_uci_cqi_bits(nprb)
        for it in range(4):
            tti = int(rng.integers(0, 10240))
            grants, start = [], 0
            while True:
                L = int(rng.choice([n for n in VALID_UL_PRB if 3 <= n <= max(3, nprb // 3)]))
                if start + L > nprb:
                    break
                qm, tbs = ul_mcs_to_mod_tbs(int(rng.integers(0, 27)), L)
                g = dict(rnti=int(rng.integers(100, 60000)), n_dmrs=int(rng.integers(0, 8)), n_prb=start, L_prb=L, mod=qm, tbs=tbs, rv=0,
                         gain_db=float(rng.uniform(-3, 3)), phase_rad=float(rng.uniform(0, 6.28)), ta_samples=float(rng.uniform(0, 3)))
                if it % 2:
                    cq = int(rng.integers(0, 2)) * cqi_bits
                    g.update(nof_ack=int(rng.integers(0, 3)), cqi_bits=cq, ri_bits=1 if cq else 0)
                grants.append(g)
                start += L + int(rng.integers(0, 3))
            iq, payloads = ul_make_subframe(ucell, tti, grants, snr_db=35.0, seed=it)
            grid = np.zeros(14 * 12 * nprb, dtype=np.complex64)
            o.o_ul_fft(C.byref(ocell), iq.ctypes.data, grid.ctypes.data)
            assert not np.any(grid[12 * 12 * nprb:])
            ngrid = np.zeros(14 * 12 * nprb, dtype=np.complex64)
            o.o_ul_fft(C.byref(ncell), iq.ctypes.data, ngrid.ctypes.data)
            for g, pl in zip(grants, payloads):
                og = OPuschGrant(g["L_prb"], g["n_prb"], 0, g["mod"], g["tbs"], 0)
                uci = OUci(g.get("nof_ack", 0), g.get("cqi_bits", 0), g.get("ri_bits", 0))
                out = np.zeros(g["tbs"] // 8 + 8, dtype=np.uint8)
                its, snr = C.c_int(0), C.c_float(0)
                crc = o.o_pusch_decode_uci(C.byref(ocell), C.byref(ucfg), tti % 10, g["rnti"], C.byref(og), g["n_dmrs"], C.byref(uci), grid.ctypes.data, 12,
                                           out.ctypes.data, C.byref(its), C.byref(snr))
                assert crc == 1 and bytes(out[:g["tbs"] // 8]) == pl, (nprb, it, g)
                ok += 1
                bad_n += o.o_pusch_decode_uci(C.byref(ncell), C.byref(ucfg), tti % 10, g["rnti"], C.byref(og), g["n_dmrs"], C.byref(uci), ngrid.ctypes.data, 4,
                                              out.ctypes.data, C.byref(its), C.byref(snr))
    assert ok >= 12 and bad_n == 0
