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
