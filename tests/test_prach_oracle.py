"""CPU tests of the PRACH restatement: independent numpy transmitter (36.211 5.7.3) -> oracle detector loop-back."""
import ctypes as C

import numpy as np
import pytest

from lsn_testlib import OCell, OPrachCfg, OPrachDet, PRACH_NCS, oracle_prach_api, prach_subframe


def run_detect(o, nprb, iq, **cfg):
    ocell = OCell(nprb, 1, 1, 1)
    zc = cfg.pop("zc_roots", None)
    pc = OPrachCfg(cfg.get("config_idx", 3), cfg.get("root_seq_idx", 10), cfg.get("zero_corr_zone", 5), cfg.get("freq_offset", 4), 0,
                   cfg.get("detect_factor", 0.0), zc.ctypes.data_as(C.POINTER(C.c_uint16)) if zc is not None else None)
    det = (OPrachDet * 64)()
    n = o.o_prach_detect(C.byref(ocell), C.byref(pc), iq.ctypes.data, det, 64, None)
    return [(det[i].preamble, det[i].offset, det[i].offset_sec, det[i].p2avg) for i in range(n)]


def test_tti_opportunities_follow_table_5_7_1_2():
    o = oracle_prach_api()
    want = {0: (True, [1]), 1: (True, [4]), 2: (True, [7]), 3: (False, [1]), 4: (False, [4]), 5: (False, [7]), 6: (False, [1, 6]),
            7: (False, [2, 7]), 8: (False, [3, 8]), 9: (False, [1, 4, 7]), 10: (False, [2, 5, 8]), 11: (False, [3, 6, 9]),
            12: (False, [0, 2, 4, 6, 8]), 13: (False, [1, 3, 5, 7, 9]), 14: (False, list(range(10))), 15: (True, [9])}
    for cfg, (even, sfs) in want.items():
        for tti in range(40):
            exp = (tti % 10 in sfs) and (not even or (tti // 10) % 2 == 0)
            assert bool(o.o_prach_tti_opportunity(cfg, tti)) == exp, (cfg, tti)
    assert not o.o_prach_tti_opportunity(16, 1)  # formats 1-3 do not fit the one subframe work_prach hands over


@pytest.mark.parametrize("nprb", [25, 100])
def test_loopback_three_ues_two_roots(nprb):
    o = oracle_prach_api()
    nsym = o.o_fft_size(nprb)
    scale = nsym / 2048.0
    ues = [(5, 0, 0.0), (40, int(60 * scale), -3.0), (63, int(130 * scale), 2.0)]  # N_CS = 26 -> 32 shifts per root, 2 roots
    iq = prach_subframe(nprb, ues, snr_db=5.0, seed=3, freq_offset=2)
    got = run_detect(o, nprb, iq, freq_offset=2)
    assert [g[0] for g in got] == [5, 40, 63], got
    for (idx, delay, _), g in zip(ues, got):
        lag = delay * 839.0 / (12 * nsym)
        assert abs(g[1] - lag) <= 1.0, (idx, delay, g)
        assert abs(g[2] - g[1] * 0.8e-3 / 839) < 1e-9 and g[3] > 60.0


def test_noise_only_and_wrong_offset_detect_nothing():
    o = oracle_prach_api()
    assert run_detect(o, 25, prach_subframe(25, [], seed=9)) == []
    iq = prach_subframe(25, [(7, 10, 0.0)], snr_db=10.0, seed=4, freq_offset=4)
    assert [g[0] for g in run_detect(o, 25, iq, freq_offset=4)] == [7]
    assert run_detect(o, 25, iq, freq_offset=12) == []          # looking at other PRBs
    assert run_detect(o, 25, iq, freq_offset=4, root_seq_idx=300) == []  # other root sequences


def test_root_table_and_zero_ncs():
    o = oracle_prach_api()
    rng = np.random.default_rng(1)
    table = (rng.permutation(838) + 1).astype(np.uint16)  # a stand-in for Table 5.7.2-4 (the real one is supplied by the caller)
    # N_CS = 0: one preamble per root, 64 roots
    assert o.o_prach_nof_roots(0) == 64 and o.o_prach_nof_roots(5) == 2 and o.o_prach_nof_roots(1) == 1
    iq = prach_subframe(25, [(3, 20, 0.0), (17, 0, 0.0)], snr_db=8.0, seed=2, zero_corr_zone=0, root_seq_idx=830, zc_roots=table)
    got = run_detect(o, 25, iq, zero_corr_zone=0, root_seq_idx=830, zc_roots=table)
    assert [g[0] for g in got] == [3, 17], got
    assert PRACH_NCS[15] == 419
