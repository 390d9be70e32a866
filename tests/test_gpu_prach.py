"""GPU parity of the PRACH detector (k_prach_bins / k_prach_corr / k_prach_peaks behind lsn_phy_prach_detect) against the
oracle on the same samples: correlation power bit-exact, identical detection lists."""
import ctypes as C

import numpy as np
import pytest

import ltesniffer_amd as la
from lsn_testlib import OCell, OPrachCfg, OPrachDet, oracle_prach_api, prach_subframe

pytestmark = pytest.mark.gpu


def oracle_run(o, nprb, iq, config_idx, root, zcz, fo, zc=None, factor=0.0):
    ocell = OCell(nprb, 1, 1, 1)
    pc = OPrachCfg(config_idx, root, zcz, fo, 0, factor, zc.ctypes.data_as(C.POINTER(C.c_uint16)) if zc is not None else None)
    det = (OPrachDet * 64)()
    nroots = o.o_prach_nof_roots(zcz)
    corr = np.zeros(nroots * 839, dtype=np.float32)
    n = o.o_prach_detect(C.byref(ocell), C.byref(pc), iq.ctypes.data, det, 64, corr.ctypes.data)
    return [(det[i].preamble, det[i].offset, det[i].offset_sec, det[i].p2avg) for i in range(n)], corr.reshape(nroots, 839)


@pytest.mark.parametrize("nprb,zcz,root,fo", [(25, 5, 10, 2), (100, 5, 10, 4), (50, 1, 700, 20), (25, 0, 830, 0), (100, 12, 837, 94)])
def test_prach_detect_matches_oracle(nprb, zcz, root, fo):
    o = oracle_prach_api()
    rng = np.random.default_rng(nprb + zcz)
    table = (rng.permutation(838) + 1).astype(np.uint16) if zcz == 0 else None
    nsym = o.o_fft_size(nprb)
    config_idx = 14  # every subframe is an occasion
    phy = la.Phy(nof_rx_antennas=1, max_batch=4)
    assert phy.setCell(nprb, 1, 1)
    assert phy.setPrachConfig(config_idx, root, zcz, fo, zc_roots=table)
    nwin = 839 // la.PRACH_NCS[zcz] if zcz else 1
    sfs, want = [], []
    for s in range(4):
        ues = [(int(rng.integers(0, 64)), int(rng.integers(0, 150) * nsym / 2048), float(rng.uniform(-3, 3))) for _ in range(s)]  # 0..3 UEs
        ues = list({u[0]: u for u in ues}.values())
        iq = prach_subframe(nprb, ues, snr_db=6.0, seed=100 + s, zero_corr_zone=zcz, root_seq_idx=root, freq_offset=fo, zc_roots=table)
        sfs.append(iq)
        det, corr = oracle_run(o, nprb, iq, config_idx, root, zcz, fo, table)
        assert sorted(d[0] for d in det) == sorted(u[0] for u in ues), (s, ues, det)
        want.append((det, corr))
    got = phy.prach_detect(np.stack(sfs), start_tti=7)
    for s, (det, corr) in enumerate(want):
        gc = phy.tap_prach_corr(s)
        assert np.array_equal(gc.view(np.uint32), corr.view(np.uint32)), (s, float(np.abs(gc - corr).max()))
        mine = [(g["preamble"], g["offset"], g["offset_sec"], g["p2avg"]) for g in got if g["sf"] == s]
        assert len(mine) == len(det)
        for a, b in zip(mine, det):
            assert a[0] == b[0] and a[1] == b[1] and np.float32(a[2]) == np.float32(b[2]) and np.float32(a[3]) == np.float32(b[3]), (s, a, b)
    phy.close()


def test_prach_occasions_and_invalid_configs():
    o = oracle_prach_api()
    phy = la.Phy(nof_rx_antennas=1, max_batch=4)
    assert phy.setCell(25, 1, 1)
    for cfg in range(17):
        for tti in range(0, 40):
            assert la.lib().lsn_prach_tti_opportunity(cfg, tti) == o.o_prach_tti_opportunity(cfg, tti)
    assert not phy.setPrachConfig(16, 0, 1, 0)            # preamble format 1
    assert not phy.setPrachConfig(3, 0, 1, 0, hs_flag=1)  # restricted set
    assert not phy.setPrachConfig(3, 0, 1, 20)            # 6 PRBs do not fit
    assert phy.setPrachConfig(3, 22, 4, 3)                # occasions: subframe 1 of every frame
    iq = np.stack([prach_subframe(25, [(9, 12, 0.0)], seed=s, zero_corr_zone=4, root_seq_idx=22, freq_offset=3) for s in range(12)])
    got = phy.prach_detect(iq, start_tti=100)             # subframes 1 and 11 of the block are tti 101 and 111
    assert [(g["sf"], g["preamble"]) for g in got] == [(1, 9), (11, 9)], got
    phy.close()


def test_ul_mode_reports_prach_occasions_through_the_sink():
    """UL_MODE batches: the detector runs on antenna 1 of every PRACH occasion (work_prach) without disturbing the record stream"""
    from lsn_testlib import gen_ul_mode_subframes, scenario
    sc = scenario("small", seed=21, nof_rx=1, ul_min=1, ul_max=2, mcs_max=16)
    tti0, iq, _ = gen_ul_mode_subframes(sc, 40)
    ttis = [tti0 + i for i in range(40)]
    occ = [i for i, t in enumerate(ttis) if la.lib().lsn_prach_tti_opportunity(6, t)]  # subframes 1 and 6 of every frame
    assert len(occ) >= 6
    hit = {occ[1]: (11, 5), occ[4]: (40, 20)}
    for i, (pre, dly) in hit.items():
        iq[i, 1] += 0.05 * prach_subframe(25, [(pre, dly, 0.0)], snr_db=60.0, seed=i, zero_corr_zone=7, root_seq_idx=50, freq_offset=10)
    seen = []
    recs = []
    for with_prach in (False, True):
        phy = la.Phy(nof_rx_antennas=2, sniffer_mode=1, max_batch=16, pcapwriter=la.PcapWriter(None))
        assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"]) and phy.setUlConfig(3, 5)
        if with_prach:
            assert phy.setPrachConfig(6, 50, 7, 10)
            phy.set_prach_sink(lambda tti, det: seen.append((tti, [(d["preamble"], d["offset"]) for d in det])))
        phy.process_host(iq, tti0, 25)
        recs.append(phy.pcapwriter.bytes())
        phy.close()
    assert [s[0] for s in seen] == [ttis[i] % 10240 for i in sorted(hit)], seen
    for (tti, det), i in zip(seen, sorted(hit)):
        assert [d[0] for d in det] == [hit[i][0]] and abs(det[0][1] - hit[i][1] * 839.0 / (12 * 384)) <= 1.0, (tti, det)
    assert recs[0] == recs[1]
