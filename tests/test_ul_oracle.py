"""CPU tests of the uplink restatement: SC-FDMA transmitter (tools/txgen) -> oracle PUSCH receiver loop-back."""
import ctypes as C

import numpy as np
import pytest

from lsn_testlib import (OCell, OPuschGrant, OUlCfg, TxgUlCell, VALID_UL_PRB, oracle_ul_api, ul_make_subframe, ul_mcs_to_mod_tbs)


def test_valid_prb_set_matches_reference_table():
    # UL_Sniffer_PUSCH.cc:3-10 lists exactly the 2^a 3^b 5^c sizes up to 100
    o = oracle_ul_api()
    ref = [1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 16, 18, 20, 24, 25, 27, 30, 32, 36, 40, 45, 48, 50, 54, 60, 64, 72, 75, 80, 81, 90, 96, 100]
    assert [n for n in range(1, 101) if o.o_ul_valid_prb(n)] == ref == VALID_UL_PRB


@pytest.mark.parametrize("nprb,cell_id", [(25, 7), (50, 101), (100, 1)])
def test_pusch_loopback_all_modulations(nprb, cell_id):
    o = oracle_ul_api()
    rng = np.random.default_rng(nprb)
    ocell, ucell, ucfg = OCell(nprb, 1, cell_id, 1), TxgUlCell(nprb, cell_id, 3, 5), OUlCfg(3, 5)
    ok = 0
    for it in range(6):
        tti = int(rng.integers(0, 10240))
        # pack a few non-overlapping grants into the band
        grants, start = [], 0
        while True:
            # one / two PRB (tabulated DMRS sequences, 36.211 Tables 5.5.1.2-1 / -2) and >= 3 PRB (Zadoff-Chu)
            L = int(rng.choice([n for n in VALID_UL_PRB if n <= max(3, nprb // 3)]))
            if start + L > nprb:
                break
            mcs = int(rng.integers(0, 29))
            qm, tbs = ul_mcs_to_mod_tbs(mcs, L)
            grants.append(dict(rnti=int(rng.integers(100, 60000)), n_dmrs=int(rng.integers(0, 8)), n_prb=start, L_prb=L, mod=qm, tbs=tbs, rv=0,
                               gain_db=float(rng.uniform(-3, 3)), phase_rad=float(rng.uniform(0, 6.28)), ta_samples=float(rng.uniform(0, 4))))
            start += L + int(rng.integers(0, 3))
        iq, payloads = ul_make_subframe(ucell, tti, grants, snr_db=35.0, seed=it)
        grid = np.zeros(14 * 12 * nprb, dtype=np.complex64)
        o.o_ul_fft(C.byref(ocell), iq.ctypes.data, grid.ctypes.data)
        for g, pl in zip(grants, payloads):
            og = OPuschGrant(g["L_prb"], g["n_prb"], 0, g["mod"], g["tbs"], 0)
            out = np.zeros(g["tbs"] // 8 + 8, dtype=np.uint8)
            its, snr = C.c_int(0), C.c_float(0)
            crc = o.o_pusch_decode(C.byref(ocell), C.byref(ucfg), tti % 10, g["rnti"], C.byref(og), g["n_dmrs"], grid.ctypes.data, 12, out.ctypes.data,
                                   C.byref(its), C.byref(snr))
            assert crc == 1, (nprb, it, g)
            assert bytes(out[:g["tbs"] // 8]) == pl
            assert snr.value > 15.0
            ok += 1
            # a wrong cyclic shift / RNTI must not decode
            assert o.o_pusch_decode(C.byref(ocell), C.byref(ucfg), tti % 10, g["rnti"] ^ 1, C.byref(og), g["n_dmrs"], grid.ctypes.data, 4, out.ctypes.data,
                                    C.byref(its), C.byref(snr)) == 0
    assert ok >= 10


def test_pusch_rejects_unsupported_grants():
    o = oracle_ul_api()
    ocell, ucfg = OCell(25, 1, 1, 1), OUlCfg(0, 0)
    grid = np.zeros(14 * 300, dtype=np.complex64)
    e = np.zeros(1 << 16, dtype=np.int16)
    for L, n in ((7, 0), (10, 20), (0, 0), (11, 0)):
        og = OPuschGrant(L, n, 0, 2, 104, 0)
        assert o.o_pusch_demod(C.byref(ocell), C.byref(ucfg), 0, 70, C.byref(og), 0, grid.ctypes.data, e.ctypes.data, None, None) == -1


def test_ul_mode_worker_loopback():
    """UL_MODE: DCI 0 found on the downlink antenna at t -> PUSCH decoded from the uplink antenna at t + 4, MCS > 20 learns the
    UE's maximum modulation; every emitted uplink record equals a transmitted payload"""
    from lsn_testlib import OracleWorkerUl, gen_ul_mode_subframes, parse_pcap, scenario
    sc = scenario("cfg2", seed=5, nof_rx=1, n_rnti=12, dl_min=2, dl_max=3, ul_min=2, ul_max=4, mcs_max=20, pct_cqi_req=40)
    tti0, iq, sent = gen_ul_mode_subframes(sc, 60)
    ow = OracleWorkerUl(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], 3, 5)
    for i in range(iq.shape[0]):
        ow.work_ul(iq[i, 0], iq[i, 1], tti0 + i)
    recs = parse_pcap(ow.pcap_bytes())
    ul = [r for r in recs if r["direction"] == 0]
    dl = [r for r in recs if r["direction"] == 1]
    assert len(ul) >= 10 and len(dl) >= 5
    sent_set = {(s["tti"], s["rnti"], s["payload"]) for s in sent}
    valid_sizes = {1, 2, 3, 4, 5, 6, 8, 9, 10}
    for r in ul:
        assert (r["sfn"] * 10 + r["sf"], r["rnti"], r["pdu"]) in sent_set
    # most decodable grants (valid PRB count >= 3) are recovered once the RNTIs are active
    late = [s for s in sent if s["tti"] >= tti0 + 30 and s["L_prb"] in valid_sizes]
    got = {(r["sfn"] * 10 + r["sf"], r["rnti"]) for r in ul}
    assert sum((s["tti"], s["rnti"]) in got for s in late) >= 0.6 * len(late)


@pytest.mark.parametrize("nprb,cell_id", [(25, 9), (100, 3)])
def test_pusch_loopback_with_uci_multiplexing(nprb, cell_id):
    """HARQ-ACK / RI / CQI multiplexed into the PUSCH (36.212 5.2.2.6-8): the transmitter fills the control cells with random bits,
    the receiver locates them, skips RI + CQI and erases the ACK cells; the payload must come back, and ignoring a CQI report
    (different rate matching) must fail"""
    from lsn_testlib import OUci
    o = oracle_ul_api()
    rng = np.random.default_rng(nprb + 1)
    ocell, ucell, ucfg = OCell(nprb, 1, cell_id, 1), TxgUlCell(nprb, cell_id, 3, 5), OUlCfg(3, 5)
    cqi_bits = o.o_uci_cqi_bits(nprb)
    assert cqi_bits == {25: 18, 100: 30}[nprb]
    tried = 0
    for it, (nof_ack, cqi, ri) in enumerate([(1, 0, 0), (2, 0, 0), (0, cqi_bits, 1), (2, cqi_bits, 1), (1, cqi_bits, 0)]):
        tti = int(rng.integers(0, 10240))
        grants, start = [], 0
        for L in (3, 6, 10):
            mcs = int(rng.integers(2, 26))
            qm, tbs = ul_mcs_to_mod_tbs(mcs, L)
            grants.append(dict(rnti=int(rng.integers(100, 60000)), n_dmrs=int(rng.integers(0, 8)), n_prb=start, L_prb=L, mod=qm, tbs=tbs, rv=0,
                               nof_ack=nof_ack, cqi_bits=cqi, ri_bits=ri))
            start += L + 1
        iq, payloads = ul_make_subframe(ucell, tti, grants, snr_db=35.0, seed=it)
        grid = np.zeros(14 * 12 * nprb, dtype=np.complex64)
        o.o_ul_fft(C.byref(ocell), iq.ctypes.data, grid.ctypes.data)
        for g, pl in zip(grants, payloads):
            og = OPuschGrant(g["L_prb"], g["n_prb"], 0, g["mod"], g["tbs"], 0)
            uci = OUci(nof_ack, cqi, ri)
            out = np.zeros(g["tbs"] // 8 + 8, dtype=np.uint8)
            its, snr = C.c_int(0), C.c_float(0)
            crc = o.o_pusch_decode_uci(C.byref(ocell), C.byref(ucfg), tti % 10, g["rnti"], C.byref(og), g["n_dmrs"], C.byref(uci), grid.ctypes.data, 12,
                                       out.ctypes.data, C.byref(its), C.byref(snr))
            assert crc == 1 and bytes(out[:g["tbs"] // 8]) == pl, (nprb, it, g)
            if cqi:  # the same samples decoded as if there were no control information: rate matching is off by Q_CQI + Q_RI
                assert o.o_pusch_decode(C.byref(ocell), C.byref(ucfg), tti % 10, g["rnti"], C.byref(og), g["n_dmrs"], grid.ctypes.data, 4, out.ctypes.data,
                                        C.byref(its), C.byref(snr)) == 0
            tried += 1
    assert tried == 15


def test_uci_layout_counts_and_positions():
    from lsn_testlib import OUci
    o = oracle_ul_api()
    M, tbs = 72, 1544  # 6 PRB
    cls = np.zeros(12 * M, dtype=np.uint8)
    didx = np.zeros(12 * M, dtype=np.int32)
    qa, qr, qc = C.c_int(), C.c_int(), C.c_int()
    n = o.o_uci_layout(M, tbs, C.byref(OUci(2, 30, 1)), cls.ctypes.data, didx.ctypes.data, C.byref(qa), C.byref(qr), C.byref(qc))
    sumk = 1568  # one code block: 1544 + 24 -> K = 1568
    assert qa.value == min(-(-2 * M * 12 * 160 // (8 * sumk)), 4 * M) and qr.value == -(-1 * M * 12 * 127 // (8 * sumk)) and qc.value == -(-38 * M * 12 * 18 // (8 * sumk))
    m = cls.reshape(M, 12)
    assert n == 12 * M - qr.value - qc.value
    assert (m == 2).sum() == qr.value and set(np.nonzero((m == 2).any(axis=0))[0]) <= {1, 4, 7, 10}
    assert (m == 3).sum() == qa.value and set(np.nonzero((m == 3).any(axis=0))[0]) <= {2, 3, 8, 9}
    assert (m == 1).sum() == qc.value and m[0, 0] == 1                      # CQI starts the row-major fill
    assert m[M - 1, 1] == 2 and m[M - 1, 10] == 2 and m[M - 1, 2] == 3      # first symbols sit in the bottom row
    d = didx.reshape(M, 12)
    data = np.sort(d[(m == 0) | (m == 3)])
    assert np.array_equal(data, np.arange(n))                              # every UL-SCH symbol index exactly once


def test_pusch_loopback_type1_frequency_hopping():
    """slot 0 and slot 1 of a grant on different PRBs (36.213 8.4.1): DMRS of slot 1 and the last six data symbols are taken from n_prb2"""
    from lsn_testlib import pusch_hop_slot1
    o = oracle_ul_api()
    nprb, cell_id = 50, 11
    ocell, ucell, ucfg = OCell(nprb, 1, cell_id, 1, 6), TxgUlCell(nprb, cell_id, 3, 5), OUlCfg(3, 5, 6)
    for hop_bits, n_prb, L in ((0, 3, 4), (2, 5, 6), (1, 30, 3)):
        n2 = pusch_hop_slot1(nprb, 6, hop_bits, n_prb)
        qm, tbs = ul_mcs_to_mod_tbs(12, L)
        g = dict(rnti=4242, n_dmrs=2, n_prb=n_prb, L_prb=L, mod=qm, tbs=tbs, rv=0, hop=1, n_prb2=n2)
        iq, payloads = ul_make_subframe(ucell, 77, [g], snr_db=30.0, seed=hop_bits)
        grid = np.zeros(14 * 12 * nprb, dtype=np.complex64)
        o.o_ul_fft(C.byref(ocell), iq.ctypes.data, grid.ctypes.data)
        out = np.zeros(tbs // 8 + 8, dtype=np.uint8)
        its, snr = C.c_int(0), C.c_float(0)
        og = OPuschGrant(L, n_prb, 0, qm, tbs, 0, n2, 1)
        assert o.o_pusch_decode(C.byref(ocell), C.byref(ucfg), 7, 4242, C.byref(og), 2, grid.ctypes.data, 12, out.ctypes.data, C.byref(its), C.byref(snr)) == 1
        assert bytes(out[:tbs // 8]) == payloads[0]
        og0 = OPuschGrant(L, n_prb, 0, qm, tbs, 0, n_prb, 0)  # the same samples read without hopping: slot 1 is elsewhere
        assert o.o_pusch_decode(C.byref(ocell), C.byref(ucfg), 7, 4242, C.byref(og0), 2, grid.ctypes.data, 4, out.ctypes.data, C.byref(its), C.byref(snr)) == 0


def test_ul_mode_worker_with_hopping_grants():
    from lsn_testlib import OracleWorkerUl, gen_ul_mode_subframes, parse_pcap, scenario
    sc = scenario("cfg2", seed=15, nof_rx=1, n_rnti=10, dl_min=2, dl_max=3, ul_min=2, ul_max=3, mcs_max=18, pct_hop=50, pusch_hop_offset=8)
    tti0, iq, sent = gen_ul_mode_subframes(sc, 60)
    assert sum(1 for s in sent if s.get("hop")) >= 5
    ow = OracleWorkerUl(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], 3, 5, 8)
    for i in range(iq.shape[0]):
        ow.work_ul(iq[i, 0], iq[i, 1], tti0 + i)
    ul = [r for r in parse_pcap(ow.pcap_bytes()) if r["direction"] == 0]
    got = {(r["sfn"] * 10 + r["sf"], r["rnti"], r["pdu"]) for r in ul}
    hop_sent = [s for s in sent if s.get("hop") and s["tti"] >= tti0 + 30]
    assert hop_sent and sum((s["tti"], s["rnti"], s["payload"]) in got for s in hop_sent) >= 0.6 * len(hop_sent)


def test_ul_mode_worker_learns_beta_offsets_and_cqi_mode():
    """UEs that received an RRCConnectionSetup transmit their PUSCH control information with their own betaOffset indices and aperiodic
    CQI mode (UL_Sniffer_PUSCH.cc:433-435): the worker that learns them from the downlink keeps decoding, a worker that is kept on the
    defaults (10 / 8 / 11, sub-band reports) loses the grants that carry control information"""
    import ctypes as C
    from lsn_testlib import OracleWorkerUl, OUeCfg, gen_ul_mode_subframes, parse_pcap, scenario
    sc = scenario("cfg2", seed=33, nof_rx=1, n_rnti=6, dl_min=3, dl_max=4, ul_min=2, ul_max=3, nof_prb=25, mcs_max=16, msg4_period=6, msg4_p_a_idx=4, pct_cqi_req=60)
    nsf = 90
    tti0, iq, sent = gen_ul_mode_subframes(sc, nsf)
    ow = OracleWorkerUl(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], 3, 5)
    for i in range(nsf):
        ow.work_ul(iq[i, 0], iq[i, 1], tti0 + i)
    recs = parse_pcap(ow.pcap_bytes())
    ul = {(r["sfn"] * 10 + r["sf"], r["rnti"], r["pdu"]) for r in recs if r["direction"] == 0}
    setups = [r for r in recs if r["direction"] == 1 and r["rnti_type"] == 3 and r["pdu"][:1] == b"\x3c"]
    assert len(setups) >= 4
    late = [s for s in sent if s["tti"] >= tti0 + 40]
    hit = sum((s["tti"] % 10240, s["rnti"], s["payload"]) in ul for s in late)
    assert late and hit >= 0.7 * len(late), (hit, len(late))
    # the learned configurations differ from the defaults for at least one UE
    ow.lib.o_worker_ue_cfg.argtypes = [C.c_void_p, C.c_uint16, C.POINTER(OUeCfg)]
    learned = set()
    for r in setups:
        c = OUeCfg()
        ow.lib.o_worker_ue_cfg(ow.h, r["rnti"], C.byref(c))
        assert c.has_ue_config == 1
        learned.add((c.i_offset_ack, c.i_offset_cqi, c.i_offset_ri, c.cqi_type))
    assert any(x != (10, 8, 11, 2) for x in learned)


def _ul_ageing_stream(nsf=3150):
    from lsn_testlib import gen_ul_mode_subframes, scenario
    sc = scenario("cfg2", seed=41, nof_rx=1, n_rnti=8, dl_min=1, dl_max=2, ul_min=1, ul_max=2, nof_prb=15, mcs_max=22, cfi=3, rar_period=180)
    return (sc,) + gen_ul_mode_subframes(sc, nsf)


def test_ul_tracking_database_statistics_and_ageing():
    """MCSTracking's uplink database (update_statistic_ul :729-754 behind the 1 dB gate of UL_Sniffer_PUSCH.cc:571-575, update_database_ul
    :86-176 every interval x 1000 subframes): every UE with a counted PUSCH decode gets an entry - so the first successful decode above
    MCS 20 already fixes its maximum modulation -, entries of UEs that left the cell are dropped one interval later"""
    import ctypes as C
    from lsn_testlib import OracleWorkerUl, parse_pcap
    sc, tti0, iq, sent = _ul_ageing_stream()
    ow = OracleWorkerUl(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], 3, 5)
    ow.set_mcs_update_interval(1)
    ow.lib.o_worker_nof_tracked_ul.argtypes = [C.c_void_p]
    ow.lib.o_worker_tracked_mod_ul.argtypes = [C.c_void_p, C.c_uint16]
    seen, counts = set(), []
    last_tx = {}
    for i in range(iq.shape[0]):
        ow.work_ul(iq[i, 0], iq[i, 1], tti0 + i)
        counts.append(ow.lib.o_worker_nof_tracked_ul(ow.h))
    for s in sent:
        seen.add(s["rnti"]); last_tx[s["rnti"]] = (s["tti"] - tti0) % 10240
    n = iq.shape[0]
    assert len(seen) > 20                     # random access brought new UEs in
    # the passes at 1000 and 2000 subframes find nothing idle for more than one whole second, the one at 3000 drops the UEs that left early
    assert counts[1001] >= counts[999] and counts[2001] >= counts[1999] and counts[3001] <= counts[2999] - 4
    gone = [r for r in seen if last_tx[r] < 990]
    live = [r for r in seen if last_tx[r] > n - 100]
    assert gone and live
    # (one survivor allowed: a false DCI 0 that happens to carry the RNTI of a tracked UE refreshes its time stamp, find_tracking_info_RNTI_ul :52-53)
    assert sum(ow.lib.o_worker_tracked_mod_ul(ow.h, r) == 0 for r in gone) >= len(gone) - 1
    assert sum(ow.lib.o_worker_tracked_mod_ul(ow.h, r) >= 2 for r in live) >= len(live) - 2   # 16QAM / 64QAM maximum learnt
    ul = [r for r in parse_pcap(ow.pcap_bytes()) if r["direction"] == 0]
    assert len(ul) > 0.7 * len([s for s in sent if (s["tti"] - tti0) % 10240 > 20])


def test_ul_mode_worker_learns_256qam_uplink_table():
    """UEs on the 256QAM uplink table (every fourth RNTI): the worker's 16QAM / 64QAM-table attempts fail, the 256QAM-table attempt passes;
    the tracking database then holds 256QAM_MAX for them (both rules of decode_run, UL_Sniffer_PUSCH.cc:287-303) and later grants decode at
    the first attempt"""
    import ctypes as C
    from lsn_testlib import OracleWorkerUl, gen_ul_mode_subframes, parse_pcap, scenario
    sc = scenario("cfg2", seed=52, nof_rx=1, n_rnti=12, dl_min=1, dl_max=2, ul_min=2, ul_max=3, nof_prb=25, mcs_max=28, cfi=3)
    tti0, iq, sent = gen_ul_mode_subframes(sc, 300, ul_256=True, ul_snr_db=32.0)
    ow = OracleWorkerUl(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], 3, 5)
    ow.lib.o_worker_tracked_mod_ul.argtypes = [C.c_void_p, C.c_uint16]
    its = []
    for i in range(300):
        ow.work_ul(iq[i, 0], iq[i, 1], tti0 + i)
    ul = {(r["sfn"] * 10 + r["sf"], r["rnti"], r["pdu"]) for r in parse_pcap(ow.pcap_bytes()) if r["direction"] == 0}
    r256 = sorted({s["rnti"] for s in sent if s["rnti"] % 4 == 0})
    assert len(r256) >= 3 and all(ow.lib.o_worker_tracked_mod_ul(ow.h, r) == 4 for r in r256)
    assert all(ow.lib.o_worker_tracked_mod_ul(ow.h, r) in (2, 3) for r in {s["rnti"] for s in sent} - set(r256))
    late = [s for s in sent if s["tti"] >= tti0 + 100 and s["rnti"] % 4 == 0]
    hit = sum((s["tti"] % 10240, s["rnti"], s["payload"]) in ul for s in late)
    assert len(late) > 30 and hit >= 0.7 * len(late), (hit, len(late))
    assert any(s["mod"] == 8 for s in late)


def test_mixed_radix_idft_matches_numpy_for_every_allocation_size():
    """o_idft_mixed (the operation order k_pusch_demod reproduces) against numpy's inverse FFT for all 34 valid PUSCH sizes"""
    import ctypes as C
    from lsn_testlib import oracle
    o = oracle()
    o.o_idft_table.argtypes = [C.c_int, C.c_void_p]
    o.o_idft_mixed.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.RandomState(1)
    sizes = [L for L in range(1, 101) if o.o_ul_valid_prb(L)]
    assert len(sizes) == 34
    for L in sizes:
        M = 12 * L
        w = np.zeros(M, np.complex64)
        o.o_idft_table(M, w.ctypes.data)
        x = (rng.randn(M) + 1j * rng.randn(M)).astype(np.complex64)
        ref = np.fft.ifft(x.astype(np.complex128)) * M
        y, t = x.copy(), np.zeros(M, np.complex64)
        o.o_idft_mixed(M, w.ctypes.data, y.ctypes.data, t.ctypes.data)
        assert np.abs(y - ref).max() <= 3e-6 * np.abs(ref).max(), L


# ---------------------------------------------------------------------------------------------------- reference signals, independently
def _gold(cinit, n):
    """36.211 7.2, written from the definition (x1 / x2 as Python ints), independent of oracle / transmitter code"""
    x1, x2 = 1, cinit & 0x7FFFFFFF
    out = []
    for i in range(1600 + n):
        if i >= 1600:
            out.append((x1 ^ x2) & 1)
        x1 = (x1 >> 1) | ((((x1 >> 3) ^ x1) & 1) << 30)
        x2 = (x2 >> 1) | ((((x2 >> 3) ^ (x2 >> 2) ^ (x2 >> 1) ^ x2) & 1) << 30)
    return out


def _np_dmrs(cell_id, cyclic_shift, delta_ss, group_hop, seq_hop, ns, n_dmrs_dci, M):
    """r_PUSCH of slot ns, 36.211 5.5.1 / 5.5.2.1.1 in numpy (double precision, closed-form exponentials)"""
    import re
    import ast
    import os
    fss = (cell_id % 30 + delta_ss) % 30
    fgh = 0
    if group_hop:
        c = _gold(cell_id // 30, 8 * 20)
        fgh = sum(c[8 * ns + i] << i for i in range(8)) % 30
    u = (fgh + fss) % 30
    v = 0
    if seq_hop and not group_hop and M >= 72:
        v = _gold((cell_id // 30) * 32 + fss, 20)[ns]
    n = np.arange(M)
    if M in (12, 24):
        src = open(os.path.join(os.path.dirname(__file__), "..", "spec", "gen_tables.py")).read()
        tab = ast.literal_eval(re.search(r"^PHI%d=(\[\[.*\]\])$" % M, src, re.M).group(1))
        base = np.exp(1j * np.pi * np.array(tab[u]) / 4)
    else:
        nzc = max(p for p in range(2, M) if all(p % d for d in range(2, int(p ** 0.5) + 1)))
        qb = nzc * (u + 1) / 31.0
        q = int(np.floor(qb + 0.5)) + v * (-1) ** int(np.floor(2 * qb))
        m = n % nzc
        base = np.exp(-1j * np.pi * q * m * (m + 1) / nzc)
    d1, d2 = [0, 2, 3, 4, 6, 8, 9, 10], [0, 6, 3, 4, 2, 8, 10, 9]
    c = _gold((cell_id // 30) * 32 + fss, 8 * 7 * 20 + 8)
    npn = sum(c[8 * 7 * ns + i] << i for i in range(8))
    ncs = (d1[cyclic_shift] + d2[n_dmrs_dci] + npn) % 12
    return base * np.exp(2j * np.pi * ncs * n / 12)


@pytest.mark.parametrize("group_hop,seq_hop", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_reference_signal_equals_an_independent_numpy_generator(group_hop, seq_hop):
    """every allocation size x a sample of cells, slots and shifts, with group / sequence hopping on and off"""
    o = oracle_ul_api()
    o.o_dmrs_pusch.argtypes = [C.POINTER(OCell), C.POINTER(OUlCfg), C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    rng = np.random.default_rng(7 + 2 * group_hop + seq_hop)
    differs = 0
    for L in VALID_UL_PRB:
        for _ in range(3):
            cid, cs, dss = int(rng.integers(0, 504)), int(rng.integers(0, 8)), int(rng.integers(0, 30))
            ns, nd = int(rng.integers(0, 20)), int(rng.integers(0, 8))
            ocell, ucfg = OCell(100, 1, cid, 1), OUlCfg(cs, dss, 0, group_hop, seq_hop)
            r = np.zeros(12 * L, dtype=np.complex64)
            assert o.o_dmrs_pusch(C.byref(ocell), C.byref(ucfg), ns, nd, 12 * L, r.ctypes.data) == 0
            ref = _np_dmrs(cid, cs, dss, group_hop, seq_hop, ns, nd, 12 * L)
            assert np.abs(r - ref).max() < 2e-6, (L, cid, ns)
            plain = _np_dmrs(cid, cs, dss, 0, 0, ns, nd, 12 * L)
            differs += int(np.abs(ref - plain).max() > 1e-3)
    if group_hop:
        assert differs > 60          # almost every slot lands in another sequence group
    elif seq_hop:
        assert 10 < differs < 60     # only >= 6 PRB, and only where c(ns) = 1
    else:
        assert differs == 0


@pytest.mark.parametrize("group_hop,seq_hop", [(1, 0), (0, 1)])
def test_pusch_loopback_with_group_and_sequence_hopping(group_hop, seq_hop):
    o = oracle_ul_api()
    nprb, cell_id = 50, 77
    rng = np.random.default_rng(31 + group_hop)
    ocell = OCell(nprb, 1, cell_id, 1)
    ucell, ucfg, plain = TxgUlCell(nprb, cell_id, 2, 9, group_hop, seq_hop), OUlCfg(2, 9, 0, group_hop, seq_hop), OUlCfg(2, 9, 0, 0, 0)
    ok = wrong_cfg_fails = 0
    for it in range(5):
        tti = int(rng.integers(0, 10240))
        grants, start = [], 0
        for L in (1, 2, 3, 6, 8, 12):
            qm, tbs = ul_mcs_to_mod_tbs(int(rng.integers(2, 20)), L)
            grants.append(dict(rnti=int(rng.integers(100, 60000)), n_dmrs=int(rng.integers(0, 8)), n_prb=start, L_prb=L, mod=qm, tbs=tbs, rv=0,
                               phase_rad=float(rng.uniform(0, 6.28)), ta_samples=float(rng.uniform(0, 3))))
            start += L + 1
        iq, payloads = ul_make_subframe(ucell, tti, grants, snr_db=30.0, seed=it)
        grid = np.zeros(14 * 12 * nprb, dtype=np.complex64)
        o.o_ul_fft(C.byref(ocell), iq.ctypes.data, grid.ctypes.data)
        for g, pl in zip(grants, payloads):
            og = OPuschGrant(g["L_prb"], g["n_prb"], 0, g["mod"], g["tbs"], 0)
            out = np.zeros(g["tbs"] // 8 + 8, dtype=np.uint8)
            its, snr = C.c_int(0), C.c_float(0)
            assert o.o_pusch_decode(C.byref(ocell), C.byref(ucfg), tti % 10, g["rnti"], C.byref(og), g["n_dmrs"], grid.ctypes.data, 12, out.ctypes.data,
                                    C.byref(its), C.byref(snr)) == 1, (it, g)
            assert bytes(out[:g["tbs"] // 8]) == pl
            ok += 1
            # a receiver that ignores the hopping configuration estimates the channel against the wrong sequence
            if o.o_pusch_decode(C.byref(ocell), C.byref(plain), tti % 10, g["rnti"], C.byref(og), g["n_dmrs"], grid.ctypes.data, 4, out.ctypes.data,
                                C.byref(its), C.byref(snr)) == 0:
                wrong_cfg_fails += 1
    assert ok == 30 and wrong_cfg_fails >= (20 if group_hop else 3)
