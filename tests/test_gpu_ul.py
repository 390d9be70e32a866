"""GPU parity of the uplink chain (k_ul_fft, k_pusch_chest, k_pusch_demod, k_turbo) against the CPU oracle: bit-exact
uplink grid, bit-exact rate-matched LLRs, identical CRC verdict / iteration count / payload / SNR estimate."""
import ctypes as C

import numpy as np
import pytest

import ltesniffer_amd as la
from lsn_testlib import (OCell, OPuschGrant, OUci, OUlCfg, TxgUlCell, VALID_UL_PRB, oracle_ul_api, ul_make_subframe, ul_mcs_to_mod_tbs)

pytestmark = pytest.mark.gpu


def _scenario(nprb, cell_id, nsf, seed, snr_db=32.0, max_share=3, rvs=(0,), uci=False, group_hop=0, seq_hop=0):
    rng = np.random.default_rng(seed)
    cqi_bits = oracle_ul_api().o_uci_cqi_bits(nprb)
    ucell = TxgUlCell(nprb, cell_id, 3, 5, group_hop, seq_hop)
    N = {25: 512, 50: 1024, 75: 1536, 100: 2048}[nprb]
    iq = np.zeros((nsf, 15 * N), dtype=np.complex64)
    grants, payloads = [], []
    tti0 = int(rng.integers(0, 10000))
    for sf in range(nsf):
        gl, start = [], 0
        while True:
            L = int(rng.choice([n for n in VALID_UL_PRB if n <= max(3, nprb // max_share)]))  # 1 / 2 PRB: tabulated DMRS, >= 3: Zadoff-Chu
            if start + L > nprb:
                break
            mcs = int(rng.integers(0, 29))
            qm, tbs = ul_mcs_to_mod_tbs(mcs, L, enable_64qam=bool(rng.integers(0, 2)))
            if rng.integers(0, 6) == 0 and L >= 3:
                qm = 8  # exercise the 256QAM demapper with the same TBS
            gl.append(dict(sf=sf, rnti=int(rng.integers(100, 60000)), n_dmrs=int(rng.integers(0, 8)), n_prb=start, L_prb=L, mod=qm, tbs=tbs,
                           rv=int(rng.choice(rvs)), gain_db=float(rng.uniform(-3, 3)), phase_rad=float(rng.uniform(0, 6.28)), ta_samples=float(rng.uniform(0, 3))))
            if uci:  # control information multiplexed into the PUSCH: HARQ-ACK bits, aperiodic CQI report + RI
                cq = int(rng.integers(0, 2)) * cqi_bits
                gl[-1].update(nof_ack=int(rng.integers(0, 3)), cqi_bits=cq, ri_bits=1 if cq else 0)
            start += L + int(rng.integers(0, 2))
        iq[sf], pl = ul_make_subframe(ucell, tti0 + sf, gl, snr_db=snr_db, seed=seed * 100 + sf)
        grants += gl
        payloads += pl
    return tti0, iq, grants, payloads


def _run(nprb, cell_id, nsf, seed, **kw):
    o = oracle_ul_api()
    tti0, iq, grants, payloads = _scenario(nprb, cell_id, nsf, seed, **kw)
    gh, sh = kw.get("group_hop", 0), kw.get("seq_hop", 0)
    ocell, ucfg = OCell(nprb, 1, cell_id, 1), OUlCfg(3, 5, 0, gh, sh)
    phy = la.Phy(nof_rx_antennas=1)
    assert phy.setCell(nprb, 1, cell_id) and phy.setUlConfig(3, 5, 0, gh, sh)
    res = phy.pusch_decode(iq, tti0, grants)
    nre = 12 * nprb
    grids = []
    for sf in range(nsf):
        g = np.zeros(14 * nre, dtype=np.complex64)
        o.o_ul_fft(C.byref(ocell), iq[sf].ctypes.data, g.ctypes.data)
        grids.append(g)
        assert np.array_equal(phy.tap_ul_grid(sf).reshape(-1).view(np.uint32), g.view(np.uint32)), sf
    n_ok = 0
    for i, (g, pl, r) in enumerate(zip(grants, payloads, res)):
        og = OPuschGrant(g["L_prb"], g["n_prb"], 0, g["mod"], g["tbs"], g["rv"], g.get("n_prb2", 0), g.get("hop", 0))
        uci = OUci(g.get("nof_ack", 0), g.get("cqi_bits", 0), g.get("ri_bits", 0))
        M = 12 * g["L_prb"]
        cls_buf, idx_buf = np.zeros(12 * M, np.uint8), np.zeros(12 * M, np.int32)
        nsym = o.o_uci_layout(M, g["tbs"], C.byref(uci), cls_buf.ctypes.data, idx_buf.ctypes.data, None, None, None)
        G = nsym * g["mod"]
        if G <= 0:  # the control information takes every resource of the allocation (one PRB with a CQI report): nothing to decode, on both sides
            assert r["crc_ok"] == 0
            out = np.zeros(g["tbs"] // 8 + 8, dtype=np.uint8)
            assert o.o_pusch_decode_uci(C.byref(ocell), C.byref(ucfg), (tti0 + g["sf"]) % 10, g["rnti"], C.byref(og), g["n_dmrs"], C.byref(uci),
                                        grids[g["sf"]].ctypes.data, 12, out.ctypes.data, None, None) == 0
            continue
        e = np.zeros(144 * g["L_prb"] * g["mod"], dtype=np.int16)
        noise, sig = C.c_float(), C.c_float()
        assert o.o_pusch_demod_uci(C.byref(ocell), C.byref(ucfg), (tti0 + g["sf"]) % 10, g["rnti"], C.byref(og), g["n_dmrs"], C.byref(uci),
                                   grids[g["sf"]].ctypes.data, e.ctypes.data, C.byref(noise), C.byref(sig)) == 0
        assert np.array_equal(phy.tap_ul_llr(i, G), e[:G]), (i, g)
        out = np.zeros(g["tbs"] // 8 + 8, dtype=np.uint8)
        its, snr = C.c_int(0), C.c_float(0)
        crc = o.o_pusch_decode_uci(C.byref(ocell), C.byref(ucfg), (tti0 + g["sf"]) % 10, g["rnti"], C.byref(og), g["n_dmrs"], C.byref(uci),
                                   grids[g["sf"]].ctypes.data, 12, out.ctypes.data, C.byref(its), C.byref(snr))
        assert r["crc_ok"] == crc and r["iterations"] == its.value, (i, g, r, crc, its.value)
        assert np.float32(r["snr_db"]).view(np.uint32) == np.float32(snr.value).view(np.uint32)
        if crc:
            assert r["payload"] == bytes(out[:g["tbs"] // 8]) == pl
            n_ok += 1
    phy.close()
    return n_ok, len(grants)


def test_pusch_25prb():
    ok, n = _run(25, 7, 6, seed=1)
    assert ok >= n * 0.6


def test_pusch_50prb_rv():
    ok, n = _run(50, 101, 5, seed=2, rvs=(0, 0, 2, 3, 1))
    assert ok >= n * 0.5


def test_pusch_100prb_wideband_and_low_snr():
    ok, n = _run(100, 1, 4, seed=3, max_share=1)      # allocations up to 100 PRB (M = 1200, 13+ code blocks)
    assert ok >= 1
    _run(100, 1, 3, seed=4, snr_db=8.0, max_share=4)   # many CRC failures: verdicts and iteration counts still identical


def test_pusch_group_and_sequence_hopping_and_two_prb():
    """SIB2 groupHoppingEnabled / sequenceHoppingEnabled (ULSchedule.cc:143-146): the reference signal of every slot comes from that slot's
    sequence group / number (36.211 5.5.1.3-4); 2-PRB allocations use Table 5.5.1.2-2 (UL_Sniffer_PUSCH.cc:3-10 accepts them)"""
    for gh, sh, seed in ((1, 0, 51), (0, 1, 52), (0, 0, 53)):
        ok, n = _run(50, 33, 5, seed=seed, max_share=2, group_hop=gh, seq_hop=sh)
        assert n >= 10 and ok >= 0.6 * n, (gh, sh, ok, n)


def test_pusch_with_uci_multiplexing():
    """HARQ-ACK puncturing, RI and CQI cells (36.212 5.2.2.6-8) located by the closed form of k_pusch_demod = the oracle's literal matrix"""
    for nprb, seed in ((25, 41), (100, 42)):
        n_ok, n = _run(nprb, 7, 4, seed, uci=True)
        assert n >= 8 and n_ok >= n - 3, (nprb, n_ok, n)   # (a 1- / 2-PRB allocation under a CQI report has too few resources left to decode - on both sides)


def test_pusch_type1_frequency_hopping():
    """slot 1 on other PRBs (36.213 8.4.1): k_pusch_chest / k_pusch_demod read each slot at its own offset"""
    from lsn_testlib import pusch_hop_slot1
    o = oracle_ul_api()
    nprb, cell_id, off = 100, 5, 10
    ocell, ucell, ucfg = OCell(nprb, 1, cell_id, 1, off), TxgUlCell(nprb, cell_id, 3, 5), OUlCfg(3, 5, off)
    rng = np.random.default_rng(3)
    grants, tti0 = [], 4321
    iq = np.zeros((3, 15 * 2048), dtype=np.complex64)
    payloads = []
    for sf in range(3):
        hop_bits = [0, 1, 2][sf]
        n_prb = [6, 40, 9][sf]
        L = [5, 4, 12][sf]
        qm, tbs = ul_mcs_to_mod_tbs(int(rng.integers(4, 20)), L)
        g = dict(sf=sf, rnti=int(rng.integers(100, 60000)), n_dmrs=int(rng.integers(0, 8)), n_prb=n_prb, L_prb=L, mod=qm, tbs=tbs, rv=0, hop=1,
                 n_prb2=pusch_hop_slot1(nprb, off, hop_bits, n_prb), nof_ack=sf % 3)
        iq[sf], pl = ul_make_subframe(ucell, tti0 + sf, [g], snr_db=30.0, seed=sf)
        grants.append(g); payloads += pl
    phy = la.Phy(nof_rx_antennas=1)
    assert phy.setCell(nprb, 1, cell_id) and phy.setUlConfig(3, 5, off)
    res = phy.pusch_decode(iq, tti0, grants)
    for i, (g, pl, r) in enumerate(zip(grants, payloads, res)):
        grid = np.zeros(14 * 12 * nprb, dtype=np.complex64)
        o.o_ul_fft(C.byref(ocell), iq[g["sf"]].ctypes.data, grid.ctypes.data)
        og = OPuschGrant(g["L_prb"], g["n_prb"], 0, g["mod"], g["tbs"], 0, g["n_prb2"], 1)
        uci = OUci(g["nof_ack"], 0, 0)
        e = np.zeros(144 * g["L_prb"] * g["mod"], dtype=np.int16)
        assert o.o_pusch_demod_uci(C.byref(ocell), C.byref(ucfg), (tti0 + g["sf"]) % 10, g["rnti"], C.byref(og), g["n_dmrs"], C.byref(uci), grid.ctypes.data,
                                   e.ctypes.data, None, None) == 0
        assert np.array_equal(phy.tap_ul_llr(i, e.size), e), i
        assert r["crc_ok"] == 1 and r["payload"] == pl, (i, g)
    phy.close()


def test_pusch_unsupported_grants_fail_cleanly():
    phy = la.Phy(nof_rx_antennas=1)
    assert phy.setCell(25, 1, 3)
    with pytest.raises(RuntimeError):
        phy.pusch_decode(np.zeros((1, 15 * 512), dtype=np.complex64), 0, [])  # no UL config yet
    assert phy.setUlConfig(0, 0) and not phy.setUlConfig(9, 0)
    iq = np.zeros((2, 15 * 512), dtype=np.complex64)
    bad = [dict(sf=0, rnti=70, n_prb=0, L_prb=2, mod=2, tbs=104), dict(sf=0, rnti=70, n_prb=0, L_prb=7, mod=2, tbs=104),
           dict(sf=0, rnti=70, n_prb=0xFFFFFFFD, L_prb=3, mod=2, tbs=104),  # n_prb + L_prb wraps around in 32 bits
           dict(sf=5, rnti=70, n_prb=0, L_prb=3, mod=2, tbs=104), dict(sf=0, rnti=70, n_prb=24, L_prb=3, mod=2, tbs=104),
           dict(sf=1, rnti=70, n_prb=0, L_prb=3, mod=3, tbs=104), dict(sf=1, rnti=70, n_prb=0, L_prb=3, mod=2, tbs=0)]
    res = phy.pusch_decode(iq, 0, bad)
    assert all(r["crc_ok"] == 0 for r in res)
    phy.close()


def _run_ul_mode(nsf, seed, batch, hopping_offset=0, ul_256=False, ul_snr_db=30.0, **over):
    from lsn_testlib import OracleWorkerUl, gen_ul_mode_subframes, parse_pcap, scenario
    from parity import gpu_records, oracle_records
    kw = dict(nof_rx=1, n_rnti=12, dl_min=2, dl_max=3, ul_min=2, ul_max=4)
    kw.update(over)
    sc = scenario("cfg2", seed=seed, **kw)
    tti0, iq, sent = gen_ul_mode_subframes(sc, nsf, ul_256=ul_256, ul_snr_db=ul_snr_db)
    ow = OracleWorkerUl(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], 3, 5, hopping_offset, cp=sc.get("cp", 0))
    for i in range(nsf):
        ow.work_ul(iq[i, 0], iq[i, 1], tti0 + i, update_meta=1 if i % 25 == 0 else 0)
    orecs = parse_pcap(ow.pcap_bytes())
    phy = la.Phy(nof_rx_antennas=2, sniffer_mode=1, max_batch=batch, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], cp=sc.get("cp", 0)) and phy.setUlConfig(3, 5, hopping_offset)
    phy.process_host(iq, tti0, 25)
    g, o = gpu_records(phy), oracle_records(orecs)
    assert g == o, "UL_MODE record streams differ: gpu %d vs oracle %d" % (len(g), len(o))
    st, ost = phy.getStats(), ow.stats()
    assert st.nof_decoded_locations == ost.nof_decoded_locations and st.nof_subframes == ost.nof_subframes
    phy.close()
    ul = [r for r in orecs if r["direction"] == 0]
    dl = [r for r in orecs if r["direction"] == 1]
    return len(ul), len(dl)


def test_ul_mode_end_to_end_matches_oracle():
    """UL_MODE through the C ABI: antenna 0 = downlink, antenna 1 = uplink; DCI 0 at t -> PUSCH at t + 4; uplink MCS-table trials;
    pcap records (downlink RAR / format 1-1A PDUs + uplink PDUs) identical to the oracle's UL_MODE worker"""
    n_ul, n_dl = _run_ul_mode(60, seed=5, batch=16, mcs_max=20)
    assert n_ul >= 10 and n_dl >= 5


def test_ul_mode_on_a_four_port_cell():
    """the downlink half of UL_MODE (one antenna) on four CRS ports: PDCCH in SFBC-FSTD, DCI 0 found, PUSCH decoded as on any other cell"""
    n_ul, n_dl = _run_ul_mode(60, seed=6, batch=16, mcs_max=20, nof_ports=4, rar_period=15)
    assert n_ul >= 10 and n_dl >= 5


def test_ul_mode_extended_cyclic_prefix():
    """UL_MODE on an extended-CP cell: both grids have 12 symbols, the PUSCH 10 data symbols with the reference signal on symbol 2 of each slot, a
    10-column channel interleaver (control information on the extended-CP column sets) - record streams identical to the oracle's worker"""
    n_ul, n_dl = _run_ul_mode(60, seed=61, batch=16, mcs_max=20, cp=1)
    assert n_ul >= 10 and n_dl >= 5
    n_ul, n_dl = _run_ul_mode(48, seed=62, batch=64, rar_period=10, mcs_max=20, pct_cqi_req=50, cp=1, nof_prb=50)
    assert n_ul >= 5


def test_ul_mode_with_rar_and_single_chunk():
    n_ul, n_dl = _run_ul_mode(48, seed=9, batch=64, rar_period=10, mcs_max=20, pct_cqi_req=50)
    assert n_ul >= 5


def test_ul_mode_with_frequency_hopping_grants():
    """DCI 0 with the hopping flag: type-1 grants are decoded from both slot positions, like the oracle's UL_MODE worker"""
    n_ul, n_dl = _run_ul_mode(60, seed=15, batch=32, hopping_offset=8, mcs_max=18, pct_hop=50, pusch_hop_offset=8, pct_cqi_req=20)
    assert n_ul >= 8


def test_ul_mode_with_256qam_uplink_ues():
    """every fourth UE transmits with the 256QAM uplink table: the 16QAM- / 64QAM-table attempts fail, the 256QAM-table attempt passes and
    fixes the UE's maximum modulation - above MCS 20 through decode_run's first rule, below through its second (UL_Sniffer_PUSCH.cc:287-303)"""
    n_ul, n_dl = _run_ul_mode(160, seed=52, batch=32, ul_256=True, ul_snr_db=32.0, nof_prb=25, mcs_max=28, cfi=3, dl_min=1, dl_max=2, ul_max=3)
    assert n_ul >= 120


def test_ul_mode_baseline_size_64_rnti_400_subframes():
    """BASELINE configs[3]: 20 MHz, 64 RNTIs, every DCI 0 answered by a PUSCH 4 ms later (MCS 0-28, half of the UEs 64QAM-capable, uplink at
    22 dB), 400 subframes in chunks of 128 - record stream identical to the oracle's UL_MODE worker"""
    from lsn_testlib import OracleWorkerUl, gen_ul_mode_subframes, parse_pcap, scenario
    from parity import gpu_records, oracle_records
    sc = scenario("cfg2", seed=4, nof_rx=1, n_rnti=64, dl_min=4, dl_max=6, ul_min=2, ul_max=4, mcs_min=0, mcs_max=28, snr_db=28.0, pct_cqi_req=20, rar_period=90)
    nsf = 400
    tti0, iq, sent = gen_ul_mode_subframes(sc, nsf, ul_snr_db=22.0)
    ow = OracleWorkerUl(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], 3, 5)
    for i in range(nsf):
        ow.work_ul(iq[i, 0], iq[i, 1], tti0 + i, update_meta=1 if i % 100 == 0 else 0)
    orecs = parse_pcap(ow.pcap_bytes())
    phy = la.Phy(nof_rx_antennas=2, sniffer_mode=1, max_batch=128, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"]) and phy.setUlConfig(3, 5)
    phy.process_host(iq, tti0, 100)
    g, o = gpu_records(phy), oracle_records(orecs)
    assert g == o, "UL_MODE record streams differ: gpu %d vs oracle %d" % (len(g), len(o))
    st, ost = phy.getStats(), ow.stats()
    assert st.nof_decoded_locations == ost.nof_decoded_locations and st.nof_subframes == ost.nof_subframes
    assert len({s_["rnti"] for s_ in sent}) >= 60 and len([r for r in orecs if r["direction"] == 0]) >= 500
    phy.close()


@pytest.mark.parametrize("batch", [16, 64])
def test_ul_mode_configures_itself_from_sib2(batch):
    """UL_MODE without lsn_phy_set_ul_config: PDSCH_Decoder::decode_SIB on every subframe until the SystemInformation with SIB2 (here in
    the middle of a chunk), that one SI-RNTI record, then DMRS / hopping offset / PRACH from it - records, learned configuration and
    search statistics identical to the oracle's worker that started without a configuration"""
    from lsn_testlib import REAL_SIB1, OracleWorkerUl, encode_sib2, gen_ul_mode_subframes, parse_pcap, scenario
    from parity import gpu_records, oracle_records
    sc = scenario("cfg2", seed=21, nof_rx=1, n_rnti=10, dl_min=2, dl_max=3, ul_min=2, ul_max=4, nof_prb=25, mcs_max=18, pusch_hop_offset=4, pct_hop=30)
    sib2 = encode_sib2(cyclic_shift=3, group_assignment_pusch=5, pusch_hop_offset=4, root_seq_idx=22, prach_config_idx=3, zero_corr_zone=1,
                       prach_freq_offset=2, ac_barring=1, srs=1, rr_ext=1, nblocks=2, fill=5)
    nsf = 70
    tti0, iq, sent = gen_ul_mode_subframes(sc, nsf, si_msgs=[REAL_SIB1, sib2])
    ow = OracleWorkerUl(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], None, None)
    for i in range(nsf):
        ow.work_ul(iq[i, 0], iq[i, 1], tti0 + i, update_meta=1 if i % 25 == 0 else 0)
    orecs = parse_pcap(ow.pcap_bytes())
    assert orecs[0]["rnti_type"] == 4 and len([r for r in orecs if r["direction"] == 0]) >= 5
    phy = la.Phy(nof_rx_antennas=2, sniffer_mode=1, max_batch=batch, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    assert phy.getUlConfig() is None
    phy.process_host(iq, tti0, 25)
    g, o = gpu_records(phy), oracle_records(orecs)
    assert g == o, "UL_MODE (self-configured) record streams differ: gpu %d vs oracle %d" % (len(g), len(o))
    assert phy.getUlConfig() == ow.ul_config()
    st, ost = phy.getStats(), ow.stats()
    assert st.nof_decoded_locations == ost.nof_decoded_locations and st.nof_subframes == ost.nof_subframes
    # a configuration given by hand afterwards replaces the learned one
    assert phy.setUlConfig(1, 2, 0) and phy.getUlConfig() == dict(cyclic_shift=1, delta_ss=2, hopping_offset=0, group_hopping=0, sequence_hopping=0, from_sib2=False, sib2=None)
    assert la.sib2_decode(sib2)[1]["root_seq_idx"] == 22 and la.sib2_decode(REAL_SIB1) == (1, None)
    phy.close()


@pytest.mark.parametrize("gh,sh", [(1, 0), (0, 1)])
def test_ul_mode_sib2_with_group_or_sequence_hopping(gh, sh):
    """a SIB2 that switches group / sequence hopping of the PUSCH reference signal on (ULSchedule.cc:143-146 forwards both flags): the
    self-configured UL_MODE decodes the hopped transmissions, 1- and 2-PRB grants included; records identical to the oracle's"""
    from lsn_testlib import REAL_SIB1, OracleWorkerUl, encode_sib2, gen_ul_mode_subframes, parse_pcap, scenario
    from parity import gpu_records, oracle_records
    sc = scenario("cfg2", seed=61 + gh, nof_rx=1, n_rnti=10, dl_min=2, dl_max=3, ul_min=3, ul_max=5, nof_prb=50, mcs_max=18)
    sib2 = encode_sib2(cyclic_shift=2, group_assignment_pusch=7, group_hopping_enabled=gh, sequence_hopping_enabled=sh, root_seq_idx=40, prach_config_idx=3,
                       zero_corr_zone=5, prach_freq_offset=4)
    nsf = 80
    tti0, iq, sent = gen_ul_mode_subframes(sc, nsf, cyclic_shift=2, delta_ss=7, si_msgs=[REAL_SIB1, sib2], group_hopping=gh, sequence_hopping=sh)
    ow = OracleWorkerUl(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], None, None)
    for i in range(nsf):
        ow.work_ul(iq[i, 0], iq[i, 1], tti0 + i, update_meta=1 if i % 25 == 0 else 0)
    orecs = parse_pcap(ow.pcap_bytes())
    ul = [r for r in orecs if r["direction"] == 0]
    sent_set = {(s["tti"], s["rnti"], s["payload"]) for s in sent}
    assert len(ul) >= 20 and all((r["sfn"] * 10 + r["sf"], r["rnti"], r["pdu"]) in sent_set for r in ul)
    small = {(s["tti"], s["rnti"]) for s in sent if s["L_prb"] <= 2}
    assert sum((r["sfn"] * 10 + r["sf"], r["rnti"]) in small for r in ul) >= 2   # 1- / 2-PRB transmissions came through
    phy = la.Phy(nof_rx_antennas=2, sniffer_mode=1, max_batch=32, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    phy.process_host(iq, tti0, 25)
    g, o = gpu_records(phy), oracle_records(orecs)
    assert g == o, "UL_MODE (hopping reference signals) record streams differ: gpu %d vs oracle %d" % (len(g), len(o))
    cfg = phy.getUlConfig()
    assert cfg == ow.ul_config() and cfg["group_hopping"] == gh and cfg["sequence_hopping"] == sh
    p = phy.perf()
    assert p.nof_pusch_2prb_skipped == 0 and p.nof_pusch_on_unverified_dmrs > 0
    phy.close()


def test_ul_mode_learns_beta_offsets_and_cqi_mode():
    """RRCConnectionSetups decoded in UL_MODE feed the PUSCH decoder's control-information layout (betaOffset indices, aperiodic CQI
    mode; UL_Sniffer_PUSCH.cc:433-435): records and learned configurations identical to the oracle's, chunked so that setups fall inside
    and between chunks"""
    import ctypes as C
    from lsn_testlib import OracleWorkerUl, OUeCfg, gen_ul_mode_subframes, parse_pcap, scenario
    from parity import gpu_records, oracle_records
    sc = scenario("cfg2", seed=33, nof_rx=1, n_rnti=6, dl_min=3, dl_max=4, ul_min=2, ul_max=3, nof_prb=25, mcs_max=16, msg4_period=6, msg4_p_a_idx=4, pct_cqi_req=60)
    nsf = 90
    tti0, iq, sent = gen_ul_mode_subframes(sc, nsf)
    ow = OracleWorkerUl(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], 3, 5)
    for i in range(nsf):
        ow.work_ul(iq[i, 0], iq[i, 1], tti0 + i, update_meta=1 if i % 25 == 0 else 0)
    orecs = parse_pcap(ow.pcap_bytes())
    ow.lib.o_worker_ue_cfg.argtypes = [C.c_void_p, C.c_uint16, C.POINTER(OUeCfg)]
    for batch in (16, 90):
        phy = la.Phy(nof_rx_antennas=2, sniffer_mode=1, max_batch=batch, pcapwriter=la.PcapWriter(None))
        assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"]) and phy.setUlConfig(3, 5)
        phy.process_host(iq, tti0, 25)
        g, o = gpu_records(phy), oracle_records(orecs)
        assert g == o, "batch %d: UL_MODE record streams differ: gpu %d vs oracle %d" % (batch, len(g), len(o))
        nlearn = 0
        for rnti in sorted({r["rnti"] for r in orecs if r["rnti_type"] == 3}):
            c = OUeCfg()
            ow.lib.o_worker_ue_cfg(ow.h, rnti, C.byref(c))
            u = phy.ue_config(rnti)
            assert (u.has_ue_config, u.i_offset_ack, u.i_offset_cqi, u.i_offset_ri, u.cqi_type) == \
                (c.has_ue_config, c.i_offset_ack, c.i_offset_cqi, c.i_offset_ri, c.cqi_type), rnti
            nlearn += c.has_ue_config
        assert nlearn >= 3
        phy.close()
    assert len([r for r in orecs if r["direction"] == 0]) >= 20


def test_ul_tracking_database_statistics_and_ageing_match_oracle():
    """3150 UL_MODE subframes with UEs joining through random access: the uplink tracking database (entries from update_statistic_ul,
    maximum modulation learnt at the first success above MCS 20, entries dropped by update_database_ul every 1000 subframes) evolves
    like the oracle's - same records, same entry count, same modulation per RNTI"""
    import ctypes as C
    from lsn_testlib import OracleWorkerUl, parse_pcap
    from parity import gpu_records, oracle_records
    from test_ul_oracle import _ul_ageing_stream
    sc, tti0, iq, sent = _ul_ageing_stream()
    ow = OracleWorkerUl(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], 3, 5)
    ow.set_mcs_update_interval(1)
    ow.lib.o_worker_nof_tracked_ul.argtypes = [C.c_void_p]
    ow.lib.o_worker_tracked_mod_ul.argtypes = [C.c_void_p, C.c_uint16]
    for i in range(iq.shape[0]):
        ow.work_ul(iq[i, 0], iq[i, 1], tti0 + i, update_meta=1 if i % 500 == 0 else 0)
    orecs = parse_pcap(ow.pcap_bytes())
    phy = la.Phy(nof_rx_antennas=2, sniffer_mode=1, max_batch=200, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"]) and phy.setUlConfig(3, 5)
    phy.setMcsUpdateInterval(1)
    phy.process_host(iq, tti0, 500)
    g, o = gpu_records(phy), oracle_records(orecs)
    assert g == o, "UL_MODE record streams differ: gpu %d vs oracle %d" % (len(g), len(o))
    assert phy.nofTrackedRnti() == ow.lib.o_worker_nof_tracked_ul(ow.h) > 5
    rntis = sorted({s["rnti"] for s in sent})
    assert [phy.trackedUlModulation(r) for r in rntis] == [ow.lib.o_worker_tracked_mod_ul(ow.h, r) for r in rntis]
    assert len([r for r in orecs if r["direction"] == 0]) > 1500
    phy.close()
