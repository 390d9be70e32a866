"""GPU parity tests proper: HIP path (through the C ABI) vs. the CPU oracle on the same seeded inputs - bit-exact."""
import numpy as np
import pytest

import ltesniffer_amd as la
from lsn_testlib import scenario
from lsn_testlib import oracle_trace
from parity import CfoLoop, compare_candidate_tables, compare_stage_c, compare_taps, gen_subframes, gpu_records, oracle_records, run_oracle

pytestmark = pytest.mark.gpu


def _run(scn, nsf, seed=1, batch=16, update_meta_period=0, exact_iters=False, cfo_correction=None, **over):
    """stage-A taps, stage-C taps (int16 soft bits, de-rate-matched streams, per-code-block verdict + iterations), record stream and statistics
    of the HIP path against the oracle; exact_iters: the engine runs without first-block gating (LSN_NO_CB_SKIP=1), so every code block of
    every decode call the oracle made carries a verdict and the iteration totals must agree"""
    import os
    sc = scenario(scn, seed=seed, **over)
    tti0, iq, truth = gen_subframes(sc, nsf)
    loop = CfoLoop(*cfo_correction, batch) if cfo_correction else None   # (mode, start offset, alpha): the engine's correction loop restated for the oracle driver
    ow, per_sf, orecs = run_oracle(sc, tti0, iq, update_meta_period=update_meta_period, trace=True, cfo_loop=loop)
    otrace = oracle_trace()
    if exact_iters:
        os.environ["LSN_NO_CB_SKIP"] = "1"   # read when the engine is made
    try:
        phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=batch)
    finally:
        os.environ.pop("LSN_NO_CB_SKIP", None)
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], {1: 0, 3: 1, 6: 2, 12: 3}[sc["phich_ng_x6"]], cp=sc.get("cp", 0))
    phy.set_stage_c_taps(True)
    if cfo_correction:
        phy.setCfoCorrection(*cfo_correction)
    bad, badc, ncall, ncb, it_o, it_g = [], [], 0, 0, 0, 0
    for base in range(0, nsf, batch):
        n = min(batch, nsf - base)
        # update_meta_period counts subframes from the start of the stream, same as the oracle driver above
        phy.process_host(iq[base:base + n], tti0 + base, update_meta_period)
        bad += [(base,) + b for b in compare_taps(phy, per_sf, sc, base, n)]
        if loop and np.float32(phy.getCfoCorrection()) != np.float32(loop.hist[base // batch]):
            bad.append((base, "cfo correction", phy.getCfoCorrection(), loop.hist[base // batch]))
        b, c, k, io, ig = compare_stage_c(phy, otrace, tti0 + base, n, exact_iters=exact_iters)
        badc += [(base,) + x for x in b]
        ncall, ncb, it_o, it_g = ncall + c, ncb + k, it_o + io, it_g + ig
    assert not bad, bad[:5]
    assert not badc, (len(badc), badc[:5])
    assert ncall == len([o for o in otrace if not o["is_ul"]]) and (ncb > 0 or not orecs), (ncall, ncb)
    assert it_o == it_g
    if exact_iters:
        assert it_o == ow.total_iters() and ncb == sum(len(o["cbs"]) for o in otrace)
    g, o = gpu_records(phy), oracle_records(orecs)
    assert len(o) > 0
    assert g == o, "record streams differ: gpu %d vs oracle %d" % (len(g), len(o))
    st, ost = phy.getStats(), ow.stats()
    for f in ("nof_locations", "nof_decoded_locations", "nof_cce", "nof_missed_cce", "nof_subframes",
              "nof_subframe_collisions_dw", "nof_subframe_collisions_up"):
        assert getattr(st, f) == getattr(ost, f), f
    phy.close()
    if loop:
        return len(o), loop
    return len(o)


def test_small_cell_25prb():
    _run("small", 40, seed=11)


def test_cfg1_10mhz_tm1_qpsk():
    _run("cfg1", 40, seed=1, batch=20)


def test_cfg2_20mhz_tm2_64qam():
    _run("cfg2", 32, seed=2)


def test_cfg3_20mhz_150rnti_tm34_256qam():
    _run("cfg3", 48, seed=3, update_meta_period=20)


def test_6prb_and_15prb():
    _run("small", 20, seed=5, nof_prb=6, cfi=3, dl_min=1, dl_max=1, n_rnti=2)
    _run("small", 20, seed=6, nof_prb=15, cfi=2, dl_min=1, dl_max=2, n_rnti=3)


def test_four_crs_ports_100prb_transmit_diversity():
    """four CRS ports (VERDICT r3 missing 1): k_chest for ports 2 / 3 (pilot symbols 1, 8), the REG map with 6-RE REGs in symbol 1, SFBC-FSTD in
    k_pcfich / k_pdcch_llr / k_pdsch_demod, the four-port RE masks and DCI sizes - every tap bit-exact, record stream identical"""
    n = _run("cfg2", 32, seed=4, nof_ports=4, cfi=0, rar_period=16, paging_period=8)
    assert n > 100


def test_four_crs_ports_small_cells_and_one_rx_antenna():
    _run("small", 30, seed=12, nof_ports=4)
    _run("small", 20, seed=13, nof_ports=4, nof_prb=6, cfi=0, dl_min=1, dl_max=1, n_rnti=2)
    _run("small", 20, seed=14, nof_ports=4, nof_prb=75, nof_rx=1, cell_id=77, cfi=0)
    _run("small", 20, seed=15, nof_ports=4, nof_prb=15, cell_id=500, phich_ng_x6=12, cfi=0, snr_db=9.0)


def test_four_crs_ports_tm3_tm4_grants_are_found_and_not_decoded():
    """format 2 / 2A grants on a four-port cell ask for spatial multiplexing: accepted by the search at their four-port sizes, no decode job, no
    record - the oracle's (and srsRAN's) behaviour; the other grants of the capture decode as usual"""
    _run("cfg3", 40, seed=16, nof_ports=4, nof_prb=50, n_rnti=20, update_meta_period=20)


def test_extended_cyclic_prefix_25_and_100_prb():
    """extended CP (round-4 review, missing 1; the reference hands whatever cell its search found to srsran_ue_dl_set_cell, SubframeWorker.cc:102):
    12-symbol subframes with a CP of N / 4, CRS on symbols 0 / 3 of each slot (ports 2, 3: symbol 1), N_CP = 0 in their sequences, the time
    interpolation between pilots 0, 3, 6, 9, the PDSCH RE masks (PSS / SSS on symbols 5 / 4, PBCH on 6 - 9), rho_B symbols - every stage-A tap,
    every stage-C tap and the record stream identical to the oracle's, TM2 / TM3 / TM4 up to 256QAM at 100 PRB"""
    n = _run("small", 40, seed=31, cp=1)
    assert n > 40
    n = _run("cfg3", 40, seed=32, cp=1, n_rnti=40, update_meta_period=20)
    assert n > 150


def test_extended_cyclic_prefix_ports_bandwidths_and_four_control_symbols():
    """one and four CRS ports, one rx antenna, 6 PRB with CFI 3 (four control symbols: symbol 3 carries CRS -> 6-RE REGs), 15 / 50 / 75 PRB"""
    _run("small", 20, seed=33, cp=1, nof_prb=6, cfi=3, dl_min=1, dl_max=1, n_rnti=2)
    _run("small", 20, seed=34, cp=1, nof_prb=6, cfi=0, nof_ports=4, dl_min=1, dl_max=1, n_rnti=2)
    _run("small", 20, seed=35, cp=1, nof_prb=50, nof_ports=1, nof_rx=1, cfo_hz=200.0)
    _run("small", 20, seed=36, cp=1, nof_prb=75, nof_ports=4, cfi=0, cell_id=77)
    _run("small", 20, seed=37, cp=1, nof_prb=15, cfi=0, cell_id=500, phich_ng_x6=12, snr_db=12.0)


def test_worker_pool_api_matches_oracle():
    """the reference's own call pattern: getAvail -> fill buffers -> prepare -> putPending ... joinPending"""
    sc = scenario("small", seed=21)
    nsf = 30
    tti0, iq, _ = gen_subframes(sc, nsf)
    _, _, orecs = run_oracle(sc, tti0, iq, taps=False)
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], nof_workers=8, max_batch=8, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    for i in range(nsf):
        w = phy.getAvail()
        bufs = w.getBuffers()
        assert len(bufs) == sc["nof_rx"] and len(bufs[0]) == 3 * iq.shape[2]
        for rx in range(sc["nof_rx"]):
            bufs[rx][:iq.shape[2]] = iq[i, rx]
        tti = tti0 + i
        w.prepare(tti % 10, (tti // 10) % 1024, False)
        assert w.getSfidx() == tti % 10
        phy.putPending(w)
    phy.joinPending()
    assert gpu_records(phy) == oracle_records(orecs) and len(orecs) > 0
    phy.close()


def test_device_resident_path_and_sink_callback():
    import torch
    sc = scenario("cfg2", seed=8)
    nsf = 24
    tti0, iq, _ = gen_subframes(sc, nsf)
    _, _, orecs = run_oracle(sc, tti0, iq, taps=False)
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=7)  # python sink callback, chunks of 7 -> pipelined
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    d_iq = torch.from_numpy(iq.view(np.float32)).to("cuda:0")
    phy.process_device(d_iq.data_ptr(), nsf, tti0, 0, torch.cuda.current_stream().cuda_stream)
    assert gpu_records(phy) == oracle_records(orecs) and len(orecs) > 0
    p = phy.perf()
    assert p.nof_pdus == len(orecs) and p.kernel_ms[la.KERNELS.index("k_turbo<64>")] + p.kernel_ms[la.KERNELS.index("k_turbo<128>")] > 0
    phy.close()


def test_invalid_inputs_are_rejected():
    phy = la.Phy(nof_rx_antennas=2)
    assert not phy.setCell(70, 2, 1)       # not an LTE bandwidth
    assert not phy.setCell(100, 3, 1)      # 1, 2 or 4 CRS ports
    assert not phy.setCell(100, 2, 504)
    with pytest.raises(RuntimeError):
        phy.process_host(np.zeros((1, 2, 30720), dtype=np.complex64), 0)  # no cell set
    phy.close()


def test_exhaustive_candidate_table_matches_oracle_decoder():
    """k_viterbi / k_cce_power alone: all 157 locations x all DCI sizes, not only the entries the search happens to look at"""
    for scn, n, over in (("small", 6, {}), ("cfg3", 4, {}), ("cfg1", 4, {}), ("cfg2", 4, dict(nof_ports=4, cfi=0))):
        sc = scenario(scn, seed=51, **over)
        tti0, iq, _ = gen_subframes(sc, n)
        _, per_sf, _ = run_oracle(sc, tti0, iq)
        phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=n)
        assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
        phy.setCandidatePruning(la.Phy.PRUNE_OFF)   # every slot (the default leaves out what the search is predicted not to read: next test)
        phy.process_host(iq, tti0, 0)
        bad = compare_candidate_tables(phy, per_sf, sc, tti0, 0, n)
        assert not bad, (scn, bad[:3])
        phy.close()


def test_pruned_candidate_table_and_decode_on_demand():
    """Round 6: by default the blind decoder leaves out the slots under a location whose candidate the search is predicted to accept (k_viterbi: four launches,
    8 -> 1 CCEs, the prediction from the search's stateless tests + a snapshot of the active RNTIs) and the search has a left-out slot decoded on demand when it
    comes there after all.  (1) every slot the pruned table DOES hold equals the oracle's decoder, and once the RNTIs are active a good part of the loaded
    subframes' slots is left out; (2) the record stream is the oracle's with the prediction as it is, with pruning off, and with a prediction that claims
    everything (PRUNE_TEST: every RNTI counts as active, so the search keeps running into left-out slots - the on-demand path carries it)"""
    sc = scenario("cfg3", seed=52, n_rnti=40)
    n = 48
    tti0, iq, _ = gen_subframes(sc, n)
    _, per_sf, orecs = run_oracle(sc, tti0, iq)
    want = oracle_records(orecs)
    assert len(want) > 300
    left_out = {}
    for mode, batch in ((la.Phy.PRUNE_ON, 8), (la.Phy.PRUNE_OFF, 8), (la.Phy.PRUNE_TEST, 8), (la.Phy.PRUNE_ON, 48)):
        phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=batch, pcapwriter=la.PcapWriter(None))
        assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
        phy.setCandidatePruning(mode)
        misses = 0
        for a in range(0, n, batch):
            phy.process_host(iq[a:a + batch], tti0 + a, 0)
            misses += phy.perf().nof_candidate_misses
        assert gpu_records(phy) == want, (mode, batch)
        # the table of the last chunk: computed slots against the oracle's decoder, left-out slots counted
        base = n - batch
        bad = compare_candidate_tables(phy, per_sf, sc, tti0, base, batch, skip_not_computed=True)
        assert not [b for b in bad if b[1] != "left_out"], (mode, bad[:3])
        left_out[(mode, batch)] = (len([b for b in bad if b[1] == "left_out"]) / float(batch), misses)
        phy.close()
    assert left_out[(la.Phy.PRUNE_OFF, 8)] == (0.0, 0)
    assert left_out[(la.Phy.PRUNE_ON, 8)][0] > 50 and left_out[(la.Phy.PRUNE_ON, 8)][1] <= 8, left_out   # slots per subframe left out; hardly any of them missed
    assert left_out[(la.Phy.PRUNE_TEST, 8)][1] > 20, left_out                                               # the forced prediction: the on-demand path was exercised


# ---------------------------------------------------------------------------------------------- edge cases
def test_redundancy_versions_1_2_3():
    _run("cfg2", 24, seed=31, pct_rv=70, mcs_min=4, mcs_max=16)


def test_mid_snr_many_crc_failures():
    n = _run("cfg3", 24, seed=33, snr_db=14.0)
    assert n > 0


def test_iteration_counts_equal_the_oracles_without_first_block_gating():
    """LSN_NO_CB_SKIP=1: every code block is decoded (the production engine skips blocks 1 .. C-1 of a transport block whose block 0 failed), so
    the per-block iteration counts of ALL blocks - hopeless 12-iteration blocks at 14 dB included - and their sum equal the oracle's"""
    _run("cfg3", 16, seed=34, snr_db=14.0, exact_iters=True)
    _run("cfg2", 16, seed=36, exact_iters=True)


def test_low_snr_subframes_are_skipped():
    """estimated SNR <= 6 dB: DCISearch::search returns before the blind search (DCISearch.cc:568-574) -> no records"""
    sc = scenario("small", seed=35, snr_db=2.0)
    tti0, iq, _ = gen_subframes(sc, 12)
    ow, per_sf, orecs = run_oracle(sc, tti0, iq)
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=12)
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    phy.process_host(iq, tti0, 0)
    assert not compare_taps(phy, per_sf, sc, 0, 12)
    assert gpu_records(phy) == oracle_records(orecs) == []
    assert phy.getStats().nof_subframes == 12 and phy.getStats().nof_locations == ow.stats().nof_locations == 0
    phy.close()


def test_frequency_selective_channel_and_cfo():
    _run("cfg2", 20, seed=37, delay_samples=9, cfo_hz=450.0)


@pytest.mark.parametrize("model,doppler,snr,at_least", [(1, 5.0, 22.0, 400), (2, 70.0, 24.0, 300), (3, 70.0, 30.0, 20), (3, 300.0, 26.0, 1)])
def test_multipath_fading_channels_of_ts_36_101(model, doppler, snr, at_least):
    """EPA 5 Hz / EVA 70 Hz / ETU 300 Hz (TS 36.101 Annex B.2: seven / nine taps, an independent Rayleigh process per rx antenna, CRS port and tap, a fractional
    sampling offset on top): the estimator's smoothing and interpolation, the equalisers and the decoders on frequency-selective, time-varying channels - every stage
    tap, every soft bit, every code block's iteration count and the record stream equal to the oracle's.  ETU's 5 us delay spread is beyond the reference's estimator
    settings (5-tap smoothing over pilots 90 kHz apart, SubframeWorker.cc:381-390; profiles/r06_frc_36101.txt): few records there, the taps still have to agree"""
    n = _run("cfg3", 40, seed=70 + model, snr_db=snr, n_rnti=24, chan_model=model, doppler_hz=doppler, timing_offset_samples=0.37, update_meta_period=20)
    assert n >= at_least, n


def test_multipath_fading_four_ports_and_extended_cp():
    _run("cfg2", 24, seed=75, nof_ports=4, nof_prb=50, snr_db=24.0, chan_model=2, doppler_hz=70.0)
    _run("cfg3", 24, seed=76, cp=1, nof_prb=75, snr_db=26.0, n_rnti=16, chan_model=3, doppler_hz=70.0)


def test_cfo_correction_fixed_offset_in_the_ofdm_kernel():
    """3 kHz of carrier offset (a fifth of the subcarrier spacing): uncorrected the receiver delivers next to nothing; with the NCO of k_ofdm set to the offset the
    stream decodes - grid, estimates, soft bits and records equal to the oracle fed the same correction (o_ofdm_rx's NCO)"""
    sc = scenario("cfg2", seed=81, cfo_hz=3000.0, snr_db=26.0)
    tti0, iq, _ = gen_subframes(sc, 16)
    n0 = len(run_oracle(sc, tti0, iq, taps=False)[2])
    n1, _ = _run("cfg2", 16, seed=81, cfo_hz=3000.0, snr_db=26.0, cfo_correction=(1, 3000.0, 0.25))
    assert n1 > 60 and n0 < 5, (n0, n1)


def test_cfo_tracking_follows_a_drifting_oscillator():
    """the offset starts at 900 Hz (the loop is told 600, as a cell search would be off) and drifts by 6 kHz/s; the loop - chunks of 8 subframes, four chunks of
    delay - pulls the residual the CRS estimator sees to a fraction of the offset; every chunk's correction and every tap equal to the oracle driven by the same
    loop rule"""
    n, loop = _run("cfg2", 96, seed=82, batch=8, cfo_hz=900.0, cfo_drift_hz_per_s=6000.0, snr_db=26.0, cfo_correction=(2, 600.0, 0.5))
    true_end = 900.0 + 6000.0 * 0.092   # at the middle of the last chunk
    assert abs(loop.meas[-1] - true_end) < 30.0, (loop.meas, true_end)      # the measurement (correction + residual) is on the true offset ...
    assert 0.0 < true_end - loop.hist[-1] < 300.0, (loop.hist, true_end)    # ... and the correction lags a ramp this steep (6 kHz/s) by its loop delay, no more
    assert n > 350, n    # (uncorrected: no record at all - the offset leaves the +-1 kHz range of the CRS estimator within 17 ms)


def test_one_rx_antenna_two_ports():
    """TM2 decodes on one rx antenna; two-codeword grants are gated off (DL_Sniffer_PDSCH.cc:887-889)"""
    _run("cfg3", 20, seed=39, nof_rx=1, n_rnti=20)


def test_full_band_two_codeword_256qam_max_size():
    """one UE over all 100 PRBs, 2 codewords, top MCS of the 256QAM table: 13+ code blocks of K = 6144 per TB"""
    _run("cfg3", 8, seed=41, n_rnti=1, dl_min=1, dl_max=1, ul_min=0, ul_max=0, mix_tm3_pct=100, mix_tm4_pct=0, pct_256qam=100,
         mcs_min=26, mcs_max=27, snr_db=40.0, rar_period=0, paging_period=0)


def test_cfi_1_2_random_and_phy_options():
    _run("cfg2", 20, seed=43, cfi=0)
    sc = scenario("small", seed=45)
    tti0, iq, _ = gen_subframes(sc, 24)
    for kw, okw in ((dict(skipSecondaryMetaFormats=True), dict(skip_secondary=1)), (dict(histogramThreshold=2), dict(threshold=2)),
                    (dict(max_turbo_iterations=3), dict(max_turbo_iter=3)), (dict(mcs_tracking_mode=2), dict(mcs_tracking_mode=2)),
                    (dict(mcs_tracking_mode=0), dict(mcs_tracking_mode=0))):
        _, _, orecs = run_oracle(sc, tti0, iq, update_meta_period=10, taps=False, **okw)
        phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=10, **kw)
        assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
        phy.process_host(iq, tti0, 10)
        assert gpu_records(phy) == oracle_records(orecs), kw
        phy.close()


def test_all_zero_and_noise_only_input():
    phy = la.Phy(nof_rx_antennas=2, max_batch=4)
    assert phy.setCell(25, 2, 7)
    z = np.zeros((4, 2, 15 * 512), dtype=np.complex64)
    phy.process_host(z, 0, 0)          # NaN SNR (0/0) -> not > 6 dB -> skipped, must not crash
    rng = np.random.default_rng(0)
    nz = (rng.standard_normal(z.shape) + 1j * rng.standard_normal(z.shape)).astype(np.complex64)
    phy.process_host(nz, 4, 0)
    assert phy.pdus == []
    phy.close()


# ---------------------------------------------------------------------------------------------- size-independent properties
def test_full_load_properties_round_trip_and_pipeline_invariance():
    """BASELINE configs[2] load (20 MHz, 150 RNTIs, up to 256QAM) without the oracle in the loop: (1) round trip - every downlink
    PDU the GPU path emits is byte-identical to a transport block the synthetic eNB sent to that RNTI in that TTI, and once the
    RNTIs are known most transmitted blocks are recovered; (2) the record stream does not depend on how the capture is cut
    into calls, chunks and pipelined submits (results never depend on pipeline timing); (3) a second pass over the same capture on
    a fresh Phy reproduces the stream bit for bit."""
    import torch
    sc = scenario("cfg3", seed=77)
    nsf = 240
    tti0, iq, truth = gen_subframes(sc, nsf)
    sent = {}
    for i, pdus in enumerate(truth):
        for p in pdus:
            if not p["is_ul"]:
                sent.setdefault(((tti0 + i) % 10240, p["rnti"]), []).append(p["payload"])

    def run(batch, mode):
        phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=batch, pcapwriter=la.PcapWriter(None))
        assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
        if mode == "host":
            phy.process_host(iq, tti0, 100)
        else:  # three pipelined submits of unequal length on a resident capture
            d = torch.from_numpy(iq.view(np.float32)).to("cuda:0")
            stride = iq[0].size * 2
            torch.cuda.synchronize()
            cuts = [0, 37, 150, nsf]
            for a, b in zip(cuts[:-1], cuts[1:]):
                # update_meta_period counts subframes of the stream (sf_cnt), so it carries across submits
                phy.submit_device(d.data_ptr() + a * stride * 4, b - a, tti0 + a, 100, torch.cuda.current_stream().cuda_stream)
            phy.wait()
        recs = parse_pcap(phy.pcapwriter.bytes())
        phy.close()
        return recs

    from lsn_testlib import parse_pcap
    ref = run(200, "host")
    assert [r["ctx"] + r["pdu"] for r in run(16, "host")] == [r["ctx"] + r["pdu"] for r in ref]
    assert [r["ctx"] + r["pdu"] for r in run(64, "submit")] == [r["ctx"] + r["pdu"] for r in ref]
    assert [r["ctx"] + r["pdu"] for r in run(200, "host")] == [r["ctx"] + r["pdu"] for r in ref]
    dl = [r for r in ref if r["direction"] == 1]
    assert len(dl) > 1500
    for r in dl:
        key = ((r["sfn"] * 10 + r["sf"]) % 10240, r["rnti"])
        if r["rnti_type"] != 3:  # SI / P / RA records carry substituted RNTI constants; C-RNTI records must match a sent block
            continue
        assert key in sent and r["pdu"] in sent[key], key
    late = [(k, pl) for k, v in sent.items() for pl in v if (k[0] - tti0) % 10240 >= 120 and 0x000B <= k[1] <= 0xFFF3]
    got = {(((r["sfn"] * 10 + r["sf"]) % 10240, r["rnti"]), r["pdu"]) for r in dl}
    assert sum(x in got for x in late) >= 0.85 * len(late), (sum(x in got for x in late), len(late))


# ---------------------------------------------------------------------------------------------- BASELINE.json sizes
def _run_stream(scn, nsf, seed, batch, update_meta_period=0, submit=False, pruning=None, min_records_per_sf=5.0, **over):
    """record stream + learned state of a long stream at the scenario's full RNTI count (no per-subframe taps: the tap comparisons
    above cover the stages; here the sequential state - RNTI manager, MCS tables, p-a - runs for hundreds of subframes)"""
    import torch
    sc = scenario(scn, seed=seed, **over)
    tti0, iq, _ = gen_subframes(sc, nsf)
    ow, _, orecs = run_oracle(sc, tti0, iq, update_meta_period=update_meta_period, taps=False)
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=batch, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], cp=sc.get("cp", 0))
    if pruning is not None:
        phy.setCandidatePruning(pruning)
    if submit:  # the pipelined entry point bench.py times: resident capture, several submits, one wait
        d = torch.from_numpy(iq.view(np.float32)).to("cuda:0")
        stride = iq[0].size * 8
        torch.cuda.synchronize()
        cuts = list(range(0, nsf, 3 * batch + 11)) + [nsf]
        for a, b in zip(cuts[:-1], cuts[1:]):
            phy.submit_device(d.data_ptr() + a * stride, b - a, tti0 + a, update_meta_period, torch.cuda.current_stream().cuda_stream)
        phy.wait()
    else:
        phy.process_host(iq, tti0, update_meta_period)
    g, o = gpu_records(phy), oracle_records(orecs)
    assert len(o) > min_records_per_sf * nsf
    assert g == o, "record streams differ: gpu %d vs oracle %d" % (len(g), len(o))
    st, ost = phy.getStats(), ow.stats()
    for f in ("nof_locations", "nof_decoded_locations", "nof_cce", "nof_missed_cce", "nof_subframes", "nof_subframe_collisions_dw", "nof_subframe_collisions_up"):
        assert getattr(st, f) == getattr(ost, f), f
    assert phy.nofTrackedRnti() == ow.nof_tracked()
    assert phy.nof_active_rnti() == ow.nof_active()
    phy.close()


@pytest.mark.parametrize("scn,nsf,over", [("small", 160, {}), ("small", 160, dict(nof_prb=6, cfi=3, dl_min=1, dl_max=1, n_rnti=2)), ("small", 160, dict(nof_prb=15, cfi=2, dl_min=1, dl_max=2, n_rnti=3)),
                                          ("cfg1", 120, {}), ("cfg2", 120, dict(nof_ports=4)), ("cfg3", 120, dict(cp=1, n_rnti=40)), ("cfg3", 160, dict(nof_prb=75, n_rnti=40))])
def test_candidate_pruning_claims_everything_on_other_cells(scn, nsf, over):
    """the decode-on-demand path of the candidate pruning (k_viterbi level launches, Engine::candidateMiss) under the prediction that claims everything
    (PRUNE_TEST), on cells whose location tables differ from the 20 MHz / CFI 3 one: few CCEs (levels without a location, blocks without an 8-CCE parent),
    one and four CRS ports, extended CP, 75 PRB - record stream, search statistics and learned state equal the oracle's"""
    _run_stream(scn, nsf, seed=77, batch=24, update_meta_period=50, pruning=la.Phy.PRUNE_TEST, min_records_per_sf=0.5, **over)


def test_baseline_cfg2_32_rnti_400_subframes():
    """BASELINE.json configs[1]: 20 MHz, 32 active RNTIs, TM2 64QAM"""
    _run_stream("cfg2", 400, seed=202, batch=64, update_meta_period=100)


def test_baseline_cfg3_150_rnti_400_subframes_pipelined():
    """BASELINE.json configs[2] (the metric's config): 20 MHz, 150 active RNTIs, TM3/TM4 up to 256QAM, through lsn_phy_submit_device"""
    _run_stream("cfg3", 400, seed=203, batch=50, update_meta_period=100, submit=True)


def test_tti_wrap_inside_a_call():
    """a call that crosses TTI 10239 -> 0: records carry SFN 0..1023 (PcapWriter.cc:102-103), raw pcap bytes equal to the oracle's"""
    sc = scenario("small", seed=61, start_tti=10240 - 13)
    tti0, iq, _ = gen_subframes(sc, 30)
    assert tti0 == 10240 - 13
    _, _, orecs = run_oracle(sc, tti0, iq, taps=False)
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=8, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    phy.process_host(iq, tti0, 0)
    from lsn_testlib import parse_pcap
    recs = parse_pcap(phy.pcapwriter.bytes())
    assert [r["ctx"] + r["pdu"] for r in recs] == oracle_records(orecs) and len(orecs) > 0
    assert all(r["sfn"] < 1024 for r in recs) and any(r["sfn"] == 1023 for r in recs) and any(r["sfn"] == 0 for r in recs)
    phy.close()


def test_shortcut_discovery_off_and_histogram_threshold_setter():
    """PhyCommon::setShortcutDiscovery(false) (LTESniffer_Core.cc:87,616) and RNTIManager::setHistogramThreshold (:620) after construction"""
    sc = scenario("cfg3", seed=63, n_rnti=40)
    tti0, iq, _ = gen_subframes(sc, 60)
    for setup, okw in ((lambda p: p.setShortcutDiscovery(False), dict(enable_shortcut=0)), (lambda p: p.setHistogramThreshold(3), dict(threshold=3))):
        ow, _, orecs = run_oracle(sc, tti0, iq, taps=False, **okw)
        phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=20, pcapwriter=la.PcapWriter(None))
        assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
        setup(phy)
        phy.process_host(iq, tti0, 0)
        assert gpu_records(phy) == oracle_records(orecs) and len(orecs) > 0
        st, ost = phy.getStats(), ow.stats()
        assert st.nof_decoded_locations == ost.nof_decoded_locations and st.nof_locations == ost.nof_locations
        txt = phy.printStats().splitlines()
        assert txt[0].startswith("nof_decoded_locations, nof_cce") and txt[1].split(", ")[0] == str(ost.nof_decoded_locations)
        phy.close()
    assert la.Phy(nof_rx_antennas=2).getShortcutDiscovery()


def test_mcs_database_ageing_matches_oracle():
    """MCSTracking::update_database_dl every interval x 1000 subframes (LTESniffer_Core.cc:473-499): RNTIs that appeared through a
    RAR and went idle leave the database, the record stream and the database size stay identical to the oracle's"""
    sc = scenario("small", seed=65, n_rnti=6, rar_period=40, pct_256qam=50, mix_tm3_pct=40, dl_min=1, dl_max=2)
    nsf = 3300
    tti0, iq, _ = gen_subframes(sc, nsf)
    ow, _, orecs = run_oracle(sc, tti0, iq, taps=False, mcs_update_interval=1)
    never, _, _ = run_oracle(sc, tti0, iq[:1], taps=False)
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=100, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    phy.setMcsUpdateInterval(1)
    phy.process_host(iq, tti0, 0)
    assert gpu_records(phy) == oracle_records(orecs) and len(orecs) > nsf
    assert phy.nofTrackedRnti() == ow.nof_tracked()
    # the test must exercise deletions: more RNTIs were introduced (one per RAR, 82 + the 6 of the cell) than the database holds at the
    # end (the ones that went idle more than one whole second before the last update are gone)
    assert ow.nof_tracked() < nsf // 40 - 10, ow.nof_tracked()
    phy.close()


def test_a_failing_chunk_is_reported_and_does_not_wedge_the_pipeline():
    """round-2 advisor finding: an error inside the pipeline left the turn counters inconsistent and the next wait() hung.  Now a failed chunk
    travels on with its error text: the call returns an error, every chunk before and after it is still written, and the engine keeps working"""
    import os
    import torch
    sc = scenario("small", seed=77)
    nsf = 40
    tti0, iq, _ = gen_subframes(sc, nsf)
    _, _, orecs = run_oracle(sc, tti0, iq, taps=False)
    os.environ["LSN_INJECT_STAGE_A_ERROR"] = "2"   # the third of five chunks of the first block (read once, when the engine is made)
    try:
        phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=8, pcapwriter=la.PcapWriter(None))
    finally:
        del os.environ["LSN_INJECT_STAGE_A_ERROR"]
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    d_iq = torch.from_numpy(iq.view(np.float32)).to("cuda:0")
    with pytest.raises(RuntimeError):
        phy.process_device(d_iq.data_ptr(), nsf, tti0, 0, torch.cuda.current_stream().cuda_stream)
    got = gpu_records(phy)
    # subframes 16..23 are missing, everything in front of them is what the oracle wrote
    head = [r for r in oracle_records(orecs)]
    n_head = len([r for r in orecs if ((r["sfn"] * 10 + r["sf"]) - tti0) % 10240 < 16])
    assert got[:n_head] == head[:n_head] and len(got) > n_head
    ttis = {(((g[10] << 8) | g[11]) >> 4) * 10 + (((g[10] << 8) | g[11]) & 15) for g in got}
    assert not any((t - tti0) % 10240 in range(16, 24) for t in ttis) and any((t - tti0) % 10240 >= 24 for t in ttis)
    # the same engine decodes the next call completely (state differs from the oracle's by the lost chunk, so only completion and volume are checked)
    phy.pcapwriter.reset()
    phy.process_device(d_iq.data_ptr(), nsf, (tti0 + nsf) % 10240, 0, torch.cuda.current_stream().cuda_stream)
    assert len(gpu_records(phy)) >= 0.8 * len(orecs)
    phy.close()


def test_harq_soft_combining_matches_oracle():
    """harq_mode = 1 (HARQ.cc:71-190, DL_Sniffer_PDSCH.cc:943-1020): retransmissions 8 subframes after a failed transport block are
    combined with the soft buffer of their HARQ process on the GPU (k_harq_combine), blocks decoded before are not decoded again - record stream
    identical to the oracle's, which differs from the stream without HARQ"""
    sc = scenario("small", seed=93, n_rnti=3, dl_min=2, dl_max=2, ul_min=0, ul_max=0, mcs_min=18, mcs_max=22, snr_db=11.0, pct_harq=60)
    tti0, iq, _ = gen_subframes(sc, 90)
    ow, _, orecs = run_oracle(sc, tti0, iq, taps=False, harq_mode=1, mcs_tracking_mode=0)
    _, _, orecs_off = run_oracle(sc, tti0, iq, taps=False, mcs_tracking_mode=0)
    st = ow.harq_stats()
    assert st[1] > 5 and st[3] > 5 and oracle_records(orecs) != oracle_records(orecs_off)
    for batch in (90, 16):  # one chunk; several chunks (a retransmission whose first transmission sits in the previous chunk)
        phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=batch, pcapwriter=la.PcapWriter(None), harq_mode=1, mcs_tracking_mode=0)
        assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
        phy.process_host(iq, tti0, 0)
        assert gpu_records(phy) == oracle_records(orecs), batch
        phy.close()
    # the same stream with 2 code words per grant (TM3) on 20 MHz: big transport blocks of several code blocks each
    sc = scenario("cfg3", seed=95, n_rnti=4, dl_min=2, dl_max=2, ul_min=0, ul_max=0, mix_tm3_pct=100, mix_tm4_pct=0, pct_256qam=0, mcs_min=20, mcs_max=24, snr_db=13.0,
                  rar_period=0, paging_period=0, pct_harq=60)
    tti0, iq, _ = gen_subframes(sc, 40)
    ow, _, orecs = run_oracle(sc, tti0, iq, taps=False, harq_mode=1, mcs_tracking_mode=0)
    assert ow.harq_stats()[1] > 3
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=40, pcapwriter=la.PcapWriter(None), harq_mode=1, mcs_tracking_mode=0)
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    phy.process_host(iq, tti0, 0)
    assert gpu_records(phy) == oracle_records(orecs) and len(orecs) > 0
    phy.close()


def test_harq_retransmissions_are_decoded_in_batches_and_on_several_engines():
    """Round 6: the retransmissions of a chunk are combined and decoded in batches AHEAD of the sequential commit walk (Engine::harqScout /
    harqRunBatch; the walk takes a result only under the key of exactly its inputs, else decodes alone as rounds 4-5 did) - same records as the
    oracle, nearly every combined decode served from a batch, also for chains of several retransmissions of one process inside a chunk; and with
    the capture spread over two engines (lsn_phy_create_multi: database, buffers and pool shared, one commit turn at a time) the same again"""
    sc = scenario("cfg3", seed=97, n_rnti=6, dl_min=3, dl_max=4, ul_min=0, ul_max=0, mix_tm3_pct=50, mix_tm4_pct=0, pct_256qam=0, mcs_min=19, mcs_max=25, snr_db=12.0,
                  rar_period=0, paging_period=0, pct_harq=70)
    nsf = 120
    tti0, iq, _ = gen_subframes(sc, nsf)
    ow, _, orecs = run_oracle(sc, tti0, iq, taps=False, harq_mode=1, mcs_tracking_mode=0)
    st = ow.harq_stats()
    assert st[1] > 40 and len(orecs) > 100
    for devices, batch in ((None, 120), (None, 40), ([0, 0], 24), ([0, 0, 0], 16)):
        import torch
        phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=batch, pcapwriter=la.PcapWriter(None), harq_mode=1, mcs_tracking_mode=0, devices=devices)
        assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
        d = torch.from_numpy(iq.view(np.float32)).to("cuda:0")
        torch.cuda.synchronize()
        stride = iq[0].size * 8
        for a in range(0, nsf, batch):   # (one submit per chunk: with several engines chunk g goes to engine g mod G)
            phy.submit_device(d.data_ptr() + a * stride, min(batch, nsf - a), tti0 + a, 0, torch.cuda.current_stream().cuda_stream)
        phy.wait()
        assert gpu_records(phy) == oracle_records(orecs), (devices, batch)
        hc = list(phy.perf().nof_harq_combines)
        assert hc[0] >= 1 and hc[1] > 0 and hc[2] <= hc[1] // 10, (devices, batch, hc)   # batches were run; the walk found (nearly) all of its combined decodes there
        phy.close()
