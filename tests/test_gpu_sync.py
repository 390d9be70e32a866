"""GPU parity of the PSS / SSS cell search (k_pss_corr + k_sync_fin behind lsn_cell_search) against the oracle - bit-exact
correlation powers and metrics, identical decisions - and the chain a recording needs: search -> MIB -> file replay at the
offset, cell id and SFN that were found."""
import numpy as np
import pytest

import ltesniffer_amd as la
from lsn_testlib import OracleWorker, parse_pcap, scenario, sync_capture
from parity import gen_subframes, gpu_records, oracle_records
from test_file_source import write_capture
from test_sync_oracle import oracle_cell_search

pytestmark = pytest.mark.gpu

FIELDS = ("found", "cell_id", "n_id_2", "n_id_1", "sf_idx", "pss_pos", "sf_start", "cp")
FLOATS = ("pss_peak", "pss_p2avg", "sss_metric", "sss_second", "cfo_hz", "cfo_coarse_hz")


def same(g, o):
    assert [getattr(g, f) for f in FIELDS] == [getattr(o, f) for f in FIELDS], ([getattr(g, f) for f in FIELDS], [getattr(o, f) for f in FIELDS])
    a = np.array([getattr(g, f) for f in FLOATS], dtype=np.float32)
    b = np.array([getattr(o, f) for f in FLOATS], dtype=np.float32)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (a, b)


@pytest.mark.parametrize("scn,over,lead,cfo,periods,force", [
    ("small", dict(cell_id=301), 1234, 0.0, 1, -1),
    ("small", dict(cell_id=2, nof_ports=1, nof_rx=1, snr_db=5.0), 38000, 900.0, 2, -1),
    ("small", dict(cell_id=503, nof_prb=6), 77, -2500.0, 3, -1),
    ("small", dict(cell_id=100, nof_prb=15), 5000, 400.0, 1, 1),
    ("cfg1", dict(cell_id=150), 60001, 0.0, 2, 0),
    ("cfg3", dict(cell_id=37, dl_min=2, dl_max=3), 20000, -700.0, 1, 1),
    ("small", dict(cell_id=301, cp=1), 1234, 0.0, 1, -1),                      # extended cyclic prefix: the SSS sits N + N / 4 in front of the PSS
    ("small", dict(cell_id=44, cp=1, nof_prb=50, nof_ports=1, nof_rx=1), 30001, 600.0, 2, -1),
    ("small", dict(cell_id=503, cp=1, nof_prb=6), 77, -2500.0, 3, 2),
])
def test_cell_search_matches_oracle(scn, over, lead, cfo, periods, force):
    sc = scenario(scn, seed=5, start_tti=10 * 77 + 3, cfo_hz=cfo, **over)
    x, _ = sync_capture(sc, lead, periods)
    ro, so, ocorr = oracle_cell_search(x, sc["nof_prb"], periods, force, 20.0)
    rg, sg, gcorr = la.cell_search(x, sc["nof_prb"], nof_periods=periods, force_n_id_2=force, threshold=20.0, with_corr=True)
    assert np.array_equal(gcorr.view(np.uint32), ocorr.view(np.uint32)), float(np.abs(gcorr - ocorr).max())
    assert rg == ro == 1
    same(sg, so)
    assert sg.cell_id == sc["cell_id"] and sg.cp == sc.get("cp", 0)


def test_cell_search_on_noise_device_input_and_invalid_arguments():
    import torch
    rng = np.random.default_rng(3)
    n = 3 * 75 * 128 + 128
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    ro, so, _ = oracle_cell_search(x, 6, 2, -1, 20.0)
    rg, sg = la.cell_search(x, 6, nof_periods=2)
    assert rg == ro == 0 and not sg.found
    same(sg, so)
    xd = torch.from_numpy(x.view(np.float32)).to("cuda:0")  # samples already in HBM
    rg2, sg2 = la.cell_search(xd.view(torch.complex64), 6, nof_periods=2)
    assert rg2 == 0
    same(sg2, so)
    for bad in (dict(nof_prb=70), dict(nof_periods=17), dict(force_n_id_2=3), dict(nof_periods=3)):  # the last one: buffer too short
        kw = dict(nof_prb=6, nof_periods=2)
        kw.update(bad)
        with pytest.raises(RuntimeError):
            la.cell_search(x, kw.pop("nof_prb"), **kw)


def test_recording_with_unknown_offset_cell_and_sfn_is_replayed_like_the_oracle(tmp_path):
    sc = scenario("small", seed=8, start_tti=10 * 700 + 4, cell_id=215)
    tti0, iq, _ = gen_subframes(sc, 50)
    lead = 4321
    p = str(tmp_path / "cap.cf32")
    write_capture(p, iq, lead=lead)
    raw = np.fromfile(p, dtype=np.complex64).reshape(-1, sc["nof_rx"])
    rc, s = la.cell_search(raw[:, 0], sc["nof_prb"], nof_periods=2)
    assert rc == 1 and s.cell_id == sc["cell_id"]
    sflen = iq.shape[2]
    # the first subframe-0/5 boundary of the recording: subframe 5 of SFN 700 is the second whole subframe
    assert s.sf_start == lead + sflen and s.sf_idx == 5
    # replay from the first subframe 0 (MIB): 5 more subframes on
    off = s.sf_start + (5 * sflen if s.sf_idx == 5 else 0)
    first = (off - lead) // sflen
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], s.cell_id, sc["nof_rx"])
    for i in range(first, 50):
        ow.work(iq[i], tti0 + i, update_meta=1 if (i - first) % 20 == 0 else 0)
    orecs = oracle_records(parse_pcap(ow.pcap_bytes()))
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=8, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], s.cell_id)
    assert phy.process_file(p, start_tti=la.TTI_FROM_MIB, offset_time=off, update_meta_period=20) == 50 - first
    assert gpu_records(phy) == orecs and orecs
    phy.close()


def test_extended_cp_recording_search_mib_and_replay(tmp_path):
    """the chain of a recording from an extended-CP cell: lsn_cell_search reports cp = 1 and the subframe boundary, lsn_phy_set_cell takes the CP, the PBCH (216
    symbols, 4 x 432 bits per 40 ms) gives the SFN, the replay writes what the oracle's worker writes"""
    sc = scenario("small", seed=18, start_tti=10 * 300 + 4, cell_id=101, cp=1)
    tti0, iq, _ = gen_subframes(sc, 50)
    lead = 2222
    p = str(tmp_path / "cap.cf32")
    write_capture(p, iq, lead=lead)
    raw = np.fromfile(p, dtype=np.complex64).reshape(-1, sc["nof_rx"])
    rc, s = la.cell_search(raw[:, 0], sc["nof_prb"], nof_periods=2)
    sflen = iq.shape[2]
    assert rc == 1 and s.cell_id == sc["cell_id"] and s.cp == 1 and s.sf_start == lead + sflen and s.sf_idx == 5
    off = s.sf_start + 5 * sflen
    first = (off - lead) // sflen
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], s.cell_id, sc["nof_rx"], cp=1)
    for i in range(first, 50):
        ow.work(iq[i], tti0 + i, update_meta=1 if (i - first) % 20 == 0 else 0)
    orecs = oracle_records(parse_pcap(ow.pcap_bytes()))
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=8, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], s.cell_id, cp=s.cp)
    assert phy.process_file(p, start_tti=la.TTI_FROM_MIB, offset_time=off, update_meta_period=20) == 50 - first
    assert gpu_records(phy) == orecs and orecs
    phy.close()


def test_recording_with_a_drifting_oscillator_is_tracked_through_the_replay(tmp_path):
    """what ue_sync's CFO tracking does for the live reference (LTESniffer_Core.cc:312-316,344): the cell search's offset starts the loop (search_cell_cfo ->
    ue_sync.cfo_current_value), the CRS estimate of every chunk keeps it on the carrier while the oscillator drifts from 1.4 kHz by 1.5 kHz/s through a replay -
    records equal to the oracle driven by the same loop rule; without correction the offset is beyond the CRS estimator's +-1 kHz and nothing decodes"""
    from parity import CfoLoop
    sc = scenario("small", seed=28, start_tti=10 * 500 + 4, cell_id=77, cfo_hz=1400.0, cfo_drift_hz_per_s=1500.0)
    nsf, batch = 126, 8
    tti0, iq, _ = gen_subframes(sc, nsf)
    p = str(tmp_path / "cap.cf32")
    write_capture(p, iq, lead=0)
    raw = np.fromfile(p, dtype=np.complex64).reshape(-1, sc["nof_rx"])
    rc, s = la.cell_search(raw[:, 0], sc["nof_prb"], nof_periods=2)
    assert rc == 1 and s.cell_id == 77 and abs(s.cfo_hz - 1400.0) < 150.0, (rc, s.cell_id, s.cfo_hz)
    sflen = iq.shape[2]
    off = s.sf_start + (5 * sflen if s.sf_idx == 5 else 0)
    first = off // sflen
    loop = CfoLoop(2, s.cfo_hz, 0.5, batch)
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], s.cell_id, sc["nof_rx"])
    n = nsf - first
    for i in range(n):
        ow.work(iq[first + i], tti0 + first + i, update_meta=1 if i % 24 == 0 else 0, cfo_hz=loop.cfo(i))
        loop.seen(i, n, ow.chest().cfo_hz)
    orecs = oracle_records(parse_pcap(ow.pcap_bytes()))
    ow0 = OracleWorker(sc["nof_prb"], sc["nof_ports"], s.cell_id, sc["nof_rx"])
    for i in range(n):
        ow0.work(iq[first + i], tti0 + first + i, update_meta=1 if i % 24 == 0 else 0)
    assert len(parse_pcap(ow0.pcap_bytes())) < len(orecs) // 4 and len(orecs) > 100, (len(parse_pcap(ow0.pcap_bytes())), len(orecs))
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=batch, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], s.cell_id)
    phy.setCfoCorrection(la.Phy.CFO_TRACK, s.cfo_hz, 0.5)
    assert phy.process_file(p, start_tti=la.TTI_FROM_MIB, offset_time=off, update_meta_period=24) == n
    assert gpu_records(phy) == orecs
    assert np.float32(phy.getCfoCorrection()) == np.float32(loop.hist[-1])
    assert abs(loop.meas[-1] - (1400.0 + 1500.0 * (nsf - batch / 2) * 1e-3)) < 40.0, loop.meas
    phy.close()
