"""GPU parity tests proper: HIP path (through the C ABI) vs. the CPU oracle on the same seeded inputs - bit-exact."""
import numpy as np
import pytest

import ltesniffer_amd as la
from lsn_testlib import scenario
from parity import compare_taps, gen_subframes, gpu_records, oracle_records, run_oracle

pytestmark = pytest.mark.gpu


def _run(scn, nsf, seed=1, batch=16, update_meta_period=0, **over):
    sc = scenario(scn, seed=seed, **over)
    tti0, iq, truth = gen_subframes(sc, nsf)
    ow, per_sf, orecs = run_oracle(sc, tti0, iq, update_meta_period=update_meta_period)
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=batch)
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    bad = []
    for base in range(0, nsf, batch):
        n = min(batch, nsf - base)
        # update_meta_period counts subframes from the start of the stream, same as the oracle driver above
        phy.process_host(iq[base:base + n], tti0 + base, update_meta_period)
        bad += [(base,) + b for b in compare_taps(phy, per_sf, sc, base, n)]
    assert not bad, bad[:5]
    g, o = gpu_records(phy), oracle_records(orecs)
    assert len(o) > 0
    assert g == o, "record streams differ: gpu %d vs oracle %d" % (len(g), len(o))
    st, ost = phy.getStats(), ow.stats()
    for f in ("nof_locations", "nof_decoded_locations", "nof_cce", "nof_missed_cce", "nof_subframes",
              "nof_subframe_collisions_dw", "nof_subframe_collisions_up"):
        assert getattr(st, f) == getattr(ost, f), f
    phy.close()
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
    assert not phy.setCell(75, 2, 1)       # 1536-point FFT not supported
    assert not phy.setCell(100, 4, 1)      # 4 CRS ports not supported
    assert not phy.setCell(100, 2, 504)
    with pytest.raises(RuntimeError):
        phy.process_host(np.zeros((1, 2, 30720), dtype=np.complex64), 0)  # no cell set
    phy.close()
