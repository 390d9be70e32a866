"""GPU parity at 15 MHz (75 PRB): the one LTE bandwidth whose symbol length (1536 = 3 x 512) is not a power of two - the
radix-3 combination in k_ofdm / k_ul_fft, and everything behind it, bit-exact against the oracle."""
import numpy as np
import pytest

import ltesniffer_amd as la
from lsn_testlib import scenario, sync_capture, TxGen
from test_gpu_parity import _run as run_dl
from test_gpu_ul import _run as run_ul
from test_pbch_oracle import oracle_mib
from test_sync_oracle import oracle_cell_search
from test_gpu_sync import same

pytestmark = pytest.mark.gpu


def test_downlink_75prb_matches_oracle():
    assert run_dl("cfg2", 16, seed=6, batch=8, nof_prb=75, cell_id=77, n_rnti=20, cfo_hz=250.0) > 20
    run_dl("cfg3", 12, seed=7, batch=12, nof_prb=75, cell_id=4, n_rnti=30, dl_min=3, dl_max=5)


def test_uplink_75prb_matches_oracle():
    ok, n = run_ul(75, 33, 4, seed=5)
    assert ok >= n * 0.5


def test_cell_search_and_mib_75prb():
    sc = scenario("cfg2", seed=9, start_tti=10 * 300 + 2, nof_prb=75, cell_id=401, n_rnti=8, dl_min=2, dl_max=3, cfo_hz=-600.0)
    x, _ = sync_capture(sc, 7777, 1)
    ro, so, ocorr = oracle_cell_search(x, 75, 1, -1, 20.0)
    rg, sg, gcorr = la.cell_search(x, 75, nof_periods=1, with_corr=True)
    assert np.array_equal(gcorr.view(np.uint32), ocorr.view(np.uint32))
    assert rg == ro == 1 and sg.cell_id == 401
    same(sg, so)
    tx = TxGen(**sc)
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=4)
    assert phy.setCell(75, sc["nof_ports"], sc["cell_id"])
    found = 0
    for _ in range(20):
        tti, iq, _ = tx.next()
        r, m = oracle_mib(sc, iq)
        g = phy.mib_decode(iq)
        assert g["found"] == r
        if r:
            assert (g["sfn"], g["nof_prb"], g["nof_ports"]) == (m.sfn, 75, sc["nof_ports"])
            found += 1
    assert found == 2
    phy.close()
