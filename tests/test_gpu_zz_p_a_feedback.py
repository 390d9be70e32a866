"""GPU parity of the p-a feedback loop (RRCConnectionSetup -> UE configuration database -> PDSCH power offset of later decodes).
Sorted last on purpose: written at the end of round 1 after the GPU budget was used up, first run on a GPU is the driver's."""
import numpy as np
import pytest

import ltesniffer_amd as la
from lsn_testlib import scenario
from parity import gen_subframes, gpu_records, oracle_records, run_oracle
from test_gpu_parity import _run

pytestmark = pytest.mark.gpu


def test_connection_setup_p_a_feedback():
    """UEs receive an RRCConnectionSetup with p-a != 0 dB and are then sent with that power offset: the record stream (and with it every
    decode decision behind it) equals the oracle's whether the decodes were planned before the connection setup was committed
    (one big chunk: planned with the old p-a, dropped and decoded again at commit) or chunk by chunk"""
    kw = dict(nof_prb=25, n_rnti=3, dl_min=3, dl_max=3, ul_min=0, ul_max=0, mcs_min=20, mcs_max=28, msg4_period=6, snr_db=36.0)
    assert _run("cfg2", 40, seed=3, batch=40, msg4_p_a_idx=0, **kw) > 60
    _run("cfg2", 40, seed=3, batch=4, msg4_p_a_idx=7, **kw)
    _run("cfg3", 30, seed=4, batch=10, nof_prb=50, n_rnti=12, dl_min=4, dl_max=6, msg4_period=4, msg4_p_a_idx=8, update_meta_period=10)


def test_learned_ue_configuration_matches_oracle():
    sc = scenario("cfg2", seed=9, nof_prb=25, n_rnti=4, dl_min=3, dl_max=3, ul_min=0, ul_max=0, mcs_min=10, mcs_max=20, msg4_period=5, msg4_p_a_idx=8, snr_db=34.0)
    tti0, iq, truth = gen_subframes(sc, 40)
    ow, _, orecs = run_oracle(sc, tti0, iq, taps=False)
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=8)
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    for base in range(0, 40, 8):
        phy.process_host(iq[base:base + 8], tti0 + base, 0)
    assert gpu_records(phy) == oracle_records(orecs)
    rntis = {p["rnti"] for pdus in truth for p in pdus} | {0x0BAD, 0xFFFF, 5}
    learned = 0
    for rnti in sorted(rntis):
        g, o = phy.ue_config(rnti), ow.ue_cfg(rnti)
        assert (g.has_ue_config, np.float32(g.p_a_db).item(), g.i_offset_ack, g.i_offset_cqi, g.i_offset_ri, g.cqi_type) == o, hex(rnti)
        learned += g.has_ue_config
    assert learned >= 3
    phy.close()
