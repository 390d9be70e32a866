"""GPU parity of the p-a feedback loop (RRCConnectionSetup -> UE configuration database -> PDSCH power offset of later decodes).
Sorted last on purpose: written at the end of round 1 after the GPU budget was used up, first run on a GPU is the driver's."""
import pytest

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
