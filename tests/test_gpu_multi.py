"""One capture over several engines (lsn_phy_create_multi): chunk g of the stream goes to engine g mod G, every engine runs
stage A and the decodes of its chunks on its own device, and the sequential host state (FALCON search, RNTI manager, MCS tracking, record
order) is shared in turns.  On a one-GPU box the engines are given the same device (two engines, two sets of streams and buffers, one GPU);
with LSN_FORCE_PEER_COPY=1 the blocks also take the peer-copy staging path that carries them to another GPU.  The record stream must be the
single-engine stream, byte for byte, and equal to the CPU oracle's."""
import os
import subprocess
import sys

import numpy as np
import pytest

import ltesniffer_amd as la
from lsn_testlib import scenario
from parity import gen_subframes, gpu_records, oracle_records, run_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stream(devices, sc, tti0, iq, batch, cuts):
    import torch
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=batch, pcapwriter=la.PcapWriter(None), devices=devices)
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    d = torch.from_numpy(iq.view(np.float32)).to("cuda:0")
    stride = iq[0].size * 8
    torch.cuda.synchronize()
    for a, b in zip(cuts[:-1], cuts[1:]):
        phy.submit_device(d.data_ptr() + a * stride, b - a, tti0 + a, 100, torch.cuda.current_stream().cuda_stream)
    phy.wait()
    recs = gpu_records(phy)
    st = phy.getStats()
    stats = tuple(getattr(st, f) for f in ("nof_locations", "nof_decoded_locations", "nof_cce", "nof_missed_cce", "nof_subframes", "nof_subframe_collisions_dw", "nof_subframe_collisions_up"))
    tracked = phy.nofTrackedRnti()
    phy.close()
    return recs, stats, tracked


def _device_sets():
    """two and three engines: on the GPUs that are there (a multi-GPU node runs them on DIFFERENT devices - peer copies of the blocks, function
    attributes per device, one HIP context each), all on device 0 on a one-GPU box"""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        return [[0, 0], [0, 0, 0]]
    return [[0, 1], list(range(min(n, 3))) if n >= 3 else [0, 1, 0]]


def test_two_and_three_engines_reproduce_the_single_engine_stream():
    sc = scenario("cfg3", seed=88, n_rnti=60)
    nsf = 200
    tti0, iq, _ = gen_subframes(sc, nsf)
    ow, _, orecs = run_oracle(sc, tti0, iq, update_meta_period=100, taps=False)
    cuts = [0, 37, 150, nsf]
    one = _stream(None, sc, tti0, iq, 16, cuts)
    assert one[0] == oracle_records(orecs) and len(orecs) > 1000
    for devs in _device_sets():
        got = _stream(devs, sc, tti0, iq, 16, cuts)
        assert got[0] == one[0], devs
        assert got[1] == one[1] and got[2] == one[2] == ow.nof_tracked(), devs


def test_peer_copy_staging_path():
    """a fresh process with LSN_FORCE_PEER_COPY=1: every block of the second engine goes through hipMemcpyPeerAsync into its staging ring"""
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import numpy as np, hashlib\n"
            "from lsn_testlib import scenario\n"
            "from parity import gen_subframes\n"
            "from test_gpu_multi import _stream\n"
            "sc = scenario('cfg2', seed=89)\n"
            "tti0, iq, _ = gen_subframes(sc, 120)\n"
            "from test_gpu_multi import _device_sets\n"
            "a = _stream(None, sc, tti0, iq, 10, [0, 55, 120]); b = _stream(_device_sets()[0], sc, tti0, iq, 10, [0, 55, 120])\n"
            "assert a == b and len(a[0]) > 300, (len(a[0]), len(b[0]))\n"
            "print('OK', len(a[0]))\n") % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, LSN_FORCE_PEER_COPY="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_ul_mode_over_two_engines_equals_one_engine_and_the_oracle():
    """UL_MODE on a capture that is spread over engines (round 2 refused it): the ULSchedule databases, the uplink tracking database and the
    uplink configuration are part of the shared sequential state now.  The stream starts WITHOUT a configuration: the SIB2 is decoded by
    whichever engine commits that chunk and the DMRS / hopping / PRACH tables of the other engine follow at its next commit turn; a PUSCH
    whose DCI 0 sits in the previous chunk (n - 4, other engine) is decoded on the engine that holds subframe n."""
    import torch
    from lsn_testlib import REAL_SIB1, OracleWorkerUl, encode_sib2, gen_ul_mode_subframes, parse_pcap
    sc = scenario("cfg2", seed=91, nof_rx=1, n_rnti=10, dl_min=2, dl_max=3, ul_min=2, ul_max=4, nof_prb=25, mcs_max=18, pusch_hop_offset=4, pct_hop=20)
    sib2 = encode_sib2(cyclic_shift=3, group_assignment_pusch=5, pusch_hop_offset=4, group_hopping_enabled=1, root_seq_idx=22, prach_config_idx=3, zero_corr_zone=1,
                       prach_freq_offset=2)
    nsf = 90
    tti0, iq, sent = gen_ul_mode_subframes(sc, nsf, si_msgs=[REAL_SIB1, sib2], group_hopping=1)
    ow = OracleWorkerUl(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], None, None)
    for i in range(nsf):
        ow.work_ul(iq[i, 0], iq[i, 1], tti0 + i, update_meta=1 if i % 25 == 0 else 0)
    orecs = oracle_records(parse_pcap(ow.pcap_bytes()))
    assert len([r for r in parse_pcap(ow.pcap_bytes()) if r["direction"] == 0]) >= 10
    d = torch.from_numpy(iq.view(np.float32)).to("cuda:0")
    got = {}
    for devs in [None] + _device_sets():
        phy = la.Phy(nof_rx_antennas=2, sniffer_mode=1, max_batch=8, pcapwriter=la.PcapWriter(None), devices=devs)
        assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"]) and phy.getUlConfig() is None
        phy.process_device(d.data_ptr(), nsf, tti0, 25, torch.cuda.current_stream().cuda_stream)
        got[str(devs)] = (gpu_records(phy), phy.getUlConfig())
        phy.close()
    assert got["None"][0] == orecs and got["None"][1] == ow.ul_config()
    for devs in _device_sets():
        assert got[str(devs)] == got["None"], devs
