"""Every LSN_* configuration switch the product still reads (README "Runtime knobs") is a configuration of a bit-exact contract: one GPU parity
run per switch (round-5 review, weak 9 / next 10: "for each switch that stays, one GPU parity test with it set").  The switches that change WHICH
kernels or results exist have their own tests (LSN_NO_CB_SKIP, LSN_INJECT_STAGE_A_ERROR: test_gpu_parity.py; LSN_FORCE_PEER_COPY: test_gpu_multi.py;
LSN_FILE_BLOCK / LSN_FILE_MMAP: test_file_source.py); here: threads, pinning, pre-sizing, timing events, trace, file-source geometry, pool linger.
All of them are read when an engine is made or a call starts, so a test sets them around its own Phy."""
import os

import numpy as np
import pytest

import ltesniffer_amd as la
from lsn_testlib import scenario
from parity import gen_subframes, gpu_records, oracle_records, run_oracle

pytestmark = pytest.mark.gpu

_CACHE = {}


def _stream():
    """one stream for all switches: 72 subframes of the 25-PRB cell, several chunks of 16, meta formats updated every 20 subframes"""
    if not _CACHE:
        sc = scenario("small", seed=77)
        tti0, iq, _ = gen_subframes(sc, 72)
        _, _, orecs = run_oracle(sc, tti0, iq, update_meta_period=20, taps=False)
        _CACHE.update(sc=sc, tti0=tti0, iq=iq, orecs=oracle_records(orecs))
        assert len(_CACHE["orecs"]) > 72
    return _CACHE["sc"], _CACHE["tti0"], _CACHE["iq"], _CACHE["orecs"]


@pytest.mark.parametrize("env", [{"LSN_DECODE_THREADS": "1", "LSN_QUIET": "1"}, {"LSN_DECODE_THREADS": "3"}, {"LSN_NO_PIN": "1"}, {"LSN_NO_PRESIZE": "1"},
                                 {"LSN_KERNEL_TIMING_PERIOD": "0"}, {"LSN_KERNEL_TIMING_PERIOD": "1"}, {"LSN_TRACE": "trace"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_engine_switches_leave_the_record_stream_alone(tmp_path, monkeypatch, env):
    sc, tti0, iq, orecs = _stream()
    for k, v in env.items():
        monkeypatch.setenv(k, str(tmp_path / v) if k == "LSN_TRACE" else v)
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=16, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    phy.process_host(iq, tti0, update_meta_period=20)
    assert gpu_records(phy) == orecs
    p = phy.perf()
    timed = sum(p.kernel_ms[i] for i in range(len(la.KERNELS)))
    if env.get("LSN_KERNEL_TIMING_PERIOD") == "0":
        assert timed == 0.0          # no timing events at all
    elif "LSN_KERNEL_TIMING_PERIOD" in env:
        assert p.kernel_ms[la.KERNELS.index("k_ofdm")] > 0 and p.kernel_ms[la.KERNELS.index("k_viterbi")] > 0   # every launch timed
    phy.close()
    if "LSN_TRACE" in env:
        assert os.path.getsize(str(tmp_path / "trace")) > 0   # the per-chunk timeline was written


@pytest.mark.parametrize("readers,slots", [(1, 3), (5, 8)])
def test_file_source_geometry_switches(tmp_path, monkeypatch, readers, slots):
    sc, tti0, iq, orecs = _stream()
    path = str(tmp_path / "cap.cf32")
    np.ascontiguousarray(np.transpose(iq, (0, 2, 1))).tofile(path)
    monkeypatch.setenv("LSN_FILE_BLOCK", "9")
    monkeypatch.setenv("LSN_FILE_READERS", str(readers))
    monkeypatch.setenv("LSN_FILE_SLOTS", str(slots))
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=16, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    assert phy.process_file(path, start_tti=tti0, update_meta_period=20) == iq.shape[0]
    assert gpu_records(phy) == orecs
    phy.close()


@pytest.mark.parametrize("linger_us", [0, 30000])
def test_worker_pool_linger_switch(monkeypatch, linger_us):
    """LSN_POOL_LINGER_US: how long the dispatcher waits for a fuller batch in the lossless (blocking getAvail) mode - batching changes, records do not"""
    sc, tti0, iq, orecs = _stream()
    monkeypatch.setenv("LSN_POOL_LINGER_US", str(linger_us))
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], nof_workers=12, max_batch=8, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    for i in range(iq.shape[0]):
        w = phy.getAvail()
        bufs = w.getBuffers()
        for rx in range(sc["nof_rx"]):
            bufs[rx][:iq.shape[2]] = iq[i, rx]
        tti = tti0 + i
        w.prepare(tti % 10, (tti // 10) % 1024, i % 20 == 0)
        phy.putPending(w)
    phy.joinPending()
    assert gpu_records(phy) == orecs
    phy.close()
