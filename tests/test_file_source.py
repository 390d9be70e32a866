"""IQ capture file source (file mode of the reference: -i file -O offset -o cfo): oracle reader vs numpy on CPU; product
replay (lsn_phy_process_file: reader thread + k_file_unpack + engine) vs the oracle worker on the oracle-read subframes on GPU."""
import ctypes as C
import os

import numpy as np
import pytest

from lsn_testlib import OracleWorker, oracle, parse_pcap, scenario
from parity import gen_subframes, oracle_records


def write_capture(path, iq, lead=0, seed=0):
    """iq [nsf, nant, sflen] -> cf32 file, antennas interleaved per sample, `lead` junk samples per antenna in front, 100 trailing
    samples that do not make a subframe"""
    nsf, nant, sflen = iq.shape
    rng = np.random.default_rng(seed)
    body = np.transpose(iq, (0, 2, 1)).reshape(-1, nant)
    junk = (rng.standard_normal((lead, nant)) + 1j * rng.standard_normal((lead, nant))).astype(np.complex64)
    tail = np.zeros((100, nant), dtype=np.complex64)
    np.concatenate([junk, body, tail]).astype(np.complex64).tofile(path)


def oracle_read(path, nprb, nant, offset_time, offset_freq, first, nsf):
    o = oracle()
    o.o_file_read.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_long, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p]
    o.o_file_read.restype = C.c_long
    sflen = 15 * o.o_fft_size(nprb)
    out = np.zeros((nsf, nant, sflen), dtype=np.complex64)
    n = o.o_file_read(os.fsencode(path), nprb, nant, offset_time, offset_freq, first, nsf, out.ctypes.data)
    return n, out


def test_oracle_reader_deinterleaves_skips_and_rotates(tmp_path):
    sc = scenario("small", seed=4)
    tti0, iq, _ = gen_subframes(sc, 7)
    p = str(tmp_path / "cap.cf32")
    write_capture(p, iq, lead=333)
    n, got = oracle_read(p, sc["nof_prb"], 2, 333, 0.0, 0, 10)
    assert n == 7 and np.array_equal(got[:7], iq)
    n, got = oracle_read(p, sc["nof_prb"], 2, 333, 0.0, 5, 10)  # starting at subframe 5; the trailing fragment is dropped
    assert n == 2 and np.array_equal(got[:2], iq[5:7])
    f = 1234.5
    n, got = oracle_read(p, sc["nof_prb"], 2, 333, f, 0, 7)
    sflen = iq.shape[2]
    rot = np.exp(-2j * np.pi * f * np.arange(sflen) / (15000.0 * sflen / 15))
    assert n == 7 and np.allclose(got, iq * rot[None, None, :], atol=2e-6 * np.abs(iq).max())
    assert oracle_read(str(tmp_path / "missing"), 25, 2, 0, 0.0, 0, 1)[0] == -1


@pytest.mark.gpu
@pytest.mark.parametrize("scn,nsf,lead,cfo,block,mm", [("small", 50, 0, 0.0, 16, 0), ("small", 45, 777, 2500.0, 7, 0), ("cfg3", 24, 64, -800.0, 10, 0),
                                                       ("small", 45, 777, 2500.0, 7, 1)])
def test_process_file_matches_oracle_worker(tmp_path, monkeypatch, scn, nsf, lead, cfo, block, mm):
    """mm = 0 (default): pread() into pinned buffers; mm = 1: blocks are page-locked in the file mapping and cross PCIe from the page cache"""
    monkeypatch.setenv("LSN_FILE_MMAP", str(mm))
    import ltesniffer_amd as la
    from parity import gpu_records
    sc = scenario(scn, seed=31)
    tti0, iq, _ = gen_subframes(sc, nsf)
    if cfo:
        sflen = iq.shape[2]
        iq = (iq * np.exp(2j * np.pi * cfo * np.arange(sflen) / (15000.0 * sflen / 15))[None, None, :]).astype(np.complex64)  # the capture is off by +cfo
    p = str(tmp_path / "cap.cf32")
    write_capture(p, iq, lead=lead)
    n, sub = oracle_read(p, sc["nof_prb"], sc["nof_rx"], lead, cfo, 0, nsf + 5)
    assert n == nsf
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"])
    for i in range(nsf):
        ow.work(sub[i], tti0 + i, update_meta=1 if i % 20 == 0 else 0)
    orecs = oracle_records(parse_pcap(ow.pcap_bytes()))
    assert len(orecs) >= nsf // 2
    monkeypatch.setenv("LSN_FILE_BLOCK", str(block))  # several blocks, the last one partial
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=8, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    done = phy.process_file(p, start_tti=tti0, offset_time=lead, offset_freq=cfo, update_meta_period=20)
    assert done == nsf
    assert gpu_records(phy) == orecs
    # max_subframes stops early; a wrong antenna count is refused
    phy2 = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=8, pcapwriter=la.PcapWriter(None))
    assert la.lib().lsn_phy_prepare_file(phy2._h, sc["nof_rx"]) == la.LSN_ERROR          # no cell yet
    assert phy2.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    assert la.lib().lsn_phy_prepare_file(phy2._h, sc["nof_rx"] + 1) == la.LSN_ERROR_INVALID_INPUTS
    phy2.prepare_file()   # block buffers reserved ahead of the replay (lsn_phy_prepare_file): the replay finds them
    assert phy2.process_file(p, start_tti=tti0, offset_time=lead, offset_freq=cfo, max_subframes=block + 3) == block + 3
    fc = la.FileCfg(sc["nof_rx"] + 1, 0, 0.0)
    assert la.lib().lsn_phy_process_file(phy2._h, os.fsencode(p), C.byref(fc), 0, 0, 0, None) == la.LSN_ERROR_INVALID_INPUTS
    phy.close(); phy2.close()
