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


def oracle_read(path, nprb, nant, offset_time, offset_freq, first, nsf, fmt=0, scale=0.0):
    o = oracle()
    o.o_file_read.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_long, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p]
    o.o_file_read.restype = C.c_long
    o.o_file_read_fmt.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_long, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]
    o.o_file_read_fmt.restype = C.c_long
    sflen = 15 * o.o_fft_size(nprb)
    out = np.zeros((nsf, nant, sflen), dtype=np.complex64)
    if fmt == 0:
        n = o.o_file_read(os.fsencode(path), nprb, nant, offset_time, offset_freq, first, nsf, out.ctypes.data)
    else:
        n = o.o_file_read_fmt(os.fsencode(path), nprb, nant, offset_time, offset_freq, first, nsf, fmt, scale, out.ctypes.data)
    return n, out


def write_integer_capture(path, iq, fmt, lead=0, pow2=True):
    """iq [nsf, nant, sflen] -> int16 (fmt 1) / int8 (fmt 2) I/Q pairs, antennas interleaved per sample, `lead` junk samples per antenna in
    front, a trailing fragment; -> (value of one LSB, the integers [nsf, nant, sflen, 2])"""
    nsf, nant, sflen = iq.shape
    full = 32767 if fmt == 1 else 127
    peak = float(max(np.abs(iq.real).max(), np.abs(iq.imag).max()))
    gain = 2.0 ** np.floor(np.log2(0.98 * full / peak)) if pow2 else 0.93 * full / peak   # (not a power of two: the conversion rounds)
    q = np.stack([np.rint(iq.real * gain), np.rint(iq.imag * gain)], axis=-1).astype(np.int16 if fmt == 1 else np.int8)
    body = np.transpose(q, (0, 2, 1, 3)).reshape(-1, nant, 2)
    junk = np.random.default_rng(5).integers(-full, full, (lead, nant, 2)).astype(q.dtype)
    tail = np.zeros((77, nant, 2), dtype=q.dtype)
    np.concatenate([junk, body, tail]).tofile(path)
    return float(np.float32(1.0 / gain)), q


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


def test_oracle_reader_converts_integer_samples(tmp_path):
    """int16 / int8 I/Q pairs (the product's extension of the file source): one LSB = scale, exact for a power of two, one float
    rounding otherwise; offset and de-interleave count SAMPLES whatever their width; the default scale is full scale +-1"""
    sc = scenario("small", seed=4)
    tti0, iq, _ = gen_subframes(sc, 6)
    for fmt, pow2 in ((1, True), (1, False), (2, True), (2, False)):
        p = str(tmp_path / ("cap%d%d" % (fmt, pow2)))
        lsb, q = write_integer_capture(p, iq, fmt, lead=211, pow2=pow2)
        want = (q[..., 0].astype(np.float32) * np.float32(lsb) + 1j * (q[..., 1].astype(np.float32) * np.float32(lsb))).astype(np.complex64)
        n, got = oracle_read(p, sc["nof_prb"], 2, 211, 0.0, 0, 9, fmt, lsb)
        assert n == 6 and np.array_equal(got[:6], want)
        n, got = oracle_read(p, sc["nof_prb"], 2, 211, 0.0, 4, 9, fmt, lsb)
        assert n == 2 and np.array_equal(got[:2], want[4:])
        if fmt == 1 and pow2:   # quantisation noise of the 16-bit capture: 80 dB under the signal
            assert np.abs(want - iq).max() <= 0.5 * lsb * 1.4143 and 10 * np.log10(np.mean(np.abs(iq) ** 2) / np.mean(np.abs(want - iq) ** 2)) > 70
        n, got = oracle_read(p, sc["nof_prb"], 2, 211, 0.0, 0, 6, fmt, 0.0)
        full = np.float32(1 / 32768.0 if fmt == 1 else 1 / 128.0)
        assert n == 6 and np.array_equal(got, (q[..., 0].astype(np.float32) * full + 1j * (q[..., 1].astype(np.float32) * full)).astype(np.complex64))
    assert oracle_read(p, sc["nof_prb"], 2, 0, 0.0, 0, 1, 3, 1.0)[0] == -1


@pytest.mark.gpu
@pytest.mark.parametrize("scn,nsf,fmt,pow2,lead,cfo,block", [("small", 45, 1, True, 0, 0.0, 16), ("cfg3", 24, 1, False, 64, -800.0, 10), ("small", 45, 2, False, 333, 1500.0, 7)])
def test_process_file_integer_samples_match_oracle_worker(tmp_path, monkeypatch, scn, nsf, fmt, pow2, lead, cfo, block):
    """lsn_file_cfg_t.sample_format: an int16 / int8 recording replays like the oracle's worker on the oracle-read subframes - and an sc16 recording
    of a 30 dB capture decodes what the cf32 one did"""
    import ltesniffer_amd as la
    from parity import gpu_records
    sc = scenario(scn, seed=32)
    tti0, iq, _ = gen_subframes(sc, nsf)
    if cfo:
        sflen = iq.shape[2]
        iq = (iq * np.exp(2j * np.pi * cfo * np.arange(sflen) / (15000.0 * sflen / 15))[None, None, :]).astype(np.complex64)
    p = str(tmp_path / "cap.int")
    lsb, _ = write_integer_capture(p, iq, fmt, lead=lead, pow2=pow2)
    n, sub = oracle_read(p, sc["nof_prb"], sc["nof_rx"], lead, cfo, 0, nsf + 5, fmt, lsb)
    assert n == nsf
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"])
    for i in range(nsf):
        ow.work(sub[i], tti0 + i, update_meta=1 if i % 20 == 0 else 0)
    orecs = oracle_records(parse_pcap(ow.pcap_bytes()))
    assert len(orecs) >= nsf // 2
    monkeypatch.setenv("LSN_FILE_BLOCK", str(block))
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=8, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    assert phy.process_file(p, start_tti=tti0, offset_time=lead, offset_freq=cfo, update_meta_period=20, sample_format=fmt, sample_scale=lsb) == nsf
    assert gpu_records(phy) == orecs
    if fmt == 1:   # the same subframes as cf32 through the same engine state: 16 bits lose nothing that decodes
        pc = str(tmp_path / "cap.cf32")
        write_capture(pc, iq, lead=lead)
        phy2 = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=8, pcapwriter=la.PcapWriter(None))
        assert phy2.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
        assert phy2.process_file(pc, start_tti=tti0, offset_time=lead, offset_freq=cfo, update_meta_period=20) == nsf
        assert gpu_records(phy2) == orecs
        phy2.close()
    fc = la.FileCfg(sc["nof_rx"], 0, 0.0, 3, 0.0)   # an unknown format is refused, and so is a scale that is not a number
    assert la.lib().lsn_phy_process_file(phy._h, os.fsencode(p), C.byref(fc), 0, 0, 0, None) == la.LSN_ERROR_INVALID_INPUTS
    fc = la.FileCfg(sc["nof_rx"], 0, 0.0, 1, float("nan"))
    assert la.lib().lsn_phy_process_file(phy._h, os.fsencode(p), C.byref(fc), 0, 0, 0, None) == la.LSN_ERROR_INVALID_INPUTS
    phy.close()


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


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,pow2", [(1, True), (2, False)])
def test_process_host_integer_samples_match_oracle_worker(tmp_path, fmt, pow2):
    """lsn_phy_process_host_int: integer I/Q pairs in the caller's memory ([subframe][antenna][sample], what a radio driver delivers unconverted) cross
    PCIe as they are and are converted behind the copy - the records are the oracle's on (float)integer * lsb; 15 blocks through a ring of 12"""
    import ltesniffer_amd as la
    from parity import gpu_records
    sc = scenario("small", seed=33)
    nsf = 118
    tti0, iq, _ = gen_subframes(sc, nsf)
    lsb, q = write_integer_capture(str(tmp_path / "unused"), iq, fmt, pow2=pow2)
    sub = (q[..., 0].astype(np.float32) * np.float32(lsb) + 1j * (q[..., 1].astype(np.float32) * np.float32(lsb))).astype(np.complex64)
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"])
    for i in range(nsf):
        ow.work(sub[i], tti0 + i, update_meta=1 if i % 20 == 0 else 0)
    orecs = oracle_records(parse_pcap(ow.pcap_bytes()))
    assert len(orecs) >= nsf // 2
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=8, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    phy.process_host_int(q, tti0, update_meta_period=20, sample_scale=lsb)
    assert gpu_records(phy) == orecs
    L = la.lib()   # cf32 belongs to lsn_phy_process_host; a scale that is not a number is refused
    assert L.lsn_phy_process_host_int(phy._h, q.ctypes.data, la.FILE_CF32, 1.0, 1, 0, 0) == la.LSN_ERROR_INVALID_INPUTS
    assert L.lsn_phy_process_host_int(phy._h, q.ctypes.data, 3, 1.0, 1, 0, 0) == la.LSN_ERROR_INVALID_INPUTS
    assert L.lsn_phy_process_host_int(phy._h, q.ctypes.data, fmt, float("nan"), 1, 0, 0) == la.LSN_ERROR_INVALID_INPUTS
    phy.close()
