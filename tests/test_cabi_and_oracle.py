"""CPU tests: the C-ABI library loads and exports everything include/ltesniffer_amd.h declares (and refuses to run
without a GPU); structural known-answer tests of the oracle's primitives; transmitter -> oracle loop-back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import ltesniffer_amd as la
from lsn_testlib import OCell, OracleWorker, ROOT, TxGen, oracle, parse_pcap, scenario


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ltesniffer_amd.h")).read()
    declared = set(re.findall(r"\b(lsn_[a-z0-9_]+)\s*\(", hdr)) - {"lsn_pdu_sink_t"}
    assert len(declared) >= 30
    lib = la.lib()
    for sym in sorted(declared):
        assert hasattr(lib, sym), sym
    assert declared == set(la.EXPORTS), declared ^ set(la.EXPORTS)
    assert lib.lsn_version().decode().startswith("ltesniffer_amd")
    assert lib.lsn_kernel_name(9) == b"k_turbo<64>" and lib.lsn_kernel_name(11) == b"k_turbo<128>"


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        la.Phy(nof_rx_antennas=1)


def _run_hpp_program():
    import subprocess
    la.lib()  # the product library must exist before the C++ program links against it
    d = os.path.join(ROOT, "tests", "native")
    subprocess.check_call(["make", "-C", d, "_build/test_hpp"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(d, "_build", "test_hpp")], capture_output=True, text=True, timeout=120)
    return r.returncode, r.stdout


def test_cpp_mirror_compiles_links_and_fails_loudly_without_a_device():
    """include/ltesniffer_amd.hpp (the drop-in Phy / SubframeWorker classes) against the shared library from plain g++; on a box without a
    HIP device the constructor throws - there is no CPU path"""
    import torch
    rc, out = _run_hpp_program()
    if torch.cuda.is_available():
        assert rc == 0 and "device path ok" in out, out
    else:
        assert rc == 10 and "no HIP device" in out, out


@pytest.mark.gpu
def test_cpp_mirror_worker_pool_round_trip_on_the_gpu():
    rc, out = _run_hpp_program()
    assert rc == 0 and "device path ok: 1 subframes" in out, out


def test_product_does_not_link_or_include_the_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "ltesniffer_amd")):
        for f in fs:
            if f.endswith((".cc", ".h", ".hip", ".py", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r'#include\s+"[^"]*oracle|lsn_oracle\.h|liblsn_oracle|from lsn_testlib|import lsn_testlib', txt):
                    bad.append(f)
    assert not bad, bad


# ---------------------------------------------------------------------------------------------- oracle primitives
def _bits(x, n):
    return np.array([(x >> (n - 1 - i)) & 1 for i in range(n)], dtype=np.uint8)


def test_crc_known_answers():
    o = oracle()
    msg = np.unpackbits(np.frombuffer(b"123456789", dtype=np.uint8))
    # CRC-16/XMODEM ("123456789") = 0x31C3: same generator 0x1021, zero init, no reflection (36.212 gCRC16)
    assert o.o_crc_bits(0x11021, 16, msg.ctypes.data, len(msg)) == 0x31C3
    # CRC-24/LTE-A and LTE-B check values
    assert o.o_crc_bits(0x1864CFB, 24, msg.ctypes.data, len(msg)) == 0xCDE703
    assert o.o_crc_bits(0x1800063, 24, msg.ctypes.data, len(msg)) == 0x23EF52
    # linearity + divisibility: data || crc leaves remainder 0
    rng = np.random.default_rng(1)
    for poly, order in ((0x1864CFB, 24), (0x1800063, 24), (0x11021, 16), (0x19B, 8)):
        d = rng.integers(0, 2, 200).astype(np.uint8)
        c = o.o_crc_bits(poly, order, d.ctypes.data, len(d))
        full = np.concatenate([d, _bits(c, order)])
        # data||crc is divisible by g(x): dividing WITHOUT a further augmentation leaves 0
        reg = 0
        for b in full:
            reg = (reg << 1) | int(b)
            if reg >> order:
                reg ^= poly
        assert reg == 0


def test_gold_sequence_two_ways():
    o = oracle()
    for cinit in (0, 1, 12345, 0x7FFFFFFF, (0x46 << 14) | (5 << 9) | 1):
        c = np.zeros(400, dtype=np.uint8)
        o.o_gold(cinit, c.ctypes.data, 400)
        # direct definition of 36.211 7.2 with explicit arrays
        n = 1600 + 400 + 31
        x1 = np.zeros(n, dtype=np.uint8); x2 = np.zeros(n, dtype=np.uint8)
        x1[0] = 1
        for i in range(31):
            x2[i] = (cinit >> i) & 1
        for i in range(n - 31):
            x1[i + 31] = x1[i + 3] ^ x1[i]
            x2[i + 31] = x2[i + 3] ^ x2[i + 2] ^ x2[i + 1] ^ x2[i]
        ref = x1[1600:2000] ^ x2[1600:2000]
        assert np.array_equal(c, ref), cinit


def test_qpp_table_is_a_permutation_and_contention_free():
    o = oracle()
    f1, f2 = C.c_int(), C.c_int()
    sizes = list(range(40, 513, 8)) + list(range(528, 1025, 16)) + list(range(1056, 2049, 32)) + list(range(2112, 6145, 64))
    assert len(sizes) == 188
    o.o_qpp_find.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    for K in sizes:
        assert o.o_qpp_find(K, C.byref(f1), C.byref(f2)) >= 0, K
        i = np.arange(K, dtype=np.int64)
        pi = (f1.value * i + f2.value * i * i) % K
        assert len(np.unique(pi)) == K, K
        P = o.o_turbo_nwin(K)
        W = K // P
        assert K % P == 0 and (W >= 32 or P == 1) and W <= 96
        # contention-free: at every step the P windows address P different memory banks x // W
        banks = (pi.reshape(P, W) // W)
        assert all(len(np.unique(banks[:, t])) == P for t in range(0, W, max(1, W // 7))), K


def test_fft_matches_numpy():
    o = oracle()
    rng = np.random.default_rng(3)
    o.o_fft_twiddle_len.argtypes = [C.c_int]
    for N in (128, 512, 1536, 2048):  # 1536 (15 MHz) = 3 x 512 with a radix-3 combination
        w = np.zeros(o.o_fft_twiddle_len(N), dtype=np.complex64)
        assert w.size == (256 + 1536 if N == 1536 else N // 2)
        o.o_fft_twiddles(N, w.ctypes.data)
        x = (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)
        y = x.copy()
        o.o_fft(N, w.ctypes.data, y.ctypes.data)
        ref = np.fft.fft(x.astype(np.complex128))
        assert np.max(np.abs(y - ref)) < 2e-3 * np.max(np.abs(ref))


def test_dci_conv_round_trip_noise_free():
    """encode (transmitter) -> rate match -> oracle candidate decoder recovers payload and RNTI for all sizes / levels"""
    tx = TxGen(**scenario("cfg2", seed=4))
    o = oracle()
    ow = OracleWorker(100, 2, 1, 2)
    ok = 0
    for _ in range(6):
        tti, iq, pdus = tx.next()
        ow.work(iq, tti)
        acc = {(a[0], a[2], a[3]) for a in ow.accepted()}
        for p in pdus:
            if p["is_ul"] or p["tb"] != 0:
                continue
            ok += (p["rnti"], p["L"], p["ncce"]) in acc
    assert ok >= 10


def test_transmitter_oracle_loopback_payload_equality():
    """every MAC PDU the oracle emits equals the transmitted one (high SNR); nothing spurious"""
    for scn, n, seed, over in (("small", 40, 2, {}), ("cfg1", 40, 1, {}), ("cfg3", 24, 3, {}), ("cfg2", 16, 6, dict(nof_prb=75, cell_id=77, n_rnti=20))):
        sc = scenario(scn, seed=seed, **over)
        tx = TxGen(**sc)
        ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"], sc["phich_ng_x6"])
        sent = {}
        for _ in range(n):
            tti, iq, pdus = tx.next()
            ow.work(iq, tti)
            for p in pdus:
                if not p["is_ul"]:
                    sent.setdefault((tti, p["rnti"]), []).append(p["payload"])
        recs = parse_pcap(ow.pcap_bytes())
        assert len(recs) > n // 2, (scn, len(recs))
        for r in recs:
            tti = r["sfn"] * 10 + r["sf"]
            assert r["pdu"] in sent.get((tti, r["rnti"]), []), (scn, tti, hex(r["rnti"]))
