"""The turbo kernel's per-lane text (ltesniffer_amd/csrc/kernels/lsn_turbo_core.h - the source k_turbo is compiled from) runs on the CPU
against the oracle's decoder: tests/native/test_turbo_core.cc emulates a workgroup lane by lane over plain memory, with every packed int16
add range-checked against 32-bit arithmetic.  A difference in iteration count, CRC verdict or any decided bit fails the run.  No GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = os.environ.get("TURBO_CXX", "/opt/rocm/lib/llvm/bin/clang++")


@pytest.mark.skipif(not (os.path.exists(CLANG) or shutil.which(CLANG)), reason="needs clang (ext_vector_type)")
def test_kernel_text_matches_the_oracle_decoder_on_every_block_size():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "native"), "_build/test_turbo_core"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(ROOT, "tests", "native", "_build", "test_turbo_core")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    cases, _, iters, _, bad, _ = out.stdout.split()[:6]
    assert int(bad) == 0 and int(cases) >= 1100 and int(iters) > 8000, out.stdout


def test_interleaver_table_offsets_cover_all_block_sizes():
    """turbo_il_offset (host) and the table order: 188 sizes, offsets = running sum of ceil(W / 2) * P words, and lsn_turbo_two_wave_class splits at 64 windows / K 3072"""
    from lsn_testlib import hosttest
    h = hosttest()
    import re
    txt = open(os.path.join(ROOT, "spec", "lte_tables.h")).read()
    m = re.search(r"lsn_qpp_table\[LSN_QPP_NSIZES\]\[3\] = \{(.*?)\};", txt, re.S)
    ks = [int(x.split(",")[0]) for x in re.findall(r"\{(\d+,\d+,\d+)\}", m.group(1))]
    assert len(ks) == 188
    def nwin(K):
        p1 = next(P for P in range(min(K // 32, 64), 0, -1) if K % P == 0)
        p2 = next(P for P in range(min(K // 32, 128), 0, -1) if K % P == 0)
        return p2 if p2 >= 96 else p1
    off = 0
    for k in ks:
        assert h.lsnh_turbo_il_offset(k) == off
        P = nwin(k)
        off += ((k // P + 1) // 2) * P  # two trellis steps per table word
    assert h.lsnh_turbo_il_offset(0) == off
    two = [k for k in ks if h.lsnh_turbo_two_wave_class(k)]
    assert min(two) == 3072 and set(two) == set(k for k in ks if k >= 3072)  # K = 3072 has 96 windows; every larger size needs either two waves or > 22 KiB of LDS


def test_layout_cycle_header_is_what_the_generator_derives():
    """lsn_turbo_cyc.h (the shuffle-free steps over seven register layouts) is generated: tools/turbo_layouts.py derives the cycle from the constituent code's
    transitions, checks every step it emits against an eight-state reference, and must reproduce the committed header byte for byte."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("turbo_layouts", os.path.join(ROOT, "tools", "turbo_layouts.py"))
    tl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tl)
    assert len(tl.T) == 7 and tl.T[0] == [(0, 4), (1, 5), (2, 6), (3, 7)]
    for L in range(7):  # every state once per layout, state 0 in the low half of register 0 (the half the normalisation subtracts)
        assert sorted(s for pr in tl.T[L] for s in pr) == list(range(8)) and tl.T[L][0][0] == 0
        assert set(map(frozenset, tl.need(tl.T[L]))) == set(map(frozenset, tl.T[(L + 1) % 7]))   # the pairing the next step's beta vector must have
    tl.check()
    committed = open(os.path.join(ROOT, "ltesniffer_amd", "csrc", "kernels", "lsn_turbo_cyc.h")).read()
    assert tl.emit() == committed
