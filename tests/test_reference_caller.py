"""The boundary, proven with the reference's own caller: /root/reference/src/src/LTESniffer_Core.cc is compiled (syntax and types; -fsyntax-only) against
include/ltesniffer_amd_compat.hpp.  The only edit is the include swap INTEGRATION.md section 2 describes, applied here to TEMPORARY copies (nothing of the
reference is kept in the repo): in LTESniffer_Core.h the worker-side headers are replaced by the compat header; LTESniffer_Core.cc is used as it is.
boost::program_options, srsue and the srsRAN radio / synchronisation API are absent from the image and outside the path: tests/native/core_shim declares them
(oracle/ref_shim_search's stand-in declarations + the RF / ue_sync / ue_mib names this file uses).  Needs /root/reference (this container; skipped on the GPU box)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SWAPPED = ('#include "include/SubframeWorker.h"', '#include "include/WorkerThread.h"', '#include "MCSTracking.h"', '#include "ULSchedule.h"', '#include "Phy.h"',
           '#include "PcapWriter.h"', '#include "HARQ.h"')


def _prepare(tmp):
    os.makedirs(os.path.join(tmp, "include"))
    h = open(os.path.join(REF, "src/include/LTESniffer_Core.h")).read()
    for i, inc in enumerate(SWAPPED):
        assert h.count(inc) == 1, inc
        h = h.replace(inc, '#include "ltesniffer_amd_compat.hpp"' if i == 0 else "// (swapped) " + inc[1:])
    open(os.path.join(tmp, "include/LTESniffer_Core.h"), "w").write(h)
    shutil.copy(os.path.join(REF, "src/src/LTESniffer_Core.cc"), os.path.join(tmp, "LTESniffer_Core.cc"))


def _compile(tmp, extra=(), our_headers=None):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-w", "-I", tmp, "-I", os.path.join(ROOT, "tests/native/core_shim"), "-I", os.path.join(ROOT, "oracle/ref_shim_search"),
           "-I", our_headers or os.path.join(ROOT, "include"), "-I", os.path.join(REF, "src"), "-I", os.path.join(REF, "src/include"), "-I", os.path.join(REF, "lib/include"),
           *extra, os.path.join(tmp, "LTESniffer_Core.cc")]
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src/src")), reason="the reference tree is not on this machine")
def test_the_references_own_core_compiles_against_the_compat_header(tmp_path):
    tmp = str(tmp_path / "core")
    _prepare(tmp)
    r = _compile(tmp)
    assert r.returncode == 0, r.stderr[-6000:]
    # (the file's own -DDISABLE_RF variant does not compile in the reference either: `rf` is used outside the guard at :551)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src/src")), reason="the reference tree is not on this machine")
def test_the_check_notices_a_missing_member(tmp_path):
    """the compile is a real check: hide one member the core calls (PhyCommon::setShortcutDiscovery, LTESniffer_Core.cc:87,616) and it fails"""
    tmp = str(tmp_path / "core")
    _prepare(tmp)
    inc = str(tmp_path / "inc")
    shutil.copytree(os.path.join(ROOT, "include"), inc)
    hpp = open(os.path.join(inc, "ltesniffer_amd.hpp")).read()
    assert "void setShortcutDiscovery(bool enable)" in hpp
    open(os.path.join(inc, "ltesniffer_amd.hpp"), "w").write(hpp.replace("void setShortcutDiscovery(bool enable)", "void setShortcutDiscovery_hidden(bool enable)"))
    r = _compile(tmp, our_headers=inc)
    assert r.returncode != 0 and "setShortcutDiscovery" in r.stderr
