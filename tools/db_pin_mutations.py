#!/usr/bin/env python3
"""tools/db_pin_mutations.py - do the pins of the MCS-tracking database and of the HARQ database on the reference's own code (tests/test_mcs_ageing.py,
tests/test_ref_harq.py; tests/golden/mcs_tracking_ref.json, harq_ref.json) notice a wrong product?  One-token mutations of the PRODUCT's host classes
(ltesniffer_amd/csrc/host/lsn_lte.cc: MCSTracking, HarqDatabase) are built into a scratch copy of the host-test glue (nothing in the repo is touched) and
the random lives and the corner scripts of the two tests are replayed against the committed answers of the reference.  -> profiles/r05_db_pin_mutations.txt"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MUTATIONS = [
    ("mcs", "look-up answers FULL_BUFFER one entry early", "return count < max_size ? TABLE_UNKNOWN : TABLE_FULL_BUFFER;", "return count + 1 < max_size ? TABLE_UNKNOWN : TABLE_FULL_BUFFER;"),
    ("mcs", "table accepted after 3 instead of more than 3 messages behind a RAR", "if (e.nof_msg_after_rar > rar_thresold) {", "if (e.nof_msg_after_rar >= rar_thresold) {"),
    ("mcs", "a RAR does not reset the table", "db[crnti].has_rar = 1; db[crnti].table = TABLE_UNKNOWN; }", "db[crnti].has_rar = 1; }"),
    ("mcs", "ageing: idle for 5 whole seconds is enough (reference: more than 5)", "if (cur_interval > interval || wrong_detect || e.nof_active == 0) {", "if (cur_interval >= interval || wrong_detect || e.nof_active == 0) {"),
    ("mcs", "success-rate floor 15 % becomes inclusive", "< 0.15f && e.table != TABLE_UNKNOWN) {", "<= 0.15f && e.table != TABLE_UNKNOWN) {"),
    ("mcs", "wrong-detection rule looks at up to 11 instead of 10 decodes", "(e.nof_active <= 10 && e.nof_success_mgs == 0", "(e.nof_active <= 11 && e.nof_success_mgs == 0"),
    ("mcs", "a successful decode of a disabled transport block is not counted", "      if (success[i]) e.nof_success_mgs++;", "      if (success[i] && tb_en[i]) e.nof_success_mgs++;"),
    ("mcs", "messages behind a RAR are counted from format 1A on (reference: above 1A)", "if (f > FORMAT1A && e.has_rar) e.nof_msg_after_rar++;", "if (f >= FORMAT1A && e.has_rar) e.nof_msg_after_rar++;"),
    ("harq", "retransmission window 9 instead of 8 subframes", "if (!(cur_tti - last_tti == 8 || cur_tti + 10240 - last_tti == 8)) r = HARQ_NEW_TX;", "if (!(cur_tti - last_tti == 9 || cur_tti + 10240 - last_tti == 9)) r = HARQ_NEW_TX;"),
    ("harq", "the TTI wrap is forgotten", "if (!(cur_tti - last_tti == 8 || cur_tti + 10240 - last_tti == 8)) r = HARQ_NEW_TX;", "if (!(cur_tti - last_tti == 8)) r = HARQ_NEW_TX;"),
    ("harq", "a changed size still counts as a retransmission", "else if (ndi != t.ndi || t.is_first || t.tbs != tbs) r = HARQ_NEW_TX;", "else if (ndi != t.ndi || t.is_first) r = HARQ_NEW_TX;"),
    ("harq", "the NDI is ignored", "else if (ndi != t.ndi || t.is_first || t.tbs != tbs) r = HARQ_NEW_TX;", "else if (t.is_first || t.tbs != tbs) r = HARQ_NEW_TX;"),
    ("harq", "a decoded block is combined again", "else r = t.last_decoded ? HARQ_DECODED : HARQ_RE_TX;", "else r = HARQ_RE_TX;"),
    ("harq", "ageing starts at 11 instead of 10 free entities", "  if (nof_aval > 10) return;", "  if (nof_aval > 11) return;"),
    ("harq", "ageing: idle for 5 whole seconds is enough (reference: more than 5)", "if ((now - e.time) / 1000u > 5u) {", "if ((now - e.time) / 1000u >= 5u) {"),
    ("harq", "the ageing pass forgets to count the entity as free", "      nof_aval++;\n    }\n}", "    }\n}"),
    ("harq", "the FIRST free entity is taken instead of the last", "    else if (ent[i].rnti == 0) avail = i;", "    else if (ent[i].rnti == 0 && avail < 0) avail = i;"),
]

CHILD = r'''
import json, os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests")]
import lsn_testlib
lsn_testlib.HOSTTEST_SO = %(so)r
import ctypes as C
lsn_testlib._host = None
_real = lsn_testlib.hosttest
import subprocess
subprocess_check_call = subprocess.check_call
subprocess.check_call = lambda *a, **k: 0     # hosttest() runs make on the repo's glue: not wanted here
if %(which)r == "mcs":
    import test_mcs_ageing as T
    fix = json.load(open(T.MCS_FIX))["lives"]
    bad = [s for s in T.LIFE_SEEDS if T._life_on(T._Product, s) != (fix[str(s)]["digest"], fix[str(s)]["counters"])]
    b = T._Product()
    print(json.dumps([len(bad), len(T.LIFE_SEEDS), T.corners(b) != json.load(open(T.MCS_FIX))["corner_script"]]))
else:
    import test_ref_harq as T
    fix = json.load(open(T.FIX))["lives"]
    bad = [s for s in T.SEEDS if T._life_on(T.Product, s) != (fix[str(s)]["digest"], fix[str(s)]["verdicts_new_retx_full_decoded_busy"])]
    b = T.Product()
    print(json.dumps([len(bad), len(T.SEEDS), T.corners(b) != json.load(open(T.FIX))["corner_script"]]))
'''


def main():
    src = open(os.path.join(ROOT, "ltesniffer_amd", "csrc", "host", "lsn_lte.cc")).read()
    lines = ["one-token mutations of the product's MCSTracking / HarqDatabase (lsn_lte.cc) against the committed answers of the reference's MCSTracking.cc / HARQ.cc, six random lives each", ""]
    for k, (which, what, old, new) in enumerate(MUTATIONS):
        assert src.count(old) == 1, (k, what, src.count(old))
        with tempfile.TemporaryDirectory() as tmp:
            for d in ("ltesniffer_amd/csrc", "tests/native", "spec", "include"):
                shutil.copytree(os.path.join(ROOT, d), os.path.join(tmp, d), ignore=shutil.ignore_patterns("_build", "*.so", "*.o"))
            open(os.path.join(tmp, "ltesniffer_amd", "csrc", "host", "lsn_lte.cc"), "w").write(src.replace(old, new))
            subprocess.check_call(["make", "-C", os.path.join(tmp, "tests", "native"), "_build/liblsn_hosttest.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            so = os.path.join(tmp, "tests", "native", "_build", "liblsn_hosttest.so")
            out = subprocess.check_output([sys.executable, "-c", CHILD % dict(root=ROOT, so=so, which=which)], stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1]
        bad, n, corner = json.loads(out)
        line = "%2d  %-5s %-85s noticed in %d of %d lives%s%s" % (k, which, what, bad, n, ", by the corner script" if corner else "", "" if bad or corner else "  <-- NOT NOTICED")
        print(line, flush=True)
        lines.append(line)
    open(os.path.join(ROOT, "profiles", "r05_db_pin_mutations.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
