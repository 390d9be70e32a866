#!/usr/bin/env python3
"""tools/search_pin_mutations.py - does the pin of the blind DCI search on the reference's own code (tests/test_ref_dci_search.py, tests/golden/dci_search_ref.json)
notice a wrong restatement?  One-token mutations of the ORACLE's search (oracle/o_worker.c: inspect / blind_search / decode_msg) are built in a scratch copy of
oracle/ (nothing in the repo is touched), the ten streams of tests/ref_dci_search.py are walked with each mutated oracle at suite length, and every subframe's
accepted-DCI list is compared with what the reference's DCISearch.cc decided (the committed per-subframe digests), the search statistics and the activation reasons of all RNTIs with
the reference's totals - what tests/test_ref_dci_search.py::test_oracle_search_decides_like_the_reference asserts.  Prints, per mutation, how many streams notice
and where first.  -> profiles/r05_search_pin_mutations.txt

  python tools/search_pin_mutations.py [--only N]"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MUTATIONS = [
    ("tie-break among formats: > becomes >=", "        if (h > hmax) {", "        if (h >= hmax) {"),
    ("an ambiguous accepted location is never disambiguated", "  if (n_ok > 0 && cand[hist_max_idx].match == 1) { /* :288-298 */", "  if (n_ok > 0 && cand[hist_max_idx].match == 3) { /* :288-298 */"),
    ("the aggregation level is not lowered after a disambiguation", "    uint32_t Ld = disamb > 0 ? L - 1 : L;", "    uint32_t Ld = L;"),
    ("RA-RNTI format filter: > RARNTI_START becomes >=", "    if (cand[fi].rnti > O_RARNTI_START && cand[fi].rnti < O_RARNTI_END) /* :181-197 */",
     "    if (cand[fi].rnti >= O_RARNTI_START && cand[fi].rnti < O_RARNTI_END) /* :181-197 */"),
    ("DCI 0 of a freshly random-accessed RNTI is not kept aside", "      if (add && w->ntemp0 < 64) {", "      if (0 && add && w->ntemp0 < 64) {"),
    ("the right half of a failed location inherits the parent's candidates (shortcut where the reference has none)",
     "        rr += inspect(w, map, ncce + (1u << (L - 1)), L - 1, max_depth - 1, metas, nformats, discovery, NULL);",
     "        rr += inspect(w, map, ncce + (1u << (L - 1)), L - 1, max_depth - 1, metas, nformats, discovery, cand);"),
    ("locations stay checked for the secondary formats", "    for (uint32_t i = 0; i < nloc; i++) locs[i].checked = 0;", "    for (uint32_t i = 0; i < nloc; i++) locs[i].checked = locs[i].checked;"),
    ("candidate decode asks for srsRAN's default mean soft-bit bound of 0.5 instead of 0", "  if (mean > 0.0) {", "  if (mean > 0.5) {"),
    ("accepted formats are not counted for the primary / secondary split", "    metas[hist_max_idx]->hits++;", "    metas[hist_max_idx]->hits += 0;"),
    ("CCE power threshold 0.7 becomes 0.75 in the location map", "    if (map[c].power < 0.7f)\n      for (int a = 0; a < 4; a++)", "    if (map[c].power < 0.75f)\n      for (int a = 0; a < 4; a++)"),
    ("a location that overlaps an accepted DCI is not marked occupied", "          map[ci].location[a]->occupied = 1;", "          map[ci].location[a]->occupied = 0;"),
    ("search-space verdict ignored: every position is valid", "    if (cand[fi].match == 0) {", "    if (0 && cand[fi].match == 0) {"),
]

CHILD = r'''
import json, os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests")]
import lsn_testlib
lsn_testlib.ORACLE_SO = %(so)r
lsn_testlib._ensure = lambda so, d: so
import ref_dci_search as R
fix = json.load(open(os.path.join(%(root)r, "tests", "golden", "dci_search_ref.json")))
out = []
for case in R.CASES:
    f = fix["cases"][case[0]]
    r = R.walk(case)
    assert r["llr_sha256"] == f["llr_sha256"]
    got = [R.sf_digest(i, a) for i, a in enumerate(r["oracle"]["per_sf"])]
    bad = [i for i, (a, b) in enumerate(zip(got, f["reference"]["per_subframe"])) if a != b]
    fr = f["reference"]
    totals = r["oracle"]["stats"][:4] != fr["stats_locations_decoded_cce_missed_subframes"][:4] or r["oracle"]["reasons"] != fr["activation_reasons_unset_evergreen_rar_shortcut_histogram_other"]
    out.append((case[0], len(bad), bad[0] if bad else None, r["subframes"], totals))
print(json.dumps(out))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", type=int, default=-1)
    a = ap.parse_args()
    src = open(os.path.join(ROOT, "oracle", "o_worker.c")).read()
    lines = ["one-token mutations of the oracle's search against the reference's decisions (tests/golden/dci_search_ref.json), ten streams at suite length", ""]
    for k, (what, old, new) in enumerate(MUTATIONS):
        if a.only >= 0 and k != a.only:
            continue
        assert src.count(old) == 1, (k, what, src.count(old))
        with tempfile.TemporaryDirectory() as tmp:
            shutil.copytree(os.path.join(ROOT, "oracle"), os.path.join(tmp, "oracle"), ignore=shutil.ignore_patterns("_build", "_ref"))
            shutil.copytree(os.path.join(ROOT, "spec"), os.path.join(tmp, "spec"))
            open(os.path.join(tmp, "oracle", "o_worker.c"), "w").write(src.replace(old, new))
            subprocess.check_call(["make", "-C", os.path.join(tmp, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            so = os.path.join(tmp, "oracle", "_build", "liblsn_oracle.so")
            res = json.loads(subprocess.check_output([sys.executable, "-c", CHILD % dict(root=ROOT, so=so)], stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1])
        hit = [r for r in res if r[1]]
        tot = [r for r in res if r[4]]
        first = min(hit, key=lambda r: r[2]) if hit else None
        line = "%2d  %-100s accepted DCI differ in %2d of %d streams%s; statistics / activation reasons differ in %2d%s" % (
            k, what, len(hit), len(res), (", first in subframe %d of %s (%d subframes differ there)" % (first[2], first[0], first[1])) if first else "", len(tot),
            "" if hit or tot else "  <-- NOT NOTICED")
        print(line, flush=True)
        lines.append(line)
    if a.only < 0:
        open(os.path.join(ROOT, "profiles", "r05_search_pin_mutations.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
