#!/usr/bin/env python3
"""tools/grant_pin_mutations.py - does the pin of the grant conversions on the reference's own code (tests/test_ref_grants.py, tests/golden/grants_ref.json) notice a wrong
restatement?  One-token mutations of the ORACLE's conversions (oracle/o_dci.c) are built in a scratch copy of oracle/, the suite's sweeps (every 23rd of 2.5 million uplink
grants, all MIMO configurations, all common-RNTI grants) are replayed and their digests compared with the committed ones of ul_sniffer_pusch.c / dl_sniffer_pdsch.c.
-> profiles/r05_grant_pin_mutations.txt"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MUTATIONS = [
    ("an odd hopping offset is not rounded up to even", "    if (n_rb_ho % 2) n_rb_ho++;\n    if (n_rb_ho + (nprb % 2) >= nprb) return -1;", "    if (n_rb_ho + (nprb % 2) >= nprb) return -1;"),
    ("+N/4 hop uses N/2", "    if (hop == 0) g->n_prb2 = (n_rb_pusch / 4 + start) % n_rb_pusch;", "    if (hop == 0) g->n_prb2 = (n_rb_pusch / 2 + start) % n_rb_pusch;"),
    ("-N/4 hop: boundary case start == N/4 wraps", "start < n_rb_pusch / 4 ? n_rb_pusch + start - n_rb_pusch / 4", "start <= n_rb_pusch / 4 ? n_rb_pusch + start - n_rb_pusch / 4"),
    ("start below half the offset is accepted", "    if (start < n_rb_ho / 2) return -1;", "    if (0 && start < n_rb_ho / 2) return -1;"),
    ("the second slot may run past the band", "    if (g->n_prb2 + L > nprb) return -1;", "    if (g->n_prb2 + L > nprb + 1) return -1;"),
    ("256QAM uplink table: I_MCS 10..13 map to I_TBS + 5 instead of + 6", "o_tbs_from_idx((int)m + (m < 10 ? 5 : 6), L); }", "o_tbs_from_idx((int)m + 5, L); }"),
    ("256QAM uplink table: row 32A not used for I_MCS 26", "      else if (m == 26) g->tbs = (L > 0 && L < 111) ? lsn_tbs_table_32A[L - 1] : 0;", "      else if (m == 26) g->tbs = o_tbs_from_idx((int)m + 6, L);"),
    ("MIMO: format 2 with one block and pinfo 0 is spatial multiplexing", "case O_FMT2: g->tx_scheme = (g->nof_tb == 1 && d->pinfo == 0) ? O_TX_DIVERSITY : O_TX_SPATIALMUX; break;", "case O_FMT2: g->tx_scheme = O_TX_SPATIALMUX; break;"),
    ("MIMO: two-block precoding information 2 is accepted", "      if (d->pinfo >= 2) return 2;", "      if (d->pinfo >= 3) return 2;"),
    ("common RNTIs, format 1A: the 2-PRB column is taken for n_prb1a = 3", "      uint32_t np = d->t2_nprb1a_is2 ? 2 : 3;", "      uint32_t np = 2;"),
]

CHILD = r'''
import json, os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests")]
import lsn_testlib
lsn_testlib.ORACLE_SO = %(so)r
lsn_testlib._ensure = lambda so, d: so
import ref_grants as G
fix = json.load(open(os.path.join(%(root)r, "tests", "golden", "grants_ref.json")))
o, ol = G.Oracle(), lsn_testlib.oracle()
ul = G.digest([G.normalise_ul(o.ul(*a), a[6]) for i, a in enumerate(G.ul_sweep()) if i %% G.SUITE_STRIDE == 0]) != fix["ul"]["suite_digest"]
mimo = G.digest([G.oracle_mimo(ol, *a) for a in G.mimo_sweep()]) != fix["mimo"]["digest"]
tbc = G.digest([G.oracle_tb_common(ol, *a) for a in G.tb_common_sweep()]) != fix["tb_common"]["digest"]
print(json.dumps([ul, mimo, tbc]))
'''


def main():
    src = open(os.path.join(ROOT, "oracle", "o_dci.c")).read()
    lines = ["one-token mutations of the oracle's grant conversions (o_dci.c) against the committed answers of the reference's ul_sniffer_pusch.c / dl_sniffer_pdsch.c", ""]
    for k, (what, old, new) in enumerate(MUTATIONS):
        assert src.count(old) == 1, (k, what, src.count(old))
        with tempfile.TemporaryDirectory() as tmp:
            shutil.copytree(os.path.join(ROOT, "oracle"), os.path.join(tmp, "oracle"), ignore=shutil.ignore_patterns("_build", "_ref"))
            shutil.copytree(os.path.join(ROOT, "spec"), os.path.join(tmp, "spec"))
            open(os.path.join(tmp, "oracle", "o_dci.c"), "w").write(src.replace(old, new))
            subprocess.check_call(["make", "-C", os.path.join(tmp, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            so = os.path.join(tmp, "oracle", "_build", "liblsn_oracle.so")
            ul, mimo, tbc = json.loads(subprocess.check_output([sys.executable, "-c", CHILD % dict(root=ROOT, so=so)], stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1])
        seen = [n for n, f in (("uplink sweep", ul), ("MIMO sweep", mimo), ("common-RNTI sweep", tbc)) if f]
        line = "%2d  %-80s %s" % (k, what, ("noticed by the " + ", ".join(seen)) if seen else "<-- NOT NOTICED")
        print(line, flush=True)
        lines.append(line)
    open(os.path.join(ROOT, "profiles", "r05_grant_pin_mutations.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
