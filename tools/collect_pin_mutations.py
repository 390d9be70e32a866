#!/usr/bin/env python3
"""tools/collect_pin_mutations.py - does the pin of the DCI collection on the reference's own code (tests/test_ref_collect.py, tests/golden/collect_ref.json) notice a
wrong restatement?  One-token mutations of the ORACLE (oracle/o_worker.c: add_candidate, oracle/o_dci.c: the downlink grant conversion, o_rar_parse) are built in a
scratch copy of oracle/, the suite's ten lives and the RAR sweep are replayed and their digests compared with the committed ones of DCICollection.cc / falcon_dci.c /
dl_sniffer_pdsch.c / collect_glue.cc.  -> profiles/r06_collect_pin_mutations.txt"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MUTATIONS = [
    ("o_worker.c", "format 1A of a C-RNTI asks the tracking database for its table", "O_RNTI_ISRAR(c->rnti) || fmt == O_FMT1A)", "O_RNTI_ISRAR(c->rnti))"),
    ("o_worker.c", "UL_MODE asks the tracking database", "table = w->ul_mode ? O_TABLE_UNKNOWN : mcs_find(w, c->rnti);", "table = mcs_find(w, c->rnti);"),
    ("o_worker.c", "a failed 256QAM uplink conversion keeps the DCI's RNTI", "if (ok && o_ra_ul_dci_to_grant_256(cell, &u->dci, &u->g256)) { ok = 0;", "if (ok && o_ra_ul_dci_to_grant_256(cell, &u->dci, &u->g256)) { ok = 1;"),
    ("o_worker.c", "a failed 64QAM-table conversion keeps the DCI's RNTI", "if (o_ra_dl_dci_to_grant(cell, w->sf_idx, w->cfi, 0, &e->dci, &e->g64)) e->dci_rnti = 0;", "if (o_ra_dl_dci_to_grant(cell, w->sf_idx, w->cfi, 0, &e->dci, &e->g64)) e->dci_rnti = e->rnti;"),
    ("o_worker.c", "an unknown-table entry gets the 64QAM-table grant only", "    if (table == O_TABLE_256QAM || table >= O_TABLE_UNKNOWN) {", "    if (table == O_TABLE_256QAM) {"),
    ("o_worker.c", "the RB map of a 256QAM-table entry is read from the (empty) 64QAM-table grant", "const o_pdsch_grant_t* gm = e->has64 ? &e->g64 : &e->g256;", "const o_pdsch_grant_t* gm = &e->g64;"),
    ("o_worker.c", "the downlink RB map is read from slot 1", "    if (gm->prb_idx[0][rb]) {", "    if (gm->prb_idx[1][rb]) {"),
    ("o_worker.c", "an uplink collision needs a different RNTI", "        if (w->rb_map_ul[u->g.n_prb + i] != 0) w->ul_collision = 1;", "        if (w->rb_map_ul[u->g.n_prb + i] != 0 && w->rb_map_ul[u->g.n_prb + i] != c->rnti) w->ul_collision = 1;"),
    ("o_worker.c", "the HARQ size is taken for I_MCS 28 too", "if (e->g64.tb[i].enabled && e->g64.tb[i].mcs_idx > 28) {", "if (e->g64.tb[i].enabled && e->g64.tb[i].mcs_idx > 27) {"),
    ("o_worker.c", "the HARQ size of the FIRST entity of the RNTI... of process 0", "tbs = w->harq[k].tb[e->dci.pid & 7][i].grant.tbs;", "tbs = w->harq[k].tb[0][i].grant.tbs;"),
    ("o_worker.c", "a transport block without bits stays enabled (64QAM-table grant)", "    if (e->g64.tb[i].nof_bits <= 0) e->g64.tb[i].enabled = 0;", "    if (e->g64.tb[i].nof_bits < 0) e->g64.tb[i].enabled = 0;"),
    ("o_worker.c", "RAR grant with the hopping flag: +N/4 instead of -N/4", "d.hop_type = d.freq_hop_fl ? 1 : -1;", "d.hop_type = d.freq_hop_fl ? 0 : -1;"),
    ("o_dci.c", "type 0: RBG 0 in the least significant bit", "        if (d->rbg_bitmask & (1u << (nb - i - 1)))", "        if (d->rbg_bitmask & (1u << i))"),
    ("o_dci.c", "type 1: the shift is not applied", "      uint32_t shift = d->t1_shift ? sub - n1 : 0;", "      uint32_t shift = 0;"),
    ("o_dci.c", "type 1: subset size of the last subset one PRB short", "        sub = ((n - 1) / (P * P)) * P + ((n - 1) % P) + 1;", "        sub = ((n - 1) / (P * P)) * P + ((n - 1) % P);"),
    ("o_dci.c", "distributed VRBs: the odd slot is not moved by half a unit", "  uint32_t o = (e % Nt + Nt / 2) % Nt + Nt * blk; /* odd slot */", "  uint32_t o = (e % Nt) % Nt + Nt * blk; /* odd slot */"),
    ("o_dci.c", "distributed VRBs: nulls in the last N_null rows instead of N_null / 2", "    e = p1 - Nrow + Nnull / 2;", "    e = p1 - Nrow + Nnull;"),
    ("o_dci.c", "distributed VRBs: second-gap table entry of 50-63 PRB", "  return nprb <= 63 ? 9 : 16;", "  return nprb <= 63 ? 12 : 16;"),
    ("o_dci.c", "format 1C: step 2 for every bandwidth", "static uint32_t n_step(uint32_t nprb) { return nprb < 50 ? 2 : 4; }", "static uint32_t n_step(uint32_t nprb) { return 2; }"),
    ("o_dci.c", "RE count: the control region of a narrow cell is not one symbol longer", "  uint32_t n = 0, l0 = cfi + (cell->nof_prb <= 10 ? 1u : 0u);", "  uint32_t n = 0, l0 = cfi;"),
    ("o_dci.c", "RE count: PBCH symbols of subframe 0 counted as data", "    if (sf_idx == 0 && l >= nsl && l <= nsl + 3) return 0;", "    if (sf_idx == 0 && l >= nsl && l <= nsl + 2) return 0;"),
    ("o_dci.c", "RE count: one-port cells lose the REs of a second port", "    if (cell->nof_ports >= 2) {\n      if ((k % 3) == (cell->id % 3)) return 0;", "    if (cell->nof_ports >= 1) {\n      if ((k % 3) == (cell->id % 3)) return 0;"),
    ("o_dci.c", "the second block of a two-block format is enabled whatever its MCS / RV", "    int en = !(d->tb[i].mcs_idx == 0 && d->tb[i].rv == 1);", "    int en = 1;"),
    ("o_dci.c", "format 1C of a paging grant keeps its redundancy version", "  if (d->format == O_FMT1C && (O_RNTI_ISRAR(d->rnti) || d->rnti == O_PRNTI))", "  if (d->format == O_FMT1C && (O_RNTI_ISRAR(d->rnti)))"),
]

CHILD = r'''
import json, os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests")]
import lsn_testlib
lsn_testlib.ORACLE_SO = %(so)r
lsn_testlib._ensure = lambda so, d: so
import ref_collect as R
fix = json.load(open(os.path.join(%(root)r, "tests", "golden", "collect_ref.json")))
bad = [l[0] for l in R.LIVES if R.digest(R.Oracle().run(l)) != fix["lives"][l[0]]["digest"]]
o = lsn_testlib.oracle()
rar = R.digest_rows([tuple(v & 0xFFFFFFFF for v in R.rar_oracle(o, *a)) for a in R.rar_sweep()]) != fix["rar"]["digest"]
print(json.dumps([bad, rar]))
'''


def main():
    srcs = {f: open(os.path.join(ROOT, "oracle", f)).read() for f in ("o_worker.c", "o_dci.c")}
    lines = ["one-token mutations of the oracle's DCI collection (o_worker.c: add_candidate, o_rar_parse; o_dci.c: downlink grant conversion) against the committed answers of the",
             "reference's DCICollection.cc / falcon_dci.c / dl_sniffer_pdsch.c (+ the standard-derived allocation functions of collect_glue.cc): tests/golden/collect_ref.json", ""]
    missed = 0
    for k, (f, what, old, new) in enumerate(MUTATIONS):
        assert srcs[f].count(old) == 1, (k, what, srcs[f].count(old))
        with tempfile.TemporaryDirectory() as tmp:
            shutil.copytree(os.path.join(ROOT, "oracle"), os.path.join(tmp, "oracle"), ignore=shutil.ignore_patterns("_build", "_ref"))
            shutil.copytree(os.path.join(ROOT, "spec"), os.path.join(tmp, "spec"))
            open(os.path.join(tmp, "oracle", f), "w").write(srcs[f].replace(old, new))
            subprocess.check_call(["make", "-C", os.path.join(tmp, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            so = os.path.join(tmp, "oracle", "_build", "liblsn_oracle.so")
            bad, rar = json.loads(subprocess.check_output([sys.executable, "-c", CHILD % dict(root=ROOT, so=so)], stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1])
        seen = (["%d of 10 lives" % len(bad)] if bad else []) + (["the RAR sweep"] if rar else [])
        missed += not seen
        line = "%2d  %-11s %-88s %s" % (k, f, what, ("noticed by " + ", ".join(seen)) if seen else "<-- NOT NOTICED")
        print(line, flush=True)
        lines.append(line)
    lines += ["", "%d of %d noticed" % (len(MUTATIONS) - missed, len(MUTATIONS))]
    open(os.path.join(ROOT, "profiles", "r06_collect_pin_mutations.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
