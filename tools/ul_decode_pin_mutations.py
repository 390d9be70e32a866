#!/usr/bin/env python3
"""tools/ul_decode_pin_mutations.py - does the pin of the uplink decode control flow on the reference's own code (tests/test_ref_ul_decode.py, tests/golden/ul_decode_ref.json)
notice a wrong restatement?  One-token mutations of the ORACLE's decode_pusch and of what it feeds (oracle/o_worker.c) are built in a scratch copy of oracle/, the suite's
five lives are replayed under the scripted uplink decoder and their digests compared with the committed ones of PUSCH_Decoder::decode.
-> profiles/r06_ul_decode_pin_mutations.txt"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MUTATIONS = [
    ("a grant without a size in the 256QAM table is tried all the same", "      if (m->g.tbs == 0 || m->g256.tbs == 0) valid = 0;", "      if (m->g.tbs == 0) valid = 0;"),
    ("PRB counts that are no product of 2, 3, 5 are tried", "      if (!o_ul_valid_prb(m->g.L_prb) || m->g.L_prb > 100) valid = 0;", "      if (m->g.L_prb > 100) valid = 0;"),
    ("RAR entries are tested like DCI 0 entries", "    if (!m->is_rar) {\n      if (m->rnti == 0) valid = 0;", "    if (1) {\n      if (m->rnti == 0) valid = 0;"),
    ("RNTI 0 is decoded", "    if (!valid || m->rnti == 0) continue;", "    if (!valid) continue;"),
    ("a 256QAM-table allocation of 110 PRB counts as usable", "    int ok256 = m->g256.L_prb < 110 && m->g256.L_prb > 0;", "    int ok256 = m->g256.L_prb <= 110 && m->g256.L_prb > 0;"),
    ("MCS 21-28, 16QAM maximum: tried with the grant's own modulation", "      if (mod == 2) { mem_mod = 2; crc = pusch_attempt(w, m, &m->g, 4, tti); LEARN(2); }", "      if (mod == 2) { mem_mod = 2; crc = pusch_attempt(w, m, &m->g, qm_base, tti); LEARN(2); }"),
    ("MCS 21-28, 64QAM maximum: capped at 16QAM", "      else if (mod == 3) { mem_mod = 3; crc = pusch_attempt(w, m, &m->g, qm_base, tti); LEARN(3); }", "      else if (mod == 3) { mem_mod = 3; crc = pusch_attempt(w, m, &m->g, 4, tti); LEARN(3); }"),
    ("MCS 21-28, unknown: 64QAM is tried first", "        crc = pusch_attempt(w, m, &m->g, 4, tti); LEARN(2);\n        if (!crc) {\n          crc = pusch_attempt(w, m, &m->g, qm_base, tti); LEARN(3);", "        crc = pusch_attempt(w, m, &m->g, qm_base, tti); LEARN(3);\n        if (!crc) {\n          crc = pusch_attempt(w, m, &m->g, 4, tti); LEARN(2);"),
    ("MCS 21-28, unknown: the 256QAM table is not tried", "          if (!crc && ok256) { crc = pusch_attempt(w, m, &m->g256, m->g256.mod, tti); LEARN(4); }", "          if (0) { crc = pusch_attempt(w, m, &m->g256, m->g256.mod, tti); LEARN(4); }"),
    ("MCS 21-28, unknown: a 16QAM success teaches 64QAM", "        crc = pusch_attempt(w, m, &m->g, 4, tti); LEARN(2);\n        if (!crc) {", "        crc = pusch_attempt(w, m, &m->g, 4, tti); LEARN(3);\n        if (!crc) {"),
    ("MCS 29 is decoded like MCS 28", "    if (mcs > 20 && mcs < 29) {\n      if (mod == 2) { mem_mod = 2;", "    if (mcs > 20 && mcs < 30) {\n      if (mod == 2) { mem_mod = 2;"),
    ("MCS 20 belongs to the upper rule", "    if (mcs > 20 && mcs < 29) {\n      if (mod == 2) { mem_mod = 2;", "    if (mcs >= 20 && mcs < 29) {\n      if (mod == 2) { mem_mod = 2;"),
    ("MCS <= 20, 256QAM maximum: the 64QAM-table grant is tried", "      else if (mod == 4) { if (ok256) { crc = pusch_attempt(w, m, &m->g256, m->g256.mod, tti); if (crc && mcs > 0) ulmod_update(w, m->rnti, 4); } }", "      else if (mod == 4) { crc = pusch_attempt(w, m, &m->g, qm_base > 4 ? 4 : qm_base, tti); }"),
    ("MCS <= 20, unknown: a 256QAM-table success at MCS 0 teaches the table", "        if (!crc && ok256) { crc = pusch_attempt(w, m, &m->g256, m->g256.mod, tti); if (crc && mcs > 0) ulmod_update(w, m->rnti, 4); }", "        if (!crc && ok256) { crc = pusch_attempt(w, m, &m->g256, m->g256.mod, tti); if (crc) ulmod_update(w, m->rnti, 4); }"),
    ("MCS <= 20, unknown: the 256QAM table is also tried after a success", "        if (!crc && ok256) { crc = pusch_attempt(w, m, &m->g256, m->g256.mod, tti); if (crc && mcs > 0) ulmod_update(w, m->rnti, 4); }", "        if (ok256) { crc = pusch_attempt(w, m, &m->g256, m->g256.mod, tti); if (crc && mcs > 0) ulmod_update(w, m->rnti, 4); }"),
    ("a success above MCS 20 teaches although the maximum was known", "#define LEARN(newmod) do { if (crc && mcs > 20 && mem_mod == 1) ulmod_update(w, m->rnti, (newmod)); } while (0)", "#define LEARN(newmod) do { if (crc && mcs > 20) ulmod_update(w, m->rnti, (newmod)); } while (0)"),
    ("the first success of an unknown RNTI sets its modulation at once", "  if (w->ulmod[rnti]) w->ulmod[rnti] = (uint8_t)mod;\n  else ul_add(w, rnti, 1);", "  if (w->ulmod[rnti]) w->ulmod[rnti] = (uint8_t)mod;\n  else ul_add(w, rnti, mod);"),
    ("the statistics ignore the SNR gate", "    if (w->last_ul_snr >= 1.0f) ul_statistic(w, m->rnti, crc, mem_mod);", "    ul_statistic(w, m->rnti, crc, mem_mod);"),
    ("the SNR gate is at 0 dB", "    if (w->last_ul_snr >= 1.0f) ul_statistic(w, m->rnti, crc, mem_mod);", "    if (w->last_ul_snr >= 0.0f) ul_statistic(w, m->rnti, crc, mem_mod);"),
    ("the statistics add an RNTI with unknown modulation whatever was tried", "  ul_add(w, rnti, mem_mod);\n  w->ul_active[rnti]++;", "  ul_add(w, rnti, 1);\n  w->ul_active[rnti]++;"),
    ("ageing keeps entries that were never counted", "    if (cur_interval > w->mcs_interval || w->ul_active[r] == 0) { w->ulmod[r] = 0; w->ulmod_count--; }", "    if (cur_interval > w->mcs_interval) { w->ulmod[r] = 0; w->ulmod_count--; }"),
    ("ageing drops an entry that is exactly `interval` seconds old", "    if (cur_interval > w->mcs_interval || w->ul_active[r] == 0) { w->ulmod[r] = 0; w->ulmod_count--; }", "    if (cur_interval >= w->mcs_interval || w->ul_active[r] == 0) { w->ulmod[r] = 0; w->ulmod_count--; }"),
    ("a look-up does not refresh the entry's time stamp", "  w->ul_time[rnti] = w->sf_count; /* :52-53 */", "  /* :52-53 */"),
    ("the database is full at 251 entries", "  if (!w->ulmod[rnti]) return w->ulmod_count < 250 ? 1 : 5;", "  if (!w->ulmod[rnti]) return w->ulmod_count < 251 ? 1 : 5;"),
    ("the UCI layout uses the default configuration for every RNTI", "  const o_ue_cfg_t uc = ue_cfg_get(w, m->rnti);\n  o_uci_t uci", "  const o_ue_cfg_t uc = w->default_cfg;\n  o_uci_t uci"),
    ("an aperiodic report carries no RI bit", "m->cqi_req ? 1u : 0u,\n                 uc.i_offset_ack + 1u", "0u,\n                 uc.i_offset_ack + 1u"),
]

CHILD = r'''
import json, os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests")]
import lsn_testlib
lsn_testlib.ORACLE_SO = %(so)r
lsn_testlib._ensure = lambda so, d: so
import ref_ul_decode as U
fix = json.load(open(os.path.join(%(root)r, "tests", "golden", "ul_decode_ref.json")))
bad = [l[0] for l in U.LIVES if U.digest(U.run(U.Oracle(), l)) != fix["lives"][l[0]]["digest"]]
print(json.dumps(bad))
'''


def main():
    src = open(os.path.join(ROOT, "oracle", "o_worker.c")).read()
    lines = ["one-token mutations of the oracle's uplink decode control flow (o_worker.c: decode_pusch_list, pusch_attempt, ulmod_*, ul_statistic, ul_update_database) against the committed",
             "behaviour of the reference's PUSCH_Decoder::decode under the scripted uplink decoder: tests/golden/ul_decode_ref.json", ""]
    missed = 0
    for k, (what, old, new) in enumerate(MUTATIONS):
        assert src.count(old) == 1, (k, what, src.count(old))
        with tempfile.TemporaryDirectory() as tmp:
            shutil.copytree(os.path.join(ROOT, "oracle"), os.path.join(tmp, "oracle"), ignore=shutil.ignore_patterns("_build", "_ref"))
            shutil.copytree(os.path.join(ROOT, "spec"), os.path.join(tmp, "spec"))
            open(os.path.join(tmp, "oracle", "o_worker.c"), "w").write(src.replace(old, new))
            subprocess.check_call(["make", "-C", os.path.join(tmp, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            so = os.path.join(tmp, "oracle", "_build", "liblsn_oracle.so")
            bad = json.loads(subprocess.check_output([sys.executable, "-c", CHILD % dict(root=ROOT, so=so)], stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1])
        missed += not bad
        line = "%2d  %-100s %s" % (k, what, ("noticed by %d of 5 lives" % len(bad)) if bad else "<-- NOT NOTICED")
        print(line, flush=True)
        lines.append(line)
    lines += ["", "%d of %d noticed" % (len(MUTATIONS) - missed, len(MUTATIONS))]
    open(os.path.join(ROOT, "profiles", "r06_ul_decode_pin_mutations.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
