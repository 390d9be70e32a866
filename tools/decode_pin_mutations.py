#!/usr/bin/env python3
"""tools/decode_pin_mutations.py - does the pin of the downlink decode control flow on the reference's own code (tests/test_ref_decode.py, tests/golden/decode_ref.json) notice a
wrong restatement?  One-token mutations of the ORACLE's decode_dl_mode and of what it feeds (oracle/o_worker.c) are built in a scratch copy of oracle/, the suite's eight lives are
replayed under the scripted decoder and their digests compared with the committed ones of PDSCH_Decoder::decode_dl_mode.  -> profiles/r06_decode_pin_mutations.txt"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MUTATIONS = [
    ("the gate lets a grant without a size through", "int gate = (cur->tb[0].tbs > 0 && e->dci_rnti > 0", "int gate = (cur->tb[0].tbs >= 0 && e->dci_rnti > 0"),
    ("a DCI whose conversion failed is decoded all the same", "&& e->dci_rnti > 0 && !(w->cfg.nof_rx == 1 && two_tb))", "&& !(w->cfg.nof_rx == 1 && two_tb))"),
    ("two-block grants are decoded on one antenna", "&& !(w->cfg.nof_rx == 1 && two_tb)) || e->rnti == O_PRNTI;", ") || e->rnti == O_PRNTI;"),
    ("paging grants pass the gate like everything else", "&& !(w->cfg.nof_rx == 1 && two_tb)) || e->rnti == O_PRNTI;", "&& !(w->cfg.nof_rx == 1 && two_tb));"),
    ("SIB1 in format 1C keeps redundancy version -1", "if (e->dci.tb[0].rv < 0 && e->rnti == O_SIRNTI) cur->tb[0].rv = 0;", "if (0) cur->tb[0].rv = 0;"),
    ("unknown table: the 256QAM attempt also runs after a partial pass", "      if (!crc[0] && !crc[1] && mimo_ret == 0) {", "      if (!(crc[0] && crc[1]) && mimo_ret == 0) {"),
    ("unknown table: a passed 64QAM-table block does not teach the table", "mcs_idx < 29 && e->format > O_FMT1A) mcs_update(w, e->rnti, O_TABLE_64QAM);", "mcs_idx < 29 && e->format > O_FMT1A && 0) mcs_update(w, e->rnti, O_TABLE_64QAM);"),
    ("unknown table: format 1A teaches the table", "mcs_idx < 29 && e->format > O_FMT1A) mcs_update(w, e->rnti, O_TABLE_64QAM);", "mcs_idx < 29 && e->format >= O_FMT1A) mcs_update(w, e->rnti, O_TABLE_64QAM);"),
    ("unknown table: I_MCS 28 of the 256QAM attempt teaches the table", "mcs_idx < 28 && e->format > O_FMT1A) mcs_update(w, e->rnti, O_TABLE_256QAM);", "mcs_idx < 29 && e->format > O_FMT1A) mcs_update(w, e->rnti, O_TABLE_256QAM);"),
    ("unknown table: I_MCS 0 teaches the table", "if (e->dci.tb[tb].mcs_idx > 0 && e->dci.tb[tb].mcs_idx < 29 && e->format > O_FMT1A)", "if (e->dci.tb[tb].mcs_idx < 29 && e->format > O_FMT1A)"),
    ("unknown table: the 256QAM attempt's verdict of a disabled block overwrites the first one", "            if (cur256->tb[tb].enabled) crc[tb] = crc2[tb];", "            crc[tb] = crc2[tb];"),
    ("unknown table: connection setups are learnt from logical channel 0 only (the oracle before round 6)", "if (name[0] == 'C') learn_conn_setup(w, w->payload + tb * 16384, len, e->rnti, 1);", "if (name[0] == 'C') learn_conn_setup(w, w->payload + tb * 16384, len, e->rnti, 0);"),
    ("known table: connection setups are learnt from every SDU", "if (name[0] == 'C') learn_conn_setup(w, w->payload + tb * 16384, len, e->rnti, 0);\n          }\n        }\n      }\n    } else {", "if (name[0] == 'C') learn_conn_setup(w, w->payload + tb * 16384, len, e->rnti, 1);\n          }\n        }\n      }\n    } else {"),
    ("a random-access response does not reach the RNTI manager", "    o_rntiman_activate_and_refresh(w->rm, t_crnti, 0, O_ACT_RAR);\n  }\n}", "  }\n}"),
    ("a random-access response does not reset the UE's table", "  w->mcs[crnti].has_rar = 1;\n  w->mcs[crnti].table = O_TABLE_UNKNOWN;", "  w->mcs[crnti].has_rar = 1;"),
    ("after a RAR three later messages are enough to fix the table", "      if (e->nof_msg_after_rar > 3) {", "      if (e->nof_msg_after_rar > 2) {"),
    ("the statistics count format 1A messages after a RAR", "  if (format > O_FMT1A && e->has_rar) e->nof_msg_after_rar++;", "  if (format >= O_FMT1A && e->has_rar) e->nof_msg_after_rar++;"),
    ("the statistics are skipped for a grant whose MIMO configuration failed", "      mcs_statistic(w, e->rnti, e->format, e->mcs_table, tb_en, crc, mimo_ret ? -mimo_ret : 0);", "      if (!mimo_ret) mcs_statistic(w, e->rnti, e->format, e->mcs_table, tb_en, crc, 0);"),
    ("ageing: a success rate of exactly 15 % resets the table", "< 0.15f && e->table != O_TABLE_UNKNOWN) {", "<= 0.15f && e->table != O_TABLE_UNKNOWN) {"),
    ("the first connection setup does not become the default configuration", "    if (!w->has_default_cfg) {\n      w->default_cfg = c;", "    if (0) {\n      w->default_cfg = c;"),
    ("HARQ: a block decoded 8 subframes ago is decoded again", "                gg.tb[i].enabled = 0; /* decoded 8 subframes ago", "                gg.tb[i].enabled = 1; /* decoded 8 subframes ago"),
    ("HARQ: the record of a retransmission keeps the old verdict", "harq_update(w, ent[i], (int)e->dci.pid, i, w->sfn, w->sf_idx, crc[i],", "harq_update(w, ent[i], (int)e->dci.pid, i, w->sfn, w->sf_idx, 0,"),
    ("HARQ: nine subframes also count as a retransmission", "  if (!(cur_tti - last_tti == 8 || cur_tti + 10240 - last_tti == 8)) return O_HARQ_NEW_TX;", "  if (!(cur_tti - last_tti == 8 || cur_tti - last_tti == 9 || cur_tti + 10240 - last_tti == 8)) return O_HARQ_NEW_TX;"),
    ("p_a of the decode call: always the default", "w->ul_mode ? -3.0f : ue_cfg_get(w, e->rnti).p_a, w->payload, w->payload + 16384, c2);", "w->ul_mode ? -3.0f : w->default_cfg.p_a, w->payload, w->payload + 16384, c2);"),
]

CHILD = r'''
import json, os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests")]
import lsn_testlib
lsn_testlib.ORACLE_SO = %(so)r
lsn_testlib._ensure = lambda so, d: so
import ref_decode as D
fix = json.load(open(os.path.join(%(root)r, "tests", "golden", "decode_ref.json")))
bad = [l[0] for l in D.LIVES if D.digest([r for r in D.run(D.Oracle(), l) if r[0] != "harq"]) != fix["lives"][l[0]]["digest"]]
print(json.dumps(bad))
'''


def main():
    src = open(os.path.join(ROOT, "oracle", "o_worker.c")).read()
    lines = ["one-token mutations of the oracle's downlink decode control flow (o_worker.c: decode_dl_mode, learn_conn_setup, unpack_rar, mcs_*, harq_*) against the committed behaviour of the",
             "reference's PDSCH_Decoder::decode_dl_mode under the scripted decoder: tests/golden/decode_ref.json", ""]
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
        line = "%2d  %-100s %s" % (k, what, ("noticed by %d of 8 lives" % len(bad)) if bad else "<-- NOT NOTICED")
        print(line, flush=True)
        lines.append(line)
    lines += ["", "%d of %d noticed" % (len(MUTATIONS) - missed, len(MUTATIONS))]
    open(os.path.join(ROOT, "profiles", "r06_decode_pin_mutations.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
