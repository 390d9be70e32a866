#!/usr/bin/env python3
"""How far is the production oracle (windowed 10-bit max-log-MAP with 3/4 extrinsic scaling, u8 Viterbi - the numerics the HIP kernels
reproduce bit for bit) from an unwindowed, unclipped text-book decoder?  (the first number that says anything about the
distance to srsran_ue_dl_decode_pdsch, /root/reference/src/src/DL_Sniffer_PDSCH.cc:997, whose sources are absent.)

For every SNR point the same synthetic subframes go through two oracle workers: the production restatement and the second-opinion decoders of
oracle/o_second.c (full-trellis turbo on 16-bit soft values, float tail-biting Viterbi).  Counted per point: MAC-LTE records of each worker,
records only one of them produced (a CRC verdict that differs), records at the same (tti, rnti, direction) whose bytes differ (both passed a
24-bit CRC with different content - should never happen), and accepted DCIs.  CPU only.

  python tools/second_opinion_sweep.py [--nsf 60] [--config cfg3] [--snr 6 9 12 15 18 21 24 30] [--out profiles/r02_second_opinion.txt]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import ctypes as C  # noqa: E402

from lsn_testlib import OracleWorker, parse_pcap, scenario  # noqa: E402
from parity import gen_subframes  # noqa: E402


def run(sc, tti0, iq, turbo, viterbi):
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"], sc["phich_ng_x6"])
    ow.lib.o_worker_set_second_opinion.argtypes = [C.c_void_p, C.c_int, C.c_int]
    ow.lib.o_worker_set_second_opinion(ow.h, turbo, viterbi)
    ndci = 0
    for i in range(iq.shape[0]):
        ow.work(iq[i], tti0 + i, update_meta=1 if i % 100 == 0 else 0)
        ndci += len(ow.accepted())
    recs = parse_pcap(ow.pcap_bytes())
    return {(r["sfn"], r["sf"], r["rnti"], r["direction"], len(r["pdu"]), k): r["pdu"] for k, r in enumerate(recs)}, recs, ndci


def compare(sc, nsf):
    from collections import Counter
    tti0, iq, truth = gen_subframes(sc, nsf)
    sent = set()
    for i, pdus in enumerate(truth):
        for p in pdus:
            if not p["is_ul"]:
                sent.add((((tti0 + i) // 10) % 1024, (tti0 + i) % 10, p["payload"]))
    _, a, da = run(sc, tti0, iq, 0, 0)
    _, b, db = run(sc, tti0, iq, 1, 1)
    ca = Counter((r["sfn"], r["sf"], r["rnti"], r["direction"], r["pdu"]) for r in a)
    cb = Counter((r["sfn"], r["sf"], r["rnti"], r["direction"], r["pdu"]) for r in b)
    only_a, only_b = sum((ca - cb).values()), sum((cb - ca).values())
    # a record whose bytes the synthetic eNB never sent in that subframe passed a 24-bit CRC by accident (or SI / paging / RAR content the
    # transmitter does not report as a PDU: those carry RNTI constants and are skipped)
    false_a = sum(1 for r in a if r["rnti_type"] == 3 and (r["sfn"], r["sf"], r["pdu"]) not in sent)
    false_b = sum(1 for r in b if r["rnti_type"] == 3 and (r["sfn"], r["sf"], r["pdu"]) not in sent)
    return dict(prod=len(a), second=len(b), only_prod=only_a, only_second=only_b, false_prod=false_a, false_second=false_b, dci_prod=da, dci_second=db)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nsf", type=int, default=60)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--snr", type=float, nargs="*", default=[6, 9, 12, 15, 18, 21, 24, 30])
    ap.add_argument("--seed", type=int, default=91)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    lines = ["# production oracle vs second-opinion decoders (oracle/o_second.c): %s, %d subframes per SNR point, seed %d" % (args.config, args.nsf, args.seed),
             "# prod / second = MAC-LTE PDUs written (CRC ok); only_* = records (tti, rnti, bytes) written by one worker only; false_* = C-RNTI records whose bytes the",
             "# synthetic eNB did not send in that subframe; dci_* = DCIs accepted by the blind search",
             "%8s %9s %9s %10s %12s %11s %13s %9s %10s" % ("snr_db", "prod", "second", "only_prod", "only_second", "false_prod", "false_second", "dci_prod", "dci_second")]
    t0 = time.time()
    for snr in args.snr:
        r = compare(scenario(args.config, seed=args.seed, snr_db=float(snr)), args.nsf)
        lines.append("%8.1f %9d %9d %10d %12d %11d %13d %9d %10d" % (snr, r["prod"], r["second"], r["only_prod"], r["only_second"], r["false_prod"], r["false_second"], r["dci_prod"], r["dci_second"]))
        print(lines[-1], flush=True)
    lines.append("# %.0f s" % (time.time() - t0))
    if args.out:
        open(args.out, "w").write("\n".join(lines) + "\n")
    return lines


if __name__ == "__main__":
    main()
