#!/usr/bin/env python3
"""Throughput of the CPU oracle's receiver on the fixed reference channels of TS 36.101 section 8.2.1.1.1 (FDD, single antenna port, 1 x 2, low correlation) over
the multipath fading channels of Annex B.2 (tools/txgen chan_model 1 / 2 / 3 = EPA / EVA / ETU), next to the PUBLISHED minimum requirement (the SNR at which a
conformant UE reaches 70 % / 30 % of the maximum throughput, Table 8.2.1.1.1-2).  This is the external yardstick the DSP restatement otherwise lacks (the
reference's srsRAN is absent): a receiver whose front end or decoders were wrong by more than the implementation margin of the requirement would miss it.

What is measured: 10 MHz, one C-RNTI scheduled in every subframe over the whole band (minus the SIB of subframe 5 in even frames), one fixed MCS -
  R.2  QPSK  1/3  = MCS  5 (TBS 4392 on 50 PRB)      R.3  16QAM 1/2 = MCS 14 (TBS 12960)      R.7  64QAM 3/4 = MCS 26 (TBS 30576)
- and the share of the sent transport blocks that the sniffer delivers (PDCCH search + PDSCH decode, first transmissions only: a passive receiver sends no HARQ
feedback, so there is NO retransmission gain here, while the requirement is stated WITH up to four HARQ transmissions - the figures below are therefore a
conservative reading: delivered share >= 70 % at the published SNR means the requirement is met without any help from HARQ).
Two properties of the REFERENCE's design (not of this restatement) bound what can be met, and the table shows both:
  * it does not search a subframe whose ESTIMATED SNR is 6 dB or less (DCISearch.cc:568-574; the estimate reads 1 - 3 dB high, tests/test_frontend_truth.py), and its
    search skips CCEs whose mean |LLR| is under 0.7 (falcon_pdcch.c:595-620) - both pinned on the compiled reference in oracle/_ref.  Every fade below that line is
    lost whatever the decoder could do, so the QPSK / 16QAM requirements at -1 ... 7 dB are out of its reach BY DESIGN; the column "DCI accepted" shows that share,
    "delivered of those" is the PDSCH chain on its own (equaliser, soft demodulation, rate matching, turbo decoder);
  * its channel estimator is configured with a 5-tap Gaussian smoothing over the pilots of a symbol + linear interpolation (SubframeWorker.cc:381-390): the 5 us delay
    spread of ETU turns by 2.8 rad between neighbouring pilots and is smoothed away - ETU is not decodable with these settings at any SNR.
The first 40 subframes of each run are left out (the sniffer has to see an RNTI a few times before it accepts it: RNTIManager histogram).

usage: tools/frc_36101.py [--seeds 8] [--subframes 240] [--out profiles/r06_frc_36101.txt]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from lsn_testlib import scenario  # noqa: E402
from parity import gen_subframes, run_oracle  # noqa: E402

# (name, mcs, channel model, Doppler, published SNR of Table 8.2.1.1.1-2, fraction of maximum throughput there, SNR points measured)
CASES = [
    ("R.2 QPSK 1/3", 5, 2, 5.0, -1.0, 0.70, (-3.0, -1.0, 1.0, 3.0)),      # test 1: EVA5
    ("R.2 QPSK 1/3", 5, 2, 5.0, None, None, (6.0, 9.0, 12.0)),             # ... and where the sniffer's PDCCH search starts to accept the RNTI at all (see note)
    ("R.2 QPSK 1/3", 5, 3, 70.0, -0.4, 0.70, (-0.4, 5.6, 11.6)),           # test 2: ETU70
    ("R.3 16QAM 1/2", 14, 2, 5.0, 6.7, 0.70, (4.7, 6.7, 8.7, 10.7)),       # test 6: EVA5
    ("R.3 16QAM 1/2", 14, 3, 70.0, 1.4, 0.30, (1.4, 3.4, 5.4)),            # test 7: ETU70, 30 %
    ("R.7 64QAM 3/4", 26, 2, 5.0, 17.7, 0.70, (15.7, 17.7, 19.7, 21.7)),   # test 11: EVA5
    ("R.7 64QAM 3/4", 26, 3, 70.0, 19.0, 0.70, (19.0, 21.0, 23.0)),        # test 12: ETU70
    ("R.7 64QAM 3/4", 26, 1, 5.0, None, None, (15.7, 17.7, 19.7)),         # EPA5: no requirement in this table; for the shape of the curve
]
NAMES = {1: "EPA", 2: "EVA", 3: "ETU"}


def measure(mcs, model, doppler, snr, seeds, nsf, skip=40):
    """-> (transport blocks sent, those whose DCI the search accepted, those delivered)"""
    sent = got = found = 0
    for s in range(seeds):
        sc = scenario("cfg1", seed=500 + s, nof_prb=50, nof_ports=1, nof_rx=2, snr_db=float(snr), cfo_hz=0.0, n_rnti=1, dl_min=1, dl_max=1, ul_min=0, ul_max=0,
                      cfi=2, mcs_min=mcs, mcs_max=mcs, chan_model=model, doppler_hz=float(doppler), timing_offset_samples=0.0)
        tti0, iq, truth = gen_subframes(sc, nsf)
        _, per_sf, recs = run_oracle(sc, tti0, iq, update_meta_period=0, taps=True)
        have = set((r["sfn"] * 10 + r["sf"], r["rnti"], r["pdu"]) for r in recs if r["direction"] == 1)  # MAC-LTE context: 1 = downlink
        for i, sf in enumerate(truth):
            if i < skip:
                continue
            for p in sf:
                if p["is_ul"] or not (0x000B <= p["rnti"] <= 0xFFF3):
                    continue
                sent += 1
                found += any(a[0] == p["rnti"] and a[3] == p["ncce"] for a in per_sf[i]["accepted"])
                got += (p["tti"] % 10240, p["rnti"], p["payload"]) in have
    return sent, found, got


def measure_harq(mcs, model, doppler, snr, seeds, nsf, skip=40):
    """the requirement's own procedure, emulated exactly for a passive receiver: EVERY transport block goes out four times, 8 subframes apart, with the redundancy
    versions 0 2 3 1 (txgen pct_harq = 100), the oracle combines them like the reference's HARQ.cc (harq_mode 1), and a block that is delivered at its k-th
    transmission counts as ONE block in k subframes (a UE would have acknowledged it there and the eNB would have used the remaining subframes for new data of the
    same statistics); an undelivered one as no block in four.  -> (blocks delivered, subframes used): their ratio is throughput / maximum throughput"""
    ok = used = 0
    for s in range(seeds):
        sc = scenario("cfg1", seed=500 + s, nof_prb=50, nof_ports=1, nof_rx=2, snr_db=float(snr), cfo_hz=0.0, n_rnti=1, dl_min=1, dl_max=1, ul_min=0, ul_max=0,
                      cfi=2, mcs_min=mcs, mcs_max=mcs, chan_model=model, doppler_hz=float(doppler), timing_offset_samples=0.0, pct_harq=100, sib_period=0)
        tti0, iq, truth = gen_subframes(sc, nsf)
        _, _, recs = run_oracle(sc, tti0, iq, update_meta_period=0, taps=False, harq_mode=1)
        have = set((r["sfn"] * 10 + r["sf"], r["rnti"], r["pdu"]) for r in recs if r["direction"] == 1)
        chains = {}
        for i, sf in enumerate(truth):
            for p in sf:
                if not p["is_ul"] and 0x000B <= p["rnti"] <= 0xFFF3:
                    chains.setdefault((p["rnti"], p["payload"]), []).append((i, p["tti"] % 10240))
        for (rnti, payload), tx in chains.items():
            if tx[0][0] < skip or tx[0][0] + 24 >= nsf:
                continue
            k = next((j for j, (_, t) in enumerate(tx) if (t, rnti, payload) in have), None)
            if k is None:
                used += len(tx)
            else:
                ok, used = ok + 1, used + k + 1
    return ok, used


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--subframes", type=int, default=240)
    ap.add_argument("--harq", action="store_true", help="the HARQ procedure of the requirement (four transmissions, soft combining) instead of first transmissions only")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if a.out is None:
        a.out = os.path.join(ROOT, "profiles", "r06_frc_36101_harq.txt" if a.harq else "r06_frc_36101.txt")
    lines = [__doc__.split("usage:")[0].rstrip(), "", ("MODE: " + measure_harq.__doc__) if a.harq else "MODE: first transmissions only", "",
             "%d seeds x %d subframes per point (first 40 of each left out); independent Rayleigh fading per seed" % (a.seeds, a.subframes), "",
             "%-15s %-8s %8s  %-26s  %s" % ("FRC", "channel", "SNR dB", "blocks / subframes used" if a.harq else "delivered / sent", "requirement (TS 36.101 Table 8.2.1.1.1-2)")]
    t0 = time.time()
    for name, mcs, model, dop, req_snr, req_frac, snrs in CASES:
        for snr in snrs:
            extra = ""
            if a.harq:
                got, sent = measure_harq(mcs, model, dop, snr, a.seeds, a.subframes)  # (blocks, subframes used)
            else:
                sent, found, got = measure(mcs, model, dop, snr, a.seeds, a.subframes)
                extra = "DCI accepted %5.1f %%, delivered of those %5.1f %%   " % (100.0 * found / max(sent, 1), 100.0 * got / max(found, 1))
            req = ""
            if req_snr is not None and abs(snr - req_snr) < 1e-6:
                ok = got >= req_frac * sent
                req = ">= %.0f %% at %.1f dB: %s" % (100 * req_frac, req_snr, ("met" if ok else "NOT met") + (" (HARQ as in the requirement)" if a.harq else " without HARQ"))
            ln = "%-15s %-8s %8.1f  %6d / %-6d = %5.1f %%   %s%s" % (name, "%s%d" % (NAMES[model], dop), snr, got, sent, 100.0 * got / max(sent, 1), extra, req)
            print(ln, flush=True)
            lines.append(ln)
    lines.append("")
    lines.append("wall %.0f s" % (time.time() - t0))
    with open(a.out, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
