#!/usr/bin/env python3
"""tests/golden/make_ul_decode_fixture.py -> tests/golden/ul_decode_ref.json

What the REFERENCE'S OWN uplink decode control flow (oracle/_ref/libref_falcon_ul_decode.so: PUSCH_Decoder::decode / decode_run of UL_Sniffer_PUSCH.cc compiled from
/root/reference on top of its own MCSTracking, oracle/Makefile.ref) does with the scripted lives of tests/ref_ul_decode.py under a scripted uplink decoder, as
digests; plus the trial order it shows for every (MCS index, tracked modulation, 256QAM-table allocation) when every attempt fails / the k-th passes."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]
import ref_ul_decode as U  # noqa: E402


def main():
    ref, orc = U.Reference(), U.Oracle()
    out = {"made_by": "tests/golden/make_ul_decode_fixture.py", "reference_sources": U.REF_SOURCES, "reference_sources_sha256": U.reference_sources_sha256(),
           "what_is_compared": "per subframe: every srsran_chest_ul_estimate_pusch + srsran_pusch_decode pair as configured (tti, rnti, L_prb, first PRB, MCS index, "
                               "modulation the decoder runs with, block size, HARQ-ACK bits, CSI request with report type and number of sub-bands, betaOffset indices, RI bits) "
                               "with the scripted verdict and SNR, every block handed to write_ul_crnti (rnti, length, FNV-1a of the bytes); after every ageing pass the "
                               "population of the uplink tracking database; at the end the tracked maximum modulation of every UE",
           "lives": {}, "oracle_equal_when_made": {}}
    for life in U.LIVES:
        a, b = U.run(ref, life), U.run(orc, life)
        out["lives"][life[0]] = dict(U.facts(a), digest=U.digest(a))
        out["oracle_equal_when_made"][life[0]] = a == b
    out["trial_table"] = U.trial_table(ref)
    json.dump(out, open(os.path.join(HERE, "ul_decode_ref.json"), "w"), indent=1)
    print(json.dumps(out["oracle_equal_when_made"]), len(out["trial_table"]))


if __name__ == "__main__":
    main()
