#!/usr/bin/env python3
"""tests/golden/make_grant_fixture.py -> tests/golden/grants_ref.json

Answers of the REFERENCE'S OWN grant conversions (oracle/_ref/libref_falcon_grant.so: ul_sniffer_pusch.c, dl_sniffer_pdsch.c compiled from /root/reference by
oracle/Makefile.ref) to the sweeps of tests/ref_grants.py, as digests: the thinned uplink sweep the suite repeats, the MIMO and common-RNTI sweeps, and - once,
here - the full uplink sweep of 2.5 million grants next to the oracle."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]
import ref_grants as G  # noqa: E402
from lsn_testlib import oracle  # noqa: E402


def main():
    r, o, ol = G.Reference(), G.Oracle(), oracle()
    full_n, full_bad, thin, accepted, hop1 = 0, 0, [], 0, 0
    for i, a in enumerate(G.ul_sweep()):
        x = G.normalise_ul(r.ul(*a), a[6])
        y = G.normalise_ul(o.ul(*a), a[6])
        full_n += 1
        full_bad += x != y
        accepted += x is not None
        hop1 += bool(x and x[3] == 1)
        if i % G.SUITE_STRIDE == 0:
            thin.append(x)
    mimo = [r.mimo(*a) for a in G.mimo_sweep()]
    tbc = [r.tb_common(*a) for a in G.tb_common_sweep()]
    out = {"made_by": "tests/golden/make_grant_fixture.py", "reference_sources": G.REF_SOURCES, "reference_sources_sha256": G.reference_sources_sha256(),
           "ul": {"row": "(L_prb, first PRB slot 0, first PRB slot 1, hopping kind 0 / 1 / 2, modulation bits, tbs, rv, nof_re) or null when the reference refuses",
                  "suite_stride": G.SUITE_STRIDE, "suite_cases": len(thin), "suite_digest": G.digest(thin), "suite_accepted": sum(x is not None for x in thin),
                  "full_sweep": {"cases": full_n, "accepted_by_the_reference": accepted, "type_1_hopping_grants": hop1, "oracle_differs_in": full_bad}},
           "mimo": {"cases": len(mimo), "digest": G.digest(mimo), "ok": sum(m[0] == 0 for m in mimo), "by_error": [sum(m[0] == -k for m in mimo) for k in (1, 2, 3)]},
           "tb_common": {"cases": len(tbc), "digest": G.digest(tbc), "ok": sum(t[0] == 0 for t in tbc),
                         "sizes_seen": sorted({t[4] for t in tbc if t[0] == 0})},
           "oracle_equal_when_made": {"mimo": mimo == [G.oracle_mimo(ol, *a) for a in G.mimo_sweep()],
                                      "tb_common": tbc == [G.oracle_tb_common(ol, *a) for a in G.tb_common_sweep()], "ul_full_sweep": full_bad == 0}}
    json.dump(out, open(os.path.join(HERE, "grants_ref.json"), "w"), indent=1)
    print(json.dumps(out["ul"]["full_sweep"]), out["oracle_equal_when_made"])


if __name__ == "__main__":
    main()
