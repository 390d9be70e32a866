#!/usr/bin/env python3
"""tests/golden/make_dci_search_fixture.py -> tests/golden/dci_search_ref.json

Runs the REFERENCE'S OWN blind DCI search (oracle/_ref/libref_falcon_search.so: DCISearch.cc, falcon_pdcch.c, MetaFormats.cc, RNTIManager.cc compiled
from /root/reference by oracle/Makefile.ref) next to the oracle worker on the streams of tests/ref_dci_search.py: CASES and writes down what the
reference decided - per subframe a digest of the accepted DCI list, the totals, the statistics, the final format split, the activation reasons - so
that tests/test_ref_dci_search.py can hold the oracle and the product's host search to it where /root/reference is absent (the GPU box).

  python tests/golden/make_dci_search_fixture.py [--long]      (--long: also the long runs, about ten minutes on one core)

A long run is evidence recorded once: its line says how many subframes the reference and the oracle walked and that their digests were equal."""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]
import ref_dci_search as R  # noqa: E402

OUT = os.path.join(HERE, "dci_search_ref.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--long", action="store_true")
    ap.add_argument("--long-missing", action="store_true", help="long runs only of the cases the present fixture has none for")
    a = ap.parse_args()
    old = json.load(open(OUT)) if os.path.exists(OUT) else {"cases": {}}
    out = {"made_by": "tests/golden/make_dci_search_fixture.py", "reference_sources": R.REF_SOURCES, "reference_sources_sha256": R.reference_sources_sha256(),
           "tuple": "(rnti, srsran_dci_format_t, L = log2 aggregation level, first CCE, DCI bits, histogram value handed to DCICollection::addCandidate)",
           "cases": {}}
    for case in R.CASES:
        name, sc_kw, nsf, nsf_long, meta, okw = case
        t = time.time()
        r = R.walk(case, with_reference=True)
        f, o = r["reference"], r["oracle"]
        c = dict(scenario=sc_kw, subframes=nsf, meta_period=meta, worker=okw, llr_sha256=r["llr_sha256"], searched=r["searched"],
                 reference=dict(digest=f["digest"], per_subframe=[R.sf_digest(i, x) for i, x in enumerate(f["per_sf"])], accepted=f["accepted"],
                                stats_locations_decoded_cce_missed_subframes=f["stats"], meta_final_primary_secondary=f["meta_final"],
                                activation_reasons_unset_evergreen_rar_shortcut_histogram_other=f["reasons"],
                                accepted_by_format_level_dci0_of_rar_rntis=f["probes"], first_subframes=[x for x in f["per_sf"][:6]]),
                 oracle_equal_when_made=o["digest"] == f["digest"])
        print("%-36s %5d subframes, %6d accepted DCI, oracle == reference: %s (%.0f s)" % (name, nsf, f["accepted"], o["digest"] == f["digest"], time.time() - t), flush=True)
        if a.long or (a.long_missing and "long_run" not in old["cases"].get(name, {})):
            t = time.time()
            prod = 10 ** 6
            r = R.walk(case, nsf=nsf_long, with_reference=True, product_subframes=prod)
            f, o = r["reference"], r["oracle"]
            c["long_run"] = dict(product_host_search_equal=r["product"]["per_sf"] == f["per_sf"],
                                 subframes=nsf_long, searched=r["searched"], llr_sha256=r["llr_sha256"], reference_digest=f["digest"], oracle_digest=o["digest"],
                                 equal=o["digest"] == f["digest"], accepted=f["accepted"], reference_stats=f["stats"], oracle_stats=o["stats"],
                                 activation_reasons=f["reasons"], accepted_by_format_level_dci0_of_rar_rntis=f["probes"], first_difference=R.first_difference(o["per_sf"], f["per_sf"]))
            if name == "cfg3_100prb_150rnti_rar":   # the same bytes as the gated bench stream's capture (tests/golden/cfg3_stream_oracle.json)
                sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "tools"))
                from make_cfg3_golden import capture_hash
                c["long_run"]["capture_xxh3_64"] = capture_hash(R.case_capture(case, nsf_long)[2])[0]
            print("%-36s %5d subframes, %6d accepted DCI, oracle == reference: %s, product == reference: %s (%.0f s)" % ("  long run", nsf_long, f["accepted"], o["digest"] == f["digest"], c["long_run"]["product_host_search_equal"], time.time() - t), flush=True)
        elif name in old["cases"] and "long_run" in old["cases"][name]:
            c["long_run"] = old["cases"][name]["long_run"]
        out["cases"][name] = c
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
