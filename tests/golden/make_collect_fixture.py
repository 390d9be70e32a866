#!/usr/bin/env python3
"""tests/golden/make_collect_fixture.py -> tests/golden/collect_ref.json

Answers of the REFERENCE'S OWN DCI collection (oracle/_ref/libref_falcon_collect.so: DCICollection.cc, falcon_dci.c, dl_sniffer_pdsch.c incl. the C-RNTI branch,
ul_sniffer_pusch.c, ULSchedule.cc, MCSTracking.cc, HARQ.cc compiled from /root/reference by oracle/Makefile.ref) to the scripts of tests/ref_collect.py, as
digests: ten lives of a DCICollection, the RAR-grant sweep, two ULSchedule scripts.  With --long the lives are walked at ten times the suite's length next to the
oracle (counts only)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]
import ref_collect as R  # noqa: E402
from lsn_testlib import hosttest, oracle  # noqa: E402


def life_facts(res):
    dl = [r for x in res for r in R.normalise(x)[1]]
    ul = [r for x in res for r in R.normalise(x)[2]]
    return {"subframes": len(res), "dl_entries": len(dl), "ul_entries": len(ul), "dl_by_format": [sum(r[1] == f for r in dl) for f in range(9)],
            "dl_by_table": [sum(r[2] == t for r in dl) for t in range(5)], "dl_conversion_failed": sum(r[3] == 0 for r in dl),
            "ul_conversion_failed": sum(r[1] == 0 for r in ul), "ul_type1_hopping": sum(r[13] == 1 for r in ul),
            "size_from_harq_database": sum(r[13] == 1 for r in dl), "two_block_grants": sum(r[16] == 2 or r[41] == 2 for r in dl),
            "slot_hopping_allocations": sum(r[17:21] != r[21:25] or r[42:46] != r[46:50] for r in dl),
            "dl_collisions": sum(x[0] & 1 for x in res), "ul_collisions": sum((x[0] >> 1) & 1 for x in res)}


def main():
    ref, orc, prod = R.Reference(), R.Oracle(), R.Product()
    o, h = oracle(), hosttest()
    out = {"made_by": "tests/golden/make_collect_fixture.py", "reference_sources": R.REF_SOURCES, "reference_sources_sha256": R.reference_sources_sha256(),
           "row_layout": "oracle/ref_shim_search/collect_glue.cc: 64 words per downlink entry, 32 per uplink entry; tests/ref_collect.py: normalise() says what is not compared",
           "lives": {}, "oracle_equal_when_made": {}, "product_equal_when_made": {}}
    for life in R.LIVES:
        a = ref.run(life)
        out["lives"][life[0]] = dict(life_facts(a), digest=R.digest(a), digest_without_maps=R.digest([prod.view(x) for x in a]))
        out["oracle_equal_when_made"][life[0]] = R.digest(orc.run(life)) == R.digest(a)
        if not life[7]:
            out["product_equal_when_made"][life[0]] = R.digest([prod.view(x) for x in prod.run(life)]) == R.digest([prod.view(x) for x in a])
    rar = [tuple(v & 0xFFFFFFFF for v in R.rar_reference(ref, *a)) for a in R.rar_sweep()]
    out["rar"] = {"cases": len(rar), "grants_converted": sum(r[6] for r in rar), "with_hopping_flag": sum(r[0] for r in rar), "digest": R.digest_rows(rar),
                  "oracle_equal_when_made": rar == [tuple(v & 0xFFFFFFFF for v in R.rar_oracle(o, *a)) for a in R.rar_sweep()],
                  "product_equal_when_made": rar == [tuple(v & 0xFFFFFFFF for v in R.rar_product(h, *a)) for a in R.rar_sweep()]}
    out["ulsche"] = {}
    for name, gaps in (("gapless", False), ("with_gaps_and_restarts", True)):
        s = R.ulsche_script(gaps=gaps)
        x = R.ulsche_reference(ref, s)
        ring = R.ulsche_ring(s)
        out["ulsche"][name] = {"fetches": len(x), "non_empty": sum(v is not None for v in x), "digest": R.digest_rows(x), "model_equal": x == R.ulsche_model(s),
                               "fetches_where_a_16_slot_ring_differs": sum(a != b for a, b in zip(x, ring))}
    if "--long" in sys.argv:
        out["long_run"] = {}
        for life in R.LIVES:
            big = life[:9] + (life[9] * 10, life[10] + 100)
            a, b = ref.run(big), orc.run(big)
            out["long_run"][life[0]] = dict(life_facts(a), oracle_differs_in=sum(R.normalise(x) != R.normalise(y) for x, y in zip(a, b)))
            if not life[7]:
                c = prod.run(big)
                out["long_run"][life[0]]["product_differs_in"] = sum(prod.view(R.normalise(x)) != prod.view(R.normalise(y)) for x, y in zip(a, c))
    else:
        old = os.path.join(HERE, "collect_ref.json")
        if os.path.exists(old):
            out["long_run"] = json.load(open(old)).get("long_run", {})
    json.dump(out, open(os.path.join(HERE, "collect_ref.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k.endswith("made")}), out["rar"], out["ulsche"])


if __name__ == "__main__":
    main()
