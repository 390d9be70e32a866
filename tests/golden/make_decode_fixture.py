#!/usr/bin/env python3
"""tests/golden/make_decode_fixture.py -> tests/golden/decode_ref.json

What the REFERENCE'S OWN downlink decode control flow (oracle/_ref/libref_falcon_decode.so: PDSCH_Decoder::decode_dl_mode of DL_Sniffer_PDSCH.cc compiled from
/root/reference on top of its own DCICollection / falcon_dci.c / MCSTracking / HARQ / RNTIManager, oracle/Makefile.ref) does with the scripts of tests/ref_decode.py
under a scripted PDSCH decoder, as digests: eight lives.  With --long the lives are walked at five times the suite's length next to the oracle (counts only)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]
import ref_decode as D  # noqa: E402


def strip(results):
    return [r for r in results if r[0] != "harq"]


def main():
    ref, orc = D.Reference(), D.Oracle()
    out = {"made_by": "tests/golden/make_decode_fixture.py", "reference_sources": D.REF_SOURCES, "reference_sources_sha256": D.reference_sources_sha256(),
           "what_is_compared": "per subframe: every call of srsran_ue_dl_decode_pdsch as configured (16 words + p_a) with the scripted verdicts, every record handed to the pcap "
                               "writer (kind, tti, rnti, length, CRC flag, FNV-1a of the bytes); at check points (after every ageing pass, at the end): population of the "
                               "tracking database and per probed RNTI its table, the RNTI manager's activation reason, the UE configuration",
           "lives": {}, "oracle_equal_when_made": {}}
    for life in D.LIVES:
        a = D.run(ref, life)
        b = D.run(orc, life)
        hq = [r[1] for r in b if r[0] == "harq"]
        out["lives"][life[0]] = dict(D.facts(a), digest=D.digest(a), check_points=sum(r[0] == "state" for r in a),
                                     tables_at_the_end=[sum(t[1] == k for t in a[-1][1][1]) for k in range(5)],
                                     rar_activated_at_the_end=sum(t[2] == 2 for t in a[-1][1][1]), ue_configs_at_the_end=sum(t[4] == 1 for t in a[-1][1][1]),
                                     harq_verdicts_new_retx_full_decoded_busy=list(hq[0]) if hq else None)
        out["oracle_equal_when_made"][life[0]] = D.digest(strip(b)) == D.digest(a)
    # the case the regular lives leave out (ref_decode.script): first block with a reserved MCS index
    life = D.LIVES[0]
    a, b = D.run(ref, life, True), strip(D.run(orc, life, True))
    diff = [i for i, (x, y) in enumerate(zip(a, b)) if x != y]
    out["reserved_first_block"] = {"life": life[0], "subframes": len(a), "first_differing": diff[0] if diff else None, "differing": len(diff)}
    if "--long" in sys.argv:
        out["long_run"] = {}
        for life in D.LIVES:
            big = life[:8] + (life[8] * 5, life[9] + 100, life[10])
            a, b = D.run(ref, big), strip(D.run(orc, big))
            out["long_run"][life[0]] = dict(D.facts(a), oracle_differs_in=sum(x != y for x, y in zip(a, b)))
    else:
        old = os.path.join(HERE, "decode_ref.json")
        if os.path.exists(old):
            out["long_run"] = json.load(open(old)).get("long_run", {})
    json.dump(out, open(os.path.join(HERE, "decode_ref.json"), "w"), indent=1)
    print(json.dumps(out["oracle_equal_when_made"]), out["reserved_first_block"], {k: v.get("oracle_differs_in") for k, v in out.get("long_run", {}).items()})


if __name__ == "__main__":
    main()
