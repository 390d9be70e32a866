#!/usr/bin/env python3
"""tests/golden/make_mcs_fixture.py -> tests/golden/mcs_tracking_ref.json: what the REFERENCE'S OWN MCSTracking.cc (oracle/_ref/libref_falcon_mcs.so) answers over the
database lives of tests/test_mcs_ageing.py: life(), as digests, plus the default UE configuration it hands out for an unknown RNTI."""
import ctypes as C
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]
import test_mcs_ageing as T  # noqa: E402


def main():
    h = hashlib.sha256()
    for f in ("src/src/MCSTracking.cc", "src/src/Sniffer_dependency.cc"):
        h.update(open(os.path.join("/root/reference", f), "rb").read())
    out = {"made_by": "tests/golden/make_mcs_fixture.py", "reference_sources_sha256": h.hexdigest(), "lives": {}}
    for seed in T.LIFE_SEEDS:
        d, info = T._life_on(T._Reference, seed)
        p, pinfo = T._life_on(T._Product, seed)
        out["lives"][str(seed)] = {"digest": d, "counters": info, "product_equal_when_made": (p, pinfo) == (d, info)}
        print(seed, d, info, (p, pinfo) == (d, info))
    r = T._Reference()
    buf = (C.c_uint32 * 6)()
    r.lib.ref_mcs_get_ue_config(r.m, 0x1234, buf)
    r.close()
    out["default_ue_config_of_an_unknown_rnti"] = list(buf)
    r = T._Reference()
    out["corner_script"] = T.corners(r)
    r.close()
    print("corner script", out["corner_script"])
    json.dump(out, open(os.path.join(HERE, "mcs_tracking_ref.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
