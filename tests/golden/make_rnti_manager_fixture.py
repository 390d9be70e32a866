#!/usr/bin/env python3
"""tests/golden/rnti_manager_ref.json: what the REFERENCE's own RNTI manager (RNTIManager.cc + Histogram.cc + Interval.cc, compiled from
/root/reference by oracle/Makefile.ref into oracle/_ref/) answers to the operation programs of tests/rnti_manager_ops.py.
Run in the build container (needs /root/reference): python tests/golden/make_rnti_manager_fixture.py
The fixture travels; the library does too (oracle/_ref/ is git-ignored, not gpurun-ignored), /root/reference does not."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import rnti_manager_ops as R  # noqa: E402


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref"], stdout=subprocess.DEVNULL)
    ref = R.Reference()
    cases = []
    for seed, nf, mc, thr, steps in R.CASES:
        c = {"seed": seed, "nformats": nf, "max_candidates_per_step_per_format": mc, "histogram_threshold": thr, "steps": steps}
        for ext in (False, True):
            prog = R.program(seed, nf, mc, thr, steps, extended=ext)
            out = ref.run(prog)
            k = "extended" if ext else "basic"
            c[k] = {"operations": len(prog), "values": len(out), "sha256_32": R.digest(out), "accepted": int(sum(1 for o, v in zip([p for p in prog if p[0] in ("vr", "val")], [v for p, v in _pairs(prog, out) if p[0] in ("vr", "val")]) if v))}
            if seed == 0 and not ext:
                c[k]["first_values"] = out[:600]  # so that a difference can be located without the library
        cases.append(c)
    sha = subprocess.check_output("cat /root/reference/lib/src/util/RNTIManager.cc /root/reference/lib/src/util/Histogram.cc /root/reference/lib/src/util/Interval.cc | sha256sum", shell=True).decode().split()[0]
    json.dump({"what": "answers of the reference's own RNTIManager (oracle/_ref/libref_falcon_util.so) to tests/rnti_manager_ops.py programs",
               "reference_sources_sha256": sha, "cases": cases}, open(os.path.join(HERE, "rnti_manager_ref.json"), "w"), indent=1)
    for c in cases:
        print(c["seed"], c["basic"]["operations"], c["basic"]["values"], c["basic"]["accepted"], c["basic"]["sha256_32"], c["extended"]["sha256_32"])


def _pairs(prog, out):
    """(operation, value) for the operations that hand a value back"""
    it = iter(out)
    return [(p, next(it)) for p in prog if p[0] in ("vr", "val", "freq", "reason", "isfb", "isev")]


if __name__ == "__main__":
    main()
