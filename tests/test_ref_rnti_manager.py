"""SURVEY 8 row a10 pinned on the REFERENCE'S OWN CODE: /root/reference/lib/src/util/RNTIManager.cc (+ Histogram.cc, Interval.cc) is the one part of
the path that compiles from its own source files (no srsRAN symbol; oracle/Makefile.ref -> oracle/_ref/libref_falcon_util.so).  Its answers to the
operation programs of tests/rnti_manager_ops.py are committed (tests/golden/rnti_manager_ref.json, made by tests/golden/make_rnti_manager_fixture.py);
the oracle's restatement (o_falcon.c: o_rntiman_*) and the product's host class (lsn_lte.cc: RNTIManager) must give the same answers, value by value.
Where the library is present (the build container; it travels to the GPU box with the snapshot) the reference itself runs again."""
import json
import os

import pytest

import rnti_manager_ops as R

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rnti_manager_ref.json")))
CASES = {c["seed"]: c for c in FIX["cases"]}


def _prog(seed, extended):
    c = CASES[seed]
    return R.program(seed, c["nformats"], c["max_candidates_per_step_per_format"], c["histogram_threshold"], c["steps"], extended=extended)


def _check(backend, seed, extended):
    c = CASES[seed]["extended" if extended else "basic"]
    prog = _prog(seed, extended)
    assert len(prog) == c["operations"], "the program generator changed: run tests/golden/make_rnti_manager_fixture.py again"
    out = backend.run(prog)
    assert len(out) == c["values"]
    if "first_values" in c and out[:len(c["first_values"])] != c["first_values"]:
        i = next(j for j, (a, b) in enumerate(zip(out, c["first_values"])) if a != b)
        ops = [p for p in prog if p[0] in ("vr", "val", "freq", "reason", "isfb", "isev")]
        raise AssertionError("%s: value %d (%s) is %d, the reference answers %d" % (backend.name, i, ops[i], out[i], c["first_values"][i]))
    assert R.digest(out) == c["sha256_32"], "%s differs from the reference's RNTIManager on program %d (%s)" % (backend.name, seed, "extended" if extended else "basic")


@pytest.mark.parametrize("seed", sorted(CASES))
def test_oracle_rnti_manager_answers_like_the_reference(seed):
    _check(R.Oracle(), seed, False)


@pytest.mark.parametrize("seed", sorted(CASES))
def test_product_rnti_manager_answers_like_the_reference(seed):
    p = R.Product()
    _check(p, seed, False)
    _check(p, seed, True)   # + validate() without refresh and isEvergreen(), which the oracle's interface does not have


def test_the_programs_reach_every_branch_of_the_manager():
    """the fixture is only worth something if the programs walk through the manager: acceptances by evergreen, by the active list, by the histogram;
    rejections by the forbidden list, by the wrong downlink format, by frequency; expiry after the lifetime; padding; over-full steps"""
    c = CASES[0]
    prog = _prog(0, False)
    vals = [p for p in prog if p[0] in ("vr", "val", "freq", "reason", "isfb", "isev")]
    first = c["basic"]["first_values"]
    by = {}
    for p, v in zip(vals, first):
        by.setdefault(p[0], set()).add(min(v, 2))
    assert by["vr"] == {0, 1} and by["isfb"] == {0, 1} and 2 in by["freq"]
    assert {0, 2, 4} <= {v for p, v in zip(vals, first) if p[0] == "reason"} | {0}   # unset, random access, histogram among the first probes
    assert any(p[0] == "step" and p[1] > 9990 for p in prog) and any(p[0] == "step" and 1 < p[1] < 400 for p in prog)
    for seed, cc in CASES.items():
        assert 0 < cc["basic"]["accepted"] < cc["basic"]["values"]
    # an over-full step: more candidates between two time steps than the per-step budget
    run, worst = 0, 0
    for p in prog:
        run = run + 1 if p[0] == "cand" else (0 if p[0] == "step" else run)
        worst = max(worst, run)
    assert worst > c["max_candidates_per_step_per_format"]


@pytest.mark.skipif(not os.path.exists(R.REF_SO), reason="oracle/_ref/libref_falcon_util.so not built (needs /root/reference: make -C oracle -f Makefile.ref)")
@pytest.mark.parametrize("seed", sorted(CASES))
def test_reference_library_reproduces_the_committed_fixture(seed):
    ref = R.Reference()
    _check(ref, seed, False)
    _check(ref, seed, True)


def test_fixture_was_made_from_the_reference_sources_that_are_here():
    src = "/root/reference/lib/src/util"
    if not os.path.isdir(src):
        pytest.skip("no /root/reference on this host")
    import hashlib
    h = hashlib.sha256()
    for f in ("RNTIManager.cc", "Histogram.cc", "Interval.cc"):
        h.update(open(os.path.join(src, f), "rb").read())
    assert h.hexdigest() == FIX["reference_sources_sha256"]
