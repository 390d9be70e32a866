"""SURVEY 8 rows a5 / a6 / a8 / a9 / a19 (and a10 in place) pinned on the REFERENCE'S OWN CODE: /root/reference/src/src/DCISearch.cc (the FALCON decision
tree), lib/src/phy/falcon_phch/falcon_pdcch.c (location map, CCE power, search-space validation, missed CCEs), src/src/MetaFormats.cc and
lib/src/util/RNTIManager.cc are compiled verbatim into oracle/_ref/libref_falcon_search.so (oracle/Makefile.ref; srsRAN - an absent dependency - is
replaced by type declarations under oracle/ref_shim_search/ and by the oracle's own two DSP primitives, see search_glue.cc) and run subframe by
subframe on the oracle's PDCCH soft bits.  What the reference decided on ten streams is committed (tests/golden/dci_search_ref.json, made by
tests/golden/make_dci_search_fixture.py, which also records long runs of up to 20 000 subframes - the whole capture of the gated bench stream): per subframe the accepted DCI - RNTI, format,
aggregation level, first CCE, size, histogram value, in the order DCICollection::addCandidate receives them - plus the search statistics, the final
primary / secondary format split and the activation reasons in the RNTI manager.  The oracle's restatement (o_worker.c: blind_search / inspect) and the
product's host search (lsn_search.cc: FalconSearch, fed with candidate tables as k_viterbi produces them) must decide the same, subframe by subframe.
Where the library is present (the build container; it travels to the GPU box with the snapshot) the reference itself runs again.

One difference this harness found, in a printed statistic only: the reference adds a subframe's search statistics to the totals only when the search ran
(SubframeWorker.cc:173-177, SNR gate of DCISearch.cc:566), so its `nof_subframes` counts searched subframes; oracle and product count every subframe."""
import json
import os

import pytest

import ref_dci_search as R

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dci_search_ref.json")))
CASES = {c[0]: c for c in R.CASES}
HAVE_LIB = os.path.exists(R.REF_SO)
_walks = {}


def _prefix(case):
    name, sc_kw, nsf, nsf_long, meta, okw = case
    kw = dict(sc_kw)
    return R.PRODUCT_SUBFRAMES[R.scenario(kw.pop("name"), **kw)["nof_prb"]]


def _walk(name):
    """one walk per case and session: oracle + product prefix + (where the library is here) the reference"""
    if name not in _walks:
        _walks[name] = R.walk(CASES[name], with_reference=HAVE_LIB, product_subframes=_prefix(CASES[name]))
    return _walks[name]


def _against_fixture(name, who, per_sf, n=None):
    f = FIX["cases"][name]
    r = _walk(name)
    assert r["llr_sha256"] == f["llr_sha256"], "the oracle's PDCCH soft bits of this stream changed: run tests/golden/make_dci_search_fixture.py again"
    exp = f["reference"]["per_subframe"][:n]
    got = [R.sf_digest(i, a) for i, a in enumerate(per_sf[:n])]
    if got != exp:
        i = next(j for j, (a, b) in enumerate(zip(got, exp)) if a != b)
        hint = (" the reference accepted %s" % f["reference"]["first_subframes"][i]) if i < len(f["reference"]["first_subframes"]) else ""
        raise AssertionError("%s: %s decides differently from the reference's DCISearch.cc in subframe %d: %s.%s" % (name, who, i, per_sf[i], hint))
    return f, r


def test_fixture_covers_the_cases_and_was_equal_when_made():
    assert set(FIX["cases"]) == set(CASES)
    for name, f in FIX["cases"].items():
        c = CASES[name]
        assert (f["scenario"], f["subframes"], f["meta_period"], f["worker"]) == (c[1], c[2], c[4], c[5]), "the case list changed: run the generator again"
        assert f["oracle_equal_when_made"] and len(f["reference"]["per_subframe"]) == f["subframes"]
        assert f["long_run"]["equal"] and f["long_run"]["subframes"] == c[3] and f["long_run"]["first_difference"] is None
        assert f["long_run"]["product_host_search_equal"] is True


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_search_decides_like_the_reference(name):
    r = _walk(name)
    f, _ = _against_fixture(name, "the oracle", r["oracle"]["per_sf"])
    fr, o = f["reference"], r["oracle"]
    assert o["accepted"] == fr["accepted"] and o["digest"] == fr["digest"]
    # statistics: locations, decoded locations, CCEs, missed CCEs equal; nof_subframes: see the module text
    assert o["stats"][:4] == fr["stats_locations_decoded_cce_missed_subframes"][:4]
    assert fr["stats_locations_decoded_cce_missed_subframes"][4] == f["searched"] == r["searched"] and o["stats"][4] == f["subframes"]
    assert o["reasons"] == fr["activation_reasons_unset_evergreen_rar_shortcut_histogram_other"]


@pytest.mark.parametrize("name", sorted(n for n in CASES if _prefix(CASES[n])))
def test_product_host_search_decides_like_the_reference(name):
    r = _walk(name)
    n = _prefix(CASES[name])
    assert len(r["product"]["per_sf"]) == min(n, r["subframes"])
    _against_fixture(name, "the product's FalconSearch", r["product"]["per_sf"], n)


def test_the_longest_run_is_the_capture_of_the_gated_bench_stream():
    """cfg3, seed 3, 20 000 subframes: the bytes the reference's search, the oracle and the product's host search walked in the long run are the bytes bench.py replays"""
    lr = FIX["cases"]["cfg3_100prb_150rnti_rar"]["long_run"]
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg3_stream_oracle.json")))
    assert lr["capture_xxh3_64"] == g["capture_xxh3_64"] and lr["subframes"] == g["stream"]["distinct_subframes"] == 20000
    assert lr["equal"] and lr["product_host_search_equal"] and lr["accepted"] > 300000


def test_candidate_tables_built_in_c_are_the_python_ones():
    """the product's search is fed with candidate tables built by lsnh_candidate_table (C, fast enough for whole streams); lsn_testlib.candidate_table (Python
    loops, used by tests/test_host_logic.py) is the definition: same entries, same CCE powers"""
    from lsn_testlib import OracleWorker, candidate_table
    case = CASES["cfg3_50prb_four_ports"]
    sc, tti0, iq = R.case_capture(case, 6)
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"], sc["phich_ng_x6"])
    p = R.ProductSearch(sc)
    for i in range(6):
        ow.work(iq[i], tti0 + i)
        c1, w1 = p.table(ow.llr(), ow.cfi(), tti0 + i)
        c2, w2 = candidate_table(ow.llr(), p.regs_cce[ow.cfi() - 1], p.sizes, (tti0 + i) % 10)
        assert bytes(c1) == bytes(c2) and w1.tobytes() == w2.tobytes()
        assert sum(1 for e in c1 if e.flags & 1) > 100
    p.close()


def test_the_streams_walk_through_the_tree():
    """the pin is only worth something if the streams reach the branches: RNTIs activated by random access, by the shortcut and by the histogram; a stream without
    shortcut discovery and without secondary formats (many missed CCEs); the SNR gate; format splits that differ; DCI 0 of freshly random-accessed RNTIs (the
    temp_dci0 path, DCISearch.cc:139-158 / 422-432); DCI of formats 0, 1, 1A, 2 and 2A at all four aggregation levels"""
    c = FIX["cases"]
    reasons = lambda n: c[n]["reference"]["activation_reasons_unset_evergreen_rar_shortcut_histogram_other"]
    probes = lambda n: c[n]["reference"]["accepted_by_format_level_dci0_of_rar_rntis"]
    assert reasons("cfg3_100prb_150rnti_rar")[2] >= 1 and reasons("cfg3_100prb_150rnti_rar")[3] >= 140
    assert reasons("cfg3_skip_secondary_no_shortcut")[3] == 0 and reasons("cfg3_skip_secondary_no_shortcut")[4] >= 100
    assert reasons("cfg3_15prb_cfi_small_region")[2] >= 5 and reasons("small_25prb_16dB")[2] >= 3
    assert c["cfg1_50prb_low_snr_gate"]["searched"] < c["cfg1_50prb_low_snr_gate"]["subframes"]
    assert c["cfg3_skip_secondary_no_shortcut"]["reference"]["stats_locations_decoded_cce_missed_subframes"][3] > 1000
    splits = {tuple(map(tuple, c[n]["reference"]["meta_final_primary_secondary"])) for n in c}
    assert len(splits) >= 4 and all(len(p) + len(s) == 9 for p, s in splits)
    assert probes("small_25prb_16dB")["dci0_of_rar_rntis"] >= 50 and probes("cfg3_15prb_cfi_small_region")["dci0_of_rar_rntis"] >= 100
    assert all(v > 50 for v in probes("cfg3_100prb_150rnti_rar")["levels"])
    assert all(probes("cfg3_100prb_150rnti_rar")["formats"][f] > 100 for f in (0, 1, 2, 6, 7))
    for n in c:
        lr = c[n]["long_run"]
        assert lr["accepted"] > lr["subframes"] and lr["reference_stats"][:4] == lr["oracle_stats"][:4]
    assert sum(c[n]["long_run"]["subframes"] for n in c) >= 35000 and sum(c[n]["long_run"]["accepted"] for n in c) >= 380000
    assert sum(c[n]["long_run"]["accepted_by_format_level_dci0_of_rar_rntis"]["dci0_of_rar_rntis"] for n in c) >= 5000


@pytest.mark.skipif(not HAVE_LIB, reason="oracle/_ref/libref_falcon_search.so not built (needs /root/reference: make -C oracle -f Makefile.ref)")
@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_library_reproduces_the_committed_fixture(name):
    r = _walk(name)
    f, _ = _against_fixture(name, "the library built here", r["reference"]["per_sf"])
    fr, g = f["reference"], r["reference"]
    assert (g["digest"], g["accepted"], g["stats"], g["meta_final"], g["reasons"]) == (
        fr["digest"], fr["accepted"], fr["stats_locations_decoded_cce_missed_subframes"], fr["meta_final_primary_secondary"],
        fr["activation_reasons_unset_evergreen_rar_shortcut_histogram_other"])


@pytest.mark.skipif(not HAVE_LIB, reason="oracle/_ref/libref_falcon_search.so not built")
def test_search_space_validation_of_the_reference_and_the_oracle_agree_everywhere_probed():
    """srsran_pdcch_validate_location (falcon_pdcch.c:223-253) against o_validate_location: every (L, nCCE) of five control-region sizes, all ten
    subframes, RNTIs from every interval of TS 36.321 7.1"""
    import ctypes as C
    from lsn_testlib import oracle
    o = oracle()
    o.o_validate_location.restype = C.c_uint32
    o.o_validate_location.argtypes = [C.c_uint32] * 4 + [C.c_uint16]
    lib = C.CDLL(R.REF_SO)
    lib.ref_search_validate_location.restype = C.c_uint32
    lib.ref_search_validate_location.argtypes = [C.c_uint32] * 4 + [C.c_uint16]
    rntis = [0x0001, 0x0005, 0x000A, 0x000B, 0x003C, 0x003D, 0x0100, 0x1234, 0x4F21, 0x8000, 0xFFF3, 0xFFF4, 0xFFFC, 0xFFFD, 0xFFFE, 0xFFFF]
    n = 0
    for nof_cce in (2, 12, 21, 41, 87):
        for sf in range(10):
            for rnti in rntis:
                for l in range(4):
                    for ncce in range(0, nof_cce, 1 << l):
                        a, b = lib.ref_search_validate_location(nof_cce, ncce, l, sf, rnti), o.o_validate_location(nof_cce, ncce, l, sf, rnti)
                        assert a == b, (nof_cce, ncce, l, sf, hex(rnti), a, b)
                        n += 1
    assert n > 40000


@pytest.mark.skipif(not HAVE_LIB, reason="oracle/_ref/libref_falcon_search.so not built")
def test_the_glues_search_space_enumerations_are_the_ones_the_reference_states():
    """search_glue.cc writes srsRAN's two enumerations of TS 36.213 9.1.1 (absent dependency).  The reference states the same sets itself, as a membership test
    (srsran_pdcch_ue_locations_check, falcon_pdcch.c:49-103): a first CCE passes it exactly when the glue's lists hold a candidate that starts there."""
    import ctypes as C
    lib = C.CDLL(R.REF_SO)
    lib.ref_search_locations_check.restype = C.c_uint32
    lib.ref_search_locations_check.argtypes = [C.c_uint32, C.c_uint32, C.c_uint16, C.c_uint32]
    lib.ref_search_glue_locations.restype = C.c_uint32
    lib.ref_search_glue_locations.argtypes = [C.c_uint32, C.c_uint32, C.c_uint16, C.c_void_p]
    buf = (C.c_uint32 * 44)()
    n = 0
    for nof_cce in (1, 2, 5, 8, 12, 21, 27, 41, 55, 84, 87):
        for sf in range(10):
            for rnti in (0x000B, 0x0100, 0x1234, 0x2AF1, 0x4F21, 0x8000, 0xC35A, 0xFFF3):
                k = lib.ref_search_glue_locations(nof_cce, sf, rnti, buf)
                starts = {buf[2 * i + 1] for i in range(k)}
                assert all(buf[2 * i + 1] + (1 << buf[2 * i]) <= nof_cce for i in range(k))
                for ncce in range(nof_cce + 2):
                    assert bool(lib.ref_search_locations_check(nof_cce, sf, rnti, ncce)) == (ncce in starts), (nof_cce, sf, hex(rnti), ncce)
                    n += 1
    assert n > 25000


def test_fixture_was_made_from_the_reference_sources_that_are_here():
    if not os.path.isdir("/root/reference/src/src"):
        pytest.skip("no /root/reference on this host")
    assert R.reference_sources_sha256() == FIX["reference_sources_sha256"]
