"""SURVEY 8(f) row 4 (downlink HARQ database) pinned on the REFERENCE'S OWN CODE: /root/reference/src/src/HARQ.cc compiled verbatim into
oracle/_ref/libref_falcon_harq.so (oracle/Makefile.ref; the three srsRAN soft-buffer calls are no-ops, clock() is a settable clock of 1 ms per subframe:
oracle/ref_shim_search/harq_glue.cc).  Random lives of a cell - a few hundred UEs, eight processes, two transport blocks, retransmissions exactly eight
subframes later with and without a toggled NDI, changed sizes, late repeats, more UEs than entities, idle gaps around the 5 s ageing interval, the TTI
wrapping at 10 240 - ask the reference and the product's HarqDatabase (lsn_lte.cc, through tests/native) the question the PDSCH decoder asks per transport
block (DL_Sniffer_PDSCH.cc:943-1020): new transmission, retransmission to combine, already decoded, or database full.  The reference's verdicts are committed
as digests (tests/golden/harq_ref.json, made by `python tests/test_ref_harq.py`); the library itself runs again where it is present.
The protocol of the caller is kept (every NEW_TX / RE_TX verdict is followed by the update of that process before the next question about it), so the
reference's per-process lock never answers BUSY - as in its own single PDSCH thread per subframe."""
import ctypes as C
import hashlib
import json
import os
import random
import sys

import pytest

sys.path[:0] = [os.path.dirname(os.path.abspath(__file__))]
from lsn_testlib import hosttest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_falcon_harq.so")
FIX = os.path.join(ROOT, "tests", "golden", "harq_ref.json")
SEEDS = (21, 22, 23, 24, 25, 26)
NEW_TX, RE_TX, FULL, DECODED, BUSY = 0, 1, 2, 3, 4


class Product:
    def __init__(self):
        h = self.h = hosttest()
        h.lsnh_harq_new.restype = C.c_void_p
        h.lsnh_harq_free.argtypes = [C.c_void_p]
        h.lsnh_harq_is_retransmission.argtypes = [C.c_void_p, C.c_uint16, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p]
        h.lsnh_harq_update.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32]
        h.lsnh_harq_update_database.argtypes = [C.c_void_p, C.c_uint32]
        self.d = h.lsnh_harq_new()
        self.now = 0
        self.entity = {}

    def set_now(self, ms): self.now = ms

    def ask(self, rnti, pid, tid, ndi, rv, tbs, sfn, sf):
        e = C.c_int(-1)
        r = self.h.lsnh_harq_is_retransmission(self.d, rnti, pid, tid, ndi, tbs, sfn, sf, C.byref(e))
        self.entity[(rnti, pid, tid)] = e.value
        return r

    def update(self, rnti, pid, tid, sfn, sf, decoded, ndi, rv, tbs):
        self.h.lsnh_harq_update(self.d, self.entity[(rnti, pid, tid)], pid, tid, sfn, sf, decoded, ndi, rv, tbs, self.now)

    def age(self): self.h.lsnh_harq_update_database(self.d, self.now)
    def close(self): self.h.lsnh_harq_free(self.d)


class Reference:
    def __init__(self):
        L = self.lib = C.CDLL(REF_SO)
        L.ref_harq_new.restype = C.c_void_p
        L.ref_harq_free.argtypes = [C.c_void_p]
        L.ref_harq_set_now_ms.argtypes = [C.c_uint64]
        L.ref_harq_is_retransmission.argtypes = [C.c_void_p, C.c_uint16] + [C.c_int] * 5 + [C.c_uint32] * 2
        L.ref_harq_update.argtypes = [C.c_void_p, C.c_uint16, C.c_int, C.c_int, C.c_uint32, C.c_uint32] + [C.c_int] * 4
        L.ref_harq_update_database.argtypes = [C.c_void_p]
        L.ref_harq_size.argtypes = [C.c_void_p]
        self.d = L.ref_harq_new()
        assert L.ref_harq_size(self.d) == 300   # 150 from the constructor + 150 from init_HARQ (HARQ.cc:15-19, 47-51)

    def set_now(self, ms): self.lib.ref_harq_set_now_ms(ms)
    def ask(self, rnti, pid, tid, ndi, rv, tbs, sfn, sf): return self.lib.ref_harq_is_retransmission(self.d, rnti, pid, tid, ndi, rv, tbs, sfn, sf)
    def update(self, rnti, pid, tid, sfn, sf, decoded, ndi, rv, tbs): self.lib.ref_harq_update(self.d, rnti, pid, tid, sfn, sf, decoded, ndi, rv, tbs)
    def age(self): self.lib.ref_harq_update_database(self.d)
    def close(self): self.lib.ref_harq_free(self.d)


def life(seed, b, steps=30000):
    """-> (digest of all verdicts, how often each verdict was given)"""
    rng = random.Random(seed)
    n_ue = (120, 280, 340, 420, 200, 330)[seed % 6]      # fewer and more UEs than the 300 entities
    ues = [rng.randrange(11, 0xFFF4) for _ in range(n_ue)]
    busy = ues[:40]
    state = {}       # (rnti, pid, tid) -> (tti, ndi, tbs, decoded, rv) of the last transmission this test sent
    pending = []     # retransmissions due: (tti, rnti, pid, tid, ndi, rv, tbs)
    tti = rng.randrange(0, 10240)
    h = hashlib.sha256()
    count = [0] * 5
    t = 0
    for step in range(steps):
        gap = rng.choice((1,) * 40 + (2, 3, 7, 8, 9, 40))
        if rng.random() < 0.002:
            gap = rng.choice((5200, 4900, 6100) if seed % 2 else (700, 2500))   # idle gaps around the 5 s ageing interval
        t += gap
        tti = (tti + gap) % 10240
        b.set_now(t)
        sfn, sf = tti // 10, tti % 10
        grants = [p[1:] for p in pending if p[0] == tti]
        pending = [p for p in pending if p[0] != tti and (p[0] - tti) % 10240 < 64]
        for _ in range(rng.randrange(0, 5)):
            r = rng.choice(busy) if rng.random() < 0.6 else rng.choice(ues)
            pid, tid = rng.randrange(8), rng.randrange(2)
            last = state.get((r, pid, tid))
            ndi = (1 - last[1]) if last and rng.random() < 0.8 else rng.randrange(2)     # mostly toggled: a new transport block
            grants.append((r, pid, tid, ndi, 0, rng.choice((1000, 2536, 4968, 11448, 75376))))
        for r, pid, tid, ndi, rv, tbs in grants:
            v = b.ask(r, pid, tid, ndi, rv, tbs, sfn, sf)
            count[v] += 1
            h.update(b"%d:%d.%d.%d=%d;" % (t, r, pid, tid, v))
            assert v != BUSY
            if v in (NEW_TX, RE_TX):
                decoded = int(rng.random() < (0.35 if v == NEW_TX else 0.7))
                b.update(r, pid, tid, sfn, sf, decoded, ndi, rv, tbs)
                state[(r, pid, tid)] = (tti, ndi, tbs, decoded, rv)
                x = rng.random()
                if x < 0.45:    # the eNB repeats it 8 subframes later: same NDI and size (retransmission), or something that only looks like one
                    y = rng.random()
                    pending.append(((tti + 8) % 10240, r, pid, tid, ndi if y < 0.8 else 1 - ndi, (rv + 2) % 4, tbs if y < 0.9 or y >= 0.95 else tbs + 8))
                elif x < 0.5:   # a late repeat (9 subframes): not a retransmission for the reference
                    pending.append(((tti + 9) % 10240, r, pid, tid, ndi, (rv + 2) % 4, tbs))
        if t // 10000 != (t - gap) // 10000:   # the 10 s timer of LTESniffer_Core.cc:487-494
            b.age()
            h.update(b"age;")
    return h.hexdigest()[:32], count


def corners(b):
    """a scripted life for what the random ones hit too rarely: the ageing pass that only runs when at most 10 of the 150 counted entities are free
    (updateHARQDatabase, HARQ.cc:206-238), on both sides of that limit, and a retransmission that arrives one whole TTI period + 8 subframes late
    (10 248 ms: the reference compares TTIs modulo 10 240 and still calls it a retransmission)"""
    out = []
    tti0 = 100
    for k in range(139):                                   # 139 UEs: 11 of the 150 counted entities stay free
        t = k
        b.set_now(t)
        tti = (tti0 + t) % 10240
        out.append(b.ask(1000 + k, 0, 0, 0, 0, 1000, tti // 10, tti % 10))
        b.update(1000 + k, 0, 0, tti // 10, tti % 10, 0, 0, 0, 1000)
    t = 10248
    b.set_now(t)
    tti = (tti0 + t) % 10240                               # = tti0 + 8
    b.age()                                                # 11 free: the pass does nothing, although every entity has been idle for 10 s
    out.append(b.ask(1000, 0, 0, 0, 2, 1000, tti // 10, tti % 10))   # UE 1000 again, same NDI and size, "8 subframes" later: a retransmission
    b.update(1000, 0, 0, tti // 10, tti % 10, 1, 0, 2, 1000)           # ... which decodes
    tti2 = (tti + 8) % 10240
    b.set_now(t + 8)
    out.append(b.ask(1000, 0, 0, 0, 3, 1000, tti2 // 10, tti2 % 10))  # and again: already decoded
    out.append(b.ask(2000, 3, 1, 1, 0, 2536, tti2 // 10, tti2 % 10))  # the 140th UE: 10 free
    b.update(2000, 3, 1, tti2 // 10, tti2 % 10, 0, 1, 0, 2536)
    b.set_now(t + 8 + 5999)
    b.age()                                                # now the pass runs: idle for more than 5 whole seconds -> freed (UE 1000 and 2000: 5.999 s, kept)
    tti3 = (tti2 + 5999) % 10240
    out.append(b.ask(1001, 0, 0, 0, 0, 1000, tti3 // 10, tti3 % 10))  # a UE of the first group: its entity is gone, a new transmission on a fresh one
    b.update(1001, 0, 0, tti3 // 10, tti3 % 10, 0, 0, 0, 1000)
    tti4 = (tti3 + 8) % 10240
    b.set_now(t + 8 + 5999 + 8)
    out.append(b.ask(1001, 0, 0, 0, 2, 1000, tti4 // 10, tti4 % 10))  # its retransmission
    out.append(b.ask(2000, 3, 1, 1, 2, 2536, tti4 // 10, tti4 % 10))  # UE 2000 survived the pass; 6007 subframes later is no retransmission
    return out


def test_product_harq_corner_script_is_the_references():
    want = json.load(open(FIX))["corner_script"]
    assert want[:139] == [NEW_TX] * 139 and want[139:] == [RE_TX, DECODED, NEW_TX, NEW_TX, RE_TX, NEW_TX]
    for cls in (Product,) + ((Reference,) if os.path.exists(REF_SO) else ()):
        b = cls()
        try:
            assert corners(b) == want, cls.__name__
        finally:
            b.close()


def _life_on(cls, seed):
    b = cls()
    try:
        return life(seed, b)
    finally:
        b.close()


@pytest.mark.parametrize("seed", SEEDS)
def test_product_harq_database_gives_the_references_verdicts(seed):
    want = json.load(open(FIX))["lives"][str(seed)]
    got, count = _life_on(Product, seed)
    assert count == want["verdicts_new_retx_full_decoded_busy"] and got == want["digest"]


def test_the_lives_reach_every_verdict():
    lives = json.load(open(FIX))["lives"]
    tot = [sum(v["verdicts_new_retx_full_decoded_busy"][i] for v in lives.values()) for i in range(5)]
    assert tot[NEW_TX] > 50000 and tot[RE_TX] > 5000 and tot[DECODED] > 3000 and tot[FULL] > 1000 and tot[BUSY] == 0
    assert any(v["verdicts_new_retx_full_decoded_busy"][FULL] == 0 for v in lives.values())      # a cell that fits the database


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libref_falcon_harq.so not built (needs /root/reference: make -C oracle -f Makefile.ref)")
@pytest.mark.parametrize("seed", SEEDS)
def test_reference_library_reproduces_the_committed_harq_fixture(seed):
    want = json.load(open(FIX))["lives"][str(seed)]
    got, count = _life_on(Reference, seed)
    assert (got, count) == (want["digest"], want["verdicts_new_retx_full_decoded_busy"])


def test_fixture_was_made_from_the_reference_sources_that_are_here():
    if not os.path.isdir("/root/reference/src/src"):
        pytest.skip("no /root/reference on this host")
    assert _sources_sha() == json.load(open(FIX))["reference_sources_sha256"]


def _sources_sha():
    s = hashlib.sha256()
    for f in ("src/src/HARQ.cc", "src/src/Sniffer_dependency.cc"):
        s.update(open(os.path.join("/root/reference", f), "rb").read())
    return s.hexdigest()


if __name__ == "__main__":   # the fixture generator
    out = {"made_by": "python tests/test_ref_harq.py", "reference_sources_sha256": _sources_sha(), "lives": {}}
    for seed in SEEDS:
        d, c = _life_on(Reference, seed)
        p = _life_on(Product, seed)
        out["lives"][str(seed)] = {"digest": d, "verdicts_new_retx_full_decoded_busy": c, "product_equal_when_made": p == (d, c)}
        print(seed, d, c, p == (d, c))
    r = Reference()
    out["corner_script"] = corners(r)
    r.close()
    print("corner script", out["corner_script"][139:])
    json.dump(out, open(FIX, "w"), indent=1)
