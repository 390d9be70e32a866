"""MCS-tracking database ageing (MCSTracking::update_database_dl, /root/reference/src/src/MCSTracking.cc:850-927, with the look-up /
statistics functions that feed it, :758-848 and :1269-1400; driven every get_interval() x 1000 subframes by
/root/reference/src/src/LTESniffer_Core.cc:473-499).

The product's MCSTracking (ltesniffer_amd/csrc/host/lsn_lte.cc, HIP-free) is driven through the host test glue with random event
sequences and compared, after every event, with a dict-based restatement of the reference's std::map logic written here from the
reference source - with the one deliberate difference the whole repo makes: time is the number of subframes processed (1 ms each)
instead of clock()."""
import ctypes as C
import random

import pytest

from lsn_testlib import hosttest

T64, T256, TUNK, TBOTH, TFULL = 0, 1, 2, 3, 4
MAX_SIZE, RAR_THRESHOLD = 250, 3


class RefModel:
    """tracking_database_dl_mode as the reference keeps it (only the fields that influence a decision)"""

    def __init__(self, interval=5):
        self.db = {}
        self.interval = interval

    def _new(self, now):
        return dict(time=now, table=TUNK, has_rar=False, after_rar=0, active=0, success=0, unsup=0, pinfo=0, other=0)

    def find(self, rnti, now):  # :758-782
        e = self.db.get(rnti)
        if e is None:
            return TUNK if len(self.db) < MAX_SIZE else TFULL
        e["time"] = now
        return e["table"]

    def add(self, rnti, now):  # :784-795
        self.db.setdefault(rnti, self._new(now))

    def update(self, rnti, table, now):  # :797-825
        e = self.db.get(rnti)
        if e is None:
            self.add(rnti, now)
        elif e["has_rar"]:
            if e["after_rar"] > RAR_THRESHOLD:
                e["table"], e["has_rar"] = table, False
            else:
                e["table"] = TUNK
        else:
            e["table"] = table

    def rar(self, rnti, now):  # :827-848
        self.add(rnti, now)
        self.db[rnti]["has_rar"] = True
        self.db[rnti]["table"] = TUNK

    def statistic(self, rnti, fmt, table, en, ok, mimo, now):  # :1269-1384, harq_mode 0
        self.add(rnti, now)
        e = self.db[rnti]
        if fmt > 2 and e["has_rar"]:  # format > 1A
            e["after_rar"] += 1
        if table in (T64, T256, TUNK):
            for i in range(2):
                if en[i]:
                    e["active"] += 1
                if ok[i]:
                    e["success"] += 1
                if en[i] and mimo == -1:
                    e["unsup"] += 1
                elif en[i] and mimo == -2:
                    e["pinfo"] += 1
                elif en[i] and mimo == -3:
                    e["other"] += 1

    def update_database(self, now):  # :850-927
        dele = []
        for rnti in sorted(self.db):
            e = self.db[rnti]
            cur_interval = (now - e["time"]) // 1000
            wrong = e["active"] == 0 or (e["active"] <= 10 and e["success"] == 0 and (e["unsup"] > 0 or e["pinfo"] > 0 or e["other"] > 0))
            if cur_interval > self.interval or wrong or e["active"] == 0:
                dele.append(rnti)
            elif e["success"] / e["active"] < 0.15 and e["table"] != TUNK:
                e["table"] = TUNK
        for r in dele:
            del self.db[r]


def _bind(h):
    h.lsnh_mcs_new.restype = C.c_void_p
    h.lsnh_mcs_free.argtypes = [C.c_void_p]
    h.lsnh_mcs_find.argtypes = [C.c_void_p, C.c_uint16, C.c_uint32]
    h.lsnh_mcs_update.argtypes = [C.c_void_p, C.c_uint16, C.c_int, C.c_uint32]
    h.lsnh_mcs_rar.argtypes = [C.c_void_p, C.c_uint16, C.c_uint32]
    h.lsnh_mcs_stat.argtypes = [C.c_void_p, C.c_uint16] + [C.c_int] * 7 + [C.c_uint32]
    h.lsnh_mcs_update_database.argtypes = [C.c_void_p, C.c_uint32]
    h.lsnh_mcs_count.argtypes = [C.c_void_p]
    h.lsnh_mcs_count.restype = C.c_uint32
    h.lsnh_mcs_peek.argtypes = [C.c_void_p, C.c_uint16]
    return h


def _same(h, m, ref, rntis):
    assert h.lsnh_mcs_count(m) == len(ref.db)
    for r in rntis:
        want = ref.db[r]["table"] if r in ref.db else -1
        assert h.lsnh_mcs_peek(m, r) == want, (r, want)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_database_ageing_matches_the_reference_logic(seed):
    h = _bind(hosttest())
    m = h.lsnh_mcs_new()
    ref = RefModel()
    rng = random.Random(seed)
    pool = [rng.randrange(11, 0xFFF4) for _ in range(400)]  # more RNTIs than max_size: FULL_BUFFER answers and unbounded growth both occur
    now = 0
    try:
        for step in range(30000):
            now += rng.choice((1, 1, 1, 2, 5))
            r = rng.choice(pool[:60]) if rng.random() < 0.7 else rng.choice(pool)  # a busy core of UEs + a long tail that goes idle
            ev = rng.random()
            if ev < 0.45:
                t = ref.find(r, now)
                assert h.lsnh_mcs_find(m, r, now) == t
                en = (1, rng.random() < 0.4)
                ok = tuple(int(e and rng.random() < (0.05 if r % 7 == 0 else 0.9)) for e in en)
                mimo = rng.choice((0, 0, 0, 0, -1, -2, -3)) if r % 11 == 0 else 0
                fmt = rng.choice((1, 2, 6, 7))
                ref.statistic(r, fmt, t, en, ok, mimo, now)
                h.lsnh_mcs_stat(m, r, fmt, t, int(en[0]), int(en[1]), ok[0], ok[1], mimo, now)
            elif ev < 0.75:
                t = rng.choice((T64, T256))
                ref.update(r, t, now)
                h.lsnh_mcs_update(m, r, t, now)
            elif ev < 0.80:
                ref.rar(r, now)
                h.lsnh_mcs_rar(m, r, now)
            if now // 5000 != (now - 5) // 5000 and rng.random() < 0.9:  # roughly every 5000 "subframes"
                ref.update_database(now)
                h.lsnh_mcs_update_database(m, now)
                _same(h, m, ref, pool)
        _same(h, m, ref, pool)
        assert len(ref.db) < len(set(pool)), "the sequence never aged anything out"
    finally:
        h.lsnh_mcs_free(m)


def test_idle_entries_leave_after_more_than_interval_whole_seconds():
    h = _bind(hosttest())
    m = h.lsnh_mcs_new()
    try:
        for r in (100, 200):
            h.lsnh_mcs_find(m, r, 0)
            h.lsnh_mcs_stat(m, r, 2, TUNK, 1, 0, 1, 0, 0, 0)  # active once, success once
        h.lsnh_mcs_find(m, 200, 4000)                          # 200 is looked up again later
        h.lsnh_mcs_update_database(m, 5999)                    # 100 idle for 5.999 s: (5999 - 0) / 1000 = 5, not > 5
        assert h.lsnh_mcs_count(m) == 2
        h.lsnh_mcs_update_database(m, 6000)                    # 6 whole seconds > 5: gone; 200 was seen 2 s ago
        assert h.lsnh_mcs_peek(m, 100) == -1 and h.lsnh_mcs_peek(m, 200) == TUNK
        h.lsnh_mcs_update(m, 200, T256, 6001)
        assert h.lsnh_mcs_peek(m, 200) == T256
        for _ in range(9):                                     # success rate 1 / 10 < 15 %: the learned table is dropped
            h.lsnh_mcs_stat(m, 200, 7, T256, 1, 0, 0, 0, 0, 6002)
        h.lsnh_mcs_update_database(m, 6500)
        assert h.lsnh_mcs_peek(m, 200) == TUNK
        h.lsnh_mcs_rar(m, 300, 6600)                           # an entry that was never active is deleted at the next update
        h.lsnh_mcs_update_database(m, 6700)
        assert h.lsnh_mcs_peek(m, 300) == -1
    finally:
        h.lsnh_mcs_free(m)


# ---------------------------------------------------------------------------------------------------------------------------------------------
# The same database pinned on the REFERENCE'S OWN CODE: /root/reference/src/src/MCSTracking.cc compiled verbatim into oracle/_ref/libref_falcon_mcs.so
# (oracle/Makefile.ref; no srsRAN function is called, clock() is bound to a settable clock of 1 ms per subframe: oracle/ref_shim_search/mcs_glue.cc).
# Random lives of a database - look-ups, statistics with MIMO errors, table updates, random-access responses, ageing passes - go through the
# reference, the product (lsn_lte.cc: MCSTracking) and the Python model above; what each answers (every look-up, and after every ageing pass the
# population and every entry's table) is reduced to a digest.  tests/golden/mcs_tracking_ref.json holds the reference's digests
# (tests/golden/make_mcs_fixture.py); the library itself runs again where it is present.
import hashlib
import json
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MCS_REF_SO = os.path.join(_ROOT, "oracle", "_ref", "libref_falcon_mcs.so")
MCS_FIX = os.path.join(_ROOT, "tests", "golden", "mcs_tracking_ref.json")
LIFE_SEEDS = (11, 12, 13, 14, 15, 16)


class _Product:
    name = "product"

    def __init__(self):
        self.h = _bind(hosttest())
        self.m = self.h.lsnh_mcs_new()
        self.now = 0

    def set_now(self, now): self.now = now
    def find(self, r): return self.h.lsnh_mcs_find(self.m, r, self.now)
    def stat(self, r, fmt, t, en, ok, mimo): self.h.lsnh_mcs_stat(self.m, r, fmt, t, int(en[0]), int(en[1]), int(ok[0]), int(ok[1]), mimo, self.now)
    def update(self, r, t): self.h.lsnh_mcs_update(self.m, r, t, self.now)
    def rar(self, r): self.h.lsnh_mcs_rar(self.m, r, self.now)
    def update_database(self): self.h.lsnh_mcs_update_database(self.m, self.now)
    def count(self): return self.h.lsnh_mcs_count(self.m)
    def peek(self, r): return self.h.lsnh_mcs_peek(self.m, r)
    def close(self): self.h.lsnh_mcs_free(self.m)


class _Model:
    name = "python model"

    def __init__(self):
        self.m = RefModel()
        self.now = 0

    def set_now(self, now): self.now = now
    def find(self, r): return self.m.find(r, self.now)
    def stat(self, r, fmt, t, en, ok, mimo): self.m.statistic(r, fmt, t, en, ok, mimo, self.now)
    def update(self, r, t): self.m.update(r, t, self.now)
    def rar(self, r): self.m.rar(r, self.now)
    def update_database(self): self.m.update_database(self.now)
    def count(self): return len(self.m.db)
    def peek(self, r): return self.m.db[r]["table"] if r in self.m.db else -1
    def close(self): pass


class _Reference:
    name = "reference"

    def __init__(self):
        L = self.lib = C.CDLL(MCS_REF_SO)
        L.ref_mcs_new.restype = C.c_void_p
        L.ref_mcs_new.argtypes = [C.c_int]
        L.ref_mcs_free.argtypes = [C.c_void_p]
        L.ref_mcs_set_now_ms.argtypes = [C.c_uint64]
        L.ref_mcs_find.argtypes = [C.c_void_p, C.c_uint16]
        L.ref_mcs_update.argtypes = [C.c_void_p, C.c_uint16, C.c_int]
        L.ref_mcs_rar.argtypes = [C.c_void_p, C.c_uint16]
        L.ref_mcs_stat.argtypes = [C.c_void_p, C.c_uint16] + [C.c_int] * 7 + [C.c_uint32] * 2
        L.ref_mcs_update_database.argtypes = [C.c_void_p]
        L.ref_mcs_count.argtypes = [C.c_void_p]
        L.ref_mcs_count.restype = C.c_uint32
        L.ref_mcs_peek.argtypes = [C.c_void_p, C.c_uint16, C.c_void_p]
        L.ref_mcs_get_ue_config.argtypes = [C.c_void_p, C.c_uint16, C.c_void_p]
        self.m = L.ref_mcs_new(0)

    def set_now(self, now): self.lib.ref_mcs_set_now_ms(now)
    def find(self, r): return self.lib.ref_mcs_find(self.m, r)
    def stat(self, r, fmt, t, en, ok, mimo): self.lib.ref_mcs_stat(self.m, r, fmt, t, int(en[0]), int(en[1]), int(ok[0]), int(ok[1]), mimo, 5, 9)
    def update(self, r, t): self.lib.ref_mcs_update(self.m, r, t)
    def rar(self, r): self.lib.ref_mcs_rar(self.m, r)
    def update_database(self): self.lib.ref_mcs_update_database(self.m)
    def count(self): return self.lib.ref_mcs_count(self.m)
    def peek(self, r): return self.lib.ref_mcs_peek(self.m, r, None)
    def close(self): self.lib.ref_mcs_free(self.m)


def life(seed, b, steps=40000):
    """one life of a database on back-end b -> (digest of everything it answered, counters)"""
    rng = random.Random(seed)
    pool = [rng.randrange(11, 0xFFF4) for _ in range(300 + 40 * (seed % 4))]
    core = pool[:40 + 10 * (seed % 3)]
    h = hashlib.sha256()
    now, passes, aged, full, after_rar = 0, 0, 0, 0, 0
    for step in range(steps):
        now += rng.choice((1, 1, 1, 2, 5, 30 if seed % 2 else 1))
        b.set_now(now)
        r = rng.choice(core) if rng.random() < 0.7 else rng.choice(pool)
        ev = rng.random()
        if ev < 0.45:
            t = b.find(r)
            h.update(b"f%d:%d;" % (r, t))
            full += t == TFULL
            en = (1, rng.random() < 0.4)
            ok = tuple(int(e and rng.random() < (0.05 if r % 7 == 0 else 0.9)) for e in en)
            mimo = rng.choice((0, 0, 0, 0, -1, -2, -3)) if r % 11 == 0 else 0
            b.stat(r, rng.choice((1, 2, 6, 7)), t if t != TFULL else T64, en, ok, mimo)
        elif ev < 0.75:
            b.update(r, rng.choice((T64, T256)))
        elif ev < 0.80:
            b.rar(r)
            after_rar += 1
        if now // 5000 != (now - 30) // 5000 and rng.random() < 0.5:
            before = b.count()
            b.update_database()
            passes += 1
            aged += before - b.count()
            h.update(b"u%d:" % b.count())
            h.update(",".join("%d" % b.peek(x) for x in pool).encode())
    h.update(b"e%d:" % b.count())
    h.update(",".join("%d" % b.peek(x) for x in pool).encode())
    return h.hexdigest()[:32], dict(ageing_passes=passes, entries_aged_out=aged, full_buffer_answers=full, rar=after_rar, population_at_the_end=b.count())


def corners(b):
    """a scripted life for what random lives hit too rarely: a success rate of EXACTLY 15 % (kept: the reference drops a table below 15 %), a success flag on a
    disabled transport block (counted by the reference), exactly rar_thresold messages behind a RAR (not enough), an entry idle for exactly the interval"""
    out = []
    b.set_now(10)
    for r in (500, 501, 502, 503):
        out.append(b.find(r))
    b.update(500, T256); b.update(500, T256)            # first call creates the entry (unknown), second sets the table
    for i in range(20):                                   # 20 decodes, 3 good: 15 %
        b.stat(500, 7, T256, (1, 0), (1 if i < 3 else 0, 0), 0)
    b.update(501, T64); b.update(501, T64)
    for i in range(20):                                   # 20 decodes, 2 good on the enabled block + 1 success flag on the disabled one: 15 % only if that one counts
        b.stat(501, 7, T64, (1, 0), (1 if i < 2 else 0, 1 if i == 5 else 0), 0)
    b.update(502, T64); b.update(502, T64)
    for i in range(20):                                   # 10 %: dropped
        b.stat(502, 7, T64, (1, 0), (1 if i < 2 else 0, 0), 0)
    b.rar(503)
    for i in range(3):                                    # exactly rar_thresold = 3 messages above format 1A behind the RAR
        b.stat(503, 7, TUNK, (1, 0), (1, 0), 0)
    b.update(503, T256)
    out += [b.peek(r) for r in (500, 501, 502, 503)]
    b.stat(503, 7, TUNK, (1, 0), (1, 0), 0)              # the fourth
    b.update(503, T256)
    out.append(b.peek(503))
    b.set_now(2000)
    b.update_database()
    out += [b.count()] + [b.peek(r) for r in (500, 501, 502, 503)]
    out += [b.find(r) for r in (500, 501, 502, 503)]      # look-ups refresh the entries' time: 2000
    b.set_now(3000)
    out.append(b.find(500))                               # 500 once more, a second later
    b.set_now(2000 + 5999)
    b.update_database()                                   # idle for 5.999 s: 5 whole seconds, not more than the interval - all stay
    out += [b.count()]
    b.set_now(2000 + 6000)
    b.update_database()                                   # 6 whole seconds: gone, except 500 (idle for exactly 5)
    out += [b.count()] + [b.peek(r) for r in (500, 501, 502, 503)]
    return out


def test_product_database_corner_script_is_the_references():
    fix = json.load(open(MCS_FIX))
    for cls in (_Product, _Model) + ((_Reference,) if os.path.exists(MCS_REF_SO) else ()):
        b = cls()
        try:
            assert corners(b) == fix["corner_script"], cls.name
        finally:
            b.close()


def _life_on(cls, seed):
    b = cls()
    try:
        return life(seed, b)
    finally:
        b.close()


@pytest.mark.parametrize("seed", LIFE_SEEDS)
def test_product_database_answers_like_the_references_own_mcs_tracking(seed):
    fix = json.load(open(MCS_FIX))
    want = fix["lives"][str(seed)]
    got, info = _life_on(_Product, seed)
    assert info == want["counters"], "the product's database ages / fills differently from the reference's MCSTracking.cc"
    assert got == want["digest"]
    assert _life_on(_Model, seed)[0] == want["digest"]   # and so does the Python model the tests above are written against


def test_the_lives_reach_the_corners():
    fix = json.load(open(MCS_FIX))
    c = [v["counters"] for v in fix["lives"].values()]
    assert all(x["entries_aged_out"] > 100 and x["ageing_passes"] > 10 and x["rar"] > 1000 for x in c)
    assert any(x["full_buffer_answers"] > 0 for x in c), "no life filled the 250 entries"
    assert fix["default_ue_config_of_an_unknown_rnti"] == [0, 10, 8, 11, 2, 0]   # p_a bits, I_offset ack / cqi / ri, CQI type (higher-layer sub-band), has_ue_config


@pytest.mark.skipif(not os.path.exists(MCS_REF_SO), reason="oracle/_ref/libref_falcon_mcs.so not built (needs /root/reference: make -C oracle -f Makefile.ref)")
@pytest.mark.parametrize("seed", LIFE_SEEDS)
def test_reference_library_reproduces_the_committed_mcs_fixture(seed):
    fix = json.load(open(MCS_FIX))
    got, info = _life_on(_Reference, seed)
    assert (got, info) == (fix["lives"][str(seed)]["digest"], fix["lives"][str(seed)]["counters"])


def test_default_ue_configuration_is_the_references():
    fix = json.load(open(MCS_FIX))
    h = _bind(hosttest())
    m = h.lsnh_mcs_new()
    out = (C.c_uint32 * 6)()
    h.lsnh_mcs_get.argtypes = [C.c_void_p, C.c_uint16, C.c_void_p]
    h.lsnh_mcs_get(m, 0x1234, out)
    h.lsnh_mcs_free(m)
    assert list(out) == fix["default_ue_config_of_an_unknown_rnti"]
    if os.path.exists(MCS_REF_SO):
        r = _Reference()
        r.lib.ref_mcs_get_ue_config(r.m, 0x1234, out)
        r.close()
        assert list(out) == fix["default_ue_config_of_an_unknown_rnti"]
