"""Operation programs for the FALCON RNTI manager and three things that can run them (test infrastructure):

  * the REFERENCE's own RNTIManager.cc / Histogram.cc / Interval.cc, compiled from /root/reference by oracle/Makefile.ref into
    oracle/_ref/libref_falcon_util.so (its C wrapper: /root/reference/lib/include/falcon/util/rnti_manager_c.h:46-66);
  * the oracle's restatement (oracle/o_falcon.c: o_rntiman_*);
  * the product's host class (ltesniffer_amd/csrc/host/lsn_lte.cc: RNTIManager) through the test glue tests/native/lsn_hosttest.cc.

A program is a list of tuples; `run(backend, program)` returns the list of every value an operation handed back.  Programs are generated from a
seed (numpy Generator, PCG64: the same numbers on every host) and look like what the path does with the manager - per subframe a handful of
candidates from a pool of UEs and from noise, validations, RAR / shortcut activations, one time step - plus what it rarely does: more
candidates than a subframe's budget, idle gaps longer than the 10 000-step lifetime, RNTIs inside evergreen / forbidden intervals, format changes.

`getActiveSet` / `printActiveSet` are left out on purpose: RNTIManager.cc:239-241 takes the manager's non-recursive mutex and calls cleanExpired(),
which takes it again (:411) - the call does not return (it has no call site in the reference: LTESniffer_Core.cc:556-559 are comments)."""
import ctypes as C
import hashlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_falcon_util.so")
RNTI_PER_SUBFRAME = 304 // 5   # RNTIManager.h:48, PhyCommon.cc:11


class _Backend:
    """ops every backend offers: ev fb cand vr act step freq reason isfb; `extended` ones also run val (validate without refresh) and isev"""
    extended = False

    def run(self, program):
        out = []
        h = None
        try:
            for op in program:
                k = op[0]
                if k == "new":
                    h = self.new(op[1], op[2], op[3])
                elif k == "ev":
                    self.add_evergreen(h, op[1], op[2], op[3])
                elif k == "fb":
                    self.add_forbidden(h, op[1], op[2], op[3])
                elif k == "cand":
                    self.add_candidate(h, op[1], op[2])
                elif k == "vr":
                    out.append(int(self.validate_and_refresh(h, op[1], op[2])))
                elif k == "act":
                    self.activate_and_refresh(h, op[1], op[2], op[3])
                elif k == "step":
                    self.step_time(h, op[1])
                elif k == "freq":
                    out.append(int(self.get_frequency(h, op[1], op[2])))
                elif k == "reason":
                    out.append(int(self.get_activation_reason(h, op[1])))
                elif k == "isfb":
                    out.append(int(self.is_forbidden(h, op[1], op[2])))
                elif k == "val":   # only in extended programs (validate() has side effects: it activates by histogram and drops an expired RNTI)
                    out.append(int(self.validate(h, op[1], op[2])))
                elif k == "isev":
                    out.append(int(self.is_evergreen(h, op[1], op[2])))
                else:
                    raise ValueError(k)
        finally:
            if h is not None:
                self.free(h)
        return out


def _bind(lib, table):
    for name, res, args in table:
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = args


class Reference(_Backend):
    """the reference's own code"""
    name = "reference (oracle/_ref)"
    extended = True

    def __init__(self, path=REF_SO):
        lib = C.CDLL(path)
        V, U16, U32, I = C.c_void_p, C.c_uint16, C.c_uint32, C.c_int
        _bind(lib, [("rnti_manager_create", V, [U32, U32, U32]), ("rnti_manager_free", None, [V]),
                    ("rnti_manager_add_evergreen", None, [V, U16, U16, U32]), ("rnti_manager_add_forbidden", None, [V, U16, U16, U32]),
                    ("rnti_manager_add_candidate", None, [V, U16, U32]), ("rnti_manager_validate", I, [V, U16, U32]),
                    ("rnti_manager_validate_and_refresh", I, [V, U16, U32]), ("rnti_manager_activate_and_refresh", None, [V, U16, U32, I]),
                    ("rnti_manager_is_evergreen", I, [V, U16, U32]), ("rnti_manager_is_forbidden", I, [V, U16, U32]),
                    ("rnti_manager_step_time_multi", None, [V, U32]), ("rnti_manager_getFrequency", U32, [V, U16, U32]),
                    ("rnti_manager_get_activation_reason", I, [V, U16])])
        self.new, self.free = lib.rnti_manager_create, lib.rnti_manager_free
        self.add_evergreen, self.add_forbidden, self.add_candidate = lib.rnti_manager_add_evergreen, lib.rnti_manager_add_forbidden, lib.rnti_manager_add_candidate
        self.validate, self.validate_and_refresh = lib.rnti_manager_validate, lib.rnti_manager_validate_and_refresh
        self.activate_and_refresh = lib.rnti_manager_activate_and_refresh
        self.is_evergreen, self.is_forbidden = lib.rnti_manager_is_evergreen, lib.rnti_manager_is_forbidden
        self.step_time, self.get_frequency, self.get_activation_reason = lib.rnti_manager_step_time_multi, lib.rnti_manager_getFrequency, lib.rnti_manager_get_activation_reason


class Oracle(_Backend):
    name = "oracle (o_falcon.c)"

    def __init__(self):
        from lsn_testlib import oracle
        lib = oracle()
        V, U16, U32, I = C.c_void_p, C.c_uint16, C.c_uint32, C.c_int
        _bind(lib, [("o_rntiman_new", V, [U32, U32, U32]), ("o_rntiman_free", None, [V]), ("o_rntiman_add_evergreen", None, [V, U16, U16, U32]),
                    ("o_rntiman_add_forbidden", None, [V, U16, U16, U32]), ("o_rntiman_add_candidate", None, [V, U16, U32]),
                    ("o_rntiman_validate_and_refresh", I, [V, U16, U32]), ("o_rntiman_activate_and_refresh", None, [V, U16, U32, I]),
                    ("o_rntiman_is_forbidden", I, [V, U16, U32]), ("o_rntiman_get_frequency", U32, [V, U16, U32]),
                    ("o_rntiman_get_activation_reason", I, [V, U16]), ("o_rntiman_step_time", None, [V])])
        self.new, self.free = lib.o_rntiman_new, lib.o_rntiman_free
        self.add_evergreen, self.add_forbidden, self.add_candidate = lib.o_rntiman_add_evergreen, lib.o_rntiman_add_forbidden, lib.o_rntiman_add_candidate
        self.validate_and_refresh, self.activate_and_refresh = lib.o_rntiman_validate_and_refresh, lib.o_rntiman_activate_and_refresh
        self.is_forbidden, self.get_frequency, self.get_activation_reason = lib.o_rntiman_is_forbidden, lib.o_rntiman_get_frequency, lib.o_rntiman_get_activation_reason
        self._step = lib.o_rntiman_step_time

    def step_time(self, h, n):
        for _ in range(n):
            self._step(h)


class Product(_Backend):
    name = "product (lsn_lte.cc)"
    extended = True

    def __init__(self):
        from lsn_testlib import hosttest
        lib = hosttest()
        V, U16, U32, I = C.c_void_p, C.c_uint16, C.c_uint32, C.c_int
        _bind(lib, [("lsnh_rm_new", V, [U32, U32, U32]), ("lsnh_rm_free", None, [V]), ("lsnh_rm_add_evergreen", None, [V, U16, U16, U32]),
                    ("lsnh_rm_add_forbidden", None, [V, U16, U16, U32]), ("lsnh_rm_add_candidate", None, [V, U16, U32]),
                    ("lsnh_rm_validate", I, [V, U16, U32]), ("lsnh_rm_validate_and_refresh", I, [V, U16, U32]),
                    ("lsnh_rm_activate_and_refresh", None, [V, U16, U32, I]), ("lsnh_rm_is_evergreen", I, [V, U16, U32]),
                    ("lsnh_rm_is_forbidden", I, [V, U16, U32]), ("lsnh_rm_step_time", None, [V, U32]), ("lsnh_rm_get_frequency", U32, [V, U16, U32]),
                    ("lsnh_rm_get_activation_reason", I, [V, U16])])
        self.new, self.free = lib.lsnh_rm_new, lib.lsnh_rm_free
        self.add_evergreen, self.add_forbidden, self.add_candidate = lib.lsnh_rm_add_evergreen, lib.lsnh_rm_add_forbidden, lib.lsnh_rm_add_candidate
        self.validate, self.validate_and_refresh = lib.lsnh_rm_validate, lib.lsnh_rm_validate_and_refresh
        self.activate_and_refresh = lib.lsnh_rm_activate_and_refresh
        self.is_evergreen, self.is_forbidden = lib.lsnh_rm_is_evergreen, lib.lsnh_rm_is_forbidden
        self.step_time, self.get_frequency, self.get_activation_reason = lib.lsnh_rm_step_time, lib.lsnh_rm_get_frequency, lib.lsnh_rm_get_activation_reason


# the cases of the fixture: (seed, formats, candidates per step and format, histogram threshold, steps)
CASES = [(0, 9, RNTI_PER_SUBFRAME, 5, 2500), (1, 9, RNTI_PER_SUBFRAME, 5, 2500), (2, 9, RNTI_PER_SUBFRAME, 0, 1500), (3, 9, RNTI_PER_SUBFRAME, 10, 2500),
         (4, 3, 7, 2, 4000), (5, 6, 20, 3, 2500), (6, 9, RNTI_PER_SUBFRAME, 5, 1200), (7, 2, RNTI_PER_SUBFRAME, 1, 1500)]


def program(seed, nformats, maxcand, threshold, steps, extended=False):
    """one manager's life as a list of operations; extended = with the two operations the oracle's interface does not have (val, isev)"""
    g = np.random.Generator(np.random.PCG64(1000 + seed))
    ops = [("new", nformats, maxcand, threshold)]
    f1a, f1c = 1 % nformats, min(3, nformats - 1)
    # what the path configures (PhyCommon.cc:13-24): RA-RNTIs, P-RNTI and SI-RNTI are evergreen for the two compact formats, RNTI 0 is forbidden everywhere
    for f in sorted({f1a, f1c}):
        ops += [("ev", 1, 10, f), ("ev", 0xFFFE, 0xFFFF, f)]
    for f in range(nformats):
        ops.append(("fb", 0, 0, f))
    if seed % 2:  # an extra forbidden band that overlaps an evergreen one for one format (evergreen is consulted first, RNTIManager.cc:157-163)
        ops += [("fb", 8, 40, f1a), ("fb", 0xFFF4, 0xFFFD, nformats - 1)]
    pool = [int(x) for x in g.choice(np.arange(11, 0xFFF4), size=48, replace=False)]
    pref = [int(g.integers(1, nformats)) if nformats > 1 else 0 for _ in pool]
    alive = list(range(12))
    recent = []

    def probe():
        for i in g.choice(len(pool), size=6, replace=False):
            r = pool[int(i)]
            ops.append(("reason", r))
            ops.append(("freq", r, 0))
            ops.append(("freq", r, pref[int(i)]))
        ops.append(("freq", 0, int(g.integers(0, nformats))))  # the padding entries (ILLEGAL_RNTI, RNTIManager.cc:424-428)

    for t in range(steps):
        burst = int(g.integers(0, 14))
        if g.random() < 0.02:
            burst = maxcand + int(g.integers(1, 9))  # more candidates than the step's budget: remainingCandidates goes negative, no padding
        for _ in range(burst):
            u = g.random()
            if u < 0.72 and alive:
                i = alive[int(g.integers(0, len(alive)))]
                r = pool[i]
                v = g.random()
                f = 0 if v < 0.3 else (pref[i] if v < 0.93 else int(g.integers(0, nformats)))
            elif u < 0.80:
                r, f = int(g.choice([0, 1, 5, 10, 11, 0xFFF3, 0xFFF4, 0xFFFD, 0xFFFE, 0xFFFF, 9, 39, 40, 41])), int(g.integers(0, nformats))
            else:
                r, f = int(g.integers(0, 65536)), int(g.integers(0, nformats))
            ops.append(("cand", r, f))
            recent.append((r, f))
            if g.random() < 0.55:
                ops.append(("vr", r, f))
            elif extended and g.random() < 0.3:
                ops.append(("val", r, f))
        recent[:] = recent[-40:]
        for _ in range(int(g.integers(0, 4))):  # a validation of something seen a while ago, perhaps in another format
            if recent:
                r, f = recent[int(g.integers(0, len(recent)))]
                ops.append(("vr", r, f if g.random() < 0.7 else int(g.integers(0, nformats))))
        if g.random() < 0.03:   # a random access response: the temporary C-RNTI is activated at once (DL_Sniffer_PDSCH.cc:782-797)
            i = int(g.integers(0, len(pool)))
            ops.append(("act", pool[i], 0, 2))
            if i not in alive:
                alive.append(i)
        if g.random() < 0.02:   # shortcut discovery (DCISearch.cc:300-340): active with its format known
            i = int(g.integers(0, len(pool)))
            ops.append(("act", pool[i], pref[i], 3))
        if g.random() < 0.01:   # an activation over an active RNTI keeps the first reason (RNTIManager.cc:386-392)
            ops.append(("act", pool[alive[0]] if alive else 77, int(g.integers(0, nformats)), int(g.integers(1, 6))))
        if g.random() < 0.015 and len(alive) > 3:   # a UE leaves
            alive.pop(int(g.integers(0, len(alive))))
        if g.random() < 0.02 and len(alive) < len(pool):
            alive.append(int(g.choice([i for i in range(len(pool)) if i not in alive])))
        if g.random() < 0.01 and nformats > 2:   # a UE changes its transmission mode
            i = int(g.integers(0, len(pool)))
            pref[i] = 1 + (pref[i] % (nformats - 1))
        if t % 50 == 49:
            probe()
            ops += [("isfb", int(g.integers(0, 65536)), int(g.integers(0, nformats))), ("isfb", int(g.choice([0, 8, 40, 41, 0xFFF4, 0xFFFD])), int(g.integers(0, nformats)))]
            if extended:
                ops.append(("isev", int(g.choice([1, 10, 11, 0xFFFE, 0xFFFF, 500])), int(g.integers(0, nformats))))
        gap = 1
        u = g.random()
        if u < 0.004:
            gap = 10000 + int(g.integers(-2, 3))   # around the lifetime: 9 998 .. 10 002 steps without a sign of life (RNTIManager.cc:399-407)
        elif u < 0.01:
            gap = int(g.integers(2, 400))
        ops.append(("step", gap))
        if gap > 9000:
            for i in range(len(pool)):   # who is still valid, in the uplink format and in its own
                ops.append(("vr", pool[i], 0 if i % 2 else pref[i]))
            probe()
    probe()
    return ops


def digest(values):
    return hashlib.sha256(np.asarray(values, dtype=np.int64).tobytes()).hexdigest()[:32]
