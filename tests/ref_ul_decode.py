"""Driver of the uplink decode-control pin: the REFERENCE'S OWN PUSCH_Decoder::decode (UL_Sniffer_PUSCH.cc compiled verbatim into
oracle/_ref/libref_falcon_ul_decode.so, oracle/ref_shim_search/ul_decode_glue.cc) and the oracle's restatement (o_worker.c: decode_pusch) walk the same
scripted lives - schedules of uplink grants per subframe - with ONE scripted uplink decoder answering both: which attempts are made (which MCS table, which
modulation, which UCI layout) in which order, which blocks are written, what the tracking database has learnt, what the ageing keeps.  Test infrastructure."""
import ctypes as C
import hashlib
import os
import random

from lsn_testlib import OracleWorker, oracle, parse_pcap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_falcon_ul_decode.so")
REF_SOURCES = ("src/src/UL_Sniffer_PUSCH.cc", "src/src/SubframePower.cc", "src/src/MCSTracking.cc", "src/include/UL_Sniffer_PUSCH.h", "src/include/MCSTracking.h")
VALID_PRB = [n for n in range(1, 101) if all(n % p ** k == 0 for p, k in ()) and (lambda m: [m := m // p for p in (2, 3, 5) for _ in range(8) if m % p == 0] and False or True)(n)]


def _valid(n):
    for p in (2, 3, 5):
        while n % p == 0:
            n //= p
    return n == 1


VALID_PRB = [n for n in range(1, 101) if _valid(n)]
UL_SCRIPT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_uint8))

# lives: (name, nof_prb, nof subframes, seed, UEs, grants per subframe up to, share of failing decodes %)
LIVES = [("mix_50prb", 50, 4000, 11, 40, 3, 15), ("quiet_100prb", 100, 10000, 12, 24, 2, 30), ("crowd_25prb_full_buffer", 25, 7000, 13, 330, 4, 10),
         ("flaky_75prb", 75, 3500, 14, 60, 3, 55), ("six_prb", 6, 2500, 15, 12, 2, 20)]


def _mix(*v):
    h = 0x9E3779B97F4A7C15
    for x in v:
        h = ((h ^ (int(x) & 0xFFFFFFFFFFFF)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        h ^= h >> 29
    return h


class ScriptedUplinkDecoder:
    """truth per UE: what its transmitter really uses - 0: 16QAM maximum (MCS 21-28 of Table 8.6.1-1 sent with 16QAM), 1: 64QAM, 2: the 256QAM table.  An attempt
    passes when its transport block size and modulation are the transmitter's (and a seeded coin allows it); the estimator's SNR is a seeded value per attempt"""

    def __init__(self, seed, truth, fail_pct, by_grant):
        self.seed, self.truth, self.fail, self.by_grant, self.log = seed, truth, fail_pct, by_grant, []
        self.fn = UL_SCRIPT(self._cb)
        self.reference_side = False

    def _cb(self, user, call, snr, payload):
        w = [int(call[i]) for i in range(16)]
        tti, rnti, qm, tbs = w[0], w[1], w[5], w[6]
        # word 15 of the normalised call: the number of sub-bands N of a higher-layer sub-band report (type 2) - what the reference CONFIGURES (cqi.N =
        # ul_sniffer_cqi_hl_get_no_subbands, :443); how many bits srsran_cqi_size makes of it is srsRAN's business (the oracle says 4 + 2 N, and 0 when the band
        # has no sub-bands - up to 7 PRB)
        if self.reference_side:   # the reference hands over the grant's modulation and enable_64qam; srsRAN's decoder caps the modulation at 16QAM without the flag
            qm = qm if w[7] else min(qm, 4)
            w = w[:5] + [qm, tbs, 0, w[8], w[9], w[10] if w[9] else 0, w[11], w[12], w[13], w[14], w[15] if (w[9] and w[10] == 2) else 0]
        else:
            w = w[:7] + [0, w[8], w[9], w[10] if w[9] else 0] + w[11:15] + [(w[15] - 4) // 2 if (w[9] and w[10] == 2 and w[15] >= 4) else 0]
        g = self.by_grant.get((tti, rnti))
        t = self.truth.get(rnti, 1)
        ok = False
        if g is not None:
            mcs, mod, tbs64, mod256, tbs256 = g
            want = (tbs256, mod256) if t == 2 else (tbs64, min(mod, 4) if t == 0 else mod)
            ok = (tbs, qm) == want and tbs > 0
        coin = _mix(self.seed, tti, rnti, tbs, qm) % 100
        crc = 1 if ok and coin >= self.fail else 0
        s = (_mix(self.seed, 7, tti, rnti, tbs, qm) % 2400) / 100.0 - 3.0   # -3 ... 21 dB: some attempts leave the estimate under the 1 dB gate of the statistics
        snr[0] = s
        if crc:
            for i in range(tbs // 8):
                payload[i] = _mix(self.seed, tti, rnti, i) & 255
        self.log.append((tuple(w), crc, round(s, 2)))
        return crc


def script(life):
    name, nprb, nsf, seed, nue, per_sf, fail = life
    rng = random.Random(seed)
    valid = [n for n in VALID_PRB if n <= nprb]
    ues = sorted(rng.sample(range(0x100, 0xFFF0), nue))
    truth = {r: rng.choice((0, 1, 1, 2)) for r in ues}
    ev, by_grant = [], {}
    active = set(rng.sample(ues, max(2, nue // 3)))
    for k in range(nsf):
        tti = (k + 37) % 10240
        if k and k % 1000 == 0:
            ev.append(("age", k))
            # the population moves: some UEs fall silent (the ageing drops them), others arrive
            for r in rng.sample(sorted(active), max(1, len(active) // 4)):
                active.discard(r)
            for r in rng.sample(ues, max(1, nue // 6)):
                active.add(r)
        if rng.random() < 0.02:
            r = rng.choice(ues)
            ev.append(("cfg", r, rng.randrange(16), rng.randrange(16), rng.randrange(13), rng.randrange(3)))
        ent = []
        for _ in range(rng.randrange(per_sf + 1)):
            r = rng.choice(sorted(active)) if rng.random() < 0.97 else rng.choice((0, 5, 0xFFF5))
            if any(e[0] == r for e in ent):
                continue
            is_rar = 1 if rng.random() < 0.06 else 0
            mcs = rng.choice((rng.randrange(0, 11), rng.randrange(11, 21), rng.randrange(21, 29), rng.randrange(21, 29), rng.randrange(29, 32))) if not is_rar else rng.randrange(0, 8)
            L = rng.choice(valid) if rng.random() < 0.93 else rng.choice([n for n in range(1, nprb + 1) if n not in valid] or [7])
            n_prb = rng.randrange(0, nprb - L + 1) if L <= nprb else 0
            mod = 2 if mcs < 11 else 4 if mcs < 21 else 6
            # (a RAR entry whose grant conversion failed - size 0 - is still handed to srsRAN by the reference, :419 lets every RAR entry through, and refused there; the
            # oracle and the product do not make that call.  Its only trace would be the estimator's SNR, and a failed conversion has no valid PRB count for the
            # estimator either: the lives keep RAR entries decodable)
            tbs = 0 if ((mcs >= 29 and rng.random() < 0.5) or (rng.random() < 0.03 and not is_rar)) else 8 * (3 + (_mix(seed, mcs, L) % 300))   # (MCS 29-31: a retransmission takes the size of the grant before it - or has none)
            L256 = L if rng.random() < 0.9 else rng.choice((0, 110, 120))
            mod256 = 2 if mcs < 6 else 4 if mcs < 14 else 6 if mcs < 23 else 8
            tbs256 = 0 if (tbs == 0 and mcs >= 29) or (rng.random() < 0.03 and not is_rar) else tbs + 8 * (1 + mcs % 5)
            ent.append((r, is_rar, mcs, L, n_prb, mod, tbs, L256, mod256, tbs256, 1 if rng.random() < 0.2 else 0, rng.randrange(3)))
            by_grant[(tti, r)] = (mcs, mod, tbs, mod256, tbs256)
        ev.append(("sf", tti, ent))
    return ues, truth, ev, by_grant


class Reference:
    name = "reference"

    def __init__(self):
        self.lib = C.CDLL(REF_SO)
        L = self.lib
        L.ref_ul_set_script.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_ul_new.restype = C.c_void_p
        L.ref_ul_new.argtypes = [C.c_uint32, C.c_uint32]
        L.ref_ul_free.argtypes = [C.c_void_p]
        L.ref_ul_set_last_snr.argtypes = [C.c_void_p, C.c_float]
        L.ref_ul_decode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.ref_decode_records.restype = C.c_uint32
        L.ref_decode_records.argtypes = [C.c_void_p, C.c_uint32]
        L.ref_ul_tracked.argtypes = [C.c_void_p, C.c_uint16]
        L.ref_ul_nof_tracked.restype = C.c_uint32
        L.ref_ul_nof_tracked.argtypes = [C.c_void_p]
        L.ref_ul_update_database.argtypes = [C.c_void_p]
        L.ref_ul_set_ue_config.argtypes = [C.c_void_p, C.c_uint16] + [C.c_uint32] * 4
        L.ref_collect_set_now_ms.argtypes = [C.c_uint64]

    def open(self, life, dec):
        dec.reference_side = True
        self.lib.ref_ul_set_script(C.cast(dec.fn, C.c_void_p), None)
        self.lib.ref_collect_set_now_ms(0)
        self.h = self.lib.ref_ul_new(life[1], 1)
        self.lib.ref_ul_set_last_snr(self.h, 0.0)

    def close(self):
        self.lib.ref_ul_free(self.h)

    def now(self, k):
        self.lib.ref_collect_set_now_ms(k)

    def age(self):
        self.lib.ref_ul_update_database(self.h)
        return self.lib.ref_ul_nof_tracked(self.h)

    def cfg(self, r, a, c, ri, t):
        self.lib.ref_ul_set_ue_config(self.h, r, a, c, ri, t)

    def subframe(self, tti, ent):
        flat = (C.c_uint32 * (12 * max(len(ent), 1)))(*[v for e in ent for v in e])
        self.lib.ref_ul_decode(self.h, tti, len(ent), flat)
        buf = (C.c_uint32 * (7 * 64))()
        n = self.lib.ref_decode_records(buf, 64)
        return [(buf[7 * i + 2], buf[7 * i + 3], buf[7 * i + 5] | (buf[7 * i + 6] << 32)) for i in range(min(n, 64)) if buf[7 * i] == 7]

    def state(self, rntis):
        code = {0: 2, 1: 3, 2: 4, 3: 1, 4: 6, 5: 5}   # ul_sniffer_mod_tracking_t (falcon_dci.h:115-122) -> the oracle's codes: 1 unknown, 2 / 3 / 4 = 16 / 64 / 256QAM maximum, 5 full buffer
        return (self.lib.ref_ul_nof_tracked(self.h), [(r, code[self.lib.ref_ul_tracked(self.h, r)]) for r in rntis])


def _fnv(b):
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


class Oracle:
    name = "oracle"

    def __init__(self):
        o = self.o = oracle()
        o.o_worker_set_ul_script.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        o.o_worker_set_last_ul_snr.argtypes = [C.c_void_p, C.c_float]
        o.o_worker_ul_decode_probe.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        o.o_worker_ul_update_database.argtypes = [C.c_void_p]
        o.o_worker_ul_set_ue_config.argtypes = [C.c_void_p, C.c_uint16] + [C.c_uint32] * 4
        o.o_worker_nof_tracked_ul.restype = C.c_uint32
        o.o_worker_nof_tracked_ul.argtypes = [C.c_void_p]
        o.o_worker_tracked_mod_ul.argtypes = [C.c_void_p, C.c_uint16]

    def open(self, life, dec):
        from lsn_testlib import OracleWorkerUl
        dec.reference_side = False
        self.w = OracleWorkerUl(life[1], 1, 1, cyclic_shift=0, delta_ss=0)
        self.o.o_worker_set_ul_script(self.w.h, C.cast(dec.fn, C.c_void_p), None)
        self.o.o_worker_set_last_ul_snr(self.w.h, 0.0)
        self.seen = 0

    def close(self):
        self.w = None

    def now(self, k):
        pass   # the oracle's clock is its subframe count (one probe call = 1 ms)

    def age(self):
        self.o.o_worker_ul_update_database(self.w.h)
        return self.o.o_worker_nof_tracked_ul(self.w.h)

    def cfg(self, r, a, c, ri, t):
        self.o.o_worker_ul_set_ue_config(self.w.h, r, a, c, ri, t)

    def subframe(self, tti, ent):
        flat = (C.c_uint32 * (12 * max(len(ent), 1)))(*[v for e in ent for v in e])
        self.o.o_worker_ul_decode_probe(self.w.h, tti, len(ent), flat)
        recs = parse_pcap(self.w.pcap_bytes())
        new = recs[self.seen:]
        self.seen = len(recs)
        return [(r["rnti"], len(r["pdu"]), _fnv(r["pdu"])) for r in new]

    def state(self, rntis):
        def tr(r):  # find_tracking_info_RNTI_ul: 1 unknown (no entry, room left), 5 full buffer
            m = self.o.o_worker_tracked_mod_ul(self.w.h, r)
            n = self.o.o_worker_nof_tracked_ul(self.w.h)
            return m if m else (1 if n < 250 else 5)
        return (self.o.o_worker_nof_tracked_ul(self.w.h), [(r, tr(r)) for r in rntis])


def run(side, life):
    ues, truth, ev, by_grant = script(life)
    dec = ScriptedUplinkDecoder(life[3], truth, life[6], by_grant)
    side.open(life, dec)
    out, k = [], 0
    for e in ev:
        if e[0] == "age":
            out.append(("age", side.age()))
        elif e[0] == "cfg":
            side.cfg(*e[1:])
        else:
            k += 1
            side.now(k)
            dec.log = []
            recs = side.subframe(e[1], e[2])
            out.append((list(dec.log), recs))
    out.append(("state", side.state(ues)))
    side.close()
    return out


def digest(results):
    h = hashlib.sha256()
    for r in results:
        h.update(repr(r).encode())
    return h.hexdigest()[:32]


def facts(results):
    calls = [c for r in results if r[0] not in ("age", "state") for c in r[0]]
    recs = [x for r in results if r[0] not in ("age", "state") for x in r[1]]
    return {"subframes": sum(r[0] not in ("age", "state") for r in results), "attempts": len(calls), "attempts_by_modulation": [sum(c[0][5] == q for c in calls) for q in (2, 4, 6, 8)],
            "attempts_with_csi": sum(c[0][9] for c in calls), "passed": sum(c[1] for c in calls), "records": len(recs),
            "tracked_after_each_ageing": [r[1] for r in results if r[0] == "age"],
            "final_tracked_by_modulation": [sum(m == q for _, m in results[-1][1][1]) for q in (1, 2, 3, 4, 5)]}


def reference_sources_sha256(ref="/root/reference"):
    h = hashlib.sha256()
    for f in REF_SOURCES:
        h.update(open(os.path.join(ref, f), "rb").read())
    return h.hexdigest()


def trial_table(ref):
    """the reference's trial order, read off its own decoder: for every MCS index 0..31, tracked state (unknown / 16 / 64 / 256QAM maximum, learnt by letting a first
    grant pass at the right attempt) and 256QAM-table allocation (usable / L_prb 0 / L_prb 110) one grant whose every attempt FAILS -> the list of attempts
    (use of the 256QAM-table grant, modulation the decoder runs with); and, from the unknown state, the modulation learnt when the k-th attempt passes.
    -> {"mcs/state/L256": [[use256, qm], ...], "learn mcs/L256/k": state afterwards}"""
    out = {}
    nprb, rnti = 50, 0x1234
    for mcs in range(32):
        mod = 2 if mcs < 11 else 4 if mcs < 21 else 6
        mod256 = 2 if mcs < 6 else 4 if mcs < 14 else 6 if mcs < 23 else 8
        tbs, tbs256 = (0, 0) if mcs >= 29 else (800, 1600)
        for L256 in (10, 0, 110):
            for state, teach in ((1, None), (2, (21, 0)), (3, (21, 1)), (4, (21, 2))):
                log = []

                def cb(user, call, snr, payload, _teach=teach, _log=log):
                    w = [int(call[i]) for i in range(16)]
                    qm = w[5] if w[7] else min(w[5], 4)
                    snr[0] = 10.0
                    if w[0] in (0, 1):   # the teaching grants: pass at the attempt that leaves the wanted state
                        k = sum(1 for x in _log if x[0] == 1 and x[3] == w[0])
                        _log.append((1, w[6], qm, w[0]))
                        return 1 if k == _teach[1] else 0
                    _log.append((2, 1 if w[6] == tbs256 and tbs256 != tbs else 0, qm))
                    return 0
                fn = UL_SCRIPT(cb)
                ref.lib.ref_ul_set_script(C.cast(fn, C.c_void_p), None)
                ref.lib.ref_collect_set_now_ms(0)
                h = ref.lib.ref_ul_new(nprb, 1)
                if teach:   # twice: update_RNTI_ul only ADDS an RNTI it does not know (as unknown); the second success sets the modulation (MCSTracking.cc:71-85)
                    e = (rnti, 0, teach[0], 10, 0, 6, 800, 10, 8, 1600, 0, 0)
                    ref.lib.ref_ul_decode(h, 0, 1, (C.c_uint32 * 12)(*e))
                    ref.lib.ref_ul_decode(h, 1, 1, (C.c_uint32 * 12)(*e))
                e = (rnti, 0, mcs, 10, 0, mod, tbs, L256, mod256, tbs256, 0, 0)
                ref.lib.ref_ul_decode(h, 2, 1, (C.c_uint32 * 12)(*e))
                out["%d/%d/%d" % (mcs, state, L256)] = [[x[1], x[2]] for x in log if x[0] == 2]
                ref.lib.ref_ul_free(h)
            # from the unknown state: what the k-th attempt teaches when it passes
            for k in range(3):
                cnt = [0]

                def cb2(user, call, snr, payload, _k=k, _cnt=cnt):
                    snr[0] = 10.0
                    if int(call[0]) == 2:
                        _cnt[0] += 1
                        return 1 if _cnt[0] - 1 == _k else 0
                    return 0
                fn = UL_SCRIPT(cb2)
                ref.lib.ref_ul_set_script(C.cast(fn, C.c_void_p), None)
                h = ref.lib.ref_ul_new(nprb, 1)
                ref.lib.ref_ul_set_ue_config(h, rnti, 10, 8, 11, 0)   # the RNTI has an entry (unknown modulation): a connection setup was seen
                e = (rnti, 0, mcs, 10, 0, mod, tbs, L256, mod256, tbs256, 0, 0)
                ref.lib.ref_ul_decode(h, 2, 1, (C.c_uint32 * 12)(*e))
                if cnt[0] > k:
                    out["learn %d/%d/%d" % (mcs, L256, k)] = {0: 2, 1: 3, 2: 4, 3: 1, 4: 6, 5: 5}[ref.lib.ref_ul_tracked(h, rnti)]
                ref.lib.ref_ul_free(h)
    return out
