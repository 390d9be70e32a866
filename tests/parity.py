"""Shared driver of the GPU-vs-oracle parity tests: runs the same seeded synthetic subframes through the HIP path
(C ABI, ltesniffer_amd.Phy) and through the CPU oracle, and compares every stage tap and the MAC-LTE record stream."""
import numpy as np

import ltesniffer_amd as la
from lsn_testlib import OracleWorker, TxGen, oracle_trace, oracle_trace_enable, parse_pcap, scenario


def gen_subframes(sc, n, **txkw):
    tx = TxGen(**txkw, **sc)
    iq = np.zeros((n, sc["nof_rx"], tx.sf_len), dtype=np.complex64)
    truth = []
    tti0 = None
    for i in range(n):
        tti, x, pdus = tx.next()
        if tti0 is None:
            tti0 = tti
        iq[i] = x
        truth.append(pdus)
    return tti0, iq, truth


def gen_capture(sc, n, threads=None, **txkw):
    """the same capture as gen_subframes(sc, n)[:2] rendered by `threads` workers (txg_generate): -> (tti0, iq[n, rx, sf_len])"""
    import ctypes as C
    import os
    tx = TxGen(**txkw, **sc)
    lib = tx.lib
    lib.txg_generate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
    lib.txg_generate.restype = C.c_int
    iq = np.empty((n, sc["nof_rx"], tx.sf_len), dtype=np.complex64)
    if threads is None:
        threads = max(1, min(32, len(os.sched_getaffinity(0))))
    tti0 = lib.txg_generate(C.byref(tx.cfg), n, iq.ctypes.data, int(threads))
    assert tti0 >= 0, "txg_generate failed"
    return tti0, iq


class CfoLoop:
    """The engine's CFO correction (lsn_engine.h, lsn_phy_set_cfo_correction) restated for the oracle driver: the offset removed from chunk g of `batch` subframes is
    c[g] = c[g-1] + alpha * (m[g-D] - c[g-1]) (mode 2; D = 4 chunks of loop delay), m[k] = c[k] + mean CRS residual of chunk k; mode 1: the fixed offset.
    Arithmetic as in the engine: doubles, rounded to float once per value."""
    D = 4

    def __init__(self, mode, cfo_hz, alpha, batch):
        self.mode, self.c, self.alpha, self.batch = mode, float(np.float32(cfo_hz)), float(np.float32(alpha)), batch
        self.meas, self.hist, self.acc = [], [], []

    def cfo(self, i):
        if i % self.batch == 0:  # a new chunk is launched
            g = i // self.batch
            if self.mode == 2 and g >= self.D:
                self.c = float(np.float32(self.c + self.alpha * (self.meas[g - self.D] - self.c)))
            self.hist.append(self.c)
            self.acc = []
        return self.c

    def seen(self, i, n_total, est_hz):
        self.acc.append(float(est_hz))
        if (i + 1) % self.batch == 0 or i + 1 == n_total:
            s = 0.0
            for v in self.acc:
                s += v
            self.meas.append(float(np.float32(self.c + s / len(self.acc))))


def run_oracle(sc, tti0, iq, update_meta_period=0, taps=True, mcs_update_interval=None, trace=False, harq_mode=0, cfo_loop=None, **okw):
    """trace=True: the oracle's stage-C recorder runs over these subframes; read it with lsn_testlib.oracle_trace() afterwards"""
    oracle_trace_enable(trace)
    okw.setdefault("cp", sc.get("cp", 0))
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"], sc["phich_ng_x6"], **okw)
    if harq_mode:
        ow.set_harq(harq_mode)
    if mcs_update_interval is not None:
        ow.set_mcs_update_interval(mcs_update_interval)
    per_sf = []
    for i in range(iq.shape[0]):
        upd = 1 if (update_meta_period and i % update_meta_period == 0) else 0
        ow.work(iq[i], tti0 + i, update_meta=upd, cfo_hz=cfo_loop.cfo(i) if cfo_loop else 0.0)
        if cfo_loop:
            cfo_loop.seen(i, iq.shape[0], ow.chest().cfo_hz)
        if taps:
            ch = ow.chest()
            # the record of la.TAP_CHEST: noise, rsrp, cepow as [rx 0..1][port 0..W-1] with W = 2 (one or two ports) or 4, then seven scalars
            W = 4 if sc["nof_ports"] == 4 else 2
            per_port = [v[rx * 4 + p] for v in (ch.noise, ch.rsrp, ch.cepow) for rx in range(2) for p in range(W)]
            per_sf.append(dict(grid=ow.grid(), ce=ow.ce(), llr=ow.llr(), cfi=ow.cfi(), accepted=ow.accepted(), rb_power=ow.rb_power(),
                               chest=np.array(per_port + list(ch.cfo_corr) +
                                              [ch.noise_avg, ch.rsrp_avg, ch.snr_db, ch.cfo_hz, ch.chan_ref], dtype=np.float32)))
    recs = parse_pcap(ow.pcap_bytes())
    return ow, per_sf, recs


def compare_stage_c(phy, otrace, first_tti, nsf, exact_iters=False):
    """Stage C behind its by-products: every decode call the ORACLE made on the subframes of the last GPU chunk (first_tti .. first_tti + nsf) must have
    a product decode job with bit-identical descrambled int16 soft bits (k_pdsch_demod), bit-identical de-rate-matched streams (k_rm, unpacked
    from the transposed 10-bit words) and the same per-code-block verdict and iteration count (k_turbo).  The product decodes MORE jobs than the
    oracle (speculative table attempts) - those are not looked at; a job planned with a p-a that commit later rejected leaves a second job with the
    same key, so a call matches when ANY job with its key is identical.  Code blocks the product skipped because the first block of their
    transport block had failed (first-block gating; none when exact_iters, i.e. under LSN_NO_CB_SKIP=1) carry no verdict to compare.
    -> (list of mismatches, number of oracle calls compared, code blocks compared, iterations the oracle ran, iterations the product ran on them)"""
    jobs = phy.stage_c_jobs()
    by_key = {}
    for j in jobs:
        if j["have"]:
            by_key.setdefault((j["tti"], j["rnti"], tuple(j["qm"]), j["nof_re"], tuple((c["tb"], c["K"], c["F"], c["E"], c["rv"]) for c in j["cbs"])), []).append(j)
    bad, ncall, ncb, it_o, it_g = [], 0, 0, 0, 0
    for o in otrace:
        if o["is_ul"] or not (0 <= (o["tti"] - first_tti) % 10240 < nsf):
            continue
        ncall += 1
        key = (o["tti"], o["rnti"], tuple(o["qm"]), o["nof_re"], tuple((c["tb"], c["K"], c["F"], c["E"], c["rv"]) for c in o["cbs"]))
        cands = by_key.get(key)
        if not cands:
            bad.append(("no product job", key[:4]))
            continue
        why = None
        for g in cands:
            why = None
            for q in range(2):
                if o["qm"][q] and not np.array_equal(g["llr"][q], o["llr"][q]):
                    d = np.nonzero(g["llr"][q] != o["llr"][q])[0]
                    why = ("llr16", key[:4], q, int(d.size), int(d[0]), int(g["llr"][q][d[0]]), int(o["llr"][q][d[0]]))
                    break
            if why:
                continue
            for k, (cg, co) in enumerate(zip(g["cbs"], o["cbs"])):
                if not np.array_equal(cg["d3"], co["d3"]):
                    d = np.argwhere(cg["d3"] != co["d3"])
                    why = ("rm_words", key[:4], k, int(len(d)), d[0].tolist(), int(cg["d3"][tuple(d[0])]), int(co["d3"][tuple(d[0])]))
                    break
                if cg["skipped"] and not exact_iters:
                    continue
                if (cg["ok"], cg["iters"]) != (co["ok"], co["iters"]):
                    why = ("cb_result", key[:4], k, (cg["ok"], cg["iters"]), (co["ok"], co["iters"]))
                    break
            if not why:
                for cg, co in zip(g["cbs"], o["cbs"]):
                    if not cg["skipped"]:
                        ncb += 1
                        it_o += co["iters"]
                        it_g += cg["iters"]
                break
        if why:
            bad.append(why)
    return bad, ncall, ncb, it_o, it_g


def gpu_records(phy):
    if phy.pcapwriter is not None:
        return oracle_records(parse_pcap(phy.pcapwriter.bytes()))
    return [la.mac_lte_record(ctx, pdu) for ctx, pdu in phy.pdus]


def oracle_records(recs):
    return [r["ctx"] + r["pdu"] for r in recs]


def compare_taps(phy, per_sf, sc, base=0, nsf=None):
    """bit-exact comparison of the stage taps of the LAST GPU batch against oracle subframes base..base+nsf"""
    A, P, nre = sc["nof_rx"], sc["nof_ports"], 12 * sc["nof_prb"]
    bad = []
    n = nsf if nsf is not None else len(per_sf) - base
    for i in range(n):
        o = per_sf[base + i]
        g = phy.tap(la.TAP_GRID, i, np.complex64, A * 14 * nre).reshape(A, 14, nre)
        if not np.array_equal(g.view(np.uint32), o["grid"].view(np.uint32)):
            bad.append((i, "grid", float(np.abs(g - o["grid"]).max())))
        ce = phy.tap(la.TAP_CE, i, np.complex64, P * A * 14 * nre).reshape(P, A, 14, nre)
        if not np.array_equal(ce.view(np.uint32), o["ce"].view(np.uint32)):
            bad.append((i, "ce", float(np.abs(ce - o["ce"]).max())))
        cfi = int(phy.tap(la.TAP_CFI, i, np.uint32, 1)[0])
        if cfi != o["cfi"]:
            bad.append((i, "cfi", cfi, o["cfi"]))
            continue
        llr = phy.tap(la.TAP_PDCCH_LLR, i, np.float32, 6400)
        if not np.array_equal(llr.view(np.uint32), o["llr"].view(np.uint32)):
            bad.append((i, "llr", len(llr), len(o["llr"])))
        ch = phy.tap(la.TAP_CHEST, i, np.float32, 31 if P == 4 else 19)
        if not np.array_equal(ch.view(np.uint32), o["chest"].view(np.uint32)):
            bad.append((i, "chest", ch.tolist(), o["chest"].tolist()))
        rbp = phy.tap(la.TAP_RB_POWER, i, np.float32, 110)
        if not np.array_equal(rbp.view(np.uint32), o["rb_power"].view(np.uint32)):
            bad.append((i, "rb_power", float(np.abs(rbp - o["rb_power"]).max())))
        acc = phy.tap(la.TAP_ACCEPTED, i, np.uint32, 64 * 6).reshape(-1, 6)
        if [tuple(int(v) for v in r) for r in acc] != [tuple(r) for r in o["accepted"]]:
            bad.append((i, "accepted", acc.tolist(), o["accepted"]))
    return bad


def compare_candidate_tables(phy, per_sf, sc, tti0, base=0, nsf=None, skip_not_computed=False):
    """the exhaustive blind-decode table of the GPU (every location x every DCI size: payload bits, CRC remainder = RNTI,
    search-space verdict) and the per-CCE LLR power against the oracle's candidate decoder run over the same LLRs"""
    import ctypes as C
    from lsn_testlib import MAX_LOC, MAX_SIZES, CCE_STRIDE, LsnCand, OCell, candidate_table, hosttest, oracle
    h = hosttest()
    cell = OCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["phich_ng_x6"])
    sizes = sorted({oracle().o_dci_format_sizeof(C.byref(cell), f) for f in range(9)})
    bad = []
    n = nsf if nsf is not None else len(per_sf) - base
    for i in range(n):
        o = per_sf[base + i]
        nof_cce = len(o["llr"]) // 72
        cand, pw = candidate_table(o["llr"], nof_cce, sizes, (tti0 + base + i) % 10)
        ref = np.frombuffer(bytes(cand), dtype=np.dtype([("bits", "<u8"), ("rnti", "<u4"), ("flags", "<u4")]))
        got = phy.tap(la.TAP_CANDIDATES, i, np.uint8, MAX_LOC * MAX_SIZES * 16).view(ref.dtype)
        gpw = phy.tap(la.TAP_CCE_POWER, i, np.float32, CCE_STRIDE)
        if not np.array_equal(gpw[:nof_cce].view(np.uint32), pw[:nof_cce].view(np.uint32)):
            bad.append((i, "cce_power"))
        for li in range(MAX_LOC):
            for si in range(len(sizes)):
                a, b = got[li * MAX_SIZES + si], ref[li * MAX_SIZES + si]
                if skip_not_computed and (int(a["flags"]) & 0x80):   # LSN_CAND_NOT_COMPUTED: left out by the pruning (reported, not an error) - unless the search had it decoded
                    bad.append((i, "left_out", li, si))
                    continue
                if (a["flags"] & 1) != (b["flags"] & 1) or ((b["flags"] & 1) and (a["bits"] != b["bits"] or a["rnti"] != b["rnti"] or a["flags"] != b["flags"])):
                    bad.append((i, "cand", li, si, int(a["bits"]), int(b["bits"]), int(a["rnti"]), int(b["rnti"]), int(a["flags"]), int(b["flags"])))
    return bad
