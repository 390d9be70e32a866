"""ctypes bindings used by the tests, smoke() and bench.py's cpu_baseline leg:
   - the CPU ORACLE (oracle/_build/liblsn_oracle.so)  -- test infrastructure, never the product path
   - the synthetic eNB transmitter (tools/txgen/_build/libtxgen.so) -- test tooling
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "liblsn_oracle.so")
TXGEN_SO = os.path.join(ROOT, "tools", "txgen", "_build", "libtxgen.so")


def _ensure(path, mkdir):
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", mkdir], stdout=subprocess.DEVNULL)
    return path


class OCell(C.Structure):
    _fields_ = [("nof_prb", C.c_uint32), ("nof_ports", C.c_uint32), ("id", C.c_uint32), ("phich_ng_x6", C.c_uint32), ("pusch_hop_offset", C.c_uint32),
                ("cp", C.c_uint32)]  # 0 = normal cyclic prefix, 1 = extended


class OWorkerCfg(C.Structure):
    _fields_ = [("cell", OCell), ("nof_rx", C.c_uint32), ("histogram_threshold", C.c_uint32), ("split_ratio", C.c_double),
                ("skip_secondary", C.c_int), ("mcs_tracking_mode", C.c_int), ("max_turbo_iter", C.c_int),
                ("enable_shortcut", C.c_int)]


class OChestRes(C.Structure):
    # noise / rsrp / cepow: [rx 0..1][port 0..3]
    _fields_ = [("noise", C.c_float * 8), ("rsrp", C.c_float * 8), ("cepow", C.c_float * 8), ("cfo_corr", C.c_float * 2),
                ("noise_avg", C.c_float), ("rsrp_avg", C.c_float), ("snr_db", C.c_float), ("cfo_hz", C.c_float),
                ("chan_ref", C.c_float)]


class OStats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("nof_decoded_locations", "nof_cce", "nof_missed_cce", "nof_subframes",
                                          "nof_subframe_collisions_dw", "nof_subframe_collisions_up", "nof_locations")]


class TxgCfg(C.Structure):
    _fields_ = [("nof_prb", C.c_uint32), ("nof_ports", C.c_uint32), ("cell_id", C.c_uint32), ("phich_ng_x6", C.c_uint32),
                ("nof_rx", C.c_uint32), ("snr_db", C.c_float), ("cfo_hz", C.c_float), ("delay_samples", C.c_uint32),
                ("seed", C.c_uint64), ("n_rnti", C.c_uint32), ("dl_min", C.c_uint32), ("dl_max", C.c_uint32),
                ("ul_min", C.c_uint32), ("ul_max", C.c_uint32), ("cfi", C.c_uint32), ("mix_tm3_pct", C.c_uint32),
                ("mix_tm4_pct", C.c_uint32), ("pct_256qam", C.c_uint32), ("mcs_min", C.c_uint32), ("mcs_max", C.c_uint32),
                ("sib_period", C.c_uint32), ("rar_period", C.c_uint32), ("paging_period", C.c_uint32),
                ("start_tti", C.c_uint32), ("fixed_L", C.c_uint32), ("pct_rv", C.c_uint32), ("pct_cqi_req", C.c_uint32), ("pct_hop", C.c_uint32), ("pusch_hop_offset", C.c_uint32),
                ("msg4_period", C.c_uint32), ("msg4_p_a_idx", C.c_uint32), ("si_len", C.c_uint32 * 2), ("si_msg", (C.c_uint8 * 96) * 2),
                ("pg_len", C.c_uint32), ("pg_msg", C.c_uint8 * 96), ("pct_harq", C.c_uint32), ("cp", C.c_uint32),
                ("chan_model", C.c_uint32), ("doppler_hz", C.c_float), ("timing_offset_samples", C.c_float), ("cfo_drift_hz_per_s", C.c_float)]  # TS 36.101 B.2 fading: 1 EPA, 2 EVA, 3 ETU


class TxgPdu(C.Structure):
    _fields_ = [("rnti", C.c_uint16), ("format", C.c_uint8), ("L", C.c_uint8), ("ncce", C.c_uint16), ("tti", C.c_uint32),
                ("nbytes", C.c_uint32), ("offset", C.c_uint32), ("tb", C.c_uint8), ("mod", C.c_uint8),
                ("table256", C.c_uint8), ("is_ul", C.c_uint8), ("nof_prb", C.c_uint32), ("mcs", C.c_uint32), ("cqi_req", C.c_uint32), ("hop_bits_plus1", C.c_uint32)]


_oracle = None
_txgen = None


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(_ensure(ORACLE_SO, os.path.join(ROOT, "oracle")))
        lib.o_crc_bits.restype = C.c_uint32
        lib.o_crc_bits.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_int]
        lib.o_gold.argtypes = [C.c_uint32, C.c_void_p, C.c_int]
        lib.o_reduce256.restype = C.c_float
        lib.o_reduce256.argtypes = [C.c_void_p, C.c_int]
        lib.o_fft_size.argtypes = [C.c_uint32]
        lib.o_fft.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        lib.o_fft_twiddles.argtypes = [C.c_int, C.c_void_p]
        lib.o_ofdm_rx.argtypes = [C.POINTER(OCell), C.c_void_p, C.c_uint32, C.c_void_p]
        lib.o_dci_format_sizeof.restype = C.c_uint32
        lib.o_dci_format_sizeof.argtypes = [C.POINTER(OCell), C.c_int]
        lib.o_dci_decode.restype = C.c_uint16
        lib.o_dci_decode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.o_validate_location.restype = C.c_uint32
        lib.o_validate_location.argtypes = [C.c_uint32] * 4 + [C.c_uint16]
        lib.o_turbo_nwin.argtypes = [C.c_int]
        lib.o_turbo_decode_cb.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.POINTER(C.c_int)]
        lib.o_rm_turbo_rx_cb.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.o_pdsch_decode_tb.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.POINTER(C.c_int)]
        lib.o_tbs_from_idx.argtypes = [C.c_int, C.c_uint32]
        lib.o_worker_new.restype = C.c_void_p
        lib.o_worker_new.argtypes = [C.POINTER(OWorkerCfg)]
        lib.o_worker_free.argtypes = [C.c_void_p]
        lib.o_worker_set_pcap.argtypes = [C.c_void_p, C.c_void_p]
        lib.o_worker_work.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_int, C.c_float]
        lib.o_worker_stats.restype = C.POINTER(OStats)
        lib.o_worker_stats.argtypes = [C.c_void_p]
        lib.o_worker_grid.restype = C.c_void_p
        lib.o_worker_grid.argtypes = [C.c_void_p]
        lib.o_worker_ce.restype = C.c_void_p
        lib.o_worker_ce.argtypes = [C.c_void_p]
        lib.o_worker_llr.restype = C.c_void_p
        lib.o_worker_llr.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        lib.o_worker_chest.restype = C.POINTER(OChestRes)
        lib.o_worker_chest.argtypes = [C.c_void_p]
        lib.o_worker_cfi.restype = C.c_uint32
        lib.o_worker_cfi.argtypes = [C.c_void_p]
        lib.o_worker_accepted.restype = C.c_uint32
        lib.o_worker_accepted.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        lib.o_worker_total_iters.restype = C.c_uint64
        lib.o_worker_total_iters.argtypes = [C.c_void_p]
        lib.o_worker_algo_bytes.restype = C.c_uint64
        lib.o_worker_algo_bytes.argtypes = [C.c_void_p]
        lib.o_worker_rntiman.restype = C.c_void_p
        lib.o_worker_rntiman.argtypes = [C.c_void_p]
        lib.o_rntiman_nof_active.restype = C.c_uint32
        lib.o_rntiman_nof_active.argtypes = [C.c_void_p]
        lib.o_pcap_open_mem.restype = C.c_void_p
        lib.o_pcap_open_file.restype = C.c_void_p
        lib.o_pcap_open_file.argtypes = [C.c_char_p]
        lib.o_pcap_write.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint16, C.c_uint8, C.c_uint8,
                                     C.c_uint8, C.c_uint32, C.c_uint32]
        lib.o_pcap_mem.restype = C.c_void_p
        lib.o_pcap_mem.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        lib.o_pcap_nof_records.restype = C.c_uint32
        lib.o_pcap_nof_records.argtypes = [C.c_void_p]
        lib.o_pcap_close.argtypes = [C.c_void_p]
        _oracle = lib
    return _oracle


def txgen():
    global _txgen
    if _txgen is None:
        lib = C.CDLL(_ensure(TXGEN_SO, os.path.join(ROOT, "tools", "txgen")))
        lib.txg_new.restype = C.c_void_p
        lib.txg_new.argtypes = [C.POINTER(TxgCfg)]
        lib.txg_free.argtypes = [C.c_void_p]
        lib.txg_next.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(TxgPdu), C.c_int, C.c_void_p, C.c_int]
        lib.txg_sf_len.restype = C.c_uint32
        lib.txg_sf_len.argtypes = [C.c_void_p]
        lib.txg_tti.restype = C.c_uint32
        lib.txg_tti.argtypes = [C.c_void_p]
        _txgen = lib
    return _txgen


# -------- scenario presets (sizes scaled by the caller) --------
def scenario(name, seed=1, **over):
    base = dict(nof_prb=100, nof_ports=2, cell_id=1, phich_ng_x6=1, nof_rx=2, snr_db=30.0, cfo_hz=0.0, delay_samples=0,
                seed=seed, n_rnti=32, dl_min=6, dl_max=6, ul_min=2, ul_max=2, cfi=3, mix_tm3_pct=0, mix_tm4_pct=0,
                pct_256qam=0, mcs_min=0, mcs_max=28, sib_period=1, rar_period=0, paging_period=0, start_tti=0, fixed_L=0, pct_rv=0, pct_cqi_req=0, pct_hop=0, pusch_hop_offset=0, msg4_period=0, msg4_p_a_idx=4)
    if "pct_harq" in over:  # (only present when asked for: the scenario dict of the gated cfg3 stream is part of cached file keys)
        base["pct_harq"] = 0
    if "cp" in over:  # extended cyclic prefix (cp = 1): same rule
        base["cp"] = 0
    if "chan_model" in over:  # multipath fading (TS 36.101 B.2: 1 EPA, 2 EVA, 3 ETU; doppler_hz; timing_offset_samples): same rule
        base.update(chan_model=0, doppler_hz=0.0, timing_offset_samples=0.0)
    if "cfo_drift_hz_per_s" in over:
        base["cfo_drift_hz_per_s"] = 0.0
    presets = {
        # config 1: 10 MHz, single RNTI, TM1 QPSK, 1 port / 1 rx
        "cfg1": dict(nof_prb=50, nof_ports=1, nof_rx=1, snr_db=20.0, cfo_hz=300.0, n_rnti=1, dl_min=1, dl_max=1, ul_min=0,
                     ul_max=1, cfi=2, mcs_min=0, mcs_max=9),
        # config 2: 20 MHz, 32 RNTIs, TM2, 64QAM
        "cfg2": dict(n_rnti=32, snr_db=28.0, dl_min=6, dl_max=6, ul_min=2, ul_max=2, mcs_min=17, mcs_max=28),
        # config 3: 20 MHz, 150 RNTIs, TM3/TM4 mix up to 256QAM (north-star)
        "cfg3": dict(n_rnti=150, snr_db=30.0, dl_min=8, dl_max=14, ul_min=3, ul_max=6, mix_tm3_pct=40, mix_tm4_pct=30,
                     pct_256qam=50, mcs_min=0, mcs_max=28, rar_period=200, paging_period=64),
        "small": dict(nof_prb=25, nof_ports=2, nof_rx=2, n_rnti=4, dl_min=2, dl_max=3, ul_min=0, ul_max=1, cfi=2,
                      mcs_min=2, mcs_max=20),
    }
    d = dict(base)
    d.update(presets[name])
    d.update(over)
    return d


class TxGen:
    def __init__(self, si_msgs=None, paging_msg=None, **kw):
        self.lib = txgen()
        self.cfg = TxgCfg(**kw)
        if paging_msg:  # PCCH message of every P-RNTI transmission
            self.cfg.pg_len = len(paging_msg)
            for j, b in enumerate(bytes(paging_msg)[:96]):
                self.cfg.pg_msg[j] = b
        for i, m in enumerate(si_msgs or []):  # BCCH-DL-SCH messages of the SI-RNTI transmissions (alternating), None / b"" = random bytes
            if m:
                self.cfg.si_len[i] = len(m)
                for j, b in enumerate(bytes(m)[:96]):
                    self.cfg.si_msg[i][j] = b
        self.h = self.lib.txg_new(C.byref(self.cfg))
        assert self.h, "txg_new failed"
        self.sf_len = self.lib.txg_sf_len(self.h)
        self.nof_rx = self.cfg.nof_rx
        self._pdus = (TxgPdu * 128)()
        self._pbuf = np.zeros(1 << 19, dtype=np.uint8)

    def next(self):
        """-> (tti, iq[nof_rx, sf_len] complex64, pdus list of dict(with payload bytes))"""
        tti = self.lib.txg_tti(self.h)
        iq = np.zeros((self.nof_rx, self.sf_len), dtype=np.complex64)
        n = self.lib.txg_next(self.h, iq.ctypes.data, self._pdus, 128, self._pbuf.ctypes.data, self._pbuf.size)
        out = []
        for i in range(n):
            p = self._pdus[i]
            out.append(dict(rnti=p.rnti, format=p.format, L=p.L, ncce=p.ncce, tti=p.tti, tb=p.tb, mod=p.mod,
                            table256=p.table256, is_ul=p.is_ul, nof_prb=p.nof_prb, mcs=p.mcs, cqi_req=p.cqi_req, hop_bits_plus1=p.hop_bits_plus1,
                            n_prb=p.offset if p.is_ul else 0,
                            payload=bytes(self._pbuf[p.offset:p.offset + p.nbytes]) if not p.is_ul else b""))
        return tti, iq, out

    def __del__(self):
        try:
            self.lib.txg_free(self.h)
        except Exception:
            pass


class OTraceJob(C.Structure):
    _fields_ = [("tti", C.c_uint32), ("rnti", C.c_uint32), ("nof_re", C.c_uint32), ("qm", C.c_uint32 * 2), ("llr_len", C.c_uint32 * 2), ("ncb", C.c_uint32),
                ("cb_first", C.c_uint32), ("is_ul", C.c_uint32)]


class OTraceCb(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("job", "tb", "K", "F", "E", "rv", "iters", "ok")]


def oracle_trace_enable(on=True):
    """stage-C recorder of the oracle (o_trace.c): switch on (clearing the log) before the subframes are worked"""
    oracle().o_trace_enable(1 if on else 0)


def oracle_trace():
    """-> list of dict(tti, rnti, nof_re, qm, is_ul, llr=[cw0, cw1] int16, cbs=[dict(tb, K, F, E, rv, iters, ok, d3 int16[3, K + 4])]) of every decode call since
    oracle_trace_enable()"""
    lib = oracle()
    lib.o_trace_njobs.restype = C.c_uint32
    lib.o_trace_job.argtypes = [C.c_uint32, C.POINTER(OTraceJob)]
    lib.o_trace_job_llr.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_uint32]
    lib.o_trace_cb_get.argtypes = [C.c_uint32, C.POINTER(OTraceCb), C.c_void_p, C.c_uint32]
    out = []
    for i in range(lib.o_trace_njobs()):
        h = OTraceJob()
        assert lib.o_trace_job(i, C.byref(h)) == 0
        llr = []
        for q in range(2):
            a = np.zeros(h.llr_len[q], dtype=np.int16)
            assert lib.o_trace_job_llr(i, q, a.ctypes.data, a.size) == a.size
            llr.append(a)
        cbs = []
        for k in range(h.cb_first, h.cb_first + h.ncb):
            c = OTraceCb()
            n = lib.o_trace_cb_get(k, C.byref(c), None, 0)
            d3 = np.zeros(n, dtype=np.int16)
            assert lib.o_trace_cb_get(k, C.byref(c), d3.ctypes.data, n) == n
            cbs.append(dict(tb=c.tb, K=c.K, F=c.F, E=c.E, rv=c.rv, iters=c.iters, ok=c.ok, d3=d3.reshape(3, c.K + 4)))
        out.append(dict(tti=h.tti, rnti=h.rnti, nof_re=h.nof_re, qm=[h.qm[0], h.qm[1]], is_ul=h.is_ul, llr=llr, cbs=cbs))
    return out


class OracleWorker:
    def __init__(self, nof_prb, nof_ports, cell_id, nof_rx, phich_ng_x6=1, threshold=5, split_ratio=0.99, skip_secondary=0,
                 mcs_tracking_mode=1, max_turbo_iter=12, enable_shortcut=1, cp=0):
        self.lib = oracle()
        self.cfg = OWorkerCfg(OCell(nof_prb, nof_ports, cell_id, phich_ng_x6, 0, cp), nof_rx, threshold, split_ratio, skip_secondary,
                              mcs_tracking_mode, max_turbo_iter, enable_shortcut)
        self.h = self.lib.o_worker_new(C.byref(self.cfg))
        assert self.h
        self.pcap = self.lib.o_pcap_open_mem()
        self.lib.o_worker_set_pcap(self.h, self.pcap)
        self.nof_rx = nof_rx
        self.nre = 12 * nof_prb
        self.nof_ports = nof_ports

    def work(self, iq, tti, update_meta=None, cfo_hz=0.0):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        ptrs = (C.c_void_p * self.nof_rx)(*[iq[r].ctypes.data for r in range(self.nof_rx)])
        if update_meta is None:
            update_meta = 0
        return self.lib.o_worker_work(self.h, ptrs, tti % 10, (tti // 10) % 1024, int(update_meta), float(cfo_hz))

    def grid(self):
        a = np.ctypeslib.as_array(C.cast(self.lib.o_worker_grid(self.h), C.POINTER(C.c_float)),
                                  shape=(self.nof_rx, 14, self.nre, 2))
        return a.copy().view(np.complex64)[..., 0]

    def ce(self):
        a = np.ctypeslib.as_array(C.cast(self.lib.o_worker_ce(self.h), C.POINTER(C.c_float)),
                                  shape=(self.nof_ports, self.nof_rx, 14, self.nre, 2))
        return a.copy().view(np.complex64)[..., 0]

    def llr(self):
        n = C.c_uint32()
        p = self.lib.o_worker_llr(self.h, C.byref(n))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n.value,)).copy()

    def chest(self):
        return self.lib.o_worker_chest(self.h).contents

    def cfi(self):
        return self.lib.o_worker_cfi(self.h)

    def ue_cfg(self, rnti):
        """MCSTracking::get_ue_config_rnti -> (has_ue_config, p_a, i_offset_ack, i_offset_cqi, i_offset_ri, cqi_type)"""
        class _Cfg(C.Structure):
            _fields_ = [("has", C.c_uint32), ("p_a", C.c_float), ("ack", C.c_uint32), ("cqi", C.c_uint32), ("ri", C.c_uint32), ("typ", C.c_uint32), ("bits", C.c_uint32)]
        c = _Cfg()
        self.lib.o_worker_ue_cfg.argtypes = [C.c_void_p, C.c_uint16, C.c_void_p]
        self.lib.o_worker_ue_cfg(self.h, rnti, C.byref(c))
        return (c.has, np.float32(c.p_a).item(), c.ack, c.cqi, c.ri, c.typ)

    def set_mcs_update_interval(self, seconds):
        self.lib.o_worker_set_mcs_update_interval.argtypes = [C.c_void_p, C.c_uint32]
        self.lib.o_worker_set_mcs_update_interval(self.h, seconds)

    def set_harq(self, mode=1):
        """DL HARQ soft combining (HARQ.cc; harq_mode is unreachable in the reference's CLI, ArgManager.cc:50,211-213)"""
        self.lib.o_worker_set_harq.argtypes = [C.c_void_p, C.c_int]
        self.lib.o_worker_set_harq(self.h, int(mode))

    def harq_stats(self):
        """verdicts of is_retransmission so far: [NEW_TX, RE_TX, FULL_BUFFER, DECODED, BUSY]"""
        st = (C.c_uint32 * 5)()
        self.lib.o_worker_harq_stats.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.o_worker_harq_stats(self.h, st)
        return list(st)

    def total_iters(self):
        """turbo iterations summed over every code block of every decode call so far"""
        return int(self.lib.o_worker_total_iters(self.h))

    def nof_tracked(self):
        self.lib.o_worker_nof_tracked.argtypes = [C.c_void_p]
        self.lib.o_worker_nof_tracked.restype = C.c_uint32
        return self.lib.o_worker_nof_tracked(self.h)

    def rb_power(self):
        self.lib.o_worker_rb_power.restype = C.POINTER(C.c_float)
        self.lib.o_worker_rb_power.argtypes = [C.c_void_p]
        return np.ctypeslib.as_array(self.lib.o_worker_rb_power(self.h), shape=(self.nre // 12,)).copy()

    def accepted(self):
        buf = (C.c_uint32 * (64 * 6))()
        n = self.lib.o_worker_accepted(self.h, buf, 64)
        return [tuple(buf[6 * i:6 * i + 6]) for i in range(min(n, 64))]

    def stats(self):
        return self.lib.o_worker_stats(self.h).contents

    def pcap_bytes(self):
        n = C.c_size_t()
        p = self.lib.o_pcap_mem(self.pcap, C.byref(n))
        return C.string_at(p, n.value)

    def nof_active(self):
        return self.lib.o_rntiman_nof_active(self.lib.o_worker_rntiman(self.h))

    def __del__(self):
        try:
            self.lib.o_worker_free(self.h)
            self.lib.o_pcap_close(self.pcap)
        except Exception:
            pass


def parse_pcap(data):
    """-> list of dict(direction, rnti_type, rnti, sfn, sf, crc, pdu) ; timestamps dropped"""
    assert data[:4] == b"\xd4\xc3\xb2\xa1"
    off, out = 24, []
    while off < len(data):
        ts, tu, il, ol = struct.unpack("<IIII", data[off:off + 16])
        off += 16
        p = data[off:off + il]
        off += il
        fs = (p[10] << 8) | p[11]
        out.append(dict(direction=p[1], rnti_type=p[2], rnti=(p[4] << 8) | p[5], sfn=fs >> 4, sf=fs & 15, crc=p[13],
                        pdu=bytes(p[19:]), ctx=bytes(p[:19])))
    return out


# -------- product host logic (HIP-free) through the test glue tests/native/liblsn_hosttest.so --------
HOSTTEST_SO = os.path.join(ROOT, "tests", "native", "_build", "liblsn_hosttest.so")
_host = None


class LsnCand(C.Structure):
    _fields_ = [("bits", C.c_uint64), ("rnti", C.c_uint32), ("flags", C.c_uint32)]


class OGrant(C.Structure):  # o_pdsch_grant_t / grant_out
    class TB(C.Structure):
        _fields_ = [("mcs_idx", C.c_uint32), ("rv", C.c_int), ("cw_idx", C.c_uint32), ("enabled", C.c_int), ("mod", C.c_int),
                    ("tbs", C.c_int), ("nof_bits", C.c_int)]
    _fields_ = [("prb_idx", (C.c_uint8 * 110) * 2), ("nof_prb", C.c_uint32), ("nof_re", C.c_uint32), ("nof_tb", C.c_uint32),
                ("tb", TB * 2), ("tx_scheme", C.c_int), ("pmi", C.c_uint32), ("nof_layers", C.c_uint32)]


def hosttest():
    global _host
    if _host is None:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "native")], stdout=subprocess.DEVNULL)
        lib = C.CDLL(HOSTTEST_SO)
        lib.lsnh_dci_format_sizeof.restype = C.c_uint32
        lib.lsnh_dci_format_sizeof.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
        lib.lsnh_validate_location.restype = C.c_uint32
        lib.lsnh_validate_location.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint16]
        lib.lsnh_validate_location_enum.restype = C.c_uint32
        lib.lsnh_validate_location_enum.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint16]
        lib.lsnh_dl_grant.argtypes = [C.c_uint32] * 5 + [C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_uint16, C.c_int, C.POINTER(OGrant)]
        lib.lsnh_ul_grant.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint16, C.c_void_p]
        lib.lsnh_cbsegm.argtypes = [C.c_int, C.c_void_p]
        lib.lsnh_turbo_il_offset.restype = C.c_uint32
        lib.lsnh_turbo_il_offset.argtypes = [C.c_int]
        lib.lsnh_turbo_two_wave_class.argtypes = [C.c_int]
        lib.lsnh_search_new.restype = C.c_void_p
        lib.lsnh_search_new.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_double, C.c_int]
        lib.lsnh_search_free.argtypes = [C.c_void_p]
        lib.lsnh_search_size_index.argtypes = [C.c_void_p, C.c_int]
        lib.lsnh_search_nof_sizes.argtypes = [C.c_void_p]
        lib.lsnh_search_nof_sizes.restype = C.c_uint32
        lib.lsnh_search_size.argtypes = [C.c_void_p, C.c_uint32]
        lib.lsnh_search_size.restype = C.c_uint32
        lib.lsnh_search_run.restype = C.c_uint32
        lib.lsnh_search_run.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32]
        lib.lsnh_search_activate_rar.argtypes = [C.c_void_p, C.c_uint16]
        lib.lsnh_search_stats.argtypes = [C.c_void_p, C.c_void_p]
        lib.lsnh_search_nof_active.argtypes = [C.c_void_p]
        lib.lsnh_search_nof_active.restype = C.c_uint32
        lib.lsnh_search_bench.restype = C.c_double
        lib.lsnh_search_bench.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _host = lib
    return _host


MAX_LOC, MAX_SIZES, CCE_STRIDE = 160, 8, 96


def candidate_table(llr, nof_cce, sizes, sf_idx=0):
    """What k_viterbi + k_cce_power produce for one subframe, computed with the ORACLE's candidate decoder:
    -> (cand[MAX_LOC*MAX_SIZES] LsnCand array, ccepow float32[CCE_STRIDE])."""
    lib = oracle()
    cand = (LsnCand * (MAX_LOC * MAX_SIZES))()
    pw = np.zeros(CCE_STRIDE, dtype=np.float32)
    llr = np.ascontiguousarray(llr, dtype=np.float32)
    for c in range(nof_cce):
        m = 0.0
        for v in llr[72 * c:72 * c + 72]:
            m += abs(float(v))
        pw[c] = np.float32(m / 72)
    lim = min(nof_cce, 84)
    li = 0
    payload = (C.c_uint8 * 256)()
    for l in (3, 2, 1, 0):
        L = 1 << l
        for i in range(lim // L):
            ncce = L * (i % (nof_cce // L))
            E = 72 * L
            ok = ncce * 72 + E <= nof_cce * 72 and all(pw[ncce + q] >= np.float32(0.7) for q in range(L))
            seg = llr[ncce * 72:ncce * 72 + E]
            if ok and not np.any(seg != 0):
                ok = False
            if ok:
                for si, nb in enumerate(sizes):
                    rnti = lib.o_dci_decode(seg.ctypes.data, E, nb, payload)
                    bits = 0
                    for b in range(nb):
                        bits |= int(payload[b]) << (63 - b)
                    e = cand[li * MAX_SIZES + si]
                    e.bits, e.rnti, e.flags = bits, rnti, 1 | (lib.o_validate_location(nof_cce, ncce, l, sf_idx, rnti) << 1)
            li += 1
    return cand, pw


# -------- uplink: transmitter + oracle bindings --------
class TxgUlCell(C.Structure):
    _fields_ = [("nof_prb", C.c_uint32), ("cell_id", C.c_uint32), ("cyclic_shift", C.c_uint32), ("delta_ss", C.c_uint32), ("group_hopping", C.c_uint32),
                ("sequence_hopping", C.c_uint32), ("cp", C.c_uint32)]  # cp = 1: extended cyclic prefix


class TxgUlGrant(C.Structure):
    _fields_ = [("rnti", C.c_uint16), ("n_dmrs", C.c_uint16), ("n_prb", C.c_uint32), ("L_prb", C.c_uint32), ("mod", C.c_uint32),
                ("tbs", C.c_uint32), ("rv", C.c_uint32), ("gain_db", C.c_float), ("phase_rad", C.c_float), ("ta_samples", C.c_float),
                ("nof_ack", C.c_uint32), ("cqi_bits", C.c_uint32), ("ri_bits", C.c_uint32), ("hop", C.c_uint32), ("n_prb2", C.c_uint32),
                ("i_ack_p1", C.c_uint32), ("i_cqi_p1", C.c_uint32), ("i_ri_p1", C.c_uint32)]


def pusch_hop_slot1(nof_prb, hop_offset, hop_bits, n_prb):
    """36.213 8.4.1 / Table 8.4-2, type-1 PUSCH hopping: first PRB of slot 1 (written from the specification, independent of oracle and product);
    None for type 2"""
    ho = hop_offset + (hop_offset % 2)
    n = nof_prb - ho - (nof_prb % 2)
    kind = {0: "half", 1: None}[hop_bits] if nof_prb < 50 else {0: "quart", 1: "quart_neg", 2: "half", 3: None}[hop_bits]
    if kind is None:
        return None
    if kind == "quart":
        return (n // 4 + n_prb) % n
    if kind == "half":
        return (n // 2 + n_prb) % n
    return n_prb - n // 4 if n_prb >= n // 4 else n + n_prb - n // 4


class OUci(C.Structure):
    _fields_ = [("nof_ack", C.c_uint32), ("cqi_bits", C.c_uint32), ("ri_bits", C.c_uint32),
                ("i_ack_p1", C.c_uint32), ("i_cqi_p1", C.c_uint32), ("i_ri_p1", C.c_uint32)]


class OUeCfg(C.Structure):  # o_ue_cfg_t
    _fields_ = [("has_ue_config", C.c_uint32), ("p_a", C.c_float), ("i_offset_ack", C.c_uint32), ("i_offset_cqi", C.c_uint32),
                ("i_offset_ri", C.c_uint32), ("cqi_type", C.c_uint32), ("bits_used", C.c_uint32)]


SIB2_FIELDS = ("n_sb", "hopping_mode", "pusch_hop_offset", "enable_64qam", "group_hopping_enabled", "group_assignment_pusch",
               "sequence_hopping_enabled", "cyclic_shift", "root_seq_idx", "prach_config_idx", "high_speed_flag", "zero_corr_zone", "prach_freq_offset")


class OSib2(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in SIB2_FIELDS + ("bits_used",)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in SIB2_FIELDS}


def oracle_sib2_decode(pdu):
    """-> (verdict, dict | None, bits_used)"""
    o = oracle()
    o.o_sib2_decode.argtypes = [C.c_char_p, C.c_int, C.POINTER(OSib2)]
    s = OSib2()
    r = o.o_sib2_decode(bytes(pdu), len(pdu), C.byref(s))
    return r, (s.as_dict() if r == 2 else None), int(s.bits_used)


def host_sib2_decode(pdu):
    """the product's parser (tests/native build of lsn_rrc.cc) -> (verdict, dict | None)"""
    h = hosttest()
    out = np.zeros(13, np.uint32)
    r = h.lsnh_sib2_decode(bytes(pdu), len(pdu), out.ctypes.data_as(C.c_void_p))
    return r, (dict(zip(SIB2_FIELDS, (int(v) for v in out))) if r == 2 else None)


class OApiEvent(C.Structure):
    _fields_ = [("tti", C.c_uint32), ("rnti", C.c_uint16), ("id_type", C.c_uint32), ("msg_type", C.c_uint32), ("value", C.c_char * 24)]


class OPagingId(C.Structure):
    _fields_ = [("is_imsi", C.c_uint32), ("nof_digits", C.c_uint32), ("digits", C.c_uint8 * 24), ("mmec", C.c_uint32), ("m_tmsi", C.c_uint32)]


def oracle_paging_decode(pdu):
    """-> list of ("imsi", digits) / ("tmsi", mmec, m_tmsi), or None when the message does not unpack"""
    o = oracle()
    o.o_paging_decode.argtypes = [C.c_char_p, C.c_int, C.POINTER(OPagingId), C.c_int]
    rec = (OPagingId * 16)()
    n = o.o_paging_decode(bytes(pdu), len(pdu), rec, 16)
    if n < 0:
        return None
    return [("imsi", "".join(str(d) for d in r.digits[:r.nof_digits])) if r.is_imsi else ("tmsi", int(r.mmec), int(r.m_tmsi)) for r in rec[:n]]


def host_paging_decode(pdu):
    h = hosttest()
    out = np.zeros(25 * 16, np.uint32)
    n = h.lsnh_paging_decode(bytes(pdu), len(pdu), out.ctypes.data_as(C.c_void_p), 16)
    if n < 0:
        return None
    return [("imsi", "".join(str(int(d)) for d in out[25 * i + 4:25 * i + 4 + out[25 * i + 1]])) if out[25 * i] else ("tmsi", int(out[25 * i + 2]), int(out[25 * i + 3]))
            for i in range(n)]


def oracle_api_events(api_mode, name, pdu, rnti, tti):
    """-> (events [(tti, rnti, id_type, msg_type, value)], to_pcap)"""
    o = oracle()
    o.o_api_dl_events.argtypes = [C.c_int, C.c_char, C.c_char_p, C.c_int, C.c_uint16, C.c_uint32, C.POINTER(OApiEvent), C.c_int, C.POINTER(C.c_int)]
    ev, n = (OApiEvent * 20)(), C.c_int(0)
    keep = o.o_api_dl_events(api_mode, name.encode(), bytes(pdu), len(pdu), rnti, tti, ev, 20, C.byref(n))
    return [(e.tti, e.rnti, e.id_type, e.msg_type, e.value.decode()) for e in ev[:n.value]], bool(keep)


def oracle_api_ul_msg3(api_mode, pdu, rnti, tti):
    o = oracle()
    o.o_api_ul_msg3_events.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_uint16, C.c_uint32, C.POINTER(OApiEvent), C.c_int, C.POINTER(C.c_int)]
    ev, n = (OApiEvent * 10)(), C.c_int(0)
    keep = o.o_api_ul_msg3_events(api_mode, bytes(pdu), len(pdu), rnti, tti, ev, 10, C.byref(n))
    return [(e.tti, e.rnti, e.id_type, e.msg_type, e.value.decode()) for e in ev[:n.value]], bool(keep)


def host_api_ul_msg3(api_mode, pdu, rnti, tti):
    h = hosttest()
    out = np.zeros(10 * 10, np.uint32)
    r = h.lsnh_api_ul_msg3_events(api_mode, bytes(pdu), len(pdu), rnti, tti, out.ctypes.data_as(C.c_void_p), 10)
    n = r & 0xFFFF
    return [(int(out[10 * i]), int(out[10 * i + 1]), int(out[10 * i + 2]), int(out[10 * i + 3]), out[10 * i + 4:10 * i + 10].tobytes().split(b"\0")[0].decode())
            for i in range(n)], bool(r >> 16)


def oracle_api_ul_dcch(api_mode, pdu, rnti, tti):
    o = oracle()
    o.o_api_ul_dcch_events.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_uint16, C.c_uint32, C.POINTER(OApiEvent), C.c_int, C.POINTER(C.c_int)]
    ev, n = (OApiEvent * 10)(), C.c_int(0)
    keep = o.o_api_ul_dcch_events(api_mode, bytes(pdu), len(pdu), rnti, tti, ev, 10, C.byref(n))
    return [(e.tti, e.rnti, e.id_type, e.msg_type, e.value.decode()) for e in ev[:n.value]], bool(keep)


def host_api_ul_dcch(api_mode, pdu, rnti, tti):
    h = hosttest()
    out = np.zeros(10 * 10, np.uint32)
    r = h.lsnh_api_ul_dcch_events(api_mode, bytes(pdu), len(pdu), rnti, tti, out.ctypes.data_as(C.c_void_p), 10)
    n = r & 0xFFFF
    return [(int(out[10 * i]), int(out[10 * i + 1]), int(out[10 * i + 2]), int(out[10 * i + 3]), out[10 * i + 4:10 * i + 10].tobytes().split(b"\0")[0].decode())
            for i in range(n)], bool(r >> 16)


def host_api_events(api_mode, name, pdu, rnti, tti):
    h = hosttest()
    out = np.zeros(10 * 20, np.uint32)
    r = h.lsnh_api_dl_events(api_mode, ord(name), bytes(pdu), len(pdu), rnti, tti, out.ctypes.data_as(C.c_void_p), 20)
    n = r & 0xFFFF
    return [(int(out[10 * i]), int(out[10 * i + 1]), int(out[10 * i + 2]), int(out[10 * i + 3]), out[10 * i + 4:10 * i + 10].tobytes().split(b"\0")[0].decode())
            for i in range(n)], bool(r >> 16)


class OUlCfg(C.Structure):
    _fields_ = [("cyclic_shift", C.c_uint32), ("delta_ss", C.c_uint32), ("hopping_offset", C.c_uint32), ("group_hopping_enabled", C.c_uint32),
                ("sequence_hopping_enabled", C.c_uint32)]


class OPuschGrant(C.Structure):  # o_pusch_grant_t (n_prb2 / hop: type-1 frequency hopping, slot 1 on other PRBs)
    _fields_ = [("L_prb", C.c_uint32), ("n_prb", C.c_uint32), ("mcs_idx", C.c_uint32), ("mod", C.c_int), ("tbs", C.c_int), ("rv", C.c_int),
                ("n_prb2", C.c_uint32), ("hop", C.c_uint32)]


VALID_UL_PRB = [n for n in range(1, 101) if (lambda m: all(m % p for p in (7, 11, 13)) and max([q for q in range(2, m + 1) if m % q == 0 and all(q % d for d in range(2, q))] or [1]) <= 5)(n)]


def ul_mcs_to_mod_tbs(mcs, L, enable_64qam=True):
    """Table 8.6.1-1 (ul_sniffer_fill_ra_mcs, ul_sniffer_pusch.c:176-200): -> (Qm, tbs)"""
    o = oracle()
    o.o_tbs_from_idx.restype = C.c_int
    if mcs < 11:
        return 2, o.o_tbs_from_idx(mcs, L)
    if mcs < 21:
        return 4, o.o_tbs_from_idx(mcs - 1, L)
    return (6 if enable_64qam else 4), o.o_tbs_from_idx(mcs - 2, L)


def ul_mcs_to_mod_tbs_256(mcs, L):
    """36.213 Table 8.6.1-3 (uplink MCS table with 256QAM), restated for the synthetic UE: -> (Qm, tbs); (0, 0) where the UE stays silent
    (MCS 26 needs the 32A TBS row, MCS 29-31 are retransmissions)"""
    o = oracle()
    o.o_tbs_from_idx.restype = C.c_int
    if mcs < 6:
        return 2, o.o_tbs_from_idx(2 * mcs, L)
    if mcs < 14:
        return 4, o.o_tbs_from_idx(mcs + (5 if mcs < 10 else 6), L)
    if mcs < 23:
        return 6, o.o_tbs_from_idx(mcs + (6 if mcs < 19 else 7), L)
    if mcs < 26:
        return 8, o.o_tbs_from_idx(mcs + 7, L)
    if mcs in (27, 28):
        return 8, o.o_tbs_from_idx(mcs + 6, L)
    return 0, 0


def ul_make_subframe(cell, tti, grants, snr_db=30.0, seed=1):
    """grants: list of dict(rnti, n_dmrs, n_prb, L_prb, mod, tbs, rv, gain_db, phase_rad, ta_samples)
    -> (iq complex64[15N], [payload bytes per grant])"""
    lib = txgen()
    lib.txg_ul_make.argtypes = [C.POINTER(TxgUlCell), C.c_uint32, C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    N = {6: 128, 15: 256, 25: 512, 50: 1024, 75: 1536, 100: 2048}[cell.nof_prb]
    iq = np.zeros(15 * N, dtype=np.complex64)
    arr = (TxgUlGrant * max(1, len(grants)))(*[TxgUlGrant(g["rnti"], g.get("n_dmrs", 0), g["n_prb"], g["L_prb"], g["mod"], g["tbs"], g.get("rv", 0),
                                                           g.get("gain_db", 0.0), g.get("phase_rad", 0.0), g.get("ta_samples", 0.0),
                                                           g.get("nof_ack", 0), g.get("cqi_bits", 0), g.get("ri_bits", 0), g.get("hop", 0), g.get("n_prb2", 0),
                                                           g.get("i_ack_p1", 0), g.get("i_cqi_p1", 0), g.get("i_ri_p1", 0))
                                               for g in grants])
    pbuf = np.zeros(sum(g["tbs"] // 8 for g in grants) + 16, dtype=np.uint8)
    offs = (C.c_uint32 * max(1, len(grants)))()
    n = lib.txg_ul_make(C.byref(cell), tti, arr, len(grants), snr_db, seed, iq.ctypes.data, pbuf.ctypes.data, offs)
    assert n >= 0
    return iq, [bytes(pbuf[offs[i]:offs[i] + g["tbs"] // 8]) for i, g in enumerate(grants)]


def oracle_ul_api():
    o = oracle()
    o.o_ul_fft.argtypes = [C.POINTER(OCell), C.c_void_p, C.c_void_p]
    o.o_pusch_demod.argtypes = [C.POINTER(OCell), C.POINTER(OUlCfg), C.c_uint32, C.c_uint16, C.POINTER(OPuschGrant), C.c_uint32, C.c_void_p, C.c_void_p,
                                C.POINTER(C.c_float), C.POINTER(C.c_float)]
    o.o_pusch_decode.argtypes = [C.POINTER(OCell), C.POINTER(OUlCfg), C.c_uint32, C.c_uint16, C.POINTER(OPuschGrant), C.c_uint32, C.c_void_p, C.c_int,
                                 C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)]
    o.o_ul_valid_prb.argtypes = [C.c_uint32]
    o.o_pusch_demod_uci.argtypes = [C.POINTER(OCell), C.POINTER(OUlCfg), C.c_uint32, C.c_uint16, C.POINTER(OPuschGrant), C.c_uint32, C.POINTER(OUci), C.c_void_p,
                                    C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    o.o_pusch_decode_uci.argtypes = [C.POINTER(OCell), C.POINTER(OUlCfg), C.c_uint32, C.c_uint16, C.POINTER(OPuschGrant), C.c_uint32, C.POINTER(OUci), C.c_void_p,
                                     C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)]
    o.o_uci_layout.argtypes = [C.c_int, C.c_int, C.POINTER(OUci), C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    o.o_uci_cqi_bits.argtypes = [C.c_uint32]
    return o


def oracle_worker_set_api(ow, api_mode):
    """-a of the reference on an oracle worker; -> the in-memory API pcap handle (read with ow.lib.o_pcap_* like ow.pcap)"""
    ow.lib.o_worker_set_api.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    ow.lib.o_worker_api_events.argtypes = [C.c_void_p, C.POINTER(OApiEvent), C.c_int]
    ow.lib.o_pcap_open_mem.restype = C.c_void_p
    ow.api_pcap = ow.lib.o_pcap_open_mem()
    ow.lib.o_worker_set_api(ow.h, api_mode, ow.api_pcap)
    return ow.api_pcap


def oracle_worker_api_events(ow):
    n = ow.lib.o_worker_api_events(ow.h, None, 0)
    ev = (OApiEvent * max(1, n))()
    ow.lib.o_worker_api_events(ow.h, ev, n)
    return [(e.tti, e.rnti, e.id_type, e.msg_type, e.value.decode()) for e in ev[:n]]


class OracleWorkerUl(OracleWorker):
    """UL_MODE worker: one downlink antenna + the uplink antenna (SubframeWorker.cc:184-199)"""

    def __init__(self, nof_prb, nof_ports, cell_id, cyclic_shift, delta_ss, hopping_offset=0, group_hopping=0, sequence_hopping=0, **kw):
        """cyclic_shift None: no configuration given - the worker configures itself from the first SIB2 (decode_SIB)"""
        super().__init__(nof_prb, nof_ports, cell_id, 1, **kw)
        self.lib.o_worker_set_ul_mode.argtypes = [C.c_void_p, C.POINTER(OUlCfg)]
        self.lib.o_worker_work_ul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
        self.lib.o_worker_ul_config.argtypes = [C.c_void_p, C.POINTER(OUlCfg), C.POINTER(OSib2)]
        if cyclic_shift is None:
            self.lib.o_worker_set_ul_mode(self.h, None)
        else:
            self._ul = OUlCfg(cyclic_shift, delta_ss, hopping_offset, int(group_hopping), int(sequence_hopping))
            self.lib.o_worker_set_ul_mode(self.h, C.byref(self._ul))

    def ul_config(self):
        """-> None | dict(cyclic_shift, delta_ss, hopping_offset, from_sib2, sib2)"""
        u, s = OUlCfg(), OSib2()
        r = self.lib.o_worker_ul_config(self.h, C.byref(u), C.byref(s))
        if r == 0:
            return None
        return dict(cyclic_shift=u.cyclic_shift, delta_ss=u.delta_ss, hopping_offset=u.hopping_offset, group_hopping=u.group_hopping_enabled,
                    sequence_hopping=u.sequence_hopping_enabled, from_sib2=r == 2,
                    sib2=s.as_dict() if r == 2 else None)

    def work_ul(self, dl_iq, ul_iq, tti, update_meta=0):
        dl_iq = np.ascontiguousarray(dl_iq, dtype=np.complex64)
        ul_iq = np.ascontiguousarray(ul_iq, dtype=np.complex64)
        return self.lib.o_worker_work_ul(self.h, dl_iq.ctypes.data, ul_iq.ctypes.data, tti % 10, (tti // 10) % 1024, int(update_meta))


# the two BCCH-DL-SCH messages of the reference's own UL_MODE / DL_MODE captures (tests/golden/pcap_records.json "si_pdus")
REAL_SIB1 = bytes.fromhex("406404ab00070019b0181460108280000000")
REAL_SIB2 = bytes.fromhex("00800ce1bf788800ca11e20140000801829945ab9c30c6a73141c21462d84ea5a40000000000000000")


def encode_sib2(n_sb=1, hopping_mode=0, pusch_hop_offset=0, enable_64qam=1, group_hopping_enabled=0, group_assignment_pusch=0,
                sequence_hopping_enabled=0, cyclic_shift=0, root_seq_idx=128, prach_config_idx=3, high_speed_flag=0, zero_corr_zone=5,
                prach_freq_offset=4, ac_barring=0, group_a=0, srs=0, mbsfn=0, ul_carrier=0, ul_bw=0, rach_ext=0, rr_ext=0, timers_ext=0,
                sib_ext=0, nblocks=1, fill=0):
    """BCCH-DL-SCH-Message { systemInformation { sib2 ... } } in unaligned PER (TS 36.331 6.2.2 / 6.3.1 / 6.3.2) - an independent
    encoder for the tests: optional components / extension additions can be switched on, `fill` seeds the don't-care fields"""
    bits = []
    rnd = np.random.RandomState(fill)

    def put(v, n):
        bits.extend((int(v) >> (n - 1 - i)) & 1 for i in range(n))

    def dc(n, lim=None):  # a don't-care field
        put(rnd.randint(0, lim if lim is not None else (1 << n)) if fill else 0, n)

    def ext_additions(octets):  # one extension addition group, `octets` bytes of open type
        put(0, 1); put(0, 6)    # normally small number: 1 addition
        put(1, 1)               # present
        put(octets, 8)          # length determinant < 128
        for _ in range(octets):
            dc(8)

    put(0, 1); put(0, 1); put(0, 1); put(0, 1)  # c1, systemInformation, systemInformation-r8, no nonCriticalExtension
    put(nblocks - 1, 5)
    put(0, 1); put(0, 4)                         # sib2
    put(sib_ext, 1); put(1 if ac_barring else 0, 1); put(1 if mbsfn else 0, 1)
    if ac_barring:
        put(1, 1); put(ac_barring > 1, 1); dc(1)
        dc(4); dc(3); dc(5)
        if ac_barring > 1:
            dc(4); dc(3); dc(5)
    put(rr_ext, 1)
    put(rach_ext, 1); put(group_a, 1); dc(4)     # rach-ConfigCommon
    if group_a:
        put(0, 1); dc(4, 15); dc(2); dc(3)
    dc(2); dc(4); dc(4, 11); dc(3); dc(3); dc(3)
    if rach_ext:
        ext_additions(2)
    dc(2)                                        # bcch-Config
    dc(2); dc(3)                                 # pcch-Config
    put(root_seq_idx, 10); put(prach_config_idx, 6); put(high_speed_flag, 1); put(zero_corr_zone, 4); put(prach_freq_offset, 7)
    dc(7, 111); dc(2)                            # pdsch-ConfigCommon
    put(n_sb - 1, 2); put(hopping_mode, 1); put(pusch_hop_offset, 7); put(enable_64qam, 1)
    put(group_hopping_enabled, 1); put(group_assignment_pusch, 5); put(sequence_hopping_enabled, 1); put(cyclic_shift, 3)
    dc(2, 3); dc(7, 99); dc(3); dc(11)           # pucch-ConfigCommon
    put(1 if srs else 0, 1)
    if srs:
        put(srs > 1, 1); dc(3); dc(4); dc(1)
    dc(8, 151); dc(3); dc(5)                     # uplinkPowerControlCommon
    dc(2, 3); dc(2, 3); dc(2); dc(2, 3); dc(2, 3)
    dc(3)
    dc(1)                                        # ul-CyclicPrefixLength
    if rr_ext:
        ext_additions(3)
    put(timers_ext, 1); dc(3); dc(3); dc(3, 7); dc(3); dc(3, 7); dc(3)
    if timers_ext:
        ext_additions(1)
    put(ul_carrier, 1); put(ul_bw, 1)
    if ul_carrier:
        dc(16)
    if ul_bw:
        dc(3, 6)
    dc(5)
    if mbsfn:
        put(mbsfn - 1, 3)
        for i in range(mbsfn):
            dc(3, 6); dc(3); put(i & 1, 1); dc(24 if i & 1 else 6)
    dc(3)                                        # timeAlignmentTimerCommon
    if sib_ext:
        ext_additions(4)
    for _ in range(nblocks - 1):                 # further blocks: not read by the decoders under test
        for _ in range(5):
            dc(8)
    while len(bits) % 8:
        bits.append(0)
    return bytes(int("".join(map(str, bits[i:i + 8])), 2) for i in range(0, len(bits), 8))


def encode_paging(records, sys_info_mod=0, etws=0, ext_record=None):
    """PCCH-Message { paging { pagingRecordList } } in unaligned PER (TS 36.331 6.2.2); records: ("imsi", "262011234567890") or
    ("tmsi", mmec, m_tmsi); ext_record: index of a record sent with an (empty-content) extension addition"""
    bits = []

    def put(v, n):
        bits.extend((int(v) >> (n - 1 - i)) & 1 for i in range(n))
    put(0, 1)
    put(1 if records else 0, 1); put(sys_info_mod, 1); put(etws, 1); put(0, 1)
    if records:
        put(len(records) - 1, 4)
        for i, r in enumerate(records):
            ext = ext_record == i
            put(ext, 1); put(0, 1)
            if r[0] == "tmsi":
                put(0, 1); put(r[1], 8); put(r[2], 32)
            else:
                put(1, 1); put(len(r[1]) - 6, 4)
                for d in r[1]:
                    put(int(d), 4)
            put(i & 1, 1)
            if ext:
                put(0, 1); put(0, 6); put(1, 1); put(1, 8); put(0xA5, 8)
    while len(bits) % 8:
        bits.append(0)
    return bytes(int("".join(map(str, bits[i:i + 8])), 2) for i in range(0, len(bits), 8))


def gen_ul_mode_subframes(sc, n, cyclic_shift=3, delta_ss=5, ul_snr_db=30.0, si_msgs=None, ul_256=False, group_hopping=0, sequence_hopping=0):  # sc['pusch_hop_offset'] = SIB2 pusch-HoppingOffset of the cell
    """DL stream from the synthetic eNB (one rx antenna) + the matching UL stream: every DCI 0 of subframe t is answered by a
    PUSCH in subframe t + 4 (UEs with an even RNTI are 64QAM-capable in the uplink; with ul_256 every fourth UE uses the 256QAM table).
    -> (tti0, iq[n, 2, sf_len] (antenna 0 = DL, 1 = UL), list of sent UL payload dicts)"""
    assert sc["nof_rx"] == 1
    tx = TxGen(si_msgs=si_msgs, **sc)
    ucell = TxgUlCell(sc["nof_prb"], sc["cell_id"], cyclic_shift, delta_ss, int(group_hopping), int(sequence_hopping), sc.get("cp", 0))
    iq = np.zeros((n, 2, tx.sf_len), dtype=np.complex64)
    pending, sent, tti0 = {}, [], None
    # the uplink control configuration every UE transmits with: its own RRCConnectionSetup once it got one, else what a sniffer that
    # follows the cell assumes - the first connection setup ever seen (update_default_ue_config), before that 10 / 8 / 11, sub-band reports
    o = oracle()
    o.o_mac_dlsch_parse.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int]
    o.o_rrc_conn_setup_decode.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    o.o_uci_cqi_bits_type.argtypes = [C.c_uint32, C.c_uint32]
    default_cfg, ue_cfg = (10, 8, 11, 2), {}

    def setups(pdu):
        sub = (C.c_uint32 * (4 * 20))()
        ns = o.o_mac_dlsch_parse(pdu, len(pdu), sub, 20)
        for k in range(ns):
            lcid, is_sdu, off, ln = sub[4 * k:4 * k + 4]
            if is_sdu and lcid == 0:
                out = (C.c_uint32 * 7)()
                if o.o_rrc_conn_setup_decode(pdu[off:off + ln], ln, out):
                    yield (out[2], out[3], out[4], out[5])  # i_offset_ack, i_offset_cqi, i_offset_ri, cqi_type
    nsetup = 0
    for i in range(n):
        tti, x, pdus = tx.next()
        if tti0 is None:
            tti0 = tti
        iq[i, 0] = x[0]
        for p in pdus:  # the downlink of a subframe is handled before its uplink (SubframeWorker.cc:299-347)
            if not p["is_ul"] and 10 < p["rnti"] < 0xFFF4 and p["format"] in (1, 2):  # TXG_FMT1 / TXG_FMT1A: what decode_ul_mode decodes
                for c in setups(p["payload"]):
                    if nsetup == 0:
                        default_cfg = c
                    ue_cfg[p["rnti"]] = c
                    nsetup += 1
        grants = pending.pop(tti, [])
        for g in grants:
            ia, ic, ir, ct = ue_cfg.get(g["rnti"], default_cfg)
            g.update(i_ack_p1=ia + 1, i_cqi_p1=ic + 1, i_ri_p1=ir + 1, cqi_bits=o.o_uci_cqi_bits_type(sc["nof_prb"], ct) if g["cqi_req"] else 0)
        ul, pl = ul_make_subframe(ucell, tti, grants, snr_db=ul_snr_db, seed=sc["seed"] * 1000 + i)
        iq[i, 1] = ul
        for g, p in zip(grants, pl):
            sent.append(dict(tti=tti, rnti=g["rnti"], payload=p, L_prb=g["L_prb"], mod=g["mod"], hop=g.get("hop", 0)))
        for p in pdus:
            if p["is_ul"]:
                if ul_256 and p["rnti"] % 4 == 0:  # every fourth UE is configured with the 256QAM uplink table
                    qm, tbs = ul_mcs_to_mod_tbs_256(p["mcs"], p["nof_prb"])
                else:
                    qm, tbs = ul_mcs_to_mod_tbs(p["mcs"], p["nof_prb"], enable_64qam=(p["rnti"] % 2 == 0))
                if tbs > 0:
                    # the UE acknowledges the downlink transport blocks it was sent in this subframe on the PUSCH 4 ms later and adds the
                    # aperiodic CSI report (type per its configuration + RI) when the DCI 0 asks for one (36.212 5.2.2.6)
                    ntb = len([q for q in pdus if not q["is_ul"] and q["rnti"] == p["rnti"]])
                    hop, n2 = 0, 0
                    if p["hop_bits_plus1"]:
                        n2 = pusch_hop_slot1(sc["nof_prb"], sc["pusch_hop_offset"], p["hop_bits_plus1"] - 1, p["n_prb"])
                        hop = 1
                    pending.setdefault((tti + 4) % 10240, []).append(dict(rnti=p["rnti"], n_dmrs=0, n_prb=p["n_prb"], L_prb=p["nof_prb"], mod=qm, tbs=tbs, rv=0,
                                                                          nof_ack=min(ntb, 2), cqi_req=bool(p["cqi_req"]), ri_bits=1 if p["cqi_req"] else 0,
                                                                          hop=hop, n_prb2=n2 or 0))
    return tti0, iq, sent


# ---------------------------------------------------------------------------------------------------- PRACH
PRACH_NCS = [0, 13, 15, 18, 22, 26, 32, 38, 46, 59, 76, 93, 119, 167, 279, 419]  # 36.211 Table 5.7.2-2 (unrestricted)


class OPrachCfg(C.Structure):
    _fields_ = [("config_idx", C.c_uint32), ("root_seq_idx", C.c_uint32), ("zero_corr_zone", C.c_uint32), ("freq_offset", C.c_uint32),
                ("hs_flag", C.c_uint32), ("detect_factor", C.c_float), ("zc_roots", C.POINTER(C.c_uint16))]


class OPrachDet(C.Structure):
    _fields_ = [("preamble", C.c_uint32), ("offset", C.c_uint32), ("offset_sec", C.c_float), ("p2avg", C.c_float)]


def oracle_prach_api():
    o = oracle()
    o.o_prach_detect.argtypes = [C.POINTER(OCell), C.POINTER(OPrachCfg), C.c_void_p, C.POINTER(OPrachDet), C.c_int, C.c_void_p]
    o.o_prach_tti_opportunity.argtypes = [C.c_uint32, C.c_uint32]
    o.o_prach_nof_roots.argtypes = [C.c_uint32]
    o.o_prach_nof_roots.restype = C.c_uint32
    return o


def prach_preamble(nof_prb, u, cv, freq_offset):
    """36.211 5.7.3 baseband PRACH signal, preamble format 0, physical root u, cyclic shift C_v = cv, at the cell's sample
    rate (N_cp + 12 N samples, unit mean power) - an independent transmitter (numpy FFTs in double)"""
    nsym = oracle().o_fft_size(nof_prb)
    n_seq, n_cp = 12 * nsym, 3168 * nsym // 2048
    n = np.arange(839, dtype=np.int64)
    x = np.exp(-1j * np.pi * ((u * n * (n + 1)) % (2 * 839)) / 839.0)
    spec = np.fft.fft(x[(n + cv) % 839])
    b0 = 7 + 12 * (12 * freq_offset - 6 * nof_prb) + 6
    full = np.zeros(n_seq, dtype=np.complex128)
    full[(b0 + n) % n_seq] = spec
    s = np.fft.ifft(full)
    s /= np.sqrt(np.mean(np.abs(s) ** 2))
    return np.concatenate([s[-n_cp:], s])


def prach_subframe(nof_prb, ues, snr_db=10.0, seed=1, zero_corr_zone=5, root_seq_idx=10, freq_offset=4, zc_roots=None):
    """one uplink subframe (15 N samples) with the PRACH preambles of `ues` = [(preamble index, delay in samples, gain dB)]
    on top of unit-power noise scaled to snr_db below a 0 dB preamble"""
    sf_len = 15 * oracle().o_fft_size(nof_prb)
    rng = np.random.default_rng(seed)
    sigma = 10.0 ** (-snr_db / 20.0)
    iq = (rng.standard_normal(sf_len) + 1j * rng.standard_normal(sf_len)) * (sigma / np.sqrt(2.0))
    ncs = PRACH_NCS[zero_corr_zone]
    nwin = 839 // ncs if ncs else 1
    for idx, delay, gain_db in ues:
        lr = (root_seq_idx + idx // nwin) % 838
        u = int(zc_roots[lr]) if zc_roots is not None else lr + 1
        p = prach_preamble(nof_prb, u, (idx % nwin) * ncs, freq_offset) * 10.0 ** (gain_db / 20.0)
        iq[delay:delay + len(p)] += p[:sf_len - delay]
    return iq.astype(np.complex64)


# -------- PSS / SSS cell search (oracle/o_sync.c) --------
class OSyncCfg(C.Structure):
    _fields_ = [("nof_periods", C.c_uint32), ("force_n_id_2", C.c_int32), ("threshold", C.c_float)]


class OSync(C.Structure):
    _fields_ = [("found", C.c_uint32), ("cell_id", C.c_uint32), ("n_id_2", C.c_uint32), ("n_id_1", C.c_uint32), ("sf_idx", C.c_uint32),
                ("pss_pos", C.c_uint32), ("sf_start", C.c_uint32), ("pss_peak", C.c_float), ("pss_p2avg", C.c_float),
                ("sss_metric", C.c_float), ("sss_second", C.c_float), ("cfo_hz", C.c_float), ("cfo_coarse_hz", C.c_float), ("cp", C.c_uint32)]


def oracle_sync_api():
    o = oracle()
    o.o_pss_seq.argtypes = [C.c_uint32, C.c_void_p]
    o.o_sss_m0m1.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    o.o_sss_seq.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    o.o_pss_time.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
    o.o_sync_min_samples.argtypes = [C.c_uint32, C.c_uint32]
    o.o_sync_min_samples.restype = C.c_uint32
    o.o_cell_search.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(OSyncCfg), C.POINTER(OSync), C.c_void_p]
    return o


def sync_capture(sc, lead, periods, seed=11):
    """antenna 0 of a txgen capture behind `lead` samples of receiver noise -> (samples, tti of the first whole subframe)"""
    tx = TxGen(**sc)
    need = oracle_sync_api().o_sync_min_samples(sc["nof_prb"], periods)
    rng = np.random.default_rng(seed)
    parts = [(0.02 * (rng.standard_normal(lead) + 1j * rng.standard_normal(lead))).astype(np.complex64)]
    first_tti, have = None, lead
    while have < need:
        tti, iq, _ = tx.next()
        if first_tti is None:
            first_tti = tti
        parts.append(iq[0])
        have += iq.shape[1]
    return np.ascontiguousarray(np.concatenate(parts)), first_tti
