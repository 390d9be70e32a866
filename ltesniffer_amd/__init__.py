"""ltesniffer_amd - MI355X-native (gfx950, HIP) replacement of LTESniffer's per-subframe worker.

The product is the C-ABI shared library ``ltesniffer_amd/lib/libltesniffer_amd.so`` (include/ltesniffer_amd.h); this
module is only the ctypes binding used by tests, ``bench.py`` and ``__graft_entry__`` plus thin Python mirrors of the
reference's ``Phy`` / ``SubframeWorker`` pair (/root/reference/src/include/Phy.h:22-66,
/root/reference/src/include/SubframeWorker.h:16-86).  There is no CPU fallback: importing works without a GPU (so
that the symbol table can be checked), but creating a ``Phy`` without a HIP device raises.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

# (GPU_MAX_HW_QUEUES: the engine's 12 decode chains + 4 stage-A chains want 16 hardware queues, the HIP runtime's default is 4 and the value is read at
#  runtime initialisation.  That is the HOST PROGRAM's business - bench.py and tests/conftest.py export it; importing this module changes nothing in the
#  process environment (round-4 advisor finding) and the engine adapts its chain count to what it finds)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LSN_LIB_PATH") or os.path.join(_HERE, "lib", "libltesniffer_amd.so")  # LSN_LIB_PATH: A/B builds of the same library (tools/)

LSN_SUCCESS, LSN_ERROR, LSN_ERROR_INVALID_INPUTS, LSN_ERROR_NO_DEVICE = 0, -1, -2, -3
TAP_GRID, TAP_CE, TAP_PDCCH_LLR, TAP_CHEST, TAP_CFI, TAP_CANDIDATES, TAP_CCE_POWER, TAP_ACCEPTED, TAP_RB_POWER = range(9)
TAP_PDSCH_JOBS, TAP_PDSCH_LLR16, TAP_RM_WORDS, TAP_CB_RESULT = 9, 10, 11, 12   # stage C: indexed by decode job (lsn_phy_set_stage_c_taps)
KERNELS = ["k_ofdm", "k_chest", "k_chest_fin", "k_pcfich", "k_pdcch_llr", "k_cce_power", "k_viterbi", "k_pdsch_prep",
           "k_pdsch_demod", "k_turbo<64>", "k_rb_power", "k_turbo<128>", "k_rm"]

# every symbol include/ltesniffer_amd.h declares
EXPORTS = ["lsn_phy_create", "lsn_phy_destroy", "lsn_phy_set_cell", "lsn_phy_get_avail", "lsn_phy_put_pending",
           "lsn_phy_join_pending", "lsn_phy_set_pdu_sink", "lsn_phy_get_stats", "lsn_phy_get_est_cfo",
           "lsn_phy_add_evergreen", "lsn_phy_add_forbidden", "lsn_phy_setup_default_rnti_intervals",
           "lsn_phy_nof_active_rnti", "lsn_phy_get_ue_config", "lsn_worker_buffers", "lsn_worker_buffer_len", "lsn_worker_prepare",
           "lsn_worker_sf_idx", "lsn_worker_sfn", "lsn_phy_process_device", "lsn_phy_process_host", "lsn_phy_process_host_int", "lsn_phy_tap",
           "lsn_phy_get_perf", "lsn_kernel_name", "lsn_version", "lsn_pcap_open", "lsn_pcap_open_mem",
           "lsn_pcap_set_wall_clock", "lsn_pcap_write", "lsn_pcap_sink", "lsn_pcap_mem", "lsn_pcap_nof_records",
           "lsn_pcap_reset", "lsn_pcap_close", "lsn_phy_set_pcap_writer", "lsn_phy_set_api_mode", "lsn_phy_tracked_ul_modulation", "lsn_phy_set_ul_config", "lsn_phy_get_ul_config", "lsn_sib2_decode", "lsn_phy_pusch_decode",
           "lsn_phy_tap_ul", "lsn_phy_set_prach_config", "lsn_phy_prach_detect", "lsn_phy_set_prach_sink", "lsn_prach_tti_opportunity", "lsn_phy_process_file", "lsn_phy_mib_decode", "lsn_phy_mib_decode_llr", "lsn_phy_submit_device", "lsn_phy_wait", "lsn_cell_search",
           "lsn_phy_set_shortcut_discovery", "lsn_phy_get_shortcut_discovery", "lsn_phy_set_histogram_threshold", "lsn_phy_print_stats",
           "lsn_phy_set_mcs_update_interval", "lsn_phy_update_mcs_database", "lsn_phy_nof_tracked_rnti", "lsn_worker_buffers_offset", "lsn_pcap_digest", "lsn_pcap_set_store", "lsn_pcap_set_digest_blocks", "lsn_pcap_block_digests", "lsn_phy_create_multi", "lsn_phy_nof_devices",
           "lsn_phy_set_cfo_correction", "lsn_phy_get_cfo_correction", "lsn_phy_set_candidate_pruning", "lsn_phy_set_stage_c_taps", "lsn_phy_prepare_file", "lsn_phy_get_meta_formats", "lsn_phy_nof_workers", "lsn_phy_worker"]


def turbo_nwin(K):
    """windows the turbo decoder cuts a block of K bits into (lsn_rm.h: lsn_turbo_nwin)"""
    p1 = p2 = 1
    for P in range(min(K // 32, 128), 0, -1):
        if K % P == 0:
            p2 = P
            break
    for P in range(min(K // 32, 64), 0, -1):
        if K % P == 0:
            p1 = P
            break
    return p2 if p2 >= 96 else p1


PRACH_NCS = [0, 13, 15, 18, 22, 26, 32, 38, 46, 59, 76, 93, 119, 167, 279, 419]  # 36.211 Table 5.7.2-2


class Mib(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("found", "sfn", "sfn_offset", "nof_prb", "nof_ports", "phich_length", "phich_resources_x6", "mib_bits")]


TTI_FROM_MIB = 0xFFFFFFFF


class UeConfig(C.Structure):
    _fields_ = [("has_ue_config", C.c_uint32), ("p_a_db", C.c_float), ("i_offset_ack", C.c_uint32), ("i_offset_cqi", C.c_uint32),
                ("i_offset_ri", C.c_uint32), ("cqi_type", C.c_uint32)]


class CellSearchCfg(C.Structure):
    _fields_ = [("nof_periods", C.c_uint32), ("force_n_id_2", C.c_int32), ("threshold", C.c_float)]


class CellSearch(C.Structure):
    _fields_ = [("found", C.c_uint32), ("cell_id", C.c_uint32), ("n_id_2", C.c_uint32), ("n_id_1", C.c_uint32), ("sf_idx", C.c_uint32),
                ("pss_pos", C.c_uint32), ("sf_start", C.c_uint32), ("pss_peak", C.c_float), ("pss_p2avg", C.c_float),
                ("sss_metric", C.c_float), ("sss_second", C.c_float), ("cfo_hz", C.c_float), ("cfo_coarse_hz", C.c_float), ("cp", C.c_uint32)]


class FileCfg(C.Structure):
    _fields_ = [("nof_antennas", C.c_uint32), ("offset_time_samples", C.c_int64), ("offset_freq_hz", C.c_float), ("sample_format", C.c_uint32),
                ("sample_scale", C.c_float)]


FILE_CF32, FILE_SC16, FILE_SC8 = 0, 1, 2


class ApiEvent(C.Structure):
    _fields_ = [("tti", C.c_uint32), ("rnti", C.c_uint16), ("id_type", C.c_uint32), ("msg_type", C.c_uint32), ("value", C.c_char * 24)]


API_SINK = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(ApiEvent))


class Sib2(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n_sb", "hopping_mode", "pusch_hop_offset", "enable_64qam", "group_hopping_enabled", "group_assignment_pusch",
                                          "sequence_hopping_enabled", "cyclic_shift", "root_seq_idx", "prach_config_idx", "high_speed_flag",
                                          "zero_corr_zone", "prach_freq_offset")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def sib2_decode(pdu):
    """BCCH-DL-SCH-Message -> (verdict 0 / 1 / 2, dict of the SIB2 fields or None)"""
    s = Sib2()
    r = lib().lsn_sib2_decode(bytes(pdu), len(pdu), C.byref(s))
    return r, (s.as_dict() if r == 2 else None)


class PrachCfg(C.Structure):
    _fields_ = [("config_idx", C.c_uint32), ("root_seq_idx", C.c_uint32), ("zero_corr_zone", C.c_uint32), ("freq_offset", C.c_uint32),
                ("hs_flag", C.c_uint32), ("detect_factor", C.c_float), ("zc_roots", C.POINTER(C.c_uint16))]


class PrachDet(C.Structure):
    _fields_ = [("sf", C.c_uint32), ("preamble", C.c_uint32), ("offset", C.c_uint32), ("offset_sec", C.c_float), ("p2avg", C.c_float)]


PRACH_SINK = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.POINTER(PrachDet), C.c_uint32)


class Cell(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("nof_prb", "nof_ports", "id", "cp", "phich_length", "phich_resources", "frame_type")]


class DlSfCfg(C.Structure):
    _fields_ = [("tti", C.c_uint32), ("cfi", C.c_uint32), ("sf_type", C.c_uint32)]


class PhyCfg(C.Structure):
    _fields_ = [("nof_rx_antennas", C.c_uint32), ("nof_workers", C.c_uint32), ("max_batch", C.c_uint32),
                ("skip_secondary_meta_formats", C.c_int), ("meta_format_split_ratio", C.c_double),
                ("histogram_threshold", C.c_uint32), ("mcs_tracking_mode", C.c_int), ("harq_mode", C.c_int),
                ("device", C.c_int), ("max_turbo_iterations", C.c_int), ("sniffer_mode", C.c_int)]


class PduCtx(C.Structure):
    _fields_ = [("tti", C.c_uint32), ("rnti", C.c_uint16), ("direction", C.c_uint8), ("rnti_type", C.c_uint8),
                ("crc_ok", C.c_uint8), ("is_retx", C.c_uint8), ("tb", C.c_uint8), ("reserved", C.c_uint8)]


class BlindStats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("nof_locations", "nof_decoded_locations", "nof_cce", "nof_missed_cce",
                                          "nof_subframes", "nof_subframe_collisions_dw", "nof_subframe_collisions_up")]


class Perf(C.Structure):
    _fields_ = [("ms_stage_a", C.c_double), ("ms_search", C.c_double), ("ms_stage_c", C.c_double), ("ms_commit", C.c_double),
                ("ms_total", C.c_double), ("kernel_ms", C.c_double * 16), ("kernel_launches", C.c_uint64 * 16),
                ("algo_bytes", C.c_uint64), ("turbo_algo_bytes", C.c_uint64), ("turbo128_algo_bytes", C.c_uint64),
                ("nof_tb_decodes", C.c_uint64),
                ("nof_cb_decodes", C.c_uint64), ("nof_turbo_iterations", C.c_uint64), ("nof_candidates_decoded", C.c_uint64),
                ("nof_ondemand_decodes", C.c_uint64), ("nof_pdus", C.c_uint64), ("ms_search_core", C.c_double), ("ms_rar", C.c_double), ("turbo_cyc_rm", C.c_uint64),
                ("turbo_cyc_map", C.c_uint64), ("turbo_cyc_out", C.c_uint64), ("ms_wait_front", C.c_double), ("ms_wait_slot", C.c_double), ("ms_drain", C.c_double),
                ("nof_turbo_iterations_run", C.c_uint64), ("nof_ondemand_commit", C.c_uint64 * 4), ("ms_ondemand_commit", C.c_double), ("nof_pusch_2prb_skipped", C.c_uint64), ("nof_pusch_on_unverified_dmrs", C.c_uint64), ("nof_tb_on_derived_tbs", C.c_uint64),
                ("nof_decode_jobs", C.c_uint64), ("nof_decode_jobs_used", C.c_uint64), ("nof_speculative_jobs", C.c_uint64),
                ("jobs_by_kind", C.c_uint64 * 5), ("jobs_unused_by_kind", C.c_uint64 * 5), ("iters_by_kind", C.c_uint64 * 5), ("iters_unused_by_kind", C.c_uint64 * 5),
                ("nof_table_hints_used", C.c_uint64), ("nof_table_hints_missed", C.c_uint64), ("nof_candidate_misses", C.c_uint64), ("ms_harq", C.c_double * 3), ("nof_harq_combines", C.c_uint64 * 4)]


class UlCfg(C.Structure):
    _fields_ = [("cyclic_shift", C.c_uint32), ("delta_ss", C.c_uint32), ("hopping_offset", C.c_uint32), ("group_hopping_enabled", C.c_uint32),
                ("sequence_hopping_enabled", C.c_uint32)]


class PuschGrant(C.Structure):
    _fields_ = [("sf", C.c_uint32), ("rnti", C.c_uint16), ("n_dmrs", C.c_uint16), ("n_prb", C.c_uint32), ("L_prb", C.c_uint32),
                ("mod", C.c_uint32), ("tbs", C.c_uint32), ("rv", C.c_int), ("nof_ack", C.c_uint32), ("cqi_bits", C.c_uint32), ("ri_bits", C.c_uint32),
                ("hop", C.c_uint32), ("n_prb_slot1", C.c_uint32),
                ("beta_offset_ack_idx_p1", C.c_uint32), ("beta_offset_cqi_idx_p1", C.c_uint32), ("beta_offset_ri_idx_p1", C.c_uint32)]


class PuschResult(C.Structure):
    _fields_ = [("crc_ok", C.c_uint32), ("iterations", C.c_uint32), ("snr_db", C.c_float), ("payload_off", C.c_uint32)]


SINK_T = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(PduCtx), C.POINTER(C.c_uint8), C.c_uint32)

_lib = None


def build(verbose=False):
    """Compile the HIP kernels + host engine for gfx950 (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"], stdout=out)
    return LIB_PATH


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1.  If this library pulled in
    /opt/rocm's copies first and torch were imported afterwards, the process would hold two HIP/HSA runtimes and device
    discovery fails.  When torch is installed, bind to its copy (one runtime per process, whichever is imported first)."""
    import importlib.util
    try:
        with open("/proc/self/maps") as f:
            if "libamdhip64" in f.read():
                return
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        d = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        for n in ("libhsa-runtime64.so", "libamdhip64.so"):
            p = os.path.join(d, n)
            if os.path.exists(p):
                C.CDLL(p, mode=C.RTLD_GLOBAL)
    except OSError:
        pass


def lib():
    """The C-ABI library. Fails loudly when it has not been built: there is no other implementation to fall back to."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("ltesniffer_amd: %s is missing - run ltesniffer_amd.build() / __graft_entry__.build() "
                               "(the HIP extension is the only implementation; there is no CPU fallback)" % LIB_PATH)
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.lsn_phy_create.argtypes = [C.POINTER(PhyCfg), C.POINTER(C.c_void_p)]
        L.lsn_phy_destroy.argtypes = [C.c_void_p]
        L.lsn_phy_create_multi.argtypes = [C.POINTER(PhyCfg), C.POINTER(C.c_int), C.c_uint32, C.POINTER(C.c_void_p)]
        L.lsn_phy_nof_devices.argtypes = [C.c_void_p]
        L.lsn_phy_nof_devices.restype = C.c_uint32
        L.lsn_phy_destroy.restype = None
        L.lsn_phy_set_cell.argtypes = [C.c_void_p, C.POINTER(Cell)]
        L.lsn_phy_get_avail.argtypes = [C.c_void_p, C.c_int]
        L.lsn_phy_get_avail.restype = C.c_void_p
        L.lsn_phy_put_pending.argtypes = [C.c_void_p, C.c_void_p]
        L.lsn_phy_join_pending.argtypes = [C.c_void_p]
        L.lsn_phy_set_pdu_sink.argtypes = [C.c_void_p, SINK_T, C.c_void_p]
        L.lsn_phy_get_stats.argtypes = [C.c_void_p, C.POINTER(BlindStats)]
        L.lsn_phy_get_est_cfo.argtypes = [C.c_void_p]
        L.lsn_phy_get_est_cfo.restype = C.c_float
        L.lsn_phy_set_cfo_correction.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float]
        L.lsn_phy_set_candidate_pruning.argtypes = [C.c_void_p, C.c_int]
        L.lsn_phy_get_cfo_correction.argtypes = [C.c_void_p]
        L.lsn_phy_get_cfo_correction.restype = C.c_float
        L.lsn_phy_add_evergreen.argtypes = [C.c_void_p, C.c_uint16, C.c_uint16, C.c_uint32]
        L.lsn_phy_add_forbidden.argtypes = [C.c_void_p, C.c_uint16, C.c_uint16, C.c_uint32]
        L.lsn_phy_setup_default_rnti_intervals.argtypes = [C.c_void_p]
        L.lsn_phy_nof_active_rnti.argtypes = [C.c_void_p]
        L.lsn_phy_nof_active_rnti.restype = C.c_uint32
        L.lsn_phy_get_ue_config.argtypes = [C.c_void_p, C.c_uint16, C.POINTER(UeConfig)]
        L.lsn_worker_buffers.argtypes = [C.c_void_p]
        L.lsn_worker_buffers.restype = C.POINTER(C.POINTER(C.c_float))
        L.lsn_worker_buffer_len.argtypes = [C.c_void_p]
        L.lsn_worker_buffer_len.restype = C.c_uint32
        L.lsn_worker_prepare.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(DlSfCfg)]
        L.lsn_worker_sf_idx.argtypes = [C.c_void_p]
        L.lsn_worker_sf_idx.restype = C.c_uint32
        L.lsn_worker_sfn.argtypes = [C.c_void_p]
        L.lsn_worker_sfn.restype = C.c_uint32
        L.lsn_phy_process_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.lsn_phy_process_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.lsn_phy_process_host_int.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32]
        L.lsn_phy_tap.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_size_t]
        L.lsn_phy_tap.restype = C.c_long
        L.lsn_phy_set_stage_c_taps.argtypes = [C.c_void_p, C.c_int]
        L.lsn_phy_get_perf.argtypes = [C.c_void_p, C.POINTER(Perf)]
        L.lsn_kernel_name.argtypes = [C.c_int]
        L.lsn_kernel_name.restype = C.c_char_p
        L.lsn_version.restype = C.c_char_p
        L.lsn_pcap_open.argtypes = [C.c_char_p]
        L.lsn_pcap_open.restype = C.c_void_p
        L.lsn_pcap_open_mem.restype = C.c_void_p
        L.lsn_pcap_set_wall_clock.argtypes = [C.c_void_p, C.c_int]
        L.lsn_pcap_set_wall_clock.restype = None
        L.lsn_pcap_write.argtypes = [C.c_void_p, C.POINTER(PduCtx), C.c_void_p, C.c_uint32]
        L.lsn_pcap_mem.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.lsn_pcap_mem.restype = C.c_void_p
        L.lsn_pcap_nof_records.argtypes = [C.c_void_p]
        L.lsn_pcap_nof_records.restype = C.c_uint32
        L.lsn_pcap_reset.argtypes = [C.c_void_p]
        L.lsn_pcap_reset.restype = None
        L.lsn_pcap_close.argtypes = [C.c_void_p]
        L.lsn_pcap_digest.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.lsn_pcap_set_store.argtypes = [C.c_void_p, C.c_int]
        L.lsn_pcap_set_store.restype = None
        L.lsn_pcap_set_digest_blocks.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.lsn_pcap_set_digest_blocks.restype = None
        L.lsn_pcap_block_digests.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.lsn_pcap_block_digests.restype = C.c_uint32
        L.lsn_pcap_close.restype = None
        L.lsn_phy_set_pcap_writer.argtypes = [C.c_void_p, C.c_void_p]
        L.lsn_phy_set_ul_config.argtypes = [C.c_void_p, C.POINTER(UlCfg)]
        L.lsn_phy_tracked_ul_modulation.argtypes = [C.c_void_p, C.c_uint16]
        L.lsn_phy_set_api_mode.argtypes = [C.c_void_p, C.c_int, API_SINK, C.c_void_p, C.c_void_p]
        L.lsn_phy_get_ul_config.argtypes = [C.c_void_p, C.POINTER(UlCfg), C.POINTER(Sib2), C.POINTER(C.c_uint32)]
        L.lsn_sib2_decode.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(Sib2)]
        L.lsn_phy_pusch_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                           C.c_void_p, C.c_size_t]
        L.lsn_phy_tap_ul.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_size_t]
        L.lsn_phy_tap_ul.restype = C.c_long
        L.lsn_phy_submit_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.lsn_phy_wait.argtypes = [C.c_void_p]
        L.lsn_phy_mib_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Mib)]
        L.lsn_phy_mib_decode_llr.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Mib), C.c_void_p]
        L.lsn_phy_process_file.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(FileCfg), C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]
        L.lsn_phy_prepare_file.argtypes = [C.c_void_p, C.c_uint32]
        L.lsn_phy_set_prach_config.argtypes = [C.c_void_p, C.POINTER(PrachCfg)]
        L.lsn_phy_prach_detect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(PrachDet), C.c_uint32]
        L.lsn_phy_set_prach_sink.argtypes = [C.c_void_p, PRACH_SINK, C.c_void_p]
        L.lsn_phy_set_prach_sink.restype = None
        L.lsn_prach_tti_opportunity.argtypes = [C.c_uint32, C.c_uint32]
        L.lsn_phy_set_shortcut_discovery.argtypes = [C.c_void_p, C.c_int]
        L.lsn_phy_get_shortcut_discovery.argtypes = [C.c_void_p]
        L.lsn_phy_set_histogram_threshold.argtypes = [C.c_void_p, C.c_uint32]
        L.lsn_phy_print_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.lsn_phy_set_mcs_update_interval.argtypes = [C.c_void_p, C.c_uint32]
        L.lsn_phy_update_mcs_database.argtypes = [C.c_void_p]
        L.lsn_phy_nof_tracked_rnti.argtypes = [C.c_void_p]
        L.lsn_phy_nof_tracked_rnti.restype = C.c_uint32
        L.lsn_worker_buffers_offset.argtypes = [C.c_void_p]
        L.lsn_worker_buffers_offset.restype = C.POINTER(C.POINTER(C.c_float))
        L.lsn_cell_search.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, C.POINTER(CellSearchCfg), C.POINTER(CellSearch), C.c_void_p]
        _lib = L
    return _lib


def _check(rc, what):
    if rc != LSN_SUCCESS:
        raise RuntimeError("%s failed: %d%s" % (what, rc, " (no HIP device: this library has no CPU path)"
                                                if rc == LSN_ERROR_NO_DEVICE else ""))


def _check_n(rc, what):
    if rc < 0:
        _check(rc, what)
    return rc


class PcapWriter:
    """Mirror of LTESniffer_pcap_writer (PcapWriter.h:39-51) on the native writer; path=None -> in-memory capture."""

    def __init__(self, path=None, wall_clock=None):
        L = lib()
        self._h = L.lsn_pcap_open(path.encode()) if path else L.lsn_pcap_open_mem()
        if not self._h:
            raise RuntimeError("cannot open pcap %r" % path)
        if wall_clock is not None:
            L.lsn_pcap_set_wall_clock(self._h, int(wall_clock))

    def write(self, ctx, pdu):
        c = PduCtx(ctx["tti"], ctx["rnti"], ctx["direction"], ctx["rnti_type"], ctx.get("crc_ok", 1), 0, ctx.get("tb", 0), 0)
        _check(lib().lsn_pcap_write(self._h, C.byref(c), pdu, len(pdu)), "pcap_write")

    def bytes(self):
        n = C.c_size_t()
        p = lib().lsn_pcap_mem(self._h, C.byref(n))
        return C.string_at(p, n.value)

    def nof_records(self):
        return lib().lsn_pcap_nof_records(self._h)

    def digest(self):
        """(64-bit digest, bytes) of the record stream since open / reset, timestamps excluded"""
        d, n = C.c_uint64(), C.c_uint64()
        _check(lib().lsn_pcap_digest(self._h, C.byref(d), C.byref(n)), "pcap_digest")
        return int(d.value), int(n.value)

    def set_store(self, on):
        lib().lsn_pcap_set_store(self._h, int(bool(on)))

    def set_digest_blocks(self, subframes_per_block, origin_tti=0):
        """cut the digest every n subframes counted from origin_tti (0 = off); clears the blocks collected so far"""
        lib().lsn_pcap_set_digest_blocks(self._h, int(subframes_per_block), int(origin_tti) % 10240)

    def block_digests(self):
        """-> [(digest, nof_records)] per block of the stream since reset / set_digest_blocks"""
        n = lib().lsn_pcap_block_digests(self._h, None, None, 0)
        d, c = (C.c_uint64 * max(1, n))(), (C.c_uint32 * max(1, n))()
        n = min(n, lib().lsn_pcap_block_digests(self._h, d, c, n))
        return [(int(d[i]), int(c[i])) for i in range(n)]

    def reset(self):
        lib().lsn_pcap_reset(self._h)

    def close(self):
        if self._h:
            lib().lsn_pcap_close(self._h)
            self._h = None


class SubframeWorker:
    """Mirror of SubframeWorker (SubframeWorker.h:16-86): buffers to fill + prepare(); work() runs inside Phy."""

    def __init__(self, phy, handle):
        self._phy, self._h = phy, handle

    def getBuffers(self):
        """-> list of numpy complex64 views (one per rx antenna, 3 * SF_LEN samples) of the pinned IQ buffers"""
        L = lib()
        n = L.lsn_worker_buffer_len(self._h)
        bufs = L.lsn_worker_buffers(self._h)
        out = []
        for rx in range(self._phy.nof_rx_antennas):
            a = np.ctypeslib.as_array(bufs[rx], shape=(2 * n,))
            out.append(a.view(np.complex64))
        return out

    def prepare(self, sf_idx, sfn, updateMetaFormats, dl_sf_cfg=None):
        cfg = dl_sf_cfg if dl_sf_cfg is not None else DlSfCfg(sfn * 10 + sf_idx, 0, 0)
        _check(lib().lsn_worker_prepare(self._h, sf_idx, sfn, int(bool(updateMetaFormats)), C.byref(cfg)), "prepare")

    def getSfidx(self):
        return lib().lsn_worker_sf_idx(self._h)

    def getSfn(self):
        return lib().lsn_worker_sfn(self._h)


class Phy:
    """Mirror of Phy (Phy.h:22-66). Constructor arguments keep the reference's names; the pcap writer argument is
    replaced by a PDU sink callable ``sink(ctx_dict, pdu_bytes)`` receiving what pack_and_write would serialise."""

    def __init__(self, nof_rx_antennas=2, nof_workers=20, skipSecondaryMetaFormats=False, metaFormatSplitRatio=0.99,
                 histogramThreshold=5, sink=None, mcs_tracking_mode=1, harq_mode=0, device=0, max_batch=64,
                 max_turbo_iterations=12, default_rnti_intervals=True, pcapwriter=None, sniffer_mode=0, devices=None):
        """devices: list of HIP devices that share ONE capture (lsn_phy_create_multi); None: the single `device`"""
        self.nof_rx_antennas = nof_rx_antennas
        self._cfg = PhyCfg(nof_rx_antennas, nof_workers, max_batch, int(skipSecondaryMetaFormats), metaFormatSplitRatio,
                           histogramThreshold, mcs_tracking_mode, harq_mode, devices[0] if devices else device, max_turbo_iterations, sniffer_mode)
        self._h = C.c_void_p()
        if devices:
            arr = (C.c_int * len(devices))(*devices)
            _check(lib().lsn_phy_create_multi(C.byref(self._cfg), arr, len(devices), C.byref(self._h)), "lsn_phy_create_multi")
        else:
            _check(lib().lsn_phy_create(C.byref(self._cfg), C.byref(self._h)), "lsn_phy_create")
        self.pdus = []
        self._user_sink = sink
        self._cb = SINK_T(self._on_pdu)
        self.pcapwriter = pcapwriter
        if pcapwriter is not None:  # native consumer, like the reference's LTESniffer_pcap_writer* ctor argument
            _check(lib().lsn_phy_set_pcap_writer(self._h, pcapwriter._h), "set_pcap_writer")
        else:
            _check(lib().lsn_phy_set_pdu_sink(self._h, self._cb, None), "set_pdu_sink")
        if default_rnti_intervals:
            _check(lib().lsn_phy_setup_default_rnti_intervals(self._h), "rnti intervals")
        self.cell = None

    def _on_pdu(self, user, ctx, pdu, n):
        c = ctx.contents
        d = dict(tti=c.tti, rnti=c.rnti, direction=c.direction, rnti_type=c.rnti_type, crc_ok=c.crc_ok, tb=c.tb)
        data = C.string_at(pdu, n)
        if self._user_sink is not None:
            self._user_sink(d, data)
        else:
            self.pdus.append((d, data))

    def setCell(self, nof_prb, nof_ports, cell_id, phich_resources=0, cp=0):
        """cp: srsran_cell_t.cp - 0 normal, 1 extended cyclic prefix (downlink path)"""
        self.cell = Cell(nof_prb, nof_ports, cell_id, cp, 0, phich_resources, 0)
        rc = lib().lsn_phy_set_cell(self._h, C.byref(self.cell))
        return rc == LSN_SUCCESS

    def getAvail(self):
        h = lib().lsn_phy_get_avail(self._h, 1)
        return SubframeWorker(self, h) if h else None

    def getAvailImmediate(self):
        h = lib().lsn_phy_get_avail(self._h, 0)
        return SubframeWorker(self, h) if h else None

    def putPending(self, worker):
        _check(lib().lsn_phy_put_pending(self._h, worker._h), "putPending")

    def joinPending(self):
        _check(lib().lsn_phy_join_pending(self._h), "joinPending")

    # ---- the calls LTESniffer_Core makes on PhyCommon / RNTIManager / MCSTracking (LTESniffer_Core.cc:87,473-499,561,616-620) ----
    def setShortcutDiscovery(self, enable):
        _check(lib().lsn_phy_set_shortcut_discovery(self._h, int(bool(enable))), "setShortcutDiscovery")

    def getShortcutDiscovery(self):
        return bool(lib().lsn_phy_get_shortcut_discovery(self._h))

    def setHistogramThreshold(self, threshold):
        _check(lib().lsn_phy_set_histogram_threshold(self._h, int(threshold)), "setHistogramThreshold")

    def printStats(self):
        """PhyCommon::printStats: the two CSV lines, returned as text (written through a FILE* by the library)"""
        import tempfile
        libc = C.CDLL(None)
        libc.fopen.restype = C.c_void_p
        libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
        libc.fclose.argtypes = [C.c_void_p]
        with tempfile.NamedTemporaryFile() as t:
            f = libc.fopen(t.name.encode(), b"w")
            _check(lib().lsn_phy_print_stats(self._h, C.c_void_p(f)), "printStats")
            libc.fclose(f)
            return open(t.name).read()

    def setMcsUpdateInterval(self, seconds):
        _check(lib().lsn_phy_set_mcs_update_interval(self._h, int(seconds)), "setMcsUpdateInterval")

    def updateMcsDatabase(self):
        _check(lib().lsn_phy_update_mcs_database(self._h), "updateMcsDatabase")

    def nofTrackedRnti(self):
        return int(lib().lsn_phy_nof_tracked_rnti(self._h))

    # ---- offline / file-replay path ----
    def process_host(self, iq, start_tti, update_meta_period=0):
        """iq: complex64 [n_subframes, nof_rx, 15*N]"""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        _check(lib().lsn_phy_process_host(self._h, iq.ctypes.data, iq.shape[0], start_tti, update_meta_period), "process_host")

    def process_host_int(self, q, start_tti, update_meta_period=0, sample_scale=0.0):
        """q: int16 or int8 [n_subframes, nof_rx, 15*N, 2] (I, Q); one LSB = sample_scale (0: full scale +-1) - lsn_phy_process_host_int"""
        assert q.dtype in (np.int16, np.int8) and q.ndim == 4 and q.shape[3] == 2
        q = np.ascontiguousarray(q)
        _check(lib().lsn_phy_process_host_int(self._h, q.ctypes.data, FILE_SC16 if q.dtype == np.int16 else FILE_SC8, float(sample_scale), q.shape[0], start_tti,
                                              update_meta_period), "process_host_int")

    def process_device(self, dev_ptr, n_subframes, start_tti, update_meta_period=0, stream=None):
        _check(lib().lsn_phy_process_device(self._h, C.c_void_p(dev_ptr), n_subframes, start_tti, update_meta_period,
                                            C.c_void_p(stream or 0)), "process_device")

    def submit_device(self, d_ptr, n_subframes, start_tti, update_meta_period=0, stream=None):
        """pipelined process_device: returns once the block is searched and queued; call wait() before reusing the buffer / reading results"""
        _check(lib().lsn_phy_submit_device(self._h, C.c_void_p(d_ptr), n_subframes, start_tti, update_meta_period, C.c_void_p(stream or 0)), "submit_device")

    def wait(self):
        _check(lib().lsn_phy_wait(self._h), "wait")

    def mib_decode(self, iq, with_llr=False):
        """srsran_ue_mib_decode on ONE subframe iq[nof_rx, 15*N] -> dict (found, sfn, sfn_offset, nof_prb, nof_ports, ...) [, raw PBCH soft bits]"""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        m = Mib()
        llr = np.zeros(480, dtype=np.float32)
        r = lib().lsn_phy_mib_decode_llr(self._h, iq.ctypes.data, 0, C.byref(m), llr.ctypes.data)
        if r < 0:
            _check(r, "mib_decode")
        d = {n: int(getattr(m, n)) for n, _ in Mib._fields_}
        return (d, llr) if with_llr else d

    def prepare_file(self):
        """reserve the file source's block buffers ahead of the first replay (lsn_phy_prepare_file)"""
        _check(lib().lsn_phy_prepare_file(self._h, self.nof_rx_antennas), "prepare_file")

    def process_file(self, path, start_tti=0, offset_time=0, offset_freq=0.0, max_subframes=0, update_meta_period=0, sample_format=FILE_CF32, sample_scale=0.0):
        """file mode of the reference (-i file -O offset_time -o offset_freq): replay a cf32 capture (antennas interleaved per sample);
        sample_format FILE_SC16 / FILE_SC8: integer I/Q pairs, one LSB = sample_scale (0: full scale +-1); returns the number of subframes processed"""
        fc = FileCfg(self.nof_rx_antennas, int(offset_time), float(offset_freq), int(sample_format), float(sample_scale))
        done = C.c_uint64(0)
        _check(lib().lsn_phy_process_file(self._h, os.fsencode(path), C.byref(fc), start_tti if start_tti == TTI_FROM_MIB else start_tti % 10240, max_subframes, update_meta_period, C.byref(done)),
               "process_file")
        return int(done.value)

    # ---- uplink ----
    def setUlConfig(self, cyclic_shift, delta_ss, hopping_offset=0, group_hopping=0, sequence_hopping=0):
        u = UlCfg(cyclic_shift, delta_ss, hopping_offset, int(group_hopping), int(sequence_hopping))
        return lib().lsn_phy_set_ul_config(self._h, C.byref(u)) == LSN_SUCCESS

    def setApiMode(self, api_mode, api_pcapwriter=None):
        """-a of the reference: identities found in decoded downlink blocks are collected in self.api_events as
        (tti, rnti, id_type, msg_type, value); blocks that carried one also go to api_pcapwriter"""
        self.api_events = []
        self._api_pcap = api_pcapwriter

        def _cb(user, ev):
            e = ev.contents
            self.api_events.append((int(e.tti), int(e.rnti), int(e.id_type), int(e.msg_type), e.value.decode()))
        self._api_cb = API_SINK(_cb)
        return lib().lsn_phy_set_api_mode(self._h, int(api_mode), self._api_cb, None, api_pcapwriter._h if api_pcapwriter else None) == LSN_SUCCESS

    def trackedUlModulation(self, rnti):
        return int(lib().lsn_phy_tracked_ul_modulation(self._h, rnti))

    def getUlConfig(self):
        """None until an uplink configuration is in use, else dict(cyclic_shift, delta_ss, hopping_offset, from_sib2, sib2 = dict or None)"""
        u, s, f = UlCfg(), Sib2(), C.c_uint32(0)
        if lib().lsn_phy_get_ul_config(self._h, C.byref(u), C.byref(s), C.byref(f)) != 1:
            return None
        return dict(cyclic_shift=u.cyclic_shift, delta_ss=u.delta_ss, hopping_offset=u.hopping_offset, group_hopping=u.group_hopping_enabled,
                    sequence_hopping=u.sequence_hopping_enabled, from_sib2=bool(f.value),
                    sib2=s.as_dict() if f.value else None)

    def pusch_decode(self, ul_iq, start_tti, grants):
        """ul_iq: complex64 [n_subframes, 15*N]; grants: list of dict(sf, rnti, n_dmrs, n_prb, L_prb, mod, tbs, rv)
        -> list of dict(crc_ok, iterations, snr_db, payload)"""
        ul_iq = np.ascontiguousarray(ul_iq, dtype=np.complex64)
        n = len(grants)
        arr = (PuschGrant * max(1, n))(*[PuschGrant(g["sf"], g["rnti"], g.get("n_dmrs", 0), g["n_prb"], g["L_prb"], g["mod"], g["tbs"], g.get("rv", 0),
                                                    g.get("nof_ack", 0), g.get("cqi_bits", 0), g.get("ri_bits", 0), g.get("hop", 0), g.get("n_prb2", 0),
                                                    g.get("i_ack_p1", 0), g.get("i_cqi_p1", 0), g.get("i_ri_p1", 0))
                                         for g in grants])
        res = (PuschResult * max(1, n))()
        cap = sum(g["tbs"] // 8 for g in grants) + 64
        pay = np.zeros(cap, dtype=np.uint8)
        _check(lib().lsn_phy_pusch_decode(self._h, ul_iq.ctypes.data, 0, ul_iq.shape[0], start_tti, arr, n, res, pay.ctypes.data, cap), "pusch_decode")
        return [dict(crc_ok=int(res[i].crc_ok), iterations=int(res[i].iterations), snr_db=float(res[i].snr_db),
                     payload=bytes(pay[res[i].payload_off:res[i].payload_off + grants[i]["tbs"] // 8]) if res[i].crc_ok else b"")
                for i in range(n)]

    def tap_ul_grid(self, sf):
        nre = 12 * self.cell.nof_prb
        buf = np.zeros(14 * nre, dtype=np.complex64)
        nb = lib().lsn_phy_tap_ul(self._h, 0, sf, buf.ctypes.data, buf.nbytes)
        if nb < 0:
            raise RuntimeError("tap_ul failed: %d" % nb)
        return buf.reshape(14, nre)

    def tap_ul_llr(self, grant_index, count):
        buf = np.zeros(count, dtype=np.int16)
        nb = lib().lsn_phy_tap_ul(self._h, 1, grant_index, buf.ctypes.data, buf.nbytes)
        if nb < 0:
            raise RuntimeError("tap_ul failed: %d" % nb)
        return buf[: nb // 2]

    # ---- PRACH ----
    def setPrachConfig(self, config_idx, root_seq_idx, zero_corr_zone, freq_offset, hs_flag=0, detect_factor=0.0, zc_roots=None):
        """PUSCH_Decoder::set_rach_config; zc_roots: 838 uint16 (36.211 Table 5.7.2-4) or None"""
        zc = np.ascontiguousarray(zc_roots, dtype=np.uint16) if zc_roots is not None else None
        p = PrachCfg(config_idx, root_seq_idx, zero_corr_zone, freq_offset, hs_flag, detect_factor,
                     zc.ctypes.data_as(C.POINTER(C.c_uint16)) if zc is not None else None)
        nwin = 839 // PRACH_NCS[zero_corr_zone] if 0 < zero_corr_zone < 16 else 1
        self._prach_nroots = (64 + nwin - 1) // nwin
        return lib().lsn_phy_set_prach_config(self._h, C.byref(p)) == LSN_SUCCESS

    def prach_detect(self, ul_iq, start_tti, cap=256):
        """PUSCH_Decoder::work_prach over ul_iq complex64 [n_subframes, 15*N] -> list of dict(sf, preamble, offset, offset_sec, p2avg)"""
        ul_iq = np.ascontiguousarray(ul_iq, dtype=np.complex64)
        det = (PrachDet * cap)()
        n = _check_n(lib().lsn_phy_prach_detect(self._h, ul_iq.ctypes.data, 0, ul_iq.shape[0], start_tti, det, cap), "prach_detect")
        return [dict(sf=int(d.sf), preamble=int(d.preamble), offset=int(d.offset), offset_sec=float(d.offset_sec), p2avg=float(d.p2avg)) for d in det[:n]]

    def set_prach_sink(self, fn):
        """fn(tti, [dict(...)]) for every PRACH occasion with detections while UL_MODE batches are processed"""
        def tramp(_user, tti, det, n):
            fn(int(tti), [dict(sf=int(det[i].sf), preamble=int(det[i].preamble), offset=int(det[i].offset), offset_sec=float(det[i].offset_sec),
                               p2avg=float(det[i].p2avg)) for i in range(n)])
        self._prach_cb = PRACH_SINK(tramp) if fn else PRACH_SINK()
        lib().lsn_phy_set_prach_sink(self._h, self._prach_cb, None)

    def tap_prach_corr(self, occasion):
        buf = np.zeros(self._prach_nroots * 839, dtype=np.float32)
        nb = lib().lsn_phy_tap_ul(self._h, 2, occasion, buf.ctypes.data, buf.nbytes)
        if nb < 0:
            raise RuntimeError("tap_ul failed: %d" % nb)
        return buf.reshape(self._prach_nroots, 839)

    # ---- taps / stats ----
    def tap(self, what, sf, dtype, count):
        buf = np.zeros(count, dtype=dtype)
        n = lib().lsn_phy_tap(self._h, what, sf, buf.ctypes.data, buf.nbytes)
        if n < 0:
            raise RuntimeError("tap %d failed: %d" % (what, n))
        return buf[: n // buf.itemsize]

    def set_stage_c_taps(self, on=True):
        """keep the soft bits / de-rate-matched words / per-code-block verdicts of every decode job of the batches processed from now on"""
        _check(lib().lsn_phy_set_stage_c_taps(self._h, 1 if on else 0), "set_stage_c_taps")

    def stage_c_jobs(self):
        """decode jobs of the LAST chunk: list of dict(sf, tti, rnti, nof_re, qm, tbs, crc, p_a_db, llr=[cw0, cw1] int16, cbs=[dict(tb, K, F, E, rv, ok, iters, skipped,
        d3=int16[3, K + 4] (the de-rate-matched streams unpacked from the kernel's transposed 10-bit words))])"""
        jt = np.dtype([("sf", "<u4"), ("tti", "<u4"), ("rnti", "<u4"), ("nof_re", "<u4"), ("qm", "<u4", 2), ("llr_len", "<u4", 2), ("tbs", "<u4", 2), ("crc", "<u4", 2),
                       ("ncb", "<u4"), ("have", "<u4"), ("done", "<u4"), ("p_a_db", "<f4")])
        ct = np.dtype([("tb", "<u4"), ("K", "<u4"), ("F", "<u4"), ("E", "<u4"), ("rv", "<u4"), ("ok", "<u4"), ("iters", "<u4"), ("skipped", "<u4")])
        hdr = self.tap(TAP_PDSCH_JOBS, 0, np.uint8, 1 << 22).view(jt)
        out = []
        for j, h in enumerate(hdr):
            d = dict(job=j, sf=int(h["sf"]), tti=int(h["tti"]), rnti=int(h["rnti"]), nof_re=int(h["nof_re"]), qm=[int(x) for x in h["qm"]], tbs=[int(x) for x in h["tbs"]],
                     crc=[int(x) for x in h["crc"]], p_a_db=float(h["p_a_db"]), have=bool(h["have"]), done=bool(h["done"]), llr=[None, None], cbs=[])
            if h["have"]:
                n0, n1 = int(h["llr_len"][0]), int(h["llr_len"][1])
                llr = self.tap(TAP_PDSCH_LLR16, j, np.int16, n0 + n1 + 8)
                d["llr"] = [llr[:n0].copy(), llr[n0:n0 + n1].copy()]
                cbs = self.tap(TAP_CB_RESULT, j, np.uint8, 64 * ct.itemsize).view(ct)
                words = self.tap(TAP_RM_WORDS, j, np.uint32, int(sum(int(c["K"]) + 12 for c in cbs)) + 8)
                o = 0
                for c in cbs:
                    K = int(c["K"])
                    w = words[o:o + K + 12]
                    o += K + 12
                    P = turbo_nwin(K)
                    W = K // P
                    x = np.arange(K)
                    ww = w[(x % W) * P + x // W]
                    d3 = np.zeros((3, K + 4), dtype=np.int16)
                    for s_ in range(3):
                        f = ((ww >> (10 * s_)) & 0x3FF).astype(np.int32)
                        d3[s_, :K] = np.where(f >= 512, f - 1024, f)
                        d3[s_, K:] = w[K + 4 * s_:K + 4 * s_ + 4].view(np.int32)
                    d["cbs"].append(dict(tb=int(c["tb"]), K=K, F=int(c["F"]), E=int(c["E"]), rv=int(c["rv"]), ok=int(c["ok"]), iters=int(c["iters"]), skipped=int(c["skipped"]), d3=d3))
            out.append(d)
        return out

    def perf(self):
        p = Perf()
        _check(lib().lsn_phy_get_perf(self._h, C.byref(p)), "get_perf")
        return p

    def getStats(self):
        s = BlindStats()
        _check(lib().lsn_phy_get_stats(self._h, C.byref(s)), "get_stats")
        return s

    def est_cfo(self):
        return lib().lsn_phy_get_est_cfo(self._h)

    CFO_OFF, CFO_FIXED, CFO_TRACK = 0, 1, 2

    def setCfoCorrection(self, mode, cfo_hz=0.0, alpha=0.25):
        """CFO correction inside the OFDM kernel - what srsran_ue_sync's tracking does ahead of the reference's workers (LTESniffer_Core.cc:312-316,344).
        CFO_FIXED removes cfo_hz; CFO_TRACK starts there and follows the CRS estimate chunk by chunk (include/ltesniffer_amd.h)"""
        _check(lib().lsn_phy_set_cfo_correction(self._h, int(mode), float(cfo_hz), float(alpha)), "setCfoCorrection")

    PRUNE_OFF, PRUNE_ON, PRUNE_TEST = 0, 1, 2

    def setCandidatePruning(self, mode):
        """which slots of the candidate table the blind decoder computes ahead of the search (lsn_phy_set_candidate_pruning): PRUNE_OFF = all, PRUNE_ON (default) =
        not those under a location the search is predicted to accept (decoded on demand if it comes there after all), PRUNE_TEST = the prediction claims everything"""
        _check(lib().lsn_phy_set_candidate_pruning(self._h, int(mode)), "setCandidatePruning")

    def getCfoCorrection(self):
        return lib().lsn_phy_get_cfo_correction(self._h)

    def nof_active_rnti(self):
        return lib().lsn_phy_nof_active_rnti(self._h)

    def ue_config(self, rnti):
        """MCSTracking::get_ue_config_rnti: what RRCConnectionSetup messages taught about this RNTI (or the default)"""
        c = UeConfig()
        _check(lib().lsn_phy_get_ue_config(self._h, rnti, C.byref(c)), "get_ue_config")
        return c

    def close(self):
        if self._h:
            lib().lsn_phy_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cell_search(iq, nof_prb, nof_periods=2, force_n_id_2=-1, threshold=20.0, device=0, with_corr=False):
    """rf_search_and_decode_mib of the reference (LTESniffer_Core.cc:195-204) on a block of samples of one antenna:
    -> (rc, CellSearch[, corr[3, 75 N]]); rc 1 found / 0 not found; iq: numpy complex64 (host) or a torch cuda tensor"""
    import numpy as np
    N = {6: 128, 15: 256, 25: 512, 50: 1024, 75: 1536, 100: 2048}.get(nof_prb, 128)
    cfg = CellSearchCfg(nof_periods, force_n_id_2, threshold)
    out = CellSearch()
    corr = np.zeros((3, 75 * N), dtype=np.float32) if with_corr else None
    if hasattr(iq, "data_ptr"):
        ptr, n, on_dev = iq.data_ptr(), iq.numel(), 1
    else:
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        ptr, n, on_dev = iq.ctypes.data, iq.size, 0
    rc = _check_n(lib().lsn_cell_search(device, ptr, on_dev, n, nof_prb, C.byref(cfg), C.byref(out), corr.ctypes.data if with_corr else None), "lsn_cell_search")
    return (rc, out, corr) if with_corr else (rc, out)


def mac_lte_record(ctx, pdu):
    """MAC-LTE (DLT 147) framing of one PDU exactly as LTESniffer_pcap_writer::pack_and_write emits it
    (/root/reference/src/src/PcapWriter.cc:93-118; byte layout SURVEY.md appendix B). -> context header + PDU bytes."""
    tti = ctx["tti"]
    fs = ((tti // 10) << 4) | (tti % 10)
    hdr = bytes([1, ctx["direction"], ctx["rnti_type"], 2, ctx["rnti"] >> 8, ctx["rnti"] & 255, 3, 0, 0, 4, (fs >> 8) & 255,
                 fs & 255, 7, ctx["crc_ok"], 10, 0, 15, 0, 1])
    return hdr + pdu


def write_pcap(path, records, ts=(0, 0)):
    """records: iterable of (ctx, pdu). Writes the same global header as the reference (network 147)."""
    with open(path, "wb") as f:
        f.write(struct.pack("<IHHiIII", 0xA1B2C3D4, 2, 4, 0, 0, 65535, 147))
        for ctx, pdu in records:
            body = mac_lte_record(ctx, pdu)
            f.write(struct.pack("<IIII", ts[0], ts[1], len(body), len(body)) + body)
