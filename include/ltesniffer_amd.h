/*
 * ltesniffer_amd.h - C ABI of the MI355X-native LTESniffer per-subframe worker (the drop-in boundary).
 *
 * Everything behind LTESniffer's Phy / SubframeWorker pair is replaced by this library:
 *   class Phy            /root/reference/src/include/Phy.h:22-66
 *   class SubframeWorker /root/reference/src/include/SubframeWorker.h:16-86
 * The caller (LTESniffer_Core::run, /root/reference/src/src/LTESniffer_Core.cc:292-299,365,434-451,547)
 * keeps filling antenna buffers and calling prepare/putPending; decoded MAC PDUs come back through a sink
 * callback that carries exactly the fields LTESniffer_pcap_writer::pack_and_write serialises
 * (/root/reference/src/src/PcapWriter.cc:93-118), in (tti, DCI acceptance order, TB) order.
 *
 * Plain C: pointers + sizes, int return codes (0 ok, <0 error: same convention as SRSRAN_SUCCESS / SRSRAN_ERROR /
 * SRSRAN_ERROR_INVALID_INPUTS), no exceptions cross the boundary.  The library REQUIRES a HIP device; there is no
 * CPU fallback: lsn_phy_create fails with LSN_ERROR_NO_DEVICE when none is visible.
 */
#ifndef LTESNIFFER_AMD_H
#define LTESNIFFER_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LSN_SUCCESS 0
#define LSN_ERROR (-1)
#define LSN_ERROR_INVALID_INPUTS (-2)
#define LSN_ERROR_NO_DEVICE (-3)

typedef struct lsn_phy lsn_phy_t;       /* replaces class Phy (Phy.h:22) */
typedef struct lsn_worker lsn_worker_t; /* replaces class SubframeWorker (SubframeWorker.h:16) */

/* srsran_cell_t subset that crosses the boundary (SubframeWorker::setCell, SubframeWorker.cc:100-107) */
typedef struct {
  uint32_t nof_prb;         /* 6, 15, 25, 50, 75, 100 */
  uint32_t nof_ports;       /* 1, 2 or 4 CRS ports (4: transmit diversity on every channel; spatial-multiplexing grants are found and not decoded, as with the reference's srsRAN) */
  uint32_t id;              /* physical cell id */
  uint32_t cp;              /* 0 = normal, 1 = extended cyclic prefix (6 symbols per slot; DL mode and UL_MODE; lsn_cell_search reports it) */
  uint32_t phich_length;    /* 0 = normal */
  uint32_t phich_resources; /* 0: 1/6, 1: 1/2, 2: 1, 3: 2 (LTESniffer_Core.cc:211-212 forces 1/6) */
  uint32_t frame_type;      /* 0 = FDD */
} lsn_cell_t;

/* srsran_dl_sf_cfg_t subset passed by SubframeWorker::prepare (SubframeWorker.cc:109-116, LTESniffer_Core.cc:429-434) */
typedef struct {
  uint32_t tti;
  uint32_t cfi;     /* ignored on input: decoded from PCFICH per subframe */
  uint32_t sf_type; /* 0 = SRSRAN_SF_NORM */
} lsn_dl_sf_cfg_t;

/* Constructor arguments of Phy::Phy (Phy.h:24-36) that influence the hot path */
typedef struct {
  uint32_t nof_rx_antennas;
  uint32_t nof_workers;             /* worker (buffer) pool size; Phy.h:19 uses 20 */
  uint32_t max_batch;               /* subframes processed per GPU batch (0 = default 64) */
  int skip_secondary_meta_formats;  /* Phy ctor skipSecondaryMetaFormats */
  double meta_format_split_ratio;   /* Settings.h:55, default 0.99 */
  uint32_t histogram_threshold;     /* Settings.h:57, default 5 */
  int mcs_tracking_mode;            /* ArgManager.cc:52, default 1 */
  int harq_mode;                    /* 0 (ArgManager.cc:50: the reference's only reachable value) or 1: DL HARQ soft combining (HARQ.cc:71-190,
                                       DL_Sniffer_PDSCH.cc:943-1020) for known-table C-RNTI grants; DL mode.  With lsn_phy_create_multi the soft-buffer pool lives on the
                                       first device and the other engines need peer access to it (the call fails otherwise) */
  int device;                       /* HIP device ordinal */
  int max_turbo_iterations;         /* SubframeWorker.cc:365, default 12 (0 = default) */
  int sniffer_mode;                 /* 0 = DL_MODE, 1 = UL_MODE (SubframeWorker.cc:166-199): antenna 0 = downlink, antenna 1 = uplink,
                                       nof_rx_antennas must be 2; the uplink configuration comes from lsn_phy_set_ul_config (+ lsn_phy_set_prach_config)
                                       or, when those are not called, from the first SIB2 decoded (PDSCH_Decoder::decode_SIB) */
} lsn_phy_cfg_t;

/* What LTESniffer_pcap_writer::pack_and_write receives (PcapWriter.cc:93-111) */
typedef struct {
  uint32_t tti;       /* sfn*10 + sf */
  uint16_t rnti;      /* SI/P constants already substituted (PcapWriter.cc:162-170) */
  uint8_t direction;  /* 0 UL, 1 DL */
  uint8_t rnti_type;  /* 0 NO, 1 P, 2 RA, 3 C, 4 SI */
  uint8_t crc_ok;
  uint8_t is_retx;
  uint8_t tb;         /* transport block index within the DCI */
  uint8_t reserved;
} lsn_pdu_ctx_t;
typedef void (*lsn_pdu_sink_t)(void* user, const lsn_pdu_ctx_t* ctx, const uint8_t* pdu, uint32_t len);

/* DCIBlindSearchStats (PhyCommon.h:12-26) */
typedef struct {
  uint32_t nof_locations, nof_decoded_locations, nof_cce, nof_missed_cce, nof_subframes, nof_subframe_collisions_dw,
      nof_subframe_collisions_up;
} lsn_blind_stats_t;

/* ---- Phy ---- */
int lsn_phy_create(const lsn_phy_cfg_t* cfg, lsn_phy_t** out);          /* Phy::Phy, Phy.cc:5 */
void lsn_phy_destroy(lsn_phy_t* phy);                                     /* Phy::~Phy */
/* ONE capture spread over several GPUs of the node: a Phy whose chunks of max_batch subframes go round-robin to one
 * engine per listed device.  Stage A, the exhaustive candidate decode and the PDSCH decodes of a chunk run on its device; the sequential
 * host state (FALCON search / RNTI manager, MCS tracking, record order) is shared and taken in turns, so the record stream is the one a
 * single device produces.  IQ blocks that live on another device travel by peer copies (xGMI).  devices[0] is the primary device (MIB,
 * uplink, taps); listing a device twice is allowed (two engines on one GPU).  Supported entry points in this mode: lsn_phy_process_device,
 * lsn_phy_submit_device, lsn_phy_wait and the getters; DL_MODE and UL_MODE (the ULSchedule / uplink tracking databases and the uplink
 * configuration are part of the shared state; a SIB2 learnt by one engine reaches the device tables of the others at their next commit turn).
 * Several CELLS are several Phys - one process per GPU, no exchange. */
int lsn_phy_create_multi(const lsn_phy_cfg_t* cfg, const int* devices, uint32_t n_devices, lsn_phy_t** out);
uint32_t lsn_phy_nof_devices(lsn_phy_t* phy);
int lsn_phy_set_cell(lsn_phy_t* phy, const lsn_cell_t* cell);             /* Phy::setCell, Phy.cc:111 */
lsn_worker_t* lsn_phy_get_avail(lsn_phy_t* phy, int blocking);            /* Phy::getAvail / getAvailImmediate, Phy.cc:79-89 */
int lsn_phy_put_pending(lsn_phy_t* phy, lsn_worker_t* w);                 /* Phy::putPending, Phy.cc:95 */
int lsn_phy_join_pending(lsn_phy_t* phy);                                 /* Phy::joinPending, Phy.cc:100 */
int lsn_phy_set_pdu_sink(lsn_phy_t* phy, lsn_pdu_sink_t cb, void* user);  /* stands in for the pcapwriter ctor argument */
int lsn_phy_get_stats(lsn_phy_t* phy, lsn_blind_stats_t* out);            /* PhyCommon::getStats, PhyCommon.cc:60 */
float lsn_phy_get_est_cfo(lsn_phy_t* phy);                                /* SubframeWorker.cc:203 est_cfo */
/* CFO correction inside the OFDM kernel (an NCO on the samples as the FFT loads them).  Replaces, for this path, what srsran_ue_sync does to the samples before
 * they reach SubframeWorker::work: ue_sync.cfo_correct_enable_track = !args.disable_cfo (LTESniffer_Core.cc:344), started from the cell search's offset
 * (ue_sync.cfo_current_value = search_cell_cfo / 15000, :312-316) and kept on the carrier by srsran_ue_sync_set_cfo_ref with the CRS estimate.
 *   mode 0: off (the reference's file replay without -o)          mode 1: remove the fixed offset cfo_hz          mode 2: track - start from cfo_hz, then
 *   each chunk of max_batch subframes is corrected by c += alpha * (m - c), m = the absolute offset (correction + mean CRS residual) measured on the chunk
 *   LSN_NSTREAM_A (4) chunks earlier; 0 < alpha <= 1.  est_cfo keeps reporting the CRS estimate of the corrected samples (the residual), as in the reference.
 * Takes effect with the next chunk that enters the pipeline; call it between process calls.  mode 2 is refused on a handle of lsn_phy_create_multi. */
int lsn_phy_set_cfo_correction(lsn_phy_t* phy, int mode, float cfo_hz, float alpha);
float lsn_phy_get_cfo_correction(lsn_phy_t* phy);                         /* the offset removed from the chunk launched last (srsran_ue_sync_get_cfo) */
/* Blind DCI decoding (srsran_pdcch_decode_msg_limit_avg_llr_power per location and format, falcon_pdcch.c:110-170; DCISearch.cc:102-447): which (location, size)
 * slots of a subframe's candidate table the GPU decodes ahead of the sequential search.
 *   mode 0: all of them (rounds 1-5)
 *   mode 1 (default): the aggregation levels in four launches, 8 -> 1 CCEs; a slot under a location that holds a candidate the search is predicted to accept (its own
 *           stateless tests + the RNTI evergreen, or active in the newest published snapshot of the RNTI manager and not forbidden) is left out, and decoded
 *           on demand if the search comes there after all (lsn_perf_t.nof_candidate_misses) - the table the search sees is the exhaustive one
 *   mode 2: test - every RNTI counts as active, so that the on-demand path carries the search
 * Takes effect with the next chunk that enters the pipeline. */
int lsn_phy_set_candidate_pruning(lsn_phy_t* phy, int mode);
int lsn_phy_add_evergreen(lsn_phy_t* phy, uint16_t rnti_start, uint16_t rnti_end, uint32_t format_idx); /* RNTIManager::addEvergreen */
int lsn_phy_add_forbidden(lsn_phy_t* phy, uint16_t rnti_start, uint16_t rnti_end, uint32_t format_idx); /* RNTIManager::addForbidden */
int lsn_phy_setup_default_rnti_intervals(lsn_phy_t* phy);                 /* LTESniffer_Core.cc:398-417 in one call */
uint32_t lsn_phy_nof_active_rnti(lsn_phy_t* phy);
/* Phy::getMetaFormats (Phy.h:45, Phy.cc:152; DCIMetaFormats, MetaFormats.h): the primary / secondary split of the nine DCI formats the search is working with,
 * as indices into falcon_ue_all_formats (DCISearch.cc:84-95; = srsran_dci_format_t values 0..8).  Each list has room for 9.  Between process calls / after wait. */
int lsn_phy_get_meta_formats(lsn_phy_t* phy, uint32_t* primary, uint32_t* nof_primary, uint32_t* secondary, uint32_t* nof_secondary);
/* Phy::getWorkers (Phy.h:46, Phy.cc:112): the pool's workers by index, whoever holds them at the moment */
uint32_t lsn_phy_nof_workers(lsn_phy_t* phy);
lsn_worker_t* lsn_phy_worker(lsn_phy_t* phy, uint32_t index);
/* UE-specific configuration learned from RRCConnectionSetup messages on the downlink (MCSTracking::get_ue_config_rnti,
 * MCSTracking.cc:1482-1516; filled by PDSCH_Decoder::decode_rrc_connection_setup, DL_Sniffer_PDSCH.cc:129-181): the entry of the
 * RNTI, or the default (the first connection setup seen; before that p_a 0 dB, offsets 10 / 8 / 11, higher-layer sub-band CQI).
 * p_a_db is the PDSCH power offset every later decode of that RNTI runs with (DL_Sniffer_PDSCH.cc:926-927). */
typedef struct {
  uint32_t has_ue_config;                           /* 1: from this UE's own connection setup */
  float p_a_db;
  uint32_t i_offset_ack, i_offset_cqi, i_offset_ri; /* betaOffset-ACK / CQI / RI-Index */
  uint32_t cqi_type;                                /* 0 wideband, 1 UE-selected sub-band, 2 higher-layer sub-band (srsran_cqi_type_t) */
} lsn_ue_config_t;
int lsn_phy_get_ue_config(lsn_phy_t* phy, uint16_t rnti, lsn_ue_config_t* out);
/* PhyCommon::setShortcutDiscovery (PhyCommon.cc:69-71; LTESniffer_Core.cc:87,616): the parent/child shortcut of the recursive blind
 * search (DCISearch.cc:200-211).  Default on, like ArgManager.cc:47. */
int lsn_phy_set_shortcut_discovery(lsn_phy_t* phy, int enable);
int lsn_phy_get_shortcut_discovery(lsn_phy_t* phy);
/* RNTIManager::setHistogramThreshold (RNTIManager.cc:442-444; LTESniffer_Core.cc:620) */
int lsn_phy_set_histogram_threshold(lsn_phy_t* phy, uint32_t threshold);
/* PhyCommon::printStats -> DCIBlindSearchStats::print (PhyCommon.cc:65-67,102-114; LTESniffer_Core.cc:561): the two CSV lines of the
 * reference's stats file; `file` is a FILE* (NULL: stdout).  The blind-search time column is the search thread's time. */
int lsn_phy_print_stats(lsn_phy_t* phy, void* file);
/* MCSTracking database ageing (MCSTracking::update_database_dl, MCSTracking.cc:850-927, driven every get_interval() x 1000 subframes
 * by LTESniffer_Core.cc:473-499).  The library ages its database by itself on the SUBFRAME count of the stream it commits (one subframe
 * = 1 ms; the reference uses clock(), i.e. replay speed): interval in seconds, 0 = never, default 5 (MCSTracking.h:162).
 * lsn_phy_update_mcs_database runs one update now (between process calls), for callers that keep the reference's own timer. */
int lsn_phy_set_mcs_update_interval(lsn_phy_t* phy, uint32_t seconds);
int lsn_phy_update_mcs_database(lsn_phy_t* phy);
uint32_t lsn_phy_nof_tracked_rnti(lsn_phy_t* phy);   /* MCSTracking::nof_RNTI_member_dl; in UL_MODE nof_RNTI_member_ul (the uplink database is aged the
                                                         same way: update_database_ul, MCSTracking.cc:86-176, fed by update_statistic_ul :729-754) */
int lsn_phy_tracked_ul_modulation(lsn_phy_t* phy, uint16_t rnti); /* UL_MODE: 0 no entry, 1 unknown, 2 / 3 / 4 = 16 / 64 / 256QAM maximum (find_tracking_info_RNTI_ul without the time stamp) */

/* ---- SubframeWorker ---- */
float** lsn_worker_buffers(lsn_worker_t* w);         /* SubframeWorker::getBuffers: [antenna] -> interleaved cf32, pinned host */
float** lsn_worker_buffers_offset(lsn_worker_t* w);  /* SubframeWorker::getBuffers_offset (SubframeWorker.h:37): the two 3 * SF_LEN scratch buffers of SubframeBuffer.cc:28 */
uint32_t lsn_worker_buffer_len(lsn_worker_t* w);     /* complex samples per antenna buffer (3 * SF_LEN, SubframeBuffer.cc:25) */
int lsn_worker_prepare(lsn_worker_t* w, uint32_t sf_idx, uint32_t sfn, int update_meta_formats, const lsn_dl_sf_cfg_t* sf); /* SubframeWorker::prepare */
uint32_t lsn_worker_sf_idx(lsn_worker_t* w);
uint32_t lsn_worker_sfn(lsn_worker_t* w);

/* ---- MAC-LTE pcap writer: replaces LTESniffer_pcap_writer (PcapWriter.h:39-51, PcapWriter.cc:75-118) ----
 * Record layout = LTE_PCAP_MAC_WritePDU [srsRAN], pinned by the captures under /root/reference/pcap_file_example/ (DLT 147). */
typedef struct lsn_pcap lsn_pcap_t;
lsn_pcap_t* lsn_pcap_open(const char* path);                 /* LTESniffer_pcap_writer::open, PcapWriter.cc:75 */
lsn_pcap_t* lsn_pcap_open_mem(void);                         /* in-memory capture with zero timestamps (tests, bench) */
void lsn_pcap_set_wall_clock(lsn_pcap_t* p, int on);         /* record timestamps: gettimeofday (default for files) or 0 */
int lsn_pcap_write(lsn_pcap_t* p, const lsn_pdu_ctx_t* ctx, const uint8_t* pdu, uint32_t len); /* pack_and_write, PcapWriter.cc:93 */
void lsn_pcap_sink(void* user /* lsn_pcap_t* */, const lsn_pdu_ctx_t* ctx, const uint8_t* pdu, uint32_t len); /* an lsn_pdu_sink_t */
const uint8_t* lsn_pcap_mem(lsn_pcap_t* p, size_t* len);
uint32_t lsn_pcap_nof_records(lsn_pcap_t* p);
void lsn_pcap_reset(lsn_pcap_t* p);
/* order-sensitive 64-bit digest + byte count of all records since open / reset, timestamps excluded (two writers fed the same record
 * sequence agree); lsn_pcap_set_store(p, 0) keeps only the count and the digest (long streams that need not be kept) */
int lsn_pcap_digest(lsn_pcap_t* p, uint64_t* digest, uint64_t* nbytes);
void lsn_pcap_set_store(lsn_pcap_t* p, int on);
/* Piecewise digests: the record stream is cut every `subframes_per_block` subframes counted from `origin_tti` (the TTI of the records
 * is unwrapped modulo 10240 in arrival order); each block has its own digest chain and record count, so a long replay can be checked
 * block by block against a reference produced elsewhere.  0 switches the blocks off; lsn_pcap_reset clears them. */
void lsn_pcap_set_digest_blocks(lsn_pcap_t* p, uint32_t subframes_per_block, uint32_t origin_tti);
uint32_t lsn_pcap_block_digests(lsn_pcap_t* p, uint64_t* digests, uint32_t* nof_records, uint32_t cap); /* -> number of blocks */
void lsn_pcap_close(lsn_pcap_t* p);                          /* LTESniffer_pcap_writer::close */
int lsn_phy_set_pcap_writer(lsn_phy_t* phy, lsn_pcap_t* p);  /* Phy ctor argument `LTESniffer_pcap_writer*` (Phy.h:31) */

/* ---- resident (offline / file-replay) path: the IQ of n subframes is ALREADY in device memory ----
 * d_iq layout: [subframe][antenna][15*N] interleaved cf32 (N = FFT size). Subframe i has tti = start_tti + i.
 * update_meta_period: metaFormats.update_formats() runs when (sf_cnt % period) == 0 (LTESniffer_Core.cc:434), 0 = never.
 * stream: a hipStream_t (may be NULL for the default stream). Blocks until all PDUs of the batch have been delivered. */
int lsn_phy_process_device(lsn_phy_t* phy, const void* d_iq, uint32_t n_subframes, uint32_t start_tti,
                           uint32_t update_meta_period, void* stream);
/* same, from host memory (copies through pinned staging) */
int lsn_phy_process_host(lsn_phy_t* phy, const float* iq, uint32_t n_subframes, uint32_t start_tti, uint32_t update_meta_period);
/* same, for INTEGER samples as a radio driver delivers them when asked not to convert (UHD cpu format sc16 / sc8; the reference asks srsran_rf for cf_t,
 * LTESniffer_Core.cc:156,591-600: srsran_rf_recv_with_time_multi into cf_t buffers, so this has no counterpart there): iq [subframe][antenna][15*N] pairs of int16 (LSN_FILE_SC16) or int8 (LSN_FILE_SC8), I first;
 * sample = (float)integer * sample_scale (0 = full scale +-1), converted on the GPU behind the copy - half / a quarter of the bytes of lsn_phy_process_host
 * cross PCIe.  Formats and conversion are lsn_file_cfg_t's (below); LSN_FILE_CF32 is refused (that is lsn_phy_process_host). */
int lsn_phy_process_host_int(lsn_phy_t* phy, const void* iq, uint32_t sample_format, float sample_scale, uint32_t n_subframes, uint32_t start_tti,
                             uint32_t update_meta_period);
/* Pipelined form of lsn_phy_process_device: submit returns as soon as every subframe of the block has been searched and queued for
 * decoding, so the decode / commit tail of one block overlaps the front of the next; lsn_phy_wait returns when everything submitted
 * has been committed (PDUs delivered, in order).  The IQ buffer of a submit must stay valid until the next lsn_phy_wait.
 * lsn_phy_get_perf then describes the whole submit ... wait span.  process_device == submit + wait. */
int lsn_phy_submit_device(lsn_phy_t* phy, const void* d_iq, uint32_t n_subframes, uint32_t start_tti, uint32_t update_meta_period, void* hip_stream);
int lsn_phy_wait(lsn_phy_t* phy);

/* ---- PBCH / MIB ----
 * Replaces srsran_ue_mib_decode + srsran_pbch_mib_unpack of the reference's DECODE_MIB state (LTESniffer_Core.cc:382-395): decode
 * the MIB on a subframe 0 and return the SFN of that subframe (MIB SFN + position of the radio frame in the 40 ms BCH period,
 * "sfn = (sfn + sfn_offset) % 1024").  The cell (bandwidth, id, number of CRS ports for the channel estimate) must have been set;
 * nof_ports is what the CRC mask says (1 / 2 / 4).  Returns 1 = found, 0 = no MIB in this subframe, < 0 = error. */
typedef struct {
  uint32_t found;
  uint32_t sfn;                 /* SFN of the subframe that was handed in */
  uint32_t sfn_offset;          /* radio-frame position inside the BCH period, 0..3 */
  uint32_t nof_prb, nof_ports;  /* dl-Bandwidth, CRC mask */
  uint32_t phich_length;        /* 0 normal, 1 extended */
  uint32_t phich_resources_x6;  /* Ng * 6: 1, 3, 6, 12 */
  uint32_t mib_bits;            /* the 24 MIB bits, first bit = MSB */
} lsn_mib_t;
/* iq: ONE subframe, [nof_rx_antennas][15*N] cf32 (host memory unless iq_on_device) */
int lsn_phy_mib_decode(lsn_phy_t* phy, const void* iq, int iq_on_device, lsn_mib_t* out);
#define LSN_TTI_FROM_MIB 0xFFFFFFFFu /* start_tti of lsn_phy_process_file: take the SFN from the first MIB that decodes (subframes 0, 10, 20, ...
                                        of the file; the subframes in front of it are dropped, as in the reference's DECODE_MIB state) */

/* ---- PSS / SSS cell search ----
 * Replaces rf_search_and_decode_mib(&rf, nof_rx_ant, &cell_detect_config, force_N_id_2, &cell, &search_cell_cfo)
 * (LTESniffer_Core.cc:195-204, configuration :108-112) on a block of samples instead of a radio: physical cell id, the position
 * of the subframe-0/5 boundaries and the carrier offset - for a recording also the -O offset and -c cell id that the
 * reference's file mode asks the user for.  The MIB (bandwidth, ports, PHICH, SFN) then comes from lsn_phy_mib_decode.
 * Needs no Phy.  iq: one antenna, contiguous cf32 at the sampling rate of nof_prb (15 kHz * N), at least
 * (nof_periods + 1) * 75 * N + N samples.  FDD; the cyclic prefix is detected (out->cp).
 * Returns 1 when a cell was found (pss_p2avg >= threshold), 0 when not (out still holds the best guess), < 0 on error. */
typedef struct {
  uint32_t nof_periods;  /* 5 ms periods whose PSS correlation powers are added (1..16; 0 = 1) */
  int32_t force_n_id_2;  /* -1: search the three PSS roots (args.force_N_id_2 of the reference) */
  float threshold;       /* on peak / mean of the PSS correlation power; 20 is a safe floor (noise alone stays below 15) */
} lsn_cell_search_cfg_t;
typedef struct {
  uint32_t found, cell_id, n_id_2, n_id_1;
  uint32_t sf_idx;    /* 0 or 5: index of the subframe that starts at sample sf_start */
  uint32_t pss_pos;   /* first sample of the useful part of the PSS symbol, 0 <= pss_pos < 75 N */
  uint32_t sf_start;  /* 0 <= sf_start < 75 N: lsn_file_cfg_t.offset_time_samples for a recording */
  float pss_peak, pss_p2avg;
  float sss_metric, sss_second;  /* best and second best of the 336 SSS hypotheses */
  float cfo_hz;                  /* from the phase turn between the SSS and PSS symbols (+-7 kHz) */
  float cfo_coarse_hz;           /* from the two halves of the PSS symbol (+-15 kHz; disturbed by the other carriers of a loaded cell) */
  uint32_t cp;                   /* 0 normal, 1 extended cyclic prefix (lsn_cell_t.cp): which of the two SSS positions in front of the PSS symbol carried the better SSS */
} lsn_cell_search_t;
int lsn_cell_search(int device, const void* iq, int iq_on_device, uint64_t nof_samples, uint32_t nof_prb, const lsn_cell_search_cfg_t* cfg,
                    lsn_cell_search_t* out, float* corr_out /* optional, host: [3][75 N] accumulated PSS correlation powers */);

/* ---- IQ capture file replay ----
 * Replaces the file source of the reference's file mode: srsran_ue_sync_init_file_multi(&ue_sync, nof_prb, file, offset_time,
 * offset_freq, nof_rx_antennas) + one srsran_ue_sync_zerocopy per subframe (LTESniffer_Core.cc:252-258,365; options -O / -o,
 * ArgManager.cc:144-149).  File format: complex float32 (or integer pairs, sample_format below), antennas interleaved sample by sample, subframe aligned after the
 * offset (file mode has no PSS tracking); every 15*N samples per antenna are one subframe, counted from start_tti.
 * offset_freq_hz != 0: every subframe is multiplied by exp(-j 2 pi offset_freq n / fs) with n restarting per subframe.
 * The SFN the reference takes from the MIB (LTESniffer_Core.cc:382-420) is an input here (start_tti).
 * Blocks of LSN_FILE_BLOCK (default 800) subframes go to the GPU and are re-laid-out there while the previous blocks are processed
 * (LSN_FILE_SLOTS, default 8, blocks in flight): LSN_FILE_READERS (default 12) threads pread() a block into a pinned buffer;
 * LSN_FILE_MMAP=1 page-locks the blocks in a mapping of the file instead (no CPU copy; slower on the boxes measured).  The block
 * buffers stay allocated between calls, so the first call pays ~0.15 s of allocation. */
/* sample_format: LSN_FILE_CF32 is the reference's (srsran_filesource_init(.., SRSRAN_COMPLEX_FLOAT_BIN) behind srsran_ue_sync_init_file_multi).
 * LSN_FILE_SC16 / LSN_FILE_SC8 are an extension: integer I/Q pairs (int16 / int8, little endian, I first) as the radio delivers them over its
 * link and as srsRAN's SRSRAN_COMPLEX_SHORT_BIN files hold them; the GPU converts, sample = (float)integer * sample_scale (exact for a power of
 * two, one float rounding otherwise), everything behind that is the cf32 path.  A subframe is 245 760 B instead of 491 520 B at 20 MHz / 2 antennas,
 * so a host link that feeds 108 k subframes/s of cf32 feeds twice that of sc16. */
#define LSN_FILE_CF32 0u
#define LSN_FILE_SC16 1u
#define LSN_FILE_SC8 2u
typedef struct {
  uint32_t nof_antennas;        /* interleaved antennas in the file = nof_rx_antennas of the Phy */
  int64_t offset_time_samples;  /* -O: samples (per antenna) skipped at the start */
  float offset_freq_hz;         /* -o: frequency offset correction */
  uint32_t sample_format;       /* LSN_FILE_CF32 (0, the reference's), LSN_FILE_SC16, LSN_FILE_SC8; anything else is refused */
  float sample_scale;           /* integer formats: value of one LSB; 0 = full scale +-1 (1/32768, 1/128); ignored for cf32 */
} lsn_file_cfg_t;
int lsn_phy_process_file(lsn_phy_t* phy, const char* path, const lsn_file_cfg_t* cfg, uint32_t start_tti, uint64_t max_subframes /* 0 = to the end */,
                         uint32_t update_meta_period, uint64_t* subframes_done);
/* Reserves the file source's block buffers (page-locked read blocks + device blocks, 4.7 GB at 20 MHz / 2 antennas) ahead of the first replay:
 * the reference knows at start-up that it replays a file (args.input_file_name, LTESniffer_Core.cc:240-262) - call this behind lsn_phy_set_cell and the
 * first lsn_phy_process_file does not pay for page-locking 1.5 GB inside the replay.  Optional (lsn_phy_process_file reserves what is missing).
 * nof_antennas = lsn_file_cfg_t.nof_antennas of the replay. */
int lsn_phy_prepare_file(lsn_phy_t* phy, uint32_t nof_antennas);

/* ---- security-API sink (the step behind the path: PDSCH_Decoder::run_api_dl_mode, DL_Sniffer_PDSCH.cc:804-879) ----
 * api_mode as ArgManager's -a (ArgManager.cc:63,218): -1 off (default), 0 identity mapping, 2 IMSI catching, 3 all.  For every CRC-ok
 * downlink block the writer thread reports, in record order: paging records (modes 2, 3; decode_imsi_tmsi_paging :84-127: IMSI as 15
 * digits, S-TMSI as 8 hex digits of the m-TMSI, rnti 65534) and the contention resolution identity next to an RRCConnectionSetup
 * (modes 0, 3; :813-877: characters 3..10 of the identity printed in hex).  In UL_MODE a decoded Msg3 (PUSCH of a RAR grant) reports the
 * initial UE identity of its RRCConnectionRequest (modes 0, 3; UL_Sniffer_PUSCH.cc:47-93,306-327: m-TMSI in hex, or the last eight hex
 * digits of the random value - the same characters the connection setup reports); every other decoded uplink block is read as SRB
 * traffic (modes 1-3; :95-247,328-372: MAC / RLC AM / PDCP walk, UL-DCCH head, NAS): UECapabilityInformation (modes 1, 3: id_type
 * 0xFFFFFFFF, value "-"), attach request and identity response identities (modes 2, 3: IMSI / IMEI / IMEISV digits, m-TMSI of a GUTI).
 * Blocks that produced an identity are also written to api_pcap (write_dl_paging_api / write_dl_crnti_api / write_ul_crnti_api,
 * PcapWriter.cc:120-145,177-190) when it is not NULL.
 * Not covered: RRCConnectionReconfiguration / NAS identities of the downlink (LCID 1, :837-850); the body of a UECapabilityInformation is
 * not unpacked. */
typedef struct {
  uint32_t tti; uint16_t rnti;
  uint32_t id_type;   /* Sniffer_dependency.h:42-47: 0 ID_RAN_VAL, 1 ID_TMSI, 2 ID_CON_RES, 3 ID_IMSI, 4 ID_IMEI, 5 ID_IMEISV, 0xFFFFFFFF none */
  uint32_t msg_type;  /* Sniffer_dependency.h:49-55: 0 MSG_CON_REQ, 1 MSG_CON_SET, 2 MSG_ATT_REQ, 3 MSG_ID_RES, 4 MSG_UE_CAP, 5 MSG_PAGING, 6 MSG_CON_RECONFIG (M-TMSI of the GUTI an attach accept inside an RRCConnectionReconfiguration assigns) */
  char value[24];     /* the string print_api_dl receives */
} lsn_api_event_t;
typedef void (*lsn_api_sink_t)(void* user, const lsn_api_event_t* ev);
int lsn_phy_set_api_mode(lsn_phy_t* phy, int api_mode, lsn_api_sink_t cb, void* user, lsn_pcap_t* api_pcap);

/* ---- uplink (PUSCH) ----
 * lsn_phy_set_ul_config replaces srsran_enb_ul_set_cell(&enb_ul, cell, &ul_cfg.dmrs, ...) (SubframeWorker.cc:258-262); the two
 * values come from SIB2 (ULSchedule::set_config, ULSchedule.cc:140-158: cyclicShift, groupAssignmentPUSCH).
 * lsn_phy_pusch_decode runs srsran_enb_ul_fft + srsran_chest_ul_estimate_pusch + srsran_pusch_decode
 * (UL_Sniffer_PUSCH.cc:392,256,262) for a whole list of grants on n_subframes of the uplink antenna.
 * Round-1 scope: one antenna, no group/sequence hopping, type-1 frequency hopping only (type 2 grants come back with crc_ok = 0), no SRS, L_prb >= 3 of the 2^a3^b5^c set; control information on the
 * PUSCH is located and skipped / erased, not decoded
 * (UL_Sniffer_PUSCH.cc:3-10); any other grant comes back with crc_ok = 0. */
typedef struct {
  uint32_t cyclic_shift;    /* SIB2 cyclicShift */
  uint32_t delta_ss;        /* SIB2 groupAssignmentPUSCH */
  uint32_t hopping_offset;  /* SIB2 pusch-HoppingOffset (ul_cfg.hopping.n_rb_ho, SubframeWorker.cc:271-273): second-slot position of type-1 hopping grants */
  uint32_t group_hopping_enabled;     /* SIB2 ul-ReferenceSignalsPUSCH.groupHoppingEnabled (ULSchedule.cc:143-146): sequence group per slot, 36.211 5.5.1.3 */
  uint32_t sequence_hopping_enabled;  /* ... sequenceHoppingEnabled: base sequence number per slot for >= 6 PRB, 36.211 5.5.1.4 */
} lsn_ul_cfg_t;
typedef struct {
  uint32_t sf;       /* subframe index inside ul_iq (tti = start_tti + sf) */
  uint16_t rnti;
  uint16_t n_dmrs;   /* cyclic shift field of DCI 0 (0 for a RAR grant) */
  uint32_t n_prb, L_prb;
  uint32_t mod;      /* bits per symbol: 2, 4, 6, 8 */
  uint32_t tbs;      /* bits */
  int rv;
  /* control information multiplexed into the PUSCH (TS 36.212 5.2.2.6-8), located so that the UL-SCH bits are de-multiplexed
   * correctly; the reference's settings are UL_Sniffer_PUSCH.cc:429-450 with the offsets I_ack = 10, I_cqi = 8, I_ri = 11 of
   * MCSTracking.cc:1534-1538 */
  uint32_t nof_ack;  /* HARQ-ACK bits 0..2 (uci_cfg.ack[0].nof_acks) */
  uint32_t cqi_bits; /* size of the CQI report, 0 = none (aperiodic request: wideband 4, UE-selected sub-band 5, higher-layer sub-band 4 + 2 N) */
  uint32_t ri_bits;  /* rank indication bits (1 with a CQI request) */
  uint32_t hop;      /* 0: both slots on n_prb; 1: type-1 frequency hopping, slot 1 starts at n_prb_slot1 (36.213 8.4.1) */
  uint32_t n_prb_slot1;
  /* 1 + betaOffset-ACK-Index / betaOffset-CQI-Index / betaOffset-RI-Index of the UE (ul_cfg.pusch.uci_offset = ue_config.uci_config,
   * UL_Sniffer_PUSCH.cc:433-435; 36.213 Tables 8.6.3-1/-2/-3); 0 = the reference's defaults 10 / 8 / 11.  A reserved index makes the
   * grant undecodable (crc_ok = 0). */
  uint32_t beta_offset_ack_idx_p1, beta_offset_cqi_idx_p1, beta_offset_ri_idx_p1;
} lsn_pusch_grant_t;
typedef struct { uint32_t crc_ok; uint32_t iterations; float snr_db; uint32_t payload_off; } lsn_pusch_result_t;
int lsn_phy_set_ul_config(lsn_phy_t* phy, const lsn_ul_cfg_t* cfg);
/* The SIB2 fields the sniffer reads (ULSchedule::set_config, ULSchedule.cc:140-158; SubframeWorker.cc:271-273). */
typedef struct {
  uint32_t n_sb, hopping_mode, pusch_hop_offset, enable_64qam;                                                  /* pusch-ConfigBasic */
  uint32_t group_hopping_enabled, group_assignment_pusch, sequence_hopping_enabled, cyclic_shift;                /* ul-ReferenceSignalsPUSCH */
  uint32_t root_seq_idx, prach_config_idx, high_speed_flag, zero_corr_zone, prach_freq_offset;                   /* prach-Config */
} lsn_sib2_t;
/* ULSchedule::get_config / getSIB2 (ULSchedule.h:90-93, DL_Sniffer_PDSCH.h:137): in UL_MODE without lsn_phy_set_ul_config the commit stage
 * runs PDSCH_Decoder::decode_SIB (DL_Sniffer_PDSCH.cc:459-560) on every subframe until a SystemInformation with SIB2 decodes, writes
 * that one SI-RNTI record, configures DMRS / hopping offset / PRACH detector from it and decodes PUSCH from the next subframe on.
 * Returns 1 when an uplink configuration is in use (ul filled; *from_sib2 = 1 and sib2 filled when it was learned from SIB2), else 0.
 * Any output pointer may be NULL.  Call between lsn_phy_wait and the next submit. */
int lsn_phy_get_ul_config(lsn_phy_t* phy, lsn_ul_cfg_t* ul, lsn_sib2_t* sib2, uint32_t* from_sib2);
/* BCCH-DL-SCH-Message -> SIB2 (the parser behind decode_SIB): 0 = does not unpack, 1 = unpacks without SIB2 in front, 2 = SIB2 (out filled) */
int lsn_sib2_decode(const uint8_t* pdu, uint32_t len, lsn_sib2_t* out);
/* ul_iq: [n_subframes][15*N] interleaved cf32 of the uplink antenna, host memory (iq_on_device = 0) or device memory (1).
 * payloads (may be NULL): decoded transport blocks, tbs/8 bytes each at results[i].payload_off. */
int lsn_phy_pusch_decode(lsn_phy_t* phy, const void* ul_iq, int iq_on_device, uint32_t n_subframes, uint32_t start_tti,
                         const lsn_pusch_grant_t* grants, uint32_t n_grants, lsn_pusch_result_t* results, uint8_t* payloads, size_t payload_cap);
long lsn_phy_tap_ul(lsn_phy_t* phy, int what /* 0: uplink grid of subframe `index`, 1: LLRs of grant `index`, 2: PRACH correlation power of occasion `index` */,
                    uint32_t index, void* out, size_t cap);

/* ---- uplink (PRACH) ----
 * lsn_phy_set_prach_config replaces PUSCH_Decoder::set_rach_config (UL_Sniffer_PUSCH.cc:640-653: srsran_prach_init,
 * srsran_prach_set_cfg, srsran_prach_set_detect_factor(60)); the fields are SIB2's prach-ConfigInfo as the reference copies
 * them in ULSchedule::set_config (ULSchedule.cc:149-154).  lsn_phy_prach_detect replaces PUSCH_Decoder::work_prach
 * (UL_Sniffer_PUSCH.cc:656-713: srsran_prach_tti_opportunity + srsran_prach_detect_offset on one uplink subframe) for a
 * block of uplink subframes.  Scope: preamble format 0 (config_idx 0..15, the format that fits the one subframe the
 * reference hands over), unrestricted cyclic shifts (hs_flag = 0); anything else is LSN_ERROR_INVALID_INPUTS.
 * The logical->physical root map (TS 36.211 Table 5.7.2-4, srsRAN's prach_zc_roots[838]) is passed in by the caller;
 * with zc_roots = NULL the numbers root_seq_idx + i (+1) are used as physical roots.
 * In UL_MODE (sniffer_mode = 1) a configured detector also runs on the uplink antenna of every processed PRACH occasion
 * and reports through the sink (the reference prints the strongest preamble there). */
typedef struct {
  uint32_t config_idx;       /* prach-ConfigIndex */
  uint32_t root_seq_idx;     /* rootSequenceIndex 0..837 */
  uint32_t zero_corr_zone;   /* zeroCorrelationZoneConfig 0..15 */
  uint32_t freq_offset;      /* prach-FreqOffset: first of the 6 PRBs */
  uint32_t hs_flag;          /* highSpeedFlag, must be 0 */
  float detect_factor;       /* peak / mean threshold, 0 = 60 (UL_Sniffer_PUSCH.cc:651) */
  const uint16_t* zc_roots;  /* 838 entries or NULL (only read during the call) */
} lsn_prach_cfg_t;
typedef struct {
  uint32_t sf;        /* subframe index inside the block (tti = start_tti + sf) */
  uint32_t preamble;  /* 0..63 */
  uint32_t offset;    /* peak lag inside the cyclic-shift window, units of T_SEQ / 839 */
  float offset_sec;   /* the same in seconds (srsran_prach_detect_offset's t_offsets) */
  float p2avg;        /* peak / mean correlation power */
} lsn_prach_det_t;
typedef void (*lsn_prach_sink_t)(void* user, uint32_t tti, const lsn_prach_det_t* det, uint32_t n_det);
int lsn_phy_set_prach_config(lsn_phy_t* phy, const lsn_prach_cfg_t* cfg);
/* returns the number of detections written to out (<= cap), in (subframe, root, cyclic shift) order, or < 0 */
int lsn_phy_prach_detect(lsn_phy_t* phy, const void* ul_iq, int iq_on_device, uint32_t n_subframes, uint32_t start_tti, lsn_prach_det_t* out,
                         uint32_t cap);
void lsn_phy_set_prach_sink(lsn_phy_t* phy, lsn_prach_sink_t cb, void* user);
int lsn_prach_tti_opportunity(uint32_t config_idx, uint32_t tti); /* srsran_prach_tti_opportunity(p, tti, -1), format 0 */

/* ---- measurement + parity taps (not part of the reference surface) ---- */
enum { LSN_TAP_GRID = 0, LSN_TAP_CE = 1, LSN_TAP_PDCCH_LLR = 2, LSN_TAP_CHEST = 3, LSN_TAP_CFI = 4, LSN_TAP_CANDIDATES = 5,
       LSN_TAP_CCE_POWER = 6, LSN_TAP_ACCEPTED = 7, LSN_TAP_RB_POWER = 8,
       /* (LSN_TAP_CANDIDATES: [160 locations][8 sizes] of {u64 payload bits, u32 CRC remainder, u32 flags}; with candidate pruning on, a slot the blind decoder left
        * out and the search never asked for has flags = 0x80 - lsn_phy_set_candidate_pruning(phy, 0) gives the exhaustive table) */
       /* stage C (a14, srsran_ue_dl_decode_pdsch as called at DL_Sniffer_PDSCH.cc:997,1110,1207): retained only for batches processed after
        * lsn_phy_set_stage_c_taps(phy, 1) - the arenas of a decode launch are recycled otherwise.  For these the index argument of lsn_phy_tap
        * is the DECODE JOB of the chunk (one job = one decode call of one accepted DCI with one MCS table), not a subframe. */
       LSN_TAP_PDSCH_JOBS = 9,    /* lsn_tap_job_t of every job of the chunk (index ignored) */
       LSN_TAP_PDSCH_LLR16 = 10,  /* descrambled int16 soft bits of the job: codeword 0 (nof_re x Qm), then codeword 1 */
       LSN_TAP_RM_WORDS = 11,     /* per code block (TB 0 blocks, then TB 1), K + 12 u32: the de-rate-matched block as k_rm hands it to k_turbo - word
                                     (x % W) P + x / W = d0[x] | d1[x] << 10 | d2[x] << 20 (10-bit two's complement), P = windows, W = K / P; words K + 4 s + j =
                                     stream s at position K + j (int32) */
       LSN_TAP_CB_RESULT = 12 };  /* lsn_tap_cb_t per code block, same order */
typedef struct { uint32_t sf, tti, rnti, nof_re, qm[2], llr_len[2], tbs[2], crc[2], ncb, have, done; float p_a_db; } lsn_tap_job_t;
typedef struct { uint32_t tb, K, F, E, rv, ok, iters, skipped; } lsn_tap_cb_t;  /* skipped: not decoded because the first block of its TB had failed */
/* copies tap `what` of subframe `sf_in_batch` of the LAST processed batch into out (host); returns bytes written or <0 */
long lsn_phy_tap(lsn_phy_t* phy, int what, uint32_t sf_in_batch, void* out, size_t cap);
int lsn_phy_set_stage_c_taps(lsn_phy_t* phy, int enable);
/* lsn_phy_mib_decode + the 480 raw (not descrambled) PBCH soft bits of the subframe */
int lsn_phy_mib_decode_llr(lsn_phy_t* phy, const void* iq, int iq_on_device, lsn_mib_t* out, float* llr_raw480);
typedef struct {
  double ms_stage_a, ms_search, ms_stage_c, ms_commit, ms_total; /* wall clock of the last process call */
  double kernel_ms[16];                                          /* HIP-event time per kernel class, last call */
  uint64_t kernel_launches[16];
  uint64_t algo_bytes;        /* algorithmic HBM bytes of the last call (SURVEY.md 8d formula) */
  uint64_t turbo_algo_bytes;  /* part of algo_bytes moved by the turbo kernels */
  uint64_t turbo128_algo_bytes; /* ... of which by k_turbo<128> */
  uint64_t nof_tb_decodes, nof_cb_decodes, nof_turbo_iterations, nof_candidates_decoded, nof_ondemand_decodes, nof_pdus;
  double ms_search_core;      /* part of ms_search inside the FALCON decision tree proper */
  double ms_rar;              /* part of ms_search spent waiting for on-demand RAR decodes */
  uint64_t turbo_cyc_rm, turbo_cyc_map, turbo_cyc_out;  /* shader cycles summed over code blocks: rate-dematch / MAP iterations / output */
  double ms_wait_front, ms_wait_slot, ms_drain;          /* search thread waiting for stage A / front thread waiting for a free chunk slot / final wait for the commits */
  uint64_t nof_turbo_iterations_run;                     /* iterations executed (equals nof_turbo_iterations) */
  uint64_t nof_ondemand_commit[4];                       /* decodes created at commit: [0] p-a changed, [1] table known at commit but unknown when planned or vice versa, [2] no job planned at all, [3] other */
  double ms_ondemand_commit;                             /* commit-thread time inside those decodes */
  uint64_t nof_pusch_2prb_skipped;                       /* (always 0 since round 3: 2-PRB grants are decoded) */
  uint64_t nof_pusch_on_unverified_dmrs;                 /* UL_MODE: PUSCH attempts on 1- / 2-PRB allocations, whose reference signals come from the restated (structure-checked, not text-verified) 36.211 Tables 5.5.1.2-1 / -2 */
  uint64_t nof_tb_on_derived_tbs;                        /* transport-block decodes whose size came from the DERIVED TBS rows I_TBS 27..33 (spec/gen_tables.py): a real capture that fails exactly there points at the table */
  uint64_t nof_decode_jobs, nof_decode_jobs_used, nof_speculative_jobs;  /* PDSCH decode calls run / of those the commit stage consumed (what the reference would have run) / run ahead only because the
                                                            RNTI's table might be known by commit time (second-table attempt although the first one passed a CRC) */
  /* decode jobs by why they were run: [0] first attempt of the plan, [1] second-table attempt after the first failed on every TB, [2] second-table attempt run
   * ahead although the first passed a CRC (speculative), [3] RA-RNTI candidates decoded ahead of the search, [4] decoded on demand (search / commit turn);
   * jobs / turbo iterations run, and the part of both the commit stage never looked at */
  uint64_t jobs_by_kind[5], jobs_unused_by_kind[5], iters_by_kind[5], iters_unused_by_kind[5];
  uint64_t nof_table_hints_used, nof_table_hints_missed;  /* DCIs planned for the 256QAM table alone on the decode threads' own evidence / of those the commit wanted the 64QAM-table attempt of after all (engine totals) */
  uint64_t nof_candidate_misses;  /* slots of the candidate table the blind decoder had left out (lsn_phy_set_candidate_pruning) and the search had decoded on demand */
  double ms_harq[3];              /* harq_mode = 1, commit-thread time: [0] predicting the retransmissions of the chunks (harqScout), [1] inside the batches (descriptors, launches, wait), [2] bringing the touched buffers home at the end of the turns */
  uint64_t nof_harq_combines[4];  /* harq_mode = 1: [0] batches of retransmissions combined and decoded ahead of the commit walk, [1] combined decodes the walk took from a batch, [2] ... it had to run alone inside its turn, [3] batch results nobody asked for */
} lsn_perf_t;
int lsn_phy_get_perf(lsn_phy_t* phy, lsn_perf_t* out);
enum { LSN_K_OFDM = 0, LSN_K_CHEST, LSN_K_CHEST_FIN, LSN_K_PCFICH, LSN_K_PDCCH_LLR, LSN_K_CCE_POWER, LSN_K_VITERBI,
       LSN_K_PDSCH_PREP, LSN_K_PDSCH_DEMOD, LSN_K_TURBO, LSN_K_RB_POWER, LSN_K_TURBO128, LSN_K_RM, LSN_K_COUNT };
/* LSN_K_TURBO = k_turbo<64> (one wavefront per code block), LSN_K_TURBO128 = k_turbo<128> (two), LSN_K_RM = k_rm (rate de-matching in front of both) */
const char* lsn_kernel_name(int k);
const char* lsn_version(void);

#ifdef __cplusplus
}
#endif
#endif
