/*
 * lsn_oracle.h - CPU ORACLE for the LTESniffer per-subframe hot path.
 *
 * >>> TEST INFRASTRUCTURE ONLY. <<<  Nothing in ltesniffer_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it (as the checker /
 * as the timed CPU baseline), never as the shipped path.
 *
 * PARITY UNPINNED: the reference's arithmetic for this path lives in an un-vendored, unpinned
 * srsRAN fork (github.com/ShaoPaoLao/srsRAN2 @ master, /root/reference/external/cmake/
 * srsRAN.CMakeLists.txt.in:8-9) that is absent from /root/reference and cannot be built here.
 * This oracle is a plain scalar C restatement of
 *   - the reference's in-tree control logic (file:line cited at each function), and
 *   - the published 3GPP algorithms (TS 36.211/36.212/36.213) for the DSP that the reference
 *     reaches through srsran_* calls (call sites cited).
 * The only golden vectors the reference ships pin the MAC-LTE pcap framing
 * (the captures under pcap_file_example/); tests/test_oracle_pcap.py checks the writer against them.
 *
 * Float arithmetic contract (so that a GPU implementation can be BIT-IDENTICAL): every float
 * expression is evaluated in binary32 with one rounding per + - * / (compile with
 * -ffp-contract=off, no fast-math); reductions use the fixed order of o_reduce256(); all
 * transcendental calls (cos/sin/exp/log10/atan2) happen on the host only.
 */
#ifndef LSN_ORACLE_H
#define LSN_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float r, i; } ocf_t;

#define O_MAX_PRB 110
#define O_MAX_PORTS 4
#define O_MAX_RX 2
#define O_NSYMB 14
#define O_MAX_CCE 87
#define O_MAX_NUM_OF_CCE 84 /* falcon_pdcch.h:36 - the search never looks past CCE 83 */
#define O_MAX_LOCATIONS 160 /* falcon_ue_dl.h:39 MAX_CANDIDATES_BLIND */
#define O_DCI_MAX_BITS 128
#define O_NOF_FORMATS 9

/* RNTI constants: srsRAN phy_common.h [not in tree], SURVEY.md appendix E */
#define O_SIRNTI 0xFFFF
#define O_PRNTI 0xFFFE
#define O_MRNTI 0xFFFD
#define O_RARNTI_START 0x0001
#define O_RARNTI_END 0x000A
#define O_CRNTI_START 0x000B
#define O_CRNTI_END 0xFFF3
#define O_RNTI_ISUSER(r) ((r) >= O_CRNTI_START && (r) <= O_CRNTI_END)
#define O_RNTI_ISRAR(r) ((r) >= O_RARNTI_START && (r) <= O_RARNTI_END)

/* DCI formats, in the order of falcon_ue_all_formats (DCISearch.cc:84-95): index == global_index */
enum { O_FMT0 = 0, O_FMT1, O_FMT1A, O_FMT1B, O_FMT1C, O_FMT1D, O_FMT2, O_FMT2A, O_FMT2B };
/* srsRAN's srsran_dci_format_t order differs (0,1,1A,1B?,...): the reference only compares formats
 * with ==, <= FORMAT1A, > FORMAT1A and >= FORMAT2; srsRAN enum order is 0,1,1A,1B,1C,1D,2,2A,2B,
 * identical to the list above, so the same integers are used for both. */

enum { O_MOD_QPSK = 2, O_MOD_16QAM = 4, O_MOD_64QAM = 6, O_MOD_256QAM = 8 };
enum { O_TX_PORT0 = 0, O_TX_DIVERSITY, O_TX_SPATIALMUX, O_TX_CDD };
enum { O_TABLE_64QAM = 0, O_TABLE_256QAM = 1, O_TABLE_UNKNOWN = 2, O_TABLE_BOTH = 3, O_TABLE_FULL = 4 };

typedef struct {
  uint32_t nof_prb;   /* 6,15,25,50,100 (power-of-two FFT sizes only) */
  uint32_t nof_ports; /* 1, 2 or 4 CRS ports (4: every channel in SFBC-FSTD transmit diversity; spatial multiplexing grants are not decodable, as in the reference's srsRAN) */
  uint32_t id;        /* physical cell id 0..503 */
  uint32_t phich_ng_x6; /* Ng*6: 1 (=1/6), 3, 6, 12 ; LTESniffer_Core.cc:211-212 forces 1/6 */
  uint32_t pusch_hop_offset; /* SIB2 pusch-HoppingOffset = n_rb_ho of the uplink grant conversion (SubframeWorker.cc:271-273); 0 until SIB2 is known */
  uint32_t cp;        /* srsran_cell_t.cp: 0 = normal cyclic prefix (7 symbols per slot), 1 = extended (6 symbols per slot, CP of N/4 samples).  The reference
                       * hands whatever the cell search found to srsran_ue_dl_set_cell (SubframeWorker.cc:102, LTESniffer_Core.cc:292-299; file and manual
                       * mode force normal, :210,243).  Grids keep 14 rows per antenna; an extended-CP subframe fills rows 0..11, rows 12, 13 stay zero. */
} o_cell_t;
/* symbols per slot / per subframe (36.211 Table 6.2.3-1) */
static inline int o_nslot(const o_cell_t* c) { return c->cp ? 6 : 7; }
static inline int o_nsym(const o_cell_t* c) { return c->cp ? 12 : 14; }
/* CRS-bearing symbols (36.211 6.10.1.2): ports 0, 1 on symbols 0 and N_symb - 3 of both slots, ports 2, 3 on symbol 1 of both slots */
static inline int o_crs_sym01(const o_cell_t* c, int s) { const int ns = o_nslot(c); return (s >> 1) * ns + ((s & 1) ? ns - 3 : 0); }
static inline int o_crs_sym23(const o_cell_t* c, int s) { return s * o_nslot(c) + 1; }
static inline int o_is_crs_sym01(const o_cell_t* c, int l) { const int ns = o_nslot(c), q = l % ns; return q == 0 || q == ns - 3; }

/* ---------- bit-level primitives (o_bits.c) ---------- */
uint32_t o_crc_bits(uint32_t poly, int order, const uint8_t* bits, int n);
#define O_CRC24A 0x1864CFBu
#define O_CRC24B 0x1800063u
#define O_CRC16 0x11021u
#define O_CRC8 0x19Bu
void o_gold(uint32_t cinit, uint8_t* c, int len);
void o_unpack_bytes(const uint8_t* bytes, uint8_t* bits, int nbits);
void o_pack_bits(const uint8_t* bits, uint8_t* bytes, int nbits);
float o_reduce256(const float* v, int n);

/* ---------- OFDM + channel estimation + control region (o_phy.c) ---------- */
int o_fft_size(uint32_t nof_prb);
int o_fft_twiddle_len(int N);
void o_fft_twiddles(int N, ocf_t* w /* N/2 */);
void o_fft(int N, const ocf_t* w, ocf_t* a /* in place, natural order in and out */);
/* in: 15*N samples of one subframe, cfo_phase_inc: NCO increment (0 = no correction); out: grid[14][12*nprb] */
void o_ofdm_rx(const o_cell_t* cell, const ocf_t* in, uint32_t nco_dphi, ocf_t* grid);
void o_nco_tables(ocf_t* coarse /*4096*/, ocf_t* fine /*1024*/);
uint32_t o_nco_dphi(float cfo_hz, int fft_size);

typedef struct {
  float noise[O_MAX_RX][O_MAX_PORTS];
  float rsrp[O_MAX_RX][O_MAX_PORTS];
  float cepow[O_MAX_RX][O_MAX_PORTS]; /* mean |smoothed pilot|^2 */
  ocf_t cfo_corr;      /* sum of pilot correlations one slot apart */
  float noise_avg, rsrp_avg, snr_db, cfo_hz, chan_ref; /* chan_ref = sum cepow */
} o_chest_res_t;

void o_crs_table(const o_cell_t* cell, uint32_t sf_idx, ocf_t* crs /* [port][4][2*nprb]; ports 2, 3: rows 0, 1 = symbols 1, 8 */);
/* grid[rx][14][nre] -> ce[port][rx][14][nre], pilots kept in work arrays */
void o_chest(const o_cell_t* cell, uint32_t nof_rx, uint32_t sf_idx, const ocf_t* grid, ocf_t* ce,
             o_chest_res_t* res);

typedef struct { uint16_t k0; uint8_t l; uint8_t assigned; } o_reg_t;
typedef struct {
  uint32_t nof_regs[3];           /* PDCCH REGs per CFI */
  uint32_t nof_cce[3];
  uint16_t pdcch_reg_k0[3][800];  /* quadruplet q (post de-interleave, = CCE*9 + i) -> first RE index */
  uint8_t pdcch_reg_l[3][800];
  uint16_t pcfich_k0[4];
  uint32_t ngroups_phich;
} o_regs_t;
void o_regs_init(const o_cell_t* cell, o_regs_t* regs);
/* PCFICH: returns cfi 1..3 (36.211 6.7), corr out optional */
uint32_t o_pcfich_decode(const o_cell_t* cell, const o_regs_t* regs, uint32_t nof_rx, uint32_t sf_idx,
                         const ocf_t* grid, const ocf_t* ce, float noise, float* corr3);
/* PDCCH LLRs for the decoded cfi: llr[nof_cce*72] (positive = bit 1) */
void o_pdcch_llr(const o_cell_t* cell, const o_regs_t* regs, uint32_t nof_rx, uint32_t sf_idx, uint32_t cfi,
                 const ocf_t* grid, const ocf_t* ce, float noise, float* llr);
void o_subframe_power(const o_cell_t* cell, const ocf_t* grid_ant0, float* rb_power_db, float* pmin, float* pmax);

/* ---------- convolutional code / DCI candidate decode (o_conv.c) ---------- */
void o_rm_conv_rx(const float* e, int E, float* out, int D3 /* 3*(n+16) */);
void o_rm_conv_rx_off(const float* e, int E, float* out, int D3, int skip /* positions of the circular buffer in front of e[0] */);
uint16_t o_dci_decode_off(const float* llr, int E, int nof_bits, uint8_t* payload, int skip);
void o_viterbi_tb(const uint8_t* sym /* 3*D quantised */, int D, uint8_t* bits);
/* decode one candidate: llr points at ncce*72, E = 72<<L, n payload bits; returns crc_rem (=RNTI) */
uint16_t o_dci_decode(const float* llr, int E, int nof_bits, uint8_t* payload);

/* ---------- DCI formats and resource allocation (o_dci.c) ---------- */
uint32_t o_dci_format_sizeof(const o_cell_t* cell, int format);
typedef struct { uint32_t mcs_idx; int rv; uint32_t ndi; uint32_t cw_idx; } o_dci_tb_t;
typedef struct {
  uint16_t rnti; int format; uint32_t L, ncce;
  int alloc_type;          /* 0,1,2 */
  uint32_t rbg_bitmask;    /* type0 */
  uint32_t t1_vrb_bitmask, t1_rbg_subset, t1_shift;
  uint32_t riv; int t2_dist; int t2_ngap2; int t2_nprb1a_is2; /* type2 */
  uint32_t pid; o_dci_tb_t tb[2]; uint32_t tb_cw_swap; uint32_t pinfo; uint32_t tpc;
  int is_ra_order;
} o_dci_dl_t;
typedef struct {
  uint16_t rnti; uint32_t L, ncce;
  uint32_t freq_hop_fl; uint32_t riv; uint32_t mcs_idx; int rv; uint32_t ndi; uint32_t tpc; uint32_t n_dmrs; uint32_t cqi_req;
  int hop_type; /* -1 no hopping; 36.213 Table 8.4-2: 0 = +N/4, 1 = -N/4, 2 = +N/2 (type 1), 3 = type 2 */
} o_dci_ul_t;
typedef struct { uint32_t mcs_idx; int rv; uint32_t cw_idx; int enabled; int mod; int tbs; int nof_bits; } o_tb_t;
typedef struct {
  uint8_t prb_idx[2][O_MAX_PRB]; uint32_t nof_prb; uint32_t nof_re; uint32_t nof_tb;
  o_tb_t tb[2]; int tx_scheme; uint32_t pmi; uint32_t nof_layers;
} o_pdsch_grant_t;
typedef struct { uint32_t L_prb, n_prb; uint32_t mcs_idx; int mod; int tbs; int rv;
                 uint32_t n_prb2; /* first PRB in slot 1 when hop == 1 */ uint32_t hop; /* 0 none, 1 type-1 hopping, 2 type-2 (not decoded) */ } o_pusch_grant_t;

int o_dci_unpack_dl(const o_cell_t* cell, const uint8_t* payload, uint32_t nof_bits, int format, uint16_t rnti, o_dci_dl_t* dci);
int o_dci_unpack_ul(const o_cell_t* cell, const uint8_t* payload, uint32_t nof_bits, uint16_t rnti, o_dci_ul_t* dci);
int o_ra_dl_dci_to_grant(const o_cell_t* cell, uint32_t sf_idx, uint32_t cfi, int use_256qam_table, const o_dci_dl_t* dci, o_pdsch_grant_t* g);
int o_ra_ul_dci_to_grant(const o_cell_t* cell, const o_dci_ul_t* dci, o_pusch_grant_t* g);
int o_ra_ul_dci_to_grant_256(const o_cell_t* cell, const o_dci_ul_t* dci, o_pusch_grant_t* g);
int o_config_mimo(const o_cell_t* cell, int format, const o_dci_dl_t* dci, o_pdsch_grant_t* g);
uint32_t o_ra_nof_re(const o_cell_t* cell, uint32_t sf_idx, uint32_t cfi, const o_pdsch_grant_t* g);
int o_tbs_from_idx(int i_tbs, uint32_t n_prb);
/* PDSCH RE position helper shared by nof_re and extraction: 1 if (l,k) carries PDSCH for this cell/subframe */
int o_pdsch_re_ok(const o_cell_t* cell, uint32_t sf_idx, uint32_t l, uint32_t k);

/* ---------- search space + FALCON bookkeeping (o_falcon.c) ---------- */
uint32_t o_validate_location(uint32_t nof_cce, uint32_t ncce, uint32_t l, uint32_t nsubframe, uint16_t rnti);
typedef struct o_rntiman o_rntiman_t;
o_rntiman_t* o_rntiman_new(uint32_t nformats, uint32_t max_cand_per_step, uint32_t threshold);
void o_rntiman_free(o_rntiman_t*);
void o_rntiman_add_evergreen(o_rntiman_t*, uint16_t a, uint16_t b, uint32_t f);
void o_rntiman_add_forbidden(o_rntiman_t*, uint16_t a, uint16_t b, uint32_t f);
void o_rntiman_add_candidate(o_rntiman_t*, uint16_t rnti, uint32_t f);
int o_rntiman_validate_and_refresh(o_rntiman_t*, uint16_t rnti, uint32_t f);
void o_rntiman_activate_and_refresh(o_rntiman_t*, uint16_t rnti, uint32_t f, int reason);
int o_rntiman_is_forbidden(o_rntiman_t*, uint16_t rnti, uint32_t f);
uint32_t o_rntiman_get_frequency(o_rntiman_t*, uint16_t rnti, uint32_t f);
int o_rntiman_get_activation_reason(o_rntiman_t*, uint16_t rnti);
void o_rntiman_step_time(o_rntiman_t*);
uint32_t o_rntiman_nof_active(o_rntiman_t*);
enum { O_ACT_UNSET = 0, O_ACT_EVERGREEN, O_ACT_RAR, O_ACT_SHORTCUT, O_ACT_HISTOGRAM, O_ACT_OTHER };

/* ---------- PDSCH (o_pdsch.c) ---------- */
typedef struct { int C, Cp, Cm, Kp, Km, F, tbs; } o_cbsegm_t;
int o_cbsegm(o_cbsegm_t* s, int tbs);
int o_qpp_find(int K, int* f1, int* f2);
/* demod: extract+equalise+soft-demod+descramble one grant. llr_out[cw][nof_re*Qm] int16 */
int o_pdsch_demod(const o_cell_t* cell, uint32_t nof_rx, uint32_t sf_idx, uint32_t cfi, uint16_t rnti,
                  const o_pdsch_grant_t* g, const ocf_t* grid, const ocf_t* ce, float noise, float chan_ref,
                  float rho_a_db, int16_t* llr_cw0, int16_t* llr_cw1);
/* rate-dematch CB r of a codeword into sys/par1/par2 arrays (K+4 each, natural turbo order incl. tail) */
void o_rm_turbo_rx_cb(const int16_t* e, int E, int K, int F, int rv, int16_t* d3 /* 3*(K+4): d0|d1|d2 */);
/* windowed max-log-MAP turbo decoder; returns number of iterations run, bits out[K]; crc_ok flag */
int o_turbo_decode_cb(const int16_t* d3, int K, int max_iter, uint32_t crc_poly, uint8_t* bits, int* crc_ok);
/* whole transport block: llr e[G] -> payload bytes (tbs/8); returns crc ok */
int o_pdsch_decode_tb(const int16_t* e, int G, int tbs, int Qm, int nof_layers_rm, int rv, int max_iter,
                      uint8_t* payload, int* iters_total);
#define O_HARQ_CB_STRIDE (3 * (6144 + 4))
#define O_HARQ_MAX_CB 16
#define O_HARQ_KEEP_STRIDE (1 + 6144) /* per code block: passed flag + decoded bits */
int o_pdsch_decode_tb_harq(const int16_t* e, int G, int tbs, int Qm, int nof_layers_rm, int rv, int max_iter,
                           uint8_t* payload, int* iters_total, int16_t* acc, int combine, uint8_t* keep);
int o_turbo_nwin(int K);
/* ---------- stage-C recorder (o_trace.c): soft bits / de-rate-matched streams / per-code-block verdicts of every decode call ---------- */
typedef struct { uint32_t tti, rnti, nof_re, qm[2], llr_len[2], ncb, cb_first, is_ul; } o_trace_job_hdr_t;
typedef struct { uint32_t job, tb, K, F, E, rv, iters, ok; } o_trace_cb_hdr_t;
void o_trace_enable(int on); /* also clears the log */
int o_trace_enabled(void);
void o_trace_begin_job(uint32_t tti, uint16_t rnti, uint32_t nof_re, const int* qm, const int16_t* llr0, const int16_t* llr1, int is_ul);
void o_trace_set_tb(int tb);
void o_trace_cb(int K, int F, int E, int rv, const int16_t* d3, int iters, int ok);
uint32_t o_trace_njobs(void);
uint32_t o_trace_ncbs(void);
int o_trace_job(uint32_t i, o_trace_job_hdr_t* out);
int o_trace_job_llr(uint32_t i, int cw, int16_t* out, uint32_t cap);
int o_trace_cb_get(uint32_t i, o_trace_cb_hdr_t* out, int16_t* d3, uint32_t cap);
/* ---------- second-opinion decoders (o_second.c): full-trellis 16-bit-input turbo, float tail-biting Viterbi ---------- */
void o_pdsch_set_llr_clip(int clip); /* demodulator soft-bit clip, 511 by contract */
int o_turbo_decode_cb_second(const int32_t* d3, int K, int max_iter, uint32_t crc_poly, uint8_t* bits, int* crc_ok);
int o_pdsch_decode_tb_second(const int16_t* e, int G, int tbs, int Qm, int nof_layers_rm, int rv, int max_iter, uint8_t* payload, int* iters_total);
uint16_t o_dci_decode_second(const float* llr, int E, int nof_bits, uint8_t* payload);

/* ---------- PBCH / MIB (o_pbch.c) ---------- */
typedef struct { int found; uint32_t sfn /* MIB SFN + radio-frame position */, sfn_offset, nof_prb, nof_ports, phich_length, phich_ng_x6, mib_bits; } o_mib_t;
int o_pbch_positions(const o_cell_t* cell, uint8_t* l, uint16_t* k);
void o_pbch_llr(const o_cell_t* cell, uint32_t nof_rx, const ocf_t* grid, const ocf_t* ce, float noise, float* llr);
int o_pbch_decode(const o_cell_t* cell, uint32_t nof_rx, const ocf_t* grid, const ocf_t* ce, float noise, o_mib_t* out, float* llr_out);
int o_mib_decode_subframe(const o_cell_t* cell, uint32_t nof_rx, const ocf_t* iq, o_mib_t* out, float* llr_out);

/* ---------- MAC DL-SCH walk + RRCConnectionSetup (o_rrc.c) ---------- */
typedef struct { uint32_t lcid, is_sdu, off, len; } o_mac_subh_t;
typedef struct { /* ltesniffer_ue_spec_config_t, MCSTracking.h:37-43 */
  uint32_t has_ue_config;
  float p_a;                                     /* dB */
  uint32_t i_offset_ack, i_offset_cqi, i_offset_ri; /* betaOffset-*-Index */
  uint32_t cqi_type;                             /* 0 wideband, 1 UE-selected sub-band, 2 higher-layer sub-band */
  uint32_t bits_used;                            /* bits of the DL-CCCH message consumed by the decoder (test aid) */
} o_ue_cfg_t;
int o_mac_dlsch_parse(const uint8_t* pdu, int len, o_mac_subh_t* out, int cap);
int o_rrc_conn_setup_decode(const uint8_t* sdu, int len, o_ue_cfg_t* out);
typedef struct { /* the SIB2 fields ULSchedule::set_config and SubframeWorker.cc:271-273 read */
  uint32_t n_sb, hopping_mode, pusch_hop_offset, enable_64qam;
  uint32_t group_hopping_enabled, group_assignment_pusch, sequence_hopping_enabled, cyclic_shift;
  uint32_t root_seq_idx, prach_config_idx, high_speed_flag, zero_corr_zone, prach_freq_offset;
  uint32_t bits_used; /* test aid */
} o_sib2_t;
int o_sib2_decode(const uint8_t* pdu, int len, o_sib2_t* out);
typedef struct { uint32_t is_imsi, nof_digits; uint8_t digits[24]; uint32_t mmec, m_tmsi; } o_paging_id_t;
int o_paging_decode(const uint8_t* pdu, int len, o_paging_id_t* out, int cap); /* PCCH-Message -> paging records, -1: does not unpack */
typedef struct { uint32_t tti; uint16_t rnti; uint32_t id_type /* 1 TMSI, 2 contention resolution, 3 IMSI */, msg_type /* 1 connection setup, 5 paging */; char value[24]; } o_api_event_t;
int o_api_ul_msg3_events(int api_mode, const uint8_t* pdu, int len, uint16_t rnti, uint32_t tti, o_api_event_t* ev, int cap, int* nev); /* decoded Msg3 -> initial UE identity */
int o_api_ul_dcch_events(int api_mode, const uint8_t* pdu, int len, uint16_t rnti, uint32_t tti, o_api_event_t* ev, int cap, int* nev); /* SRB messages: UE capability, attach request / identity response */
int o_api_dl_events(int api_mode, char name, const uint8_t* pdu, int len, uint16_t rnti, uint32_t tti, o_api_event_t* ev, int cap, int* nev);

/* ---------- PSS / SSS cell search (o_sync.c) ---------- */
typedef struct { uint32_t nof_periods; int32_t force_n_id_2 /* -1: all three roots */; float threshold /* peak / mean of the PSS correlation power */; } o_sync_cfg_t;
typedef struct {
  uint32_t found, cell_id, n_id_2, n_id_1;
  uint32_t sf_idx;   /* 0 or 5: index of the subframe that starts at sample sf_start */
  uint32_t pss_pos;  /* first sample of the PSS symbol's useful part, 0 <= pss_pos < 5 ms */
  uint32_t sf_start; /* 0 <= sf_start < 5 ms */
  float pss_peak, pss_p2avg, sss_metric, sss_second, cfo_hz, cfo_coarse_hz;
  uint32_t cp;       /* 0 normal, 1 extended cyclic prefix: which of the two SSS positions in front of the PSS symbol carried the better SSS */
} o_sync_t;
void o_pss_seq(uint32_t n_id_2, ocf_t* d);
void o_sss_m0m1(uint32_t n_id_1, uint32_t* m0, uint32_t* m1);
void o_sss_seq(uint32_t n_id_1, uint32_t n_id_2, int sf5, int8_t* d);
void o_pss_time(uint32_t n_id_2, uint32_t N, ocf_t* p);
uint32_t o_sync_min_samples(uint32_t nof_prb, uint32_t nof_periods);
int o_cell_search(const ocf_t* x, uint64_t nsamples, uint32_t nof_prb, const o_sync_cfg_t* cfg, o_sync_t* out, float* corr_out);

/* ---------- IQ capture file source (o_file.c) ---------- */
long o_file_read(const char* path, uint32_t nof_prb, uint32_t nant, long offset_time, float offset_freq, uint32_t first_sf, uint32_t nsf, ocf_t* out);
long o_file_read_fmt(const char* path, uint32_t nof_prb, uint32_t nant, long offset_time, float offset_freq, uint32_t first_sf, uint32_t nsf,
                     uint32_t fmt /* 0 cf32, 1 int16 pairs, 2 int8 pairs */, float scale /* one LSB; 0 = full scale +-1 */, ocf_t* out);

/* ---------- uplink: PRACH detection (o_prach.c) ---------- */
typedef struct {
  uint32_t config_idx, root_seq_idx, zero_corr_zone, freq_offset, hs_flag; /* SIB2 prach-ConfigInfo (ULSchedule.cc:149-154) */
  float detect_factor;                                                     /* 0 -> 60 (UL_Sniffer_PUSCH.cc:651) */
  const uint16_t* zc_roots;                                                /* 36.211 Table 5.7.2-4 (838 entries) or NULL */
} o_prach_cfg_t;
typedef struct { uint32_t preamble, offset; float offset_sec, p2avg; } o_prach_det_t;
uint32_t o_prach_ncs(uint32_t zero_corr_zone);
uint32_t o_prach_nof_roots(uint32_t zero_corr_zone);
int o_prach_tti_opportunity(uint32_t config_idx, uint32_t tti);
int o_prach_first_bin(uint32_t nof_prb, uint32_t freq_offset);
void o_prach_root_spectrum(uint32_t u, ocf_t* D);
int o_prach_detect(const o_cell_t* cell, const o_prach_cfg_t* cfg, const ocf_t* samples, o_prach_det_t* out, int cap, float* corr_out);

/* ---------- uplink: SC-FDMA demodulation + PUSCH (o_pusch.c) ---------- */
typedef struct { uint32_t cyclic_shift; /* SIB2 cyclicShift 0..7 */ uint32_t delta_ss; /* SIB2 groupAssignmentPUSCH 0..29 */
                 uint32_t hopping_offset; /* SIB2 pusch-HoppingOffset */
                 uint32_t group_hopping_enabled, sequence_hopping_enabled; /* SIB2 ul-ReferenceSignalsPUSCH (ULSchedule.cc:143-146) */ } o_ul_cfg_t;
typedef struct { uint32_t nof_ack; uint32_t cqi_bits; uint32_t ri_bits; /* HARQ-ACK bits 0..2, CQI report size (0 = none), RI bits */
                 uint32_t i_ack_p1, i_cqi_p1, i_ri_p1; /* 1 + betaOffset-ACK / -CQI / -RI-Index of the UE (uci_offset, UL_Sniffer_PUSCH.cc:435); 0 = the defaults 10 / 8 / 11 of MCSTracking.cc:1534-1538 */ } o_uci_t;
int o_uci_cqi_bits_type(uint32_t nof_prb, uint32_t cqi_type); /* srsran_cqi_size: 0 wideband (4), 1 UE-selected sub-band (4 + 1), 2 higher-layer sub-band (4 + 2 N) */
int o_uci_cqi_bits(uint32_t nof_prb);
int o_uci_layout(int M, int tbs, const o_uci_t* uci, uint8_t* cls, int* didx, int* q_ack, int* q_ri, int* q_cqi);
int o_uci_layout_cp(int M, int tbs, const o_uci_t* uci, int cp /* 1: extended CP - 10 columns, RI on 0 3 5 8, HARQ-ACK on 1 2 6 7 */, uint8_t* cls, int* didx, int* q_ack, int* q_ri, int* q_cqi);
int o_ul_valid_prb(uint32_t L);
void o_ul_shift_table(int N, ocf_t* t);
void o_ul_fft(const o_cell_t* cell, const ocf_t* in, ocf_t* grid);
int o_dmrs_base(uint32_t u, uint32_t v, int M_sc, ocf_t* r);
void o_dmrs_uv(const o_cell_t* cell, const o_ul_cfg_t* ul, uint32_t ns, int M_sc, uint32_t* u, uint32_t* v); /* 36.211 5.5.1.3 / 5.5.1.4 */
int o_dmrs_pusch(const o_cell_t* cell, const o_ul_cfg_t* ul, uint32_t ns, uint32_t n_dmrs_dci, int M_sc, ocf_t* r); /* the reference signal of slot ns as transmitted */
uint32_t o_dmrs_ncs(const o_cell_t* cell, const o_ul_cfg_t* ul, uint32_t ns, uint32_t n_dmrs_dci);
void o_idft_table(int M, ocf_t* w);
void o_idft_mixed(int M, const ocf_t* w, ocf_t* x, ocf_t* tmp); /* in-place (via tmp) mixed-radix IDFT, radices 4, 2, 3, 5; defines the operation order of the product */
int o_pusch_demod(const o_cell_t* cell, const o_ul_cfg_t* ul, uint32_t sf_idx, uint16_t rnti, const o_pusch_grant_t* g, uint32_t n_dmrs_dci,
                  const ocf_t* grid, int16_t* e, float* noise_out, float* sigpow_out);
int o_pusch_decode(const o_cell_t* cell, const o_ul_cfg_t* ul, uint32_t sf_idx, uint16_t rnti, const o_pusch_grant_t* g, uint32_t n_dmrs_dci,
                   const ocf_t* grid, int max_iter, uint8_t* payload, int* iters, float* snr_db);
int o_pusch_demod_uci(const o_cell_t* cell, const o_ul_cfg_t* ul, uint32_t sf_idx, uint16_t rnti, const o_pusch_grant_t* g, uint32_t n_dmrs_dci,
                      const o_uci_t* uci, const ocf_t* grid, int16_t* e, float* noise_out, float* sigpow_out);
int o_pusch_decode_uci(const o_cell_t* cell, const o_ul_cfg_t* ul, uint32_t sf_idx, uint16_t rnti, const o_pusch_grant_t* g, uint32_t n_dmrs_dci,
                       const o_uci_t* uci, const ocf_t* grid, int max_iter, uint8_t* payload, int* iters, float* snr_db);

/* ---------- pcap (o_pcap.c) ---------- */
typedef struct o_pcap o_pcap_t;
o_pcap_t* o_pcap_open_mem(void);
o_pcap_t* o_pcap_open_file(const char* path);
void o_pcap_write(o_pcap_t*, const uint8_t* pdu, uint32_t len, uint32_t tti, uint16_t rnti, uint8_t direction, uint8_t rnti_type, uint8_t crc_ok, uint32_t ts_sec, uint32_t ts_usec);
const uint8_t* o_pcap_mem(o_pcap_t*, size_t* len);
uint32_t o_pcap_nof_records(o_pcap_t*);
void o_pcap_close(o_pcap_t*);
enum { O_PCAP_NO_RNTI = 0, O_PCAP_P_RNTI = 1, O_PCAP_RA_RNTI = 2, O_PCAP_C_RNTI = 3, O_PCAP_SI_RNTI = 4 };

/* ---------- the per-subframe worker (o_worker.c) ---------- */
typedef struct o_worker o_worker_t;
typedef struct {
  uint32_t nof_decoded_locations, nof_cce, nof_missed_cce, nof_subframes, nof_subframe_collisions_dw,
      nof_subframe_collisions_up, nof_locations;
} o_stats_t;
typedef struct {
  o_cell_t cell; uint32_t nof_rx; uint32_t histogram_threshold; double split_ratio; int skip_secondary;
  int mcs_tracking_mode; /* 1 = on (default) */ int max_turbo_iter; int enable_shortcut;
} o_worker_cfg_t;
o_worker_t* o_worker_new(const o_worker_cfg_t* cfg);
void o_worker_free(o_worker_t*);
void o_worker_set_pcap(o_worker_t*, o_pcap_t*);
/* iq[rx] -> 15*N samples each; returns number of pcap records written for this subframe */
int o_worker_work(o_worker_t*, const ocf_t* const* iq, uint32_t sf_idx, uint32_t sfn, int update_meta_formats, float cfo_correct_hz);
/* UL_MODE (SubframeWorker.cc:184-199,236-345): iq[0] = downlink antenna, iq[1] = uplink antenna; worker created with nof_rx = 1.
 * The SIB2-derived DMRS configuration is given instead of parsed (ASN.1 is out of scope). */
typedef struct { uint32_t rapid, ta, hopping, riv, mcs, tpc, ul_delay, csi_req; uint16_t t_crnti; int grant_ok; o_pusch_grant_t grant; } o_rar_t;
int o_rar_parse(const o_cell_t* cell, const uint8_t* p, int len, o_rar_t* out, int cap); /* MAC RAR PDU -> RAR entries (o_worker.c) */
void o_worker_set_ul_mode(o_worker_t*, const o_ul_cfg_t* ul); /* ul == NULL: configure from the first SIB2 (decode_SIB) */
int o_worker_ul_config(o_worker_t* w, o_ul_cfg_t* ul, o_sib2_t* sib2); /* 0 not configured, 1 given, 2 learned from SIB2 */
int o_worker_work_ul(o_worker_t*, const ocf_t* dl_iq, const ocf_t* ul_iq, uint32_t sf_idx, uint32_t sfn, int update_meta_formats);
const o_stats_t* o_worker_stats(o_worker_t*);
void o_worker_ue_cfg(o_worker_t* w, uint16_t rnti, o_ue_cfg_t* out); /* MCSTracking::get_ue_config_rnti */
int o_rrc_reconfig_tmsi(const uint8_t* sdu, int len, uint32_t* m_tmsi); /* decode_rrc_connection_reconfig: M-TMSI of the attach accept's GUTI */
void o_worker_set_api(o_worker_t* w, int api_mode, o_pcap_t* api_pcap); /* -a: -1 off, 0 identity mapping, 2 IMSI catching, 3 all */
int o_worker_api_events(o_worker_t* w, o_api_event_t* out, int cap);     /* events reported so far (print_api_dl) */
void o_worker_set_second_opinion(o_worker_t* w, int turbo, int viterbi); /* decode transport blocks / DCI candidates with o_second.c */
void o_worker_set_mcs_update_interval(o_worker_t* w, uint32_t seconds); /* MCSTracking::interval (5 s): ageing every interval x 1000 subframes, 0 = never */
uint32_t o_worker_nof_tracked_ul(o_worker_t* w);                      /* MCSTracking::nof_RNTI_member_ul */
int o_worker_tracked_mod_ul(o_worker_t* w, uint16_t rnti);            /* 0 no entry, 1 unknown, 2 / 3 / 4 = 16 / 64 / 256QAM max */
uint32_t o_worker_nof_tracked(o_worker_t* w);                         /* MCSTracking::nof_RNTI_member_dl */
int o_worker_tracked_table(o_worker_t* w, uint16_t rnti);            /* tracked table of an RNTI, -1 without entry */
/* stage taps for parity tests (valid until the next work()) */
const ocf_t* o_worker_grid(o_worker_t*);
const ocf_t* o_worker_ce(o_worker_t*);
const float* o_worker_llr(o_worker_t*, uint32_t* n);
const o_chest_res_t* o_worker_chest(o_worker_t*);
uint32_t o_worker_cfi(o_worker_t*);
const float* o_worker_rb_power(o_worker_t*); /* SubframePower::computePower of antenna 0, dB per PRB */
/* accepted DCIs of the last subframe, flat: {rnti, format, L, ncce, nof_bits, histval} x n */
uint32_t o_worker_accepted(o_worker_t*, uint32_t* out6, uint32_t max);
o_rntiman_t* o_worker_rntiman(o_worker_t*);
uint64_t o_worker_total_iters(o_worker_t*);
uint64_t o_worker_algo_bytes(o_worker_t*);
/* probes for tests/test_ref_collect.py: DCICollection::addCandidate on given DCI bits (flat words: 64 per downlink entry, 32 per uplink entry - the layout of
 * oracle/ref_shim_search/collect_glue.cc) and what the decoders feed back between subframes */
void o_worker_collect_begin(o_worker_t* w, uint32_t sfn, uint32_t sf_idx, uint32_t cfi);
void o_worker_collect_add(o_worker_t* w, uint16_t rnti, int format, uint32_t L, uint32_t ncce, uint32_t histval, const uint8_t* payload, uint32_t nof_bits);
uint32_t o_worker_collect_end(o_worker_t* w, uint32_t* dl, uint32_t dl_cap, uint32_t* ul, uint32_t ul_cap, uint16_t* map_dl, uint16_t* map_ul, uint32_t* counts2);
void o_worker_collect_mcs_update(o_worker_t* w, uint16_t rnti, int table);
void o_worker_collect_harq_update(o_worker_t* w, uint16_t rnti, int pid, int tid, uint32_t sfn, uint32_t sf_idx, int decoded, int ndi, int rv, int tbs);
void o_worker_collect_set_hop_offset(o_worker_t* w, uint32_t n_rb_ho);
/* probes for tests/test_ref_ul_decode.py: PUSCH_Decoder::decode on a given schedule, every attempt answered by a scripted uplink decoder (o_worker.c) */
void o_worker_set_ul_script(o_worker_t* w, int (*fn)(void* user, const uint32_t* call16, float* snr_db, uint8_t* payload), void* user);
void o_worker_set_last_ul_snr(o_worker_t* w, float snr_db);
void o_worker_ul_decode_probe(o_worker_t* w, uint32_t tti, uint32_t n, const uint32_t* e12);
void o_worker_ul_update_database(o_worker_t* w);
void o_worker_ul_set_ue_config(o_worker_t* w, uint16_t rnti, uint32_t i_ack, uint32_t i_cqi, uint32_t i_ri, uint32_t cqi_type);
/* probes for tests/test_ref_decode.py: decode_dl_mode on the collected entries, every decode call answered by a scripted decoder (call16 = {tti, rnti, nof_re, tx_scheme,
 * pmi, nof_layers, per block: enabled, modulation bits, tbs, rv, cw_idx}; it writes the payloads and two CRC verdicts) */
void o_worker_set_script_decoder(o_worker_t* w, int (*fn)(void* user, const uint32_t* call16, float p_a, uint8_t* payload0, uint8_t* payload1, int32_t* crc2), void* user);
void o_worker_collect_set_now(o_worker_t* w, uint32_t subframes);
void o_worker_collect_decode_dl_mode(o_worker_t* w);
int o_worker_collect_find_table(o_worker_t* w, uint16_t rnti);
void o_worker_collect_update_database(o_worker_t* w);

#ifdef __cplusplus
}
#endif
#endif
