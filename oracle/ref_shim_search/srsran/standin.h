/* oracle/_ref build stand-in for the srsRAN 4G API (test infrastructure, NOT product, NOT a copy of srsRAN).
 *
 * Purpose: compile the REFERENCE'S OWN blind-search code from where it lies - /root/reference/src/src/DCISearch.cc (the FALCON decision tree,
 * SURVEY 8 row a8), lib/src/phy/falcon_phch/falcon_pdcch.c (location map a5, CCE power a6), src/src/MetaFormats.cc (a19), lib/src/util/
 * RNTIManager.cc (a10) - and run it on the oracle's PDCCH soft bits (oracle/Makefile.ref: libref_falcon_search.so, tests/test_ref_dci_search.py).
 * srsRAN is an un-vendored dependency that is absent from /root/reference; the reference's headers name about forty of its types.  This
 * file DECLARES those types - only the members the compiled reference files or their headers touch, laid out freely (every translation unit
 * of the harness sees the same stand-in, so layout is irrelevant) - and the handful of macros they use.  It implements nothing: the srsRAN
 * FUNCTIONS the compiled files call are supplied by search_glue.cc next to this file, each either bound to the oracle's primitive
 * (format size, tail-biting Viterbi + CRC) or written from TS 36.213 9.1.1 where the reference itself shows the same loops
 * (falcon_pdcch.c:49-103).  Nothing of this reaches the product or the oracle proper. */
#pragma once
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include <sys/time.h>

typedef __complex__ float cf_t; /* GNU spelling: valid in C and inside the extern "C" blocks from which the reference's C++ headers include this */
#ifdef __cplusplus
extern "C" {
#endif

#define SRSRAN_API
#define SRSRAN_SUCCESS 0
#define SRSRAN_ERROR -1
#define SRSRAN_ERROR_INVALID_INPUTS -2
#define SRSRAN_MAX_PORTS 4
#define SRSRAN_MAX_CODEWORDS 2
#define SRSRAN_MAX_TB SRSRAN_MAX_CODEWORDS
#define SRSRAN_MAX_LAYERS 4
#define SRSRAN_MAX_PRB 110
#define SRSRAN_DCI_MAX_BITS 128
#define SRSRAN_RAR_GRANT_LEN 20
#define SRSRAN_NOF_SF_X_FRAME 10
#define SRSRAN_NRE 12
#define SRSRAN_MIN(a, b) ((a) < (b) ? (a) : (b))
#define SRSRAN_MAX(a, b) ((a) > (b) ? (a) : (b))
/* RNTI intervals: the values of srsRAN's phy_common.h (RA-RNTI 1..10: FDD, 1 + t_id; TS 36.321 Table 7.1-1 reserves up to 0x3C for TDD) - the
 * same values the oracle (lsn_oracle.h) and the product (lsn_lte.h) carry.  A first version of this file had 0x003C here: the long runs of
 * tests/golden/make_dci_search_fixture.py caught it in subframe 2160 of the first stream, where a spurious CRC match on RNTI 57 is an evergreen
 * RA-RNTI under one value and an unknown C-RNTI under the other. */
#define SRSRAN_RARNTI_START 0x0001
#define SRSRAN_RARNTI_END 0x000A
#define SRSRAN_CRNTI_START 0x000B
#define SRSRAN_CRNTI_END 0xFFF3
#define SRSRAN_MRNTI 0xFFFD
#define SRSRAN_PRNTI 0xFFFE
#define SRSRAN_SIRNTI 0xFFFF
#define SRSRAN_RNTI_ISRAR(rnti) ((rnti) >= SRSRAN_RARNTI_START && (rnti) <= SRSRAN_RARNTI_END)

/* log macros: silent */
#define DEBUG(...) do { } while (0)
#define INFO(...) do { } while (0)
#ifdef LSN_REF_QUIET /* the collection harness sweeps invalid grants by the thousand */
#define ERROR(...) do { } while (0)
#else
#define ERROR(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } while (0)
#endif
#define SRSRAN_VERBOSE_ISINFO() (0)
#define SRSRAN_VERBOSE_ISDEBUG() (0)

typedef enum { SRSRAN_CP_NORM = 0, SRSRAN_CP_EXT } srsran_cp_t;
typedef enum { SRSRAN_SF_NORM = 0, SRSRAN_SF_MBSFN } srsran_sf_t;
typedef enum { SRSRAN_MOD_BPSK = 0, SRSRAN_MOD_QPSK, SRSRAN_MOD_16QAM, SRSRAN_MOD_64QAM, SRSRAN_MOD_256QAM } srsran_mod_t;
typedef struct { uint32_t nof_prb, nof_ports, id; srsran_cp_t cp; int phich_length, phich_resources, frame_type; } srsran_cell_t;

typedef enum {
  SRSRAN_DCI_FORMAT0 = 0, SRSRAN_DCI_FORMAT1, SRSRAN_DCI_FORMAT1A, SRSRAN_DCI_FORMAT1B, SRSRAN_DCI_FORMAT1C, SRSRAN_DCI_FORMAT1D,
  SRSRAN_DCI_FORMAT2, SRSRAN_DCI_FORMAT2A, SRSRAN_DCI_FORMAT2B, SRSRAN_DCI_FORMATN0, SRSRAN_DCI_FORMATN1, SRSRAN_DCI_FORMATN2,
  SRSRAN_DCI_FORMAT_RAR, SRSRAN_DCI_NOF_FORMATS
} srsran_dci_format_t;
typedef struct { uint32_t L; /* log2 of the aggregation level */ uint32_t ncce; } srsran_dci_location_t;
typedef struct { bool multiple_csi_request_enabled, cif_enabled, cif_present, srs_request_enabled, ra_format_enabled, is_not_ue_ss; } srsran_dci_cfg_t;
typedef struct {
  uint8_t payload[SRSRAN_DCI_MAX_BITS];
  uint32_t nof_bits;
  srsran_dci_location_t location;
  srsran_dci_format_t format;
  uint16_t rnti;
} srsran_dci_msg_t;
typedef struct { uint32_t sf_config, tdd_special_sf; bool configured; } srsran_tdd_config_t;
typedef struct { uint32_t tti; uint32_t cfi; srsran_sf_t sf_type; uint32_t non_mbsfn_region; srsran_tdd_config_t tdd_config; } srsran_dl_sf_cfg_t;
typedef struct { uint32_t tti; bool shortened; srsran_tdd_config_t tdd_config; } srsran_ul_sf_cfg_t;

/* names the grant conversions use (ul_sniffer_pusch.c, dl_sniffer_pdsch.c).  Hopping: the values the reference itself restates in falcon_define.h
 * (FALCON_RA_PUSCH_HOP_*); transmission schemes and n_prb1a: symbolic in the compiled files, the harness maps them by name */
enum { SRSRAN_RA_PUSCH_HOP_DISABLED = -1, SRSRAN_RA_PUSCH_HOP_QUART = 0, SRSRAN_RA_PUSCH_HOP_QUART_NEG = 1, SRSRAN_RA_PUSCH_HOP_HALF = 2, SRSRAN_RA_PUSCH_HOP_TYPE2 = 3 };
typedef enum { SRSRAN_TXSCHEME_PORT0 = 0, SRSRAN_TXSCHEME_DIVERSITY, SRSRAN_TXSCHEME_SPATIALMUX, SRSRAN_TXSCHEME_CDD } srsran_tx_scheme_t;
enum { SRSRAN_RA_TYPE2_NPRB1A_2 = 0, SRSRAN_RA_TYPE2_NPRB1A_3 = 1 };
#define SRSRAN_RNTI_ISUSER(rnti) ((rnti) >= SRSRAN_CRNTI_START && (rnti) <= SRSRAN_CRNTI_END)
#define SRSRAN_DCI_IS_TB_EN(tb) (!((tb).mcs_idx == 0 && (tb).rv == 1)) /* 36.212 5.3.3.1.5: I_MCS = 0 and rv = 1 disables a transport block */
/* resource allocation (only named by the reference's headers) */
typedef enum { SRSRAN_RA_ALLOC_TYPE0 = 0, SRSRAN_RA_ALLOC_TYPE1, SRSRAN_RA_ALLOC_TYPE2 } srsran_ra_type_t;
typedef struct { uint32_t rbg_bitmask; } srsran_ra_type0_t;
typedef struct { uint32_t vrb_bitmask, rbg_subset; bool shift; } srsran_ra_type1_t;
typedef struct { uint32_t riv; int n_prb1a, n_gap, mode; } srsran_ra_type2_t;
typedef struct { srsran_mod_t mod; int tbs, rv; uint32_t nof_bits, cw_idx; bool enabled; uint32_t mcs_idx; } srsran_ra_tb_t;
typedef struct {
  uint16_t rnti; srsran_dci_format_t format; srsran_dci_location_t location; uint32_t ue_cc_idx;
  srsran_ra_type_t alloc_type; srsran_ra_type0_t type0_alloc; srsran_ra_type1_t type1_alloc; srsran_ra_type2_t type2_alloc;
  struct { uint32_t mcs_idx; int rv; bool ndi; uint32_t cw_idx; } tb[SRSRAN_MAX_CODEWORDS];
  bool tb_cw_swap; uint32_t pinfo, pid, dai; bool is_tdd, is_dwpts, sram_id, pconf, power_offset; uint8_t tpc_pucch;
  bool is_ra_order; uint32_t ra_preamble, ra_mask_idx; uint32_t cif; bool cif_present, srs_request, srs_request_present;
} srsran_dci_dl_t;
typedef struct {
  uint16_t rnti; srsran_dci_format_t format; srsran_dci_location_t location; uint32_t ue_cc_idx;
  srsran_ra_type2_t type2_alloc; int freq_hop_fl; struct { uint32_t mcs_idx; int rv; bool ndi; } tb;
  uint32_t n_dmrs; bool cqi_request; uint32_t dai, ul_idx; bool is_tdd; uint8_t tpc_pusch; uint32_t cif; bool cif_present;
  uint8_t multiple_csi_request; bool multiple_csi_request_present, srs_request, srs_request_present; int ra_type; bool ra_type_present;
} srsran_dci_ul_t;
typedef struct { uint32_t rba, trunc_mcs; uint16_t tpc_pusch; bool ul_delay, cqi_request, hopping_flag; } srsran_dci_rar_grant_t;
typedef struct {
  int tx_scheme; uint32_t pmi; bool prb_idx[2][SRSRAN_MAX_PRB]; uint32_t nof_prb, nof_re;
  uint32_t nof_symb_slot[2]; srsran_ra_tb_t tb[SRSRAN_MAX_CODEWORDS]; int last_tbs[SRSRAN_MAX_CODEWORDS]; uint32_t nof_tb, nof_layers;
} srsran_pdsch_grant_t;
typedef struct {
  uint32_t n_prb[2], n_prb_tilde[2], L_prb, freq_hopping, nof_re, nof_symb; srsran_ra_tb_t tb; srsran_ra_tb_t last_tb; uint32_t n_dmrs; bool is_rar;
} srsran_pusch_grant_t;
typedef struct {
  enum { SRSRAN_PUSCH_HOP_MODE_INTER_SF = 1, SRSRAN_PUSCH_HOP_MODE_INTRA_SF = 0 } hop_mode;
  uint32_t hopping_offset, n_sb, n_rb_ho, current_tx_nb; bool hopping_enabled;
} srsran_pusch_hopping_cfg_t;
typedef struct { uint32_t I_offset_cqi, I_offset_ri, I_offset_ack; } srsran_uci_offset_cfg_t;
typedef enum { SRSRAN_CQI_TYPE_WIDEBAND = 0, SRSRAN_CQI_TYPE_SUBBAND_UE, SRSRAN_CQI_TYPE_SUBBAND_HL } srsran_cqi_type_t;
typedef struct { bool data_enable, pmi_present, four_antenna_ports, rank_is_not_one, subband_label_2_bits; uint32_t scell_index, L, N; srsran_cqi_type_t type; uint32_t ri_len; } srsran_cqi_cfg_t;
typedef struct { uint32_t n_prb_lowest, n_dmrs, I_phich; } srsran_phich_grant_t;
typedef struct { uint32_t max_cb; int16_t** buffer_f; uint8_t** data; bool* cb_crc; bool tb_crc; } srsran_softbuffer_rx_t;
typedef struct { uint32_t cyclic_shift, cyclic_shift_for_dmrs, delta_ss; bool group_hopping_en, sequence_hopping_en; } srsran_refsignal_dmrs_pusch_cfg_t;
typedef struct { bool is_nr; uint32_t config_idx, root_seq_idx, zero_corr_zone, freq_offset, num_ra_preambles; bool hs_flag; int tdd_config[2]; bool enable_successive_cancellation, enable_freq_domain_offset_calc; } srsran_prach_cfg_t;

/* what DCISearch.cc / falcon_pdcch.c reach through the UE object: pdcch.{nof_cce, nof_regs, llr, max_bits, rm_f}, cell, chest_res.snr_db,
 * sf_symbols[0], nof_rx_antennas */
typedef struct { int _; } srsran_viterbi_t;
typedef struct { int _; } srsran_crc_t;
typedef struct {
  srsran_cell_t cell; uint32_t nof_regs[3]; uint32_t nof_cce[3]; uint32_t max_bits; uint32_t nof_rx_antennas;
  float rm_f[3 * (SRSRAN_DCI_MAX_BITS + 16)]; float* llr; srsran_viterbi_t decoder; srsran_crc_t crc;
} srsran_pdcch_t;
typedef struct { float noise_estimate, noise_estimate_dbm, snr_db, snr_ant_port_db[4][4], rsrp, rsrp_dbm, rsrq, rsrq_db, cfo, sync_error; } srsran_chest_dl_res_t;
typedef struct { srsran_dci_cfg_t dci; } srsran_ue_dl_cfg_inner_t;
typedef struct { srsran_ue_dl_cfg_inner_t cfg; } srsran_ue_dl_cfg_t;
typedef struct {
  srsran_cell_t cell; uint32_t nof_rx_antennas; srsran_pdcch_t pdcch; srsran_chest_dl_res_t chest_res; cf_t* sf_symbols[SRSRAN_MAX_PORTS];
} srsran_ue_dl_t;

/* what PDSCH_Decoder hands to the PDSCH decoder and gets back (DL_Sniffer_PDSCH.cc): the members it touches */
typedef struct { srsran_softbuffer_rx_t* rx[SRSRAN_MAX_CODEWORDS]; } srsran_pdsch_softbuffers_standin_t;
typedef struct {
  srsran_pdsch_grant_t grant; uint16_t rnti; srsran_pdsch_softbuffers_standin_t softbuffers; float p_a; bool use_tbs_index_alt, power_scale, csi_enable, meas_evm_en, meas_time_en;
  uint32_t max_nof_iterations; int decoder_type;
} srsran_pdsch_cfg_t;
typedef struct { uint8_t* payload; bool crc; float avg_iterations_block, evm; uint32_t ack_value; } srsran_pdsch_res_t;
#define ZERO_OBJECT(x) memset(&(x), 0x0, sizeof((x)))
uint8_t* srsran_vec_u8_malloc(uint32_t len);
void srsran_vec_u8_zero(uint8_t* ptr, uint32_t nsamples);
void srsran_softbuffer_rx_reset_tbs(srsran_softbuffer_rx_t* q, uint32_t tbs);
int srsran_ue_dl_decode_pdsch(srsran_ue_dl_t* q, srsran_dl_sf_cfg_t* sf, srsran_pdsch_cfg_t* pdsch_cfg, srsran_pdsch_res_t data[SRSRAN_MAX_CODEWORDS]);

/* functions the compiled reference files call (definitions: search_glue.cc) */
uint32_t srsran_dci_format_sizeof(const srsran_cell_t* cell, srsran_dl_sf_cfg_t* sf, srsran_dci_cfg_t* cfg, srsran_dci_format_t format);
const char* srsran_dci_format_string(srsran_dci_format_t format);
int srsran_pdcch_dci_decode(srsran_pdcch_t* q, float* e, uint8_t* data, uint32_t E, uint32_t nof_bits, uint16_t* crc);
void srsran_pdcch_dci_encode_conv(srsran_pdcch_t* q, uint8_t* data, uint32_t nof_bits, uint8_t* coded_data, uint16_t rnti);
uint32_t srsran_pdcch_common_locations_ncce(uint32_t nof_cce, srsran_dci_location_t* c, uint32_t max_candidates);
uint32_t srsran_pdcch_ue_locations_ncce(uint32_t nof_cce, srsran_dci_location_t* c, uint32_t max_candidates, uint32_t sf_idx, uint16_t rnti);
int srsran_ue_dl_decode_fft_estimate(srsran_ue_dl_t* q, srsran_dl_sf_cfg_t* sf, srsran_ue_dl_cfg_t* cfg);
int srsran_rm_conv_rx(float* input, uint32_t in_len, float* output, uint32_t out_len);
int srsran_rm_conv_tx(uint8_t* input, uint32_t in_len, uint8_t* output, uint32_t out_len);
int srsran_viterbi_decode_f(srsran_viterbi_t* q, float* symbols, uint8_t* data, uint32_t frame_length);
uint32_t srsran_crc_checksum(srsran_crc_t* h, uint8_t* data, int len);
uint32_t srsran_bit_pack(uint8_t** bits, int nof_bits);
void srsran_bit_fprint(FILE* stream, uint8_t* bits, int nof_bits);
void srsran_bit_unpack(uint32_t value, uint8_t** bits, int nof_bits);
uint32_t srsran_mod_bits_x_symbol(srsran_mod_t mod);
int srsran_dci_msg_unpack_pdsch(srsran_cell_t* cell, srsran_dl_sf_cfg_t* sf, srsran_dci_cfg_t* cfg, srsran_dci_msg_t* msg, srsran_dci_dl_t* dci);
int srsran_dci_msg_unpack_pusch(srsran_cell_t* cell, srsran_dl_sf_cfg_t* sf, srsran_dci_cfg_t* cfg, srsran_dci_msg_t* msg, srsran_dci_ul_t* dci);
int srsran_ra_ul_dci_to_grant(srsran_cell_t* cell, srsran_ul_sf_cfg_t* sf, srsran_pusch_hopping_cfg_t* hopping_cfg, srsran_dci_ul_t* dci, srsran_pusch_grant_t* grant);
int srsran_softbuffer_rx_init(srsran_softbuffer_rx_t* q, uint32_t nof_prb);
void srsran_softbuffer_rx_free(srsran_softbuffer_rx_t* q);
void srsran_softbuffer_rx_reset(srsran_softbuffer_rx_t* q);
#ifdef __cplusplus
}
#endif
