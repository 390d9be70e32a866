/* oracle/_ref build stand-in for the srsRAN uplink receiver types, the UL RRC / S1AP / NAS message classes and the RLC helpers that the reference's
 * src/src/UL_Sniffer_PUSCH.cc names (test infrastructure, NOT product, NOT a copy of srsRAN; see standin.h, standin_l2.h).
 *
 * Purpose: compile the REFERENCE'S OWN uplink decode control flow from where it lies - PUSCH_Decoder::decode (UL_Sniffer_PUSCH.cc:389-583: which uplink MCS table
 * is tried in which order for which tracked modulation, 16QAM -> 64QAM -> 256QAM for an unknown one), decode_run (:248-385: what a CRC-ok block writes and what it
 * teaches the tracking database), investigate_valid_ul_grant / check_valid_prb_ul (:894-918) - and drive it with scripted decoder verdicts next to the oracle
 * (oracle/Makefile.ref: libref_falcon_ul_decode.so, tests/test_ref_ul_decode.py).
 * The receiver structs have the MEMBERS that file touches and nothing else.  The message classes of the identity API (api_mode >= 0: RRCConnectionRequest,
 * UL-DCCH, NAS) DO NOT PARSE here - every unpack reports failure - so the harness runs with api_mode -1 only; that part of the reference (:47-247, 306-381) is
 * pinned on the reference's own captures instead (tests/test_rrc_oracle.py, tests/test_api_sink.py). */
#pragma once
#include "srsran/standin.h"
#include "srsran/standin_l2.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { bool ack_value[8]; bool scheduling_request; struct { bool data_crc; struct { uint32_t wideband_cqi, pmi, spatial_diff_cqi; } wideband; struct { uint32_t subband_label, subband_cqi; } subband_ue; struct { uint32_t wideband_cqi_cw0, wideband_cqi_cw1; } subband_hl; } cqi; uint8_t ri; } srsran_uci_value_t;
typedef struct { srsran_uci_value_t uci; } srsran_uci_value_holder_standin_t;
typedef struct { uint8_t* data; bool crc; srsran_uci_value_t uci; float avg_iterations_block, evm; float epre_dbfs; } srsran_pusch_res_t;
typedef struct { float snr_db, noise_estimate, noise_estimate_dbm, ta_us, epre_dBfs, rsrp_dBfs, cfo_hz; cf_t* ce; } srsran_chest_ul_res_t;
typedef struct { int _; } srsran_chest_ul_t;
typedef struct { int _; } srsran_pusch_t;
typedef struct { uint32_t nof_acks; } srsran_uci_cfg_ack_t;
typedef struct { srsran_uci_cfg_ack_t ack[5]; srsran_cqi_cfg_t cqi; bool is_scheduling_request_tti; } srsran_uci_cfg_t;
typedef struct { srsran_softbuffer_rx_t* rx; } srsran_pusch_softbuffers_standin_t;
typedef struct {
  uint16_t rnti; srsran_uci_cfg_t uci_cfg; srsran_uci_offset_cfg_t uci_offset; srsran_pusch_grant_t grant; uint32_t max_nof_iterations, last_O_cqi, K_segm, current_tx_nb;
  bool csi_enable, enable_64qam, meas_time_en, meas_epre_en, meas_ta_en, meas_evm_en; uint32_t meas_time_value; srsran_pusch_softbuffers_standin_t softbuffers;
} srsran_pusch_cfg_t;
typedef struct { srsran_refsignal_dmrs_pusch_cfg_t dmrs; srsran_pusch_hopping_cfg_t hopping; srsran_pusch_cfg_t pusch; } srsran_ul_cfg_t;
typedef struct { srsran_cell_t cell; cf_t* in_buffer; cf_t* sf_symbols; srsran_chest_ul_res_t chest_res; srsran_chest_ul_t chest; srsran_pusch_t pusch; } srsran_enb_ul_t;
typedef struct { uint32_t N_cp; float T_tot; int _; } srsran_prach_t;
void srsran_enb_ul_fft(srsran_enb_ul_t* q);
int srsran_chest_ul_estimate_pusch(srsran_chest_ul_t* q, srsran_ul_sf_cfg_t* sf, srsran_pusch_cfg_t* cfg, cf_t* input, srsran_chest_ul_res_t* res);
int srsran_pusch_decode(srsran_pusch_t* q, srsran_ul_sf_cfg_t* sf, srsran_pusch_cfg_t* cfg, srsran_chest_ul_res_t* channel, cf_t* sf_symbols, srsran_pusch_res_t* data);
uint8_t* srsran_vec_u8_malloc(uint32_t n);
void srsran_vec_u8_zero(uint8_t* p, uint32_t n);
int srsran_symbol_sz(uint32_t nof_prb);
uint32_t srsran_ri_nof_bits(const srsran_cell_t* cell);
int srsran_prach_init(srsran_prach_t* p, uint32_t max_N_ifft_ul);
int srsran_prach_set_cfg(srsran_prach_t* p, srsran_prach_cfg_t* cfg, uint32_t nof_prb);
void srsran_prach_set_detect_factor(srsran_prach_t* p, float factor);
bool srsran_prach_tti_opportunity(srsran_prach_t* p, uint32_t current_tti, int allowed_subframe);
int srsran_prach_detect_offset(srsran_prach_t* p, uint32_t freq_offset, cf_t* signal, uint32_t sig_len, uint32_t* indices, float* t_offsets, float* peak_to_avg, uint32_t* ind_len);
#ifndef SRSRAN_SF_LEN_PRB
#define SRSRAN_SF_LEN_PRB(nof_prb) (15 * srsran_symbol_sz(nof_prb))
#endif
#ifdef __cplusplus
}
#include <memory>
#include <sstream>
namespace srsran {
struct byte_buffer_t { uint32_t N_bytes = 0; uint8_t buffer[8192]; uint8_t* msg = buffer; };
typedef std::unique_ptr<byte_buffer_t> unique_byte_buffer_t;
inline unique_byte_buffer_t make_byte_buffer() { return unique_byte_buffer_t(new byte_buffer_t()); }
struct rlc_amd_pdu_header_t { uint8_t dc = 0, rf = 0, p = 0, fi = 0; uint16_t sn = 0; uint8_t lsf = 0; uint16_t so = 0; uint32_t N_li = 0; uint16_t li[16] = {}; };
bool rlc_am_is_control_pdu(uint8_t* payload);
void rlc_am_read_data_pdu_header(uint8_t** payload, uint32_t* nof_bytes, rlc_amd_pdu_header_t* header);
}  // namespace srsran
/* NAS (liblte_mme.h), uplink side: declared, never reached with api_mode -1 */
#define LIBLTE_MME_SECURITY_HDR_TYPE_PLAIN_NAS 0x0
#define LIBLTE_MME_SECURITY_HDR_TYPE_INTEGRITY 0x1
#define LIBLTE_MME_SECURITY_HDR_TYPE_INTEGRITY_AND_CIPHERED 0x2
#define LIBLTE_MME_SECURITY_HDR_TYPE_INTEGRITY_WITH_NEW_EPS_SECURITY_CONTEXT 0x3
#define LIBLTE_MME_SECURITY_HDR_TYPE_INTEGRITY_AND_CIPHERED_WITH_NEW_EPS_SECURITY_CONTEXT 0x4
#define LIBLTE_MME_MSG_TYPE_ATTACH_REQUEST 0x41
#define LIBLTE_MME_MSG_TYPE_IDENTITY_RESPONSE 0x56
#define LIBLTE_MME_MOBILE_ID_TYPE_IMSI 0x1
#define LIBLTE_MME_MOBILE_ID_TYPE_IMEI 0x2
#define LIBLTE_MME_MOBILE_ID_TYPE_IMEISV 0x3
#define LIBLTE_MME_EPS_MOBILE_ID_TYPE_IMSI 0x1
#define LIBLTE_MME_EPS_MOBILE_ID_TYPE_GUTI 0x6
#define LIBLTE_MME_EPS_MOBILE_ID_TYPE_IMEI 0x3
struct LIBLTE_MME_MOBILE_ID_STRUCT { uint8 type_of_id; uint8 imsi[15], imei[15], imeisv[16]; };
struct LIBLTE_MME_ID_RESPONSE_MSG_STRUCT { LIBLTE_MME_MOBILE_ID_STRUCT mobile_id; };
struct LIBLTE_MME_EPS_MOBILE_ID_STRUCT { uint8 type_of_id; uint8 imsi[15], imei[15]; struct { uint32_t m_tmsi; } guti; };
struct LIBLTE_MME_ATTACH_REQUEST_MSG_STRUCT { LIBLTE_MME_EPS_MOBILE_ID_STRUCT eps_mobile_id; };
LIBLTE_ERROR_ENUM liblte_mme_parse_msg_sec_header(LIBLTE_BYTE_MSG_STRUCT* msg, uint8* pd, uint8* sec_hdr_type);
LIBLTE_ERROR_ENUM liblte_mme_unpack_identity_response_msg(LIBLTE_BYTE_MSG_STRUCT* msg, LIBLTE_MME_ID_RESPONSE_MSG_STRUCT* id_resp);
LIBLTE_ERROR_ENUM liblte_mme_unpack_attach_request_msg(LIBLTE_BYTE_MSG_STRUCT* msg, LIBLTE_MME_ATTACH_REQUEST_MSG_STRUCT* attach_req);
namespace asn1 {
namespace s1ap { struct s1ap_pdu_c { SRSASN_CODE unpack(cbit_ref&) { return SRSASN_ERROR_DECODE_FAIL; } }; }
namespace rrc {
struct bitstring_standin { uint64_t v = 0; uint32_t nbits = 0; uint64_t to_number() const { return v; } std::string to_string() const { std::string s; for (uint32_t i = 0; i < nbits; i++) s.push_back(((v >> (nbits - 1 - i)) & 1) ? '1' : '0'); return s; } };
struct init_ue_id_c {
  struct types { enum options { s_tmsi, random_value, nulltype }; };
  struct s_tmsi_t { bitstring_standin mmec, m_tmsi; };
  types::options type() const { return t; }
  s_tmsi_t& s_tmsi() { return st; }
  bitstring_standin& random_value() { return rv; }
  types::options t = types::nulltype; s_tmsi_t st; bitstring_standin rv;
};
struct rrc_conn_request_r8_ies_s { init_ue_id_c ue_id; };
struct rrc_conn_request_s { struct crit_exts_t { rrc_conn_request_r8_ies_s r8; rrc_conn_request_r8_ies_s& rrc_conn_request_r8() { return r8; } } crit_exts; };
struct ul_ccch_msg_type_c {
  struct types_opts { enum options { c1, msg_class_ext, nulltype }; };
  struct c1_c_ {
    struct types { enum options { rrc_conn_reest_request, rrc_conn_request, nulltype }; };
    typed<types::options> type() const { return typed<types::options>{t}; }
    rrc_conn_request_s& rrc_conn_request() { return req; }
    types::options t = types::nulltype; rrc_conn_request_s req;
  };
  types_opts::options type() const { return t; }
  c1_c_& c1() { return c1_; }
  types_opts::options t = types_opts::nulltype; c1_c_ c1_;
};
struct ul_ccch_msg_s { ul_ccch_msg_type_c msg; SRSASN_CODE unpack(cbit_ref&) { return SRSASN_ERROR_DECODE_FAIL; } };
struct rrc_conn_setup_complete_r8_ies_s { nas_pdu_standin ded_info_nas; };
struct rrc_conn_setup_complete_s {
  struct c1_t { rrc_conn_setup_complete_r8_ies_s r8; rrc_conn_setup_complete_r8_ies_s& rrc_conn_setup_complete_r8() { return r8; } };
  struct crit_exts_t { c1_t c1_; c1_t& c1() { return c1_; } } crit_exts;
};
struct ul_info_transfer_s {
  struct ded_info_type_t { nas_pdu_standin n; nas_pdu_standin& ded_info_nas() { return n; } };
  struct r8_t { ded_info_type_t ded_info_type; };
  struct c1_t { r8_t r8; r8_t& ul_info_transfer_r8() { return r8; } };
  struct crit_exts_t { c1_t c1_; c1_t& c1() { return c1_; } } crit_exts;
};
struct ul_dcch_msg_type_c {
  struct types_opts { enum options { c1, msg_class_ext, nulltype }; };
  struct c1_c_ {
    struct types { enum options { csfb_params_request_cdma2000, meas_report, rrc_conn_recfg_complete, rrc_conn_reest_complete, rrc_conn_setup_complete, security_mode_complete, security_mode_fail, ue_cap_info, ul_ho_prep_transfer, ul_info_transfer, nulltype }; };
    types::options type() const { return t; }
    rrc_conn_setup_complete_s& rrc_conn_setup_complete() { return sc; }
    ul_info_transfer_s& ul_info_transfer() { return it; }
    types::options t = types::nulltype; rrc_conn_setup_complete_s sc; ul_info_transfer_s it;
  };
  types_opts::options type() const { return t; }
  c1_c_& c1() { return c1_; }
  types_opts::options t = types_opts::nulltype; c1_c_ c1_;
};
struct ul_dcch_msg_s { ul_dcch_msg_type_c msg; SRSASN_CODE unpack(cbit_ref&) { return SRSASN_ERROR_DECODE_FAIL; } };
}  // namespace rrc
}  // namespace asn1
#endif
