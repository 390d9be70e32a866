/* oracle/_ref build stand-in for the srsRAN MAC PDU walkers, the RRC ASN.1 message classes and the NAS helpers that the reference's DL_Sniffer_PDSCH.cc /
 * UL_Sniffer_PUSCH.cc name (test infrastructure, NOT product, NOT a copy of srsRAN; see standin.h).
 *
 * Purpose: compile the REFERENCE'S OWN decode control flow from where it lies - PDSCH_Decoder::decode_dl_mode / run_decode / decode_SIB / decode_ul_mode
 * (src/src/DL_Sniffer_PDSCH.cc) - and drive it with scripted decoder verdicts (oracle/Makefile.ref: libref_falcon_decode.so, tests/test_ref_decode.py).
 * The classes below have the MEMBER NAMES those files touch and nothing else.  Every one of them that would parse bytes (MAC sub-header walk, RAR walk, the four RRC
 * message types, the NAS attach accept) hands the bytes to a function pointer that the test binds to the ORACLE's parser of the same thing (o_rrc.c, o_worker.c) and
 * copies the handful of fields the reference reads into its members: the layer-2/3 parsing stays the oracle's (pinned on the reference's own captures,
 * tests/test_rrc_oracle.py), the DECISIONS taken on it are the reference's machine code. */
#pragma once
#include "srsran/standin.h"
#include "srsran/asn1/rrc/standin_rrc.h"
#ifdef __cplusplus
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

/* ---- what decode_glue.cc binds (the oracle's parsers, flat signatures) ---- */
extern "C" {
/* MAC DL-SCH PDU -> n sub-headers x {lcid, is_sdu, offset, length}; 0 when the PDU does not parse */
extern int (*lsn_l2_mac_parse)(const uint8_t* pdu, int len, uint32_t* out4, int cap);
/* DL-CCCH message -> 1 when it is an RRCConnectionSetup: out = {p_a index, beta ack, beta cqi, beta ri, aperiodic CQI mode present, mode 0..4 (rm12 rm20 rm22 rm30 rm31)} */
extern int (*lsn_l2_conn_setup)(const uint8_t* sdu, int len, uint32_t* out6);
/* MAC RAR PDU -> n x {t_crnti, ta, 20 grant bits packed MSB first}; -1 when it does not parse */
extern int (*lsn_l2_rar_parse)(const uint8_t* pdu, int len, uint32_t* out3, int cap);
/* PCCH message -> n records x {is_imsi, nof_digits, m_tmsi, 15 digits ...} (18 words each); -1 when it does not unpack */
extern int (*lsn_l2_paging)(const uint8_t* pdu, int len, uint32_t* out18, int cap);
/* DL-DCCH message with an attach accept in its first NAS PDU -> 1 and the M-TMSI of its GUTI */
extern int (*lsn_l2_reconfig_tmsi)(const uint8_t* sdu, int len, uint32_t* m_tmsi);
/* BCCH-DL-SCH message -> 0 not a system information message with SIB2, 1 SIB1, 2 carries SIB2: out = the o_sib2_t words */
extern int (*lsn_l2_sib2)(const uint8_t* pdu, int len, uint32_t* out14);
}

namespace srslog { struct basic_logger { int _ = 0; }; inline basic_logger& fetch_basic_logger(const char*) { static basic_logger l; return l; } }

namespace srsran {
/* PcapWriter.h holds one by value in a class LTESniffer does not use ("not using now") */
class mac_pcap {
public:
  void write_dl_sirnti(uint8_t*, uint32_t, bool, uint32_t, uint8_t) {}
  void write_dl_pch(uint8_t*, uint32_t, bool, uint32_t, uint8_t) {}
  void write_dl_ranti(uint8_t*, uint32_t, uint16_t, bool, uint32_t, uint8_t) {}
  void write_dl_crnti(uint8_t*, uint32_t, uint16_t, bool, uint32_t, uint8_t) {}
  uint32_t open(std::string, uint32_t = 0) { return 0; }
  uint32_t close() { return 0; }
};
enum class dl_sch_lcid { CCCH = 0, CON_RES_ID = 0x1c, PADDING = 0x1f };
class sch_subh {
public:
  bool is_sdu() const { return sdu; }
  uint32_t get_sdu_lcid() const { return lcid; }
  uint32_t lcid_value() const { return lcid; }
  int get_payload_size() const { return (int)len; }
  uint8_t* get_sdu_ptr() const { return ptr; }
  uint64_t get_con_res_id() const { uint64_t v = 0; for (uint32_t i = 0; i < 6 && i < len; i++) v = (v << 8) | ptr[i]; return v; }
  bool sdu = false; uint32_t lcid = 0, len = 0; uint8_t* ptr = nullptr;
};
class sch_pdu {
public:
  sch_pdu(uint32_t, srslog::basic_logger&) {}
  void init_rx(uint32_t len_, bool) { len = len_; }
  void parse_packet(uint8_t* p)
  {
    uint32_t w[4 * 32];
    const int n = lsn_l2_mac_parse(p, (int)len, w, 32);
    sub.clear(); cur = -1;
    for (int i = 0; i < n; i++) { sch_subh s; s.lcid = w[4 * i]; s.sdu = w[4 * i + 1] != 0; s.ptr = p + w[4 * i + 2]; s.len = w[4 * i + 3]; sub.push_back(s); }
  }
  bool next() { if (cur + 1 < (int)sub.size()) { cur++; return true; } return false; }
  sch_subh* get() { return &sub[(size_t)cur]; }
private:
  uint32_t len = 0; std::vector<sch_subh> sub; int cur = -1;
};
class rar_subh {
public:
  uint16_t get_temp_crnti() const { return t_crnti; }
  uint32_t get_ta_cmd() const { return ta; }
  void get_sched_grant(uint8_t grant[SRSRAN_RAR_GRANT_LEN]) const { for (int i = 0; i < SRSRAN_RAR_GRANT_LEN; i++) grant[i] = (uint8_t)((g20 >> (19 - i)) & 1u); }
  uint16_t t_crnti = 0; uint32_t ta = 0, g20 = 0;
};
class rar_pdu {
public:
  void init_rx(uint32_t len_) { len = len_; }
  int parse_packet(uint8_t* p)
  {
    uint32_t w[3 * 32];
    const int n = lsn_l2_rar_parse(p, (int)len, w, 32);
    sub.clear(); cur = -1;
    if (n < 0) return SRSRAN_ERROR;
    for (int i = 0; i < n; i++) { rar_subh s; s.t_crnti = (uint16_t)w[3 * i]; s.ta = w[3 * i + 1]; s.g20 = w[3 * i + 2]; sub.push_back(s); }
    return SRSRAN_SUCCESS;
  }
  bool next() { if (cur + 1 < (int)sub.size()) { cur++; return true; } return false; }
  rar_subh* get() { return &sub[(size_t)cur]; }
private:
  uint32_t len = 0; std::vector<rar_subh> sub; int cur = -1;
};
}  // namespace srsran
using namespace srsran;

/* ---- NAS (liblte_mme.h): the attach accept behind an RRCConnectionReconfiguration, decode_rrc_connection_reconfig DL_Sniffer_PDSCH.cc:181-220 ---- */
typedef uint8_t uint8;
typedef enum { LIBLTE_SUCCESS = 0, LIBLTE_ERROR_INVALID_INPUTS } LIBLTE_ERROR_ENUM;
struct LIBLTE_BYTE_MSG_STRUCT { uint32_t N_bytes = 0; uint8_t msg[4096]; };
#define LIBLTE_MME_MSG_TYPE_ATTACH_ACCEPT 0x42
struct LIBLTE_MME_ATTACH_ACCEPT_MSG_STRUCT { LIBLTE_BYTE_MSG_STRUCT esm_msg; bool guti_present; struct { struct { uint32_t m_tmsi; } guti; } guti; };
struct LIBLTE_MME_ACTIVATE_DEFAULT_EPS_BEARER_CONTEXT_REQUEST_MSG_STRUCT { int _; };
/* the three calls see the NAS PDU only; the oracle's walk starts at the DL-DCCH message.  The stand-in message class below keeps the whole SDU and hands it over here */
LIBLTE_ERROR_ENUM liblte_mme_parse_msg_header(LIBLTE_BYTE_MSG_STRUCT* msg, uint8* pd, uint8* msg_type);
LIBLTE_ERROR_ENUM liblte_mme_unpack_attach_accept_msg(LIBLTE_BYTE_MSG_STRUCT* msg, LIBLTE_MME_ATTACH_ACCEPT_MSG_STRUCT* attach_accept);
LIBLTE_ERROR_ENUM liblte_mme_unpack_activate_default_eps_bearer_context_request_msg(LIBLTE_BYTE_MSG_STRUCT* msg, LIBLTE_MME_ACTIVATE_DEFAULT_EPS_BEARER_CONTEXT_REQUEST_MSG_STRUCT* out);

namespace asn1 {
enum SRSASN_CODE { SRSASN_SUCCESS = 0, SRSASN_ERROR_ENCODE_FAIL, SRSASN_ERROR_DECODE_FAIL };
struct cbit_ref { cbit_ref(const uint8_t* p_, uint32_t n_) : p(p_), n(n_) {} const uint8_t* p; uint32_t n; };
template <class T> struct typed { T value; bool operator==(T o) const { return value == o; } bool operator!=(T o) const { return value != o; } };

namespace rrc {
/* -- RRCConnectionSetup (DL-CCCH) -- */
/* (RRCConnectionSetup-r8-IEs and SystemInformationBlockType2: srsran/asn1/rrc/standin_rrc.h) */
struct rrc_conn_setup_s {
  struct c1_t { rrc_conn_setup_r8_ies_s r8; rrc_conn_setup_r8_ies_s& rrc_conn_setup_r8() { return r8; } };
  struct crit_exts_t { c1_t c1_; c1_t& c1() { return c1_; } } crit_exts;
};
struct dl_ccch_msg_type_c {
  struct types_opts { enum options { c1, msg_class_ext, nulltype }; };
  struct c1_c_ {
    struct types { enum options { rrc_conn_reest, rrc_conn_reest_reject, rrc_conn_reject, rrc_conn_setup, nulltype }; };
    typed<types::options> type() const { return typed<types::options>{t}; }
    rrc_conn_setup_s& rrc_conn_setup() { return setup; }
    types::options t = types::nulltype; rrc_conn_setup_s setup;
  };
  types_opts::options type() const { return t; }
  c1_c_& c1() { return c1_; }
  types_opts::options t = types_opts::nulltype; c1_c_ c1_;
};
struct dl_ccch_msg_s {
  dl_ccch_msg_type_c msg;
  SRSASN_CODE unpack(cbit_ref& b)
  {
    uint32_t w[6];
    msg.t = dl_ccch_msg_type_c::types_opts::c1;
    if (lsn_l2_conn_setup(b.p, (int)b.n, w) != 1) { msg.c1_.t = dl_ccch_msg_type_c::c1_c_::types::rrc_conn_reject; return SRSASN_SUCCESS; }  /* some other DL-CCCH message */
    msg.c1_.t = dl_ccch_msg_type_c::c1_c_::types::rrc_conn_setup;
    phys_cfg_ded_standin& p = msg.c1_.setup.crit_exts.c1_.r8.rr_cfg_ded.phys_cfg_ded;
    p.pdsch_cfg_ded.p_a = w[0]; p.pusch_cfg_ded.beta_offset_ack_idx = w[1]; p.pusch_cfg_ded.beta_offset_cqi_idx = w[2]; p.pusch_cfg_ded.beta_offset_ri_idx = w[3];
    p.cqi_report_cfg.cqi_report_mode_aperiodic_present = w[4] != 0; p.cqi_report_cfg.cqi_report_mode_aperiodic = (cqi_report_mode_aperiodic_e)w[5];
    return SRSASN_SUCCESS;
  }
};
/* -- RRCConnectionReconfiguration (DL-DCCH) -- */
struct nas_pdu_standin { std::vector<uint8_t> b; uint32_t size() const { return (uint32_t)b.size(); } const uint8_t* data() const { return b.data(); } };
struct rrc_conn_recfg_s {
  struct r8_t { std::vector<nas_pdu_standin> ded_info_nas_list; };
  struct c1_t { r8_t r8; r8_t& rrc_conn_recfg_r8() { return r8; } };
  struct crit_exts_t { c1_t c1_; c1_t& c1() { return c1_; } } crit_exts;
};
struct dl_dcch_msg_type_c {
  struct types { enum options { c1, msg_class_ext, nulltype }; };
  struct c1_c_ {
    struct types { enum options { csfb_params_resp_cdma2000, dl_info_transfer, ho_from_eutra_prep_request, mob_from_eutra_cmd, rrc_conn_recfg, rrc_conn_release, nulltype }; };
    types::options type() const { return t; }
    rrc_conn_recfg_s& rrc_conn_recfg() { return recfg; }
    types::options t = types::nulltype; rrc_conn_recfg_s recfg;
  };
  types::options type() const { return t; }
  c1_c_& c1() { return c1_; }
  types::options t = types::nulltype; c1_c_ c1_;
};
struct dl_dcch_msg_s {
  dl_dcch_msg_type_c msg;
  /* the reference reads ded_info_nas_list[0] and runs the three liblte calls on it; here the whole SDU travels in that slot and the liblte stand-ins (decode_glue.cc)
   * ask the oracle's walk (o_rrc_reconfig_tmsi) once: "attach accept with a GUTI -> its M-TMSI" */
  SRSASN_CODE unpack(cbit_ref& b)
  {
    uint32_t tmsi = 0;
    msg.t = dl_dcch_msg_type_c::types::c1;
    if (lsn_l2_reconfig_tmsi(b.p, (int)b.n, &tmsi) != 1) { msg.c1_.t = dl_dcch_msg_type_c::c1_c_::types::dl_info_transfer; return SRSASN_SUCCESS; }
    msg.c1_.t = dl_dcch_msg_type_c::c1_c_::types::rrc_conn_recfg;
    nas_pdu_standin n;
    n.b.assign(8, 0); n.b[0] = 0x07; n.b[1] = LIBLTE_MME_MSG_TYPE_ATTACH_ACCEPT; n.b[4] = (uint8_t)(tmsi >> 24); n.b[5] = (uint8_t)(tmsi >> 16); n.b[6] = (uint8_t)(tmsi >> 8); n.b[7] = (uint8_t)tmsi;
    msg.c1_.recfg.crit_exts.c1_.r8.ded_info_nas_list.assign(1, n);
    return SRSASN_SUCCESS;
  }
};
/* -- Paging (PCCH) -- */
struct paging_ue_id_c {
  struct types_opts { enum options { s_tmsi, imsi, nulltype }; };
  struct m_tmsi_t { uint32_t v = 0; uint32_t to_number() const { return v; } };
  struct s_tmsi_t { m_tmsi_t m_tmsi; };
  types_opts::options type() const { return t; }
  const std::vector<uint8_t>& imsi() const { return digits; }
  const s_tmsi_t& s_tmsi() const { return st; }
  types_opts::options t = types_opts::nulltype; std::vector<uint8_t> digits; s_tmsi_t st;
};
struct paging_record_s { paging_ue_id_c ue_id; };
typedef std::vector<paging_record_s> paging_record_list_l;
struct paging_s { bool paging_record_list_present = false; paging_record_list_l paging_record_list; };
struct pcch_msg_type_c {
  struct types_opts { enum options { c1, msg_class_ext, nulltype }; };
  struct c1_c_ { paging_s pg; paging_s& paging() { return pg; } };
  typed<types_opts::options> type() const { return typed<types_opts::options>{t}; }
  c1_c_& c1() { return c1_; }
  types_opts::options t = types_opts::c1; c1_c_ c1_;
};
struct pcch_msg_s {
  pcch_msg_type_c msg;
  SRSASN_CODE unpack(cbit_ref& b)
  {
    uint32_t w[18 * 16];
    const int n = lsn_l2_paging(b.p, (int)b.n, w, 16);
    msg.c1_.pg = paging_s();
    if (n < 0) { msg.t = pcch_msg_type_c::types_opts::c1; return SRSASN_ERROR_DECODE_FAIL; }   /* (the reference's condition is "unpacked OR type c1", DL_Sniffer_PDSCH.cc:89) */
    msg.c1_.pg.paging_record_list_present = n > 0;
    for (int i = 0; i < n; i++) {
      paging_record_s r;
      const uint32_t* q = w + 18 * i;
      if (q[0]) { r.ue_id.t = paging_ue_id_c::types_opts::imsi; for (uint32_t k = 0; k < 15; k++) r.ue_id.digits.push_back((uint8_t)(k < q[1] ? q[3 + k] : 0)); }
      else { r.ue_id.t = paging_ue_id_c::types_opts::s_tmsi; r.ue_id.st.m_tmsi.v = q[2]; }
      msg.c1_.pg.paging_record_list.push_back(r);
    }
    return SRSASN_SUCCESS;
  }
};
/* -- SystemInformation with SIB2 (BCCH-DL-SCH) -- */
struct sib_info_item_c {
  struct types { enum options { sib2, sib3, nulltype }; };
  typed<types::options> type() const { return typed<types::options>{t}; }
  sib_type2_s& sib2() { return s2; }
  types::options t = types::nulltype; sib_type2_s s2;
};
struct sys_info_r8_ies_s { typedef std::vector<sib_info_item_c> sib_type_and_info_l_; sib_type_and_info_l_ sib_type_and_info; };
struct sys_info_s {
  struct crit_exts_t { sys_info_r8_ies_s r8; sys_info_r8_ies_s& sys_info_r8() { return r8; } } crit_exts;
};
struct bcch_dl_sch_msg_type_c {
  struct c1_c_ {
    struct types { enum options { sys_info, sib_type1, nulltype }; };
    types::options type() const { return t; }
    sys_info_s& sys_info() { return si; }
    types::options t = types::nulltype; sys_info_s si;
  };
  c1_c_& c1() { return c1_; }
  c1_c_ c1_;
};
struct bcch_dl_sch_msg_s {
  bcch_dl_sch_msg_type_c msg;
  SRSASN_CODE unpack(cbit_ref& b)
  {
    uint32_t w[14];
    const int k = lsn_l2_sib2(b.p, (int)b.n, w);
    msg.c1_.si = sys_info_s();
    if (k == 1) { msg.c1_.t = bcch_dl_sch_msg_type_c::c1_c_::types::sib_type1; return SRSASN_SUCCESS; }
    msg.c1_.t = bcch_dl_sch_msg_type_c::c1_c_::types::sys_info;
    if (k == 2) {
      sib_info_item_c it; it.t = sib_info_item_c::types::sib2;
      /* o_sib2_t order: n_sb, hopping_mode, pusch_hop_offset, enable_64qam, group_hopping_enabled, group_assignment_pusch, sequence_hopping_enabled, cyclic_shift,
       * root_seq_idx, prach_config_idx, high_speed_flag, zero_corr_zone, prach_freq_offset, bits_used */
      it.s2.rr_cfg_common.pusch_cfg_common.pusch_cfg_basic.n_sb = w[0]; it.s2.rr_cfg_common.pusch_cfg_common.pusch_cfg_basic.pusch_hop_offset = w[2];
      it.s2.rr_cfg_common.pusch_cfg_common.ul_ref_sigs_pusch.group_hop_enabled = w[4] != 0; it.s2.rr_cfg_common.pusch_cfg_common.ul_ref_sigs_pusch.group_assign_pusch = w[5];
      it.s2.rr_cfg_common.pusch_cfg_common.ul_ref_sigs_pusch.seq_hop_enabled = w[6] != 0; it.s2.rr_cfg_common.pusch_cfg_common.ul_ref_sigs_pusch.cyclic_shift = w[7];
      it.s2.rr_cfg_common.prach_cfg.root_seq_idx = w[8]; it.s2.rr_cfg_common.prach_cfg.prach_cfg_info.prach_cfg_idx = w[9];
      it.s2.rr_cfg_common.prach_cfg.prach_cfg_info.high_speed_flag = w[10] != 0; it.s2.rr_cfg_common.prach_cfg.prach_cfg_info.zero_correlation_zone_cfg = w[11];
      it.s2.rr_cfg_common.prach_cfg.prach_cfg_info.prach_freq_offset = w[12];
      msg.c1_.si.crit_exts.r8.sib_type_and_info.push_back(it);
    }
    return k == 0 ? SRSASN_ERROR_DECODE_FAIL : SRSASN_SUCCESS;
  }
};
}  // namespace rrc
}  // namespace asn1
#endif
