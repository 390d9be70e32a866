// lsn_lte.h - host-side LTE control logic of the product (integer logic that the reference also keeps on the CPU:
// SURVEY.md section 8a rows a5, a8-a12, a19).  C++ mirror of the reference's own types:
//   RNTIManager / Histogram / Interval  /root/reference/lib/src/util/{RNTIManager,Histogram,Interval}.cc
//   DCIMetaFormats                      /root/reference/src/src/MetaFormats.cc
//   dl_sniffer_* grant logic            /root/reference/lib/src/phy/falcon_phch/dl_sniffer_pdsch.c
//   search-space validation             /root/reference/lib/src/phy/falcon_phch/falcon_pdcch.c:183-250
#pragma once
#include <cstdint>
#include <cstring>
#include <list>
#include <memory>
#include <string>
#include <vector>

namespace lsn {

enum DciFormat { FORMAT0 = 0, FORMAT1, FORMAT1A, FORMAT1B, FORMAT1C, FORMAT1D, FORMAT2, FORMAT2A, FORMAT2B, NOF_FORMATS };
extern const DciFormat falcon_ue_all_formats[NOF_FORMATS];  // DCISearch.cc:84-95
enum TxScheme { TXSCHEME_PORT0 = 0, TXSCHEME_DIVERSITY, TXSCHEME_SPATIALMUX, TXSCHEME_CDD };
enum McsTable { TABLE_64QAM = 0, TABLE_256QAM = 1, TABLE_UNKNOWN = 2, TABLE_BOTH = 3, TABLE_FULL_BUFFER = 4 };  // falcon_dci.h dl_sniffer_mcs_table_t
enum ActivationReason { RM_ACT_UNSET = 0, RM_ACT_EVERGREEN, RM_ACT_RAR, RM_ACT_SHORTCUT, RM_ACT_HISTOGRAM, RM_ACT_OTHER };

constexpr uint16_t SIRNTI = 0xFFFF, PRNTI = 0xFFFE, MRNTI = 0xFFFD, RARNTI_START = 0x0001, RARNTI_END = 0x000A,
                   CRNTI_START = 0x000B, CRNTI_END = 0xFFF3;
inline bool rnti_isuser(uint16_t r) { return r >= CRNTI_START && r <= CRNTI_END; }
inline bool rnti_israr(uint16_t r) { return r >= RARNTI_START && r <= RARNTI_END; }

struct Cell {
  uint32_t nof_prb = 0, nof_ports = 0, id = 0, phich_ng_x6 = 1;
  uint32_t pusch_hop_offset = 0;  // SIB2 pusch-HoppingOffset = n_rb_ho of the uplink grant conversion (SubframeWorker.cc:271-273); 0 until known
  uint32_t cp = 0;                // srsran_cell_t.cp: 0 normal (7 symbols per slot), 1 extended (6 symbols per slot; 36.211 Table 6.2.3-1)
  uint32_t nslot() const { return cp ? 6u : 7u; }
  uint32_t nsym() const { return cp ? 12u : 14u; }
  bool crs_symbol01(uint32_t l) const { const uint32_t q = l % nslot(); return q == 0 || q == nslot() - 3; }  // CRS of ports 0, 1: symbols 0 and N_symb - 3 of both slots
  // optional acceleration table built by cell_build_re_tables(): PDSCH-capable REs per (subframe class, first PDSCH
  // symbol l0 0..4, slot, PRB); class 0: subframe 0, 1: subframe 5, 2: any other (srsran_ra_dl_compute_nof_re [srsRAN])
  std::shared_ptr<std::vector<uint16_t>> re_count;
  uint32_t fmt_size[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // cached dci_format_sizeof (0 = not cached), filled by cell_build_re_tables()
};
void cell_build_re_tables(Cell& cell);

// ---- DCI ----
uint32_t dci_format_sizeof(const Cell& cell, DciFormat f);
struct DciTb { uint32_t mcs_idx = 0; int rv = 0; uint32_t ndi = 0; uint32_t cw_idx = 0; };
struct DciDl {
  uint16_t rnti = 0; DciFormat format = FORMAT1; uint32_t L = 0, ncce = 0;
  int alloc_type = 0; uint32_t rbg_bitmask = 0, vrb_bitmask = 0, rbg_subset = 0, shift = 0;
  uint32_t riv = 0; bool distributed = false, ngap2 = false, nprb1a_is2 = false;
  uint32_t pid = 0; DciTb tb[2]; uint32_t tb_cw_swap = 0, pinfo = 0, tpc = 0;
};
struct DciUl {
  uint16_t rnti = 0; uint32_t L = 0, ncce = 0, hopping = 0, riv = 0, mcs_idx = 0, ndi = 0, tpc = 0, n_dmrs = 0, cqi_req = 0;
  int hop_type = -1;  // -1 no hopping; 36.213 Table 8.4-2: 0 = +N/4, 1 = -N/4, 2 = +N/2 (type 1), 3 = type 2
};
struct GrantTb { uint32_t mcs_idx = 0; int rv = 0; uint32_t cw_idx = 0; bool enabled = false; int mod = 0; int tbs = 0; int nof_bits = 0; };
struct PdschGrant {
  bool prb_idx[2][110]; uint32_t nof_prb = 0, nof_re = 0, nof_tb = 0; GrantTb tb[2]; TxScheme tx_scheme = TXSCHEME_PORT0;
  uint32_t pmi = 0, nof_layers = 0;
  uint32_t prb_lo = 110, prb_hi = 0;  // allocated PRBs (either slot) lie in [prb_lo, prb_hi]; lo > hi: none
  PdschGrant() { std::memset(prb_idx, 0, sizeof(prb_idx)); }
};
struct PuschGrant {
  uint32_t L_prb = 0, n_prb = 0, mcs_idx = 0; int mod = 0, tbs = 0, rv = 0;
  uint32_t n_prb2 = 0, hop = 0;  // first PRB of slot 1 when hop == 1 (type-1 hopping); hop == 2: type 2 (not decoded)
};

bool dci_msg_unpack_pdsch(const Cell& cell, const uint8_t* bits, uint32_t nof_bits, DciFormat f, uint16_t rnti, DciDl& out);
bool dci_msg_unpack_pusch(const Cell& cell, const uint8_t* bits, uint32_t nof_bits, uint16_t rnti, DciUl& out);
bool dl_sniffer_ra_dl_dci_to_grant(const Cell& cell, uint32_t sf_idx, uint32_t cfi, bool use_tbs_index_alt, const DciDl& dci, PdschGrant& g);
// both MCS tables at once (falcon_dci.c:284-310 calls the function above twice): same results as two separate calls
void dl_sniffer_ra_dl_dci_to_grant_both(const Cell& cell, uint32_t sf_idx, uint32_t cfi, const DciDl& dci, PdschGrant& g64, bool& ok64,
                                        PdschGrant& g256, bool& ok256);
void dl_sniffer_grant_finish_both(const Cell& cell, uint32_t sf_idx, uint32_t cfi, const DciDl& dci, PdschGrant& g64, bool& ok64, PdschGrant& g256,
                                  bool& ok256);
// true when `tbs` on `nof_prb` PRBs is a value only the DERIVED rows I_TBS 27..33 of spec/lte_tables.h hold (see spec/gen_tables.py)
bool tbs_from_derived_rows(int tbs, uint32_t nof_prb);
bool ra_dl_grant_to_grant_prb_allocation(const Cell& cell, const DciDl& dci, PdschGrant& g);
bool ra_ul_dci_to_grant(const Cell& cell, const DciUl& dci, PuschGrant& g);
// MAC RAR PDU (TS 36.321 6.1.5, 6.2.2, 6.2.3) -> one entry per sub-header (a sub-header without a body - backoff indicator - gives T-CRNTI 0,
// like the reference's loop, DL_Sniffer_PDSCH.cc:632-671); the 20-bit grant as ul_sniffer_dci_rar_unpack reads it (falcon_dci.c:648-683)
struct RarEntry { uint32_t rapid = 0, ta = 0, hopping = 0, riv = 0, mcs = 0, tpc = 0, ul_delay = 0, csi_req = 0; uint16_t t_crnti = 0; bool grant_ok = false; PuschGrant grant; };
int rar_parse(const Cell& cell, const uint8_t* p, int len, RarEntry* out, int cap);
bool ra_ul_dci_to_grant_256(const Cell& cell, const DciUl& dci, PuschGrant& g);  // ulsniffer_ra_ul_dci_to_grant_256, ul_sniffer_pusch.c:138-172
// PUSCH_Decoder::decode's trial order for one scheduled grant (UL_Sniffer_PUSCH.cc:451-568), given the tracked maximum modulation of its RNTI
// (find_tracking_info_RNTI_ul: 1 unknown, 2 / 3 / 4 = 16 / 64 / 256QAM maximum, 5 database full): up to three decode attempts, tried in order until one passes -
// use256: the 256QAM-table grant, qm: modulation bits the decoder runs with (16QAM where the reference leaves enable_64qam off), learn: what update_RNTI_ul is told
// when this attempt passes (decode_run :288-303; 0 = nothing).  Pinned on the compiled reference: tests/test_ref_ul_decode.py.
struct UlTry { bool use256; int qm; int learn; };
int ulTrialPlan(uint32_t mcs_idx, int grant_mod_bits, uint32_t L_prb_256, int mod_bits_256, int tracked, UlTry out[3]);
// investigate_valid_ul_grant (UL_Sniffer_PUSCH.cc:894-918) + the RNTI test of :419
bool ulGrantValid(uint16_t rnti, bool is_rar, int tbs, int tbs_256, uint32_t L_prb);
bool ul_valid_prb(uint32_t L);                                                    // valid_prb_ul, UL_Sniffer_PUSCH.cc:3-10
int dl_sniffer_config_mimo(const Cell& cell, DciFormat f, const DciDl& dci, PdschGrant& g);  // 0 ok
bool pdsch_re_usable(const Cell& cell, uint32_t sf_idx, uint32_t l, uint32_t k);
int ra_tbs_from_idx(int i_tbs, uint32_t n_prb);
uint32_t pdcch_validate_location(uint32_t nof_cce, uint32_t ncce, uint32_t l, uint32_t nsubframe, uint16_t rnti);

struct CbSegm { int C = 0, Cp = 0, Cm = 0, Kp = 0, Km = 0, F = 0; };
bool cbsegm(int tbs, CbSegm& s);
bool qpp_params(int K, uint32_t& f1, uint32_t& f2);
uint32_t turbo_nwin(int K);      // lsn_turbo_nwin(K) from a table (the number of trellis windows of a code block of K bits)
uint32_t turbo_il_offset(int K);  // word offset of block size K in the interleaver address tables (sizes in table order, lsn_turbo_il_words(K) words each); K = 0: total

// ---- Histogram / RNTIManager ----
class Histogram {
public:
  Histogram(uint32_t itemCount, uint32_t valueRange);
  void add(uint16_t item, uint32_t nTimes = 1);
  void addZeros(uint32_t nTimes);  // == add(0, nTimes), without touching the counters of slots that already hold 0
  void setTotals(uint32_t* t) { total = t; }  // shared per-RNTI sum over all formats (fast reject in RNTIManager::validate)
  uint32_t getFrequency(uint16_t item) const { return rnti_histogram[item]; }
private:
  std::vector<uint32_t> rnti_histogram;
  std::vector<uint16_t> rnti_history;
  uint32_t rnti_history_current, rnti_history_end;
  bool rnti_histogram_ready;
  std::vector<uint32_t> nz_pos;   // positions of the window's non-zero entries, oldest first (a ring of its own)
  size_t nz_head, nz_tail;
  uint32_t* total = nullptr;
};
struct Interval { uint16_t start, end; bool matches(uint16_t v) const { return v >= start && v <= end; } };

class RNTIManager {
public:
  RNTIManager(uint32_t nformats, uint32_t maxCandidatesPerStepPerFormat, uint32_t histogramThreshold);
  void addEvergreen(uint16_t a, uint16_t b, uint32_t formatIdx) { evergreen[formatIdx].push_back({a, b}); }
  void addForbidden(uint16_t a, uint16_t b, uint32_t formatIdx) { forbidden[formatIdx].push_back({a, b}); }
  void addCandidate(uint16_t rnti, uint32_t formatIdx);
  bool validate(uint16_t rnti, uint32_t formatIdx);
  bool validateAndRefresh(uint16_t rnti, uint32_t formatIdx);
  void activateAndRefresh(uint16_t rnti, uint32_t formatIdx, ActivationReason reason);
  bool isEvergreen(uint16_t rnti, uint32_t formatIdx) const;
  bool isForbidden(uint16_t rnti, uint32_t formatIdx) const;
  void stepTime();
  uint32_t getFrequency(uint16_t rnti, uint32_t formatIdx) const { return histograms[formatIdx].getFrequency(rnti); }
  ActivationReason getActivationReason(uint16_t rnti) const { return active[rnti] ? (ActivationReason)reason[rnti] : RM_ACT_UNSET; }
  bool activatedByRar(uint16_t rnti) const { return (rar_bits[rnti >> 5] >> (rnti & 31u)) & 1u; }   // getActivationReason(rnti) == RM_ACT_RAR, one bit test (the search asks it of every format 0 candidate)
  uint32_t nofActive() const { return nactive; }
  void setHistogramThreshold(uint32_t t) { threshold = t; }  // RNTIManager.cc:442-444
  // what the candidate pruning of the blind decoder wants to know (k_viterbi): the active RNTIs as bits, the static intervals of a format
  const uint32_t* activeBits() const { return active_bits.data(); }
  // ... without the RNTIs whose activation has run out (validate, :705-708, only notices when the RNTI shows up again: a cell's departed UEs stay "active" for
  // good, and a random CRC that hits one of them would be claimed acceptable by the pruning and refused by the search)
  void activeFreshBits(uint32_t* out2048) const
  {
    for (uint32_t w = 0; w < 2048; w++) {
      uint32_t m = active_bits[w], keep = 0;
      for (uint32_t t = m; t; t &= t - 1) { const uint32_t b = (uint32_t)__builtin_ctz(t); if (timestamp - lastSeen[w * 32 + b] < lifetime) keep |= 1u << b; }
      out2048[w] = keep;
    }
  }
  const std::vector<Interval>& evergreenOf(uint32_t f) const { return evergreen[f]; }
  const std::vector<Interval>& forbiddenOf(uint32_t f) const { return forbidden[f]; }
private:
  uint32_t getLikelyDlFormatIdx(uint16_t rnti) const;
  void activateRNTI(uint16_t rnti, ActivationReason r);
  void deactivateRNTI(uint16_t rnti);
  uint32_t nformats;
  std::vector<Histogram> histograms;
  std::vector<std::vector<Interval>> evergreen, forbidden;
  std::vector<uint8_t> active, reason;
  std::vector<uint32_t> active_bits;   // active[], one bit per RNTI (2048 words)
  std::vector<uint32_t> rar_bits;      // active[] && reason[] == RM_ACT_RAR
  std::vector<uint32_t> lastSeen, assocFormatIdx, totals;
  uint32_t nactive, timestamp, lifetime, threshold, maxCandidatesPerStepPerFormat;
  std::vector<int32_t> remainingCandidates;
};

// ---- DCIMetaFormats ----
struct MetaFormat { DciFormat format; uint32_t global_index; uint32_t hits; };
class DCIMetaFormats {
public:
  explicit DCIMetaFormats(uint32_t nformats, double split_ratio = 1.0);
  void update_formats();
  MetaFormat** getPrimaryMetaFormats() { return primary.data(); }
  MetaFormat** getSecondaryMetaFormats() { return secondary.data(); }
  uint32_t getNofPrimaryMetaFormats() const { return nprimary; }
  uint32_t getNofSecondaryMetaFormats() const { return nsecondary; }
  uint32_t primaryMask() const { uint32_t m = 0; for (uint32_t i = 0; i < nprimary; i++) m |= 1u << primary[i]->global_index; return m; }   // bit f: format f is tried in the primary pass
  void setSkipSecondaryMetaFormats(bool s) { skip_secondary = s; }
  bool skipSecondaryMetaFormats() const { return skip_secondary; }
private:
  std::vector<MetaFormat> all;
  std::vector<MetaFormat*> primary, secondary;
  uint32_t nprimary = 0, nsecondary = 0;
  bool skip_secondary = false;
  double split_ratio;
};

// ---- MAC DL-SCH walk and RRCConnectionSetup (lsn_rrc.cc) ----
struct MacSubheader { uint32_t lcid = 0; bool is_sdu = false; uint32_t off = 0, len = 0; };
// sch_pdu::parse_packet: number of subheaders with payload offsets / lengths, 0 when the PDU does not parse
int mac_dlsch_parse(const uint8_t* pdu, int len, MacSubheader* out, int cap);
struct UeSpecConfig {  // ltesniffer_ue_spec_config_t, MCSTracking.h:37-43
  bool has_ue_config = false;
  float p_a = 0.0f;                                         // dB
  uint32_t i_offset_ack = 10, i_offset_cqi = 8, i_offset_ri = 11;  // MCSTracking::set_default_of_default_config, MCSTracking.cc:1531-1540
  uint32_t cqi_type = 2;                                    // 0 wideband, 1 UE-selected sub-band, 2 higher-layer sub-band
  bool from_lcid0 = true;                                   // (book-keeping of setups_of_pdu: the SDU this came from sat on logical channel 0)
};
// PDSCH_Decoder::decode_rrc_connection_setup: true when the CCCH SDU is an RRCConnectionSetup (out filled)
bool rrc_conn_setup_decode(const uint8_t* sdu, int len, UeSpecConfig& out);

// SystemInformationBlockType2 as far as the sniffer uses it (ULSchedule::set_config, ULSchedule.cc:140-158; SubframeWorker.cc:271-273)
struct Sib2Config {
  uint32_t n_sb = 1, hopping_mode = 0, pusch_hop_offset = 0, enable_64qam = 0;            // pusch-ConfigBasic
  uint32_t group_hopping_enabled = 0, group_assignment_pusch = 0, sequence_hopping_enabled = 0, cyclic_shift = 0;  // ul-ReferenceSignalsPUSCH
  uint32_t root_seq_idx = 0, prach_config_idx = 0, high_speed_flag = 0, zero_corr_zone = 0, prach_freq_offset = 0;  // prach-Config
};
// PDSCH_Decoder::decode_SIB, DL_Sniffer_PDSCH.cc:531-557: 0 = not a BCCH-DL-SCH message that unpacks, 1 = unpacks but carries no SIB2
// (SystemInformationBlockType1, or a SystemInformation whose first entry is another block), 2 = SIB2 found (out filled)
int sib2_decode(const uint8_t* pdu, int len, Sib2Config& out);

// ---- security-API view of decoded downlink blocks (PDSCH_Decoder::run_api_dl_mode, DL_Sniffer_PDSCH.cc:804-879) ----
struct PagingId { bool is_imsi = false; uint32_t nof_digits = 0; uint8_t digits[21] = {0}; uint32_t mmec = 0, m_tmsi = 0; };
// PCCH-Message -> paging records (decode_imsi_tmsi_paging, DL_Sniffer_PDSCH.cc:84-127): number of records, -1 when the message does not unpack
int paging_decode(const uint8_t* pdu, int len, PagingId* out, int cap);
struct ApiEvent { uint32_t tti = 0; uint16_t rnti = 0; uint32_t id_type = 0, msg_type = 0; char value[24] = {0}; };  // print_api_dl's arguments
enum { API_ID_RAN_VAL = 0, API_ID_TMSI = 1, API_ID_CON_RES = 2, API_ID_IMSI = 3, API_ID_IMEI = 4, API_ID_IMEISV = 5, API_ID_NONE = 0xFFFFFFFFu,
       API_MSG_CON_REQ = 0, API_MSG_CON_SET = 1, API_MSG_ATT_REQ = 2, API_MSG_ID_RES = 3, API_MSG_UE_CAP = 4, API_MSG_PAGING = 5, API_MSG_CON_RECONFIG = 6 };  // Sniffer_dependency.h:42-55
// what run_api_dl_mode reports for one CRC-ok downlink block (name = first letter of the reference's RNTI name, api_mode 0 / 2 / 3):
// events appended to ev (at most cap), return value = true when the block also goes to the API pcap (write_dl_paging_api / write_dl_crnti_api)
// the uplink side for a decoded Msg3 (PUSCH of a RAR grant, api_mode 0 / 3; PUSCH_Decoder::decode_run :306-327 + decode_rrc_connection_request
// :47-93): the initial UE identity of an RRCConnectionRequest; return value = the block goes to the API pcap (write_ul_crnti_api)
bool api_ul_msg3_events(int api_mode, const uint8_t* pdu, int len, uint16_t rnti, uint32_t tti, ApiEvent* ev, int cap, int* nev);
// any other decoded uplink block (api_mode 1 / 2 / 3; decode_run :328-372, decode_ul_dcch :95-143, decode_nas_ul :146-247): SRB messages behind
// RLC AM + PDCP - UECapabilityInformation (modes 1, 3), attach request / identity response identities (modes 2, 3)
bool api_ul_dcch_events(int api_mode, const uint8_t* pdu, int len, uint16_t rnti, uint32_t tti, ApiEvent* ev, int cap, int* nev);
// RRCConnectionReconfiguration (DL-DCCH) whose first dedicatedInfoNAS is an attach accept with a GUTI -> its M-TMSI (decode_rrc_connection_reconfig, DL_Sniffer_PDSCH.cc:181-220)
bool rrc_reconfig_attach_accept_tmsi(const uint8_t* sdu, int len, uint32_t& m_tmsi);
bool api_dl_events(int api_mode, char name, const uint8_t* pdu, int len, uint16_t rnti, uint32_t tti, ApiEvent* ev, int cap, int* nev);

// ---- MCSTracking (DL table learning + UE-specific configuration + database ageing; MCSTracking.cc:758-927,1269-1400,1444-1540) ----
// Time is counted in SUBFRAMES processed so far (`now`), not in clock() ticks: the reference ages its database by CPU time consumed
// (MCSTracking.cc:778,854-858), i.e. by replay speed; one subframe = 1 ms of air time is the deterministic equivalent.
class MCSTracking {
public:
  UeSpecConfig get_ue_config_rnti(uint16_t rnti) const;         // the entry's configuration, else the default
  void update_ue_config_rnti(uint16_t rnti, const UeSpecConfig& c, uint32_t now);
  bool check_default_config() const { return has_default; }
  float default_p_a() const { return default_cfg.p_a; }
  UeSpecConfig default_config() const { UeSpecConfig c = default_cfg; c.has_ue_config = false; return c; }
  void update_default_ue_config(const UeSpecConfig& c) { default_cfg = c; has_default = true; }
  // one decoded C-RNTI transport block: every CCCH SDU is tried as RRCConnectionSetup (DL_Sniffer_PDSCH.cc:1041-1070); true when one was
  bool learn_from_pdu(const uint8_t* pdu, int len, uint16_t rnti, uint32_t now);
  // the same with the PDU already walked (setups_of_pdu): the commit thread applies what a decode thread parsed
  // any_lcid: also the ones that did not come on logical channel 0 - the 64QAM-table attempt of decode_dl_mode's unknown-table branch tries EVERY SDU of the block
  // (DL_Sniffer_PDSCH.cc:1140 has no LCID test; the known-table branch, :1049, and run_decode, :286, have)
  bool learn_setups(const UeSpecConfig* c, int n, uint16_t rnti, uint32_t now, bool any_lcid = false);
  static int setups_of_pdu(const uint8_t* pdu, int len, UeSpecConfig* out, int cap, bool any_lcid = false);  // RRCConnectionSetups among the SDUs, in order (from_lcid0 says which)
  McsTable find_tracking_info_RNTI_dl(uint16_t rnti, uint32_t now);  // refreshes the entry's time stamp (:778-779)
  void update_RNTI_dl(uint16_t rnti, McsTable t, uint32_t now);
  void update_rar_time_crnti(uint16_t crnti, uint32_t now);
  // update_statistic_dl (:1269-1400) without the HARQ branches (harq_mode is 0): table = the table fixed when the DCI was collected,
  // tb_en = enabled flags of the statistic grant, success = final CRC verdicts, mimo_ret = 0 / -1 / -2 / -3 of dl_sniffer_config_mimo
  void update_statistic_dl(uint16_t rnti, DciFormat f, McsTable table, const bool tb_en[2], const bool success[2], int mimo_ret, uint32_t now);
  // update_database_dl (:850-927): entries idle for more than `interval` whole seconds, never active, or wrongly detected are deleted;
  // a known table with a success rate under 15 % is reset.  Appends the RNTIs it touched to `changed`.
  void update_database_dl(uint32_t now, std::vector<uint16_t>* changed = nullptr);
  uint32_t get_interval() const { return interval; }   // seconds (MCSTracking.h:162: 5)
  void set_interval(uint32_t seconds) { interval = seconds; }
  uint32_t nof_RNTI_member_dl() const { return count; }
  McsTable peek(uint16_t rnti) const { return db[rnti].present ? (McsTable)db[rnti].table : TABLE_UNKNOWN; }
  bool present(uint16_t rnti) const { return db[rnti].present != 0; }
  MCSTracking() : db(65536), ue_cfg(65536) {}
private:
  struct Entry {
    uint8_t present = 0, has_rar = 0, table = TABLE_UNKNOWN; uint16_t nof_msg_after_rar = 0;
    uint32_t time = 0, nof_active = 0, nof_success_mgs = 0, nof_unsupport_mimo = 0, nof_pinfo = 0, nof_other_mimo = 0;
  };
  void add_RNTI_dl(uint16_t rnti, uint32_t now);
  std::vector<Entry> db;
  std::vector<UeSpecConfig> ue_cfg;  // [65536], valid where db[].present
  UeSpecConfig default_cfg;
  bool has_default = false;
  uint32_t count = 0;
  uint32_t interval = 5;
  static constexpr uint32_t max_size = 250;  // MCSTracking.h:30
  static constexpr uint16_t rar_thresold = 3;
};

// ---- HARQ (src/include/HARQ.h, src/src/HARQ.cc): which transport block of a known-table C-RNTI grant is a new transmission, a retransmission to be
// soft-combined with the buffer of its (RNTI, process, TB), or already decoded.  150 entities from the constructor + 150 from init_HARQ (HARQ.cc:15-20,
// 48-52), 8 processes x 2 transport blocks; clock() is replaced by the subframe count; the per-TB mutexes only matter between worker threads
// (DL_SNIFFER_HARQ_BUSY cannot occur in the sequential commit order).  Off in the reference (ArgManager.cc:50,211-213).
enum HarqRet { HARQ_NEW_TX = 0, HARQ_RE_TX = 1, HARQ_FULL_BUFFER = 2, HARQ_DECODED = 3, HARQ_BUSY = 4 };
class HarqDatabase {
public:
  static constexpr int NENT = 300, NPID = 8;
  HarqDatabase() : ent(NENT), ent_of_rnti(65536, (int16_t)-1) {}
  // HARQ::is_retransmission (HARQ.cc:71-135); entity: index of the RNTI's entity (-1: none) - with pid and tid it names the soft buffer
  HarqRet is_retransmission(uint16_t rnti, uint32_t pid, int tid, bool ndi, int tbs, uint32_t sfn, uint32_t sf_idx, int& entity);
  // HARQ::updateHARQRNTI / updateProcess (HARQ.cc:155-190)
  void update(int entity, uint32_t pid, int tid, uint32_t sfn, uint32_t sf_idx, bool last_decoded, bool ndi, int rv, int tbs, uint32_t now);
  void update_database(uint32_t now);   // HARQ::updateHARQDatabase (HARQ.cc:206-238), the 10 s timer of LTESniffer_Core.cc:487-494
  int getlastTbs(uint16_t rnti, uint32_t pid, int tid) const;  // HARQ::getlastTbs (HARQ.cc:262-274): the size the RNTI's (last) entity remembers for (process, block), 0 without entity
  uint64_t stats[5] = {0, 0, 0, 0, 0};  // verdicts so far, by HarqRet
private:
  struct Tb { uint32_t sfn = 0, sf_idx = 0; bool last_decoded = false, ndi = false, is_first = true; int rv = 0, tbs = 0; };
  struct Entity { uint16_t rnti = 0; uint32_t time = 0; Tb tb[NPID][2]; };
  std::vector<Entity> ent;
  std::vector<int16_t> ent_of_rnti;   // [65536] the entity an RNTI owns, -1: none (is_retransmission)
  int nof_aval = 150;
};

// ---- GF(2) helpers for the transport-block CRC combine ----
uint32_t crc24a_xpow(uint64_t n);                 // x^n mod g_CRC24A
uint32_t crc24a_xpow_bytes(uint32_t nbytes);      // x^(8 nbytes) mod g_CRC24A, memoised per code-block payload size
uint32_t crc24a_mulmod(uint32_t a, uint32_t b);   // a*b mod g_CRC24A
uint32_t crc_bits(uint32_t poly, int order, const uint8_t* bits, int n);

}  // namespace lsn
