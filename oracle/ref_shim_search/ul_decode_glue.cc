/* oracle/_ref harness around the REFERENCE'S OWN uplink decode control flow (test infrastructure, NOT product; see srsran/standin_ul.h).
 *
 * Compiled verbatim from /root/reference by oracle/Makefile.ref into _ref/libref_falcon_ul_decode.so:
 *   src/src/UL_Sniffer_PUSCH.cc     PUSCH_Decoder::decode (:389-583): which grants of the schedule are tried at all (investigate_valid_ul_grant :894-918, the valid
 *                                   PRB counts :3-10), with which uplink MCS table and enable_64qam in which order for which tracked maximum modulation, the UCI
 *                                   layout each attempt is configured with (:429-450), decode_run (:248-310): what a CRC-ok block writes and teaches the tracking
 *                                   database, the SNR gate of the statistics (:571-575)                                     
 *   src/src/SubframePower.cc        computePower / getRBPowerUL (read by the unknown-modulation branch)
 *   + everything of libref_falcon_decode.so: the MCSTracking database the decoder talks to is the reference's own (MCSTracking.cc)
 * What is NOT the reference here:
 *   - srsran_chest_ul_estimate_pusch + srsran_pusch_decode: the uplink receiver itself (srsRAN).  Every decode call is RECORDED - the grant as configured, enable_64qam,
 *     the UCI configuration - and answered by a callback of the test (CRC verdict, payload bytes, the SNR the estimator would report).  The same callback answers the
 *     oracle's attempts (o_worker_set_ul_script).
 *   - srsran_enb_ul_fft: a no-op on a zeroed grid; the PRACH detector (work_prach is not driven here; tests/test_gpu_prach.py covers that row).
 *   - the identity API (api_mode >= 0): its message classes do not parse in this build (standin_ul.h); the harness runs with api_mode -1. */
#include "decode_glue.cc"
#include "include/UL_Sniffer_PUSCH.h"
#include "include/SubframePower.h"

/* one decode attempt as the reference configured it: 16 words {tti, rnti, L_prb, n_prb[0], mcs_idx, modulation bits, tbs, enable_64qam, nof_acks, cqi data_enable,
 * cqi type, I_offset_ack, I_offset_cqi, I_offset_ri, ri_len, cqi N} -> CRC verdict; the callback also sets the SNR of the channel estimate */
typedef int (*ul_script_fn)(void* user, const uint32_t* call16, float* snr_db, uint8_t* payload);
static ul_script_fn g_ul_script; static void* g_ul_user;
struct ul_call_rec { uint32_t w[16]; int crc; float snr; };
static std::vector<ul_call_rec> g_ul_calls;
static ul_call_rec g_pending; static int g_pending_crc; static std::vector<uint8_t> g_pending_payload;

extern "C" {
void srsran_enb_ul_fft(srsran_enb_ul_t*) {}
float srsran_vec_avg_power_cf(const cf_t* x, const uint32_t len) { float s = 0; for (uint32_t i = 0; i < len; i++) s += __real__ x[i] * __real__ x[i] + __imag__ x[i] * __imag__ x[i]; return len ? s / len : 0; }
int srsran_symbol_sz(uint32_t nof_prb) { return nof_prb <= 6 ? 128 : nof_prb <= 15 ? 256 : nof_prb <= 25 ? 384 : nof_prb <= 50 ? 768 : nof_prb <= 75 ? 1024 : 1536; }
uint32_t srsran_ri_nof_bits(const srsran_cell_t*) { return 1; }
int srsran_prach_init(srsran_prach_t*, uint32_t) { return 0; }
int srsran_prach_set_cfg(srsran_prach_t*, srsran_prach_cfg_t*, uint32_t) { return 0; }
void srsran_prach_set_detect_factor(srsran_prach_t*, float) {}
bool srsran_prach_tti_opportunity(srsran_prach_t*, uint32_t, int) { return false; }
int srsran_prach_detect_offset(srsran_prach_t*, uint32_t, cf_t*, uint32_t, uint32_t*, float*, float*, uint32_t* n) { *n = 0; return 0; }
/* the estimate runs first (decode_run :256): the script is asked here, once per attempt, and srsran_pusch_decode hands its verdict on */
int srsran_chest_ul_estimate_pusch(srsran_chest_ul_t*, srsran_ul_sf_cfg_t* sf, srsran_pusch_cfg_t* cfg, cf_t*, srsran_chest_ul_res_t* res)
{
  ul_call_rec& c = g_pending;
  memset(&c, 0, sizeof(c));
  const srsran_pusch_grant_t& g = cfg->grant;
  c.w[0] = sf->tti; c.w[1] = cfg->rnti; c.w[2] = g.L_prb; c.w[3] = g.n_prb[0]; c.w[4] = (uint32_t)g.tb.mcs_idx; c.w[5] = mod_bits(g.tb.mod); c.w[6] = (uint32_t)g.tb.tbs;
  c.w[7] = cfg->enable_64qam; c.w[8] = cfg->uci_cfg.ack[0].nof_acks; c.w[9] = cfg->uci_cfg.cqi.data_enable; c.w[10] = (uint32_t)cfg->uci_cfg.cqi.type;
  c.w[11] = cfg->uci_offset.I_offset_ack; c.w[12] = cfg->uci_offset.I_offset_cqi; c.w[13] = cfg->uci_offset.I_offset_ri; c.w[14] = cfg->uci_cfg.cqi.ri_len; c.w[15] = cfg->uci_cfg.cqi.N;
  g_pending_payload.assign(g.tb.tbs > 0 ? (size_t)g.tb.tbs / 8 : 0, 0);
  float snr = res->snr_db;
  g_pending_crc = g_ul_script(g_ul_user, c.w, &snr, g_pending_payload.data());
  res->snr_db = snr; res->ta_us = 0.0f; res->noise_estimate_dbm = -100.0f;
  c.snr = snr;
  return SRSRAN_SUCCESS;
}
int srsran_pusch_decode(srsran_pusch_t*, srsran_ul_sf_cfg_t*, srsran_pusch_cfg_t*, srsran_chest_ul_res_t*, cf_t*, srsran_pusch_res_t* out)
{
  out->crc = g_pending_crc != 0;
  if (out->crc && !g_pending_payload.empty()) memcpy(out->data, g_pending_payload.data(), g_pending_payload.size());
  g_pending.crc = g_pending_crc != 0;
  g_ul_calls.push_back(g_pending);
  return SRSRAN_SUCCESS;
}
}
namespace srsran {
bool rlc_am_is_control_pdu(uint8_t*) { return true; }
void rlc_am_read_data_pdu_header(uint8_t**, uint32_t*, rlc_amd_pdu_header_t*) {}
}
LIBLTE_ERROR_ENUM liblte_mme_parse_msg_sec_header(LIBLTE_BYTE_MSG_STRUCT*, uint8* pd, uint8* sec) { *pd = 0; *sec = LIBLTE_MME_SECURITY_HDR_TYPE_INTEGRITY_AND_CIPHERED; return LIBLTE_SUCCESS; }
LIBLTE_ERROR_ENUM liblte_mme_unpack_identity_response_msg(LIBLTE_BYTE_MSG_STRUCT*, LIBLTE_MME_ID_RESPONSE_MSG_STRUCT*) { return LIBLTE_ERROR_INVALID_INPUTS; }
LIBLTE_ERROR_ENUM liblte_mme_unpack_attach_request_msg(LIBLTE_BYTE_MSG_STRUCT*, LIBLTE_MME_ATTACH_REQUEST_MSG_STRUCT*) { return LIBLTE_ERROR_INVALID_INPUTS; }

struct ref_ul_t {
  srsran_cell_t cell;
  std::atomic<float> cfo{0.f};
  MCSTracking* mcs;
  UL_HARQ ul_harq;
  ULSchedule* ulsche;
  LTESniffer_pcap_writer pcap;
  srsran_enb_ul_t enb_ul;
  srsran_ul_sf_cfg_t ul_sf;
  srsran_ul_cfg_t ul_cfg;
  std::vector<cf_t> grid, buf0, buf1;
  cf_t* bufs[2]; cf_t* bufs_off[2];
  SubframePower* power;
  PUSCH_Decoder* dec;
  std::vector<DCI_UL> dci, rar;
};

extern "C" {
void ref_ul_set_script(void* fn, void* user) { g_ul_script = (ul_script_fn)fn; g_ul_user = user; }
ref_ul_t* ref_ul_new(uint32_t nof_prb, uint32_t cell_id)
{
  const uint8_t fill = g_fill;
  g_fill = 0;
  ref_ul_t* h = new ref_ul_t();
  memset(&h->cell, 0, sizeof(h->cell));
  h->cell.nof_prb = nof_prb; h->cell.nof_ports = 1; h->cell.id = cell_id;
  char cwd[4096];
  char* here = getcwd(cwd, sizeof(cwd));
  if (chdir("/tmp") != 0) here = nullptr;   /* MCSTracking's constructor opens mcs_statistic.csv in the working directory */
  h->mcs = new MCSTracking(1, 0, false, UL_MODE, -1, h->cfo);
  if (here && chdir(here) != 0) { }
  h->ulsche = new ULSchedule(0, &h->ul_harq, false);
  h->ulsche->set_multi_offset(UL_MODE);
  memset(&h->enb_ul, 0, sizeof(h->enb_ul)); memset(&h->ul_sf, 0, sizeof(h->ul_sf)); memset(&h->ul_cfg, 0, sizeof(h->ul_cfg));
  h->grid.assign((size_t)14 * 12 * nof_prb, cf_t());
  for (auto& v : h->grid) { __real__ v = 1.0f; __imag__ v = 0.0f; }   /* a flat grid: 0 dB in every PRB */
  h->buf0.assign(3 * 15 * 1536, cf_t()); h->buf1 = h->buf0;
  h->bufs[0] = h->buf0.data(); h->bufs[1] = h->buf1.data(); h->bufs_off[0] = h->buf0.data(); h->bufs_off[1] = h->buf1.data();
  h->enb_ul.cell = h->cell; h->enb_ul.sf_symbols = h->grid.data();
  h->power = new SubframePower(h->cell);
  h->dec = new PUSCH_Decoder(h->enb_ul, h->ul_sf, h->ulsche, h->bufs, h->bufs_off, h->ul_cfg, &h->pcap, h->mcs, false);   /* SubframeWorker.cc:45-55 */
  g_fill = fill;
  return h;
}
void ref_ul_free(ref_ul_t* h) { if (!h) return; delete h->dec; delete h->power; delete h->ulsche; delete h->mcs; delete h; }
/* the estimator's last result is a member of enb_ul: it persists between attempts, as in the reference (the SNR gate reads whatever the last estimate left) */
void ref_ul_set_last_snr(ref_ul_t* h, float snr_db) { h->enb_ul.chest_res.snr_db = snr_db; }
/* one subframe of PUSCH_Decoder::decode: n entries x 12 words {rnti, is_rar, mcs_idx, L_prb, n_prb0, modulation bits, tbs, L_prb of the 256QAM-table grant, its modulation
 * bits, its tbs, cqi_request, nof_ack}; entries with is_rar go to the RAR list, the others to the DCI 0 list - an empty list is handed over as a null pointer, the way
 * ULSchedule hands them out */
void ref_ul_decode(ref_ul_t* h, uint32_t tti, uint32_t n, const uint32_t* e12)
{
  g_ul_calls.clear(); g_pcap.clear();
  h->dci.clear(); h->rar.clear();
  auto mod_of = [](uint32_t bits) { return bits == 2 ? SRSRAN_MOD_QPSK : bits == 4 ? SRSRAN_MOD_16QAM : bits == 6 ? SRSRAN_MOD_64QAM : bits == 8 ? SRSRAN_MOD_256QAM : SRSRAN_MOD_BPSK; };
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t* e = e12 + 12 * i;
    DCI_UL d;
    d.rnti = (uint16_t)e[0]; d.is_rar_gant = (int)e[1]; d.nof_ack = (int)e[11];
    d.ran_ul_dci = std::make_shared<srsran_dci_ul_t>(); memset(d.ran_ul_dci.get(), 0, sizeof(srsran_dci_ul_t));
    d.ran_ul_dci->rnti = d.rnti; d.ran_ul_dci->cqi_request = e[10] != 0;
    d.ran_ul_grant = std::make_shared<srsran_pusch_grant_t>(); memset(d.ran_ul_grant.get(), 0, sizeof(srsran_pusch_grant_t));
    d.ran_ul_grant_256 = std::make_shared<srsran_pusch_grant_t>(); memset(d.ran_ul_grant_256.get(), 0, sizeof(srsran_pusch_grant_t));
    d.ran_ul_grant->L_prb = e[3]; d.ran_ul_grant->n_prb[0] = d.ran_ul_grant->n_prb[1] = e[4]; d.ran_ul_grant->tb.mcs_idx = (int)e[2]; d.ran_ul_grant->tb.mod = mod_of(e[5]); d.ran_ul_grant->tb.tbs = (int)e[6];
    d.ran_ul_grant_256->L_prb = e[7]; d.ran_ul_grant_256->n_prb[0] = d.ran_ul_grant_256->n_prb[1] = e[4]; d.ran_ul_grant_256->tb.mcs_idx = (int)e[2]; d.ran_ul_grant_256->tb.mod = mod_of(e[8]); d.ran_ul_grant_256->tb.tbs = (int)e[9];
    (d.is_rar_gant ? h->rar : h->dci).push_back(d);
  }
  memset(&h->ul_sf, 0, sizeof(h->ul_sf));
  srsran_ul_sf_cfg_t sf; memset(&sf, 0, sizeof(sf)); sf.tti = tti;
  h->dec->init_pusch_decoder(h->dci.empty() ? nullptr : &h->dci, h->rar.empty() ? nullptr : &h->rar, sf, h->power);
  h->dec->decode();
}
uint32_t ref_ul_calls(uint32_t* out18, uint32_t cap)
{
  for (uint32_t i = 0; i < g_ul_calls.size() && i < cap; i++) { uint32_t* o = out18 + 18 * i; memcpy(o, g_ul_calls[i].w, 64); o[16] = (uint32_t)g_ul_calls[i].crc; memcpy(o + 17, &g_ul_calls[i].snr, 4); }
  return (uint32_t)g_ul_calls.size();
}
int ref_ul_tracked(ref_ul_t* h, uint16_t rnti) { return (int)h->mcs->find_tracking_info_RNTI_ul(rnti); }   /* (refreshes the entry's time stamp, like every look-up) */
uint32_t ref_ul_nof_tracked(ref_ul_t* h) { return (uint32_t)h->mcs->nof_RNTI_member_ul(); }
void ref_ul_update_database(ref_ul_t* h) { h->mcs->update_database_ul(); }
/* RRCConnectionSetup seen on the downlink (SubframeWorker.cc:299-347 -> update_ue_config_rnti): betaOffset indices and the aperiodic report type of one RNTI */
void ref_ul_set_ue_config(ref_ul_t* h, uint16_t rnti, uint32_t i_ack, uint32_t i_cqi, uint32_t i_ri, uint32_t cqi_type)
{
  ltesniffer_ue_spec_config_t c = h->mcs->get_ue_config_rnti(rnti);
  c.has_ue_config = true; c.uci_config.I_offset_ack = i_ack; c.uci_config.I_offset_cqi = i_cqi; c.uci_config.I_offset_ri = i_ri; c.cqi_config.type = (srsran_cqi_type_t)cqi_type;
  h->mcs->update_ue_config_rnti(rnti, c);
}
}
