/* oracle/_ref harness around the REFERENCE'S OWN downlink decode control flow (test infrastructure, NOT product; see srsran/standin.h, srsran/standin_l2.h).
 *
 * Compiled verbatim from /root/reference by oracle/Makefile.ref into _ref/libref_falcon_decode.so:
 *   src/src/DL_Sniffer_PDSCH.cc     PDSCH_Decoder::decode_dl_mode (:881-1291): the gate, known-table decode or the 64QAM-then-256QAM trial, HARQ verdicts, which transport
 *                                   blocks become pcap records, RAR -> RNTI manager + tracking database, RRCConnectionSetup -> UE configuration, what goes back into
 *                                   MCSTracking; run_api_dl_mode (:804-879)
 *   + everything of libref_falcon_collect.so (DCICollection.cc, falcon_dci.c, dl_sniffer_pdsch.c, ul_sniffer_pusch.c, ULSchedule.cc, MCSTracking.cc, HARQ.cc) and the
 *     RNTI manager (lib/src/util/RNTIManager.cc, Histogram.cc, Interval.cc): the objects the decoder talks to are the reference's own
 * What is NOT the reference here:
 *   - srsran_ue_dl_decode_pdsch: the PDSCH decoder itself (srsRAN).  Every call is RECORDED (RNTI, grant as configured, p_a) and answered by a callback of the test - a
 *     scripted decoder: CRC verdicts and payload bytes per transport block.  The same callback answers the oracle's decode calls (o_worker_set_script_decoder).
 *   - the byte parsers behind the stand-in MAC / RRC / NAS classes: bound to the oracle's (standin_l2.h).
 *   - LTESniffer_pcap_writer: a recorder (PcapWriter.cc's framing is pinned on the reference's own captures elsewhere: tests/test_pcap_golden.py).
 *   - the srsRAN resource-allocation functions: collect_glue.cc's (this file includes it).
 * Memory the reference reads without having written it - the grant of the table it did not compute (DL_Sniffer_PDSCH.cc:887-888 reads ran_pdsch_grant->tb[0].tbs and
 * both nof_tb for EVERY entry) - comes from this library's operator new, filled with a byte the test chooses: 0x01 ("a positive size, not two blocks": what a recycled
 * heap block typically holds) or 0x00. */
#include "collect_glue.cc"
#include "include/DL_Sniffer_PDSCH.h"
#include "../lsn_oracle.h"
#include <new>

static uint8_t g_fill = 0x01;
void* operator new(size_t n) { void* p = malloc(n ? n : 1); if (!p) throw std::bad_alloc(); memset(p, g_fill, n); return p; }
void* operator new[](size_t n) { void* p = malloc(n ? n : 1); if (!p) throw std::bad_alloc(); memset(p, g_fill, n); return p; }
void operator delete(void* p) noexcept { free(p); }
void operator delete[](void* p) noexcept { free(p); }
void operator delete(void* p, size_t) noexcept { free(p); }
void operator delete[](void* p, size_t) noexcept { free(p); }

/* ---- bound from the test: the oracle's parsers ---- */
typedef int (*o_mac_fn)(const uint8_t*, int, o_mac_subh_t*, int);
typedef int (*o_setup_fn)(const uint8_t*, int, o_ue_cfg_t*);
typedef int (*o_rar_fn)(const o_cell_t*, const uint8_t*, int, o_rar_t*, int);
typedef int (*o_paging_fn)(const uint8_t*, int, o_paging_id_t*, int);
typedef int (*o_reconf_fn)(const uint8_t*, int, uint32_t*);
typedef int (*o_sib2_fn)(const uint8_t*, int, o_sib2_t*);
static o_mac_fn g_o_mac; static o_setup_fn g_o_setup; static o_rar_fn g_o_rar; static o_paging_fn g_o_paging; static o_reconf_fn g_o_reconf; static o_sib2_fn g_o_sib2;
static o_cell_t g_ocell;

static int l2_mac(const uint8_t* pdu, int len, uint32_t* out4, int cap) { return g_o_mac(pdu, len, (o_mac_subh_t*)out4, cap); }
static int l2_setup(const uint8_t* sdu, int len, uint32_t* w)
{
  static const float p_a_db[8] = {-6.0f, -4.77f, -3.0f, -1.77f, 0.0f, 1.0f, 2.0f, 3.0f};
  o_ue_cfg_t c;
  if (!g_o_setup(sdu, len, &c)) return 0;
  w[0] = 4;
  for (uint32_t i = 0; i < 8; i++) if (c.p_a == p_a_db[i]) w[0] = i;
  w[1] = c.i_offset_ack; w[2] = c.i_offset_cqi; w[3] = c.i_offset_ri;
  w[4] = 1; w[5] = c.cqi_type == 0 ? 0u : c.cqi_type == 1 ? 1u : 3u; /* wideband -> rm12, UE-selected sub-band -> rm20, higher-layer sub-band -> rm30 */
  return 1;
}
static int l2_rar(const uint8_t* pdu, int len, uint32_t* w, int cap)
{
  o_rar_t r[32];
  const int n = g_o_rar(&g_ocell, pdu, len, r, cap < 32 ? cap : 32);
  for (int i = 0; i < n; i++) {
    w[3 * i] = r[i].t_crnti; w[3 * i + 1] = r[i].ta;
    w[3 * i + 2] = (r[i].hopping << 19) | (r[i].riv << 9) | (r[i].mcs << 5) | (r[i].tpc << 2) | (r[i].ul_delay << 1) | r[i].csi_req;
  }
  return n;
}
static int l2_paging(const uint8_t* pdu, int len, uint32_t* w, int cap)
{
  o_paging_id_t r[16];
  const int n = g_o_paging(pdu, len, r, cap < 16 ? cap : 16);
  for (int i = 0; i < n; i++) {
    uint32_t* q = w + 18 * i;
    q[0] = r[i].is_imsi; q[1] = r[i].nof_digits; q[2] = r[i].m_tmsi;
    for (int k = 0; k < 15; k++) q[3 + k] = r[i].digits[k];
  }
  return n;
}
static int l2_reconf(const uint8_t* sdu, int len, uint32_t* tmsi) { return g_o_reconf(sdu, len, tmsi) == 1 ? 1 : 0; }
static int l2_sib2(const uint8_t* pdu, int len, uint32_t* w)
{
  o_sib2_t s;
  memset(&s, 0, sizeof(s));
  const int k = g_o_sib2(pdu, len, &s);
  memcpy(w, &s, 14 * sizeof(uint32_t));
  return k;
}
extern "C" {
int (*lsn_l2_mac_parse)(const uint8_t*, int, uint32_t*, int) = l2_mac;
int (*lsn_l2_conn_setup)(const uint8_t*, int, uint32_t*) = l2_setup;
int (*lsn_l2_rar_parse)(const uint8_t*, int, uint32_t*, int) = l2_rar;
int (*lsn_l2_paging)(const uint8_t*, int, uint32_t*, int) = l2_paging;
int (*lsn_l2_reconfig_tmsi)(const uint8_t*, int, uint32_t*) = l2_reconf;
int (*lsn_l2_sib2)(const uint8_t*, int, uint32_t*) = l2_sib2;
}
/* the NAS PDU the stand-in DL-DCCH message hands over is {0x07, attach accept, -, -, M-TMSI} (standin_l2.h): the three liblte calls read it back */
LIBLTE_ERROR_ENUM liblte_mme_parse_msg_header(LIBLTE_BYTE_MSG_STRUCT* msg, uint8* pd, uint8* msg_type) { *pd = msg->msg[0]; *msg_type = msg->msg[1]; return LIBLTE_SUCCESS; }
LIBLTE_ERROR_ENUM liblte_mme_unpack_attach_accept_msg(LIBLTE_BYTE_MSG_STRUCT* msg, LIBLTE_MME_ATTACH_ACCEPT_MSG_STRUCT* a)
{
  a->guti_present = true;
  a->guti.guti.m_tmsi = ((uint32_t)msg->msg[4] << 24) | ((uint32_t)msg->msg[5] << 16) | ((uint32_t)msg->msg[6] << 8) | msg->msg[7];
  return LIBLTE_SUCCESS;
}
LIBLTE_ERROR_ENUM liblte_mme_unpack_activate_default_eps_bearer_context_request_msg(LIBLTE_BYTE_MSG_STRUCT*, LIBLTE_MME_ACTIVATE_DEFAULT_EPS_BEARER_CONTEXT_REQUEST_MSG_STRUCT*) { return LIBLTE_SUCCESS; }

/* ---- the scripted PDSCH decoder ---- */
/* one decode call as the reference configured it: 16 words {tti, rnti, nof_re, tx_scheme, pmi, nof_layers, then per block: enabled, modulation bits, tbs, rv, cw_idx} */
typedef int (*script_fn)(void* user, const uint32_t* call16, float p_a, uint8_t* payload0, uint8_t* payload1, int32_t* crc2);
static script_fn g_script; static void* g_script_user;
struct call_rec { uint32_t w[16]; float p_a; int crc[2]; };
static std::vector<call_rec> g_calls;
struct pcap_rec { uint32_t kind, tti, rnti, len, crc_ok; uint64_t hash; };
static std::vector<pcap_rec> g_pcap;
static uint32_t g_tti;

extern "C" {
uint8_t* srsran_vec_u8_malloc(uint32_t len) { return (uint8_t*)(::operator new[](len)); }
void srsran_vec_u8_zero(uint8_t* ptr, uint32_t n) { memset(ptr, 0, n); }
void srsran_softbuffer_rx_reset_tbs(srsran_softbuffer_rx_t*, uint32_t) {}
int srsran_ue_dl_decode_pdsch(srsran_ue_dl_t*, srsran_dl_sf_cfg_t* sf, srsran_pdsch_cfg_t* cfg, srsran_pdsch_res_t data[SRSRAN_MAX_CODEWORDS])
{
  call_rec c;
  memset(&c, 0, sizeof(c));
  const srsran_pdsch_grant_t& g = cfg->grant;
  c.w[0] = sf->tti; c.w[1] = cfg->rnti; c.w[2] = g.nof_re; c.w[3] = (uint32_t)g.tx_scheme; c.w[4] = g.pmi; c.w[5] = g.nof_layers;
  for (int i = 0; i < 2; i++) {
    uint32_t* t = c.w + 6 + 5 * i;
    t[0] = g.tb[i].enabled; t[1] = g.tb[i].enabled ? mod_bits(g.tb[i].mod) : 0; t[2] = g.tb[i].enabled ? (uint32_t)g.tb[i].tbs : 0; t[3] = g.tb[i].enabled ? (uint32_t)g.tb[i].rv : 0; t[4] = g.tb[i].enabled ? g.tb[i].cw_idx : 0;
  }
  c.p_a = cfg->p_a;
  int32_t crc[2] = {0, 0};
  g_script(g_script_user, c.w, c.p_a, data[0].payload, data[1].payload, crc);
  for (int i = 0; i < 2; i++)
    if (g.tb[i].enabled && g.tb[i].tbs > 0) { data[i].crc = crc[i] != 0; c.crc[i] = crc[i] != 0; }
  g_calls.push_back(c);
  return SRSRAN_SUCCESS;
}
}

/* ---- LTESniffer_pcap_writer as a recorder (kinds: 1 dl_crnti, 2 dl_ranti, 3 dl_sirnti, 4 dl_pch, 5 dl_crnti_api, 6 dl_paging_api, 7 ul_crnti, 8 ul_crnti_api, 9 bch) ---- */
static void rec(uint32_t kind, const uint8_t* pdu, uint32_t len, uint32_t rnti, bool crc_ok, uint32_t tti)
{
  uint64_t h = 1469598103934665603ull;
  for (uint32_t i = 0; i < len; i++) { h ^= pdu[i]; h *= 1099511628211ull; }
  g_pcap.push_back(pcap_rec{kind, tti, rnti, len, crc_ok ? 1u : 0u, h});
}
void LTESniffer_pcap_writer::enable(bool) {}
void LTESniffer_pcap_writer::open(const std::string, const std::string, uint32_t) {}
void LTESniffer_pcap_writer::close() {}
void LTESniffer_pcap_writer::set_ue_id(uint16_t) {}
void LTESniffer_pcap_writer::write_dl_crnti(uint8_t* pdu, uint32_t n, uint16_t crnti, bool crc_ok, uint32_t tti, bool) { rec(1, pdu, n, crnti, crc_ok, tti); }
void LTESniffer_pcap_writer::write_dl_ranti(uint8_t* pdu, uint32_t n, uint16_t ranti, bool crc_ok, uint32_t tti) { rec(2, pdu, n, ranti, crc_ok, tti); }
void LTESniffer_pcap_writer::write_dl_sirnti(uint8_t* pdu, uint32_t n, bool crc_ok, uint32_t tti) { rec(3, pdu, n, SRSRAN_SIRNTI, crc_ok, tti); }
void LTESniffer_pcap_writer::write_dl_bch(uint8_t* pdu, uint32_t n, bool crc_ok, uint32_t tti) { rec(9, pdu, n, 0, crc_ok, tti); }
void LTESniffer_pcap_writer::write_dl_pch(uint8_t* pdu, uint32_t n, bool crc_ok, uint32_t tti) { rec(4, pdu, n, SRSRAN_PRNTI, crc_ok, tti); }
void LTESniffer_pcap_writer::write_ul_crnti(uint8_t* pdu, uint32_t n, uint16_t crnti, uint32_t tti) { rec(7, pdu, n, crnti, true, tti); }
void LTESniffer_pcap_writer::write_ul_crnti_api(uint8_t* pdu, uint32_t n, uint16_t crnti, uint32_t tti) { rec(8, pdu, n, crnti, true, tti); }
void LTESniffer_pcap_writer::write_dl_crnti_api(uint8_t* pdu, uint32_t n, uint16_t crnti, bool crc_ok, uint32_t tti, bool) { rec(5, pdu, n, crnti, crc_ok, tti); }
void LTESniffer_pcap_writer::write_dl_paging_api(uint8_t* pdu, uint32_t n, uint16_t rnti, bool crc_ok, uint32_t tti, bool) { rec(6, pdu, n, rnti, crc_ok, tti); }

/* ---- the harness ---- */
struct ref_decode_t {
  ref_collect_t* c;          /* cell, MCSTracking, HARQ, ULSchedule, the per-subframe DCICollection: collect_glue.cc */
  RNTIManager* rm;
  LTESniffer_pcap_writer pcap;
  PDSCH_Decoder* dec;
  srsran_ue_dl_t ue;
  falcon_ue_dl_t fq;
  srsran_ue_dl_cfg_t ue_dl_cfg;
};

extern "C" {

void ref_decode_bind(void* mac, void* setup, void* rar, void* paging, void* reconf, void* sib2)
{
  g_o_mac = (o_mac_fn)mac; g_o_setup = (o_setup_fn)setup; g_o_rar = (o_rar_fn)rar; g_o_paging = (o_paging_fn)paging; g_o_reconf = (o_reconf_fn)reconf; g_o_sib2 = (o_sib2_fn)sib2;
}
void ref_decode_set_script(void* fn, void* user) { g_script = (script_fn)fn; g_script_user = user; }
void ref_decode_set_fill(int byte) { g_fill = (uint8_t)byte; }

ref_decode_t* ref_decode_new(uint32_t nof_prb, uint32_t nof_ports, uint32_t cell_id, uint32_t cp, int mcs_tracking_mode, int harq_mode, int nof_antenna, int api_mode)
{
  const uint8_t fill = g_fill;
  g_fill = 0; /* the long-lived objects start from zeroed memory; the fill is for what addCandidate allocates per DCI */
  ref_decode_t* h = new ref_decode_t();
  h->c = ref_collect_new(nof_prb, nof_ports, cell_id, cp, mcs_tracking_mode, harq_mode, DL_MODE);
  if (api_mode >= 0) { delete h->c->mcs; char cwd[4096]; char* here = getcwd(cwd, sizeof(cwd)); if (chdir("/tmp") != 0) here = nullptr;
    h->c->mcs = new MCSTracking(mcs_tracking_mode, 0, false, DL_MODE, api_mode, h->c->cfo); if (here && chdir(here) != 0) { } }
  memset(&g_ocell, 0, sizeof(g_ocell));
  g_ocell.nof_prb = nof_prb; g_ocell.nof_ports = nof_ports; g_ocell.id = cell_id; g_ocell.phich_ng_x6 = 1; g_ocell.cp = cp;
  h->rm = new RNTIManager(9, RNTI_PER_SUBFRAME, 5);   /* PhyCommon.cc:11: nof_falcon_ue_all_formats (DCISearch.cc:84-95) = 9 */
  memset(&h->ue, 0, sizeof(h->ue));
  h->ue.cell = h->c->cell; h->ue.nof_rx_antennas = (uint32_t)nof_antenna;
  memset(&h->fq, 0, sizeof(h->fq));
  h->fq.q = &h->ue;
  memset(&h->ue_dl_cfg, 0, sizeof(h->ue_dl_cfg));
  h->dec = new PDSCH_Decoder(0, &h->pcap, h->c->mcs, *h->rm, h->c->harq, mcs_tracking_mode, harq_mode, nof_antenna);   /* SubframeWorker.cc:36-44 */
  g_fill = fill;
  return h;
}
void ref_decode_free(ref_decode_t* h)
{
  if (!h) return;
  delete h->dec; delete h->rm; ref_collect_free(h->c); delete h;
}
ref_collect_t* ref_decode_collect(ref_decode_t* h) { return h->c; }

/* after ref_collect_begin / ref_collect_add on ref_decode_collect(h): SubframeWorker::run_dl_mode (SubframeWorker.cc:205-231) */
void ref_decode_dl_mode(ref_decode_t* h, uint32_t sfn, uint32_t sf_idx)
{
  g_calls.clear(); g_pcap.clear();
  g_tti = sfn * 10 + sf_idx;
  h->dec->init_pdsch_decoder(&h->fq, &h->c->sf, &h->ue_dl_cfg, &h->c->coll->getDLSnifferDCI_DL(), sfn, sf_idx);
  h->dec->decode_dl_mode();
}
uint32_t ref_decode_calls(uint32_t* out19, uint32_t cap)
{
  for (uint32_t i = 0; i < g_calls.size() && i < cap; i++) {
    uint32_t* o = out19 + 19 * i;
    memcpy(o, g_calls[i].w, sizeof(g_calls[i].w));
    memcpy(o + 16, &g_calls[i].p_a, 4); o[17] = (uint32_t)g_calls[i].crc[0]; o[18] = (uint32_t)g_calls[i].crc[1];
  }
  return (uint32_t)g_calls.size();
}
uint32_t ref_decode_records(uint32_t* out7, uint32_t cap)
{
  for (uint32_t i = 0; i < g_pcap.size() && i < cap; i++) {
    uint32_t* o = out7 + 7 * i;
    o[0] = g_pcap[i].kind; o[1] = g_pcap[i].tti; o[2] = g_pcap[i].rnti; o[3] = g_pcap[i].len; o[4] = g_pcap[i].crc_ok; o[5] = (uint32_t)g_pcap[i].hash; o[6] = (uint32_t)(g_pcap[i].hash >> 32);
  }
  return (uint32_t)g_pcap.size();
}
/* state probes */
int ref_decode_table(ref_decode_t* h, uint16_t rnti) { return (int)h->c->mcs->find_tracking_info_RNTI_dl(rnti); } /* (refreshes the entry's time stamp, like every look-up) */
int ref_decode_rnti_reason(ref_decode_t* h, uint16_t rnti) { return (int)h->rm->getActivationReason(rnti); }
void ref_decode_ue_config(ref_decode_t* h, uint16_t rnti, float* p_a, uint32_t* out5)
{
  const ltesniffer_ue_spec_config_t c = h->c->mcs->get_ue_config_rnti(rnti);
  *p_a = c.p_a; out5[0] = c.has_ue_config; out5[1] = c.uci_config.I_offset_ack; out5[2] = c.uci_config.I_offset_cqi; out5[3] = c.uci_config.I_offset_ri; out5[4] = (uint32_t)c.cqi_config.type;
}
uint32_t ref_decode_nof_tracked(ref_decode_t* h) { return (uint32_t)h->c->mcs->nof_RNTI_member_dl(); }
void ref_decode_update_database(ref_decode_t* h) { h->c->mcs->update_database_dl(); }
void ref_decode_step_time(ref_decode_t* h) { h->rm->stepTime(); }

} /* extern "C" */
