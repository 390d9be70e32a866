/* o_worker.c - ORACLE (test infrastructure only): one subframe through the LTESniffer downlink worker with
 * strictly sequential-subframe semantics.
 *   SubframeWorker::work / run_dl_mode      /root/reference/src/src/SubframeWorker.cc:142-235
 *   DCISearch::search / recursive / inspect /root/reference/src/src/DCISearch.cc:102-578
 *   falcon location map + CCE power         /root/reference/lib/src/phy/falcon_phch/falcon_pdcch.c:110-170,321-367,561-620
 *   DCIMetaFormats::update_formats          /root/reference/src/src/MetaFormats.cc:41-89
 *   DCICollection::addCandidate             /root/reference/src/src/DCICollection.cc:97-298
 *   srsran_dci_msg_to_trace_timestamp       /root/reference/lib/src/phy/falcon_phch/falcon_dci.c:148-352
 *   PDSCH_Decoder::decode_dl_mode           /root/reference/src/src/DL_Sniffer_PDSCH.cc:881-1291 (+ :611-632, :782-797, :1398-1418)
 *   MCSTracking (DL table learning)         /root/reference/src/src/MCSTracking.cc:758-848,1269-1291
 *   evergreen / forbidden setup             /root/reference/src/src/LTESniffer_Core.cc:398-417
 * Deviations that any deterministic restatement needs (DESIGN.md "determinism"): wall-clock / clock() fields are
 * dropped; grants that the reference leaves uninitialised (new srsran_pdsch_grant_t without the table's
 * dci_to_grant call, DL_Sniffer_PDSCH.cc:887 reads them) are defined as "not computed" and the decode gate is
 * evaluated on the grant that is actually used; the p-a of a UE comes from its RRCConnectionSetup (o_rrc.c), until then the
 * MCSTracking default 0 dB (MCSTracking.cc:1536). */
#include "lsn_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t L, ncce; uint8_t used, occupied, checked, sufficient_power; float power; } floc_t;
typedef struct { floc_t* location[4]; float power; } cce_map_t;
typedef struct { uint8_t payload[O_DCI_MAX_BITS]; uint32_t nof_bits; int format; uint16_t rnti; } dci_msg_t;
typedef struct { uint16_t rnti; dci_msg_t msg; uint32_t match; } cand_t;
typedef struct { int format; uint32_t global_index; uint32_t hits; } meta_t;
typedef struct { uint16_t rnti; uint32_t L, ncce; int format; cand_t cand; } temp_dci0_t;

typedef struct { /* DL_Sniffer_DCI_DL */
  uint16_t rnti; int format; uint32_t nof_bits, L, ncce, histval; int mcs_table;
  o_dci_dl_t dci; uint16_t dci_rnti; /* ran_dci_dl->rnti, zeroed on grant failure */
  o_pdsch_grant_t g64, g256; int has64, has256;
  int check; /* DCI_BASE::check: a size was taken from the HARQ database */
} dl_entry_t;
typedef struct { uint16_t rnti; uint32_t nof_bits, L, ncce, histval; o_dci_ul_t dci; o_pusch_grant_t g, g256; int ok; } ul_entry_t;
/* one entry of ULSchedule (DCI_UL): grants of both UL MCS tables */
typedef struct { uint16_t rnti; o_pusch_grant_t g, g256; uint32_t n_dmrs, hopping; int is_rar; uint32_t nof_ack, cqi_req; } ulg_t;
typedef struct { uint32_t tti; int valid; int n; ulg_t g[96]; } ulslot_t;

typedef struct {
  uint8_t present, has_rar; uint16_t nof_msg_after_rar; uint8_t table;
  /* database ageing (MCSTracking.cc:850-927): last look-up in subframes processed, activity / success / MIMO-error counters */
  uint32_t time, nof_active, nof_success_mgs, nof_unsupport_mimo, nof_pinfo, nof_other_mimo;
} mcs_entry_t;

/* ---------------- HARQ (src/include/HARQ.h, src/src/HARQ.cc) ----------------
 * 150 entities from the constructor + 150 from init_HARQ(harq_mode) (HARQ.cc:15-20,48-52), 8 processes x 2 transport blocks each; the soft buffer of
 * a (RNTI, process, TB) is allocated when first used.  clock() is replaced by the subframe count; the per-TB mutexes only
 * matter between worker threads (DL_SNIFFER_HARQ_BUSY cannot occur in the sequential order this oracle defines). */
#define O_HARQ_ENTITIES 300
enum { O_HARQ_NEW_TX = 0, O_HARQ_RE_TX, O_HARQ_FULL_BUFFER, O_HARQ_DECODED, O_HARQ_BUSY };
typedef struct { int last_decoded, ndi, rv, tbs, is_first; } o_harq_grant_t;
typedef struct { uint32_t sfn, sf_idx; o_harq_grant_t grant; int16_t* acc; uint8_t* keep; /* per code block: passed flag + decoded bits (cb_crc / data of softbuffer_rx) */ } o_harq_tb_t;
struct o_harq_entity { uint16_t rnti; uint32_t time; uint32_t nof_active, nof_success, nof_retx_success, nof_retx[8]; o_harq_tb_t tb[8][2]; };

struct o_worker {
  o_worker_cfg_t cfg;
  o_regs_t regs;
  o_rntiman_t* rm;
  meta_t all[O_NOF_FORMATS];
  meta_t* primary[O_NOF_FORMATS];
  meta_t* secondary[O_NOF_FORMATS];
  uint32_t nprimary, nsecondary;
  o_pcap_t* pcap;
  o_stats_t stats;
  mcs_entry_t* mcs; /* [65536] */
  uint32_t mcs_count;
  uint32_t sf_count;          /* subframes worked so far = the database's clock (1 subframe = 1 ms instead of clock()) */
  uint32_t mcs_update_period; /* get_interval() x 1000 subframes (LTESniffer_Core.cc:473-485), 0 = never */
  uint32_t mcs_interval;      /* seconds, MCSTracking.h:162 */
  uint32_t nof_mcs_updates;
  int harq_mode;                    /* -m harq: DL HARQ soft combining (HARQ.cc; off in the reference: ArgManager.cc:50,211-213) */
  struct o_harq_entity* harq;       /* [O_HARQ_ENTITIES] */
  int harq_nof_aval;
  uint32_t harq_stats[5];           /* verdicts of is_retransmission so far: NEW_TX, RE_TX, FULL_BUFFER, DECODED, BUSY */
  int second_turbo, second_viterbi; /* second-opinion decoders (o_second.c) instead of the production restatement */
  o_ue_cfg_t* uecfg; /* [65536] ue_spec_config of the tracking-database entries (MCSTracking.h:37-43) */
  o_ue_cfg_t default_cfg;
  int has_default_cfg;
  /* per-subframe state */
  ocf_t *grid, *ce;
  float* llr;
  o_chest_res_t chest;
  uint32_t cfi, sf_idx, sfn;
  float rb_power[O_MAX_PRB];
  temp_dci0_t temp0[64];
  uint32_t ntemp0;
  dl_entry_t dl[64];
  uint32_t ndl;
  ul_entry_t ul[64];
  uint32_t nul;
  uint16_t rb_map_dl[O_MAX_PRB], rb_map_ul[O_MAX_PRB];
  int dl_collision, ul_collision;
  uint32_t acc[64][6];
  uint32_t nacc;
  int16_t *llr0, *llr1;
  uint8_t* payload;
  uint64_t total_iters, algo_bytes;
  int records;
  /* UL mode */
  int ul_mode;
  o_ul_cfg_t ulcfg;
  int api_mode; o_pcap_t* api_pcap; o_api_event_t* api_ev; int api_n, api_cap; /* security-API sink (run_api_dl_mode) */
  int ul_configured;   /* ULSchedule::get_config */
  o_sib2_t sib2;       /* ULSchedule::sib2, valid when the configuration was learned */
  int sib2_learned;
  ocf_t* ul_grid;
  ulslot_t* ul_sched;  /* [16] DCI-0 grants by tti % 16 (ULSchedule::pushULSche) */
  ulslot_t* rar_sched; /* [16] RAR grants */
  uint8_t* ulmod;      /* [65536] 0 absent, 1 unknown, 2 16QAM max, 3 64QAM max, 4 256QAM max (MCSTracking UL) */
  uint32_t ulmod_count;
  uint32_t *ul_time, *ul_active, *ul_success; /* [65536] ul_sniffer_tracking_t::time (subframes), nof_active, nof_success_mgs */
  float last_ul_snr;  /* enb_ul.chest_res.snr_db: the estimate of the most recent PUSCH attempt (UL_Sniffer_PUSCH.cc:572) */
  uint32_t nof_ul_updates;
  /* scripted decoder (tests/test_ref_decode.py): when set, a decode call is described to this function instead of being demodulated and decoded */
  int (*script_fn)(void* user, const uint32_t* call16, float p_a, uint8_t* payload0, uint8_t* payload1, int32_t* crc2);
  void* script_user;
  /* scripted uplink decoder (tests/test_ref_ul_decode.py): one PUSCH attempt is described to this function - 16 words {tti, rnti, L_prb, n_prb, mcs_idx, modulation bits
   * tried, tbs, -, nof_ack, CSI requested, report type, I_offset_ack, I_offset_cqi, I_offset_ri, RI bits, CQI bits} - which answers with the CRC verdict and the SNR the
   * estimator would report */
  int (*ul_script_fn)(void* user, const uint32_t* call16, float* snr_db, uint8_t* payload);
  void* ul_script_user;
};


/* MetaFormats.cc:41-89 */
static void update_formats(o_worker_t* w)
{
  meta_t* sorted[O_NOF_FORMATS];
  double total = 0;
  int n = O_NOF_FORMATS;
  for (int i = 0; i < n; i++) {
    sorted[i] = &w->all[i];
    total += sorted[i]->hits;
  }
  for (int i = 0; i < n - 1; i++) {
    int mx = i;
    for (int j = mx; j < n; j++)
      if (sorted[j]->hits > sorted[mx]->hits) mx = j;
    meta_t* d = sorted[i];
    sorted[i] = sorted[mx];
    sorted[mx] = d;
  }
  double thr = total * w->cfg.split_ratio, cum = 0;
  w->nprimary = w->nsecondary = 0;
  for (int i = 0; i < n; i++) {
    if (cum <= thr)
      w->primary[w->nprimary++] = sorted[i];
    else
      w->secondary[w->nsecondary++] = sorted[i];
    cum += sorted[i]->hits;
    sorted[i]->hits = 0;
  }
}

o_worker_t* o_worker_new(const o_worker_cfg_t* cfg)
{
  if (o_fft_size(cfg->cell.nof_prb) < 0 || cfg->cell.nof_ports < 1 || cfg->cell.nof_ports == 3 || cfg->cell.nof_ports > 4 || cfg->nof_rx < 1 ||
      cfg->nof_rx > O_MAX_RX)
    return NULL;
  o_worker_t* w = (o_worker_t*)calloc(1, sizeof(*w));
  w->cfg = *cfg;
  if (w->cfg.max_turbo_iter <= 0) w->cfg.max_turbo_iter = 12; /* SubframeWorker.cc:365 */
  o_regs_init(&cfg->cell, &w->regs);
  w->rm = o_rntiman_new(O_NOF_FORMATS, 304 / 5, cfg->histogram_threshold); /* PhyCommon.cc:11 */
  /* LTESniffer_Core.cc:402-417 */
  o_rntiman_add_evergreen(w->rm, O_RARNTI_START, O_RARNTI_END, O_FMT1A);
  o_rntiman_add_evergreen(w->rm, O_PRNTI, O_SIRNTI, O_FMT1A);
  o_rntiman_add_evergreen(w->rm, O_RARNTI_START, O_RARNTI_END, O_FMT1C);
  o_rntiman_add_evergreen(w->rm, O_PRNTI, O_SIRNTI, O_FMT1C);
  for (uint32_t f = 0; f < O_NOF_FORMATS; f++) o_rntiman_add_forbidden(w->rm, 0, 0, f);
  for (int i = 0; i < O_NOF_FORMATS; i++) {
    w->all[i].format = i;
    w->all[i].global_index = (uint32_t)i;
    w->all[i].hits = 0;
  }
  update_formats(w);
  size_t nre = 12u * cfg->cell.nof_prb;
  w->grid = (ocf_t*)calloc(cfg->nof_rx * 14u * nre, sizeof(ocf_t));
  w->ce = (ocf_t*)calloc((size_t)cfg->cell.nof_ports * cfg->nof_rx * 14u * nre, sizeof(ocf_t));
  w->llr = (float*)calloc(8 * 800, sizeof(float));
  w->mcs = (mcs_entry_t*)calloc(65536, sizeof(mcs_entry_t));
  w->uecfg = (o_ue_cfg_t*)calloc(65536, sizeof(o_ue_cfg_t));
  /* MCSTracking::set_default_of_default_config (:1531-1540) */
  w->default_cfg.p_a = 0.0f;
  w->default_cfg.i_offset_ack = 10;
  w->default_cfg.i_offset_cqi = 8;
  w->default_cfg.i_offset_ri = 11;
  w->default_cfg.cqi_type = 2;
  w->api_mode = -1; /* ArgManager.cc:63 */
  w->mcs_interval = 5; /* MCSTracking.h:162 */
  w->mcs_update_period = 5000;
  w->llr0 = (int16_t*)calloc(14u * nre * 8u, sizeof(int16_t));
  w->llr1 = (int16_t*)calloc(14u * nre * 8u, sizeof(int16_t));
  w->payload = (uint8_t*)calloc(32768, 1);
  return w;
}

void o_worker_free(o_worker_t* w)
{
  if (!w) return;
  o_rntiman_free(w->rm);
  free(w->grid); free(w->ce); free(w->llr); free(w->mcs); free(w->uecfg); free(w->api_ev); free(w->llr0); free(w->llr1); free(w->payload);
  free(w->ul_grid); free(w->ul_sched); free(w->rar_sched); free(w->ulmod); free(w->ul_time); free(w->ul_active); free(w->ul_success);
  if (w->harq) {
    for (int i = 0; i < O_HARQ_ENTITIES; i++)
      for (int p = 0; p < 8; p++)
        for (int t = 0; t < 2; t++) { free(w->harq[i].tb[p][t].acc); free(w->harq[i].tb[p][t].keep); }
    free(w->harq);
  }
  free(w);
}
void o_worker_set_pcap(o_worker_t* w, o_pcap_t* p) { w->pcap = p; }
const o_stats_t* o_worker_stats(o_worker_t* w) { return &w->stats; }
const ocf_t* o_worker_grid(o_worker_t* w) { return w->grid; }
const ocf_t* o_worker_ce(o_worker_t* w) { return w->ce; }
const float* o_worker_llr(o_worker_t* w, uint32_t* n) { if (n) *n = w->regs.nof_cce[w->cfi - 1] * 72; return w->llr; }
const o_chest_res_t* o_worker_chest(o_worker_t* w) { return &w->chest; }
uint32_t o_worker_cfi(o_worker_t* w) { return w->cfi; }
const float* o_worker_rb_power(o_worker_t* w) { return w->rb_power; }
o_rntiman_t* o_worker_rntiman(o_worker_t* w) { return w->rm; }
uint64_t o_worker_total_iters(o_worker_t* w) { return w->total_iters; }
uint64_t o_worker_algo_bytes(o_worker_t* w) { return w->algo_bytes; }
uint32_t o_worker_accepted(o_worker_t* w, uint32_t* out6, uint32_t max)
{
  uint32_t n = w->nacc < max ? w->nacc : max;
  memcpy(out6, w->acc, n * 6 * sizeof(uint32_t));
  return w->nacc;
}

/* ---------------- MCSTracking, DL part ---------------- */
static int mcs_find(o_worker_t* w, uint16_t rnti) /* MCSTracking.cc:758-782 */
{
  if (!w->mcs[rnti].present) return w->mcs_count < 250 ? O_TABLE_UNKNOWN : O_TABLE_FULL;
  w->mcs[rnti].time = w->sf_count; /* :778-779 */
  return w->mcs[rnti].table;
}
static void mcs_add(o_worker_t* w, uint16_t rnti)
{
  if (!w->mcs[rnti].present) {
    memset(&w->mcs[rnti], 0, sizeof(mcs_entry_t));
    w->mcs[rnti].present = 1;
    w->mcs[rnti].time = w->sf_count;
    w->mcs[rnti].table = O_TABLE_UNKNOWN;
    w->mcs_count++;
    w->uecfg[rnti] = w->default_cfg; /* add_RNTI_dl, MCSTracking.cc:785-795 */
    w->uecfg[rnti].has_ue_config = 0;
  }
}
/* MCSTracking::get_ue_config_rnti (:1482-1516): the entry's configuration, or the default for an RNTI without entry */
static o_ue_cfg_t ue_cfg_get(o_worker_t* w, uint16_t rnti)
{
  if (w->ul_mode ? (w->ulmod && w->ulmod[rnti]) : w->mcs[rnti].present) return w->uecfg[rnti]; /* UL_MODE: tracking_database_ul_mode, :1466-1479,1503-1513 */
  o_ue_cfg_t c = w->default_cfg;
  c.has_ue_config = 0;
  return c;
}
/* MCSTracking::add_RNTI_ul (MCSTracking.cc:57-69): time stamp, modulation, a copy of the default configuration, counters at zero */
static void ul_add(o_worker_t* w, uint16_t rnti, int mod)
{
  if (w->ulmod[rnti]) return; /* std::map::insert keeps an existing entry */
  w->ulmod[rnti] = (uint8_t)mod;
  w->ulmod_count++;
  w->ul_time[rnti] = w->sf_count;
  w->ul_active[rnti] = w->ul_success[rnti] = 0;
  w->uecfg[rnti] = w->default_cfg;
  w->uecfg[rnti].has_ue_config = 0;
}
/* a decoded C-RNTI transport block: sch_pdu walk, every CCCH SDU (LCID 0) is tried as RRCConnectionSetup; the first one ever
 * seen also becomes the default of RNTIs without entry (DL_Sniffer_PDSCH.cc:1041-1070).  any_lcid: the 64QAM-table attempt of the unknown-table branch
 * tries EVERY SDU, whatever its logical channel (:1140 has no LCID test) - found by the pin on the compiled reference (tests/test_ref_decode.py) */
static void learn_conn_setup(o_worker_t* w, const uint8_t* pdu, int len, uint16_t rnti, int any_lcid)
{
  o_mac_subh_t sub[20];
  const int n = o_mac_dlsch_parse(pdu, len, sub, 20);
  for (int i = 0; i < n; i++) {
    if (!(sub[i].is_sdu && (any_lcid || sub[i].lcid == 0))) continue;
    o_ue_cfg_t c;
    if (!o_rrc_conn_setup_decode(pdu + sub[i].off, (int)sub[i].len, &c)) continue;
    if (!w->has_default_cfg) {
      w->default_cfg = c; /* update_default_ue_config, :1518-1529 */
      w->has_default_cfg = 1;
    }
    if (w->ul_mode) { /* update_ue_config_rnti, UL_MODE branch (:1464-1479): add_RNTI_ul(UNKNOWN_MOD) copies the default first */
      ul_add(w, rnti, 1);
    } else {
      mcs_add(w, rnti); /* update_ue_config_rnti, :1446-1463 */
    }
    w->uecfg[rnti] = c;
  }
}
static void mcs_update(o_worker_t* w, uint16_t rnti, int table) /* MCSTracking.cc:797-825 */
{
  mcs_entry_t* e = &w->mcs[rnti];
  if (e->present) {
    if (e->has_rar) {
      if (e->nof_msg_after_rar > 3) {
        e->table = (uint8_t)table;
        e->has_rar = 0;
      } else {
        e->table = O_TABLE_UNKNOWN;
      }
    } else {
      e->table = (uint8_t)table;
    }
  } else {
    mcs_add(w, rnti);
  }
}
static void mcs_rar(o_worker_t* w, uint16_t crnti) /* MCSTracking.cc:827-848 */
{
  mcs_add(w, crnti);
  w->mcs[crnti].has_rar = 1;
  w->mcs[crnti].table = O_TABLE_UNKNOWN;
}
/* MCSTracking::update_statistic_dl, MCSTracking.cc:1269-1400, without the HARQ branches (harq_mode 0: every transmission is NEW_TX) */
static void mcs_statistic(o_worker_t* w, uint16_t rnti, int format, int table, const int* tb_en, const int* success, int mimo_ret)
{
  mcs_add(w, rnti);
  mcs_entry_t* e = &w->mcs[rnti];
  if (format > O_FMT1A && e->has_rar) e->nof_msg_after_rar++;
  if (table == O_TABLE_64QAM || table == O_TABLE_256QAM || table == O_TABLE_UNKNOWN)
    for (int i = 0; i < 2; i++) {
      if (tb_en[i]) e->nof_active++;
      if (success[i]) e->nof_success_mgs++;
      if (mimo_ret == -1 && tb_en[i]) e->nof_unsupport_mimo++;
      else if (mimo_ret == -2 && tb_en[i]) e->nof_pinfo++;
      else if (mimo_ret == -3 && tb_en[i]) e->nof_other_mimo++;
    }
}
/* MCSTracking::update_database_dl, MCSTracking.cc:850-927 (the all_database copies are statistics only) */
static void mcs_update_database(o_worker_t* w)
{
  const uint32_t now = w->sf_count;
  for (uint32_t r = 0; r < 65536; r++) {
    mcs_entry_t* e = &w->mcs[r];
    if (!e->present) continue;
    const uint32_t cur_interval = (now - e->time) / 1000u; /* whole seconds */
    const int wrong_detect = e->nof_active == 0 ||
                             (e->nof_active <= 10 && e->nof_success_mgs == 0 && (e->nof_unsupport_mimo > 0 || e->nof_pinfo > 0 || e->nof_other_mimo > 0));
    if (cur_interval > w->mcs_interval || wrong_detect || e->nof_active == 0) {
      memset(e, 0, sizeof(*e));
      w->mcs_count--;
    } else if ((float)e->nof_success_mgs / (float)e->nof_active < 0.15f && e->table != O_TABLE_UNKNOWN) {
      e->table = O_TABLE_UNKNOWN;
    }
  }
  w->nof_mcs_updates++;
}
void o_worker_set_second_opinion(o_worker_t* w, int turbo, int viterbi) { w->second_turbo = turbo; w->second_viterbi = viterbi; }
void o_worker_set_mcs_update_interval(o_worker_t* w, uint32_t seconds) { w->mcs_interval = seconds; w->mcs_update_period = seconds * 1000u; }
uint32_t o_worker_nof_tracked(o_worker_t* w) { return w->mcs_count; }
int o_worker_tracked_table(o_worker_t* w, uint16_t rnti) { return w->mcs[rnti].present ? (int)w->mcs[rnti].table : -1; }

/* ---------------- DCICollection::addCandidate ---------------- */
static void add_candidate(o_worker_t* w, const cand_t* c, uint32_t L, uint32_t ncce, uint32_t histval)
{
  const o_cell_t* cell = &w->cfg.cell;
  int fmt = c->msg.format;
  if (w->nacc < 64) {
    uint32_t* a = w->acc[w->nacc++];
    a[0] = c->rnti; a[1] = (uint32_t)fmt; a[2] = L; a[3] = ncce; a[4] = c->msg.nof_bits; a[5] = histval;
  }
  int table;
  if (w->cfg.mcs_tracking_mode == 1) { /* DCICollection.cc:107-134 */
    if (c->rnti == O_SIRNTI || c->rnti == O_PRNTI || O_RNTI_ISRAR(c->rnti) || fmt == O_FMT1A)
      table = O_TABLE_64QAM;
    else
      table = w->ul_mode ? O_TABLE_UNKNOWN : mcs_find(w, c->rnti); /* DCICollection.cc:117-121: the database is asked in DL_MODE only */
  } else if (w->cfg.mcs_tracking_mode == 2) {
    table = O_TABLE_UNKNOWN;
  } else {
    table = O_TABLE_64QAM;
  }
  if (fmt == O_FMT0) { /* falcon_dci.c:204-265 */
    if (w->nul >= 64) return;
    ul_entry_t* u = &w->ul[w->nul];
    memset(u, 0, sizeof(*u));
    u->rnti = c->rnti; u->nof_bits = c->msg.nof_bits; u->L = L; u->ncce = ncce; u->histval = histval;
    u->dci.L = L; u->dci.ncce = ncce;
    int ok = c->msg.payload[0] == 0 && o_dci_unpack_ul(cell, c->msg.payload, c->msg.nof_bits, c->rnti, &u->dci) == 0 &&
             o_ra_ul_dci_to_grant(cell, &u->dci, &u->g) == 0;
    if (ok && o_ra_ul_dci_to_grant_256(cell, &u->dci, &u->g256)) { ok = 0; memset(&u->g256, 0, sizeof(u->g256)); } /* falcon_dci.c:222-231 */
    if (!ok) u->dci.rnti = 0;
    u->ok = ok;
    if (ok) /* DCICollection.cc:275-280 */
      for (uint32_t i = 0; i < u->g.L_prb; i++) {
        if (w->rb_map_ul[u->g.n_prb + i] != 0) w->ul_collision = 1;
        w->rb_map_ul[u->g.n_prb + i] = c->rnti;
      }
    w->nul++;
    return;
  }
  if (w->ndl >= 64) return;
  dl_entry_t* e = &w->dl[w->ndl];
  memset(e, 0, sizeof(*e));
  e->rnti = c->rnti; e->format = fmt; e->nof_bits = c->msg.nof_bits; e->L = L; e->ncce = ncce; e->histval = histval;
  e->mcs_table = table;
  e->dci.L = L; e->dci.ncce = ncce;
  e->dci_rnti = c->rnti;
  if (o_dci_unpack_dl(cell, c->msg.payload, c->msg.nof_bits, fmt, c->rnti, &e->dci) == 0) { /* falcon_dci.c:271-310 */
    if (table == O_TABLE_64QAM || table >= O_TABLE_UNKNOWN) {
      e->has64 = 1;
      if (o_ra_dl_dci_to_grant(cell, w->sf_idx, w->cfi, 0, &e->dci, &e->g64)) e->dci_rnti = 0;
    }
    if (table == O_TABLE_256QAM || table >= O_TABLE_UNKNOWN) {
      e->has256 = 1;
      if (o_ra_dl_dci_to_grant(cell, w->sf_idx, w->cfi, 1, &e->dci, &e->g256)) e->dci_rnti = 0;
    }
  }
  const o_pdsch_grant_t* gm = e->has64 ? &e->g64 : &e->g256; /* convert_dl_grant source, falcon_dci.c:290,297,307 */
  for (uint32_t rb = 0; rb < cell->nof_prb; rb++) /* DCICollection.cc:215-223 */
    if (gm->prb_idx[0][rb]) {
      if (w->rb_map_dl[rb] != 0) w->dl_collision = 1;
      w->rb_map_dl[rb] = c->rnti;
    }
  /* DCICollection.cc:236-251: with HARQ on, a reserved MCS (29-31: "same size as the previous transmission") of a 64QAM-table grant takes the size the HARQ
   * database remembers for (RNTI, process, block).  The reference's 256QAM-table branch writes into the grant it did not compute for that entry (no effect). */
  if (w->harq_mode && table == O_TABLE_64QAM && e->has64)
    for (int i = 0; i < 2; i++)
      if (e->g64.tb[i].enabled && e->g64.tb[i].mcs_idx > 28) {
        int tbs = 0; /* HARQ::getlastTbs, HARQ.cc:262-274: the last entity of that RNTI */
        for (int k = 0; k < O_HARQ_ENTITIES; k++)
          if (w->harq[k].rnti == c->rnti) tbs = w->harq[k].tb[e->dci.pid & 7][i].grant.tbs;
        e->g64.tb[i].tbs = tbs;
        e->check = 1;
      }
  for (int i = 0; i < 2; i++) { /* DCICollection.cc:252-259 */
    if (e->g64.tb[i].nof_bits <= 0) e->g64.tb[i].enabled = 0;
    if (e->g256.tb[i].nof_bits <= 0) e->g256.tb[i].enabled = 0;
  }
  w->ndl++;
}

/* ---------------- candidate decode (falcon_pdcch.c:110-170) ---------------- */
static void decode_msg(o_worker_t* w, const floc_t* loc, int format, cand_t* c)
{
  uint32_t ncce_tot = w->regs.nof_cce[w->cfi - 1];
  uint32_t E = 72u << loc->L;
  if (loc->ncce * 72 + E > ncce_tot * 72) return;
  uint32_t nof_bits = o_dci_format_sizeof(&w->cfg.cell, format);
  const float* l = w->llr + loc->ncce * 72;
  double mean = 0;
  for (uint32_t i = 0; i < E; i++) mean += (l[i] < 0 ? -l[i] : l[i]);
  mean /= E;
  if (mean > 0.0) {
    c->rnti = w->second_viterbi ? o_dci_decode_second(l, (int)E, (int)nof_bits, c->msg.payload) : o_dci_decode(l, (int)E, (int)nof_bits, c->msg.payload);
    c->msg.nof_bits = nof_bits;
    if (format == O_FMT0 || format == O_FMT1A)
      c->msg.format = c->msg.payload[0] == 0 ? O_FMT0 : O_FMT1A;
    else
      c->msg.format = format;
  }
}

/* DCISearch::inspect_dci_location_recursively, DCISearch.cc:102-447 */
static int inspect(o_worker_t* w, cce_map_t* map, uint32_t ncce, uint32_t L, uint32_t max_depth, meta_t** metas,
                   uint32_t nformats, uint32_t discovery, const cand_t* parent)
{
  int hist_max_idx = -1;
  uint32_t hist_max_val = 0, n_ok = 0;
  cand_t cand[O_NOF_FORMATS];
  memset(cand, 0, sizeof(cand));
  floc_t* loc = map[ncce].location[L];
  if (!(loc && !loc->occupied && !loc->checked && loc->sufficient_power)) return 0;

  for (uint32_t fi = 0; fi < nformats; fi++) {
    decode_msg(w, loc, metas[fi]->format, &cand[fi]);
    w->stats.nof_decoded_locations++;
    if (o_rntiman_get_activation_reason(w->rm, cand[fi].rnti) == O_ACT_RAR && cand[fi].msg.format == O_FMT0) { /* :139-158 */
      int add = 1;
      for (uint32_t i = 0; i < w->ntemp0; i++)
        if (w->temp0[i].format == cand[fi].msg.format && w->temp0[i].rnti == cand[fi].rnti && w->temp0[i].ncce == ncce) add = 0;
      if (add && w->ntemp0 < 64) {
        temp_dci0_t* t = &w->temp0[w->ntemp0++];
        t->rnti = cand[fi].rnti; t->L = L; t->ncce = ncce; t->format = cand[fi].msg.format; t->cand = cand[fi];
      }
    }
    if (metas[fi]->format != cand[fi].msg.format) { /* :163 */
      cand[fi].rnti = 0;
      continue;
    }
    if (metas[fi]->format == O_FMT1C && cand[fi].rnti > O_RARNTI_END && cand[fi].rnti < O_PRNTI) { /* :174 */
      cand[fi].rnti = 0;
      continue;
    }
    if (cand[fi].rnti > O_RARNTI_START && cand[fi].rnti < O_RARNTI_END) /* :181-197 */
      if (metas[fi]->format != O_FMT1A && metas[fi]->format != O_FMT1C) {
        cand[fi].rnti = 0;
        continue;
      }
    if (w->cfg.enable_shortcut && discovery && parent != NULL && parent[fi].rnti == cand[fi].rnti &&
        !o_rntiman_is_forbidden(w->rm, cand[fi].rnti, metas[fi]->global_index)) /* :200-211 */
      return -((int)fi + 1);
    cand[fi].match = o_validate_location(w->regs.nof_cce[w->cfi - 1], ncce, L, w->sf_idx, cand[fi].rnti); /* :214 */
    if (cand[fi].match == 0) {
      cand[fi].rnti = 0;
      continue;
    }
    if (o_rntiman_validate_and_refresh(w->rm, cand[fi].rnti, metas[fi]->global_index)) { /* :245-250 */
      n_ok++;
      hist_max_idx = (int)fi;
      hist_max_val = o_rntiman_get_frequency(w->rm, cand[fi].rnti, metas[fi]->global_index);
    }
  }
  if (n_ok > 1) { /* :255-280 */
    hist_max_idx = -1;
    uint32_t hmax = 0;
    for (uint32_t fi = 0; fi < nformats; fi++)
      if (cand[fi].rnti != 0) {
        uint32_t h = o_rntiman_get_frequency(w->rm, cand[fi].rnti, metas[fi]->global_index);
        if (h > hmax) {
          hmax = h;
          hist_max_idx = (int)fi;
          hist_max_val = h;
        }
      }
    if (hist_max_idx == -1) n_ok = 0;
  }
  loc->checked = 1; /* :282 */
  int disamb = 0;
  if (n_ok > 0 && cand[hist_max_idx].match == 1) { /* :288-298 */
    if (L > 0 && max_depth > 0)
      disamb = inspect(w, map, ncce + (1u << (L - 1)), L - 1, max_depth - 1, metas, nformats, 0, NULL);
  } else if (n_ok == 0) { /* :302-368 */
    int rr = 0;
    if (L > 0 && max_depth > 0) {
      rr += inspect(w, map, ncce, L - 1, max_depth - 1, metas, nformats, discovery, cand);
      if (rr < 0) {
        hist_max_idx = -rr - 1;
        hist_max_val = o_rntiman_get_frequency(w->rm, cand[hist_max_idx].rnti, metas[hist_max_idx]->global_index);
        n_ok = 1;
        if (cand[hist_max_idx].match == 1) {
          uint32_t md = max_depth < 99 ? max_depth : 99;
          disamb = inspect(w, map, ncce + (1u << (L - 1)), L - 1, md - 1, metas, nformats, 0, NULL);
        }
        o_rntiman_activate_and_refresh(w->rm, cand[hist_max_idx].rnti, metas[hist_max_idx]->global_index, O_ACT_SHORTCUT);
      } else {
        rr += inspect(w, map, ncce + (1u << (L - 1)), L - 1, max_depth - 1, metas, nformats, discovery, NULL);
      }
    }
    if (rr == 0) {
      if (discovery)
        for (uint32_t fi = 0; fi < nformats; fi++)
          if (cand[fi].rnti != 0) o_rntiman_add_candidate(w->rm, cand[fi].rnti, metas[fi]->global_index);
      return 0;
    } else if (rr > 0) {
      return rr;
    }
  }
  if (n_ok > 0) { /* :371-439 */
    loc->used = 1;
    for (uint32_t ci = ncce; ci < ncce + (1u << L); ci++)
      for (int a = 0; a < 4; a++)
        if (map[ci].location[a]) {
          map[ci].location[a]->occupied = 1;
          map[ci].location[a]->checked = 1;
        }
    o_rntiman_add_candidate(w->rm, cand[hist_max_idx].rnti, metas[hist_max_idx]->global_index);
    metas[hist_max_idx]->hits++;
    uint32_t Ld = disamb > 0 ? L - 1 : L;
    cand[hist_max_idx].msg.rnti = cand[hist_max_idx].rnti;
    if (cand[hist_max_idx].rnti != 0) {
      int add = 1;
      if (cand[hist_max_idx].msg.format == O_FMT0)
        for (uint32_t i = 0; i < w->ntemp0; i++)
          if (w->temp0[i].format == O_FMT0 && w->temp0[i].rnti == cand[hist_max_idx].rnti && w->temp0[i].ncce == ncce) add = 0;
      if (add) add_candidate(w, &cand[hist_max_idx], Ld, ncce, hist_max_val);
      for (uint32_t i = 0; i < w->ntemp0; i++) { /* :422-432 - temp_dci0 entries all carry format 0 */
        uint32_t hv = o_rntiman_get_frequency(w->rm, w->temp0[i].rnti, (uint32_t)w->temp0[i].format);
        add_candidate(w, &w->temp0[i].cand, w->temp0[i].L, w->temp0[i].ncce, hv);
      }
      w->ntemp0 = 0;
    }
    return 1 + disamb;
  }
  return 0;
}

/* DCISearch::recursive_blind_dci_search, DCISearch.cc:449-528 */
static void blind_search(o_worker_t* w)
{
  floc_t locs[O_MAX_LOCATIONS];
  cce_map_t map[O_MAX_NUM_OF_CCE];
  memset(map, 0, sizeof(map));
  uint32_t ncce = w->regs.nof_cce[w->cfi - 1];
  uint32_t lim = ncce < O_MAX_NUM_OF_CCE ? ncce : O_MAX_NUM_OF_CCE;
  w->stats.nof_cce += ncce;
  uint32_t k = 0;
  for (int l = 3; l >= 0; l--) { /* falcon_pdcch.c:321-356 */
    uint32_t L = 1u << l;
    for (uint32_t i = 0; i < lim / L; i++)
      if (k < O_MAX_LOCATIONS) {
        memset(&locs[k], 0, sizeof(floc_t));
        locs[k].L = (uint32_t)l;
        locs[k].ncce = L * (i % (ncce / L));
        locs[k].sufficient_power = 1;
        for (uint32_t m = locs[k].ncce; m < locs[k].ncce + L; m++) map[m].location[l] = &locs[k];
        k++;
      }
  }
  uint32_t nloc = k;
  w->stats.nof_locations += nloc;
  for (uint32_t c = 0; c < ncce && c < O_MAX_NUM_OF_CCE; c++) { /* falcon_pdcch.c:595-620 */
    double mean = 0;
    for (int i = 0; i < 72; i++) {
      float v = w->llr[c * 72 + (uint32_t)i];
      mean += (v < 0 ? -v : v);
    }
    map[c].power = (float)(mean / 72);
    if (map[c].power < 0.7f)
      for (int a = 0; a < 4; a++)
        if (map[c].location[a]) map[c].location[a]->sufficient_power = 0;
  }
  for (uint32_t i = 0; i < nloc; i++)
    inspect(w, map, locs[i].ncce, locs[i].L, 99, w->primary, w->nprimary, 1, NULL);
  if (!w->cfg.skip_secondary) {
    for (uint32_t i = 0; i < nloc; i++) locs[i].checked = 0;
    for (uint32_t i = 0; i < nloc; i++)
      inspect(w, map, locs[i].ncce, locs[i].L, 99, w->secondary, w->nsecondary, 1, NULL);
  }
  if (w->dl_collision) w->stats.nof_subframe_collisions_dw++;
  if (w->ul_collision) w->stats.nof_subframe_collisions_up++;
  uint32_t missed = 0; /* falcon_pdcch.c:561-593 */
  for (uint32_t c = 0; c < ncce && c < O_MAX_NUM_OF_CCE; c++) {
    if (map[c].power < 0.7f) continue;
    int m = 1;
    for (int a = 0; a < 4; a++)
      if (map[c].location[a] && map[c].location[a]->used) {
        m = 0;
        break;
      }
    if (m) missed++;
  }
  w->stats.nof_missed_cce += missed;
  o_rntiman_step_time(w->rm);
}

/* ---------------- PDSCH ---------------- */
static void write_pcap(o_worker_t* w, const char* name, const uint8_t* pdu, uint32_t len, uint16_t rnti, uint32_t tti)
{
  if (!w->pcap) return;
  w->records++;
  if (name[0] == 'S') o_pcap_write(w->pcap, pdu, len, tti, O_SIRNTI, 1, O_PCAP_SI_RNTI, 1, 0, 0);
  else if (name[0] == 'P') o_pcap_write(w->pcap, pdu, len, tti, O_PRNTI, 1, O_PCAP_P_RNTI, 1, 0, 0);
  else if (name[0] == 'R') o_pcap_write(w->pcap, pdu, len, tti, rnti, 1, O_PCAP_RA_RNTI, 1, 0, 0);
  else o_pcap_write(w->pcap, pdu, len, tti, rnti, 1, O_PCAP_C_RNTI, 1, 0, 0);
  if (w->api_mode >= 0 && (name[0] == 'P' || name[0] == 'C')) { /* run_api_dl_mode, DL_Sniffer_PDSCH.cc:804-879 */
    o_api_event_t ev[20];
    int nev = 0;
    if (o_api_dl_events(w->api_mode, name[0], pdu, (int)len, rnti, tti, ev, 20, &nev) && w->api_pcap)
      o_pcap_write(w->api_pcap, pdu, len, tti, name[0] == 'P' ? O_PRNTI : rnti, 1, name[0] == 'P' ? O_PCAP_P_RNTI : O_PCAP_C_RNTI, 1, 0, 0);
    for (int i = 0; i < nev; i++) {
      if (w->api_n == w->api_cap) { w->api_cap = w->api_cap ? 2 * w->api_cap : 64; w->api_ev = (o_api_event_t*)realloc(w->api_ev, sizeof(o_api_event_t) * (size_t)w->api_cap); }
      w->api_ev[w->api_n++] = ev[i];
    }
  }
}
void o_worker_set_api(o_worker_t* w, int api_mode, o_pcap_t* api_pcap) { w->api_mode = api_mode; w->api_pcap = api_pcap; }
int o_worker_api_events(o_worker_t* w, o_api_event_t* out, int cap)
{
  int n = w->api_n < cap ? w->api_n : cap;
  if (out) memcpy(out, w->api_ev, sizeof(o_api_event_t) * (size_t)n);
  return w->api_n;
}

static const char* rnti_name(uint16_t r) /* DL_Sniffer_PDSCH.cc:1398-1418 */
{
  if (r == O_SIRNTI) return "SI_RNTI";
  if (r == O_PRNTI) return "P_RNTI";
  if (r > O_RARNTI_START && r < O_RARNTI_END) return "RA_RNTI";
  return "C_RNTI";
}

/* MAC RAR PDU (TS 36.321 6.1.5, 6.2.2, 6.2.3): E/T/RAPID sub-headers (a T = 0 sub-header is the backoff indicator and has no body),
 * then one 6-byte RAR per RAPID sub-header: R(1) TA(11) UL grant(20) T-CRNTI(16); the grant as ul_sniffer_dci_rar_unpack /
 * _rar_to_ul_dci read it (falcon_dci.c:648-683: hopping(1) RIV(10) MCS(4) TPC(3) UL delay(1) CSI request(1)) and convert it
 * with srsran_ra_ul_dci_to_grant.  A sub-header without a body yields an entry with T-CRNTI 0, like the reference's loop. */
int o_rar_parse(const o_cell_t* cell, const uint8_t* p, int len, o_rar_t* out, int cap)
{
  int nsub = 0, is_rapid[32], rapid[32], pos = 0, n = 0;
  while (pos < len && nsub < 32) {
    uint8_t b = p[pos++];
    is_rapid[nsub] = (b & 0x40) ? 1 : 0;
    rapid[nsub] = b & 0x3F;
    nsub++;
    if (!(b & 0x80)) break;
  }
  for (int i = 0; i < nsub && n < cap; i++) {
    o_rar_t* r = &out[n];
    memset(r, 0, sizeof(*r));
    uint32_t grant20 = 0;
    if (is_rapid[i]) {
      if (pos + 6 > len) break;
      r->rapid = (uint32_t)rapid[i];
      r->ta = ((uint32_t)(p[pos] & 0x7F) << 4) | (p[pos + 1] >> 4);
      grant20 = ((uint32_t)(p[pos + 1] & 0x0F) << 16) | ((uint32_t)p[pos + 2] << 8) | p[pos + 3];
      r->t_crnti = (uint16_t)((p[pos + 4] << 8) | p[pos + 5]);
      pos += 6;
    }
    o_dci_ul_t d;
    memset(&d, 0, sizeof(d));
    d.rnti = r->t_crnti;
    d.freq_hop_fl = (grant20 >> 19) & 1u;
    d.hop_type = d.freq_hop_fl ? 1 : -1; /* ul_sniffer_dci_rar_to_ul_dci, falcon_dci.c:665-670: "freq_hop_fl = 1" = the -N/4 type-1 pattern on the full RIV */
    d.riv = (grant20 >> 9) & 0x3FFu;
    d.mcs_idx = (grant20 >> 5) & 0xFu;
    r->hopping = d.freq_hop_fl; r->riv = d.riv; r->mcs = d.mcs_idx; r->tpc = (grant20 >> 2) & 7u; r->ul_delay = (grant20 >> 1) & 1u; r->csi_req = grant20 & 1u;
    r->grant_ok = o_ra_ul_dci_to_grant(cell, &d, &r->grant) == 0;
    if (!r->grant_ok) memset(&r->grant, 0, sizeof(r->grant));
    n++;
  }
  return n;
}

/* MAC RAR PDU (TS 36.321 6.1.5 / 6.2.2-6.2.3) walked like srsran::rar_pdu; DL_Sniffer_PDSCH.cc:782-797 */
static void unpack_rar(o_worker_t* w, const uint8_t* p, int len)
{
  int nsub = 0, is_rapid[32], pos = 0;
  while (pos < len && nsub < 32) {
    uint8_t b = p[pos++];
    is_rapid[nsub++] = (b & 0x40) ? 1 : 0;
    if (!(b & 0x80)) break;
  }
  for (int i = 0; i < nsub; i++) {
    uint16_t t_crnti = 0;
    if (is_rapid[i]) {
      if (pos + 6 > len) break;
      t_crnti = (uint16_t)((p[pos + 4] << 8) | p[pos + 5]);
      pos += 6;
    }
    mcs_rar(w, t_crnti);
    o_rntiman_activate_and_refresh(w->rm, t_crnti, 0, O_ACT_RAR);
  }
}

void o_worker_set_harq(o_worker_t* w, int mode)
{
  w->harq_mode = mode;
  if (mode && !w->harq) {
    w->harq = (struct o_harq_entity*)calloc(O_HARQ_ENTITIES, sizeof(struct o_harq_entity));
    for (int i = 0; i < O_HARQ_ENTITIES; i++)
      for (int p = 0; p < 8; p++)
        for (int t = 0; t < 2; t++) w->harq[i].tb[p][t].grant.is_first = 1; /* dl_sniffer_harq_grant_t default */
    w->harq_nof_aval = 150; /* nof_aval = DL_SNIFFER_MAX_HARQ_SIZE */
  }
}
void o_worker_harq_stats(o_worker_t* w, uint32_t* out5) { memcpy(out5, w->harq_stats, sizeof(w->harq_stats)); }

/* HARQ::is_retransmission, HARQ.cc:71-135 */
static int harq_is_retx(o_worker_t* w, uint16_t rnti, int pid, int tid, int ndi, int tbs, uint32_t sfn, uint32_t sf_idx, struct o_harq_entity** ent)
{
  struct o_harq_entity *found = NULL, *avail = NULL;
  for (int i = 0; i < O_HARQ_ENTITIES; i++) {
    if (w->harq[i].rnti == rnti) found = &w->harq[i];
    else if (w->harq[i].rnti == 0) avail = &w->harq[i]; /* the LAST free entity */
  }
  *ent = found;
  if (!found && avail) {
    avail->rnti = rnti;
    if (w->harq_nof_aval > 0) w->harq_nof_aval--;
    *ent = avail;
    return O_HARQ_NEW_TX;
  }
  if (!found) return O_HARQ_FULL_BUFFER;
  o_harq_tb_t* t = &found->tb[pid][tid];
  uint32_t last_tti = t->sfn * 10 + t->sf_idx, cur_tti = sfn * 10 + sf_idx;
  if (!(cur_tti - last_tti == 8 || cur_tti + 10240 - last_tti == 8)) return O_HARQ_NEW_TX; /* comparetti */
  if (ndi != t->grant.ndi || t->grant.is_first || t->grant.tbs != tbs) return O_HARQ_NEW_TX;
  return t->grant.last_decoded ? O_HARQ_DECODED : O_HARQ_RE_TX;
}
/* HARQ::updateHARQRNTI / updateProcess, HARQ.cc:155-190 */
static void harq_update(o_worker_t* w, struct o_harq_entity* e, int pid, int tid, uint32_t sfn, uint32_t sf_idx, int last_decoded, int ndi, int rv, int tbs)
{
  e->time = w->sf_count;
  o_harq_tb_t* t = &e->tb[pid][tid];
  t->sfn = sfn; t->sf_idx = sf_idx;
  t->grant.last_decoded = last_decoded; t->grant.ndi = ndi; t->grant.rv = rv; t->grant.tbs = tbs; t->grant.is_first = 0;
}
/* HARQ::updateHARQDatabase, HARQ.cc:206-238, driven by the 10 s timer of LTESniffer_Core.cc:487-494: once fewer than 11 entities are free, the ones
 * idle for more than `interval` = 5 whole seconds are released */
static void harq_update_database(o_worker_t* w)
{
  if (w->harq_nof_aval > 10) return;
  for (int i = 0; i < O_HARQ_ENTITIES; i++) {
    struct o_harq_entity* e = &w->harq[i];
    if ((w->sf_count - e->time) / 1000u > 5u) {
      e->rnti = 0; e->time = 0;
      for (int p = 0; p < 8; p++)
        for (int t = 0; t < 2; t++) { e->tb[p][t].grant.is_first = 1; e->tb[p][t].grant.last_decoded = 0; e->tb[p][t].grant.tbs = 0; e->tb[p][t].sf_idx = 0; }
      w->harq_nof_aval++;
    }
  }
}

/* one srsran_ue_dl_decode_pdsch call: returns crc[2]; payload of TB i at w->payload + i*8192 */
static void decode_grant_harq(o_worker_t* w, const dl_entry_t* e, const o_pdsch_grant_t* g, int* crc, int16_t* const* acc, const int* combine, uint8_t* const* keep);
static void decode_grant(o_worker_t* w, const dl_entry_t* e, const o_pdsch_grant_t* g, int* crc) { decode_grant_harq(w, e, g, crc, NULL, NULL, NULL); }
static void decode_grant_harq(o_worker_t* w, const dl_entry_t* e, const o_pdsch_grant_t* g, int* crc, int16_t* const* acc, const int* combine, uint8_t* const* keep)
{
  crc[0] = crc[1] = 0;
  if (!(g->tb[0].enabled || g->tb[1].enabled)) return;
  if (w->script_fn) { /* the decode call as srsran_ue_dl_decode_pdsch would receive it: {tti, rnti, nof_re, tx_scheme, pmi, nof_layers, per block: enabled, modulation bits, tbs, rv, cw_idx} */
    uint32_t call[16] = {w->sfn * 10 + w->sf_idx, e->rnti, g->nof_re, (uint32_t)g->tx_scheme, g->pmi, g->nof_layers};
    for (int i = 0; i < 2; i++) {
      uint32_t* t = call + 6 + 5 * i;
      t[0] = (uint32_t)g->tb[i].enabled; t[1] = g->tb[i].enabled ? (uint32_t)g->tb[i].mod : 0; t[2] = g->tb[i].enabled ? (uint32_t)g->tb[i].tbs : 0;
      t[3] = g->tb[i].enabled ? (uint32_t)g->tb[i].rv : 0; t[4] = g->tb[i].enabled ? g->tb[i].cw_idx : 0;
    }
    int32_t c2[2] = {0, 0};
    w->script_fn(w->script_user, call, w->ul_mode ? -3.0f : ue_cfg_get(w, e->rnti).p_a, w->payload, w->payload + 16384, c2);
    for (int i = 0; i < 2; i++)
      if (g->tb[i].enabled && g->tb[i].tbs > 0) crc[i] = c2[i] != 0;
    (void)acc; (void)combine; (void)keep;
    return;
  }
  memset(w->llr0, 0, sizeof(int16_t) * g->nof_re * 8);
  memset(w->llr1, 0, sizeof(int16_t) * g->nof_re * 8);
  /* pdsch_cfg->p_a: the UE's p-a from its RRCConnectionSetup (or the default) in DL mode, DL_Sniffer_PDSCH.cc:926-927; the UL-mode
   * decoders never set it and run with the initial -3 dB of SubframeWorker::set_pdsch_uecfg (SubframeWorker.cc:370) */
  const float p_a = w->ul_mode ? -3.0f : ue_cfg_get(w, e->rnti).p_a;
  if (w->second_turbo) o_pdsch_set_llr_clip(32767);
  const int demod_rc = o_pdsch_demod(&w->cfg.cell, w->cfg.nof_rx, w->sf_idx, w->cfi, e->rnti, g, w->grid, w->ce, w->chest.noise_avg,
                                     w->chest.chan_ref, p_a, w->llr0, w->llr1);
  if (w->second_turbo) o_pdsch_set_llr_clip(511);
  if (demod_rc) return;
  if (o_trace_enabled()) {
    int qm_cw[2] = {0, 0};
    for (int i = 0; i < 2; i++)
      if (g->tb[i].enabled) qm_cw[g->tb[i].cw_idx & 1] = g->tb[i].mod;
    o_trace_begin_job(w->sfn * 10 + w->sf_idx, e->rnti, g->nof_re, qm_cw, w->llr0, w->llr1, 0);
  }
  for (int i = 0; i < 2; i++)
    if (g->tb[i].enabled && g->tb[i].tbs > 0) {
      const int16_t* llr = (g->tb[i].cw_idx & 1) ? w->llr1 : w->llr0;
      int its = 0;
      o_trace_set_tb(i);
      if (acc && acc[i])
        crc[i] = o_pdsch_decode_tb_harq(llr, g->tb[i].nof_bits, g->tb[i].tbs, g->tb[i].mod, g->tx_scheme == O_TX_DIVERSITY ? 2 : 1, g->tb[i].rv, w->cfg.max_turbo_iter,
                                        w->payload + i * 8192 * 2, &its, acc[i], combine[i], keep ? keep[i] : NULL);
      else
      crc[i] = (w->second_turbo ? o_pdsch_decode_tb_second : o_pdsch_decode_tb)(llr, g->tb[i].nof_bits, g->tb[i].tbs, g->tb[i].mod, g->tx_scheme == O_TX_DIVERSITY ? 2 : 1,
                                 g->tb[i].rv, w->cfg.max_turbo_iter, w->payload + i * 8192 * 2, &its);
      w->total_iters += (uint64_t)its;
      w->algo_bytes += 2ull * (uint64_t)g->tb[i].nof_bits * 2ull + (uint64_t)g->tb[i].tbs / 8ull;
    }
}

/* PDSCH_Decoder::decode_dl_mode, DL_Sniffer_PDSCH.cc:881-1291 */
static void decode_dl_mode(o_worker_t* w)
{
  uint32_t tti = w->sfn * 10 + w->sf_idx;
  for (uint32_t di = 0; di < w->ndl; di++) {
    dl_entry_t* e = &w->dl[di];
    o_pdsch_grant_t* cur = (e->mcs_table == O_TABLE_256QAM) ? &e->g256 : &e->g64;
    o_pdsch_grant_t* cur256 = &e->g256;
    int two_tb = (e->has64 && e->g64.nof_tb == 2) || (e->has256 && e->g256.nof_tb == 2);
    int gate = (cur->tb[0].tbs > 0 && e->dci_rnti > 0 && !(w->cfg.nof_rx == 1 && two_tb)) || e->rnti == O_PRNTI; /* :887-889 */
    if (!gate) continue;
    if (e->dci.tb[0].rv < 0 && e->rnti == O_SIRNTI) cur->tb[0].rv = 0; /* :891-897 */
    const char* name = rnti_name(e->rnti);
    int crc[2] = {0, 0};
    int mimo_ret;
    if (e->mcs_table == O_TABLE_64QAM || e->mcs_table == O_TABLE_256QAM) { /* :932-1083 */
      mimo_ret = o_config_mimo(&w->cfg.cell, e->format, &e->dci, cur);
      if (mimo_ret == 0) {
        if (w->harq_mode && name[0] == 'C') { /* :943-1020: new transmission / retransmission / already decoded, per transport block */
          o_pdsch_grant_t gg = *cur; /* pdsch_cfg->grant */
          int harq_ret[2] = {O_HARQ_NEW_TX, O_HARQ_NEW_TX}, combine[2] = {0, 0};
          int16_t* acc[2] = {NULL, NULL};
          uint8_t* keep[2] = {NULL, NULL};
          struct o_harq_entity* ent[2] = {NULL, NULL};
          for (int i = 0; i < 2; i++)
            if (gg.tb[i].enabled) {
              harq_ret[i] = harq_is_retx(w, e->rnti, (int)e->dci.pid, i, (int)e->dci.tb[i].ndi, gg.tb[i].tbs, w->sfn, w->sf_idx, &ent[i]);
              w->harq_stats[harq_ret[i]]++;
              if (harq_ret[i] == O_HARQ_NEW_TX || harq_ret[i] == O_HARQ_RE_TX) {
                o_harq_tb_t* t = &ent[i]->tb[e->dci.pid][i];
                if (!t->acc) t->acc = (int16_t*)calloc((size_t)O_HARQ_MAX_CB * O_HARQ_CB_STRIDE, sizeof(int16_t));
                if (!t->keep) t->keep = (uint8_t*)calloc((size_t)O_HARQ_MAX_CB * O_HARQ_KEEP_STRIDE, 1);
                acc[i] = t->acc; keep[i] = t->keep; combine[i] = harq_ret[i] == O_HARQ_RE_TX;
              } else if (harq_ret[i] == O_HARQ_DECODED) {
                gg.tb[i].enabled = 0; /* decoded 8 subframes ago: the block is not decoded again (and nothing is written for it) */
              }
            }
          if (gg.tb[0].enabled || gg.tb[1].enabled) decode_grant_harq(w, e, &gg, crc, acc, combine, keep);
          for (int i = 0; i < 2; i++)
            if (gg.tb[i].enabled && (harq_ret[i] == O_HARQ_NEW_TX || harq_ret[i] == O_HARQ_RE_TX))
              harq_update(w, ent[i], (int)e->dci.pid, i, w->sfn, w->sf_idx, crc[i], (int)e->dci.tb[i].ndi, e->dci.tb[i].rv, cur->tb[i].tbs);
        } else {
          decode_grant(w, e, cur, crc);
        }
        for (int tb = 0; tb < 2; tb++) {
          int len = cur->tb[tb].tbs / 8;
          if (crc[tb] && len > 0) {
            write_pcap(w, name, w->payload + tb * 16384, (uint32_t)len, e->rnti, tti);
            if (name[0] == 'R') unpack_rar(w, w->payload + tb * 16384, len);
            if (name[0] == 'C') learn_conn_setup(w, w->payload + tb * 16384, len, e->rnti, 0);
          }
        }
      }
    } else { /* unknown table: 64QAM first, then 256QAM if both TBs failed, :1089-1243 */
      mimo_ret = o_config_mimo(&w->cfg.cell, e->format, &e->dci, cur);
      if (mimo_ret == 0) {
        decode_grant(w, e, cur, crc);
        for (int tb = 0; tb < 2; tb++) {
          int len = cur->tb[tb].tbs / 8;
          if (crc[tb] && len > 0) {
            write_pcap(w, name, w->payload + tb * 16384, (uint32_t)len, e->rnti, tti);
            if (name[0] == 'R') unpack_rar(w, w->payload + tb * 16384, len);
            if (name[0] == 'C') learn_conn_setup(w, w->payload + tb * 16384, len, e->rnti, 1);
            if (e->dci.tb[tb].mcs_idx > 0 && e->dci.tb[tb].mcs_idx < 29 && e->format > O_FMT1A) mcs_update(w, e->rnti, O_TABLE_64QAM);
          }
        }
      }
      if (!crc[0] && !crc[1] && mimo_ret == 0) {
        mimo_ret = o_config_mimo(&w->cfg.cell, e->format, &e->dci, cur256);
        if (mimo_ret == 0) {
          int crc2[2];
          decode_grant(w, e, cur256, crc2);
          for (int tb = 0; tb < 2; tb++) {
            if (cur256->tb[tb].enabled) crc[tb] = crc2[tb];
            int len = cur256->tb[tb].tbs / 8;
            if (crc[tb] && len > 0) {
              write_pcap(w, name, w->payload + tb * 16384, (uint32_t)len, e->rnti, tti);
              if (e->dci.tb[tb].mcs_idx > 0 && e->dci.tb[tb].mcs_idx < 28 && e->format > O_FMT1A) mcs_update(w, e->rnti, O_TABLE_256QAM);
            }
          }
        }
      }
    }
    if (name[0] == 'C' && w->cfg.mcs_tracking_mode) { /* :1268-1285 */
      const int tb_en[2] = {cur->tb[0].enabled, cur->tb[1].enabled};
      mcs_statistic(w, e->rnti, e->format, e->mcs_table, tb_en, crc, mimo_ret ? -mimo_ret : 0);
    }
  }
}

void o_worker_ue_cfg(o_worker_t* w, uint16_t rnti, o_ue_cfg_t* out) { *out = ue_cfg_get(w, rnti); }

int o_worker_work(o_worker_t* w, const ocf_t* const* iq, uint32_t sf_idx, uint32_t sfn, int update_meta, float cfo_hz)
{
  const o_cell_t* cell = &w->cfg.cell;
  size_t nre = 12u * cell->nof_prb;
  w->sf_idx = sf_idx;
  w->sfn = sfn;
  w->records = 0;
  w->ndl = w->nul = w->nacc = w->ntemp0 = 0;
  w->dl_collision = w->ul_collision = 0;
  memset(w->rb_map_dl, 0, sizeof(w->rb_map_dl));
  memset(w->rb_map_ul, 0, sizeof(w->rb_map_ul));
  if (update_meta) update_formats(w); /* SubframeWorker.cc:148-151 */
  /* LTESniffer_Core.cc:473-499: every get_interval() x 1000 subframes the tracking database is aged (DL mode, mcs_tracking_mode on) */
  if (w->cfg.mcs_tracking_mode && w->mcs_update_period && w->sf_count && (w->sf_count % w->mcs_update_period) == 0) mcs_update_database(w);
  if (w->harq_mode && w->sf_count && (w->sf_count % 10000u) == 0) harq_update_database(w); /* the 10 s timer, LTESniffer_Core.cc:487-494 */
  uint32_t dphi = cfo_hz != 0.0f ? o_nco_dphi(cfo_hz, o_fft_size(cell->nof_prb)) : 0;
  /* srsran_ue_dl_decode_fft_estimate, DCISearch.cc:562 */
  for (uint32_t rx = 0; rx < w->cfg.nof_rx; rx++) o_ofdm_rx(cell, iq[rx], dphi, w->grid + rx * 14u * nre);
  o_chest(cell, w->cfg.nof_rx, sf_idx, w->grid, w->ce, &w->chest);
  w->cfi = o_pcfich_decode(cell, &w->regs, w->cfg.nof_rx, sf_idx, w->grid, w->ce, w->chest.noise_avg, NULL);
  o_pdcch_llr(cell, &w->regs, w->cfg.nof_rx, sf_idx, w->cfi, w->grid, w->ce, w->chest.noise_avg, w->llr);
  o_subframe_power(cell, w->grid, w->rb_power, NULL, NULL); /* DCISearch.cc:565 */
  uint32_t A = w->cfg.nof_rx, P = cell->nof_ports;
  w->algo_bytes += A * (uint64_t)(15 * o_fft_size(cell->nof_prb)) * 8ull + 2ull * A * 14ull * nre * 8ull + 2ull * P * A * 14ull * nre * 8ull +
                   2ull * w->regs.nof_cce[w->cfi - 1] * 72ull * 4ull;
  if (w->chest.snr_db > 6.0f) { /* DCISearch.cc:568-574 */
    blind_search(w);
    w->stats.nof_subframes++;
    decode_dl_mode(w); /* SubframeWorker.cc:224 */
  } else {
    w->stats.nof_subframes++;
  }
  w->sf_count++;
  return w->records;
}


/* ================================================================================================ UL mode */
/* ul == NULL: no configuration yet - the worker runs PDSCH_Decoder::decode_SIB until a SIB2 configures it (SubframeWorker.cc:238-252) */
void o_worker_set_ul_mode(o_worker_t* w, const o_ul_cfg_t* ul)
{
  w->ul_mode = 1;
  w->ul_configured = ul != NULL;
  w->sib2_learned = 0;
  if (ul) {
    w->ulcfg = *ul;
    w->cfg.cell.pusch_hop_offset = ul->hopping_offset; /* ul_cfg.hopping.n_rb_ho, SubframeWorker.cc:271-273 */
  }
  if (!w->ul_grid) {
    w->ul_grid = (ocf_t*)calloc(14u * 12u * w->cfg.cell.nof_prb, sizeof(ocf_t));
    w->ul_sched = (ulslot_t*)calloc(16, sizeof(ulslot_t));
    w->rar_sched = (ulslot_t*)calloc(16, sizeof(ulslot_t));
    w->ulmod = (uint8_t*)calloc(65536, 1);
    w->ul_time = (uint32_t*)calloc(65536, 4); w->ul_active = (uint32_t*)calloc(65536, 4); w->ul_success = (uint32_t*)calloc(65536, 4);
  }
}

static void write_pcap_ul(o_worker_t* w, const uint8_t* pdu, uint32_t len, uint16_t rnti, uint32_t tti, int is_rar)
{
  if (!w->pcap) return;
  w->records++;
  o_pcap_write(w->pcap, pdu, len, tti, rnti, 0, O_PCAP_C_RNTI, 1, 0, 0); /* write_ul_crnti, PcapWriter.cc:172-175 */
  if (w->api_mode >= 0) { /* decode_run's API part, UL_Sniffer_PUSCH.cc:306-372: Msg3 of a RAR grant (modes 0, 3), else SRB messages (modes 1-3) */
    o_api_event_t ev[10];
    int nev = 0;
    const int msg3 = is_rar && (w->api_mode == 0 || w->api_mode == 3);
    if ((msg3 ? o_api_ul_msg3_events(w->api_mode, pdu, (int)len, rnti, tti, ev, 10, &nev) : o_api_ul_dcch_events(w->api_mode, pdu, (int)len, rnti, tti, ev, 10, &nev)) && w->api_pcap)
      o_pcap_write(w->api_pcap, pdu, len, tti, rnti, 0, O_PCAP_C_RNTI, 1, 0, 0); /* write_ul_crnti_api */
    for (int i = 0; i < nev; i++) {
      if (w->api_n == w->api_cap) { w->api_cap = w->api_cap ? 2 * w->api_cap : 64; w->api_ev = (o_api_event_t*)realloc(w->api_ev, sizeof(o_api_event_t) * (size_t)w->api_cap); }
      w->api_ev[w->api_n++] = ev[i];
    }
  }
}

/* unpack_rar_response_ul_mode, DL_Sniffer_PDSCH.cc:632-671: every sub-header activates its temporary C-RNTI; the UL grant
 * that survives is the LAST one of the PDU (the result object is overwritten per sub-header) */
static int unpack_rar_ul(o_worker_t* w, const uint8_t* p, int len, ulg_t* out)
{
  o_rar_t r[32];
  int n = o_rar_parse(&w->cfg.cell, p, len, r, 32);
  for (int i = 0; i < n; i++) {
    memset(out, 0, sizeof(*out));
    out->rnti = r[i].t_crnti;
    out->is_rar = 1;
    out->hopping = 0; /* a hopping RAR grant is a type-1 grant (see o_rar_parse) */
    if (r[i].grant_ok) out->g = r[i].grant; /* ran_ul_grant_256 stays empty for RAR grants */
    o_rntiman_activate_and_refresh(w->rm, r[i].t_crnti, 0, O_ACT_RAR);
  }
  return n > 0;
}

/* PDSCH_Decoder::decode_ul_mode (DL_Sniffer_PDSCH.cc:362-457) with rnti == 0 (no target); RRC parsing is out of scope */
static void decode_ul_mode_dl(o_worker_t* w, ulslot_t* rar_out)
{
  uint32_t tti = w->sfn * 10 + w->sf_idx;
  for (uint32_t di = 0; di < w->ndl; di++) {
    dl_entry_t* e = &w->dl[di];
    int crc[2] = {0, 0};
    if (e->rnti >= O_RARNTI_START && e->rnti <= O_RARNTI_END) { /* run_rar_decode, :673-740 */
      o_pdsch_grant_t* cur = &e->g64;
      if (o_config_mimo(&w->cfg.cell, e->format, &e->dci, cur) != 0) continue;
      for (int i = 0; i < 2; i++)
        if (cur->tb[i].enabled && cur->tb[i].rv < 0) cur->tb[i].rv = (int)((uint32_t)ceilf(1.5f * (float)((w->sfn / 2) % 4)) % 4u);
      decode_grant(w, e, cur, crc);
      for (int tb = 0; tb < 2; tb++)
        if (crc[tb]) {
          int len = cur->tb[tb].tbs / 8;
          write_pcap(w, "RA_RNTI", w->payload + tb * 16384, (uint32_t)len, e->rnti, tti);
          ulg_t g;
          if (unpack_rar_ul(w, w->payload, len, &g) && rar_out->n < 96) rar_out->g[rar_out->n++] = g; /* pdsch_res->payload = TB 0 buffer */
          break;
        }
    } else if (e->rnti > O_RARNTI_END) {
      if ((e->format == O_FMT1 || e->format == O_FMT1A) && e->rnti != O_SIRNTI) { /* run_decode with the 64QAM table, :223-360 */
        o_pdsch_grant_t* cur = &e->g64;
        if (o_config_mimo(&w->cfg.cell, e->format, &e->dci, cur) != 0) continue;
        decode_grant(w, e, cur, crc);
        for (int tb = 0; tb < 2; tb++)
          if (crc[tb]) {
            write_pcap(w, rnti_name(e->rnti), w->payload + tb * 16384, (uint32_t)(cur->tb[tb].tbs / 8), e->rnti, tti);
            learn_conn_setup(w, w->payload + tb * 16384, cur->tb[tb].tbs / 8, e->rnti, 0); /* run_decode, :279-306: betaOffset indices + CQI mode for the PUSCH decoder */
          }
      }
    }
  }
}

/* PDSCH_Decoder::decode_SIB, DL_Sniffer_PDSCH.cc:459-560: SI-RNTI grants with the 64QAM table; the first transport block that carries a
 * SystemInformation with SIB2 is written to the pcap and ends the subframe */
static int decode_sib(o_worker_t* w)
{
  uint32_t tti = w->sfn * 10 + w->sf_idx;
  for (uint32_t di = 0; di < w->ndl; di++) {
    dl_entry_t* e = &w->dl[di];
    int crc[2] = {0, 0};
    if (e->rnti != O_SIRNTI) continue;
    o_pdsch_grant_t* cur = &e->g64;
    if (o_config_mimo(&w->cfg.cell, e->format, &e->dci, cur) != 0) continue;
    for (int i = 0; i < 2; i++)
      if (cur->tb[i].enabled && cur->tb[i].rv < 0) cur->tb[i].rv = (int)((uint32_t)ceilf(1.5f * (float)((w->sfn / 2) % 4)) % 4u);
    decode_grant(w, e, cur, crc);
    for (int tb = 0; tb < 2; tb++)
      if (crc[tb]) {
        int len = cur->tb[tb].tbs / 8;
        o_sib2_t s;
        if (o_sib2_decode(w->payload + tb * 16384, len, &s) == 2) {
          w->sib2 = s;
          write_pcap(w, "SI_RNTI", w->payload + tb * 16384, (uint32_t)len, e->rnti, tti);
          return 1;
        }
      }
  }
  return 0;
}
int o_worker_ul_config(o_worker_t* w, o_ul_cfg_t* ul, o_sib2_t* sib2) /* ULSchedule::get_config / getSIB2: 0 none, 1 given, 2 learned from SIB2 */
{
  if (!w->ul_configured) return 0;
  if (ul) *ul = w->ulcfg;
  if (sib2 && w->sib2_learned) *sib2 = w->sib2;
  return w->sib2_learned ? 2 : 1;
}

/* one srsran_chest_ul_estimate_pusch + srsran_pusch_decode attempt (PUSCH_Decoder::decode_run, UL_Sniffer_PUSCH.cc:250-310) */
static int pusch_attempt(o_worker_t* w, const ulg_t* m, const o_pusch_grant_t* g, int qm, uint32_t tti)
{
  if (m->hopping || g->hop == 2 || g->tbs <= 0) return 0; /* type-2 hopping is not applied by the reference either (hopping_enabled stays false, SubframeWorker.cc:269): the attempt fails */
  o_pusch_grant_t gg = *g;
  gg.mod = qm;
  int its = 0;
  float snr = 0;
  /* uci_cfg of this attempt, UL_Sniffer_PUSCH.cc:429-450: HARQ-ACK bits counted 4 ms earlier, aperiodic CQI (higher-layer sub-band) + 1 RI bit on request */
  /* ... with the UE's report type and betaOffset indices from the tracking database (get_ue_config_rnti, :433-435) */
  const o_ue_cfg_t uc = ue_cfg_get(w, m->rnti);
  o_uci_t uci = {m->nof_ack, m->cqi_req ? (uint32_t)o_uci_cqi_bits_type(w->cfg.cell.nof_prb, uc.cqi_type) : 0u, m->cqi_req ? 1u : 0u,
                 uc.i_offset_ack + 1u, uc.i_offset_cqi + 1u, uc.i_offset_ri + 1u};
  if (w->ul_script_fn) {
    const uint32_t call[16] = {tti, m->rnti, gg.L_prb, gg.n_prb, gg.mcs_idx, (uint32_t)qm, (uint32_t)gg.tbs, 0, m->nof_ack, m->cqi_req, uc.cqi_type,
                               uc.i_offset_ack, uc.i_offset_cqi, uc.i_offset_ri, uci.ri_bits, uci.cqi_bits};
    memset(w->payload, 0, (size_t)(gg.tbs / 8));
    const int c = w->ul_script_fn(w->ul_script_user, call, &snr, w->payload);
    w->last_ul_snr = snr;
    if (c) write_pcap_ul(w, w->payload, (uint32_t)(gg.tbs / 8), m->rnti, tti, m->is_rar);
    return c;
  }
  if (o_trace_enabled()) o_trace_begin_job(tti, m->rnti, 0, NULL, NULL, NULL, 1);
  int crc = o_pusch_decode_uci(&w->cfg.cell, &w->ulcfg, tti % 10, m->rnti, &gg, m->n_dmrs, &uci, w->ul_grid, w->cfg.max_turbo_iter, w->payload, &its, &snr);
  w->total_iters += (uint64_t)its;
  if (its > 0) w->last_ul_snr = snr; /* the channel estimate ran (an invalid grant leaves the previous estimate in place) */
  if (crc) write_pcap_ul(w, w->payload, (uint32_t)(gg.tbs / 8), m->rnti, tti, m->is_rar);
  return crc;
}

static int ulmod_find(o_worker_t* w, uint16_t rnti) /* MCSTracking::find_tracking_info_RNTI_ul, MCSTracking.cc:32-57: 5 = FULL_BUFFER */
{
  if (!w->ulmod[rnti]) return w->ulmod_count < 250 ? 1 : 5;
  w->ul_time[rnti] = w->sf_count; /* :52-53 */
  return w->ulmod[rnti];
}
static void ulmod_update(o_worker_t* w, uint16_t rnti, int mod) /* update_RNTI_ul, :71-85; add_RNTI_ul (:57-69) copies the default configuration */
{
  if (w->ulmod[rnti]) w->ulmod[rnti] = (uint8_t)mod;
  else ul_add(w, rnti, 1);
}
/* MCSTracking::update_statistic_ul, :729-754 (called when the last channel estimate reports >= 1 dB, UL_Sniffer_PUSCH.cc:571-575) */
static void ul_statistic(o_worker_t* w, uint16_t rnti, int success, int mem_mod)
{
  ul_add(w, rnti, mem_mod);
  w->ul_active[rnti]++;
  if (success) w->ul_success[rnti]++;
}
/* MCSTracking::update_database_ul, :86-176: entries idle for more than `interval` whole seconds or without a counted decode are dropped
 * (the all_database copies are statistics only) */
static void ul_update_database(o_worker_t* w)
{
  for (uint32_t r = 0; r < 65536; r++) {
    if (!w->ulmod[r]) continue;
    const uint32_t cur_interval = (w->sf_count - w->ul_time[r]) / 1000u;
    if (cur_interval > w->mcs_interval || w->ul_active[r] == 0) { w->ulmod[r] = 0; w->ulmod_count--; }
  }
  w->nof_ul_updates++;
}
uint32_t o_worker_nof_tracked_ul(o_worker_t* w) { return w->ulmod_count; }
int o_worker_tracked_mod_ul(o_worker_t* w, uint16_t rnti) { return w->ulmod ? (int)w->ulmod[rnti] : 0; }

/* PUSCH_Decoder::decode, UL_Sniffer_PUSCH.cc:389-583 (statistics / debug printing dropped) */
static void decode_pusch_list(o_worker_t* w, uint32_t tti, ulg_t* list, int n);
static void decode_pusch(o_worker_t* w, uint32_t tti)
{
  uint32_t t4 = (tti + 10240 - 4) % 10240, t6 = (tti + 10240 - 6) % 10240;
  ulslot_t* a = &w->ul_sched[t4 % 16];
  ulslot_t* r = &w->rar_sched[t6 % 16];
  int have_a = a->valid && a->tti == t4, have_r = r->valid && r->tti == t6;
  ulg_t list[192];
  int n = 0;
  if (have_a) for (int i = 0; i < a->n; i++) list[n++] = a->g[i];
  if (have_r) for (int i = 0; i < r->n; i++) list[n++] = r->g[i];
  if (have_a) a->valid = 0;
  if (have_r) r->valid = 0;
  decode_pusch_list(w, tti, list, n);
}
static void decode_pusch_list(o_worker_t* w, uint32_t tti, ulg_t* list, int n)
{
  for (int i = 0; i < n; i++) {
    ulg_t* m = &list[i];
    int valid = 1; /* investigate_valid_ul_grant, :894-918 */
    if (!m->is_rar) {
      if (m->rnti == 0) valid = 0;
      if (m->g.tbs == 0 || m->g256.tbs == 0) valid = 0;
      if (!o_ul_valid_prb(m->g.L_prb) || m->g.L_prb > 100) valid = 0;
    }
    if (!valid || m->rnti == 0) continue;
    uint32_t mcs = m->g.mcs_idx;
    int mod = ulmod_find(w, m->rnti), mem_mod = 1 /* decoding_mem.mcs_mod, UNKNOWN unless set below */;
    int ok256 = m->g256.L_prb < 110 && m->g256.L_prb > 0;
    int qm_base = m->g.mod; /* modulation of Table 8.6.1-1 with 64QAM allowed */
    int crc = 0;
#define LEARN(newmod) do { if (crc && mcs > 20 && mem_mod == 1) ulmod_update(w, m->rnti, (newmod)); } while (0)
    if (mcs > 20 && mcs < 29) {
      if (mod == 2) { mem_mod = 2; crc = pusch_attempt(w, m, &m->g, 4, tti); LEARN(2); }
      else if (mod == 3) { mem_mod = 3; crc = pusch_attempt(w, m, &m->g, qm_base, tti); LEARN(3); }
      else if (mod == 4) { mem_mod = 4; if (ok256) { crc = pusch_attempt(w, m, &m->g256, m->g256.mod, tti); LEARN(4); } }
      else if (mod == 1) {
        crc = pusch_attempt(w, m, &m->g, 4, tti); LEARN(2);
        if (!crc) {
          crc = pusch_attempt(w, m, &m->g, qm_base, tti); LEARN(3);
          if (!crc && ok256) { crc = pusch_attempt(w, m, &m->g256, m->g256.mod, tti); LEARN(4); }
        }
      }
    } else if (mcs <= 20) {
      /* decode_run's second rule (UL_Sniffer_PUSCH.cc:300-303): below MCS 21 decoding_mem.mcs_mod is never set, so a passing
       * "[PUSCH-256]" attempt with MCS > 0 calls update_RNTI_ul(256QAM_MAX) - whether the entry said 256QAM already or nothing */
      if (mod == 2 || mod == 3) crc = pusch_attempt(w, m, &m->g, qm_base > 4 ? 4 : qm_base, tti);
      else if (mod == 4) { if (ok256) { crc = pusch_attempt(w, m, &m->g256, m->g256.mod, tti); if (crc && mcs > 0) ulmod_update(w, m->rnti, 4); } }
      else if (mod == 1) {
        crc = pusch_attempt(w, m, &m->g, qm_base > 4 ? 4 : qm_base, tti);
        if (!crc && ok256) { crc = pusch_attempt(w, m, &m->g256, m->g256.mod, tti); if (crc && mcs > 0) ulmod_update(w, m->rnti, 4); }
      }
    }
#undef LEARN
    if (w->last_ul_snr >= 1.0f) ul_statistic(w, m->rnti, crc, mem_mod); /* UL_Sniffer_PUSCH.cc:571-575 */
  }
}

int o_worker_work_ul(o_worker_t* w, const ocf_t* dl_iq, const ocf_t* ul_iq, uint32_t sf_idx, uint32_t sfn, int update_meta)
{
  const o_cell_t* cell = &w->cfg.cell;
  if (!w->ul_mode || w->cfg.nof_rx != 1) return -1;
  uint32_t tti = sfn * 10 + sf_idx;
  w->sf_idx = sf_idx; w->sfn = sfn; w->records = 0;
  w->ndl = w->nul = w->nacc = w->ntemp0 = 0;
  w->dl_collision = w->ul_collision = 0;
  memset(w->rb_map_dl, 0, sizeof(w->rb_map_dl));
  memset(w->rb_map_ul, 0, sizeof(w->rb_map_ul));
  if (update_meta) update_formats(w);
  /* LTESniffer_Core.cc:473-499: every get_interval() x 1000 subframes the uplink tracking database is aged (UL mode, mcs_tracking_mode on) */
  if (w->cfg.mcs_tracking_mode && w->mcs_update_period && w->sf_count && (w->sf_count % w->mcs_update_period) == 0) ul_update_database(w);
  o_ofdm_rx(cell, dl_iq, 0, w->grid); /* DCISearch::prepareDCISearch: one rx antenna for the downlink, DCISearch.cc:592 */
  o_chest(cell, 1, sf_idx, w->grid, w->ce, &w->chest);
  w->cfi = o_pcfich_decode(cell, &w->regs, 1, sf_idx, w->grid, w->ce, w->chest.noise_avg, NULL);
  o_pdcch_llr(cell, &w->regs, 1, sf_idx, w->cfi, w->grid, w->ce, w->chest.noise_avg, w->llr);
  o_ul_fft(cell, ul_iq, w->ul_grid); /* srsran_enb_ul_fft, UL_Sniffer_PUSCH.cc:392 */
  ulslot_t* cur = &w->ul_sched[tti % 16];
  ulslot_t* rar = &w->rar_sched[tti % 16];
  cur->tti = tti; cur->valid = 1; cur->n = 0;
  rar->tti = tti; rar->valid = 1; rar->n = 0;
  if (!w->ul_configured) { /* run_ul_mode without a configuration: decode_SIB, then ULSchedule::set_SIB2 + set_config; nothing else this subframe */
    if (w->chest.snr_db > 6.0f) {
      blind_search(w);
      if (decode_sib(w)) {
        w->ulcfg.cyclic_shift = w->sib2.cyclic_shift;           /* ULSchedule.cc:143-146 */
        w->ulcfg.delta_ss = w->sib2.group_assignment_pusch;
        w->ulcfg.hopping_offset = w->sib2.pusch_hop_offset;     /* SubframeWorker.cc:271-273 */
        w->ulcfg.group_hopping_enabled = w->sib2.group_hopping_enabled;
        w->ulcfg.sequence_hopping_enabled = w->sib2.sequence_hopping_enabled;
        w->cfg.cell.pusch_hop_offset = w->sib2.pusch_hop_offset;
        w->ul_configured = 1;
        w->sib2_learned = 1;
      }
    }
    w->stats.nof_subframes++;
    w->sf_count++;
    return w->records;
  }
  if (w->chest.snr_db > 6.0f) {
    blind_search(w);
    decode_ul_mode_dl(w, rar);
    for (uint32_t i = 0; i < w->nul && cur->n < 96; i++) { /* ULSchedule::pushULSche(tti, dci_ul), SubframeWorker.cc:340 */
      ulg_t* g = &cur->g[cur->n++];
      memset(g, 0, sizeof(*g));
      g->rnti = w->ul[i].rnti; g->g = w->ul[i].g; g->g256 = w->ul[i].g256;
      g->n_dmrs = w->ul[i].dci.n_dmrs; g->hopping = 0; /* DCI 0 hopping lives in the grants (hop = 1 decoded, 2 not) */
      g->cqi_req = w->ul[i].dci.cqi_req;
      for (uint32_t di = 0; di < w->ndl; di++) /* "check nof_ack for uplink pusch decoder", SubframeWorker.cc:318-336 */
        if (w->dl[di].rnti == g->rnti) {
          if (w->dl[di].g64.nof_tb == 1) g->nof_ack = 1;
          else if (w->dl[di].g64.nof_tb == 2) g->nof_ack = 2;
        }
      if (!w->ul[i].ok) { memset(&g->g, 0, sizeof(g->g)); memset(&g->g256, 0, sizeof(g->g256)); }
    }
  }
  w->stats.nof_subframes++;
  decode_pusch(w, tti); /* SubframeWorker.cc:343-347 */
  w->sf_count++;
  return w->records;
}


/* ================================================================================================ probes for tests/test_ref_collect.py
 * DCICollection::addCandidate (add_candidate above) on given DCI bits, and what the decoders feed back between subframes.  Flat words, the layout of
 * oracle/ref_shim_search/collect_glue.cc: 64 per downlink entry, 32 per uplink entry. */
void o_worker_collect_begin(o_worker_t* w, uint32_t sfn, uint32_t sf_idx, uint32_t cfi)
{
  w->sfn = sfn; w->sf_idx = sf_idx; w->cfi = cfi;
  w->ndl = w->nul = w->nacc = 0;
  w->dl_collision = w->ul_collision = 0;
  memset(w->rb_map_dl, 0, sizeof(w->rb_map_dl));
  memset(w->rb_map_ul, 0, sizeof(w->rb_map_ul));
}
void o_worker_collect_add(o_worker_t* w, uint16_t rnti, int format, uint32_t L, uint32_t ncce, uint32_t histval, const uint8_t* payload, uint32_t nof_bits)
{
  cand_t c;
  memset(&c, 0, sizeof(c));
  c.rnti = rnti; c.msg.rnti = rnti; c.msg.format = format; c.msg.nof_bits = nof_bits;
  memcpy(c.msg.payload, payload, nof_bits);
  add_candidate(w, &c, L, ncce, histval);
}
static void put_mask(uint32_t* o, const uint8_t* prb, uint32_t n)
{
  o[0] = o[1] = o[2] = o[3] = 0;
  for (uint32_t i = 0; i < n && i < 128; i++)
    if (prb[i]) o[i >> 5] |= 1u << (i & 31);
}
static void put_dl_grant(uint32_t* o, const o_pdsch_grant_t* g, uint32_t nof_prb)
{
  o[0] = g->nof_prb; o[1] = g->nof_re; o[2] = g->nof_tb;
  put_mask(o + 3, g->prb_idx[0], nof_prb);
  put_mask(o + 7, g->prb_idx[1], nof_prb);
  for (int i = 0; i < 2; i++) {
    uint32_t* t = o + 11 + 7 * i;
    t[0] = (uint32_t)g->tb[i].enabled; t[1] = g->tb[i].enabled ? (uint32_t)g->tb[i].mod : 0; t[2] = (uint32_t)g->tb[i].tbs; t[3] = (uint32_t)g->tb[i].nof_bits;
    t[4] = (uint32_t)g->tb[i].rv; t[5] = g->tb[i].mcs_idx; t[6] = g->tb[i].cw_idx;
  }
}
static void put_ul_grant(uint32_t* o, const o_pusch_grant_t* g, const o_cell_t* cell, int ok)
{
  memset(o, 0, 9 * sizeof(uint32_t));
  if (!ok) return;
  o[0] = g->L_prb; o[1] = g->n_prb; o[2] = g->hop == 1 ? g->n_prb2 : g->n_prb; o[3] = g->hop; o[4] = g->L_prb ? (uint32_t)g->mod : 0; o[5] = (uint32_t)g->tbs;
  o[6] = (uint32_t)g->rv; o[7] = g->mcs_idx; o[8] = g->L_prb * 12u * (uint32_t)(2 * (o_nslot(cell) - 1));
}
uint32_t o_worker_collect_end(o_worker_t* w, uint32_t* dl, uint32_t dl_cap, uint32_t* ul, uint32_t ul_cap, uint16_t* map_dl, uint16_t* map_ul, uint32_t* counts2)
{
  const o_cell_t* cell = &w->cfg.cell;
  for (uint32_t i = 0; i < w->ndl && i < dl_cap; i++) {
    const dl_entry_t* e = &w->dl[i];
    uint32_t* o = dl + (size_t)i * 64;
    memset(o, 0, 64 * sizeof(uint32_t));
    o[0] = e->rnti; o[1] = (uint32_t)e->format; o[2] = (uint32_t)e->mcs_table; o[3] = e->dci_rnti; o[4] = e->dci.pid; o[5] = e->dci.pinfo; o[6] = e->dci.tb_cw_swap;
    for (int t = 0; t < 2; t++) { o[7 + 3 * t] = e->dci.tb[t].mcs_idx; o[8 + 3 * t] = (uint32_t)e->dci.tb[t].rv; o[9 + 3 * t] = e->dci.tb[t].ndi; }
    o[13] = (uint32_t)e->check;
    if (e->has64) put_dl_grant(o + 14, &e->g64, cell->nof_prb);
    if (e->has256) put_dl_grant(o + 39, &e->g256, cell->nof_prb);
  }
  for (uint32_t i = 0; i < w->nul && i < ul_cap; i++) {
    const ul_entry_t* u = &w->ul[i];
    uint32_t* o = ul + (size_t)i * 32;
    memset(o, 0, 32 * sizeof(uint32_t));
    o[0] = u->rnti; o[1] = u->dci.rnti; o[2] = u->dci.n_dmrs; o[3] = u->dci.cqi_req; o[4] = u->dci.ndi; o[5] = u->dci.tpc; o[6] = (uint32_t)u->dci.hop_type; o[7] = u->dci.riv;
    o[8] = u->dci.mcs_idx; o[9] = (uint32_t)u->dci.rv;
    put_ul_grant(o + 10, &u->g, cell, u->ok);
    put_ul_grant(o + 19, &u->g256, cell, u->ok);
    o[28] = u->ok ? u->g.L_prb : 0; o[29] = u->ok ? u->g.n_prb : 0;
  }
  for (uint32_t i = 0; i < cell->nof_prb; i++) { map_dl[i] = w->rb_map_dl[i]; map_ul[i] = w->rb_map_ul[i]; }
  counts2[0] = w->ndl; counts2[1] = w->nul;
  return (w->dl_collision ? 1u : 0u) | (w->ul_collision ? 2u : 0u);
}
void o_worker_collect_mcs_update(o_worker_t* w, uint16_t rnti, int table) { mcs_update(w, rnti, table); }
void o_worker_collect_harq_update(o_worker_t* w, uint16_t rnti, int pid, int tid, uint32_t sfn, uint32_t sf_idx, int decoded, int ndi, int rv, int tbs)
{
  struct o_harq_entity* ent = NULL;
  const int verdict = harq_is_retx(w, rnti, pid, tid, ndi, tbs, sfn, sf_idx, &ent); /* finds or takes an entity, DL_Sniffer_PDSCH.cc:954 */
  if (ent && (verdict == O_HARQ_NEW_TX || verdict == O_HARQ_RE_TX)) harq_update(w, ent, pid, tid, sfn, sf_idx, decoded, ndi, rv, tbs); /* :1008-1014 */
}
void o_worker_collect_set_hop_offset(o_worker_t* w, uint32_t n_rb_ho) { w->cfg.cell.pusch_hop_offset = n_rb_ho; }

/* ---- probes for tests/test_ref_ul_decode.py: PUSCH_Decoder::decode on a given schedule with a scripted uplink decoder ---- */
void o_worker_set_ul_script(o_worker_t* w, int (*fn)(void*, const uint32_t*, float*, uint8_t*), void* user) { w->ul_script_fn = fn; w->ul_script_user = user; }
void o_worker_set_last_ul_snr(o_worker_t* w, float snr_db) { w->last_ul_snr = snr_db; }
/* n entries x 12 words {rnti, is_rar, mcs_idx, L_prb, n_prb, modulation bits, tbs, L_prb of the 256QAM-table grant, its modulation bits, its tbs, cqi_request, nof_ack};
 * the DCI 0 entries first, then the RAR entries - the order PUSCH_Decoder::decode walks them (UL_Sniffer_PUSCH.cc:395-412) */
void o_worker_ul_decode_probe(o_worker_t* w, uint32_t tti, uint32_t n, const uint32_t* e12)
{
  ulg_t list[192];
  int k = 0;
  w->records = 0;
  for (int pass = 0; pass < 2; pass++)
    for (uint32_t i = 0; i < n && k < 192; i++) {
      const uint32_t* e = e12 + 12 * i;
      if ((int)(e[1] != 0) != pass) continue;
      ulg_t* m = &list[k++];
      memset(m, 0, sizeof(*m));
      m->rnti = (uint16_t)e[0]; m->is_rar = e[1] != 0; m->cqi_req = e[10]; m->nof_ack = e[11];
      m->g.mcs_idx = m->g256.mcs_idx = e[2]; m->g.L_prb = e[3]; m->g.n_prb = m->g256.n_prb = e[4]; m->g.mod = (int)e[5]; m->g.tbs = (int)e[6];
      m->g256.L_prb = e[7]; m->g256.mod = (int)e[8]; m->g256.tbs = (int)e[9];
    }
  decode_pusch_list(w, tti, list, k);
  w->sf_count++;
}
void o_worker_ul_update_database(o_worker_t* w) { ul_update_database(w); }
void o_worker_ul_set_ue_config(o_worker_t* w, uint16_t rnti, uint32_t i_ack, uint32_t i_cqi, uint32_t i_ri, uint32_t cqi_type)
{
  o_ue_cfg_t c = ue_cfg_get(w, rnti);
  c.i_offset_ack = i_ack; c.i_offset_cqi = i_cqi; c.i_offset_ri = i_ri; c.cqi_type = cqi_type;
  ul_add(w, rnti, 1); /* update_ue_config_rnti, UL_MODE branch (MCSTracking.cc:1464-1479) */
  c.has_ue_config = 1;
  w->uecfg[rnti] = c;
}

/* ---- probes for tests/test_ref_decode.py: PDSCH_Decoder::decode_dl_mode (decode_dl_mode above) on the collected entries with a scripted decoder ---- */
void o_worker_set_script_decoder(o_worker_t* w, int (*fn)(void*, const uint32_t*, float, uint8_t*, uint8_t*, int32_t*), void* user) { w->script_fn = fn; w->script_user = user; }
void o_worker_collect_set_now(o_worker_t* w, uint32_t subframes) { w->sf_count = subframes; }
void o_worker_collect_decode_dl_mode(o_worker_t* w) { decode_dl_mode(w); }
int o_worker_collect_find_table(o_worker_t* w, uint16_t rnti) { return mcs_find(w, rnti); } /* MCSTracking::find_tracking_info_RNTI_dl: refreshes the entry's time stamp */
void o_worker_collect_update_database(o_worker_t* w) { mcs_update_database(w); }
