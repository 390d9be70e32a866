/* oracle/_ref harness around the REFERENCE'S OWN blind DCI search (test infrastructure, NOT product; see srsran/standin.h).
 *
 * Compiled verbatim from /root/reference by oracle/Makefile.ref into _ref/libref_falcon_search.so:
 *   src/src/DCISearch.cc                         the FALCON decision tree           
 *   lib/src/phy/falcon_phch/falcon_pdcch.c       location map, CCE power, search-space validation, missed CCEs (a5, a6, a9)
 *   src/src/MetaFormats.cc                       primary / secondary format split   (a19)
 *   src/src/SubframeInfo.cc                      holder of the two per-subframe objects
 *   lib/src/util/{RNTIManager,Histogram,Interval}.cc                                (a10)
 * This file supplies what those files call and the reference keeps elsewhere:
 *   - the srsRAN primitives (absent dependency).  Two are BOUND TO THE ORACLE through function pointers set by the test: the DCI size of a
 *     format and "de-rate-match + tail-biting Viterbi + CRC remainder of one candidate".  Two are the search-space enumerations of
 *     TS 36.213 9.1.1, written here with the loops the reference itself shows in falcon_pdcch.c:49-103.  The OFDM / channel-estimation
 *     front end is a no-op: the test hands in the oracle's PDCCH soft bits.  The rest is never reached on this path and aborts.
 *   - three one-line helpers of falcon_ue_dl.c / falcon_dci.c (candidate array alloc / free, location range check);
 *   - a RECORDING DCICollection (the reference's unpacks DCI into grants through srsRAN; here only what the search hands over is kept),
 *     an idle SubframePower, a zeroed DCIBlindSearchStats;
 *   - the set-up the reference does in PhyCommon.cc:11, Phy.cc:21-26 and LTESniffer_Core.cc:397-417, and the per-subframe sequence of
 *     SubframeWorker.cc:142-170, behind a C interface for ctypes.
 * What a run proves: the reference's search code and the oracle's restatement (o_worker.c: inspect / search_subframe) take the same
 * decisions - same accepted DCI (RNTI, format, L, nCCE, bits, histogram value) in the same order, same statistics - on the same soft bits. */
#include "include/DCISearch.h"
#include "include/DCICollection.h"
#include <vector>

typedef uint32_t (*size_fn_t)(const void* ocell, int format);
typedef uint16_t (*decode_fn_t)(const float* llr, int E, int nof_bits, uint8_t* payload);
static size_fn_t g_size_fn = nullptr;
static decode_fn_t g_decode_fn = nullptr;
static const void* g_ocell = nullptr;

struct accepted_t { uint32_t rnti, format, L, ncce, nof_bits, histval; uint8_t payload[SRSRAN_DCI_MAX_BITS]; };
static std::vector<accepted_t> g_accepted;

static void unreachable(const char* what)
{
  fprintf(stderr, "ref search harness: %s is not on the search path\n", what);
  abort();
}

extern "C" {

/* ---- srsRAN primitives bound to the oracle ---- */
uint32_t srsran_dci_format_sizeof(const srsran_cell_t*, srsran_dl_sf_cfg_t*, srsran_dci_cfg_t*, srsran_dci_format_t format)
{
  return g_size_fn(g_ocell, (int)format);
}
/* srsRAN: E <= max_bits and nof_bits <= SRSRAN_DCI_MAX_BITS, then rate de-matching, Viterbi, *crc = (last 16 decoded bits) xor CRC16 */
int srsran_pdcch_dci_decode(srsran_pdcch_t* q, float* e, uint8_t* data, uint32_t E, uint32_t nof_bits, uint16_t* crc)
{
  if (!q || !data || E > q->max_bits || nof_bits > SRSRAN_DCI_MAX_BITS) return SRSRAN_ERROR_INVALID_INPUTS;
  uint16_t r = g_decode_fn(e, (int)E, (int)nof_bits, data);
  if (crc) *crc = r;
  return SRSRAN_SUCCESS;
}
int srsran_ue_dl_decode_fft_estimate(srsran_ue_dl_t*, srsran_dl_sf_cfg_t*, srsran_ue_dl_cfg_t*) { return SRSRAN_SUCCESS; }

/* ---- TS 36.213 9.1.1: common and UE-specific search spaces (same loops as falcon_pdcch.c:49-103, which only tests membership) ---- */
uint32_t srsran_pdcch_common_locations_ncce(uint32_t nof_cce, srsran_dci_location_t* c, uint32_t max_candidates)
{
  uint32_t k = 0;
  for (int l = 3; l > 1; l--) {
    uint32_t L = 1u << l;
    for (uint32_t i = 0; i < SRSRAN_MIN(nof_cce, 16) / L; i++) {
      uint32_t ncce = L * (i % (nof_cce / L));
      if (k < max_candidates && ncce + L <= nof_cce) { c[k].L = (uint32_t)l; c[k].ncce = ncce; k++; }
    }
  }
  return k;
}
uint32_t srsran_pdcch_ue_locations_ncce(uint32_t nof_cce, srsran_dci_location_t* c, uint32_t max_candidates, uint32_t sf_idx, uint16_t rnti)
{
  static const uint32_t nof_candidates[4] = {6, 6, 2, 2};
  uint32_t Yk = rnti, k = 0;
  for (uint32_t m = 0; m < sf_idx + 1; m++) Yk = (39827 * Yk) % 65537;
  for (int l = 3; l >= 0; l--) {
    uint32_t L = 1u << l;
    for (uint32_t i = 0; i < nof_candidates[l]; i++)
      if (nof_cce >= L) {
        uint32_t ncce = L * ((Yk + i) % (nof_cce / L));
        if (k < max_candidates && ncce + L <= nof_cce) { c[k].L = (uint32_t)l; c[k].ncce = ncce; k++; }
      }
  }
  return k;
}

/* ---- named by the compiled files, never reached from DCISearch::search ---- */
const char* srsran_dci_format_string(srsran_dci_format_t) { return "?"; }
void srsran_pdcch_dci_encode_conv(srsran_pdcch_t*, uint8_t*, uint32_t, uint8_t*, uint16_t) { unreachable("srsran_pdcch_dci_encode_conv"); }
int srsran_rm_conv_rx(float*, uint32_t, float*, uint32_t) { unreachable("srsran_rm_conv_rx"); return -1; }
int srsran_rm_conv_tx(uint8_t*, uint32_t, uint8_t*, uint32_t) { unreachable("srsran_rm_conv_tx"); return -1; }
int srsran_viterbi_decode_f(srsran_viterbi_t*, float*, uint8_t*, uint32_t) { unreachable("srsran_viterbi_decode_f"); return -1; }
uint32_t srsran_crc_checksum(srsran_crc_t*, uint8_t*, int) { unreachable("srsran_crc_checksum"); return 0; }
uint32_t srsran_bit_pack(uint8_t**, int) { unreachable("srsran_bit_pack"); return 0; }
void srsran_bit_fprint(FILE*, uint8_t*, int) {}

/* ---- falcon_ue_dl.c:212-219, falcon_dci.c:527-534 ---- */
dci_candidate_t* falcon_alloc_candidates(uint32_t nof_candidates) { return (dci_candidate_t*)calloc(nof_candidates, sizeof(dci_candidate_t)); }
void falcon_free_candidates(dci_candidate_t* candidates) { free(candidates); }
bool falcon_dci_location_isvalid(falcon_dci_location_t* c) { return c->L <= 3 && c->ncce <= 87; }

} /* extern "C" */

/* ---- the per-subframe objects the search writes into ---- */
DCIBlindSearchStats::DCIBlindSearchStats() :
  nof_locations(0), nof_decoded_locations(0), nof_cce(0), nof_missed_cce(0), nof_subframes(0), nof_subframe_collisions_dw(0),
  nof_subframe_collisions_up(0), time_blindsearch() {}
SubframePower::SubframePower(const srsran_cell_t& cell) : nof_prb(cell.nof_prb), max(0), min(0), rb_power_dl(cell.nof_prb, 0) {}
SubframePower::~SubframePower() {}
void SubframePower::computePower(const cf_t*) {}
DCICollection::DCICollection(const srsran_cell_t& cell_, int mcs_tracking_mode_, MCSTracking* mcs_tracking_, int harq_mode_, HARQ* harq_, ULSchedule* ulsche_) :
  cell(cell_), sfn(0), sf_idx(0), cfi(0), timestamp(), dl_collision(false), ul_collision(false), mcs_tracking_mode(mcs_tracking_mode_),
  mcs_tracking(mcs_tracking_), ulsche(ulsche_), harq_mode(harq_mode_), harq(harq_) {}
DCICollection::~DCICollection() {}
void DCICollection::setTimestamp(timeval t) { timestamp = t; }
void DCICollection::setSubframe(uint32_t sfn_, uint32_t sf_idx_, uint32_t cfi_) { sfn = sfn_; sf_idx = sf_idx_; cfi = cfi_; }
bool DCICollection::hasCollisionDL() const { return dl_collision; }
bool DCICollection::hasCollisionUL() const { return ul_collision; }
void DCICollection::addCandidate(dci_candidate_t& cand, const srsran_dci_location_t& location, uint32_t histval, srsran_dl_sf_cfg_t*, srsran_dci_cfg_t*)
{
  accepted_t a;
  a.rnti = cand.rnti; a.format = (uint32_t)cand.dci_msg.format; a.L = location.L; a.ncce = location.ncce;
  a.nof_bits = cand.dci_msg.nof_bits; a.histval = histval;
  memcpy(a.payload, cand.dci_msg.payload, sizeof(a.payload));
  g_accepted.push_back(a);
}

/* ---- the harness ---- */
struct ref_search_t {
  srsran_ue_dl_t ue;
  falcon_ue_dl_t* fq;
  RNTIManager* rm;
  DCIMetaFormats* meta;
  srsran_ue_dl_cfg_t ue_dl_cfg;
  bool shortcut;
  DCIBlindSearchStats stats;
  std::vector<float> llr;
};

extern "C" {

void ref_search_bind(void* size_fn, void* decode_fn, const void* ocell)
{
  g_size_fn = (size_fn_t)size_fn; g_decode_fn = (decode_fn_t)decode_fn; g_ocell = ocell;
}

ref_search_t* ref_search_new(uint32_t nof_prb, uint32_t nof_ports, uint32_t cell_id, uint32_t histogram_threshold, double split_ratio,
                             int skip_secondary, int enable_shortcut)
{
  ref_search_t* h = new ref_search_t();
  memset(&h->ue, 0, sizeof(h->ue));
  h->ue.cell.nof_prb = nof_prb; h->ue.cell.nof_ports = nof_ports; h->ue.cell.id = cell_id;
  h->ue.nof_rx_antennas = 1;
  h->ue.pdcch.cell = h->ue.cell;
  h->fq = (falcon_ue_dl_t*)calloc(1, sizeof(falcon_ue_dl_t));
  h->fq->q = &h->ue;
  memset(&h->ue_dl_cfg, 0, sizeof(h->ue_dl_cfg));
  /* PhyCommon.cc:11 */
  h->rm = new RNTIManager(nof_falcon_ue_all_formats, RNTI_PER_SUBFRAME, histogram_threshold);
  /* Phy.cc:21-26 */
  h->meta = new DCIMetaFormats(nof_falcon_ue_all_formats, split_ratio);
  h->meta->setSkipSecondaryMetaFormats(skip_secondary != 0);
  /* LTESniffer_Core.cc:397-417 (falcon_dci_index_of_format_in_list = position in falcon_ue_all_formats) */
  for (int pass = 0; pass < 2; pass++) {
    srsran_dci_format_t want = pass == 0 ? SRSRAN_DCI_FORMAT1A : SRSRAN_DCI_FORMAT1C;
    int idx = -1;
    for (uint32_t i = 0; i < nof_falcon_ue_all_formats; i++) if (falcon_ue_all_formats[i] == want) { idx = (int)i; break; }
    if (idx > -1) {
      h->rm->addEvergreen(SRSRAN_RARNTI_START, SRSRAN_RARNTI_END, (uint32_t)idx);
      h->rm->addEvergreen(SRSRAN_PRNTI, SRSRAN_SIRNTI, (uint32_t)idx);
    }
  }
  for (uint32_t f = 0; f < nof_falcon_ue_all_formats; f++) h->rm->addForbidden(0x0, 0x0, f);
  h->shortcut = enable_shortcut != 0;
  return h;
}

void ref_search_free(ref_search_t* h)
{
  if (!h) return;
  delete h->meta; delete h->rm; free(h->fq); delete h;
}

/* one subframe, SubframeWorker.cc:142-170 (DL_MODE).  llr: the PDCCH soft bits of the control region (72 per CCE), nof_cce of this cfi.
 * out: 6 words per accepted DCI (rnti, format, L, ncce, nof_bits, histogram value), payload_out: 128 bytes per accepted DCI (may be NULL).
 * Returns the number of accepted DCI (may exceed cap; only cap are written), -1 when the search did not run (SNR gate, DCISearch.cc:566). */
int ref_search_subframe(ref_search_t* h, const float* llr, uint32_t nof_cce, uint32_t cfi, uint32_t sf_idx, uint32_t sfn, float snr_db,
                        int update_meta_formats, uint32_t* out, uint8_t* payload_out, uint32_t cap)
{
  if (update_meta_formats) h->meta->update_formats();
  h->llr.assign(llr, llr + (size_t)nof_cce * 72);
  h->llr.resize((size_t)nof_cce * 72 + 8 * 72, 0.f);
  for (int i = 0; i < 3; i++) { h->ue.pdcch.nof_cce[i] = 0; h->ue.pdcch.nof_regs[i] = 0; }
  if (cfi >= 1 && cfi <= 3) { h->ue.pdcch.nof_cce[cfi - 1] = nof_cce; h->ue.pdcch.nof_regs[cfi - 1] = nof_cce * 9; }
  h->ue.pdcch.llr = h->llr.data();
  h->ue.pdcch.max_bits = nof_cce * 72;
  h->ue.chest_res.snr_db = snr_db;
  srsran_dl_sf_cfg_t sf;
  memset(&sf, 0, sizeof(sf));
  sf.tti = sfn * 10 + sf_idx; sf.cfi = cfi;
  g_accepted.clear();
  int ret;
  {
    SubframeInfo subframeInfo(h->ue.cell, 1, nullptr, 0, nullptr, nullptr);
    DCISearch dciSearch(*h->fq, *h->meta, *h->rm, subframeInfo, sf_idx, sfn, &sf, &h->ue_dl_cfg);
    dciSearch.setShortcutDiscovery(h->shortcut);
    ret = dciSearch.search();
    if (ret == SRSRAN_SUCCESS) {
      DCIBlindSearchStats& s = dciSearch.getStats();
      h->stats.nof_locations += s.nof_locations; h->stats.nof_decoded_locations += s.nof_decoded_locations; h->stats.nof_cce += s.nof_cce;
      h->stats.nof_missed_cce += s.nof_missed_cce; h->stats.nof_subframes += s.nof_subframes;
    }
  }
  if (ret != SRSRAN_SUCCESS) return -1;
  uint32_t n = (uint32_t)g_accepted.size();
  for (uint32_t i = 0; i < n && i < cap; i++) {
    const accepted_t& a = g_accepted[i];
    uint32_t* o = out + 6 * i;
    o[0] = a.rnti; o[1] = a.format; o[2] = a.L; o[3] = a.ncce; o[4] = a.nof_bits; o[5] = a.histval;
    if (payload_out) memcpy(payload_out + (size_t)i * SRSRAN_DCI_MAX_BITS, a.payload, SRSRAN_DCI_MAX_BITS);
  }
  return (int)n;
}

/* a temporary C-RNTI of a decoded random-access response: DL_Sniffer_PDSCH.cc:782-797 -> RNTIManager::activateAndRefresh(.., 0, RM_ACT_RAR) */
void ref_search_activate_rar(ref_search_t* h, uint16_t t_crnti) { h->rm->activateAndRefresh(t_crnti, 0, RM_ACT_RAR); }

/* state probes */
uint32_t ref_search_meta_formats(ref_search_t* h, uint32_t* out18)
{
  uint32_t np = h->meta->getNofPrimaryMetaFormats(), ns = h->meta->getNofSecondaryMetaFormats();
  for (uint32_t i = 0; i < np; i++) out18[i] = h->meta->getPrimaryMetaFormats()[i]->global_index;
  for (uint32_t i = 0; i < ns; i++) out18[9 + i] = h->meta->getSecondaryMetaFormats()[i]->global_index;
  return np | (ns << 8);
}
void ref_search_stats(ref_search_t* h, uint32_t* out5)
{
  out5[0] = h->stats.nof_locations; out5[1] = h->stats.nof_decoded_locations; out5[2] = h->stats.nof_cce; out5[3] = h->stats.nof_missed_cce;
  out5[4] = h->stats.nof_subframes;
}
uint32_t ref_search_rnti_frequency(ref_search_t* h, uint16_t rnti, uint32_t format_idx) { return h->rm->getFrequency(rnti, format_idx); }
int ref_search_rnti_reason(ref_search_t* h, uint16_t rnti) { return (int)h->rm->getActivationReason(rnti); }
uint32_t ref_search_validate_location(uint32_t nof_cce, uint32_t ncce, uint32_t l, uint32_t nsubframe, uint16_t rnti)
{
  return srsran_pdcch_validate_location(nof_cce, ncce, l, nsubframe, rnti);
}
/* the two search-space enumerations written in this file against the reference's own statement of them: srsran_pdcch_ue_locations_check
 * (falcon_pdcch.c:49-103) answers whether a first CCE belongs to the UE-specific or the common search space of an RNTI in a subframe */
uint32_t ref_search_locations_check(uint32_t nof_cce, uint32_t nsubframe, uint16_t rnti, uint32_t ncce)
{
  srsran_pdcch_t q;
  memset(&q, 0, sizeof(q));
  q.nof_cce[0] = nof_cce;
  return srsran_pdcch_ue_locations_check(&q, nsubframe, 1, rnti, ncce);
}
uint32_t ref_search_glue_locations(uint32_t nof_cce, uint32_t nsubframe, uint16_t rnti, uint32_t* out2 /* (L, ncce) pairs, room for 22 */)
{
  srsran_dci_location_t loc[22];
  uint32_t n = srsran_pdcch_ue_locations_ncce(nof_cce, loc, 22, nsubframe, rnti);
  n += srsran_pdcch_common_locations_ncce(nof_cce, &loc[n], 22 - n);
  for (uint32_t i = 0; i < n; i++) { out2[2 * i] = loc[i].L; out2[2 * i + 1] = loc[i].ncce; }
  return n;
}

} /* extern "C" */
