// lsn_engine.cc - the batched GPU pipeline behind Phy / SubframeWorker (see lsn_engine.h).
// Control flow mirrors the reference's worker, re-staged for batches of subframes:
//   SubframeWorker::work / run_dl_mode      /root/reference/src/src/SubframeWorker.cc:142-235
//   DCISearch::search / recursive / inspect /root/reference/src/src/DCISearch.cc:102-578
//   DCICollection::addCandidate             /root/reference/src/src/DCICollection.cc:97-298
//   PDSCH_Decoder::decode_dl_mode           /root/reference/src/src/DL_Sniffer_PDSCH.cc:881-1291
// Product code: no CPU fallback, nothing from oracle/ is included or linked.
#include "lsn_engine.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <string>

#define HIP_CHECK(x)                                                                                       \
  do {                                                                                                     \
    hipError_t _e = (x);                                                                                   \
    if (_e != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #x); \
  } while (0)

size_t lsn_turbo_lds_bytes(uint32_t K);

namespace lsn {

static double now_ms()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------------------------------------ life cycle
Engine::Engine(const lsn_phy_cfg_t& c) : cfg(c)
{
  if (cfg.nof_rx_antennas < 1 || cfg.nof_rx_antennas > LSN_MAX_RX) throw std::invalid_argument("nof_rx_antennas");
  if (cfg.harq_mode != 0) throw std::invalid_argument("harq_mode");  // ArgManager.cc:50: always 0 in the reference
  max_batch = cfg.max_batch ? cfg.max_batch : 64;
  if (cfg.max_turbo_iterations <= 0) cfg.max_turbo_iterations = 12;  // SubframeWorker.cc:365
  if (cfg.meta_format_split_ratio <= 0.0) cfg.meta_format_split_ratio = 0.99;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device");
  HIP_CHECK(hipSetDevice(cfg.device));
  rnti_manager.reset(new RNTIManager(NOF_FORMATS, 304 / 5, cfg.histogram_threshold));  // PhyCommon.cc:11
  meta_formats.reset(new DCIMetaFormats(NOF_FORMATS, cfg.meta_format_split_ratio));
  meta_formats->setSkipSecondaryMetaFormats(cfg.skip_secondary_meta_formats != 0);
  for (auto& e : ev) HIP_CHECK(hipEventCreate(&e));
}

Engine::~Engine()
{
  (void)hipDeviceSynchronize();
  freeDevice();
  for (auto& e : ev) (void)hipEventDestroy(e);
}

int Engine::setCell(const lsn_cell_t& c)
{
  static const uint32_t ng_x6[4] = {1, 3, 6, 12};
  if (c.cp != 0 || c.frame_type != 0 || c.phich_length != 0 || c.phich_resources > 3) return LSN_ERROR_INVALID_INPUTS;
  if (c.nof_ports < 1 || c.nof_ports > 2 || c.id > 503) return LSN_ERROR_INVALID_INPUTS;
  switch (c.nof_prb) { case 6: case 15: case 25: case 50: case 100: break; default: return LSN_ERROR_INVALID_INPUTS; }
  (void)hipDeviceSynchronize();
  freeDevice();
  cell.nof_prb = c.nof_prb; cell.nof_ports = c.nof_ports; cell.id = c.id; cell.phich_ng_x6 = ng_x6[c.phich_resources];
  buildTables();
  cell_set = true;
  return LSN_SUCCESS;
}

void Engine::setupDefaultIntervals()
{
  // LTESniffer_Core.cc:402-417
  rnti_manager->addEvergreen(RARNTI_START, RARNTI_END, FORMAT1A);
  rnti_manager->addEvergreen(PRNTI, SIRNTI, FORMAT1A);
  rnti_manager->addEvergreen(RARNTI_START, RARNTI_END, FORMAT1C);
  rnti_manager->addEvergreen(PRNTI, SIRNTI, FORMAT1C);
  for (uint32_t f = 0; f < NOF_FORMATS; f++) rnti_manager->addForbidden(0, 0, f);
}

// ------------------------------------------------------------------------------------------------ stage A
void Engine::stageA(const void* d_iq, uint32_t nsf, hipStream_t st)
{
  std::vector<uint32_t> sfidx(nsf);
  for (uint32_t i = 0; i < nsf; i++) sfidx[i] = ctx[i].sf_idx;
  HIP_CHECK(hipMemcpyAsync(d_sfidx, sfidx.data(), nsf * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  const cf32* iq = (const cf32*)d_iq;
  auto timed = [&](int k, auto&& fn) {
    HIP_CHECK(hipEventRecord(ev[2 * k], st));
    fn();
    HIP_CHECK(hipEventRecord(ev[2 * k + 1], st));
    perf.kernel_launches[k]++;
  };
  timed(LSN_K_OFDM, [&] { lsn_launch_ofdm(cd, iq, d_dphi, d_grid, nsf, st); });
  timed(LSN_K_CHEST, [&] { lsn_launch_chest(cd, d_grid, d_sfidx, d_ce, d_chest_raw, nsf, st); });
  timed(LSN_K_CHEST_FIN, [&] { lsn_launch_chest_fin(cd, d_chest_raw, d_chest, nsf, st); });
  timed(LSN_K_PCFICH, [&] { lsn_launch_pcfich(cd, d_grid, d_ce, d_chest, d_sfidx, d_cfi, d_pcfich_corr, nsf, st); });
  timed(LSN_K_PDCCH_LLR, [&] { lsn_launch_pdcch_llr(cd, d_grid, d_ce, d_chest, d_sfidx, d_cfi, d_llr, nsf, st); });
  timed(LSN_K_CCE_POWER, [&] { lsn_launch_cce_power(cd, d_llr, d_cfi, d_ccepow, nsf, st); });
  timed(LSN_K_VITERBI, [&] { lsn_launch_viterbi(cd, d_llr, d_ccepow, d_cfi, d_cand, nsf, st); });
  timed(LSN_K_RB_POWER, [&] { lsn_launch_rb_power(cd, d_grid, d_rbp, nsf, st); });
  HIP_CHECK(hipMemcpyAsync(h_cand, d_cand, (size_t)nsf * LSN_MAX_LOC * LSN_MAX_SIZES * sizeof(LsnCand), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(h_ccepow, d_ccepow, (size_t)nsf * LSN_CCE_STRIDE * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(h_chest, d_chest, (size_t)nsf * sizeof(LsnChest), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(h_cfi, d_cfi, (size_t)nsf * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(h_rbp, d_rbp, (size_t)nsf * 128 * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  static const int ks[] = {LSN_K_OFDM, LSN_K_CHEST, LSN_K_CHEST_FIN, LSN_K_PCFICH, LSN_K_PDCCH_LLR, LSN_K_CCE_POWER, LSN_K_VITERBI, LSN_K_RB_POWER};
  for (int k : ks) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]) == hipSuccess) perf.kernel_ms[k] += ms;
  }
  const uint64_t A = cfg.nof_rx_antennas, P = cell.nof_ports;
  for (uint32_t i = 0; i < nsf; i++) {
    SubframeCtx& c = ctx[i];
    c.cfi = h_cfi[i];
    // host-side scalars, same expressions as the device-free part of the estimator
    c.snr_db = 10.0f * log10f(h_chest[i].rsrp_avg / h_chest[i].noise_avg);
    c.cfo_hz = atan2f(h_chest[i].corr_i, h_chest[i].corr_r) / (2.0f * (float)M_PI * 0.0005f);
    perf.algo_bytes += A * cd.sflen * 8ull + 2ull * A * 14ull * cd.nre * 8ull + 2ull * P * A * 14ull * cd.nre * 8ull +
                       2ull * cd.nof_cce[c.cfi - 1] * 72ull * 4ull;
  }
}

// ------------------------------------------------------------------------------------------------ FALCON search (stage B)
// srsran_pdcch_decode_msg_limit_avg_llr_power (falcon_pdcch.c:110-170) as a lookup in the exhaustive candidate table
void Engine::decodeCandidate(uint32_t sf, const FalconLocation& loc, DciFormat format, DciCandidate& cand)
{
  const LsnCand& c = h_cand[((size_t)sf * LSN_MAX_LOC + loc.index) * LSN_MAX_SIZES + size_index_of_format[format]];
  perf.nof_candidates_decoded++;
  if (!c.flags) return;
  const uint32_t n = size_of_format[format];
  for (uint32_t i = 0; i < n; i++) cand.msg.payload[i] = (uint8_t)((c.bits >> (63 - i)) & 1ull);
  cand.msg.nof_bits = n;
  cand.rnti = (uint16_t)c.rnti;
  if (format == FORMAT0 || format == FORMAT1A) cand.msg.format = cand.msg.payload[0] == 0 ? FORMAT0 : FORMAT1A;  // falcon_pdcch.c:147-148
  else cand.msg.format = format;
}

// DCICollection::addCandidate (DCICollection.cc:97-298) + srsran_dci_msg_to_trace_timestamp (falcon_dci.c:148-352).
// Both MCS tables' grants are computed here; which of them "exists" for the reference is resolved at commit time,
// when the MCS-tracking state of this subframe is known.
void Engine::addCandidate(uint32_t sf, const DciCandidate& cand, uint32_t L, uint32_t ncce, uint32_t histval)
{
  SubframeCtx& c = ctx[sf];
  const DciFormat fmt = cand.msg.format;
  if (c.accepted.size() < 64 * 6) {
    const uint32_t a[6] = {cand.rnti, (uint32_t)fmt, L, ncce, cand.msg.nof_bits, histval};
    c.accepted.insert(c.accepted.end(), a, a + 6);
  }
  if (fmt == FORMAT0) {
    if (c.ul.size() >= 64) return;
    UlEntry u;
    u.rnti = cand.rnti; u.nof_bits = cand.msg.nof_bits; u.L = L; u.ncce = ncce; u.histval = histval;
    u.dci.L = L; u.dci.ncce = ncce;
    u.ok = cand.msg.payload[0] == 0 && dci_msg_unpack_pusch(cell, cand.msg.payload, cand.msg.nof_bits, cand.rnti, u.dci) &&
           ra_ul_dci_to_grant(cell, u.dci, u.grant);
    if (u.ok)
      for (uint32_t i = 0; i < u.grant.L_prb; i++) {  // DCICollection.cc:275-280
        if (rb_map_ul[u.grant.n_prb + i] != 0) ul_collision = true;
        rb_map_ul[u.grant.n_prb + i] = cand.rnti;
      }
    c.ul.push_back(u);
    return;
  }
  if (c.dl.size() >= 64) return;
  DlEntry e;
  e.rnti = cand.rnti; e.format = fmt; e.nof_bits = cand.msg.nof_bits; e.L = L; e.ncce = ncce; e.histval = histval;
  e.dci.L = L; e.dci.ncce = ncce;
  e.unpack_ok = dci_msg_unpack_pdsch(cell, cand.msg.payload, cand.msg.nof_bits, fmt, cand.rnti, e.dci);
  if (e.unpack_ok) {
    e.ok64 = dl_sniffer_ra_dl_dci_to_grant(cell, c.sf_idx, c.cfi, false, e.dci, e.grant64);
    e.ok256 = dl_sniffer_ra_dl_dci_to_grant(cell, c.sf_idx, c.cfi, true, e.dci, e.grant256);
    for (uint32_t rb = 0; rb < cell.nof_prb; rb++)  // DCICollection.cc:215-223 (the PRB set does not depend on the MCS table)
      if (e.grant64.prb_idx[0][rb]) {
        if (rb_map_dl[rb] != 0) dl_collision = true;
        rb_map_dl[rb] = cand.rnti;
      }
    for (int i = 0; i < 2; i++) {  // DCICollection.cc:252-259
      if (e.grant64.tb[i].nof_bits <= 0) e.grant64.tb[i].enabled = false;
      if (e.grant256.tb[i].nof_bits <= 0) e.grant256.tb[i].enabled = false;
    }
  }
  c.dl.push_back(e);
}

// DCISearch::inspect_dci_location_recursively, DCISearch.cc:102-447
int Engine::inspect_dci_location_recursively(uint32_t sf, CceMap* cce_map, uint32_t ncce, uint32_t L, uint32_t max_depth, MetaFormat** metas,
                                             uint32_t nof_formats, uint32_t enable_discovery, const DciCandidate* parent_cand)
{
  SubframeCtx& c = ctx[sf];
  int hist_max_format_idx = -1;
  uint32_t hist_max_format_value = 0, nof_cand_above_threshold = 0;
  DciCandidate cand[NOF_FORMATS];
  FalconLocation* loc = cce_map[ncce].location[L];
  if (!(loc && !loc->occupied && !loc->checked && loc->sufficient_power)) return 0;  // :124-127

  for (uint32_t fi = 0; fi < nof_formats; fi++) {
    decodeCandidate(sf, *loc, metas[fi]->format, cand[fi]);
    stats.nof_decoded_locations++;
    if (rnti_manager->getActivationReason(cand[fi].rnti) == RM_ACT_RAR && cand[fi].msg.format == FORMAT0) {  // :139-158
      bool add = true;
      for (auto& t : temp_dci0)
        if (t.format == cand[fi].msg.format && t.rnti == cand[fi].rnti && t.ncce == ncce) add = false;
      if (add && temp_dci0.size() < 64) temp_dci0.push_back({cand[fi].rnti, L, ncce, cand[fi].msg.format, cand[fi]});
    }
    if (metas[fi]->format != cand[fi].msg.format) { cand[fi].rnti = 0; continue; }  // :163
    if (metas[fi]->format == FORMAT1C && cand[fi].rnti > RARNTI_END && cand[fi].rnti < PRNTI) { cand[fi].rnti = 0; continue; }  // :174
    if (cand[fi].rnti > RARNTI_START && cand[fi].rnti < RARNTI_END)  // :181-197
      if (metas[fi]->format != FORMAT1A && metas[fi]->format != FORMAT1C) { cand[fi].rnti = 0; continue; }
    if (enable_discovery && parent_cand != nullptr && parent_cand[fi].rnti == cand[fi].rnti &&
        !rnti_manager->isForbidden(cand[fi].rnti, metas[fi]->global_index))  // :200-211 (shortcut discovery)
      return -((int)fi + 1);
    cand[fi].search_space_match_result = pdcch_validate_location(cd.nof_cce[c.cfi - 1], ncce, L, c.sf_idx, cand[fi].rnti);  // :214
    if (cand[fi].search_space_match_result == 0) { cand[fi].rnti = 0; continue; }
    if (rnti_manager->validateAndRefresh(cand[fi].rnti, metas[fi]->global_index)) {  // :245-250
      nof_cand_above_threshold++;
      hist_max_format_idx = (int)fi;
      hist_max_format_value = rnti_manager->getFrequency(cand[fi].rnti, metas[fi]->global_index);
    }
  }
  if (nof_cand_above_threshold > 1) {  // :255-280
    hist_max_format_idx = -1;
    uint32_t hmax = 0;
    for (uint32_t fi = 0; fi < nof_formats; fi++)
      if (cand[fi].rnti != 0) {
        const uint32_t h = rnti_manager->getFrequency(cand[fi].rnti, metas[fi]->global_index);
        if (h > hmax) { hmax = h; hist_max_format_idx = (int)fi; hist_max_format_value = h; }
      }
    if (hist_max_format_idx == -1) nof_cand_above_threshold = 0;
  }
  loc->checked = true;  // :282
  int disamb = 0;
  if (nof_cand_above_threshold > 0 && cand[hist_max_format_idx].search_space_match_result == 1) {  // :288-298
    if (L > 0 && max_depth > 0)
      disamb = inspect_dci_location_recursively(sf, cce_map, ncce + (1u << (L - 1)), L - 1, max_depth - 1, metas, nof_formats, 0, nullptr);
  } else if (nof_cand_above_threshold == 0) {  // :302-368
    int rr = 0;
    if (L > 0 && max_depth > 0) {
      rr += inspect_dci_location_recursively(sf, cce_map, ncce, L - 1, max_depth - 1, metas, nof_formats, enable_discovery, cand);
      if (rr < 0) {
        hist_max_format_idx = -rr - 1;
        hist_max_format_value = rnti_manager->getFrequency(cand[hist_max_format_idx].rnti, metas[hist_max_format_idx]->global_index);
        nof_cand_above_threshold = 1;
        if (cand[hist_max_format_idx].search_space_match_result == 1) {
          const uint32_t md = max_depth < 99 ? max_depth : 99;
          disamb = inspect_dci_location_recursively(sf, cce_map, ncce + (1u << (L - 1)), L - 1, md - 1, metas, nof_formats, 0, nullptr);
        }
        rnti_manager->activateAndRefresh(cand[hist_max_format_idx].rnti, metas[hist_max_format_idx]->global_index, RM_ACT_SHORTCUT);
      } else {
        rr += inspect_dci_location_recursively(sf, cce_map, ncce + (1u << (L - 1)), L - 1, max_depth - 1, metas, nof_formats, enable_discovery, nullptr);
      }
    }
    if (rr == 0) {
      if (enable_discovery)
        for (uint32_t fi = 0; fi < nof_formats; fi++)
          if (cand[fi].rnti != 0) rnti_manager->addCandidate(cand[fi].rnti, metas[fi]->global_index);
      return 0;
    } else if (rr > 0) {
      return rr;
    }
  }
  if (nof_cand_above_threshold > 0) {  // :371-439
    loc->used = true;
    for (uint32_t ci = ncce; ci < ncce + (1u << L); ci++)
      for (int a = 0; a < 4; a++)
        if (cce_map[ci].location[a]) { cce_map[ci].location[a]->occupied = true; cce_map[ci].location[a]->checked = true; }
    DciCandidate& best = cand[hist_max_format_idx];
    rnti_manager->addCandidate(best.rnti, metas[hist_max_format_idx]->global_index);
    metas[hist_max_format_idx]->hits++;
    const uint32_t Ld = disamb > 0 ? L - 1 : L;
    if (best.rnti != 0) {
      bool add = true;
      if (best.msg.format == FORMAT0)
        for (auto& t : temp_dci0)
          if (t.format == FORMAT0 && t.rnti == best.rnti && t.ncce == ncce) add = false;
      if (add) addCandidate(sf, best, Ld, ncce, hist_max_format_value);
      for (auto& t : temp_dci0)  // :422-432
        addCandidate(sf, t.cand, t.L, t.ncce, rnti_manager->getFrequency(t.rnti, (uint32_t)t.format));
      temp_dci0.clear();
    }
    return 1 + disamb;
  }
  return 0;
}

// DCISearch::recursive_blind_dci_search, DCISearch.cc:449-528
void Engine::recursive_blind_dci_search(uint32_t sf)
{
  SubframeCtx& c = ctx[sf];
  CceMap cce_map[LSN_MAX_NUM_OF_CCE];
  std::memset(cce_map, 0, sizeof(cce_map));
  const uint32_t ncce = cd.nof_cce[c.cfi - 1];
  const uint32_t lim = std::min<uint32_t>(ncce, LSN_MAX_NUM_OF_CCE);
  stats.nof_cce += ncce;
  uint32_t k = 0;
  for (int l = 3; l >= 0; l--) {  // srsran_pdcch_ue_locations_all_map, falcon_pdcch.c:321-356
    const uint32_t L = 1u << l;
    for (uint32_t i = 0; i < lim / L; i++)
      if (k < LSN_MAX_LOC) {
        FalconLocation& f = locations[k];
        f = FalconLocation{(uint32_t)l, L * (i % (ncce / L)), false, false, false, true, k};
        for (uint32_t m = f.ncce; m < f.ncce + L; m++) cce_map[m].location[l] = &f;
        k++;
      }
  }
  const uint32_t nloc = k;
  stats.nof_locations += nloc;
  for (uint32_t cc = 0; cc < lim; cc++) {  // srsran_pdcch_cce_avg_llr_power, falcon_pdcch.c:595-620
    cce_map[cc].power = h_ccepow[sf * LSN_CCE_STRIDE + cc];
    if (cce_map[cc].power < 0.7f)
      for (int a = 0; a < 4; a++)
        if (cce_map[cc].location[a]) cce_map[cc].location[a]->sufficient_power = false;
  }
  for (uint32_t i = 0; i < nloc; i++)
    inspect_dci_location_recursively(sf, cce_map, locations[i].ncce, locations[i].L, 99, meta_formats->getPrimaryMetaFormats(),
                                     meta_formats->getNofPrimaryMetaFormats(), 1, nullptr);
  if (!meta_formats->skipSecondaryMetaFormats()) {
    for (uint32_t i = 0; i < nloc; i++) locations[i].checked = false;
    for (uint32_t i = 0; i < nloc; i++)
      inspect_dci_location_recursively(sf, cce_map, locations[i].ncce, locations[i].L, 99, meta_formats->getSecondaryMetaFormats(),
                                       meta_formats->getNofSecondaryMetaFormats(), 1, nullptr);
  }
  if (dl_collision) stats.nof_subframe_collisions_dw++;
  if (ul_collision) stats.nof_subframe_collisions_up++;
  uint32_t missed = 0;  // falcon_pdcch.c:561-593
  for (uint32_t cc = 0; cc < lim; cc++) {
    if (cce_map[cc].power < 0.7f) continue;
    bool m = true;
    for (int a = 0; a < 4; a++)
      if (cce_map[cc].location[a] && cce_map[cc].location[a]->used) { m = false; break; }
    if (m) missed++;
  }
  stats.nof_missed_cce += missed;
  rnti_manager->stepTime();
}

static const char* rnti_name(uint16_t r)  // DL_Sniffer_PDSCH.cc:1398-1418
{
  if (r == SIRNTI) return "SI_RNTI";
  if (r == PRNTI) return "P_RNTI";
  if (r > RARNTI_START && r < RARNTI_END) return "RA_RNTI";
  return "C_RNTI";
}

void Engine::searchSubframe(uint32_t sf, bool update_meta)
{
  SubframeCtx& c = ctx[sf];
  temp_dci0.clear();
  dl_collision = ul_collision = false;
  std::fill(rb_map_dl.begin(), rb_map_dl.end(), 0);
  std::fill(rb_map_ul.begin(), rb_map_ul.end(), 0);
  if (update_meta) meta_formats->update_formats();  // SubframeWorker.cc:148-151
  c.searched = c.snr_db > 6.0f;                     // DCISearch.cc:568-574
  if (c.searched) recursive_blind_dci_search(sf);
  stats.nof_subframes++;
  est_cfo = c.cfo_hz;  // SubframeWorker.cc:203
  if (!c.searched) return;
  // RAR grants feed the RNTI manager before the next subframe is searched (DL_Sniffer_PDSCH.cc:782-797): decode them now
  for (auto& e : c.dl) {
    if (rnti_name(e.rnti)[0] != 'R') continue;
    const bool dci_ok = e.unpack_ok && e.ok64;
    const bool two_tb = e.grant64.nof_tb == 2;
    if (!(e.grant64.tb[0].tbs > 0 && dci_ok && !(cfg.nof_rx_antennas == 1 && two_tb))) continue;
    const int j = newJob(sf, e, 0);
    e.job[0] = j;
    if (j < 0) continue;
    ensureJob(j, cur_stream);
    perf.nof_ondemand_decodes++;
    for (int tb = 0; tb < 2; tb++) {
      const int len = jobs[j].grant.tb[tb].tbs / 8;
      if (jobs[j].crc[tb] && len > 0) unpackRar(h_payload.data() + jobs[j].payload_off[tb], len, true);
    }
  }
}

// MAC RAR PDU (TS 36.321 6.1.5) walked like srsran::rar_pdu; DL_Sniffer_PDSCH.cc:782-797
void Engine::unpackRar(const uint8_t* p, int len, bool at_search)
{
  int nsub = 0, pos = 0;
  bool is_rapid[32];
  while (pos < len && nsub < 32) {
    const uint8_t b = p[pos++];
    is_rapid[nsub++] = (b & 0x40) != 0;
    if (!(b & 0x80)) break;
  }
  for (int i = 0; i < nsub; i++) {
    uint16_t t_crnti = 0;
    if (is_rapid[i]) {
      if (pos + 6 > len) break;
      t_crnti = (uint16_t)((p[pos + 4] << 8) | p[pos + 5]);
      pos += 6;
    }
    if (at_search) rnti_manager->activateAndRefresh(t_crnti, 0, RM_ACT_RAR);
    else mcs_tracking.update_rar_time_crnti(t_crnti);
  }
}

// ------------------------------------------------------------------------------------------------ stage C planning
// one srsran_ue_dl_decode_pdsch call = one job; returns -1 when dl_sniffer_config_mimo rejects the grant
int Engine::newJob(uint32_t sf, const DlEntry& e, int table)
{
  DecodeJob j;
  j.sf = sf; j.rnti = e.rnti;
  j.grant = table ? e.grant256 : e.grant64;
  if (dl_sniffer_config_mimo(cell, e.format, e.dci, j.grant) != 0) return -1;
  if (e.dci.tb[0].rv < 0 && e.rnti == SIRNTI) j.grant.tb[0].rv = 0;  // DL_Sniffer_PDSCH.cc:891-897
  jobs.push_back(j);
  return (int)jobs.size() - 1;
}

template <typename T>
static void grow(T*& p, size_t& cap, size_t need)
{
  if (need <= cap) return;
  HIP_CHECK(hipDeviceSynchronize());
  if (p) HIP_CHECK(hipFree(p));
  cap = need + need / 2 + 1024;
  HIP_CHECK(hipMalloc((void**)&p, cap * sizeof(T)));
}

void Engine::runJobs(std::vector<int>& ids, hipStream_t st)
{
  std::vector<int> todo;
  for (int j : ids)
    if (j >= 0 && !jobs[j].done && !jobs[j].planned) { jobs[j].planned = true; todo.push_back(j); }
  if (todo.empty()) return;
  const uint32_t nprb = cell.nof_prb;
  h_jobs.clear(); h_cbs.clear();
  size_t llr_n = 0, prefix_n = 0;
  const size_t pay0 = h_payload.size();
  size_t pay_n = pay0;
  struct TbRef { int job, tb; uint32_t cb_first, cb_count; uint32_t bits[16]; };
  std::vector<TbRef> tbrefs;
  for (int jid : todo) {
    DecodeJob& j = jobs[jid];
    const PdschGrant& g = j.grant;
    const SubframeCtx& c = ctx[j.sf];
    LsnGrantDev d{};
    d.sf = j.sf; d.sf_idx = c.sf_idx; d.l0 = c.cfi + (nprb <= 10 ? 1u : 0u);
    for (int s = 0; s < 2; s++)
      for (uint32_t rb = 0; rb < nprb; rb++)
        if (g.prb_idx[s][rb]) d.prb_mask[s][rb >> 5] |= 1u << (rb & 31);
    d.nof_re = g.nof_re; d.tx_scheme = (uint32_t)g.tx_scheme; d.pmi = g.pmi; d.nof_layers = g.nof_layers;
    for (int i = 0; i < 2; i++)
      if (g.tb[i].enabled) d.qm[g.tb[i].cw_idx & 1] = (uint32_t)g.tb[i].mod;
    // demodulation possible? (o_pdsch_demod / srsran_pdsch_decode preconditions)
    bool demod_ok = (g.tb[0].enabled || g.tb[1].enabled) && g.nof_re > 0;
    if (g.tx_scheme == TXSCHEME_SPATIALMUX || g.tx_scheme == TXSCHEME_CDD) {
      if (cell.nof_ports < 2) demod_ok = false;
      if (g.nof_layers != 1 && cfg.nof_rx_antennas < 2) demod_ok = false;
    }
    if (g.tx_scheme == TXSCHEME_DIVERSITY && cell.nof_ports < 2) demod_ok = false;
    if (!demod_ok) { j.done = true; continue; }
    for (int q = 0; q < 2; q++) {
      d.cinit[q] = ((uint32_t)j.rnti << 14) | ((uint32_t)q << 13) | (c.sf_idx << 9) | cell.id;
      d.llr_off[q] = (uint32_t)llr_n;
      if (d.qm[q]) llr_n += ((size_t)g.nof_re * d.qm[q] + 7) & ~(size_t)7;
    }
    d.prefix_off = (uint32_t)prefix_n;
    prefix_n += 14 * nprb + 16;
    // power allocation 36.213 5.2 with p_a = 0 dB (MCSTracking.cc:1536), p_b = 1 (SubframeWorker.cc:372)
    const float rho_a = powf(10.0f, 0.0f / 20.0f);
    const float rho_b = cell.nof_ports == 1 ? rho_a * sqrtf(0.8f) : rho_a;
    d.inv_amp_a = 1.0f / rho_a; d.inv_amp_b = 1.0f / rho_b;
    // transport blocks -> code blocks (36.212 5.1.2, 5.1.4.1.2)
    j.cb_first = (uint32_t)h_cbs.size();
    for (int i = 0; i < 2; i++) {
      j.cb_count[i] = 0;
      const GrantTb& tb = g.tb[i];
      if (!(tb.enabled && tb.tbs > 0)) continue;
      CbSegm s;
      const int Qm = tb.mod, G = tb.nof_bits, NL = g.tx_scheme == TXSCHEME_DIVERSITY ? 2 : 1;
      if (!cbsegm(tb.tbs, s) || Qm <= 0 || G <= 0) continue;
      const int Gp = G / (NL * Qm), gamma = Gp % s.C;
      j.payload_off[i] = (uint32_t)pay_n;
      TbRef ref{jid, i, (uint32_t)h_cbs.size(), (uint32_t)s.C, {}};
      int rp = 0;
      uint32_t wp = 0;
      for (int r = 0; r < s.C; r++) {
        LsnCbDev cb{};
        const int K = r < s.Cm ? s.Km : s.Kp, F = r == 0 ? s.F : 0;
        int E = (r <= s.C - gamma - 1) ? NL * Qm * (Gp / s.C) : NL * Qm * ((Gp + s.C - 1) / s.C);
        if (rp + E > G) E = G - rp;
        cb.e_off = d.llr_off[tb.cw_idx & 1] + (uint32_t)rp; cb.E = (uint32_t)E; cb.K = (uint32_t)K; cb.F = (uint32_t)F; cb.rv = (uint32_t)tb.rv;
        cb.crc_b = s.C > 1 ? 1u : 0u;
        cb.out_bytes = (uint32_t)(K - F - (s.C > 1 ? 24 : 0)) / 8;
        cb.out_off = (uint32_t)(pay_n - pay0) + wp;
        qpp_params(K, cb.f1, cb.f2);
        cb.max_iter = (uint32_t)cfg.max_turbo_iterations;
        wp += cb.out_bytes;
        rp += E;
        h_cbs.push_back(cb);
      }
      j.cb_count[i] = (uint32_t)s.C;
      pay_n += (wp + 15) & ~15u;
      tbrefs.push_back(ref);
      perf.nof_tb_decodes++;
      perf.nof_cb_decodes += (uint64_t)s.C;
      const uint64_t tbbytes = 2ull * (uint64_t)tb.nof_bits * 2ull + (uint64_t)tb.tbs / 8ull;
      perf.algo_bytes += tbbytes;
      perf.turbo_algo_bytes += (uint64_t)tb.nof_bits * 2ull + (uint64_t)tb.tbs / 8ull;
    }
    h_jobs.push_back(d);
  }
  const uint32_t njobs = (uint32_t)h_jobs.size(), ncb = (uint32_t)h_cbs.size();
  if (njobs == 0) return;
  {
    size_t cap;
    cap = jobs_cap; grow(d_jobs, cap, njobs); jobs_cap = (uint32_t)cap;
    if (ncb > cbs_cap) {
      size_t c1 = cbs_cap, c2 = cbs_cap;
      grow(d_cbs, c1, ncb); grow(d_cbres, c2, ncb); cbs_cap = (uint32_t)std::min(c1, c2);
    }
    grow(d_prefix, prefix_cap, prefix_n);
    grow(d_llr16, llr16_cap, llr_n + 8);
    grow(d_payload, payload_cap, pay_n - pay0 + 16);
  }
  if (h_cbres_cap < ncb) {
    if (h_cbres_pinned) HIP_CHECK(hipHostFree(h_cbres_pinned));
    h_cbres_cap = ncb + ncb / 2 + 256;
    HIP_CHECK(hipHostMalloc((void**)&h_cbres_pinned, h_cbres_cap * sizeof(LsnCbRes)));
  }
  if (h_payload_cap < pay_n - pay0) {
    if (h_payload_pinned) HIP_CHECK(hipHostFree(h_payload_pinned));
    h_payload_cap = (pay_n - pay0) * 2 + 4096;
    HIP_CHECK(hipHostMalloc((void**)&h_payload_pinned, h_payload_cap));
  }
  HIP_CHECK(hipMemcpyAsync(d_jobs, h_jobs.data(), njobs * sizeof(LsnGrantDev), hipMemcpyHostToDevice, st));
  if (ncb) HIP_CHECK(hipMemcpyAsync(d_cbs, h_cbs.data(), ncb * sizeof(LsnCbDev), hipMemcpyHostToDevice, st));
  HIP_CHECK(hipMemsetAsync(d_llr16, 0, llr_n * sizeof(int16_t), st));
  auto timed = [&](int k, auto&& fn) {
    HIP_CHECK(hipEventRecord(ev[2 * k], st));
    fn();
    HIP_CHECK(hipEventRecord(ev[2 * k + 1], st));
    perf.kernel_launches[k]++;
  };
  timed(LSN_K_PDSCH_PREP, [&] { lsn_launch_pdsch_prep(cd, d_jobs, d_prefix, njobs, st); });
  timed(LSN_K_PDSCH_DEMOD, [&] { lsn_launch_pdsch_demod(cd, d_jobs, d_prefix, d_grid, d_ce, d_chest, d_llr16, njobs, st); });
  if (ncb) {
    timed(LSN_K_TURBO, [&] { lsn_launch_turbo(cd, d_cbs, d_llr16, d_payload, d_cbres, ncb, st); });
    HIP_CHECK(hipMemcpyAsync(h_cbres_pinned, d_cbres, ncb * sizeof(LsnCbRes), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(h_payload_pinned, d_payload, pay_n - pay0, hipMemcpyDeviceToHost, st));
  }
  HIP_CHECK(hipStreamSynchronize(st));
  for (int k : {LSN_K_PDSCH_PREP, LSN_K_PDSCH_DEMOD, LSN_K_TURBO}) {
    if (k == LSN_K_TURBO && !ncb) continue;
    float ms = 0;
    if (hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]) == hipSuccess) perf.kernel_ms[k] += ms;
  }
  h_payload.resize(pay_n);
  if (pay_n > pay0) std::memcpy(h_payload.data() + pay0, h_payload_pinned, pay_n - pay0);
  // transport-block verdicts: every code block ok, CRC24A over data||parity zero (combined from the per-block
  // remainders), parity word non-zero
  for (auto& r : tbrefs) {
    DecodeJob& j = jobs[r.job];
    bool all_ok = true;
    uint32_t rem = 0;
    uint64_t bits_after = 0;
    for (int q = (int)r.cb_count - 1; q >= 0; q--) {
      const LsnCbRes& cr = h_cbres_pinned[r.cb_first + q];
      all_ok = all_ok && cr.ok != 0;
      j.iters += cr.iters;
      perf.nof_turbo_iterations += cr.iters;
      rem ^= crc24a_mulmod(cr.rem_a, crc24a_xpow(bits_after));
      bits_after += 8ull * h_cbs[r.cb_first + q].out_bytes;
    }
    const int tbs = j.grant.tb[r.tb].tbs;
    const uint8_t* pl = h_payload.data() + j.payload_off[r.tb];
    const uint32_t par = ((uint32_t)pl[tbs / 8] << 16) | ((uint32_t)pl[tbs / 8 + 1] << 8) | pl[tbs / 8 + 2];
    j.crc[r.tb] = all_ok && rem == 0 && par != 0 && bits_after == (uint64_t)tbs + 24;
  }
  for (int jid : todo) jobs[jid].done = true;
}

void Engine::ensureJob(int j, hipStream_t st)
{
  if (j < 0 || jobs[j].done) return;
  std::vector<int> one{j};
  jobs[j].planned = false;
  runJobs(one, st);
}

// wave 1: the first decode the reference would attempt for every accepted DL DCI, predicted from the MCS-tracking
// state as of now; wave 2: the 256QAM-table retry of "unknown table" grants whose first attempt failed on both TBs
void Engine::planJobs(uint32_t nsf)
{
  std::vector<int> wave;
  struct Pending { uint32_t sf; size_t di; };
  std::vector<Pending> retry;
  for (uint32_t sf = 0; sf < nsf; sf++) {
    SubframeCtx& c = ctx[sf];
    if (!c.searched) continue;
    for (size_t di = 0; di < c.dl.size(); di++) {
      DlEntry& e = c.dl[di];
      if (!e.unpack_ok) continue;
      McsTable table;
      if (cfg.mcs_tracking_mode == 1)
        table = (e.rnti == SIRNTI || e.rnti == PRNTI || rnti_israr(e.rnti) || e.format == FORMAT1A) ? TABLE_64QAM : mcs_tracking.peek(e.rnti);
      else
        table = cfg.mcs_tracking_mode == 2 ? TABLE_UNKNOWN : TABLE_64QAM;
      const int first = table == TABLE_256QAM ? 1 : 0;
      const PdschGrant& g = first ? e.grant256 : e.grant64;
      const bool ok = first ? e.ok256 : e.ok64;
      if (!ok || !(g.tb[0].tbs > 0)) continue;
      if (cfg.nof_rx_antennas == 1 && (e.grant64.nof_tb == 2 || e.grant256.nof_tb == 2)) continue;
      if (e.job[first] < 0) e.job[first] = newJob(sf, e, first);
      if (e.job[first] >= 0) wave.push_back(e.job[first]);
      if (table >= TABLE_UNKNOWN && e.ok256) retry.push_back({sf, di});
    }
  }
  runJobs(wave, cur_stream);
  wave.clear();
  for (auto& p : retry) {
    DlEntry& e = ctx[p.sf].dl[p.di];
    if (e.job[0] < 0 || !jobs[e.job[0]].done) continue;
    if (jobs[e.job[0]].crc[0] || jobs[e.job[0]].crc[1]) continue;
    if (e.job[1] < 0) e.job[1] = newJob(p.sf, e, 1);
    if (e.job[1] >= 0) wave.push_back(e.job[1]);
  }
  runJobs(wave, cur_stream);
}

// ------------------------------------------------------------------------------------------------ commit
void Engine::emitPdu(const char* name, const uint8_t* pdu, uint32_t len, uint16_t rnti, uint32_t tti, uint8_t tb)
{
  perf.nof_pdus++;
  if (!sink) return;
  lsn_pdu_ctx_t c{};
  c.tti = tti; c.direction = 1; c.crc_ok = 1; c.is_retx = 0; c.tb = tb;
  // LTESniffer_pcap_writer::write_dl_* (PcapWriter.cc:162-190)
  if (name[0] == 'S') { c.rnti = SIRNTI; c.rnti_type = 4; }
  else if (name[0] == 'P') { c.rnti = PRNTI; c.rnti_type = 1; }
  else if (name[0] == 'R') { c.rnti = rnti; c.rnti_type = 2; }
  else { c.rnti = rnti; c.rnti_type = 3; }
  sink(sink_user, &c, pdu, len);
}

// PDSCH_Decoder::decode_dl_mode (DL_Sniffer_PDSCH.cc:881-1291) over the decode results of this subframe
void Engine::commitSubframe(uint32_t sf, hipStream_t st)
{
  SubframeCtx& c = ctx[sf];
  if (!c.searched) return;
  // DCICollection.cc:107-134: the table of every DCI of this subframe is fixed before any of them is decoded
  std::vector<McsTable> tables(c.dl.size());
  for (size_t di = 0; di < c.dl.size(); di++) {
    const DlEntry& e = c.dl[di];
    if (cfg.mcs_tracking_mode == 1)
      tables[di] = (e.rnti == SIRNTI || e.rnti == PRNTI || rnti_israr(e.rnti) || e.format == FORMAT1A) ? TABLE_64QAM
                                                                                                      : mcs_tracking.find_tracking_info_RNTI_dl(e.rnti);
    else
      tables[di] = cfg.mcs_tracking_mode == 2 ? TABLE_UNKNOWN : TABLE_64QAM;
  }
  for (size_t di = 0; di < c.dl.size(); di++) {
    DlEntry& e = c.dl[di];
    const McsTable table = tables[di];
    const bool has64 = e.unpack_ok && (table == TABLE_64QAM || table >= TABLE_UNKNOWN);
    const bool has256 = e.unpack_ok && (table == TABLE_256QAM || table >= TABLE_UNKNOWN);
    const bool dci_rnti_ok = e.rnti > 0 && !(has64 && !e.ok64) && !(has256 && !e.ok256);  // falcon_dci.c:286,293,300,305
    const int cur_t = table == TABLE_256QAM ? 1 : 0;
    static const PdschGrant empty_grant;
    const PdschGrant& cur = cur_t ? (has256 ? e.grant256 : empty_grant) : (has64 ? e.grant64 : empty_grant);
    const bool two_tb = (has64 && e.grant64.nof_tb == 2) || (has256 && e.grant256.nof_tb == 2);
    const bool gate = (cur.tb[0].tbs > 0 && dci_rnti_ok && !(cfg.nof_rx_antennas == 1 && two_tb)) || e.rnti == PRNTI;  // :887-889
    if (!gate) continue;
    const char* name = rnti_name(e.rnti);
    auto run = [&](int t) -> int {
      if (!(t ? has256 : has64)) return -1;
      if (e.job[t] < 0) e.job[t] = newJob(sf, e, t);
      if (e.job[t] >= 0 && !jobs[e.job[t]].done) { ensureJob(e.job[t], st); perf.nof_ondemand_decodes++; }
      return e.job[t];
    };
    bool crc[2] = {false, false};
    if (table == TABLE_64QAM || table == TABLE_256QAM) {  // :932-1083
      const int j = run(cur_t);
      if (j >= 0)
        for (int tb = 0; tb < 2; tb++) {
          const int len = jobs[j].grant.tb[tb].tbs / 8;
          if (jobs[j].crc[tb] && len > 0) {
            emitPdu(name, h_payload.data() + jobs[j].payload_off[tb], (uint32_t)len, e.rnti, c.tti, (uint8_t)tb);
            if (name[0] == 'R') unpackRar(h_payload.data() + jobs[j].payload_off[tb], len, false);
          }
        }
    } else {  // unknown table: 64QAM table first, the 256QAM table only if both TBs failed, :1089-1243
      const int j = run(0);
      if (j >= 0) {
        for (int tb = 0; tb < 2; tb++) {
          const int len = jobs[j].grant.tb[tb].tbs / 8;
          crc[tb] = jobs[j].crc[tb];
          if (crc[tb] && len > 0) {
            emitPdu(name, h_payload.data() + jobs[j].payload_off[tb], (uint32_t)len, e.rnti, c.tti, (uint8_t)tb);
            if (name[0] == 'R') unpackRar(h_payload.data() + jobs[j].payload_off[tb], len, false);
            if (e.dci.tb[tb].mcs_idx > 0 && e.dci.tb[tb].mcs_idx < 29 && e.format > FORMAT1A) mcs_tracking.update_RNTI_dl(e.rnti, TABLE_64QAM);
          }
        }
        if (!crc[0] && !crc[1]) {
          const int j2 = run(1);
          if (j2 >= 0)
            for (int tb = 0; tb < 2; tb++) {
              const int len = jobs[j2].grant.tb[tb].tbs / 8;
              if (jobs[j2].crc[tb] && len > 0) {
                emitPdu(name, h_payload.data() + jobs[j2].payload_off[tb], (uint32_t)len, e.rnti, c.tti, (uint8_t)tb);
                if (e.dci.tb[tb].mcs_idx > 0 && e.dci.tb[tb].mcs_idx < 28 && e.format > FORMAT1A) mcs_tracking.update_RNTI_dl(e.rnti, TABLE_256QAM);
              }
            }
        }
      }
    }
    if (name[0] == 'C' && cfg.mcs_tracking_mode) mcs_tracking.update_statistic_dl(e.rnti, e.format);  // :1268-1285
  }
}

// ------------------------------------------------------------------------------------------------ batch driver
int Engine::process(const void* d_iq, uint32_t nsf_total, uint32_t start_tti, uint32_t update_meta_period, hipStream_t stream)
{
  if (!cell_set) return LSN_ERROR;
  if (!d_iq && nsf_total) return LSN_ERROR_INVALID_INPUTS;
  try {
    HIP_CHECK(hipSetDevice(cfg.device));
    perf = lsn_perf_t{};
    cur_stream = stream;
    const double t_all = now_ms();
    const size_t sf_stride = (size_t)cfg.nof_rx_antennas * cd.sflen * sizeof(cf32);
    for (uint32_t base = 0; base < nsf_total; base += max_batch) {
      const uint32_t nsf = std::min(max_batch, nsf_total - base);
      jobs.clear(); h_payload.clear();
      for (uint32_t i = 0; i < nsf; i++) {
        SubframeCtx& c = ctx[i];
        c = SubframeCtx();
        c.tti = start_tti + base + i;
        c.sf_idx = c.tti % 10; c.sfn = (c.tti / 10) % 1024;
      }
      double t0 = now_ms();
      stageA((const uint8_t*)d_iq + (size_t)base * sf_stride, nsf, stream);
      double t1 = now_ms();
      perf.ms_stage_a += t1 - t0;
      for (uint32_t i = 0; i < nsf; i++) {
        const bool upd = (update_meta_period && (sf_cnt % update_meta_period) == 0) || force_meta_next;  // LTESniffer_Core.cc:434
        force_meta_next = false;
        sf_cnt++;
        searchSubframe(i, upd);
      }
      double t2 = now_ms();
      perf.ms_search += t2 - t1;
      planJobs(nsf);
      double t3 = now_ms();
      perf.ms_stage_c += t3 - t2;
      for (uint32_t i = 0; i < nsf; i++) commitSubframe(i, stream);
      perf.ms_commit += now_ms() - t3;
      last_nsf = nsf;
    }
    perf.ms_total = now_ms() - t_all;
    return LSN_SUCCESS;
  } catch (const std::exception& ex) {
    fprintf(stderr, "ltesniffer_amd: %s\n", ex.what());
    return LSN_ERROR;
  }
}

int Engine::processHost(const float* iq, uint32_t nsf_total, uint32_t start_tti, uint32_t update_meta_period)
{
  if (!cell_set) return LSN_ERROR;
  try {
    HIP_CHECK(hipSetDevice(cfg.device));
    const size_t sf_stride = (size_t)cfg.nof_rx_antennas * cd.sflen * sizeof(cf32);
    lsn_perf_t acc{};
    for (uint32_t base = 0; base < nsf_total; base += max_batch) {
      const uint32_t nsf = std::min(max_batch, nsf_total - base);
      HIP_CHECK(hipMemcpy(d_iq_staging, (const uint8_t*)iq + (size_t)base * sf_stride, (size_t)nsf * sf_stride, hipMemcpyHostToDevice));
      const int r = process(d_iq_staging, nsf, start_tti + base, update_meta_period, nullptr);
      if (r != LSN_SUCCESS) return r;
      // accumulate the per-batch perf records so that the caller sees the whole call
      acc.ms_stage_a += perf.ms_stage_a; acc.ms_search += perf.ms_search; acc.ms_stage_c += perf.ms_stage_c; acc.ms_commit += perf.ms_commit;
      acc.ms_total += perf.ms_total; acc.algo_bytes += perf.algo_bytes; acc.turbo_algo_bytes += perf.turbo_algo_bytes;
      acc.nof_tb_decodes += perf.nof_tb_decodes; acc.nof_cb_decodes += perf.nof_cb_decodes; acc.nof_turbo_iterations += perf.nof_turbo_iterations;
      acc.nof_candidates_decoded += perf.nof_candidates_decoded; acc.nof_ondemand_decodes += perf.nof_ondemand_decodes; acc.nof_pdus += perf.nof_pdus;
      for (int k = 0; k < 16; k++) { acc.kernel_ms[k] += perf.kernel_ms[k]; acc.kernel_launches[k] += perf.kernel_launches[k]; }
    }
    perf = acc;
    return LSN_SUCCESS;
  } catch (const std::exception& ex) {
    fprintf(stderr, "ltesniffer_amd: %s\n", ex.what());
    return LSN_ERROR;
  }
}

// ------------------------------------------------------------------------------------------------ parity taps
long Engine::tap(int what, uint32_t sf, void* out, size_t cap)
{
  if (!cell_set || sf >= last_nsf) return LSN_ERROR_INVALID_INPUTS;
  const size_t A = cfg.nof_rx_antennas, P = cell.nof_ports, nre = cd.nre;
  auto d2h = [&](const void* src, size_t n) -> long {
    if (n > cap) return LSN_ERROR_INVALID_INPUTS;
    if (hipMemcpy(out, src, n, hipMemcpyDeviceToHost) != hipSuccess) return LSN_ERROR;
    return (long)n;
  };
  auto h2h = [&](const void* src, size_t n) -> long {
    if (n > cap) return LSN_ERROR_INVALID_INPUTS;
    std::memcpy(out, src, n);
    return (long)n;
  };
  switch (what) {
    case LSN_TAP_GRID: return d2h(d_grid + (size_t)sf * A * 14 * nre, A * 14 * nre * sizeof(cf32));
    case LSN_TAP_CE: return d2h(d_ce + (size_t)sf * P * A * 14 * nre, P * A * 14 * nre * sizeof(cf32));
    case LSN_TAP_PDCCH_LLR: return d2h(d_llr + (size_t)sf * LSN_LLR_STRIDE, (size_t)cd.nof_cce[ctx[sf].cfi - 1] * 72 * sizeof(float));
    case LSN_TAP_CHEST: {
      // layout of the test-side record: noise[2][2], rsrp[2][2], cepow[2][2], corr_r, corr_i, noise_avg, rsrp_avg, snr_db, cfo_hz, chan_ref
      float r[19] = {0};
      const LsnChest& h = h_chest[sf];
      for (size_t rx = 0; rx < A; rx++)
        for (size_t p = 0; p < P; p++) {
          r[rx * 2 + p] = h.noise[rx * P + p]; r[4 + rx * 2 + p] = h.rsrp[rx * P + p]; r[8 + rx * 2 + p] = h.cepow[rx * P + p];
        }
      r[12] = h.corr_r; r[13] = h.corr_i; r[14] = h.noise_avg; r[15] = h.rsrp_avg; r[16] = ctx[sf].snr_db; r[17] = ctx[sf].cfo_hz; r[18] = h.chan_ref;
      return h2h(r, sizeof(r));
    }
    case LSN_TAP_CFI: return h2h(&ctx[sf].cfi, sizeof(uint32_t));
    case LSN_TAP_CANDIDATES: return h2h(h_cand + (size_t)sf * LSN_MAX_LOC * LSN_MAX_SIZES, (size_t)LSN_MAX_LOC * LSN_MAX_SIZES * sizeof(LsnCand));
    case LSN_TAP_CCE_POWER: return h2h(h_ccepow + (size_t)sf * LSN_CCE_STRIDE, LSN_CCE_STRIDE * sizeof(float));
    case LSN_TAP_ACCEPTED: return h2h(ctx[sf].accepted.data(), ctx[sf].accepted.size() * sizeof(uint32_t));
    case LSN_TAP_RB_POWER: {
      // SubframePower.cc:34-41: dB conversion on the host
      float r[110];
      const float logdiv = 10.0f * log10f(14.0f);
      for (uint32_t i = 0; i < cell.nof_prb; i++) r[i] = 10.0f * log10f(h_rbp[sf * 128 + i]) - logdiv;
      return h2h(r, cell.nof_prb * sizeof(float));
    }
    default: return LSN_ERROR_INVALID_INPUTS;
  }
}

}  // namespace lsn
