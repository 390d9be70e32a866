// lsn_search.cc - see lsn_search.h.  Product code (HIP-free): must not include anything from oracle/.
#include "lsn_search.h"
#include <new>
#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace lsn {

const char* rnti_name(uint16_t r)
{
  if (r == SIRNTI) return "SI_RNTI";
  if (r == PRNTI) return "P_RNTI";
  if (r > RARNTI_START && r < RARNTI_END) return "RA_RNTI";
  return "C_RNTI";
}

// ------------------------------------------------------------------------------------------------ search space
void SearchSpace::init(const uint32_t n[3])
{
  std::memset(common, 0, sizeof(common));
  for (int c = 0; c < 3; c++) {
    nof_cce[c] = n[c];
    // srsran_pdcch_common_locations_ncce [srsRAN]: L = 8, 4 inside the first 16 CCEs
    for (int l = 3; l > 1; l--) {
      const uint32_t L = 1u << l;
      if (n[c] < L) continue;
      const uint32_t lim = std::min(n[c], 16u) / L;
      for (uint32_t i = 0; i < lim; i++) {
        const uint32_t ncce = L * (i % (n[c] / L));
        if (ncce + L <= n[c] && ncce < LSN_CCE_STRIDE) common[c][l][ncce] = 1;
      }
    }
  }
}

uint32_t SearchSpace::validate(uint32_t cfi, uint32_t ncce, uint32_t l, uint32_t nsubframe, uint16_t rnti) const
{
  static const uint32_t ncand[4] = {6, 6, 2, 2};
  const uint32_t n = nof_cce[cfi - 1];
  const uint8_t(*cm)[LSN_CCE_STRIDE] = common[cfi - 1];
  bool ue = false;
  if (rnti_isuser(rnti)) ue = true;
  else if (!(rnti_israr(rnti) || rnti >= MRNTI)) return 0;  // reserved interval or RNTI 0: no locations at all
  auto member = [&](uint32_t lv, uint32_t Yk) -> bool {
    if (cm[lv][ncce]) return true;
    if (!ue) return false;
    const uint32_t L = 1u << lv;
    if (n < L || (ncce & (L - 1))) return false;
    const uint32_t M = n / L, q = ncce / L;
    if (q >= M) return false;
    return (q + M - Yk % M) % M < ncand[lv];
  };
  uint32_t Yk = rnti;
  if (ue)
    for (uint32_t m = 0; m <= nsubframe; m++) {  // Yk = (39827 * Yk) mod 65537 with 65537 = 2^16 + 1: x mod = lo16 - hi16 (+ 65537)
      const uint32_t x = 39827u * Yk;
      const int32_t t = (int32_t)(x & 0xFFFFu) - (int32_t)(x >> 16);
      Yk = (uint32_t)(t < 0 ? t + 65537 : t);
    }
  if (!member(l, Yk)) return 0;
  if (l > 0 && member(l - 1, Yk)) return 1;
  return 2;
}

// ------------------------------------------------------------------------------------------------ FalconSearch
FalconSearch::FalconSearch(uint32_t threshold, double split_ratio, bool skip_secondary)
{
  rnti_manager.reset(new RNTIManager(NOF_FORMATS, 304 / 5, threshold));  // PhyCommon.cc:11
  meta_formats.reset(new DCIMetaFormats(NOF_FORMATS, split_ratio));
  meta_formats->setSkipSecondaryMetaFormats(skip_secondary);
}

void FalconSearch::setupDefaultIntervals()
{
  rnti_manager->addEvergreen(RARNTI_START, RARNTI_END, FORMAT1A);
  rnti_manager->addEvergreen(PRNTI, SIRNTI, FORMAT1A);
  rnti_manager->addEvergreen(RARNTI_START, RARNTI_END, FORMAT1C);
  rnti_manager->addEvergreen(PRNTI, SIRNTI, FORMAT1C);
  for (uint32_t f = 0; f < NOF_FORMATS; f++) rnti_manager->addForbidden(0, 0, f);
}

void FalconSearch::setCell(const Cell& c, const uint32_t n[3])
{
  cell = c;
  cell_build_re_tables(cell);
  for (int i = 0; i < 3; i++) nof_cce[i] = n[i];
  sspace.init(n);
  std::vector<uint32_t> sizes;
  for (int f = 0; f < NOF_FORMATS; f++) {
    size_of_format[f] = dci_format_sizeof(cell, (DciFormat)f);
    if (std::find(sizes.begin(), sizes.end(), size_of_format[f]) == sizes.end()) sizes.push_back(size_of_format[f]);
  }
  std::sort(sizes.begin(), sizes.end());
  if (sizes.size() > LSN_MAX_SIZES || sizes.back() > 64) throw std::runtime_error("unsupported DCI size set");
  nsizes = (uint32_t)sizes.size();
  for (size_t i = 0; i < sizes.size(); i++) size_list[i] = sizes[i];
  for (int f = 0; f < NOF_FORMATS; f++) size_index_of_format[f] = (int)(std::find(sizes.begin(), sizes.end(), size_of_format[f]) - sizes.begin());
  for (int cfi = 0; cfi < 3; cfi++) {
    LocTemplate& tp = loc_template[cfi];
    const uint32_t ncce = nof_cce[cfi], lim = std::min<uint32_t>(ncce, LSN_MAX_NUM_OF_CCE);
    for (auto& row : tp.map) for (auto& v : row) v = -1;
    for (auto& cs : tp.cover) cs.clear();
    uint32_t k = 0;
    for (int l = 3; l >= 0; l--) {
      const uint32_t L = 1u << l;
      if (ncce < L) continue;
      for (uint32_t i = 0; i < lim / L; i++)
        if (k < LSN_MAX_LOC) {
          tp.locations[k] = FalconLocation{(uint32_t)l, L * (i % (ncce / L))};
          for (uint32_t m = tp.locations[k].ncce; m < tp.locations[k].ncce + L && m < LSN_MAX_NUM_OF_CCE; m++) tp.map[m][l] = (int16_t)k;
          k++;
        }
    }
    tp.nloc = k;
    for (uint32_t m = 0; m < LSN_MAX_NUM_OF_CCE; m++)  // (a later location of the same level overwrites an earlier one in the map, like the reference's pointer map)
      for (int a = 0; a < 4; a++)
        if (tp.map[m][a] >= 0) tp.cover[m].set((uint32_t)tp.map[m][a]);
    for (auto& cm : tp.ccemask) cm[0] = cm[1] = 0;
    for (uint32_t m = 0; m < LSN_MAX_NUM_OF_CCE; m++)
      for (uint32_t q = 0; q < LSN_MAX_LOC; q++)
        if (tp.cover[m].test(q)) tp.ccemask[q][m >> 6] |= 1ull << (m & 63);
  }
}

// DCICollection::addCandidate (DCICollection.cc:97-298) + srsran_dci_msg_to_trace_timestamp (falcon_dci.c:148-352).
// The sequential search only RECORDS the accepted DCI (nothing it decides later depends on the unpacked fields); unpacking, the grant
// conversion for both MCS tables and the PRB collision statistics happen in finishSubframe, on the decode threads.  Which table's grant
// "exists" for the reference is resolved at commit time, when the MCS-tracking state of the subframe is known.
void FalconSearch::addCandidate(SubframeCtx& c, const DciCandidate& cand, uint32_t L, uint32_t ncce, uint32_t histval)
{
  c.raw.push_back(AcceptedDci{cand.rnti, (uint8_t)cand.msg.format, (uint8_t)L, (uint16_t)ncce, (uint16_t)cand.msg.nof_bits, histval, (unsigned long long)cand.slot});   // (bits: the slot for now - FalconSearch::search puts the payload in)
}

void FalconSearch::materialize(SubframeCtx& c)
{
  if (c.materialized) return;
  c.materialized = true;
  for (const AcceptedDci& r : c.raw) {
    if (c.accepted.size() < 64 * 6) {
      const uint32_t a[6] = {r.rnti, (uint32_t)r.format, r.L, r.ncce, r.nof_bits, r.histval};
      c.accepted.insert(c.accepted.end(), a, a + 6);
    }
    if ((DciFormat)r.format == FORMAT0) {
      if (c.ul.size() >= 64) continue;
      c.ul.emplace_back();
      UlEntry& u = c.ul.back();
      u.rnti = r.rnti; u.nof_bits = r.nof_bits; u.L = r.L; u.ncce = r.ncce; u.histval = r.histval; u.bits = r.bits;
      continue;
    }
    if (c.dl.size() >= 64) continue;
    c.dl.emplace_back();
    DlEntry& e = c.dl.back();
    e.rnti = r.rnti; e.format = (DciFormat)r.format; e.nof_bits = r.nof_bits; e.L = r.L; e.ncce = r.ncce; e.histval = r.histval; e.bits = r.bits;
  }
}

void FalconSearch::finishUlEntry(UlEntry& u) const
{
  if (u.finished) return;
  u.finished = true;
  DciMsg msg;
  msg.bits = u.bits; msg.nof_bits = u.nof_bits;
  uint8_t payload[64] = {0};
  msg.unpack(payload);
  u.dci.L = u.L; u.dci.ncce = u.ncce;
  u.ok = payload[0] == 0 && dci_msg_unpack_pusch(cell, payload, u.nof_bits, u.rnti, u.dci) && ra_ul_dci_to_grant(cell, u.dci, u.grant);
  if (u.ok && !ra_ul_dci_to_grant_256(cell, u.dci, u.grant256)) { u.ok = false; u.grant256 = PuschGrant(); }  // falcon_dci.c:222-231
}

void FalconSearch::finishDlEntry(DlEntry& e, uint32_t sf_idx, uint32_t cfi) const
{
  if (e.finished) return;
  e.finished = true;
  if (!e.unpacked) {
    e.unpacked = true;
    DciMsg msg;
    msg.bits = e.bits; msg.nof_bits = e.nof_bits;
    uint8_t payload[64] = {0};
    msg.unpack(payload);
    e.dci.L = e.L; e.dci.ncce = e.ncce;
    e.unpack_ok = dci_msg_unpack_pdsch(cell, payload, e.nof_bits, e.format, e.rnti, e.dci);
    if (!e.unpack_ok) return;
    e.alloc_ok = ra_dl_grant_to_grant_prb_allocation(cell, e.dci, e.grant64);
  }
  if (!e.unpack_ok) return;
  if (!e.alloc_ok) { e.grant256 = e.grant64; return; }
  dl_sniffer_grant_finish_both(cell, sf_idx, cfi, e.dci, e.grant64, e.ok64, e.grant256, e.ok256);
  for (int i = 0; i < 2; i++) {  // DCICollection.cc:252-259
    if (e.grant64.tb[i].nof_bits <= 0) e.grant64.tb[i].enabled = false;
    if (e.grant256.tb[i].nof_bits <= 0) e.grant256.tb[i].enabled = false;
  }
}

void FalconSearch::finishSubframe(SubframeCtx& c)
{
  if (c.finished) return;
  c.finished = true;
  materialize(c);
  uint16_t rb_dl[110] = {0}, rb_ul[110] = {0};
  bool dl_collision = false, ul_collision = false;
  for (auto& e : c.dl) {
    finishDlEntry(e, c.sf_idx, c.cfi);
    if (e.unpack_ok && e.alloc_ok)
      for (uint32_t rb = e.grant64.prb_lo; rb <= e.grant64.prb_hi && rb < cell.nof_prb; rb++)  // DCICollection.cc:215-223 (the PRB set does not depend on the MCS table)
        if (e.grant64.prb_idx[0][rb]) {
          if (rb_dl[rb] != 0) dl_collision = true;
          rb_dl[rb] = e.rnti;
        }
  }
  for (auto& u : c.ul) {
    finishUlEntry(u);
    if (u.ok)  // convert_ul_grant runs only after both conversions succeeded (falcon_dci.c:222-232); DCICollection.cc:275-280
      for (uint32_t i = 0; i < u.grant.L_prb && u.grant.n_prb + i < 110; i++) {
        if (rb_ul[u.grant.n_prb + i] != 0) ul_collision = true;
        rb_ul[u.grant.n_prb + i] = u.rnti;
      }
  }
  if (dl_collision) coll_dw.fetch_add(1, std::memory_order_relaxed);
  if (ul_collision) coll_up.fetch_add(1, std::memory_order_relaxed);
}

bool FalconSearch::buildDlEntry(const SubframeCtx& c, uint16_t rnti, DciFormat fmt, unsigned long long bits, DlEntry& e) const
{
  DciMsg msg;
  msg.bits = bits; msg.nof_bits = size_of_format[fmt]; msg.format = fmt;
  uint8_t payload[64] = {0};
  msg.unpack(payload);
  e = DlEntry();
  e.rnti = rnti; e.format = fmt; e.nof_bits = msg.nof_bits; e.bits = bits;
  e.unpack_ok = dci_msg_unpack_pdsch(cell, payload, msg.nof_bits, fmt, rnti, e.dci);
  e.finished = true; e.unpacked = true;
  if (!e.unpack_ok) return false;
  dl_sniffer_ra_dl_dci_to_grant_both(cell, c.sf_idx, c.cfi, e.dci, e.grant64, e.ok64, e.grant256, e.ok256);
  for (int i = 0; i < 2; i++) {
    if (e.grant64.tb[i].nof_bits <= 0) e.grant64.tb[i].enabled = false;
    if (e.grant256.tb[i].nof_bits <= 0) e.grant256.tb[i].enabled = false;
  }
  return true;
}

// DCISearch::inspect_dci_location_recursively, DCISearch.cc:102-447
int FalconSearch::inspect_dci_location_recursively(SubframeCtx& c, const int16_t (*cce_map)[4], uint32_t ncce, uint32_t L, uint32_t max_depth, MetaFormat** metas,
                                                   uint32_t nof_formats, uint32_t enable_discovery, const DciCandidate* parent_cand)
{
  int hist_max_format_idx = -1;
  uint32_t hist_max_format_value = 0, nof_cand_above_threshold = 0;
  const int li = cce_map[ncce][L];  // the level-L location that covers this CCE
  if (li < 0 || f_occupied.test((uint32_t)li) || f_checked.test((uint32_t)li) || f_nopower.test((uint32_t)li)) return 0;  // :124-127
  // only the first nof_formats entries exist (children index their parent's candidates with the same format list)
  alignas(DciCandidate) unsigned char cand_raw[sizeof(DciCandidate) * NOF_FORMATS];
  DciCandidate* cand = reinterpret_cast<DciCandidate*>(cand_raw);
  stats.nof_decoded_locations += nof_formats;
  nof_lookups += nof_formats;

  for (uint32_t fi = 0; fi < nof_formats; fi++) {
    {  // srsran_pdcch_decode_msg_limit_avg_llr_power (falcon_pdcch.c:110-170) as a lookup in the exhaustive candidate table (decodeCandidate)
      // (the one-word view of the slot, LSN_CAND_HOT: everything the decisions below read; the payload bits stay in the table until a candidate is accepted)
      const DciFormat format = metas[fi]->format;
      const uint32_t slot = (uint32_t)li * LSN_MAX_SIZES + (uint32_t)size_index_of_format[format];
      auto hot = [&]() -> uint32_t { return cur_cand4 ? cur_cand4[slot] : LSN_CAND_HOT(cur_cand[slot].bits, cur_cand[slot].rnti, cur_cand[slot].flags); };
      uint32_t t = hot();
      if ((t & (LSN_CAND_NOT_COMPUTED << 16)) && cand_miss) { cand_miss(cand_miss_ctx, (uint32_t)li, (uint32_t)size_index_of_format[format]); t = hot(); }
      DciCandidate& d = cand[fi];
      d.slot = (uint16_t)slot;
      d.msg.bits = 0;
      if (t & (1u << 16)) {
        d.rnti = (uint16_t)t;
        d.search_space_match_result = (t >> 17) & 3u;
        d.msg.nof_bits = size_of_format[format];
        d.msg.format = (format == FORMAT0 || format == FORMAT1A) ? (((t >> 19) & 1u) == 0 ? FORMAT0 : FORMAT1A) : format;  // falcon_pdcch.c:147-148
        if (d.search_space_match_result) __builtin_prefetch(&cur_cand[slot], 0, 1);   // it may be accepted: its payload is read when the subframe's search ends
      } else {
        d.rnti = 0; d.search_space_match_result = 0; d.msg.nof_bits = 0; d.msg.format = FORMAT0;
      }
    }
    if (cand[fi].msg.format == FORMAT0 && rnti_manager->activatedByRar(cand[fi].rnti)) {  // :139-158 (getActivationReason(rnti) == RM_ACT_RAR)
      bool add = true;
      for (auto& t : temp_dci0)
        if (t.format == cand[fi].msg.format && t.rnti == cand[fi].rnti && t.ncce == ncce) add = false;
      if (add && temp_dci0.size() < 64) temp_dci0.push_back({cand[fi].rnti, L, ncce, cand[fi].msg.format, cand[fi]});
    }
    if (metas[fi]->format != cand[fi].msg.format) { cand[fi].rnti = 0; cand[fi].search_space_match_result = 0; continue; }  // :163
    if (metas[fi]->format == FORMAT1C && cand[fi].rnti > RARNTI_END && cand[fi].rnti < PRNTI) { cand[fi].rnti = 0; cand[fi].search_space_match_result = 0; continue; }  // :174
    if (cand[fi].rnti > RARNTI_START && cand[fi].rnti < RARNTI_END)  // :181-197
      if (metas[fi]->format != FORMAT1A && metas[fi]->format != FORMAT1C) { cand[fi].rnti = 0; cand[fi].search_space_match_result = 0; continue; }
    if (shortcut_discovery && enable_discovery && parent_cand != nullptr && parent_cand[fi].rnti == cand[fi].rnti &&
        !rnti_manager->isForbidden(cand[fi].rnti, metas[fi]->global_index)) {  // :200-211 (shortcut discovery)
      stats.nof_decoded_locations -= nof_formats - fi - 1;  // the formats behind fi are not decoded
      nof_lookups -= nof_formats - fi - 1;
      return -((int)fi + 1);
    }
    // :214 srsran_pdcch_validate_location: the verdict travels with the candidate (decodeCandidate)
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
  f_checked.set((uint32_t)li);  // :282
  int disamb = 0;
  if (nof_cand_above_threshold > 0 && cand[hist_max_format_idx].search_space_match_result == 1) {  // :288-298
    if (L > 0 && max_depth > 0)
      disamb = inspect_dci_location_recursively(c, cce_map, ncce + (1u << (L - 1)), L - 1, max_depth - 1, metas, nof_formats, 0, nullptr);
  } else if (nof_cand_above_threshold == 0) {  // :302-368
    int rr = 0;
    if (L > 0 && max_depth > 0) {
      rr += inspect_dci_location_recursively(c, cce_map, ncce, L - 1, max_depth - 1, metas, nof_formats, enable_discovery, cand);
      if (rr < 0) {
        hist_max_format_idx = -rr - 1;
        hist_max_format_value = rnti_manager->getFrequency(cand[hist_max_format_idx].rnti, metas[hist_max_format_idx]->global_index);
        nof_cand_above_threshold = 1;
        if (cand[hist_max_format_idx].search_space_match_result == 1) {
          const uint32_t md = max_depth < 99 ? max_depth : 99;
          disamb = inspect_dci_location_recursively(c, cce_map, ncce + (1u << (L - 1)), L - 1, md - 1, metas, nof_formats, 0, nullptr);
        }
        rnti_manager->activateAndRefresh(cand[hist_max_format_idx].rnti, metas[hist_max_format_idx]->global_index, RM_ACT_SHORTCUT);
      } else {
        rr += inspect_dci_location_recursively(c, cce_map, ncce + (1u << (L - 1)), L - 1, max_depth - 1, metas, nof_formats, enable_discovery, nullptr);
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
    f_used.set((uint32_t)li);
    for (uint32_t ci = ncce; ci < ncce + (1u << L); ci++) { f_occupied |= cur_tp->cover[ci]; f_checked |= cur_tp->cover[ci]; }
    DciCandidate& best = cand[hist_max_format_idx];
    rnti_manager->addCandidate(best.rnti, metas[hist_max_format_idx]->global_index);
    metas[hist_max_format_idx]->hits++;
    const uint32_t Ld = disamb > 0 ? L - 1 : L;
    if (best.rnti != 0) {
      bool add = true;
      if (best.msg.format == FORMAT0)
        for (auto& t : temp_dci0)
          if (t.format == FORMAT0 && t.rnti == best.rnti && t.ncce == ncce) add = false;
      if (add) addCandidate(c, best, Ld, ncce, hist_max_format_value);
      for (auto& t : temp_dci0)  // :422-432
        addCandidate(c, t.cand, t.L, t.ncce, rnti_manager->getFrequency(t.rnti, (uint32_t)t.format));
      temp_dci0.clear();
    }
    return 1 + disamb;
  }
  return 0;
}

// DCISearch::recursive_blind_dci_search, DCISearch.cc:449-528
void FalconSearch::recursive_blind_dci_search(SubframeCtx& c)
{
  const uint32_t ncce = nof_cce[c.cfi - 1];
  const uint32_t lim = std::min<uint32_t>(ncce, LSN_MAX_NUM_OF_CCE);
  stats.nof_cce += ncce;
  // srsran_pdcch_ue_locations_all_map (falcon_pdcch.c:321-356): the enumeration and the CCE -> covering-location map only depend on
  // the CFI: the map is used straight from the template (indices instead of pointers), the per-subframe flags are copied
  const LocTemplate& tp = loc_template[c.cfi - 1];
  cur_tp = &tp;
  const int16_t (*cce_map)[4] = tp.map;
  const uint32_t nloc = tp.nloc;
  f_used.clear(); f_occupied.clear(); f_checked.clear(); f_nopower.clear();
  stats.nof_locations += nloc;
  // srsran_pdcch_cce_avg_llr_power, falcon_pdcch.c:595-620: CCEs below the power bound switch off every location that covers them
  static_assert(LSN_MAX_NUM_OF_CCE <= 128, "two words of CCE flags");
  uint64_t low[2] = {0, 0};  // bit cc: CCE cc is below the bound
  for (uint32_t cc = 0; cc < lim; cc++) low[cc >> 6] |= (uint64_t)(cur_ccepow[cc] < 0.7f) << (cc & 63);
  for (int k = 0; k < 2; k++)
    for (uint64_t m = low[k]; m; m &= m - 1) f_nopower |= tp.cover[64 * k + __builtin_ctzll(m)];
  // (the entry test of inspect_dci_location_recursively, DCISearch.cc:124-127, is repeated here so that dead locations cost no call:
  // the next location in index order that is neither occupied nor checked nor without power, re-evaluated after every call)
  auto pass = [&](MetaFormat** metas, uint32_t nmetas) {
    for (uint32_t i = 0; i < nloc;) {
      const uint32_t w = i >> 6;
      const uint64_t alive = ~(f_occupied.w[w] | f_checked.w[w] | f_nopower.w[w]) & (~0ull << (i & 63));
      if (!alive) { i = (w + 1) << 6; continue; }
      i = (w << 6) + (uint32_t)__builtin_ctzll(alive);
      if (i >= nloc) break;
      const FalconLocation& l = tp.locations[i];
      inspect_dci_location_recursively(c, cce_map, l.ncce, l.L, 99, metas, nmetas, 1, nullptr);
      i++;
    }
  };
  pass(meta_formats->getPrimaryMetaFormats(), meta_formats->getNofPrimaryMetaFormats());
  if (!meta_formats->skipSecondaryMetaFormats()) {
    f_checked.clear();
    pass(meta_formats->getSecondaryMetaFormats(), meta_formats->getNofSecondaryMetaFormats());
  }
  // falcon_pdcch.c:561-593: CCEs with power that no used location covers.  (f_used.intersects(cover[cc]) for every CCE, turned round: the CCEs of the used
  // locations - a dozen - are collected once)
  uint64_t covered[2] = {0, 0};
  for (int k = 0; k < LOCW; k++)
    for (uint64_t m = f_used.w[k]; m; m &= m - 1) { const uint32_t q = 64u * (uint32_t)k + (uint32_t)__builtin_ctzll(m); covered[0] |= tp.ccemask[q][0]; covered[1] |= tp.ccemask[q][1]; }
  const uint64_t in0 = lim >= 64 ? ~0ull : ((1ull << lim) - 1ull), in1 = lim > 64 ? ((1ull << (lim - 64)) - 1ull) : 0ull;
  const uint32_t missed = (uint32_t)__builtin_popcountll(~low[0] & ~covered[0] & in0) + (uint32_t)__builtin_popcountll(~low[1] & ~covered[1] & in1);
  stats.nof_missed_cce += missed;
  rnti_manager->stepTime();
}

void FalconSearch::search(SubframeCtx& c, const LsnCand* cand, const float* ccepow, bool update_meta, const uint32_t* cand4)
{
  cur_cand = cand; cur_cand4 = cand4; cur_ccepow = ccepow;
  temp_dci0.clear();
  if (update_meta) meta_formats->update_formats();  // SubframeWorker.cc:148-151
  c.searched = c.snr_db > 6.0f;                     // DCISearch.cc:568-574
  const size_t first = c.raw.size();
  if (c.searched) recursive_blind_dci_search(c);
  for (size_t i = first; i < c.raw.size(); i++) c.raw[i].bits = cur_cand[c.raw[i].bits].bits;   // the payloads of the accepted DCIs (their lines were asked for when the candidates showed up)
  stats.nof_subframes++;
}

}  // namespace lsn
