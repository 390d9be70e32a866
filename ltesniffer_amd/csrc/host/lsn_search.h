// lsn_search.h - FALCON blind-DCI decision logic over an exhaustive candidate table (HIP-free host code).
//   DCISearch::search / recursive_blind_dci_search / inspect_dci_location_recursively   /root/reference/src/src/DCISearch.cc:102-578
//   srsran_pdcch_ue_locations_all_map, srsran_pdcch_cce_avg_llr_power (thresholding)     /root/reference/lib/src/phy/falcon_phch/falcon_pdcch.c:321-367,595-620
//   DCICollection::addCandidate + srsran_dci_msg_to_trace_timestamp                      /root/reference/src/src/DCICollection.cc:97-298, falcon_dci.c:148-352
// The GPU decodes EVERY (location, DCI size) pair of a subframe; srsran_pdcch_decode_msg_limit_avg_llr_power
// (falcon_pdcch.c:110-170) becomes a table lookup and the order-dependent decisions below stay sequential on the host.
#pragma once
#include "lsn_lte.h"
#include "lsn_types.h"
#include <functional>
#include <atomic>
#include <memory>
#include <vector>

namespace lsn {

struct DciMsg {
  unsigned long long bits = 0;  // payload bit i at position 63-i
  uint32_t nof_bits = 0;
  DciFormat format = FORMAT0;
  void unpack(uint8_t* payload) const { for (uint32_t i = 0; i < nof_bits; i++) payload[i] = (uint8_t)((bits >> (63 - i)) & 1ull); }
};
struct DciCandidate { uint16_t rnti = 0; uint16_t slot = 0; DciMsg msg; uint32_t search_space_match_result = 0; };   // slot: location * LSN_MAX_SIZES + size index (where the payload bits are; msg.bits is filled in for accepted candidates only)

struct DlEntry {  // DL_Sniffer_DCI_DL (Sniffer_dependency.h:90)
  uint16_t rnti = 0; DciFormat format = FORMAT1; uint32_t nof_bits = 0, L = 0, ncce = 0, histval = 0;
  unsigned long long bits = 0;  // DCI payload as decoded (bit i at position 63-i)
  DciDl dci; bool unpack_ok = false;
  PdschGrant grant64, grant256; bool ok64 = false, ok256 = false;  // both tables computed; selection happens at commit
  bool alloc_ok = false, finished = false, unpacked = false;  // the sequential search only records (rnti, format, location, payload bits); unpacking, PRB allocation
                                                              // and MCS/TBS/RE counts are filled in by finishDlEntry / finishSubframe, off the search thread
  int job[2] = {-1, -1};                                          // decode job index per table
  bool hinted = false;                                            // planned for the 256QAM table alone on the decode threads' own evidence (SharedSeq::hint_pos)
};
struct UlEntry { uint16_t rnti = 0; uint32_t nof_bits = 0, L = 0, ncce = 0, histval = 0; unsigned long long bits = 0; DciUl dci; PuschGrant grant, grant256; bool ok = false, finished = false; };

// what the sequential search records of an accepted DCI; the DlEntry / UlEntry objects are built from it off the search thread
struct AcceptedDci { uint16_t rnti; uint8_t format, L; uint16_t ncce, nof_bits; uint32_t histval; unsigned long long bits; };

struct SubframeCtx {
  uint32_t tti = 0, sf_idx = 0, sfn = 0, cfi = 0;
  float snr_db = 0, cfo_hz = 0;
  bool searched = false;
  bool finished = false;           // finishSubframe has run (grants converted, collision statistics counted)
  bool materialized = false;       // dl / ul / accepted have been built from raw (FalconSearch::materialize)
  std::vector<AcceptedDci> raw;    // accepted DCIs in acceptance order
  std::vector<DlEntry> dl;
  std::vector<UlEntry> ul;
  std::vector<uint32_t> accepted;  // 6 words per accepted DCI: rnti, format, L, ncce, nof_bits, histval
  // the TTI counter wraps with the SFN (10 * 1024 subframes): records carry sfn 0..1023 like PcapWriter.cc:102-103, also when a call crosses the wrap
  void reset(uint32_t tti_) { tti_ %= 10240u; tti = tti_; sf_idx = tti_ % 10; sfn = (tti_ / 10) % 1024; cfi = 0; snr_db = cfo_hz = 0; searched = false; finished = false; materialized = false; raw.clear(); dl.clear(); ul.clear(); accepted.clear(); }
};

struct BlindStats { uint32_t nof_locations = 0, nof_decoded_locations = 0, nof_cce = 0, nof_missed_cce = 0, nof_subframes = 0, nof_subframe_collisions_dw = 0, nof_subframe_collisions_up = 0; };

// closed-form 36.213 9.1.1 search-space membership (srsran_pdcch_validate_location, falcon_pdcch.c:223-250)
class SearchSpace {
public:
  void init(const uint32_t nof_cce_per_cfi[3]);
  // 0 invalid, 1 valid but ambiguous with aggregation level l-1 at the same CCE, 2 valid
  uint32_t validate(uint32_t cfi, uint32_t ncce, uint32_t l, uint32_t nsubframe, uint16_t rnti) const;
private:
  uint32_t nof_cce[3] = {0, 0, 0};
  uint8_t common[3][4][LSN_CCE_STRIDE];
};

class FalconSearch {
public:
  FalconSearch(uint32_t histogram_threshold, double split_ratio, bool skip_secondary);
  void setCell(const Cell& cell, const uint32_t nof_cce_per_cfi[3]);
  void setPuschHopOffset(uint32_t n_rb_ho) { cell.pusch_hop_offset = n_rb_ho; }  // SIB2 pusch-HoppingOffset, once known
  // cand: [LSN_MAX_LOC][LSN_MAX_SIZES] of this subframe, ccepow: [LSN_CCE_STRIDE]; c.cfi / c.snr_db / c.sf_idx must be set
  // cand4: the one-word view of the same slots (LSN_CAND_HOT; what the GPU writes next to the table), nullptr: derived from cand on the fly
  void search(SubframeCtx& c, const LsnCand* cand, const float* ccepow, bool update_meta, const uint32_t* cand4 = nullptr);
  RNTIManager& rntiManager() { return *rnti_manager; }
  DCIMetaFormats& metaFormats() { return *meta_formats; }
  BlindStats getStats() const { BlindStats b = stats; b.nof_subframe_collisions_dw = coll_dw.load(); b.nof_subframe_collisions_up = coll_up.load(); return b; }
  uint32_t sizeOfFormat(int f) const { return size_of_format[f]; }
  int sizeIndexOfFormat(int f) const { return size_index_of_format[f]; }
  uint32_t nofSizes() const { return nsizes; }
  const uint32_t* sizes() const { return size_list; }
  void setupDefaultIntervals();  // LTESniffer_Core.cc:398-417
  void setCandMiss(void (*fn)(void*, uint32_t, uint32_t), void* ctx) { cand_miss = fn; cand_miss_ctx = ctx; }   // who decodes a slot the blind decoder left out
  void setShortcutDiscovery(bool enable) { shortcut_discovery = enable; }  // PhyCommon::setShortcutDiscovery, PhyCommon.cc:69-71
  bool getShortcutDiscovery() const { return shortcut_discovery; }
  // the DL entry addCandidate() would build for this candidate (no state is touched): used to decode RA-RNTI grants ahead
  void finishDlEntry(DlEntry& e, uint32_t sf_idx, uint32_t cfi) const;  // idempotent: the rest of addCandidate's DCI unpack + grant conversion (falcon_dci.c:148-352)
  void finishUlEntry(UlEntry& u) const;
  // everything of DCICollection::addCandidate the decisions of the search do not depend on, for all accepted DCIs of a subframe:
  // unpack + grant conversion + the PRB collision statistics (DCICollection.cc:215-223,275-280).  Thread-safe (decode threads).
  void finishSubframe(SubframeCtx& c);
  // DCICollection::addCandidate's bookkeeping for the DCIs the search accepted: c.raw -> c.accepted / c.dl / c.ul (idempotent)
  static void materialize(SubframeCtx& c);
  bool buildDlEntry(const SubframeCtx& c, uint16_t rnti, DciFormat fmt, unsigned long long bits, DlEntry& e) const;
  uint64_t nof_lookups = 0;

private:
  // falcon_dci_location_t (falcon_dci.h:91-99): the static part (L, ncce) lives in the per-CFI template, the four per-subframe flags
  // (used / occupied / checked / !sufficient_power) are bit sets over the location index
  static constexpr int LOCW = (LSN_MAX_LOC + 63) / 64;
  struct LocSet {
    uint64_t w[LOCW] = {0};
    bool test(uint32_t i) const { return (w[i >> 6] >> (i & 63)) & 1ull; }
    void set(uint32_t i) { w[i >> 6] |= 1ull << (i & 63); }
    void operator|=(const LocSet& o) { for (int k = 0; k < LOCW; k++) w[k] |= o.w[k]; }
    bool intersects(const LocSet& o) const { uint64_t a = 0; for (int k = 0; k < LOCW; k++) a |= w[k] & o.w[k]; return a != 0; }
    void clear() { for (int k = 0; k < LOCW; k++) w[k] = 0; }
  };
  struct FalconLocation { uint32_t L, ncce; };
  struct LocTemplate {
    FalconLocation locations[LSN_MAX_LOC];
    int16_t map[LSN_MAX_NUM_OF_CCE][4];   // CCE -> the location of every aggregation level that covers it (-1: none)
    LocSet cover[LSN_MAX_NUM_OF_CCE];     // the same as a set
    uint64_t ccemask[LSN_MAX_LOC][2];     // ... and turned round: the CCEs whose cover holds the location (bit cc)
    uint32_t nloc = 0;
  };
  struct TempDci0 { uint16_t rnti; uint32_t L, ncce; DciFormat format; DciCandidate cand; };
  int inspect_dci_location_recursively(SubframeCtx& c, const int16_t (*cce_map)[4], uint32_t ncce, uint32_t L, uint32_t max_depth, MetaFormat** meta_formats_,
                                       uint32_t nof_formats, uint32_t enable_discovery, const DciCandidate* parent_cand);
  void recursive_blind_dci_search(SubframeCtx& c);
  void addCandidate(SubframeCtx& c, const DciCandidate& cand, uint32_t L, uint32_t ncce, uint32_t histval);

  Cell cell;
  uint32_t nof_cce[3] = {0, 0, 0};
  SearchSpace sspace;
  std::unique_ptr<RNTIManager> rnti_manager;
  std::unique_ptr<DCIMetaFormats> meta_formats;
  uint32_t size_of_format[NOF_FORMATS] = {0};
  int size_index_of_format[NOF_FORMATS] = {0};
  uint32_t size_list[LSN_MAX_SIZES] = {0}, nsizes = 0;
  std::vector<TempDci0> temp_dci0;
  std::atomic<uint32_t> coll_dw{0}, coll_up{0};  // nof_subframe_collisions_dw / _up, counted by finishSubframe
  LocSet f_used, f_occupied, f_checked, f_nopower;
  const LocTemplate* cur_tp = nullptr;
  LocTemplate loc_template[3];
  const LsnCand* cur_cand = nullptr;
  // a slot the blind decoder left out (LSN_CAND_NOT_COMPUTED): the owner of the table decodes it now and returns it (Engine::candidateMiss)
  void (*cand_miss)(void* ctx, uint32_t li, uint32_t size_index) = nullptr;   // (fills the slot in both views)
  const uint32_t* cur_cand4 = nullptr;
  void* cand_miss_ctx = nullptr;
  const float* cur_ccepow = nullptr;
  BlindStats stats;
  bool shortcut_discovery = true;  // Settings.h / ArgManager default (DCISearch.cc:200: enableShortcutDiscovery)
};

const char* rnti_name(uint16_t r);  // DL_Sniffer_PDSCH.cc:1398-1418

// DCICollection::addCandidate, DCICollection.cc:107-134: the MCS table an accepted DCI is collected under (the tracking database is asked for every user
// RNTI that does not come in format 1A - the look-up refreshes the entry's time stamp)
inline McsTable collection_table(int mcs_tracking_mode, uint16_t rnti, DciFormat format, MCSTracking& mcs_tracking, uint32_t now)
{
  if (mcs_tracking_mode == 1)
    return (rnti == SIRNTI || rnti == PRNTI || rnti_israr(rnti) || format == FORMAT1A) ? TABLE_64QAM : mcs_tracking.find_tracking_info_RNTI_dl(rnti, now);
  return mcs_tracking_mode == 2 ? TABLE_UNKNOWN : TABLE_64QAM;
}
// srsran_dci_msg_to_trace_timestamp, falcon_dci.c:284-310: which of the two grants of a downlink entry the reference computes under that table, and
// whether the unpacked DCI keeps its RNTI (a failed conversion sets it to 0: "to avoid decode")
// DCICollection.cc:236-251: with HARQ on, a reserved MCS index (29-31: "the size of the previous transmission") of a 64QAM-table grant takes the size the HARQ
// database remembers for (RNTI, process, block).  (The reference's 256QAM-table branch writes into the grant it did not compute for that entry: no effect.)
// Returns true when a size was put in.
inline bool collection_last_tbs(bool harq_mode, McsTable table, DlEntry& e, const HarqDatabase& harq)
{
  bool put = false;
  if (harq_mode && table == TABLE_64QAM && e.unpack_ok)
    for (int i = 0; i < 2; i++)
      if (e.grant64.tb[i].enabled && e.grant64.tb[i].mcs_idx > 28) { e.grant64.tb[i].tbs = harq.getlastTbs(e.rnti, e.dci.pid, i); put = true; }
  return put;
}
struct TableView { bool has64, has256, dci_rnti_ok; };
inline TableView table_view(McsTable table, uint16_t rnti, bool unpack_ok, bool ok64, bool ok256)
{
  TableView v;
  v.has64 = unpack_ok && (table == TABLE_64QAM || table >= TABLE_UNKNOWN);
  v.has256 = unpack_ok && (table == TABLE_256QAM || table >= TABLE_UNKNOWN);
  v.dci_rnti_ok = rnti > 0 && !(v.has64 && !ok64) && !(v.has256 && !ok256);
  return v;
}

}  // namespace lsn
