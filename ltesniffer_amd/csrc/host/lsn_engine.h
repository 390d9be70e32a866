// lsn_engine.h - the batched GPU pipeline behind Phy / SubframeWorker (product; no CPU fallback).
//   stage A (GPU): OFDM -> CRS estimation -> PCFICH -> PDCCH LLRs -> CCE power -> exhaustive Viterbi candidate table
//   stage B (CPU): FALCON decision tree per subframe in TTI order over the candidate table (DCISearch.cc:102-528),
//                  RAR grants decoded on demand because they feed the RNTI manager (DL_Sniffer_PDSCH.cc:782-797)
//   stage C (GPU): PDSCH demodulation + turbo decoding of every accepted grant, both MCS tables where the table is unknown
//   commit  (CPU): decode_dl_mode selection logic in (tti, DCI, TB) order, MCS-table learning, PDU sink
#pragma once
#include "../../../include/ltesniffer_amd.h"
#include "../kernels/lsn_dev.h"
#include "lsn_lte.h"
#include <memory>
#include <vector>

namespace lsn {

struct DciMsg { uint8_t payload[64] = {}; uint32_t nof_bits = 0; DciFormat format = FORMAT0; };
struct DciCandidate { uint16_t rnti = 0; DciMsg msg; uint32_t search_space_match_result = 0; };

struct DlEntry {  // DL_Sniffer_DCI_DL (Sniffer_dependency.h:90)
  uint16_t rnti = 0; DciFormat format = FORMAT1; uint32_t nof_bits = 0, L = 0, ncce = 0, histval = 0;
  DciDl dci; bool unpack_ok = false;
  PdschGrant grant64, grant256; bool ok64 = false, ok256 = false;  // both tables computed; selection happens at commit
  int job[2] = {-1, -1};                                          // decode job index per table
};
struct UlEntry { uint16_t rnti = 0; uint32_t nof_bits = 0, L = 0, ncce = 0, histval = 0; DciUl dci; PuschGrant grant; bool ok = false; };

struct DecodeJob {
  uint32_t sf = 0; PdschGrant grant; uint16_t rnti = 0;
  bool planned = false, done = false;
  uint32_t cb_first = 0, cb_count[2] = {0, 0};
  uint32_t payload_off[2] = {0, 0};
  bool crc[2] = {false, false};
  uint32_t iters = 0;
};

struct SubframeCtx {
  uint32_t tti = 0, sf_idx = 0, sfn = 0, cfi = 0;
  float snr_db = 0, cfo_hz = 0;
  bool searched = false;
  std::vector<DlEntry> dl;
  std::vector<UlEntry> ul;
  std::vector<uint32_t> accepted;  // 6 words per accepted DCI: rnti, format, L, ncce, nof_bits, histval
};

class Engine {
public:
  explicit Engine(const lsn_phy_cfg_t& cfg);
  ~Engine();
  int setCell(const lsn_cell_t& cell);
  bool hasCell() const { return cell_set; }
  int process(const void* d_iq, uint32_t nsf, uint32_t start_tti, uint32_t update_meta_period, hipStream_t stream);
  int processHost(const float* iq, uint32_t nsf, uint32_t start_tti, uint32_t update_meta_period);
  void setSink(lsn_pdu_sink_t cb, void* user) { sink = cb; sink_user = user; }
  long tap(int what, uint32_t sf, void* out, size_t cap);
  void getPerf(lsn_perf_t* p) const { *p = perf; }
  void getStats(lsn_blind_stats_t* s) const { *s = stats; }
  float estCfo() const { return est_cfo; }
  RNTIManager& rntiManager() { return *rnti_manager; }
  uint32_t sfLen() const { return cd.sflen; }
  uint32_t nofRx() const { return cfg.nof_rx_antennas; }
  uint32_t maxBatch() const { return max_batch; }
  void* stagingDevice() { return d_iq_staging; }
  void setupDefaultIntervals();
  void forceMetaUpdateNext() { force_meta_next = true; }
  void setCfoCorrection(float hz) { cfo_correct_hz = hz; }

private:
  void freeDevice();
  void buildTables();
  void stageA(const void* d_iq, uint32_t nsf, hipStream_t st);
  void searchSubframe(uint32_t sf, bool update_meta);
  void planJobs(uint32_t nsf);
  void runJobs(std::vector<int>& job_ids, hipStream_t st);
  void ensureJob(int j, hipStream_t st);
  void commitSubframe(uint32_t sf, hipStream_t st);
  int newJob(uint32_t sf, const DlEntry& e, int table);
  // FALCON search
  struct FalconLocation { uint32_t L, ncce; bool used, occupied, checked, sufficient_power; uint32_t index; };
  struct CceMap { FalconLocation* location[4]; float power; };
  int inspect_dci_location_recursively(uint32_t sf, CceMap* cce_map, uint32_t ncce, uint32_t L, uint32_t max_depth, MetaFormat** meta_formats,
                                       uint32_t nof_formats, uint32_t enable_discovery, const DciCandidate* parent_cand);
  void recursive_blind_dci_search(uint32_t sf);
  void decodeCandidate(uint32_t sf, const FalconLocation& loc, DciFormat format, DciCandidate& cand);
  void addCandidate(uint32_t sf, const DciCandidate& cand, uint32_t L, uint32_t ncce, uint32_t histval);
  void unpackRar(const uint8_t* p, int len, bool at_search);
  void emitPdu(const char* name, const uint8_t* pdu, uint32_t len, uint16_t rnti, uint32_t tti, uint8_t tb);

  lsn_phy_cfg_t cfg;
  Cell cell;
  bool cell_set = false;
  uint32_t max_batch = 64;
  LsnCellDev cd{};
  std::vector<void*> dev_allocs;
  // batch buffers (device)
  cf32 *d_grid = nullptr, *d_ce = nullptr;
  float *d_chest_raw = nullptr, *d_llr = nullptr, *d_ccepow = nullptr, *d_pcfich_corr = nullptr, *d_rbp = nullptr;
  LsnChest* d_chest = nullptr;
  uint32_t *d_cfi = nullptr, *d_sfidx = nullptr, *d_dphi = nullptr;
  LsnCand* d_cand = nullptr;
  void* d_iq_staging = nullptr;
  // stage C arenas
  LsnGrantDev* d_jobs = nullptr; uint32_t jobs_cap = 0;
  LsnCbDev* d_cbs = nullptr; LsnCbRes* d_cbres = nullptr; uint32_t cbs_cap = 0;
  uint16_t* d_prefix = nullptr; size_t prefix_cap = 0;
  int16_t* d_llr16 = nullptr; size_t llr16_cap = 0;
  uint8_t* d_payload = nullptr; size_t payload_cap = 0;
  // host mirrors (pinned)
  LsnCand* h_cand = nullptr; float* h_ccepow = nullptr; LsnChest* h_chest = nullptr; uint32_t* h_cfi = nullptr; float* h_rbp = nullptr;
  std::vector<LsnGrantDev> h_jobs; std::vector<LsnCbDev> h_cbs; std::vector<LsnCbRes> h_cbres;
  std::vector<uint8_t> h_payload;
  uint8_t* h_payload_pinned = nullptr; size_t h_payload_cap = 0;
  LsnCbRes* h_cbres_pinned = nullptr; size_t h_cbres_cap = 0;
  // DCI size table
  uint32_t size_of_format[NOF_FORMATS]; int size_index_of_format[NOF_FORMATS];
  // state
  std::unique_ptr<RNTIManager> rnti_manager;
  std::unique_ptr<DCIMetaFormats> meta_formats;
  MCSTracking mcs_tracking;
  std::vector<SubframeCtx> ctx;
  std::vector<DecodeJob> jobs;
  std::vector<std::vector<uint32_t>> job_cbs;  // per job: indices into the current cb list (scratch)
  struct TempDci0 { uint16_t rnti; uint32_t L, ncce; DciFormat format; DciCandidate cand; };
  std::vector<TempDci0> temp_dci0;
  std::vector<uint16_t> rb_map_dl, rb_map_ul;
  bool dl_collision = false, ul_collision = false;
  FalconLocation locations[LSN_MAX_LOC];
  lsn_blind_stats_t stats{};
  lsn_perf_t perf{};
  float est_cfo = 0;
  lsn_pdu_sink_t sink = nullptr; void* sink_user = nullptr;
  uint64_t sf_cnt = 0;
  uint32_t last_nsf = 0;
  hipEvent_t ev[2 * LSN_K_COUNT + 2];
  size_t llr16_used = 0, prefix_used = 0, payload_used = 0;
  hipStream_t cur_stream = nullptr;
  bool force_meta_next = false;
  float cfo_correct_hz = 0.0f;
};

// table builders (lsn_tables.cc)
void gold_sequence(uint32_t cinit, uint8_t* c, int len);

}  // namespace lsn
