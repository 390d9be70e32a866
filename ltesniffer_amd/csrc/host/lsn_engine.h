// lsn_engine.h - the batched, software-pipelined GPU engine behind Phy / SubframeWorker (product; no CPU fallback).
//   stage A (GPU, stream A): OFDM -> CRS estimation -> PCFICH -> PDCCH LLRs -> CCE power -> exhaustive Viterbi candidate table
//   stage B (CPU, caller thread): FALCON decision tree per subframe in TTI order over the candidate table
//                  (DCISearch.cc:102-528); RAR grants are decoded on demand because they feed the RNTI manager
//                  (DL_Sniffer_PDSCH.cc:782-797)
//   stage C (GPU, stream C): PDSCH demodulation + turbo decoding of every accepted grant
//   commit  (CPU, commit thread): decode_dl_mode selection logic in (tti, DCI, TB) order, MCS-table learning, PDU sink
// A call is cut into chunks of max_batch subframes; chunk i+1 runs stage A while chunk i is searched and chunk i-1 is
// decoded/committed (three chunk slots).
#pragma once
#include "../../../include/ltesniffer_amd.h"
#include "../kernels/lsn_dev.h"
#include "lsn_search.h"
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

namespace lsn {

struct DecodeJob {
  uint32_t sf = 0; PdschGrant grant; uint16_t rnti = 0;
  float p_a = 0.0f;  // pdsch_cfg->p_a this decode runs with (dB)
  bool planned = false, done = false;
  bool risky = false;  // a table guess for a UE whose MCS table is unknown: hopeless (12 iterations per first block) when the guess is wrong
  uint8_t kind = 0, used = 0;  // lsn_perf_t::jobs_by_kind; used: the commit stage looked at the result
  uint32_t cb_first = 0, cb_count[2] = {0, 0};
  uint32_t payload_off[2] = {0, 0};
  bool crc[2] = {false, false};
  uint32_t iters = 0;
  uint32_t keep_first[2] = {0, 0}, keep_count[2] = {0, 0};  // harq_mode: this job's code blocks in Chunk::keep_cbs (their soft data sits in Chunk::d_keep)
};

// Compact views for the sequential commit thread (it walks them linearly instead of chasing the wide DlEntry / DecodeJob records that
// other threads wrote): one JobRes per decode job (same index), one CommitDci per accepted downlink DCI in (subframe, acceptance) order.
struct JobRes {
  uint8_t done = 0, crc[2] = {0, 0}, enabled[2] = {0, 0};
  uint8_t nsetup[2] = {0, 0};          // RRCConnectionSetups found in a CRC-ok transport block (pre-parsed by the thread that ran the decode)
  float p_a = 0.0f;
  uint32_t payload_off[2] = {0, 0};
  int32_t len[2] = {0, 0};             // tbs / 8
  uint32_t setup_first[2] = {0, 0};    // into Chunk::setup_cfgs
};
struct CommitDci {
  uint16_t rnti = 0; uint8_t format = 0, flags = 0;   // flags: 1 unpack_ok, 2 ok64, 4 ok256, 8 grant64 has two TBs, 16 grant256 has two TBs
  uint8_t en64 = 0, en256 = 0, mcs_idx[2] = {0, 0};   // en*: bit i = tb[i].enabled of that table's grant
  int32_t tbs0_64 = 0, tbs0_256 = 0;                  // tb[0].tbs of the two grants
  int32_t job[2] = {-1, -1};
  uint32_t di = 0;                                    // index in SubframeCtx::dl (slow path: a decode has to be created at commit)
};

// stage-C taps (lsn_phy_set_stage_c_taps): what one decode job left behind in the launch arenas, copied out before they are recycled
struct TapCb { LsnCbDev cb; uint32_t tb = 0; LsnCbRes res{}; std::vector<uint32_t> words; };
struct TapJob { bool have = false; LsnGrantDev d{}; std::vector<int16_t> llr[2]; std::vector<TapCb> cbs; };

// everything one chunk of subframes owns while it travels through the pipeline
struct Chunk {
  uint32_t nsf = 0, start_tti = 0;
  cf32 *d_grid = nullptr, *d_ce = nullptr, *d_ul_grid = nullptr;
  hipStream_t st_a = nullptr;        // stage-A stream of this trip through the pipeline
  const cf32* d_iq_src = nullptr;    // the caller's samples of this chunk ([sf][antenna][sflen]), valid until the call returns
  float *d_chest_raw = nullptr, *d_llr = nullptr, *d_ccepow = nullptr, *d_pcfich_corr = nullptr, *d_rbp = nullptr, *d_rbp_part = nullptr;
  LsnChest* d_chest = nullptr;
  uint32_t *d_cfi = nullptr, *d_sfidx = nullptr;
  uint32_t* d_prune_snap = nullptr; uint8_t* d_acc = nullptr;   // candidate pruning (k_viterbi): the RNTI manager's snapshot this chunk was decoded with; what each decoded slot lets the search do
  LsnCand* d_cand = nullptr;
  uint32_t *d_cand4 = nullptr, *h_cand4 = nullptr;   // the search's one-word view of the candidate table (LSN_CAND_HOT) and its host mirror
  LsnCand* h_cand = nullptr; float* h_ccepow = nullptr; LsnChest* h_chest = nullptr; uint32_t* h_cfi = nullptr; float* h_rbp = nullptr;
  uint32_t* h_sfidx = nullptr;
  uint32_t *d_dphi = nullptr, *h_dphi = nullptr;  // per-subframe NCO increment of k_ofdm (CFO correction on), device + pinned mirror
  float cfo_corr_hz = 0.0f;          // the carrier offset k_ofdm removes from this chunk (0 = correction off)
  int cfo_slot = -1;                 // where finishStageA leaves this chunk's absolute CFO measurement for the tracking loop (-1: correction off)
  std::vector<SubframeCtx> ctx;
  uint32_t ul_epoch = 0;             // Engine::ul_cfg_epoch when the DCI 0 grants of this chunk were converted
  std::vector<DecodeJob> jobs;
  std::vector<JobRes> jres;            // per job, same index as jobs
  // harq_mode: the de-rate-matched soft data of every decode launch of this chunk is kept until the chunk is committed - the commit decides which
  // transport blocks go into (new transmission that failed) or are combined with (retransmission) the soft buffers of their HARQ processes
  uint32_t* d_keep = nullptr; size_t keep_cap = 0, keep_n = 0;
  std::vector<LsnCbDev> keep_cbs;      // descriptors of the kept blocks (spp_off = word offset in d_keep)
  std::vector<LsnCbRes> keep_res;      // ... and their verdicts from the decode of THIS transmission (same index)
  std::vector<TapJob> tapjobs;         // per job, filled only while the stage-C taps are switched on
  std::vector<CommitDci> cdci;         // built by planJobs
  std::vector<uint32_t> cdci_first;    // [nsf + 1] first CommitDci of each subframe
  std::vector<UeSpecConfig> setup_cfgs;
  std::vector<uint8_t> h_payload;
  struct PduRec { lsn_pdu_ctx_t ctx; size_t off; uint32_t len; uint8_t msg3 = 0; };  // payload at h_payload[off .. off + len); msg3: uplink block of a RAR grant
  std::vector<PduRec> recs;          // records of this chunk in emission order (commit thread -> writer thread)
  std::string err;                   // first error on this chunk's way through the pipeline
  hipEvent_t ev_a[2 * 8 + 1] = {};  // per stage-A kernel class start/stop + "mirrors on host"
  bool timed_a = false;             // this pass of the chunk carries the per-kernel timing events (every Engine::timing_period-th does)
  struct SpecRar { uint32_t sf; uint16_t rnti; DciFormat format; unsigned long long bits; int job; };
  std::vector<SpecRar> spec_rar;     // RA-RNTI grants decoded ahead of the search (front thread)
  bool busy = false;                 // owned by the pipeline (slot not reusable yet)
  uint64_t seq = 0;                  // position in the commit order
  uint64_t gseq = 0;                 // global chunk number (SharedSeq turn taking)
  uint32_t trace_id = 0;             // chunk number inside its submit (LSN_TRACE)
  uint32_t gpos0 = 0;                // stream position (subframes searched before) of the chunk's first subframe
  uint32_t update_meta_period = 0;   // of the submit this chunk belongs to
  bool force_meta = false;           // SubframeWorker::prepare(updateMetaFormats = true) on the first subframe of this chunk (worker pool)
};

// one stream + its device/host arenas for PDSCH decode launches
struct JobRunner {
  hipStream_t stream = nullptr;
  LsnGrantDev* d_jobs = nullptr; size_t jobs_cap = 0;
  LsnCbDev* d_cbs = nullptr; size_t cbs_cap = 0;
  LsnCbRes* d_cbres = nullptr; size_t cbres_cap = 0;
  uint16_t* d_prefix = nullptr; size_t prefix_cap = 0;
  int16_t* d_llr16 = nullptr; size_t llr16_cap = 0;
  uint32_t* d_items = nullptr; size_t items_cap = 0;   // demodulator work items: job << 8 | group of 16 PRBs
  uint32_t* h_items_pinned = nullptr; size_t h_items_cap = 0;
  std::vector<uint32_t> h_items;
  uint32_t* d_spp = nullptr; size_t spp_cap = 0;   // de-rate-matched soft data of every code block of the launch (k_rm -> k_turbo)
  uint8_t* d_payload = nullptr; size_t payload_cap = 0;
  uint8_t* h_payload_pinned = nullptr; size_t h_payload_cap = 0;
  LsnCbRes* h_cbres_pinned = nullptr; size_t h_cbres_cap = 0;
  LsnGrantDev* h_jobs_pinned = nullptr; size_t h_jobs_cap = 0;
  LsnCbDev* h_cbs_pinned = nullptr; size_t h_cbs_cap = 0;
  std::vector<LsnGrantDev> h_jobs; std::vector<LsnCbDev> h_cbs;
  hipEvent_t ev[10] = {};
  uint32_t launches = 0;            // decode launches so far (per-kernel timing events ride on every Engine::timing_period-th)
  hipEvent_t ev_done = nullptr;  // blocking-sync event the owning thread waits on
  lsn_perf_t perf{};
};

// one entry of the ULSchedule databases (ULSchedule.cc:11-138): a DCI 0 / RAR grant waiting for its PUSCH subframe
struct UlSchedGrant { uint16_t rnti = 0; PuschGrant g, g256; uint32_t n_dmrs = 0; bool hopping = false, is_rar = false; uint32_t nof_ack = 0; bool cqi_req = false; };

// The SEQUENTIAL host state of one cell: the FALCON search with its RNTI manager, the MCS-tracking database and the clocks both run on.
// One engine owns one of these; the engines of a capture that is spread over several GPUs (lsn_phy_create_multi: chunk g goes to engine
// g mod G) share one and take turns on it - chunk g is searched, committed and written when chunks 0 .. g-1 have been, whichever engine
// holds them - so the record stream is the one a single engine would produce.
// harq_mode = 1: what srsran_softbuffer_rx_t holds besides the soft values (round-4 advisor finding): per code block cb_crc and the decoded data of a block whose
// CRC passed - a retransmission neither combines nor decodes such a block again (sch.c decode_tb_cb [srsRAN]).  Kept on the host, by soft-buffer slot, touched in
// the commit turn only: verdict, CRC24A contribution and payload bytes of every code block of the transport block the slot holds.
// ver / loc (round 6): WHICH content the buffer holds (a number that names the chain of stores and combinations that produced it - equal numbers, equal
// soft values) and WHERE each of its blocks lies at this point of the commit turn (HARQ_LOC_*: pool, the turn's scratch area, the chunk's keep store).
enum : uint32_t { HARQ_LOC_POOL = 0u, HARQ_LOC_SCRATCH = 1u << 30, HARQ_LOC_KEEP = 2u << 30, HARQ_LOC_PLACE = 3u << 30 };
struct HarqKeep { uint32_t ncb = 0; uint8_t ok[16] = {}; uint32_t rem_a[16] = {}; std::vector<uint8_t> bytes[16]; uint64_t ver = 0; uint32_t loc[16] = {}, K[16] = {}; };

struct SharedSeq {
  std::unique_ptr<FalconSearch> search;
  MCSTracking mcs_tracking;
  std::atomic<float> default_p_a{0.0f};  // p-a of RNTIs without tracking entry (RA-RNTIs): readable by the search / front thread without mcs_mtx
  std::mutex mcs_mtx;                     // commit thread (authoritative updates) vs the API getter; the decode threads read the prediction arrays below
  // per RNTI: tracked table (0xFF: no entry) and p-a as of the last commit that touched the RNTI; relaxed atomics, prediction only
  std::unique_ptr<std::atomic<uint8_t>[]> pred_table{new std::atomic<uint8_t>[65536]};
  std::unique_ptr<std::atomic<float>[]> pred_p_a{new std::atomic<float>[65536]};
  // a RAR the search has seen for this RNTI but the commit has not reached yet will reset the RNTI's table (update_rar_time_crnti): predict that
  std::unique_ptr<std::atomic<uint32_t>[]> pred_rar_at{new std::atomic<uint32_t>[65536]};  // 1 + subframe count (search side) of the latest RAR naming the RNTI, 0 = none
  // Plan-side knowledge that bridges the plan -> commit lag (thousands of subframes with eight chunks in decode at once): positions of the last
  // "teaching" decodes the DECODE threads have seen for an RNTI - an unknown-table DCI of a format > 1A whose 64QAM-table attempt failed on every
  // block and whose 256QAM-table attempt passed with a learnable MCS index, i.e. exactly what makes the commit set the RNTI's table to 256QAM
  // (update_RNTI_dl).  Once LSN_HINT_EVENTS (6: the first adds the entry, a RAR-introduced entry needs more than three messages first) such events lie
  // between the last reset of the entry (RAR naming the RNTI, database ageing) and a DCI, the commit WILL find the table known when it gets there,
  // and the plan leaves the hopeless 64QAM-table attempt out.  A prediction only: a commit that does want that attempt decodes it on demand.
  static constexpr int HINT_RING = 16, HINT_EVENTS = 6;
  std::unique_ptr<std::atomic<uint32_t>[]> hint_pos{new std::atomic<uint32_t>[65536 * HINT_RING]};  // position + 1 of an event, 0 = empty
  std::unique_ptr<std::atomic<uint8_t>[]> hint_next{new std::atomic<uint8_t>[65536]};
  std::atomic<uint32_t> hint_floor{0};     // events at positions below are void (the caller aged the database by hand)
  std::atomic<uint64_t> hint_used{0}, hint_missed{0};
  std::atomic<uint32_t> commit_pos{0};    // subframes committed so far, published
  uint32_t commit_sf_cnt = 0;             // subframes committed so far = the tracking database's clock (1 subframe = 1 ms)
  // HARQ soft combining (harq_mode = 1): the process database, the host side of the soft buffers and the device pool they live in.  With several engines on
  // one capture the pool belongs to the FIRST engine's device and the others reach it over the peer link (lsn_phy_create_multi enables the access): the
  // commit turns are sequential, so one engine at a time reads or writes it
  HarqDatabase harq_db;
  std::unordered_map<size_t, HarqKeep> harq_keep;
  uint32_t* d_harq_pool = nullptr;
  const void* harq_pool_owner = nullptr;   // the engine whose device holds the pool (and frees it)
  uint32_t mcs_update_period = 5000;      // MCSTracking::get_interval() x 1000 subframes (LTESniffer_Core.cc:473-485); 0: never
  uint64_t nof_mcs_db_updates = 0;
  uint64_t sf_cnt = 0;                    // subframes searched so far
  double search_time_us = 0;              // time_blindsearch of the statistics (PhyCommon.cc:111-112)
  float est_cfo = 0;
  bool force_meta_next = false;
  // UL_MODE (touched in the commit turn, or under mcs_mtx where the caller may look): ULSchedule databases, the uplink tracking database
  // (MCSTracking UL: 0 absent, 1 unknown, 2/3/4 = 16/64/256QAM max; ue_spec_config, time in subframes, nof_active, nof_success_mgs) and the
  // uplink configuration in force.  An engine whose device tables were built for an older epoch rebuilds them at its next commit turn
  // (Engine::syncUlConfig), so a configuration learnt from SIB2 by one engine of a multi-GPU capture reaches the others in stream order.
  std::map<uint32_t, std::vector<UlSchedGrant>> ul_sched, rar_sched;
  std::vector<uint8_t> ulmod; uint32_t ulmod_count = 0;
  std::vector<UeSpecConfig> ul_uecfg;
  std::vector<uint32_t> ul_time, ul_active, ul_success;
  float last_ul_snr = 0.0f;
  lsn_ul_cfg_t ul_cfg{}; bool ul_set = false;
  bool sib2_learned = false; Sib2Config sib2;
  std::atomic<uint32_t> ul_cfg_epoch{0};
  lsn_prach_cfg_t prach_cfg{}; bool prach_cfg_set = false; uint32_t prach_epoch = 0;
  // turn taking
  std::atomic<uint64_t> next_gseq{0};     // global chunk numbers in submission order
  std::mutex turn_mtx;
  std::condition_variable turn_cv;
  uint64_t search_turn = 0, commit_turn = 0, write_turn = 0;
  SharedSeq()
  {
    for (uint32_t i = 0; i < 65536; i++) { pred_table[i].store(0xFF, std::memory_order_relaxed); pred_p_a[i].store(0.0f, std::memory_order_relaxed); pred_rar_at[i].store(0, std::memory_order_relaxed); hint_next[i].store(0, std::memory_order_relaxed); }
    for (uint32_t i = 0; i < 65536u * HINT_RING; i++) hint_pos[i].store(0, std::memory_order_relaxed);
  }
};

class Engine {
public:
  explicit Engine(const lsn_phy_cfg_t& cfg, std::shared_ptr<SharedSeq> shared = nullptr);
  std::shared_ptr<SharedSeq> sharedState() { return sh; }
  int device() const { return cfg.device; }
  // a block that lives on ANOTHER device (one capture spread over several GPUs): copied over the peer link into this engine's staging ring
  int submitFrom(const void* d_iq, int src_device, uint32_t nsf, uint32_t start_tti, uint32_t update_meta_period, hipStream_t stream);
  ~Engine();
  int setCell(const lsn_cell_t& cell);
  bool hasCell() const { return cell_set; }
  int process(const void* d_iq, uint32_t nsf, uint32_t start_tti, uint32_t update_meta_period, hipStream_t stream);
  int submit(const void* d_iq, uint32_t nsf, uint32_t start_tti, uint32_t update_meta_period, hipStream_t stream, bool force_meta_first = false);
  // `nrows` = nsf x antennas rows of one subframe each (sflen samples), `row_pitch` bytes apart in PINNED host memory (the worker pool's slab:
  // SubframeBuffer keeps 3 SF_LEN per antenna): one strided copy into the staging ring, then submit.  `copied` (optional) is recorded behind the
  // copy: the caller may overwrite the rows once it has completed.  At most max_batch subframes.
  int submitHostRows(const void* host_rows, size_t row_pitch, uint32_t nsf, uint32_t start_tti, bool force_meta_first, hipEvent_t copied);
  int wait();
  uint64_t submitMark() { std::unique_lock<std::mutex> lk(mtx); return chunks_expected; }            // position of the last submitted chunk
  // chunks up to `mark` have finished stage A: nothing reads their IQ block any more in DL mode (UL_MODE keeps it for the PRACH detector -> waitMark)
  void waitStageAMark(uint64_t mark) { std::unique_lock<std::mutex> lk(mtx); cv_done.wait(lk, [&] { return seq_a_done >= mark || stop; }); }
  void waitIqConsumed(uint64_t mark) { if (cfg.sniffer_mode == 1) waitMark(mark); else waitStageAMark(mark); }
  void waitMark(uint64_t mark) { std::unique_lock<std::mutex> lk(mtx); cv_done.wait(lk, [&] { return seq_written >= mark || stop; }); }
  int mibDecode(const void* iq, bool on_device, lsn_mib_t* out, float* llr_raw480);
  int processFile(const char* path, const lsn_file_cfg_t& fc, uint32_t start_tti, uint64_t max_subframes, uint32_t update_meta_period,
                  uint64_t* subframes_done);
  int reserveFileBuffers(uint32_t nof_antennas);   // lsn_phy_prepare_file: pinned read blocks + device blocks of the file source, ahead of the first replay
  int processHost(const void* iq, uint32_t nsf, uint32_t start_tti, uint32_t update_meta_period, uint32_t sample_format = 0 /* LSN_FILE_* */, float sample_scale = 0.0f);
  void setSink(lsn_pdu_sink_t cb, void* user) { sink = cb; sink_user = user; }
  void setApi(int mode, lsn_api_sink_t cb, void* user, lsn_pdu_sink_t pcap_cb, void* pcap) { api_mode = mode; api_sink = cb; api_user = user; api_pcap_sink = pcap_cb; api_pcap = pcap; }
  long tap(int what, uint32_t sf, void* out, size_t cap);
  void setStageCTaps(bool on) { keep_stage_c.store(on); }
  void getPerf(lsn_perf_t* p) const { *p = perf; p->nof_table_hints_used = sh->hint_used.load(); p->nof_table_hints_missed = sh->hint_missed.load(); }
  void getStats(lsn_blind_stats_t* s) const;
  float estCfo() const { return est_cfo; }
  int setCfoCorrection(int mode, float cfo_hz, float alpha);
  // Candidate pruning of the blind decoder (stage_a.hip: k_viterbi): 0 = exhaustive table, 1 = slots under a location whose candidate the search is predicted
  // to accept are left out and decoded on demand if the search comes there after all (default), 2 = test: every RNTI counts as active (the prediction claims
  // far too much, the on-demand path carries the search)
  int setCandidatePruning(int mode) { if (mode < 0 || mode > 2) return LSN_ERROR_INVALID_INPUTS; prune_mode.store(mode); return LSN_SUCCESS; }
  float cfoCorrection() const { return cfo_current.load(); }
  RNTIManager& rntiManager() { return search->rntiManager(); }
  FalconSearch& searchRef() { return *search; }
  double searchTimeUs() const { return search_time_us; }
  uint32_t sfLen() const { return cd.sflen; }
  uint32_t nofRx() const { return cfg.nof_rx_antennas; }
  uint32_t dlRx() const { return cfg.sniffer_mode == 1 ? 1u : cfg.nof_rx_antennas; }  // DCISearch::prepareDCISearch, DCISearch.cc:592
  uint32_t maxBatch() const { return max_batch; }
  void setupDefaultIntervals() { search->setupDefaultIntervals(); }
  int setUlConfig(const lsn_ul_cfg_t& u);
  void uploadUlStatic();   // tables of the uplink OFDM demodulator that do not depend on SIB2
  bool getUlConfig(lsn_ul_cfg_t* u, lsn_prach_cfg_t* p, Sib2Config* sib) const;
  bool sib2Learned() const { return sib2_learned; }
  int trackedModUl(uint16_t rnti) { std::lock_guard<std::mutex> lk(mcs_mtx); return ulmod.empty() ? 0 : (int)ulmod[rnti]; }
  int puschDecode(const void* ul_iq, bool on_device, uint32_t nsf, uint32_t start_tti, const lsn_pusch_grant_t* grants, uint32_t ngrants,
                  lsn_pusch_result_t* results, uint8_t* payloads, size_t payload_cap);
  long tapUl(int what, uint32_t index, void* out, size_t cap);
  int setPrachConfig(const lsn_prach_cfg_t& p);
  int prachDetect(const void* ul_iq, bool on_device, uint32_t nsf, uint32_t start_tti, lsn_prach_det_t* out, uint32_t cap);
  long tapPrach(uint32_t index, void* out, size_t cap);
  void setPrachSink(lsn_prach_sink_t cb, void* user) { prach_sink = cb; prach_sink_user = user; }
  void forceMetaUpdateNext() { force_meta_next = true; }

private:
  static constexpr int NDEC = 16;            // max decode threads (each: plan + stage-C launches of one chunk; commits stay in order)
  static constexpr int NSLOTS = NDEC + 8;
  int ndec = 12, nslots = 20;                 // in use (LSN_DECODE_THREADS; 8 until round 4: 12 threads on 16 hardware queues measured + 3 %)
  void createCopyStream();
  void freeDevice(bool keep_file_buffers = false);
  void buildTables();
  void allocChunk(Chunk& ch);
  void allocRunner(JobRunner& r);
  void freeRunner(JobRunner& r);
  void launchStageA(Chunk& ch, const void* d_iq);
  void finishStageA(Chunk& ch);
  void searchChunk(Chunk& ch, uint32_t update_meta_period);
  void speculateRar(Chunk& ch);
  void planJobs(Chunk& ch, JobRunner& r);
  void buildCommitView(Chunk& ch);
  void runJobs(Chunk& ch, JobRunner& r, std::vector<int>& job_ids);
  void ensureJob(Chunk& ch, JobRunner& r, int j);
  void commitChunk(Chunk& ch, JobRunner& r);
  void commitChunkUl(Chunk& ch, JobRunner& r);
  void prachDetectDev(const cf32* d_iq, uint32_t nant, uint32_t ant, uint32_t nsf, uint32_t start_tti, std::vector<lsn_prach_det_t>& out, uint32_t first_sf = 0);
  void puschDecodeGrid(const cf32* d_grid, uint32_t nsf, uint32_t start_tti, const lsn_pusch_grant_t* grants, uint32_t ngrants,
                       lsn_pusch_result_t* results, std::vector<uint8_t>& payload_out);
  // RA-RNTI grants whose content feeds the RNTI manager: DL mode 2..9 (rnti_name == RA_RNTI, DL_Sniffer_PDSCH.cc:1409), UL mode 1..10 (:373)
  bool isRarFeedbackRnti(uint16_t r) const { return cfg.sniffer_mode == 1 ? (r >= RARNTI_START && r <= RARNTI_END) : (r > RARNTI_START && r < RARNTI_END); }
  bool ulModeDecodesDl(const DlEntry& e) const
  {
    if (e.rnti >= RARNTI_START && e.rnti <= RARNTI_END) return true;
    return e.rnti > RARNTI_END && (e.format == FORMAT1 || e.format == FORMAT1A) && e.rnti != SIRNTI;
  }
  int newJob(Chunk& ch, uint32_t sf, const DlEntry& e, int table, float p_a = 0.0f, int kind = 0);
  void learnUeConfig(const uint8_t* pdu, int len, uint16_t rnti);
  // HARQ soft combining (harq_mode = 1, DL mode): database + device pool of soft buffers (SharedSeq), driven by the commit stage
  static constexpr size_t HARQ_CB_WORDS = LSN_SPP_WORDS(6144u), HARQ_MAX_CB = 16, HARQ_SLOT_WORDS = HARQ_CB_WORDS * HARQ_MAX_CB;
  // Retransmissions are combined and decoded in BATCHES ahead of the sequential commit walk (round-5 review: one GPU round trip per retransmission inside
  // the commit turn held the gated HARQ leg at 4.6 k subframes/s).  A request names its inputs completely: the current transmission (job, block), the content
  // of the buffer it meets (HarqKeep::ver) and the blocks that have passed already; its key is a hash of exactly those, so a result found under the key of
  // the request the walk makes IS the result the walk would have computed.  harqScout predicts the walk's requests on copies of the HARQ state (a guess -
  // nothing but speed depends on it), harqRunBatch runs them into the scratch area, the walk takes what fits and decodes the rest alone, as before.
  struct HarqReq { int job = -1, tb = 0; size_t slot = 0; uint64_t key = 0, ver = 0; uint32_t n = 0; uint8_t ok[16] = {}; uint32_t loc[16] = {}; };
  struct HarqDone { HarqReq req; uint8_t ok[16] = {}; uint32_t rem_a[16] = {}, iters[16] = {}, loc[16] = {}; std::vector<uint8_t> bytes[16]; bool used = false; };
  std::unordered_map<uint64_t, HarqDone> harq_cache;   // results of this commit turn's batches, by request key
  std::vector<size_t> harq_touched;                    // buffers whose content is not (all) in the pool: copied there at the end of the turn
  uint32_t* d_harq_scratch = nullptr; size_t harq_scratch_cap = 0, harq_scratch_n = 0;
  LsnCbDev *harq_h_store = nullptr, *harq_d_store = nullptr; size_t harq_h_store_cap = 0, harq_d_store_cap = 0;   // descriptors of the end-of-turn copies
  static uint64_t harqMix(uint64_t a, uint64_t b, uint64_t c, uint64_t d);
  bool harqRequest(const Chunk& ch, int job, int tb, size_t slot, uint32_t n, uint32_t ncb_have, uint64_t ver, const uint8_t* ok, const uint32_t* loc, HarqReq& q) const;
  void harqStore(Chunk& ch, JobRunner& r, int job, int tb, size_t slot);                       // a failed new transmission becomes the buffer's content (it stays in the keep store until harqFlush)
  void harqFlush(Chunk& ch, JobRunner& r);                                                       // end of the turn: every touched buffer into the pool
  struct HarqEvent { int job = -1; uint32_t now = 0, sfn = 0, sf_idx = 0, n = 0; int tbs = 0; uint16_t rnti = 0; uint8_t pid = 0, tb = 0, rv = 0; bool ndi = false, crc = false; };
  std::vector<HarqEvent> harq_events;   // harqScout: the transport blocks of the chunk in commit that go to the process database
  void harqScout(Chunk& ch, std::vector<HarqReq>& out, bool first_pass);                                          // the combined decodes the walk over this chunk will probably ask for and that have no result yet
  void harqRunBatch(Chunk& ch, JobRunner& r, const std::vector<HarqReq>& reqs);                  // combine + decode all of them, results into harq_cache
  bool harqCombinedDecode(Chunk& ch, JobRunner& r, int job, int tb, size_t slot, uint32_t& payload_off);  // retransmission: result of the combined decode (from a batch, else decoded now)
public:
  UeSpecConfig ueConfig(uint16_t rnti) { std::lock_guard<std::mutex> lk(mcs_mtx); return cfg.sniffer_mode == 1 ? ulUeConfig(rnti) : mcs_tracking.get_ue_config_rnti(rnti); }
private:
  void unpackRar(const uint8_t* p, int len, bool at_search);
  void emitPdu(Chunk& ch, JobRunner& r, const char* name, size_t payload_off, uint32_t len, uint16_t rnti, uint32_t tti, uint8_t tb);
  void writerLoop();
  void decodeLoop(int idx);
  void commitLoop();
  void frontLoop();
  void specLoop();
  void mergePerf(const lsn_perf_t& p);
  void detectNumaCpus();
  bool pinThisThread(void* saved_mask);   // bind the calling thread to the CPUs of the GPU's NUMA node
  void unpinThisThread(const void* saved_mask);

  std::shared_ptr<SharedSeq> sh;  // (declared first: the reference members below bind to it)
  lsn_phy_cfg_t cfg;
  Cell cell;
  bool cell_set = false;
  uint32_t max_batch = 64;
  LsnCellDev cd{};
  std::vector<void*> dev_allocs, host_allocs;
  // CFO correction in the OFDM front end (the north-star's "OFDM FFT + CFO correction"): what srsran_ue_sync does for the reference AHEAD of the worker
  // (cfo_correct_enable_track, LTESniffer_Core.cc:312-316,344; srsran_ue_sync_set_cfo_ref fed by the CRS estimate) happens here inside k_ofdm - an NCO on the
  // samples as they are loaded for the FFT, no extra pass over the capture.  mode 0: off.  mode 1: the fixed offset cfo_start_hz.  mode 2: a tracking loop over
  // the chunks in flight: chunk g is corrected by c[g] = c[g-1] + alpha * (m[g-D] - c[g-1]) with D = NSTREAM_A chunks of loop delay (stage A of D chunks is in
  // flight at once, so m[g-D] is the newest measurement that is ALWAYS there when g is launched - the loop is deterministic, whatever the GPU's timing) and
  // m[k] = c[k] + mean of the chunk's CRS residual estimates: an average of ABSOLUTE offsets, stable for any 0 < alpha <= 1 whatever the delay.
  // All loop state lives on the front thread; the setter only posts a request (cfo_epoch) that the next launched chunk picks up.
  std::atomic<int> cfo_mode{0};
  std::atomic<float> cfo_start_hz{0.0f}, cfo_alpha{0.25f}, cfo_current{0.0f};
  std::atomic<uint32_t> cfo_epoch{0};
  uint32_t cfo_epoch_seen = 0;
  uint64_t cfo_launched = 0;
  float cfo_c = 0.0f, cfo_meas[16] = {};
  void* d_iq_staging = nullptr;
  void* d_iq_raw = nullptr;                   // processHost on integer samples: the raw blocks in front of the conversion (made on first use)
  size_t d_iq_raw_bytes = 0;
  hipStream_t copy_stream = nullptr;          // host -> staging copies of processHost
  hipEvent_t copy_done[3] = {};
  uint64_t peer_marks[12] = {};               // submitFrom / submitHostRows: staging slot -> mark of the chunk that used it last
  uint32_t peer_slot = 0;
  size_t staging_sf = 0;
  Chunk chunks[NSLOTS];
  JobRunner runner_c[NDEC], runner_s, runner_f, runner_k;  // decode threads / search thread (on-demand RAR decodes) / front thread (speculative RAR decodes) / commit thread (on-demand decodes)
#ifndef LSN_NSTREAM_A
#define LSN_NSTREAM_A 4
#endif
  static constexpr int NSTREAM_A = LSN_NSTREAM_A;        // stage A of consecutive chunks overlaps on the GPU (kernels of one stream serialise)
  hipStream_t stream_a[NSTREAM_A] = {};
  hipEvent_t ev_in = nullptr;
  std::vector<hipEvent_t> ev_pool;            // recycled "block ready" events (guarded by mtx)
  hipEvent_t peer_ev[16] = {};                 // submitFrom: "source block ready" events, one per source device (created on that device)
  std::unique_ptr<FalconSearch>& search = sh->search;
  MCSTracking& mcs_tracking = sh->mcs_tracking;
  HarqDatabase& harq_db = sh->harq_db;   // harq_mode = 1: process database, host side of the soft buffers, device pool (shared by the engines of a capture)
  std::unordered_map<size_t, HarqKeep>& harq_keep = sh->harq_keep;
  uint32_t*& d_harq_pool = sh->d_harq_pool;
  std::atomic<float>& default_p_a = sh->default_p_a;
  std::mutex& mcs_mtx = sh->mcs_mtx;
  std::unique_ptr<std::atomic<uint8_t>[]>& pred_table = sh->pred_table;
  std::unique_ptr<std::atomic<float>[]>& pred_p_a = sh->pred_p_a;
  std::unique_ptr<std::atomic<uint32_t>[]>& pred_rar_at = sh->pred_rar_at;
  std::atomic<uint32_t>& commit_pos = sh->commit_pos;
  McsTable predictedTable(uint16_t rnti) const
  {
    const uint32_t ra = pred_rar_at[rnti].load(std::memory_order_relaxed);
    if (ra && ra > commit_pos.load(std::memory_order_relaxed)) return TABLE_UNKNOWN;
    const uint8_t t = pred_table[rnti].load(std::memory_order_relaxed);
    return t == 0xFF ? TABLE_UNKNOWN : (McsTable)t;
  }
  float predictedPa(uint16_t rnti) const { return pred_table[rnti].load(std::memory_order_relaxed) == 0xFF ? default_p_a.load(std::memory_order_relaxed) : pred_p_a[rnti].load(std::memory_order_relaxed); }
  void publishPrediction(uint16_t rnti);
  // Per-kernel HIP timing events cost 2.6 % of the resident rate when every launch carries them (36 event records per chunk; measured in round 4).
  // The two decoder kernels - the roofline's subject - are timed on every launch; the other kernels on every timing_period-th chunk / decode launch,
  // their sums scaled by the period (LSN_KERNEL_TIMING_PERIOD, default 8; 1 = every launch).
  uint32_t timing_period = 8, stage_a_passes = 0;
  bool hintedTable256(uint16_t rnti, uint32_t pos) const;   // SharedSeq::hint_pos: will the commit find this RNTI on the 256QAM table at stream position pos?
  void hintEvent(uint16_t rnti, uint32_t pos);
  int hintEvents(uint16_t rnti, uint32_t pos, uint32_t* lo_out) const;  // teaching events in the ring between the entry's last reset and pos; -1: hints are off
  void ageTrackingDatabase();
  uint32_t& commit_sf_cnt = sh->commit_sf_cnt;
  uint32_t& mcs_update_period = sh->mcs_update_period;
  uint64_t& nof_mcs_db_updates = sh->nof_mcs_db_updates;
public:
  void setMcsUpdateInterval(uint32_t seconds) { mcs_tracking.set_interval(seconds); mcs_update_period = seconds * 1000u; }
  void updateMcsDatabase() { sh->hint_floor.store((uint32_t)sf_cnt, std::memory_order_relaxed); std::lock_guard<std::mutex> lk(mcs_mtx); if (cfg.sniffer_mode == 1) ulAgeDatabase(); else ageTrackingDatabase(); }  // between process calls only
  uint32_t nofTrackedRnti() { std::lock_guard<std::mutex> lk(mcs_mtx); return cfg.sniffer_mode == 1 ? ulmod_count : mcs_tracking.nof_RNTI_member_dl(); }  // nof_RNTI_member_dl / _ul
private:
  // front thread: launches stage A chunk after chunk, hands finished chunks to the search (caller) thread
  std::thread front_thread;
  struct FrontJob { const void* d_iq = nullptr; uint32_t nsf_total = 0, start_tti = 0, update_meta_period = 0; uint64_t gseq0 = 0; bool force_meta = false; hipEvent_t ready = nullptr; /* "block is in place" on the caller's stream */
                    int inject_fail = -1; /* test hook (LSN_INJECT_STAGE_A_ERROR=<chunk of this block>): that chunk fails in stage A */ };
  std::deque<FrontJob> front_jobs;            // submits not yet cut into chunks (front thread)
  std::atomic<bool> keep_stage_c{false};      // lsn_phy_set_stage_c_taps: runJobs copies LLRs / de-rate-matched words / verdicts of every job out of its arenas
  std::mutex tap_mtx;                         // (several runners may add jobs to one chunk: decode thread, search thread, commit thread)
  bool cb_skip = true;                        // first-block gating of transport blocks (LSN_NO_CB_SKIP=1 switches it off: every code block is decoded)
  int inject_stage_a_fail = -1;               // LSN_INJECT_STAGE_A_ERROR (test hook, read by the constructor, consumed by the first submit)
  uint64_t chunks_expected = 0;               // chunks of all submits so far; wait() returns when as many have been written
  std::thread search_thread;                  // stage B: the sequential FALCON search, chunk after chunk
  lsn_perf_t perf_search{};
  // candidate pruning: snapshots of the RNTI manager's state (active RNTIs as bits + the primary-format mask), published by the search after every chunk into a
  // pinned ring; the front thread uploads the newest one with the stage-A launches of its next chunk.  A prediction only: a snapshot that is old, or torn by a
  // publish that overtakes the upload, costs on-demand decodes, never a wrong table
  static constexpr uint32_t PRUNE_RING = 64;
  uint32_t* h_prune_ring = nullptr;
  std::atomic<uint32_t> prune_pub{0};
  std::atomic<int> prune_mode{1};
  LsnPruneCfg pruneConfig();
  void publishPruneSnapshot();
  struct CandMissCtx { Engine* e; Chunk* ch; uint32_t sf; } cand_miss_ctx{nullptr, nullptr, 0};
  static void candMissTramp(void* ctx, uint32_t li, uint32_t szi);
  void candidateMiss(Chunk& ch, uint32_t sf, uint32_t li, uint32_t szi);
  void searchLoop();
  std::deque<Chunk*> search_queue, spec_queue;   // front -> spec (speculative RA-RNTI decodes) -> search
  std::thread spec_thread;
  std::condition_variable cv_spec;
  std::condition_variable cv_front, cv_search;
  std::string front_error;
  lsn_perf_t perf_front{};
  // decode threads
  std::thread decode_threads[NDEC];
  std::thread commit_thread;                    // commits decoded chunks in queue order (PDU order and MCS-table learning stay in TTI order)
  std::map<uint64_t, std::pair<Chunk*, std::string>> decoded;  // seq -> chunk whose decode launches have completed (+ error text)
  std::condition_variable cv_commit;
  std::thread writer_thread;                    // hands the committed records to the PDU sink / pcap writer, chunk by chunk in commit order
  std::deque<Chunk*> write_queue;
  std::condition_variable cv_write;
  uint64_t seq_written = 0;
  uint64_t seq_a_done = 0;                      // chunks whose stage A has completed (front thread, in submission order)
  uint64_t slot_counter = 0;                    // chunks acquired so far (front thread)
  uint64_t seq_pushed = 0, seq_committed = 0;  // chunks queued / committed (commit order = queue order)
  std::mutex mtx;
  std::condition_variable cv_work, cv_done;
  std::deque<Chunk*> commit_queue;
  bool stop = false;
  std::string commit_error, submit_error;
  bool batch_open = false;  // submits since the last wait()
  double t_batch = 0;
  lsn_perf_t perf{};
  float& est_cfo = sh->est_cfo;
  lsn_pdu_sink_t sink = nullptr; void* sink_user = nullptr;
  int api_mode = -1; lsn_api_sink_t api_sink = nullptr; void* api_user = nullptr; lsn_pdu_sink_t api_pcap_sink = nullptr; void* api_pcap = nullptr;  // run_api_dl_mode
  uint64_t& sf_cnt = sh->sf_cnt;
  double& search_time_us = sh->search_time_us;
  Chunk* last_chunk = nullptr;
  bool& force_meta_next = sh->force_meta_next;
  // uplink: the sequential state lives in SharedSeq (see there); per engine: the device tables of the configuration and the buffers
  lsn_ul_cfg_t& ul_cfg = sh->ul_cfg;
  bool& ul_set = sh->ul_set;
  uint32_t ul_tables_epoch = 0, prach_tables_epoch = 0;   // shared epochs this engine's device tables were built for
  int buildUlTables(const lsn_ul_cfg_t& u);             // DMRS base sequences, n_PN, u(ns) / v(ns), hopping offset: the device side of setUlConfig
  void syncUlConfig();                                    // commit turn: pick up a configuration another engine (or the caller) has set since
  // device / pinned blocks of the file source, kept between lsn_phy_process_file calls (allocating them costs more than replaying a short capture)
  struct FileBuf { cf32* h_raw = nullptr; cf32* d_raw = nullptr; cf32* d_iq = nullptr; size_t bytes = 0; };
  FileBuf file_buf[8];
  std::atomic<uint32_t>& ul_cfg_epoch = sh->ul_cfg_epoch;  // bumped by every (re)configuration: chunks whose DCI 0 grants were converted earlier are converted again at commit
  bool& sib2_learned = sh->sib2_learned; Sib2Config& sib2 = sh->sib2;  // the SIB2 the UL-mode commit stage configured itself from (decode_SIB), if any
  bool decodeSib(Chunk& ch, JobRunner& r, uint32_t sf, Sib2Config& out, size_t& payload_off, uint32_t& len, uint8_t& tb);
  std::vector<int> ul_off;       // allocation size L -> offset into ul_base / ul_idft (-1: unsupported)
  uint32_t ul_npn[20] = {0};
  uint32_t ul_base_stride = 0;   // cf32 per (u, v) variant of the base-sequence table
  uint8_t ul_u[20] = {0}, ul_v[20] = {0};  // sequence group / base sequence number of the 20 slots (group / sequence hopping of SIB2)
  std::map<uint32_t, std::vector<UlSchedGrant>>& ul_sched = sh->ul_sched;  // ULSchedule databases (touched in the commit turn only)
  std::map<uint32_t, std::vector<UlSchedGrant>>& rar_sched = sh->rar_sched;
  std::vector<uint8_t>& ulmod = sh->ulmod; uint32_t& ulmod_count = sh->ulmod_count;
  std::vector<UeSpecConfig>& ul_uecfg = sh->ul_uecfg;
  std::vector<uint32_t>& ul_time = sh->ul_time; std::vector<uint32_t>& ul_active = sh->ul_active; std::vector<uint32_t>& ul_success = sh->ul_success;
  float& last_ul_snr = sh->last_ul_snr;
  void ulTrackAdd(uint16_t rnti, int mod = 1);                        // add_RNTI_ul, MCSTracking.cc:57-69
  void ulAgeDatabase();                                               // update_database_ul, MCSTracking.cc:86-176

  UeSpecConfig ulUeConfig(uint16_t rnti) const { return (!ulmod.empty() && ulmod[rnti]) ? ul_uecfg[rnti] : mcs_tracking.default_config(); }
  std::vector<LsnUlGrantDev> ul_last_gd; std::vector<int> ul_last_idx;  // descriptors of the last puschDecode call (taps)
  JobRunner runner_u;
  cf32 *ul_d_iq = nullptr, *ul_d_grid = nullptr, *ul_d_hs = nullptr; float* ul_d_stat = nullptr; LsnUlGrantDev* ul_d_grants = nullptr;
  size_t ul_iq_cap = 0, ul_grid_cap = 0, ul_hs_cap = 0, ul_stat_cap = 0, ul_grants_cap = 0;
  LsnUlGrantDev* ul_h_grants = nullptr; size_t ul_h_grants_cap = 0;
  float* ul_h_stat = nullptr; size_t ul_h_stat_cap = 0;  // pinned mirror of the grant descriptors
  struct Prach {
    bool set = false;
    lsn_prach_cfg_t cfg{};
    float factor = 60.0f;
    uint32_t ncs = 0, nwin = 1, nroots = 1, last_nocc = 0;
    int N12 = 0, Ncp = 0, b0 = 0;
    cf32 *d_W = nullptr, *d_V = nullptr, *d_D = nullptr, *d_Y = nullptr;
    float *d_corr = nullptr, *d_out = nullptr;
    uint64_t* d_off = nullptr;
    size_t off_cap = 0, y_cap = 0, corr_cap = 0, out_cap = 0;
  } prach;
  cf32* mib_d_iq = nullptr; float* mib_d_llr = nullptr; LsnCand* mib_d_cand = nullptr;
  lsn_prach_sink_t prach_sink = nullptr; void* prach_sink_user = nullptr;
  std::vector<int> numa_cpus;  // CPUs local to the GPU (empty: unknown, no pinning)
  // LSN_TRACE=<file>: pipeline event log (thread, event, chunk, ms since the first event), written by wait(); tools/trace_gantt.py reads it
  struct TraceEv { double t; uint8_t thr, ev; uint32_t chunk; };
  std::vector<TraceEv> trace_log;
  std::mutex trace_mtx;
  const char* trace_path = nullptr;
  void trace(uint8_t thr, uint8_t ev, uint32_t chunk);
  void traceDump();
};

bool prach_tti_opportunity(uint32_t config_idx, uint32_t tti);  // lsn_prach.cc
int cell_search(int device, const cf32* iq, bool on_device, uint64_t nsamples, uint32_t nof_prb, const lsn_cell_search_cfg_t& cfg, lsn_cell_search_t& out,
                float* corr_out);  // lsn_sync.cc
// table builders (lsn_tables.cc)
void gold_sequence(uint32_t cinit, uint8_t* c, int len);

}  // namespace lsn
