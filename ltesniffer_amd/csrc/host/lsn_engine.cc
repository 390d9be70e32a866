// lsn_engine.cc - the batched, software-pipelined GPU engine behind Phy / SubframeWorker (see lsn_engine.h).
// Control flow mirrors the reference's worker, re-staged for chunks of subframes:
//   SubframeWorker::work / run_dl_mode      /root/reference/src/src/SubframeWorker.cc:142-235
//   DCISearch::search                       /root/reference/src/src/DCISearch.cc:553-578   (decision tree: lsn_search.cc)
//   PDSCH_Decoder::decode_dl_mode           /root/reference/src/src/DL_Sniffer_PDSCH.cc:881-1291
// Product code: no CPU fallback, nothing from oracle/ is included or linked.
#include "lsn_engine.h"
#include <pthread.h>
#include <atomic>
#include <time.h>
#include <sys/prctl.h>
#include "../kernels/lsn_rm.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <sched.h>
#include <stdexcept>
#include <string>

#define HIP_CHECK(x)                                                                                       \
  do {                                                                                                     \
    hipError_t _e = (x);                                                                                   \
    if (_e != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #x); \
  } while (0)

namespace lsn {

// Host wait for a pipeline event.  hipEventSynchronize keeps the calling core busy for the whole wait on this runtime even
// for hipEventBlockingSync events (measured: thread CPU time == wall time in the wait, six decode threads = four cores of
// polling per rank), which starves the search thread when several ranks share a CPU quota.  Poll-and-sleep instead: the
// waits are 2-15 ms long; the decode threads nap 50 us, the front thread (which feeds the sequential search) 15 us, with the
// threads' timer slack set to 1 us so the naps are that short.  (Spinning instead was measured in round 5: no gain, profiles/r05_exp_session21.txt.)
// LSN_NO_CB_SKIP=1: decode every code block even when the first block of its transport block has already failed (iteration counts then equal the oracle's)
// (read when an engine is made: Engine::cb_skip; the GPU suite runs with it)
// (Round 6: naps of 200 / 500 us in the decode threads and 60 us in the front thread were measured - same rate, same busy cores: the polling is not what the
// decode threads' CPU time goes into, profiles/r06_host_cost.txt.)
static void waitEvent(hipEvent_t ev, long nap_ns = 50000)
{
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return;
    if (e != hipErrorNotReady) HIP_CHECK(e);
    timespec ts{0, nap_ns};
    nanosleep(&ts, nullptr);
  }
}

static double now_ms()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// pipeline event log (LSN_TRACE): thread 0 = caller/search, 1 = front, 2.. = decode threads
enum { TR_ACQ_BEGIN = 0, TR_ACQ_END, TR_A_DONE, TR_SPEC_DONE, TR_SEARCH_BEGIN, TR_SEARCH_END, TR_DEC_BEGIN, TR_W1_LAUNCHED, TR_W1_DONE, TR_W2_DONE,
       TR_COMMIT_BEGIN, TR_COMMIT_END };
void Engine::trace(uint8_t thr, uint8_t ev, uint32_t chunk)
{
  if (!trace_path) return;
  const double t = now_ms();
  std::lock_guard<std::mutex> lk(trace_mtx);
  trace_log.push_back({t, thr, ev, chunk});
}
void Engine::traceDump()
{
  if (!trace_path) return;
  std::lock_guard<std::mutex> lk(trace_mtx);
  if (trace_log.empty()) return;
  static const char* nm[] = {"acq_begin", "acq_end", "stage_a_done", "spec_done", "search_begin", "search_end", "dec_begin", "w1_launched", "w1_done",
                             "w2_done", "commit_begin", "commit_end"};
  if (FILE* f = fopen(trace_path, "w")) {
    const double t0 = trace_log[0].t;
    for (auto& e : trace_log) fprintf(f, "%u %s %u %.4f\n", (unsigned)e.thr, nm[e.ev], e.chunk, e.t - t0);
    fclose(f);
  }
  trace_log.clear();
}

static const int kStageA[8] = {LSN_K_OFDM, LSN_K_CHEST, LSN_K_CHEST_FIN, LSN_K_PCFICH, LSN_K_PDCCH_LLR, LSN_K_CCE_POWER, LSN_K_VITERBI, LSN_K_RB_POWER};

// ------------------------------------------------------------------------------------------------ NUMA placement
// The sequential search reads candidate tables the GPU has just DMA-written; on a two-socket host a thread on the far
// socket pays a remote-memory miss for every lookup.  All engine threads (and the caller while it is inside the
// library) are therefore bound to the CPUs of the GPU's NUMA node (sysfs: PCI device -> numa_node -> cpulist).
void Engine::detectNumaCpus()
{
  numa_cpus.clear();
  if (getenv("LSN_NO_PIN")) return;
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof(bus), cfg.device) != hipSuccess) return;
  std::string id(bus);
  for (auto& ch : id) ch = (char)tolower(ch);
  int node = -1;
  { std::ifstream f("/sys/bus/pci/devices/" + id + "/numa_node"); if (!(f >> node)) node = -1; }
  if (node < 0) return;
  std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
  std::string list;
  if (!(f >> list)) return;
  size_t pos = 0;
  while (pos < list.size()) {  // "0-63,128-191"
    size_t end = list.find(',', pos);
    if (end == std::string::npos) end = list.size();
    const std::string tok = list.substr(pos, end - pos);
    const size_t dash = tok.find('-');
    const int a = atoi(tok.c_str()), b = dash == std::string::npos ? a : atoi(tok.c_str() + dash + 1);
    for (int c = a; c <= b && c < CPU_SETSIZE; c++) numa_cpus.push_back(c);
    pos = end + 1;
  }
}

bool Engine::pinThisThread(void* saved)
{
  if (numa_cpus.empty()) return false;
  cpu_set_t* old = (cpu_set_t*)saved;
  if (old && sched_getaffinity(0, sizeof(cpu_set_t), old) != 0) return false;
  cpu_set_t set;
  CPU_ZERO(&set);
  int n = 0;
  for (int c : numa_cpus)
    if (!old || CPU_ISSET(c, old)) { CPU_SET(c, &set); n++; }  // never widen what the caller was allowed to use
  if (n == 0) return false;
  return sched_setaffinity(0, sizeof(set), &set) == 0;
}

void Engine::unpinThisThread(const void* saved) { (void)sched_setaffinity(0, sizeof(cpu_set_t), (const cpu_set_t*)saved); }

// ------------------------------------------------------------------------------------------------ life cycle
Engine::Engine(const lsn_phy_cfg_t& c, std::shared_ptr<SharedSeq> shared) : sh(shared ? shared : std::make_shared<SharedSeq>()), cfg(c)
{
  if (cfg.nof_rx_antennas < 1 || cfg.nof_rx_antennas > LSN_MAX_RX) throw std::invalid_argument("nof_rx_antennas");
  if (cfg.sniffer_mode != 0 && cfg.sniffer_mode != 1) throw std::invalid_argument("sniffer_mode");
  if (cfg.sniffer_mode == 1 && cfg.nof_rx_antennas != 2) throw std::invalid_argument("UL_MODE needs two antenna buffers");
  if (cfg.harq_mode != 0 && cfg.harq_mode != 1) throw std::invalid_argument("harq_mode");  // 0 (the reference's only reachable value, ArgManager.cc:50) or 1
  if (cfg.harq_mode && cfg.sniffer_mode != 0) throw std::invalid_argument("harq_mode: DL mode only");
  max_batch = cfg.max_batch ? cfg.max_batch : 64;
  if (cfg.max_turbo_iterations <= 0) cfg.max_turbo_iterations = 12;  // SubframeWorker.cc:365
  if (cfg.meta_format_split_ratio <= 0.0) cfg.meta_format_split_ratio = 0.99;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device");
  HIP_CHECK(hipSetDevice(cfg.device));
  detectNumaCpus();
  if (!search) search.reset(new FalconSearch(cfg.histogram_threshold, cfg.meta_format_split_ratio, cfg.skip_secondary_meta_formats != 0));  // (shared: the first engine of a group makes it)
  {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    // Stage A runs at the SAME (lowest) stream priority as the bulk decode chains since the second half of round 5.  Rounds 1-4 gave it the highest
    // ("stage A feeds the sequential search"): its kernels - k_viterbi alone is 384 000 wavefronts per chunk - then take the chip whenever they are
    // queued and the decode chains, which bound the engine, stand still meanwhile: 208-212 k subframes/s against 220-222 k with equal priorities,
    // 119 k against 129 k at 16 dB (profiles/r05_exp_session14.txt, r05_exp_session15.txt; decode chains ABOVE stage A: 219-220 k, then stage A is what
    // the decode threads wait for).
    (void)hi;
    for (auto& sa : stream_a) HIP_CHECK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, lo));
  }
  HIP_CHECK(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
  trace_path = getenv("LSN_TRACE");
  cb_skip = !(getenv("LSN_NO_CB_SKIP") && atoi(getenv("LSN_NO_CB_SKIP")));
  if (cfg.harq_mode) cb_skip = false;  // the soft buffer remembers every code block that passed, also behind a failed first block (HarqKeep): all blocks are decoded
  // test hook of the error path, read once per engine: chunk <n> of the first submitted block fails in stage A (and must not wedge the pipeline)
  if (const char* e = getenv("LSN_INJECT_STAGE_A_ERROR")) inject_stage_a_fail = atoi(e);
  if (const char* e = getenv("LSN_DECODE_THREADS")) ndec = std::max(1, std::min((int)NDEC, atoi(e)));
  else {
    // The default of twelve decode chains (+ four stage-A chains + copies) is tuned for sixteen hardware queues; the HIP runtime gives a process four
    // unless GPU_MAX_HW_QUEUES is exported BEFORE its first HIP call - something a library cannot do for its host (round-4 advisor finding: the Python
    // binding used to set it on import).  On fewer queues the chains wait for each other's kernels: the engine then runs the eight chains measured best
    // on four queues and says so once.
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    const int nq = q ? atoi(q) : 4;
    if (nq < 12) {
      ndec = std::min(ndec, 8);
      static std::atomic<bool> told{false};
      if (!told.exchange(true) && !getenv("LSN_QUIET"))
        fprintf(stderr, "ltesniffer_amd: %d hardware queues (GPU_MAX_HW_QUEUES %s): running %d decode chains instead of %d; export GPU_MAX_HW_QUEUES=16 before the "
                        "first HIP call of the process for the tuned configuration (INTEGRATION.md section 2)\n", nq, q ? q : "unset", ndec, (int)NDEC);
    }
  }
  if (const char* e = getenv("LSN_KERNEL_TIMING_PERIOD")) timing_period = (uint32_t)std::max(0, atoi(e));
  nslots = ndec + 8;
  front_thread = std::thread([this] { pthread_setname_np(pthread_self(), "lsn-front"); frontLoop(); });
  commit_thread = std::thread([this] { pthread_setname_np(pthread_self(), "lsn-commit"); commitLoop(); });
  spec_thread = std::thread([this] { pthread_setname_np(pthread_self(), "lsn-spec"); specLoop(); });
  search_thread = std::thread([this] { pthread_setname_np(pthread_self(), "lsn-search"); searchLoop(); });
  writer_thread = std::thread([this] { pthread_setname_np(pthread_self(), "lsn-writer"); writerLoop(); });
  for (int i = 0; i < ndec; i++)
    decode_threads[i] = std::thread([this, i] {
      char nm[16];
      snprintf(nm, sizeof nm, "lsn-dec%d", i);
      pthread_setname_np(pthread_self(), nm);
      decodeLoop(i);
    });
}

Engine::~Engine()
{
  {
    std::unique_lock<std::mutex> lk(mtx);
    stop = true;
  }
  cv_work.notify_all();
  cv_front.notify_all();
  cv_commit.notify_all();
  cv_spec.notify_all();
  cv_write.notify_all();
  cv_search.notify_all();
  cv_done.notify_all();
  sh->turn_cv.notify_all();
  if (search_thread.joinable()) search_thread.join();
  if (writer_thread.joinable()) writer_thread.join();
  if (front_thread.joinable()) front_thread.join();
  if (spec_thread.joinable()) spec_thread.join();
  if (commit_thread.joinable()) commit_thread.join();
  for (auto& t : decode_threads)
    if (t.joinable()) t.join();
  (void)hipDeviceSynchronize();
  freeDevice();
  for (auto& sa : stream_a)
    if (sa) (void)hipStreamDestroy(sa);
  if (ev_in) (void)hipEventDestroy(ev_in);
  for (auto& e : peer_ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : ev_pool) if (e) (void)hipEventDestroy(e);
  if (copy_stream) (void)hipStreamDestroy(copy_stream);
  for (auto& e : copy_done) if (e) (void)hipEventDestroy(e);
}

int Engine::setCell(const lsn_cell_t& c)
{
  static const uint32_t ng_x6[4] = {1, 3, 6, 12};
  if (c.cp > 1 || c.frame_type != 0 || c.phich_length != 0 || c.phich_resources > 3) return LSN_ERROR_INVALID_INPUTS;
  if ((c.nof_ports != 1 && c.nof_ports != 2 && c.nof_ports != 4) || c.id > 503) return LSN_ERROR_INVALID_INPUTS;
  switch (c.nof_prb) { case 6: case 15: case 25: case 50: case 75: case 100: break; default: return LSN_ERROR_INVALID_INPUTS; }
  cpu_set_t saved_mask;
  const bool pinned = pinThisThread(&saved_mask);  // pinned host buffers are first touched on the GPU's node
  struct Unpin { Engine* e; bool on; cpu_set_t* m; ~Unpin() { if (on) e->unpinThisThread(m); } } unpin{this, pinned, &saved_mask};
  try {
    HIP_CHECK(hipSetDevice(cfg.device));
    (void)hipDeviceSynchronize();
    freeDevice(true);
    cell.nof_prb = c.nof_prb; cell.nof_ports = c.nof_ports; cell.id = c.id; cell.phich_ng_x6 = ng_x6[c.phich_resources]; cell.cp = c.cp;
    buildTables();
    sib2_learned = false;
    if (cfg.sniffer_mode == 1) {
      uploadUlStatic();
      std::lock_guard<std::mutex> lk(mcs_mtx);  // uplink tracking database (shared by the engines of a capture): sized here, never lazily on the commit thread
      ulmod.assign(65536, 0); ul_uecfg.assign(65536, UeSpecConfig()); ul_time.assign(65536, 0); ul_active.assign(65536, 0); ul_success.assign(65536, 0);
      ulmod_count = 0;
    }
    // hipMemset on device memory (dalloc) is asynchronous to the host and runs on the null stream, which the engine's non-blocking streams do not
    // wait for: without this barrier a clear still in flight could wipe what the first stage-A kernels of a fresh engine had just written (seen once
    // the process ran on 16 hardware queues: the records of the first subframes missing in 1 of 40 runs)
    HIP_CHECK(hipDeviceSynchronize());
  } catch (const std::exception& ex) {
    fprintf(stderr, "ltesniffer_amd: %s\n", ex.what());
    return LSN_ERROR;
  }
  cell_set = true;
  return LSN_SUCCESS;
}

void Engine::getStats(lsn_blind_stats_t* s) const
{
  const BlindStats b = search->getStats();
  s->nof_locations = b.nof_locations; s->nof_decoded_locations = b.nof_decoded_locations; s->nof_cce = b.nof_cce;
  s->nof_missed_cce = b.nof_missed_cce; s->nof_subframes = b.nof_subframes;
  s->nof_subframe_collisions_dw = b.nof_subframe_collisions_dw; s->nof_subframe_collisions_up = b.nof_subframe_collisions_up;
}

void Engine::mergePerf(const lsn_perf_t& p)
{
  perf.ms_stage_a += p.ms_stage_a; perf.ms_search += p.ms_search; perf.ms_stage_c += p.ms_stage_c; perf.ms_commit += p.ms_commit;
  perf.algo_bytes += p.algo_bytes; perf.turbo_algo_bytes += p.turbo_algo_bytes; perf.turbo128_algo_bytes += p.turbo128_algo_bytes;
  perf.nof_tb_decodes += p.nof_tb_decodes; perf.nof_cb_decodes += p.nof_cb_decodes; perf.nof_turbo_iterations += p.nof_turbo_iterations;
  perf.ms_search_core += p.ms_search_core; perf.ms_rar += p.ms_rar;
  perf.turbo_cyc_rm += p.turbo_cyc_rm; perf.turbo_cyc_map += p.turbo_cyc_map; perf.turbo_cyc_out += p.turbo_cyc_out; perf.nof_turbo_iterations_run += p.nof_turbo_iterations_run; perf.ms_wait_slot += p.ms_wait_slot;
  perf.nof_candidates_decoded += p.nof_candidates_decoded; perf.nof_ondemand_decodes += p.nof_ondemand_decodes; perf.nof_pdus += p.nof_pdus;
  perf.nof_candidate_misses += p.nof_candidate_misses;
  for (int k = 0; k < 4; k++) perf.nof_ondemand_commit[k] += p.nof_ondemand_commit[k];
  perf.ms_ondemand_commit += p.ms_ondemand_commit;
  for (int k = 0; k < 4; k++) perf.nof_harq_combines[k] += p.nof_harq_combines[k];
  for (int k = 0; k < 3; k++) perf.ms_harq[k] += p.ms_harq[k];
  perf.nof_pusch_2prb_skipped += p.nof_pusch_2prb_skipped;
  perf.nof_pusch_on_unverified_dmrs += p.nof_pusch_on_unverified_dmrs;
  perf.nof_tb_on_derived_tbs += p.nof_tb_on_derived_tbs;
  perf.nof_decode_jobs += p.nof_decode_jobs; perf.nof_decode_jobs_used += p.nof_decode_jobs_used; perf.nof_speculative_jobs += p.nof_speculative_jobs;
  for (int k = 0; k < 5; k++) { perf.jobs_by_kind[k] += p.jobs_by_kind[k]; perf.jobs_unused_by_kind[k] += p.jobs_unused_by_kind[k]; perf.iters_by_kind[k] += p.iters_by_kind[k]; perf.iters_unused_by_kind[k] += p.iters_unused_by_kind[k]; }
  for (int k = 0; k < 16; k++) { perf.kernel_ms[k] += p.kernel_ms[k]; perf.kernel_launches[k] += p.kernel_launches[k]; }
}

// ------------------------------------------------------------------------------------------------ stage A
void Engine::launchStageA(Chunk& ch, const void* d_iq)
{
  hipStream_t st = ch.st_a;
  const uint32_t nsf = ch.nsf;
  for (uint32_t i = 0; i < nsf; i++) ch.h_sfidx[i] = ch.ctx[i].sf_idx;
  lsn_launch_upload(ch.d_sfidx, ch.h_sfidx, nsf * sizeof(uint32_t), st);  // (pinned mirror; not through the copy engine, see lsn_dev.h)
  const cf32* iq = (const cf32*)d_iq;
  ch.d_iq_src = iq;
  // CFO correction (lsn_engine.h): this chunk's offset from the loop, as the NCO increment of every subframe
  const uint32_t* d_dphi = nullptr;
  ch.cfo_corr_hz = 0.0f; ch.cfo_slot = -1;
  if (const int mode = cfo_mode.load()) {
    if (cfo_epoch_seen != cfo_epoch.load()) { cfo_epoch_seen = cfo_epoch.load(); cfo_launched = 0; cfo_c = cfo_start_hz.load(); }
    if (mode == 2 && cfo_launched >= (uint64_t)NSTREAM_A) {
      const float m = cfo_meas[(cfo_launched - (uint64_t)NSTREAM_A) % 16];
      cfo_c = (float)((double)cfo_c + (double)cfo_alpha.load() * ((double)m - (double)cfo_c));  // (products of two floats are exact in double: no contraction can change this)
    }
    ch.cfo_corr_hz = cfo_c; ch.cfo_slot = (int)(cfo_launched % 16);
    cfo_meas[ch.cfo_slot] = cfo_c;  // until finishStageA has measured (a chunk that fails keeps the loop where it is)
    cfo_launched++;
    cfo_current.store(cfo_c);
    // phase increment per sample, 2^32 = one turn, that REMOVES cfo_c (the oracle's o_nco_dphi, same expression)
    const uint32_t dphi = (uint32_t)(int32_t)llrint(-(double)cfo_c / (15000.0 * (double)cd.N) * 4294967296.0);
    for (uint32_t i = 0; i < nsf; i++) ch.h_dphi[i] = dphi;
    lsn_launch_upload(ch.d_dphi, ch.h_dphi, nsf * sizeof(uint32_t), st);
    d_dphi = ch.d_dphi;
  }
  int n = 0;
  ch.timed_a = timing_period && (stage_a_passes++ % timing_period) == 0;
  auto timed = [&](auto&& fn) {
    if (ch.timed_a) HIP_CHECK(hipEventRecord(ch.ev_a[2 * n], st));
    fn();
    if (ch.timed_a) HIP_CHECK(hipEventRecord(ch.ev_a[2 * n + 1], st));
    n++;
  };
  timed([&] { lsn_launch_ofdm(cd, iq, d_dphi, ch.d_grid, nsf, st, ch.d_rbp_part); });
  timed([&] { lsn_launch_chest(cd, ch.d_grid, ch.d_sfidx, ch.d_ce, ch.d_chest_raw, nsf, st); });
  timed([&] { lsn_launch_chest_fin(cd, ch.d_chest_raw, ch.d_chest, nsf, st); });
  timed([&] { lsn_launch_pcfich(cd, ch.d_grid, ch.d_ce, ch.d_chest, ch.d_sfidx, ch.d_cfi, ch.d_pcfich_corr, nsf, st); });
  timed([&] { lsn_launch_pdcch_llr(cd, ch.d_grid, ch.d_ce, ch.d_chest, ch.d_sfidx, ch.d_cfi, ch.d_llr, nsf, st); });
  timed([&] { lsn_launch_cce_power(cd, ch.d_llr, ch.d_cfi, ch.d_ccepow, nsf, st); });
  timed([&] {
    const LsnPruneCfg pc = pruneConfig();
    if (pc.on) {  // the newest snapshot of the RNTI manager's state the search has published
      const uint32_t pub = prune_pub.load(std::memory_order_acquire);
      lsn_launch_upload(ch.d_prune_snap, h_prune_ring + (size_t)((pub + PRUNE_RING - 1) % PRUNE_RING) * LSN_PRUNE_SNAP_WORDS, LSN_PRUNE_SNAP_WORDS * sizeof(uint32_t), st);
    }
    lsn_launch_viterbi(cd, ch.d_llr, ch.d_ccepow, ch.d_cfi, ch.d_sfidx, ch.d_cand, ch.d_cand4, nsf, pc, ch.d_prune_snap, ch.d_acc, st);
  });
  timed([&] { lsn_launch_rb_power(cd, ch.d_rbp_part, ch.d_rbp, nsf, st); });
  if (cfg.sniffer_mode == 1) lsn_launch_ul_fft(cd, iq, cd.iq_nant, 1, ch.d_ul_grid, nsf, st);  // srsran_enb_ul_fft on antenna 1, UL_Sniffer_PUSCH.cc:391-392
  // mirrors for the host stages: posted writes of a copy kernel into the pinned buffers (not the copy engine, lsn_dev.h)
  {
    LsnCopySegs sg;
    sg.add(ch.h_cand, ch.d_cand, (size_t)nsf * LSN_MAX_LOC * LSN_MAX_SIZES * sizeof(LsnCand));
    sg.add(ch.h_cand4, ch.d_cand4, (size_t)nsf * LSN_MAX_LOC * LSN_MAX_SIZES * sizeof(uint32_t));
    sg.add(ch.h_ccepow, ch.d_ccepow, (size_t)nsf * LSN_CCE_STRIDE * sizeof(float));
    sg.add(ch.h_chest, ch.d_chest, (size_t)nsf * sizeof(LsnChest));
    sg.add(ch.h_cfi, ch.d_cfi, (size_t)nsf * sizeof(uint32_t));
    sg.add(ch.h_rbp, ch.d_rbp, (size_t)nsf * 128 * sizeof(float));
    lsn_launch_copy_multi(sg, true, st);
  }
  HIP_CHECK(hipEventRecord(ch.ev_a[16], st));
}

// candidate pruning (stage_a.hip: k_viterbi): the stateless half of the prediction, from the search's format table and the RNTI manager's intervals
LsnPruneCfg Engine::pruneConfig()
{
  LsnPruneCfg pc{};
  pc.on = prune_mode.load() != 0 ? 1u : 0u;
  RNTIManager& rm = search->rntiManager();
  for (uint32_t f = 0; f < 9u && f < (uint32_t)NOF_FORMATS; f++) {
    pc.fmt_size[f] = (uint32_t)search->sizeIndexOfFormat((int)f);
    const auto& ev = rm.evergreenOf(f);
    const auto& fb = rm.forbiddenOf(f);
    if (ev.size() > 4 || fb.size() > 4) pc.on = 0;   // (more intervals than the kernel's table holds: exhaustive decode)
    pc.n_ever[f] = (uint32_t)std::min<size_t>(ev.size(), 4); pc.n_forb[f] = (uint32_t)std::min<size_t>(fb.size(), 4);
    for (uint32_t i = 0; i < pc.n_ever[f]; i++) pc.ever[f][i] = (uint32_t)ev[i].start | ((uint32_t)ev[i].end << 16);
    for (uint32_t i = 0; i < pc.n_forb[f]; i++) pc.forb[f][i] = (uint32_t)fb[i].start | ((uint32_t)fb[i].end << 16);
  }
  return pc;
}
// the search thread, after a chunk: what the RNTI manager and the meta formats hold now
void Engine::publishPruneSnapshot()
{
  if (!h_prune_ring || !search) return;
  const uint32_t pub = prune_pub.load(std::memory_order_relaxed);
  uint32_t* slot = h_prune_ring + (size_t)(pub % PRUNE_RING) * LSN_PRUNE_SNAP_WORDS;
  if (prune_mode.load() == 2) std::memset(slot, 0xFF, 2048 * sizeof(uint32_t));
  else search->rntiManager().activeFreshBits(slot);
  slot[2048] = search->metaFormats().primaryMask();
  prune_pub.store(pub + 1, std::memory_order_release);
}
void Engine::candMissTramp(void* ctx, uint32_t li, uint32_t szi)
{
  CandMissCtx* c = (CandMissCtx*)ctx;
  c->e->candidateMiss(*c->ch, c->sf, li, szi);
}
// The search came to a slot the blind decoder had left out (the prediction claimed an acceptance that the sequential state did not bear out): decoded now, on the
// search runner's stream - a round trip inside the sequential search, which is why the prediction only claims what is all but certain
void Engine::candidateMiss(Chunk& ch, uint32_t sf, uint32_t li, uint32_t szi)
{
  hipStream_t st = runner_s.stream;
  const size_t idx = ((size_t)sf * LSN_MAX_LOC + li) * LSN_MAX_SIZES + szi;
  // A claim that did not hold leaves a whole sub-tree undecoded, and the search walks all of it: everything still missing in the 8-CCE block of this slot is
  // decoded in ONE launch and mirrored level by level (the first version decoded slot by slot: 24 round trips per false claim)
  const uint32_t ncce_tot = cd.nof_cce[ch.ctx[sf].cfi - 1], lim = std::min<uint32_t>(ncce_tot, LSN_MAX_NUM_OF_CCE);
  int L = -1; uint32_t r = li;
  for (int l = 3; l >= 0; l--) { const uint32_t cnt = lim >> l; if (r < cnt) { L = l; break; } r -= cnt; }
  if (L < 0) throw std::runtime_error("candidate miss outside the location table");
  const uint32_t block = r >> (3 - L);
  lsn_launch_viterbi_block(cd, ch.d_llr, ch.d_ccepow, ch.d_cfi, ch.d_sfidx, ch.d_cand, ch.d_cand4, sf, block, pruneConfig(), ch.d_prune_snap, ch.d_acc, st);
  uint32_t off = 0;
  for (int l = 3; l >= 0; l--) {
    const uint32_t cnt = lim >> l, first = block << (3 - l);
    if (first < cnt) {
      const uint32_t n = std::min<uint32_t>(1u << (3 - l), cnt - first);
      const size_t at = ((size_t)sf * LSN_MAX_LOC + off + first) * LSN_MAX_SIZES;
      HIP_CHECK(hipMemcpyAsync(ch.h_cand + at, ch.d_cand + at, (size_t)n * LSN_MAX_SIZES * sizeof(LsnCand), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipMemcpyAsync(ch.h_cand4 + at, ch.d_cand4 + at, (size_t)n * LSN_MAX_SIZES * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    }
    off += cnt;
  }
  HIP_CHECK(hipStreamSynchronize(st));
  perf_search.nof_candidate_misses++;
  if ((ch.h_cand[idx].flags & LSN_CAND_NOT_COMPUTED) || (ch.h_cand4[idx] & (LSN_CAND_NOT_COMPUTED << 16))) throw std::runtime_error("candidate decode on demand left the slot marked");
}

void Engine::finishStageA(Chunk& ch)
{
  waitEvent(ch.ev_a[16], 15000);
  for (int n = 0; n < 8; n++) {
    float ms = 0;
    if (ch.timed_a && hipEventElapsedTime(&ms, ch.ev_a[2 * n], ch.ev_a[2 * n + 1]) == hipSuccess) perf_front.kernel_ms[kStageA[n]] += ms * (float)timing_period;  // (a sample of the chunks, scaled)
    perf_front.kernel_launches[kStageA[n]]++;
  }
  const uint64_t A = dlRx(), P = cell.nof_ports;
  for (uint32_t i = 0; i < ch.nsf; i++) {
    SubframeCtx& c = ch.ctx[i];
    c.cfi = ch.h_cfi[i];
    if (c.cfi < 1 || c.cfi > 3) throw std::runtime_error("stage A mirror of the CFI is out of range (device -> host mirror not in place?)");  // never index tables with it
    // host-side scalars, same expressions as the device-free part of the estimator
    c.snr_db = 10.0f * log10f(ch.h_chest[i].rsrp_avg / ch.h_chest[i].noise_avg);
    c.cfo_hz = atan2f(ch.h_chest[i].corr_i, ch.h_chest[i].corr_r) / (2.0f * (float)M_PI * 0.0005f);
    perf_front.algo_bytes += A * cd.sflen * 8ull + 2ull * A * 14ull * cd.nre * 8ull + 2ull * P * A * 14ull * cd.nre * 8ull +
                       2ull * cd.nof_cce[c.cfi - 1] * 72ull * 4ull;
  }
  if (ch.cfo_slot >= 0) {  // the tracking loop's measurement of this chunk: what was removed + the mean of what the CRS estimator still saw
    double s = 0.0;
    for (uint32_t i = 0; i < ch.nsf; i++) s += (double)ch.ctx[i].cfo_hz;
    cfo_meas[ch.cfo_slot] = (float)((double)ch.cfo_corr_hz + s / (double)ch.nsf);
  }
}

int Engine::setCfoCorrection(int mode, float cfo_hz, float alpha)
{
  if (mode < 0 || mode > 2 || !std::isfinite(cfo_hz) || !(alpha > 0.0f && alpha <= 1.0f)) return LSN_ERROR_INVALID_INPUTS;
  cfo_start_hz.store(cfo_hz); cfo_alpha.store(alpha); cfo_mode.store(mode);
  cfo_current.store(mode ? cfo_hz : 0.0f);
  cfo_epoch.fetch_add(1);  // the next chunk that is launched starts the loop again from cfo_hz
  return LSN_SUCCESS;
}

// RA-RNTI grants feed the RNTI manager between two subframes of the sequential search (DL_Sniffer_PDSCH.cc:782-797).
// Every candidate the search could possibly accept as such a grant - CRC remainder 2..9, format 1A / 1C, a location of
// the common search space - is decoded here, ahead of the search and in one batch per chunk, so that the search thread
// finds the result instead of waiting for a GPU round trip.  Unused speculative decodes are simply dropped.
void Engine::speculateRar(Chunk& ch)
{
  ch.spec_rar.clear();
  std::vector<int> ids;
  const DciFormat fmts[2] = {FORMAT1A, FORMAT1C};
  for (uint32_t sf = 0; sf < ch.nsf; sf++) {
    const SubframeCtx& c = ch.ctx[sf];
    if (!(c.snr_db > 6.0f)) continue;
    const uint32_t ncce = cd.nof_cce[c.cfi - 1], lim = std::min<uint32_t>(ncce, LSN_MAX_NUM_OF_CCE);
    uint32_t li = 0;
    for (int l = 3; l >= 2; l--) {  // aggregation levels 8 and 4 come first in the location enumeration
      const uint32_t L = 1u << l, cnt = lim / L;
      for (uint32_t i = 0; i < cnt; i++, li++) {
        if (L * (i % (ncce / L)) >= 16) continue;  // common search space: first 16 CCEs
        for (DciFormat f : fmts) {
          const LsnCand& q = ch.h_cand[((size_t)sf * LSN_MAX_LOC + li) * LSN_MAX_SIZES + search->sizeIndexOfFormat(f)];
          if (!q.flags || !isRarFeedbackRnti((uint16_t)q.rnti)) continue;
          if (f == FORMAT1A && (q.bits >> 63) == 0) continue;  // that payload is a format 0
          bool dup = false;
          for (auto& s : ch.spec_rar) dup = dup || (s.sf == sf && s.rnti == q.rnti && s.format == f && s.bits == q.bits);
          if (dup) continue;
          DlEntry e;
          if (!search->buildDlEntry(c, (uint16_t)q.rnti, f, q.bits, e) || !e.ok64) continue;
          if (cfg.sniffer_mode == 0 && (!(e.grant64.tb[0].tbs > 0) || (dlRx() == 1 && e.grant64.nof_tb == 2))) continue;
          const int j = newJob(ch, sf, e, 0, default_p_a.load(std::memory_order_relaxed), 3);
          if (j < 0) continue;
          ch.spec_rar.push_back({sf, (uint16_t)q.rnti, f, q.bits, j});
          ids.push_back(j);
        }
      }
    }
  }
  if (!ids.empty()) runJobs(ch, runner_f, ids);
}

// ------------------------------------------------------------------------------------------------ stage B (caller thread)
// The candidate tables were just written by DMA, i.e. none of their lines is in a CPU cache: pull the next subframe's
// one-word view (157 locations x 32 B; round 6 - the 16-byte entries, 20 KB per subframe, cost the search thread a third of its time) towards the core while the
// current subframe is searched.
static inline void prefetch_cand(const uint32_t* cand4, const float* ccepow)
{
  const char* p = (const char*)cand4;   // (the one-word view: 5 KB per subframe; the payload lines are asked for one by one when a candidate shows up)
  for (size_t off = 0; off < (size_t)LSN_MAX_LOC * LSN_MAX_SIZES * sizeof(uint32_t); off += 64) __builtin_prefetch(p + off, 0, 3);
  const char* q = (const char*)ccepow;
  for (size_t off = 0; off < LSN_CCE_STRIDE * sizeof(float); off += 64) __builtin_prefetch(q + off, 0, 3);
}

void Engine::searchChunk(Chunk& ch, uint32_t update_meta_period)
{
  ch.gpos0 = (uint32_t)sf_cnt;
  prefetch_cand(ch.h_cand4, ch.h_ccepow);
  if (ch.nsf > 1) prefetch_cand(ch.h_cand4 + (size_t)LSN_MAX_LOC * LSN_MAX_SIZES, ch.h_ccepow + LSN_CCE_STRIDE);
  for (uint32_t sf = 0; sf < ch.nsf; sf++) {
    SubframeCtx& c = ch.ctx[sf];
    if (sf + 2 < ch.nsf) prefetch_cand(ch.h_cand4 + (size_t)(sf + 2) * LSN_MAX_LOC * LSN_MAX_SIZES, ch.h_ccepow + (size_t)(sf + 2) * LSN_CCE_STRIDE);
    const bool upd = (update_meta_period && (sf_cnt % update_meta_period) == 0) || force_meta_next || (sf == 0 && ch.force_meta);  // LTESniffer_Core.cc:434
    force_meta_next = false;
    sf_cnt++;
    const double ts0 = now_ms();
    cand_miss_ctx = {this, &ch, sf};
    search->setCandMiss(&Engine::candMissTramp, &cand_miss_ctx);
    search->search(c, ch.h_cand + (size_t)sf * LSN_MAX_LOC * LSN_MAX_SIZES, ch.h_ccepow + (size_t)sf * LSN_CCE_STRIDE, upd, ch.h_cand4 + (size_t)sf * LSN_MAX_LOC * LSN_MAX_SIZES);
    { const double dt = now_ms() - ts0; perf_search.ms_search_core += dt; search_time_us += dt * 1e3; }
    est_cfo = c.cfo_hz;  // SubframeWorker.cc:203
    if (!c.searched) continue;
    // RAR grants feed the RNTI manager before the next subframe is searched (DL_Sniffer_PDSCH.cc:782-797): decode them now
    // (the entry objects of the accepted DCIs are otherwise built by the decode threads, FalconSearch::finishSubframe)
    for (const AcceptedDci& a : c.raw)
      if ((DciFormat)a.format != FORMAT0 && isRarFeedbackRnti(a.rnti)) { FalconSearch::materialize(c); break; }
    for (auto& e : c.dl) {
      if (!isRarFeedbackRnti(e.rnti)) continue;
      search->finishDlEntry(e, c.sf_idx, c.cfi);
      const bool dci_ok = e.unpack_ok && e.ok64;
      const bool two_tb = e.grant64.nof_tb == 2;
      if (cfg.sniffer_mode == 0 && !(e.grant64.tb[0].tbs > 0 && dci_ok && !(dlRx() == 1 && two_tb))) continue;
      if (cfg.sniffer_mode == 1 && !e.unpack_ok) continue;
      int j = -1;
      for (auto& s : ch.spec_rar)  // decoded ahead by the front thread?
        if (s.sf == sf && s.rnti == e.rnti && s.format == e.format && s.bits == e.bits && ch.jobs[s.job].done) { j = s.job; break; }
      if (j < 0) {
        j = newJob(ch, sf, e, 0, default_p_a.load(std::memory_order_relaxed), 4);
        if (j < 0) continue;
        const double tr0 = now_ms();
        ensureJob(ch, runner_s, j);
        runner_s.perf.ms_rar += now_ms() - tr0;
        runner_s.perf.nof_ondemand_decodes++;
      }
      e.job[0] = j;
      ch.jobs[j].used = 1;  // the search read it (and the commit will)
      for (int tb = 0; tb < 2; tb++) {
        const int len = ch.jobs[j].grant.tb[tb].tbs / 8;
        if (cfg.sniffer_mode == 1) {  // run_rar_decode parses pdsch_res->payload (TB 0) at the first CRC-ok TB and returns
          if (ch.jobs[j].crc[tb]) { unpackRar(ch.h_payload.data() + ch.jobs[j].payload_off[0], len, true); break; }
        } else if (ch.jobs[j].crc[tb] && len > 0) {
          unpackRar(ch.h_payload.data() + ch.jobs[j].payload_off[tb], len, true);
        }
      }
    }
  }
  search->setCandMiss(nullptr, nullptr);
  publishPruneSnapshot();   // the blind decoder of the chunks that are launched from now on predicts the search with this state (k_viterbi)
}

// MAC RAR PDU (TS 36.321 6.1.5) walked like srsran::rar_pdu; DL_Sniffer_PDSCH.cc:782-797
void Engine::unpackRar(const uint8_t* p, int len, bool at_search)
{
  RarEntry r[32];
  const int n = rar_parse(cell, p, len, r, 32);
  for (int i = 0; i < n; i++) {
    if (at_search) {  // search thread owns the RNTI manager
      search->rntiManager().activateAndRefresh(r[i].t_crnti, 0, RM_ACT_RAR);
      pred_rar_at[r[i].t_crnti].store((uint32_t)sf_cnt, std::memory_order_relaxed);  // sf_cnt already counts the subframe being searched (+1 form)
    }
    else { mcs_tracking.update_rar_time_crnti(r[i].t_crnti, commit_sf_cnt); publishPrediction(r[i].t_crnti); }  // commit thread owns the MCS tracking
  }
}

// ------------------------------------------------------------------------------------------------ stage C planning
// one srsran_ue_dl_decode_pdsch call = one job; returns -1 when dl_sniffer_config_mimo rejects the grant
int Engine::newJob(Chunk& ch, uint32_t sf, const DlEntry& e, int table, float p_a, int kind)
{
  DecodeJob j;
  j.sf = sf; j.rnti = e.rnti; j.kind = (uint8_t)kind;
  j.risky = kind == 2;  // the speculative second-table attempt
  // pdsch_cfg->p_a: DL mode looks the UE's p-a up before every decode (DL_Sniffer_PDSCH.cc:926-927); the UL-mode decoders never set
  // it and run with the -3 dB of SubframeWorker::set_pdsch_uecfg (SubframeWorker.cc:370)
  j.p_a = cfg.sniffer_mode == 1 ? -3.0f : p_a;
  j.grant = table ? e.grant256 : e.grant64;
  if (dl_sniffer_config_mimo(cell, e.format, e.dci, j.grant) != 0) return -1;
  if (cfg.sniffer_mode == 1) {  // run_decode / run_rar_decode, DL_Sniffer_PDSCH.cc:240-247,694-701
    const uint32_t sfn = ch.ctx[sf].sfn;
    for (auto& tb : j.grant.tb)
      if (tb.enabled && tb.rv < 0) tb.rv = (int)((uint32_t)ceilf(1.5f * (float)((sfn / 2) % 4)) % 4u);
  } else if (e.dci.tb[0].rv < 0 && e.rnti == SIRNTI) j.grant.tb[0].rv = 0;  // DL_Sniffer_PDSCH.cc:891-897
  ch.jobs.push_back(j);
  JobRes jr;
  jr.p_a = j.p_a;
  for (int i = 0; i < 2; i++) { jr.enabled[i] = j.grant.tb[i].enabled ? 1 : 0; jr.len[i] = j.grant.tb[i].tbs / 8; }
  ch.jres.push_back(jr);
  return (int)ch.jobs.size() - 1;
}

template <typename T>
static void grow_dev(T*& p, size_t& cap, size_t need, hipStream_t st)
{
  if (need <= cap) return;
  if (getenv("LSN_HOST_DEBUG")) fprintf(stderr, "grow_dev: %zu -> %zu elements of %zu B\n", cap, need + need / 2 + 1024, sizeof(T));
  HIP_CHECK(hipStreamSynchronize(st));
  if (p) HIP_CHECK(hipFree(p));
  cap = need + need / 2 + 1024;
  HIP_CHECK(hipMalloc((void**)&p, cap * sizeof(T)));
}
template <typename T>
static void grow_host(T*& p, size_t& cap, size_t need, hipStream_t st)
{
  if (need <= cap) return;
  if (getenv("LSN_HOST_DEBUG")) fprintf(stderr, "grow_host: %zu -> %zu elements of %zu B\n", cap, need + need / 2 + 1024, sizeof(T));
  HIP_CHECK(hipStreamSynchronize(st));
  if (p) HIP_CHECK(hipHostFree(p));
  cap = need + need / 2 + 1024;
  HIP_CHECK(hipHostMalloc((void**)&p, cap * sizeof(T), hipHostMallocCoherent | hipHostMallocMapped));
}

void Engine::runJobs(Chunk& ch, JobRunner& r, std::vector<int>& ids)
{
  std::vector<int> todo;
  for (int j : ids)
    if (j >= 0 && !ch.jobs[j].done && !ch.jobs[j].planned) { ch.jobs[j].planned = true; todo.push_back(j); }
  if (todo.empty()) return;
  hipStream_t st = r.stream;
  lsn_perf_t& pf = r.perf;
  const uint32_t nprb = cell.nof_prb;
  r.h_jobs.clear(); r.h_cbs.clear(); r.h_items.clear();
  size_t llr_n = 0, prefix_n = 0;
  const size_t pay0 = ch.h_payload.size();
  size_t pay_n = pay0;
  struct TbRef { int job, tb; uint32_t cb_first, cb_count; };
  std::vector<TbRef> tbrefs;
  std::vector<int> jid_of_hjob;
  for (int jid : todo) {
    DecodeJob& j = ch.jobs[jid];
    const PdschGrant& g = j.grant;
    const SubframeCtx& c = ch.ctx[j.sf];
    LsnGrantDev d{};
    d.sf = j.sf; d.sf_idx = c.sf_idx; d.l0 = c.cfi + (nprb <= 10 ? 1u : 0u);
    for (int s = 0; s < 2; s++)
      for (uint32_t rb = g.prb_lo; rb <= g.prb_hi && rb < nprb; rb++)
        if (g.prb_idx[s][rb]) d.prb_mask[s][rb >> 5] |= 1u << (rb & 31);   // (per slot: the kernels index it with l >= nslot)
    d.nof_re = g.nof_re; d.tx_scheme = (uint32_t)g.tx_scheme; d.pmi = g.pmi; d.nof_layers = g.nof_layers;
    for (int i = 0; i < 2; i++)
      if (g.tb[i].enabled) d.qm[g.tb[i].cw_idx & 1] = (uint32_t)g.tb[i].mod;
    // demodulation possible? (srsran_pdsch_decode preconditions)
    bool demod_ok = (g.tb[0].enabled || g.tb[1].enabled) && g.nof_re > 0;
    if (g.tx_scheme == TXSCHEME_SPATIALMUX || g.tx_scheme == TXSCHEME_CDD) {
      if (cell.nof_ports != 2) demod_ok = false;  // one port: no such transmission; four ports: the reference's srsRAN precodes them for transmit diversity only
      if (g.nof_layers != 1 && dlRx() < 2) demod_ok = false;
    }
    if (g.tx_scheme == TXSCHEME_DIVERSITY && cell.nof_ports < 2) demod_ok = false;
    if (!demod_ok) continue;
    for (int q = 0; q < 2; q++) {
      d.cinit[q] = ((uint32_t)j.rnti << 14) | ((uint32_t)q << 13) | (c.sf_idx << 9) | cell.id;
      d.llr_off[q] = (uint32_t)llr_n;
      if (d.qm[q]) llr_n += ((size_t)g.nof_re * d.qm[q] + 7) & ~(size_t)7;
    }
    d.prefix_off = (uint32_t)prefix_n;
    prefix_n += 14 * nprb + 16;
    // power allocation 36.213 5.2: rho_A from the job's p_a, p_b = 1 (SubframeWorker.cc:372)
    const float rho_a = powf(10.0f, j.p_a / 20.0f);
    const float rho_b = cell.nof_ports == 1 ? rho_a * sqrtf(0.8f) : rho_a;
    d.inv_amp_a = 1.0f / rho_a; d.inv_amp_b = 1.0f / rho_b;
    // transport blocks -> code blocks (36.212 5.1.2, 5.1.4.1.2)
    j.cb_first = (uint32_t)r.h_cbs.size();
    for (int i = 0; i < 2; i++) {
      j.cb_count[i] = 0;
      const GrantTb& tb = g.tb[i];
      if (!(tb.enabled && tb.tbs > 0)) continue;
      CbSegm s;
      const int Qm = tb.mod, G = tb.nof_bits, NL = g.tx_scheme == TXSCHEME_DIVERSITY ? 2 : 1;
      if (!cbsegm(tb.tbs, s) || Qm <= 0 || G <= 0) continue;
      const int Gp = G / (NL * Qm), gamma = Gp % s.C;
      j.payload_off[i] = (uint32_t)pay_n;
      TbRef ref{jid, i, (uint32_t)r.h_cbs.size(), (uint32_t)s.C};
      int rp = 0;
      uint32_t wp = 0;
      for (int q = 0; q < s.C; q++) {
        LsnCbDev cb{};
        const int K = q < s.Cm ? s.Km : s.Kp, F = q == 0 ? s.F : 0;
        int E = (q <= s.C - gamma - 1) ? NL * Qm * (Gp / s.C) : NL * Qm * ((Gp + s.C - 1) / s.C);
        if (rp + E > G) E = G - rp;
        cb.e_off = d.llr_off[tb.cw_idx & 1] + (uint32_t)rp; cb.E = (uint32_t)E; cb.K = (uint32_t)K; cb.F = (uint32_t)F; cb.rv = (uint32_t)tb.rv;
        cb.crc_b = s.C > 1 ? 1u : 0u;
        cb.out_bytes = (uint32_t)(K - F - (s.C > 1 ? 24 : 0)) / 8;
        cb.out_off = (uint32_t)(pay_n - pay0) + wp;
        cb.il_off = turbo_il_offset(K);
        cb.nwin = turbo_nwin(K);
        cb.max_iter = (uint32_t)cfg.max_turbo_iterations;
        // code blocks 1 .. C-1 are launched behind block 0 and skipped when it failed (the TB CRC verdict needs every block)
        cb.dep = (q > 0 && cb_skip) ? (uint32_t)(r.h_cbs.size() - (size_t)q) : LSN_CB_NODEP;
        wp += cb.out_bytes;
        rp += E;
        r.h_cbs.push_back(cb);
      }
      j.cb_count[i] = (uint32_t)s.C;
      pay_n += (wp + 15) & ~15u;
      tbrefs.push_back(ref);
      pf.nof_tb_decodes++;
      if (tbs_from_derived_rows(tb.tbs, g.nof_prb)) pf.nof_tb_on_derived_tbs++;
      pf.nof_cb_decodes += (uint64_t)s.C;
      pf.algo_bytes += 2ull * (uint64_t)tb.nof_bits * 2ull + (uint64_t)tb.tbs / 8ull;
    }
    if (g.prb_lo <= g.prb_hi)
      for (uint32_t grp = g.prb_lo / 16; grp <= std::min<uint32_t>(g.prb_hi, nprb - 1) / 16; grp++) r.h_items.push_back(((uint32_t)r.h_jobs.size() << 8) | grp);
    r.h_jobs.push_back(d);
    jid_of_hjob.push_back(jid);
  }
  // ONE decoder launch per phase (late round 4: + 5.7 % against one launch per wavefront class).  Round 5: the blocks of at most 64 windows - one working
  // wavefront - share workgroups two by two (k_turbo, stage_c.hip) instead of holding a whole LDS / wavefront slot each with the second wavefront idle:
  // half of the metric's code blocks, half of the decoder's slot time (every block alone in its workgroup, rounds 2-4: - 3 %, profiles/r05_ab_session*.txt).
  auto pairable = [&](uint32_t K) { return K <= LSN_TURBO_PAIR_KMAX && turbo_nwin((int)K) <= 64; };
  const uint32_t njobs = (uint32_t)r.h_jobs.size(), ncb = (uint32_t)r.h_cbs.size();
  uint32_t kmax_solo = 0, kmax_pair = 0, emax = 0, nsolo[2] = {0, 0}, npair[2] = {0, 0};
  size_t spp_n = 0;
  std::vector<uint32_t> order;
  if (njobs) {
    grow_dev(r.d_jobs, r.jobs_cap, njobs, st);
    grow_dev(r.d_cbs, r.cbs_cap, ncb, st);
    grow_dev(r.d_cbres, r.cbres_cap, ncb, st);
    grow_dev(r.d_prefix, r.prefix_cap, prefix_n, st);
    grow_dev(r.d_llr16, r.llr16_cap, llr_n + 8, st);
    grow_dev(r.d_payload, r.payload_cap, pay_n - pay0 + 16, st);
    grow_host(r.h_cbres_pinned, r.h_cbres_cap, ncb, st);
    grow_host(r.h_payload_pinned, r.h_payload_cap, pay_n - pay0 + 16, st);
    grow_host(r.h_jobs_pinned, r.h_jobs_cap, njobs, st);
    grow_host(r.h_cbs_pinned, r.h_cbs_cap, ncb, st);
    const uint32_t nitems = (uint32_t)r.h_items.size();
    grow_dev(r.d_items, r.items_cap, nitems + 1, st);
    grow_host(r.h_items_pinned, r.h_items_cap, nitems + 1, st);
    LsnCopySegs up;   // items, jobs and code-block descriptors go up in one launch
    std::memcpy(r.h_items_pinned, r.h_items.data(), nitems * sizeof(uint32_t));
    up.add(r.d_items, r.h_items_pinned, nitems * sizeof(uint32_t));
    std::memcpy(r.h_jobs_pinned, r.h_jobs.data(), njobs * sizeof(LsnGrantDev));
    if (ncb) {
      // launch order: per phase the blocks that get a workgroup of their own first, then the blocks that share one; each class by descending size (longest
      // jobs first; the two blocks of a pair are neighbours in size, so their wavefronts run for about the same time)
      // two phases: first every block that nothing depends on having passed (block 0 of each transport block), then the dependants.  The key (phase, solo before
      // paired, K descending, index ascending) has 2 x 6 144 values: a counting sort - the comparison sort this replaces cost 1.5 us of decode-thread CPU per
      // subframe (round 6, thread-CPU sections)
      order.resize(ncb);
      {
        static thread_local std::vector<uint32_t> cnt;
        cnt.assign(2 * 2 * 6145 + 1, 0);
        auto key = [&](const LsnCbDev& q) { return ((q.dep != LSN_CB_NODEP ? 1u : 0u) * 2u + (pairable(q.K) ? 1u : 0u)) * 6145u + (6144u - std::min<uint32_t>(q.K, 6144u)); };
        for (uint32_t i = 0; i < ncb; i++) { r.h_cbs[i].res_idx = i; cnt[key(r.h_cbs[i]) + 1]++; }
        for (size_t k = 1; k < cnt.size(); k++) cnt[k] += cnt[k - 1];
        for (uint32_t i = 0; i < ncb; i++) order[cnt[key(r.h_cbs[i])]++] = i;   // stable: equal keys keep ascending index
      }
      for (uint32_t i = 0; i < ncb; i++) {
        LsnCbDev q = r.h_cbs[order[i]];
        q.spp_off = (uint32_t)spp_n; spp_n += LSN_SPP_WORDS(q.K);
        emax = std::max(emax, q.E);
        r.h_cbs_pinned[i] = q;
        const int ph = q.dep != LSN_CB_NODEP ? 1 : 0;
        if (pairable(q.K)) { npair[ph]++; kmax_pair = std::max(kmax_pair, q.K); } else { nsolo[ph]++; kmax_solo = std::max(kmax_solo, q.K); }
      }
      grow_dev(r.d_spp, r.spp_cap, spp_n + 16, st);
      up.add(r.d_cbs, r.h_cbs_pinned, ncb * sizeof(LsnCbDev));
    }
    const bool tk = timing_period && (r.launches++ % timing_period) == 0;   // prep / demod / rm: timed on a sample of the launches; the decoders on every launch
    hipStream_t sl = st;
    if (tk) HIP_CHECK(hipEventRecord(r.ev[0], sl));  // (no clear of the LLR arena: k_pdsch_demod writes every soft bit of every codeword it is given, zeros of unpaired SFBC REs included)
    lsn_launch_pdsch_prep_up(cd, r.h_jobs_pinned, r.d_jobs, njobs, up, r.d_prefix, sl);   // descriptors up + prefix tables in one launch
    if (tk) HIP_CHECK(hipEventRecord(r.ev[1], sl));
    lsn_launch_pdsch_demod(cd, r.d_jobs, r.d_items, nitems, r.d_prefix, ch.d_grid, ch.d_ce, ch.d_chest, r.d_llr16, sl);
    if (tk) HIP_CHECK(hipEventRecord(r.ev[2], sl));
    if (ncb) {
      lsn_launch_rm(r.d_cbs, r.d_llr16, r.d_spp, ncb, emax, sl);
      if (timing_period) HIP_CHECK(hipEventRecord(r.ev[5], sl));
      // phase 0: the independent blocks [solo | paired], phase 1: the same of the dependants (descriptor order = launch order)
      lsn_launch_turbo_packed(cd, r.d_cbs, r.d_spp, r.d_payload, r.d_cbres, nsolo[0], kmax_solo, npair[0], kmax_pair, st);
      const uint32_t o1 = nsolo[0] + npair[0];
      if (ncb > o1) {
        if (timing_period) HIP_CHECK(hipEventRecord(r.ev[8], st));
        lsn_launch_turbo_packed(cd, r.d_cbs + o1, r.d_spp, r.d_payload, r.d_cbres, nsolo[1], kmax_solo, npair[1], kmax_pair, st);
      }
      if (timing_period) HIP_CHECK(hipEventRecord(r.ev[3], st));
      {
        LsnCopySegs dn;
        dn.add(r.h_cbres_pinned, r.d_cbres, ncb * sizeof(LsnCbRes));
        dn.add(r.h_payload_pinned, r.d_payload, pay_n - pay0);
        lsn_launch_copy_multi(dn, true, st);
      }
      if (cfg.harq_mode) {  // the soft data of this launch stays with the chunk until its commit (HARQ buffers are filled / combined there)
        if (ch.keep_n + spp_n >= ((size_t)1 << 30)) throw std::runtime_error("harq_mode: the soft data of one chunk exceeds 4 GB (2^30 words) - process smaller batches");   // (HarqKeep::loc: 30-bit word offsets)
        if (ch.keep_n + spp_n > ch.keep_cap) {
          const size_t cap = (ch.keep_n + spp_n) * 2 + (1u << 20);
          uint32_t* nb = nullptr;
          HIP_CHECK(hipMalloc((void**)&nb, cap * sizeof(uint32_t)));
          if (ch.keep_n) HIP_CHECK(hipMemcpy(nb, ch.d_keep, ch.keep_n * sizeof(uint32_t), hipMemcpyDeviceToDevice));
          if (ch.d_keep) HIP_CHECK(hipFree(ch.d_keep));
          ch.d_keep = nb; ch.keep_cap = cap;
        }
        HIP_CHECK(hipMemcpyAsync(ch.d_keep + ch.keep_n, r.d_spp, spp_n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        std::vector<uint32_t> spp_of(ncb, 0);
        for (uint32_t i = 0; i < ncb; i++) spp_of[order[i]] = r.h_cbs_pinned[i].spp_off;
        for (auto& tr : tbrefs) {
          DecodeJob& j = ch.jobs[tr.job];
          j.keep_first[tr.tb] = (uint32_t)ch.keep_cbs.size(); j.keep_count[tr.tb] = tr.cb_count;
          for (uint32_t q = 0; q < tr.cb_count; q++) {
            LsnCbDev cb = r.h_cbs[tr.cb_first + q];
            cb.spp_off = (uint32_t)(ch.keep_n + spp_of[tr.cb_first + q]);
            ch.keep_cbs.push_back(cb);
          }
        }
        ch.keep_n += spp_n;
      }
    }
    HIP_CHECK(hipEventRecord(r.ev_done, st));
    waitEvent(r.ev_done);
    float ms = 0;
    const float scale = (float)timing_period;
    if (tk && hipEventElapsedTime(&ms, r.ev[0], r.ev[1]) == hipSuccess) pf.kernel_ms[LSN_K_PDSCH_PREP] += ms * scale;
    if (tk && hipEventElapsedTime(&ms, r.ev[1], r.ev[2]) == hipSuccess) pf.kernel_ms[LSN_K_PDSCH_DEMOD] += ms * scale;
    pf.kernel_launches[LSN_K_PDSCH_PREP]++; pf.kernel_launches[LSN_K_PDSCH_DEMOD]++;
    if (ncb) {
      if (tk && hipEventElapsedTime(&ms, r.ev[2], r.ev[5]) == hipSuccess) pf.kernel_ms[LSN_K_RM] += ms * scale;
      pf.kernel_launches[LSN_K_RM]++;
      {
        const bool ph1 = ncb > nsolo[0] + npair[0];
        auto acc = [&](int k, hipEvent_t a, hipEvent_t b) { if (timing_period && hipEventElapsedTime(&ms, a, b) == hipSuccess) { pf.kernel_ms[k] += ms; pf.kernel_launches[k]++; } };
        acc(LSN_K_TURBO128, r.ev[5], ph1 ? r.ev[8] : r.ev[3]);
        if (ph1) acc(LSN_K_TURBO128, r.ev[8], r.ev[3]);
      }
      // algorithmic bytes of the decoder kernels: every code block reads its K + 12 packed soft words (k_rm's output) and writes its payload
      for (uint32_t i = 0; i < ncb; i++) {
        const uint64_t b = 4ull * (r.h_cbs_pinned[i].K + 12u) + r.h_cbs_pinned[i].out_bytes;
        pf.turbo_algo_bytes += b;
        pf.turbo128_algo_bytes += b;   // (every downlink block runs in the two-wavefront instance)
      }
    }
    if (keep_stage_c.load()) {  // parity taps: the arenas are recycled by the next launch of this runner
      std::vector<int16_t> llr(llr_n);
      std::vector<uint32_t> spp(spp_n);
      if (llr_n) HIP_CHECK(hipMemcpy(llr.data(), r.d_llr16, llr_n * sizeof(int16_t), hipMemcpyDeviceToHost));
      if (spp_n) HIP_CHECK(hipMemcpy(spp.data(), r.d_spp, spp_n * sizeof(uint32_t), hipMemcpyDeviceToHost));
      std::vector<uint32_t> spp_of(ncb, 0);
      for (uint32_t i = 0; i < ncb; i++) spp_of[order[i]] = r.h_cbs_pinned[i].spp_off;
      std::lock_guard<std::mutex> lk(tap_mtx);
      if (ch.tapjobs.size() < ch.jobs.size()) ch.tapjobs.resize(ch.jobs.size());
      for (uint32_t h = 0; h < njobs; h++) {
        TapJob& t = ch.tapjobs[jid_of_hjob[h]];
        t = TapJob{};
        t.have = true; t.d = r.h_jobs[h];
        for (int q = 0; q < 2; q++)
          if (t.d.qm[q]) t.llr[q].assign(llr.begin() + t.d.llr_off[q], llr.begin() + t.d.llr_off[q] + (size_t)t.d.nof_re * t.d.qm[q]);
      }
      for (auto& tr : tbrefs)
        for (uint32_t q = 0; q < tr.cb_count; q++) {
          TapCb c;
          c.cb = r.h_cbs[tr.cb_first + q]; c.tb = (uint32_t)tr.tb; c.res = r.h_cbres_pinned[tr.cb_first + q];
          c.words.assign(spp.begin() + spp_of[tr.cb_first + q], spp.begin() + spp_of[tr.cb_first + q] + c.cb.K + 12);
          ch.tapjobs[tr.job].cbs.push_back(std::move(c));
        }
    }
    ch.h_payload.resize(pay_n);
    if (pay_n > pay0) std::memcpy(ch.h_payload.data() + pay0, r.h_payload_pinned, pay_n - pay0);
    // transport-block verdicts: every code block ok, CRC24A over data||parity zero (combined from the per-block
    // remainders), parity word non-zero
    for (auto& t : tbrefs) {
      DecodeJob& j = ch.jobs[t.job];
      bool all_ok = true;
      uint32_t rem = 0;
      uint64_t bits_after = 0;
      uint32_t shift = 1;   // x^bits_after mod g, carried along: one or two multiplications per block instead of a modular power
      for (int q = (int)t.cb_count - 1; q >= 0; q--) {
        const LsnCbRes& cr = r.h_cbres_pinned[t.cb_first + q];
        all_ok = all_ok && cr.ok != 0;
        j.iters += cr.iters;
        pf.nof_turbo_iterations += cr.iters; pf.nof_turbo_iterations_run += cr.iters_run;
        pf.turbo_cyc_rm += cr.cyc_rm; pf.turbo_cyc_map += cr.cyc_map; pf.turbo_cyc_out += cr.cyc_out;
        rem ^= bits_after ? crc24a_mulmod(cr.rem_a, shift) : (cr.rem_a & 0xFFFFFFu);
        const uint32_t nb = r.h_cbs[t.cb_first + q].out_bytes;
        bits_after += 8ull * nb;
        if (q > 0) shift = crc24a_mulmod(shift, crc24a_xpow_bytes(nb));
      }
      const int tbs = j.grant.tb[t.tb].tbs;
      const uint8_t* pl = ch.h_payload.data() + j.payload_off[t.tb];
      const uint32_t par = ((uint32_t)pl[tbs / 8] << 16) | ((uint32_t)pl[tbs / 8 + 1] << 8) | pl[tbs / 8 + 2];
      j.crc[t.tb] = all_ok && rem == 0 && par != 0 && bits_after == (uint64_t)tbs + 24;
      if (cfg.harq_mode) {  // per-block verdicts of this transmission, next to the kept soft data (harqStore / harqCombinedDecode)
        if (ch.keep_res.size() < ch.keep_cbs.size()) ch.keep_res.resize(ch.keep_cbs.size());
        for (uint32_t q = 0; q < t.cb_count && j.keep_count[t.tb] == t.cb_count; q++) ch.keep_res[j.keep_first[t.tb] + q] = r.h_cbres_pinned[t.cb_first + q];
      }
      JobRes& jr = ch.jres[t.job];
      jr.crc[t.tb] = j.crc[t.tb] ? 1 : 0;
      jr.payload_off[t.tb] = j.payload_off[t.tb];
      // DL_Sniffer_PDSCH.cc:1041-1070: a decoded C-RNTI block is walked for RRCConnectionSetups - here, by the thread that ran the decode,
      // so that the commit thread never touches the payload
      if (j.crc[t.tb] && tbs >= 8 && cfg.sniffer_mode == 0 && rnti_name(j.rnti)[0] == 'C') {
        UeSpecConfig sc[20];
        const int n = MCSTracking::setups_of_pdu(pl, tbs / 8, sc, 20, true);   // every SDU: which of them count is the commit's decision (known-table branch: LCID 0 only)
        if (n > 0) {
          jr.setup_first[t.tb] = (uint32_t)ch.setup_cfgs.size();
          jr.nsetup[t.tb] = (uint8_t)n;
          ch.setup_cfgs.insert(ch.setup_cfgs.end(), sc, sc + n);
        }
      }
    }
  }
  for (int jid : todo) { ch.jobs[jid].done = true; ch.jres[jid].done = 1; }
  pf.nof_decode_jobs += todo.size();
}

void Engine::ensureJob(Chunk& ch, JobRunner& r, int j)
{
  if (j < 0 || ch.jobs[j].done) return;
  std::vector<int> one{j};
  ch.jobs[j].planned = false;
  runJobs(ch, r, one);
}

// the two tables give the same decode (low MCS indices map to the same modulation and transport block size in both): one job serves both attempts
static bool same_decode(const PdschGrant& a, const PdschGrant& b)
{
  if (a.nof_tb != b.nof_tb || a.nof_re != b.nof_re || a.tx_scheme != b.tx_scheme || a.pmi != b.pmi || a.nof_layers != b.nof_layers) return false;
  for (int i = 0; i < 2; i++) {
    const GrantTb &x = a.tb[i], &y = b.tb[i];
    if (x.enabled != y.enabled) return false;
    if (x.enabled && (x.mod != y.mod || x.tbs != y.tbs || x.rv != y.rv || x.nof_bits != y.nof_bits || x.cw_idx != y.cw_idx)) return false;
  }
  return true;
}
// The 256QAM-table attempt of an unknown-table DCI (format > 1A) is run ahead even when its 64QAM-table attempt passed a CRC: the plan runs
// thousands of subframes ahead of the commit, and a UE whose table the commit learns in between is then committed with the KNOWN table, which
// wants exactly that attempt.  Measured in round 4 (LSN_SPECULATE_SECOND_TABLE=0): without them 6 % of the subframes need a decode inside the
// sequential commit turn (one GPU round trip each) and the rate falls from 156 k to 56 k subframes/s; with them the engine runs 7 % more
// turbo iterations.  Results are the same either way (the commit re-derives every decision).
static const bool g_speculate_second_table = true;

static const bool g_hints = true;
int Engine::hintEvents(uint16_t rnti, uint32_t pos, uint32_t* lo_out) const
{
  if (!g_hints || cfg.mcs_tracking_mode != 1) return -1;
  // the hints switch themselves off when the commit keeps asking for the attempts they left out (each costs a GPU round trip in the sequential turn)
  const uint64_t used = sh->hint_used.load(std::memory_order_relaxed), missed = sh->hint_missed.load(std::memory_order_relaxed);
  if (missed * 20 > used + 200) return -1;
  // events count from the last reset of the RNTI's entry: the database ageing in front of this DCI, a RAR naming the RNTI, an update by hand
  uint32_t lo = mcs_update_period ? (pos / mcs_update_period) * mcs_update_period : 0u;
  lo = std::max(lo, pred_rar_at[rnti].load(std::memory_order_relaxed));   // (position + 1 of the RAR subframe = first position after it)
  lo = std::max(lo, sh->hint_floor.load(std::memory_order_relaxed));
  if (lo_out) *lo_out = lo;
  int n = 0;
  const std::atomic<uint32_t>* ring = &sh->hint_pos[(size_t)rnti * SharedSeq::HINT_RING];
  for (int i = 0; i < SharedSeq::HINT_RING; i++) {
    const uint32_t q = ring[i].load(std::memory_order_relaxed);  // position + 1
    if (q && q - 1 >= lo && q - 1 < pos) n++;
  }
  return n;
}
bool Engine::hintedTable256(uint16_t rnti, uint32_t pos) const { return hintEvents(rnti, pos, nullptr) >= SharedSeq::HINT_EVENTS; }
void Engine::hintEvent(uint16_t rnti, uint32_t pos)
{
  const uint8_t k = sh->hint_next[rnti].fetch_add(1, std::memory_order_relaxed);
  sh->hint_pos[(size_t)rnti * SharedSeq::HINT_RING + (k % SharedSeq::HINT_RING)].store(pos + 1, std::memory_order_relaxed);
}

// wave 1: the first decode the reference would attempt for every accepted DL DCI, predicted from the MCS-tracking
// state as of now; wave 2: the 256QAM-table retry of "unknown table" grants whose first attempt failed on both TBs
void Engine::planJobs(Chunk& ch, JobRunner& r)
{
  std::vector<int> wave;
  struct Pending { uint32_t sf; size_t di; bool always; };
  std::vector<Pending> retry, deferred;
  // In-chunk learning (round 4): a UE whose table the plan does not know yet gets its first HINT_EVENTS teachable grants of the chunk decoded the
  // reference's way (64QAM table, then 256QAM table) - the rest wait for those verdicts in a third wave.  Before, all grants of a new 256QAM UE in
  // the chunks in flight (about 200 per UE) ran a hopeless 64QAM-table attempt the commit never read: 19 % of all turbo iterations.
  static const bool g_defer = true;
  std::vector<std::pair<uint16_t, uint8_t>> seen;   // (rnti, teachable unknown-table grants so far in this chunk): a handful of UEs
  ch.ul_epoch = ul_cfg_epoch.load(std::memory_order_acquire);
  for (uint32_t sf = 0; sf < ch.nsf; sf++) search->finishSubframe(ch.ctx[sf]);  // DCI unpack, grants and collision statistics of every accepted DCI (deferred from the sequential search)
  for (uint32_t sf = 0; sf < ch.nsf; sf++) {
    SubframeCtx& c = ch.ctx[sf];
    if (!c.searched) continue;
    for (size_t di = 0; di < c.dl.size(); di++) {
      DlEntry& e = c.dl[di];
      if (!e.unpack_ok) continue;
      if (cfg.sniffer_mode == 1) {  // decode_ul_mode: RAR + format 1 / 1A (not SI) with the 64QAM table only
        if (!(ul_set ? ulModeDecodesDl(e) : e.rnti == SIRNTI)) continue;  // before the SIB2 configuration: decode_SIB (a prediction, commit decides)
        if (e.job[0] < 0) e.job[0] = newJob(ch, sf, e, 0);
        if (e.job[0] >= 0) wave.push_back(e.job[0]);
        continue;
      }
      McsTable table;
      if (cfg.mcs_tracking_mode == 1)
        table = (e.rnti == SIRNTI || e.rnti == PRNTI || rnti_israr(e.rnti) || e.format == FORMAT1A) ? TABLE_64QAM : predictedTable(e.rnti);
      else
        table = cfg.mcs_tracking_mode == 2 ? TABLE_UNKNOWN : TABLE_64QAM;
      e.hinted = false;
      if (table >= TABLE_UNKNOWN && e.format > FORMAT1A && e.ok64 && e.ok256 && e.job[0] < 0 && e.job[1] < 0 && hintedTable256(e.rnti, ch.gpos0 + sf)) {
        table = TABLE_256QAM;   // the commit will have learnt it by then: no 64QAM-table attempt
        e.hinted = true;
        sh->hint_used.fetch_add(1, std::memory_order_relaxed);
      }
      const int first = table == TABLE_256QAM ? 1 : 0;
      const PdschGrant& g = first ? e.grant256 : e.grant64;
      const bool ok = first ? e.ok256 : e.ok64;
      if (!ok || !(g.tb[0].tbs > 0)) continue;
      if (dlRx() == 1 && (e.grant64.nof_tb == 2 || e.grant256.nof_tb == 2)) continue;
      if (g_defer && g_hints && cfg.mcs_tracking_mode == 1 && table >= TABLE_UNKNOWN && e.format > FORMAT1A && e.ok64 && e.ok256 && e.job[0] < 0 && e.job[1] < 0 &&
          e.grant256.tb[0].tbs > 0 && !same_decode(e.grant64, e.grant256)) {
        size_t k = 0;
        while (k < seen.size() && seen[k].first != e.rnti) k++;
        if (k == seen.size()) seen.push_back({e.rnti, (uint8_t)0});
        if (seen[k].second >= SharedSeq::HINT_EVENTS) { deferred.push_back({sf, di, false}); continue; }
        seen[k].second++;
      }
      if (e.job[first] < 0) {
        e.job[first] = newJob(ch, sf, e, first, predictedPa(e.rnti));  // as of now; commit checks it
        if (e.job[first] >= 0 && table >= TABLE_UNKNOWN && e.format > FORMAT1A) ch.jobs[e.job[first]].risky = true;
      }
      if (e.job[first] >= 0) wave.push_back(e.job[first]);
      if (e.job[first] >= 0 && e.job[1 - first] < 0 && e.ok64 && e.ok256 && same_decode(e.grant64, e.grant256)) e.job[1 - first] = e.job[first];
      // the 256QAM-table attempt: the reference makes it when both TBs failed with the 64QAM table.  A DCI of a format that can teach the
      // table (> 1A, DL_Sniffer_PDSCH.cc:1168-1171) may find its RNTI's table KNOWN by the time it is committed (the plan runs thousands of
      // subframes ahead of the commit while a new UE is being learned), and then commit wants exactly that attempt: decode it now rather
      // than as a GPU round trip of the sequential commit thread
      if (table >= TABLE_UNKNOWN && e.ok256 && e.job[1] < 0) retry.push_back({sf, di, g_speculate_second_table && e.format > FORMAT1A});
    }
  }
  const uint8_t trk = (uint8_t)(2 + (&r - runner_c));
  trace(trk, TR_W1_LAUNCHED, ch.trace_id);  // host-side planning done
  runJobs(ch, r, wave);
  trace(trk, TR_W1_DONE, ch.trace_id);
  wave.clear();
  for (auto& p : retry) {
    DlEntry& e = ch.ctx[p.sf].dl[p.di];
    if (e.job[0] < 0 || !ch.jobs[e.job[0]].done) continue;
    if (!p.always && (ch.jobs[e.job[0]].crc[0] || ch.jobs[e.job[0]].crc[1])) continue;
    const bool spec = ch.jobs[e.job[0]].crc[0] || ch.jobs[e.job[0]].crc[1];
    if (spec) {
      // The speculative attempt serves a commit that finds the RNTI's table KNOWN as 256QAM.  A UE whose 64QAM-table attempt passed the CRC of
      // EVERY enabled transport block is on the 64QAM table: nothing can have taught the commit otherwise, so the attempt (hopeless by
      // construction: 12 iterations per block) is left out - measured in round 4: 3 326 speculative jobs per 6 400 subframes, 280 of them used,
      // 15 % of all turbo iterations.  Kept: partial passes (one block of a two-block grant whose MCS index means the same in both tables).
      const DecodeJob& j0 = ch.jobs[e.job[0]];
      bool all_ok = true;
      for (int i = 0; i < 2; i++) all_ok = all_ok && (!j0.grant.tb[i].enabled || !(j0.grant.tb[i].tbs > 0) || j0.crc[i]);
      if (all_ok) continue;
    }
    if (spec) r.perf.nof_speculative_jobs++;
    if (e.job[1] < 0) e.job[1] = newJob(ch, p.sf, e, 1, ch.jobs[e.job[0]].p_a, spec ? 2 : 1);
    if (e.job[1] >= 0) wave.push_back(e.job[1]);
  }
  if (!deferred.empty()) {
    // the grants held back above join the second wave.  Evidence = the 64QAM-table verdicts of THIS chunk's first wave in front of the grant + the ring
    // (a third wave that waits for the 256QAM-table verdicts as well was measured: 11 % fewer turbo iterations and 6 % FEWER subframes/s - one more
    // round of launches per chunk costs more than the hopeless attempts it saves)
    struct Ev { uint16_t rnti; uint32_t pos; bool fail64, full64; };  // fail64: no block passed with the 64QAM table; full64: every enabled block passed
    std::vector<Ev> ev;
    for (auto& p : retry) {
      const DlEntry& e = ch.ctx[p.sf].dl[p.di];
      if (!(e.format > FORMAT1A) || e.job[0] < 0 || !ch.jobs[e.job[0]].done || same_decode(e.grant64, e.grant256)) continue;
      const DecodeJob& j0 = ch.jobs[e.job[0]];
      bool full64 = true;
      for (int i = 0; i < 2; i++) full64 = full64 && (!j0.grant.tb[i].enabled || !(j0.grant.tb[i].tbs > 0) || j0.crc[i]);
      ev.push_back({e.rnti, ch.gpos0 + p.sf, !j0.crc[0] && !j0.crc[1], full64});
    }
    for (auto& p : deferred) {
      DlEntry& e = ch.ctx[p.sf].dl[p.di];
      const uint32_t pos = ch.gpos0 + p.sf;
      uint32_t lo = 0;
      const int ring = hintEvents(e.rnti, pos, &lo);
      int nfail = 0, n64 = 0, nother = 0;
      for (auto& x : ev)
        if (x.rnti == e.rnti && x.pos < pos && x.pos >= lo) { nfail += x.fail64 ? 1 : 0; n64 += x.full64 ? 1 : 0; nother += (!x.fail64 && !x.full64) ? 1 : 0; }
      const float p_a = predictedPa(e.rnti);
      if (ring >= 0 && (ring >= SharedSeq::HINT_EVENTS || (nfail >= SharedSeq::HINT_EVENTS && n64 == 0 && nother == 0))) {
        // the 256QAM-table attempt only: the ring says the commit knows the table, or every one of this UE's HINT_EVENTS grants in front failed with
        // the 64QAM table (their 256QAM-table attempts run in this same wave: a UE that fails those too makes the commit ask for the attempt left
        // out here, which counts as a miss and closes the hints)
        e.hinted = true;
        sh->hint_used.fetch_add(1, std::memory_order_relaxed);
        if (e.job[1] < 0) e.job[1] = newJob(ch, p.sf, e, 1, p_a);
        if (e.job[1] >= 0) wave.push_back(e.job[1]);
        continue;
      }
      // otherwise the reference's own order: its 64QAM-table attempt and, unless everything of this UE passed with that table so far, the
      // 256QAM-table attempt with it
      if (e.job[0] < 0) e.job[0] = newJob(ch, p.sf, e, 0, p_a);
      if (e.job[0] >= 0) wave.push_back(e.job[0]);
      if (!(n64 > 0 && nfail == 0 && nother == 0) && e.job[1] < 0) {
        e.job[1] = newJob(ch, p.sf, e, 1, p_a, 2);
        if (e.job[1] >= 0) { wave.push_back(e.job[1]); r.perf.nof_speculative_jobs++; }
      }
    }
  }
  runJobs(ch, r, wave);
  // teaching decodes of this chunk (SharedSeq::hint_pos): 64QAM-table attempt failed on every block, 256QAM-table attempt passed with a learnable MCS index
  if (cfg.mcs_tracking_mode == 1)
    for (auto& p : retry) {
      const DlEntry& e = ch.ctx[p.sf].dl[p.di];
      if (!(e.format > FORMAT1A) || e.job[0] < 0 || e.job[1] < 0 || e.job[0] == e.job[1]) continue;
      const DecodeJob &j0 = ch.jobs[e.job[0]], &j1 = ch.jobs[e.job[1]];
      if (!j0.done || !j1.done || j0.crc[0] || j0.crc[1]) continue;
      bool teach = false;
      for (int i = 0; i < 2; i++) teach = teach || (j1.crc[i] && e.dci.tb[i].mcs_idx > 0 && e.dci.tb[i].mcs_idx < 28);
      if (teach) hintEvent(e.rnti, ch.gpos0 + p.sf);
    }
  buildCommitView(ch);
}

void Engine::buildCommitView(Chunk& ch)
{
  ch.cdci.clear();
  ch.cdci_first.assign(ch.nsf + 1, 0);
  for (uint32_t sf = 0; sf < ch.nsf; sf++) {
    ch.cdci_first[sf] = (uint32_t)ch.cdci.size();
    const SubframeCtx& c = ch.ctx[sf];
    if (!c.searched) continue;
    for (size_t di = 0; di < c.dl.size(); di++) {
      const DlEntry& e = c.dl[di];
      CommitDci d;
      d.rnti = e.rnti; d.format = (uint8_t)e.format; d.di = (uint32_t)di;
      d.flags = (uint8_t)((e.unpack_ok ? 1 : 0) | (e.ok64 ? 2 : 0) | (e.ok256 ? 4 : 0) | (e.grant64.nof_tb == 2 ? 8 : 0) | (e.grant256.nof_tb == 2 ? 16 : 0));
      for (int i = 0; i < 2; i++) {
        if (e.grant64.tb[i].enabled) d.en64 |= (uint8_t)(1u << i);
        if (e.grant256.tb[i].enabled) d.en256 |= (uint8_t)(1u << i);
        d.mcs_idx[i] = (uint8_t)e.dci.tb[i].mcs_idx;
        d.job[i] = e.job[i];
      }
      d.tbs0_64 = e.grant64.tb[0].tbs; d.tbs0_256 = e.grant256.tb[0].tbs;
      ch.cdci.push_back(d);
    }
  }
  ch.cdci_first[ch.nsf] = (uint32_t)ch.cdci.size();
}

// ------------------------------------------------------------------------------------------------ commit
void Engine::emitPdu(Chunk& ch, JobRunner& r, const char* name, size_t payload_off, uint32_t len, uint16_t rnti, uint32_t tti, uint8_t tb)
{
  r.perf.nof_pdus++;
  if (!sink && api_mode < 0) return;  // records feed the PDU sink and / or the security-API sinks
  lsn_pdu_ctx_t c{};
  c.tti = tti; c.direction = 1; c.crc_ok = 1; c.is_retx = 0; c.tb = tb;
  // LTESniffer_pcap_writer::write_dl_* (PcapWriter.cc:162-190)
  if (name[0] == 'S') { c.rnti = SIRNTI; c.rnti_type = 4; }
  else if (name[0] == 'P') { c.rnti = PRNTI; c.rnti_type = 1; }
  else if (name[0] == 'R') { c.rnti = rnti; c.rnti_type = 2; }
  else { c.rnti = rnti; c.rnti_type = 3; }
  ch.recs.push_back({c, payload_off, len, 0});  // written by the writer thread, in commit order
}

// a decoded C-RNTI transport block: RRCConnectionSetup -> UE configuration database (commit thread)
void Engine::learnUeConfig(const uint8_t* pdu, int len, uint16_t rnti)
{
  if (mcs_tracking.learn_from_pdu(pdu, len, rnti, commit_sf_cnt)) default_p_a.store(mcs_tracking.default_p_a(), std::memory_order_relaxed);
}

// what the decode threads may read of the tracking database while they plan a chunk (a prediction: commit re-derives everything)
void Engine::publishPrediction(uint16_t rnti)
{
  pred_table[rnti].store(mcs_tracking.present(rnti) ? (uint8_t)mcs_tracking.peek(rnti) : (uint8_t)0xFF, std::memory_order_relaxed);
  pred_p_a[rnti].store(mcs_tracking.get_ue_config_rnti(rnti).p_a, std::memory_order_relaxed);
}

// MCSTracking::update_database_dl as LTESniffer_Core drives it (LTESniffer_Core.cc:473-499): every get_interval() x 1000 subframes
void Engine::ageTrackingDatabase()
{
  std::vector<uint16_t> changed;
  mcs_tracking.update_database_dl(commit_sf_cnt, &changed);
  for (uint16_t r : changed) publishPrediction(r);
  nof_mcs_db_updates++;
}

// PDSCH_Decoder::decode_dl_mode (DL_Sniffer_PDSCH.cc:881-1291) over the decode results of every subframe of the chunk.
// The loop reads the compact CommitDci / JobRes views (buildCommitView, runJobs); the wide DlEntry / DecodeJob records are only touched
// on the slow path (a decode that has to be created here because the plan-time prediction of table or p-a was wrong).
void Engine::commitChunk(Chunk& ch, JobRunner& r)
{
  std::vector<McsTable> tables;
  if (cfg.harq_mode) {
    // the retransmissions of this chunk, combined and decoded in a few batches ahead of the walk (harqScout): pass p serves the p-th retransmission in a row
    // of the same buffer.  The scratch area is empty here (harqFlush of the previous turn): it may be given a new size
    harq_scratch_n = 0;
    const size_t want = std::min<size_t>(ch.keep_n + 4096, (size_t)(1u << 30) - 1);
    if (want > harq_scratch_cap) grow_dev(d_harq_scratch, harq_scratch_cap, want, r.stream);
    std::vector<HarqReq> reqs;
    for (int pass = 0; pass < 8; pass++) {
      const double t0 = now_ms();
      harqScout(ch, reqs, pass == 0);
      r.perf.ms_harq[0] += now_ms() - t0;
      if (reqs.empty()) break;
      size_t need = 0;
      for (const HarqReq& q : reqs)
        for (uint32_t b = 0; b < q.n; b++)
          if (!q.ok[b]) need += LSN_SPP_WORDS(ch.keep_cbs[ch.jobs[q.job].keep_first[q.tb] + b].K);
      if (harq_scratch_n + need > harq_scratch_cap) break;   // (what does not fit is decoded by the walk itself)
      harqRunBatch(ch, r, reqs);
    }
  }
  for (uint32_t sf = 0; sf < ch.nsf; sf++, commit_sf_cnt++) {
    SubframeCtx& c = ch.ctx[sf];
    // the tracking database is only WRITTEN here (commit thread); decode threads read the published prediction arrays, the API getter
    // (lsn_phy_get_ue_config) takes this lock
    std::unique_lock<std::mutex> mcs_lk(mcs_mtx);
    const uint32_t now = commit_sf_cnt;
    commit_pos.store(commit_sf_cnt, std::memory_order_relaxed);
    if (cfg.mcs_tracking_mode && mcs_update_period && commit_sf_cnt && (commit_sf_cnt % mcs_update_period) == 0) ageTrackingDatabase();
    if (cfg.harq_mode && commit_sf_cnt && (commit_sf_cnt % 10000u) == 0) harq_db.update_database(commit_sf_cnt);  // the 10 s timer, LTESniffer_Core.cc:487-494
    if (!c.searched) continue;
    const uint32_t k0 = ch.cdci_first[sf], k1 = ch.cdci_first[sf + 1];
    // DCICollection.cc:107-134: the table of every DCI of this subframe is fixed before any of them is decoded
    tables.resize(k1 - k0);
    for (uint32_t k = k0; k < k1; k++) {
      const CommitDci& d = ch.cdci[k];
      tables[k - k0] = collection_table(cfg.mcs_tracking_mode, d.rnti, (DciFormat)d.format, mcs_tracking, now);
    }
    // addCandidate looks the table up for EVERY accepted DCI, format 0 included: an uplink grant refreshes the entry's time stamp too
    if (cfg.mcs_tracking_mode == 1)
      for (const UlEntry& u : c.ul)
        if (!(u.rnti == SIRNTI || u.rnti == PRNTI || rnti_israr(u.rnti))) (void)mcs_tracking.find_tracking_info_RNTI_dl(u.rnti, now);
    for (uint32_t k = k0; k < k1; k++) {
      CommitDci& d = ch.cdci[k];
      const McsTable table = tables[k - k0];
      const bool unpack_ok = d.flags & 1, ok64 = d.flags & 2, ok256 = d.flags & 4;
      const TableView tv = table_view(table, d.rnti, unpack_ok, ok64, ok256);  // falcon_dci.c:284-310
      const bool has64 = tv.has64, has256 = tv.has256, dci_rnti_ok = tv.dci_rnti_ok;
      // DCICollection.cc:236-251: a reserved MCS index of a 64QAM-table grant takes its size from the HARQ database (harq_mode only).  The plan knew no size for
      // it (0): whatever it decoded for this entry is dropped and the grant is decoded on demand with the size in
      if (cfg.harq_mode && has64 && collection_last_tbs(true, table, c.dl[d.di], harq_db)) {
        d.tbs0_64 = c.dl[d.di].grant64.tb[0].tbs;
        d.job[0] = -1; c.dl[d.di].job[0] = -1;
      }
      const int cur_t = table == TABLE_256QAM ? 1 : 0;
      const bool cur_has = cur_t ? has256 : has64;
      const int32_t cur_tbs0 = cur_has ? (cur_t ? d.tbs0_256 : d.tbs0_64) : 0;
      const uint8_t cur_en = cur_has ? (cur_t ? d.en256 : d.en64) : 0;
      const bool two_tb = (has64 && (d.flags & 8)) || (has256 && (d.flags & 16));
      const bool gate = (cur_tbs0 > 0 && dci_rnti_ok && !(dlRx() == 1 && two_tb)) || d.rnti == PRNTI;  // :887-889
      if (!gate) continue;
      const char* name = rnti_name(d.rnti);
      // :926-927: the p-a in force when this DCI is decoded.  A job planned (or speculated) with another value - a connection setup
      // was committed in between - is dropped and decoded again, so results do not depend on how far ahead the pipeline planned
      const float p_a_now = mcs_tracking.get_ue_config_rnti(d.rnti).p_a;
      auto run = [&](int t) -> int {
        if (!(t ? has256 : has64)) return -1;
        if (d.job[t] >= 0 && ch.jres[d.job[t]].p_a != p_a_now) d.job[t] = -1;
        if (d.job[t] >= 0 && ch.jres[d.job[t]].done) { r.perf.nof_decode_jobs_used++; ch.jobs[d.job[t]].used = 1; return d.job[t]; }
        DlEntry& e = c.dl[d.di];  // slow path
        if (e.hinted && t == 0 && d.job[0] < 0) sh->hint_missed.fetch_add(1, std::memory_order_relaxed);
        const int why = (e.job[t] >= 0 && d.job[t] < 0) ? 0 : (e.job[t] < 0 ? (e.job[1 - t] >= 0 ? 1 : 2) : 3);
        e.job[t] = d.job[t];
        if (e.job[t] < 0) e.job[t] = newJob(ch, sf, e, t, p_a_now, 4);
        if (e.job[t] >= 0 && !ch.jobs[e.job[t]].done) {
          const double t0 = now_ms();
          static const bool dbg = getenv("LSN_DEBUG_ONDEMAND") != nullptr;
          if (dbg) fprintf(stderr, "ondemand: sf_cnt %u tti %u rnti %u fmt %d table %d t %d job0 %d job1 %d pred %d mcs %u/%u crc0job %d%d\n", commit_sf_cnt, c.tti, d.rnti, (int)d.format, (int)table, t,
                           e.job[0], e.job[1], (int)pred_table[d.rnti].load(), d.mcs_idx[0], d.mcs_idx[1], e.job[1 - t] >= 0 ? (int)ch.jres[e.job[1 - t]].crc[0] : -1, e.job[1 - t] >= 0 ? (int)ch.jres[e.job[1 - t]].crc[1] : -1);
          ensureJob(ch, r, e.job[t]);
          r.perf.nof_ondemand_decodes++; r.perf.nof_ondemand_commit[why]++; r.perf.ms_ondemand_commit += now_ms() - t0;
        }
        d.job[t] = e.job[t];
        if (d.job[t] >= 0) { r.perf.nof_decode_jobs_used++; ch.jobs[d.job[t]].used = 1; }
        return d.job[t];
      };
      // dl_sniffer_config_mimo's verdict 0 / -1 / -2 / -3 for the statistics: a job exists exactly when it was 0 (newJob), so the
      // function itself only runs again for the rare rejected grant
      auto mimo_of = [&](int t, int job) {
        if (job >= 0) return 0;
        const DlEntry& e = c.dl[d.di];
        PdschGrant g = t ? e.grant256 : e.grant64;
        return -dl_sniffer_config_mimo(cell, e.format, e.dci, g);
      };
      auto learn = [&](const JobRes& jr, int tb, bool any_lcid) {  // :1041-1070 / :1133-1160 with the PDU walked ahead of time (runJobs)
        if (jr.nsetup[tb] && mcs_tracking.learn_setups(ch.setup_cfgs.data() + jr.setup_first[tb], jr.nsetup[tb], d.rnti, now, any_lcid))
          default_p_a.store(mcs_tracking.default_p_a(), std::memory_order_relaxed);
      };
      bool crc[2] = {false, false};   // pdsch_res[].crc as the statistics see it at the end of the iteration
      int mimo_ret = 0;
      if (table == TABLE_64QAM || table == TABLE_256QAM) {  // :932-1083
        const int j = run(cur_t);
        mimo_ret = cur_has ? mimo_of(cur_t, j) : -1;
        if (j >= 0) {
          const JobRes jr = ch.jres[j];  // (by value: a combined decode below appends to the chunk's vectors)
          for (int tb = 0; tb < 2; tb++) {
            crc[tb] = jr.crc[tb] != 0;
            uint32_t poff = jr.payload_off[tb];
            bool combined = false;
            if (cfg.harq_mode && name[0] == 'C' && jr.enabled[tb]) {  // :943-1020: new transmission / retransmission / already decoded, per transport block
              const DlEntry& e = c.dl[d.di];
              const int tbs = ch.jobs[j].grant.tb[tb].tbs;
              int ent = -1;
              const HarqRet hr = harq_db.is_retransmission(d.rnti, e.dci.pid, tb, e.dci.tb[tb].ndi != 0, tbs, c.sfn, c.sf_idx, ent);
              const size_t slot = ent < 0 ? 0 : ((size_t)ent * HarqDatabase::NPID + (e.dci.pid & 7u)) * 2 + (size_t)tb;
              if (hr == HARQ_NEW_TX) {
                if (!crc[tb]) harqStore(ch, r, j, tb, slot);   // srsran_softbuffer_rx_reset_tbs + this transmission (the buffer is only read again if the block failed)
              } else if (hr == HARQ_RE_TX) {
                crc[tb] = harqCombinedDecode(ch, r, j, tb, slot, poff);
                combined = true;
              } else if (hr == HARQ_DECODED) {
                crc[tb] = false;                 // decoded 8 subframes ago: not decoded again, nothing written
              }
              if (hr == HARQ_NEW_TX || hr == HARQ_RE_TX) harq_db.update(ent, e.dci.pid, tb, c.sfn, c.sf_idx, crc[tb], e.dci.tb[tb].ndi != 0, e.dci.tb[tb].rv, tbs, now);
            }
            if (crc[tb] && jr.len[tb] > 0) {
              emitPdu(ch, r, name, poff, (uint32_t)jr.len[tb], d.rnti, c.tti, (uint8_t)tb);
              if (name[0] == 'R') unpackRar(ch.h_payload.data() + poff, jr.len[tb], false);
              if (name[0] == 'C') {
                if (combined) learnUeConfig(ch.h_payload.data() + poff, jr.len[tb], d.rnti);  // (not pre-parsed: the block was decoded in this turn)
                else learn(jr, tb, false);
              }
            }
          }
        }
      } else {  // unknown table: 64QAM table first, the 256QAM table only if both TBs failed, :1089-1243
        const int j = run(0);
        mimo_ret = has64 ? mimo_of(0, j) : -1;
        if (j >= 0) {
          const JobRes& jr = ch.jres[j];
          for (int tb = 0; tb < 2; tb++) {
            crc[tb] = jr.crc[tb] != 0;
            if (crc[tb] && jr.len[tb] > 0) {
              emitPdu(ch, r, name, jr.payload_off[tb], (uint32_t)jr.len[tb], d.rnti, c.tti, (uint8_t)tb);
              if (name[0] == 'R') unpackRar(ch.h_payload.data() + jr.payload_off[tb], jr.len[tb], false);
              if (name[0] == 'C') learn(jr, tb, true);  // :1133-1160: every SDU, whatever its logical channel
              if (d.mcs_idx[tb] > 0 && d.mcs_idx[tb] < 29 && d.format > FORMAT1A) mcs_tracking.update_RNTI_dl(d.rnti, TABLE_64QAM, now);
            }
          }
        }
        if (!crc[0] && !crc[1] && mimo_ret == 0) {
          const int j2 = run(1);
          mimo_ret = has256 ? mimo_of(1, j2) : -1;
          if (j2 >= 0) {
            const JobRes& jr = ch.jres[j2];
            for (int tb = 0; tb < 2; tb++) {
              if (jr.enabled[tb]) crc[tb] = jr.crc[tb] != 0;
              if (jr.crc[tb] && jr.len[tb] > 0) {
                emitPdu(ch, r, name, jr.payload_off[tb], (uint32_t)jr.len[tb], d.rnti, c.tti, (uint8_t)tb);
                if (d.mcs_idx[tb] > 0 && d.mcs_idx[tb] < 28 && d.format > FORMAT1A) mcs_tracking.update_RNTI_dl(d.rnti, TABLE_256QAM, now);
              }
            }
          }
        }
      }
      if (name[0] == 'C' && cfg.mcs_tracking_mode) {  // :1268-1285
        const bool tb_en[2] = {(cur_en & 1) != 0, (cur_en & 2) != 0};
        mcs_tracking.update_statistic_dl(d.rnti, (DciFormat)d.format, table, tb_en, crc, mimo_ret, now);
      }
      publishPrediction(d.rnti);
    }
  }
  if (cfg.harq_mode) harqFlush(ch, r);   // the chunk's keep store is recycled with the chunk: what its buffers hold there (and in the scratch area) goes home now
  for (const DecodeJob& j : ch.jobs) {
    if (!j.done) continue;
    const int k = j.kind < 5 ? j.kind : 0;
    r.perf.jobs_by_kind[k]++; r.perf.iters_by_kind[k] += j.iters;
    if (!j.used) { r.perf.jobs_unused_by_kind[k]++; r.perf.iters_unused_by_kind[k] += j.iters; }
  }
}

// ------------------------------------------------------------------------------------------------ HARQ soft buffers (harq_mode = 1)
// Block q of the buffer of a (RNTI entity, process, transport block) has its home in the pool at slot * HARQ_SLOT_WORDS + q * HARQ_CB_WORDS.  Inside a commit
// turn the content may lie elsewhere (HarqKeep::loc): a failed new transmission stays in the chunk's keep store, a combination in the turn's scratch area;
// harqFlush brings everything home before the chunk (and with it the keep store) is recycled.  Round 6: retransmissions are combined and decoded in batches
// ahead of the walk (lsn_engine.h: HarqReq) - rounds 4-5 paid one GPU round trip per retransmission inside the sequential turn.
uint64_t Engine::harqMix(uint64_t a, uint64_t b, uint64_t c, uint64_t d)
{
  auto sm = [](uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); };
  return sm(sm(sm(sm(a) ^ b) ^ c) ^ d);
}

// The request a retransmission (job, block) makes when it meets a buffer in the state (ncb_have, ver, ok, loc) - the SAME function serves the walk and the
// scout, so that equal states give equal keys.  A buffer without a first transmission on record for this geometry (ncb_have != n) is taken as it lies in
// the pool, nothing passed (rounds 4-5: hk = HarqKeep{}); the pool does not change inside a turn, so (slot, n) names that content.  false: every block has
// passed already, nothing to combine or decode.
bool Engine::harqRequest(const Chunk& ch, int job, int tb, size_t slot, uint32_t n, uint32_t ncb_have, uint64_t ver, const uint8_t* ok, const uint32_t* loc, HarqReq& q) const
{
  (void)ch;
  q = HarqReq{};
  q.job = job; q.tb = tb; q.slot = slot; q.n = n;
  uint64_t okmask = 0;
  if (ncb_have != n) {
    q.ver = harqMix(0x52455345u /* reset */, slot, n, 0);
    for (uint32_t i = 0; i < n; i++) { q.ok[i] = 0; q.loc[i] = HARQ_LOC_POOL | (uint32_t)(slot * HARQ_SLOT_WORDS + i * HARQ_CB_WORDS); }
  } else {
    q.ver = ver;
    for (uint32_t i = 0; i < n; i++) { q.ok[i] = ok[i] ? 1 : 0; q.loc[i] = loc[i]; okmask |= (uint64_t)(ok[i] ? 1 : 0) << i; }
  }
  q.key = harqMix(q.ver, ((uint64_t)(uint32_t)job << 1) | (uint64_t)(tb & 1), okmask, n);
  return okmask != ((1ull << n) - 1ull);
}

void Engine::harqStore(Chunk& ch, JobRunner& r, int job, int tb, size_t slot)
{
  (void)r;
  const DecodeJob& j = ch.jobs[job];
  const uint32_t n = j.keep_count[tb];
  if (!n || n > HARQ_MAX_CB) { harq_keep.erase(slot); return; }   // nothing stored for this transmission: what the slot held belongs to an older one and must not be combined with
  // cb_crc / data of the soft buffer: what passed in this (failed) transmission is remembered, a retransmission decodes the other blocks only
  HarqKeep& hk = harq_keep[slot];
  hk.ncb = n;
  hk.ver = harqMix(0x53544F52u /* store */, ch.gseq, (uint64_t)(uint32_t)job, (uint64_t)tb);
  uint32_t boff = 0;
  for (uint32_t q = 0; q < n; q++) {
    const LsnCbDev& cb = ch.keep_cbs[j.keep_first[tb] + q];
    const LsnCbRes cr = j.keep_first[tb] + q < ch.keep_res.size() ? ch.keep_res[j.keep_first[tb] + q] : LsnCbRes{};
    hk.ok[q] = cr.ok ? 1 : 0; hk.rem_a[q] = cr.rem_a; hk.K[q] = cb.K;
    const uint8_t* pb = ch.h_payload.data() + j.payload_off[tb] + boff;
    hk.bytes[q].assign(pb, pb + cb.out_bytes);
    boff += cb.out_bytes;
    hk.loc[q] = HARQ_LOC_KEEP | cb.spp_off;   // this transmission, where it lies in the chunk's keep store: nothing reads it before a retransmission combines with it
  }
  harq_touched.push_back(slot);
}

// end of the commit turn: the blocks that do not lie at home go there in at most one launch (rounds 4 / early 5 paid an upload, a launch and a stream
// synchronisation per failed transport block: 2.7 k subframes/s on the gated HARQ leg)
void Engine::harqFlush(Chunk& ch, JobRunner& r)
{
  struct Timer { double t0, &acc; ~Timer() { acc += now_ms() - t0; } } timer{now_ms(), r.perf.ms_harq[2]};
  std::vector<LsnCbDev> cp;
  std::sort(harq_touched.begin(), harq_touched.end());
  harq_touched.erase(std::unique(harq_touched.begin(), harq_touched.end()), harq_touched.end());
  for (size_t slot : harq_touched) {
    auto it = harq_keep.find(slot);
    if (it == harq_keep.end()) continue;
    HarqKeep& hk = it->second;
    for (uint32_t q = 0; q < hk.ncb && q < HARQ_MAX_CB; q++) {
      const uint32_t home = HARQ_LOC_POOL | (uint32_t)(slot * HARQ_SLOT_WORDS + q * HARQ_CB_WORDS);
      if (hk.loc[q] == home) continue;
      LsnCbDev cb{};
      cb.K = hk.K[q]; cb.reserved = hk.loc[q]; cb.spp_off = home;
      cp.push_back(cb);
      hk.loc[q] = home;
    }
  }
  harq_touched.clear();
  for (auto& kv : harq_cache) if (!kv.second.used) r.perf.nof_harq_combines[3]++;
  harq_cache.clear();
  harq_scratch_n = 0;
  (void)ch;
  if (cp.empty()) return;
  const uint32_t n = (uint32_t)cp.size();
  grow_host(harq_h_store, harq_h_store_cap, n, r.stream);
  grow_dev(harq_d_store, harq_d_store_cap, n, r.stream);
  std::memcpy(harq_h_store, cp.data(), n * sizeof(LsnCbDev));
  lsn_launch_upload(harq_d_store, harq_h_store, n * sizeof(LsnCbDev), r.stream);
  lsn_launch_harq_combine(harq_d_store, n, ch.d_keep, d_harq_pool, d_harq_scratch, true, r.stream);
  HIP_CHECK(hipStreamSynchronize(r.stream));   // the chunk's keep store is recycled with the chunk
}

// combine + decode a batch of requests: one descriptor upload, one combination launch, one decoder launch per wavefront class, one download, one wait
void Engine::harqRunBatch(Chunk& ch, JobRunner& r, const std::vector<HarqReq>& reqs)
{
  if (reqs.empty()) return;
  struct Timer { double t0, &acc; ~Timer() { acc += now_ms() - t0; } } timer{now_ms(), r.perf.ms_harq[1]};
  hipStream_t st = r.stream;
  struct Ref { uint32_t req, q, out; };
  std::vector<LsnCbDev> cbs;
  std::vector<Ref> refs;
  uint32_t out = 0;
  size_t words = harq_scratch_n;
  for (uint32_t i = 0; i < reqs.size(); i++) {
    const HarqReq& q = reqs[i];
    const DecodeJob& j = ch.jobs[q.job];
    for (uint32_t b = 0; b < q.n; b++) {
      if (q.ok[b]) continue;   // srsRAN: if (!softbuffer->cb_crc[cb_idx]) { rate de-matching into the buffer, decoding } - a passed block is left alone
      LsnCbDev cb = ch.keep_cbs[j.keep_first[q.tb] + b];
      cb.e_off = cb.spp_off;                       // this transmission, in the chunk's keep store
      cb.reserved = q.loc[b];                      // what the buffer holds, wherever it lies
      cb.spp_off = (uint32_t)words; words += LSN_SPP_WORDS(cb.K);
      cb.res_idx = (uint32_t)cbs.size(); cb.dep = LSN_CB_NODEP; cb.out_off = out;
      refs.push_back({i, b, out});
      out += cb.out_bytes;
      cbs.push_back(cb);
    }
  }
  const uint32_t nd = (uint32_t)cbs.size();
  if (!nd) return;
  if (words > harq_scratch_cap || words >= (1u << 30)) throw std::runtime_error("HARQ scratch area exhausted");   // (sized by the caller: harqEnsureScratch)
  harq_scratch_n = words;
  grow_host(r.h_cbs_pinned, r.h_cbs_cap, nd, st);
  grow_dev(r.d_cbs, r.cbs_cap, nd, st);
  if (nd > r.cbres_cap) grow_dev(r.d_cbres, r.cbres_cap, nd, st);
  grow_host(r.h_cbres_pinned, r.h_cbres_cap, nd, st);
  grow_dev(r.d_payload, r.payload_cap, (size_t)out + 16, st);
  grow_host(r.h_payload_pinned, r.h_payload_cap, (size_t)out + 16, st);
  // launch order: two-wavefront class first, each class by descending size (the longest first); results stay addressable through res_idx
  std::vector<uint32_t> order(nd);
  for (uint32_t i = 0; i < nd; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    const bool ca = turbo_nwin((int)cbs[a].K) > 64 || cbs[a].K > LSN_TURBO_ONE_WAVE_KMAX, cb2 = turbo_nwin((int)cbs[b].K) > 64 || cbs[b].K > LSN_TURBO_ONE_WAVE_KMAX;
    if (ca != cb2) return ca;
    return cbs[a].K > cbs[b].K;
  });
  uint32_t n128 = 0, kmax128 = 0, kmax64 = 0;
  for (uint32_t i = 0; i < nd; i++) {
    const LsnCbDev& c = cbs[order[i]];
    r.h_cbs_pinned[i] = c;
    if (lsn_turbo_two_wave_class((int)c.K)) { n128++; kmax128 = std::max(kmax128, c.K); } else kmax64 = std::max(kmax64, c.K);
  }
  lsn_launch_upload(r.d_cbs, r.h_cbs_pinned, nd * sizeof(LsnCbDev), st);
  lsn_launch_harq_combine(r.d_cbs, nd, ch.d_keep, d_harq_pool, d_harq_scratch, false, st);
  lsn_launch_turbo(cd, r.d_cbs, d_harq_scratch, r.d_payload, r.d_cbres, n128, kmax128, nd - n128, kmax64, st, nullptr);
  {
    LsnCopySegs dn;   // verdicts + payload bytes down in one launch
    dn.add(r.h_cbres_pinned, r.d_cbres, nd * sizeof(LsnCbRes));
    dn.add(r.h_payload_pinned, r.d_payload, out);
    lsn_launch_copy_multi(dn, true, st);
  }
  HIP_CHECK(hipEventRecord(r.ev_done, st));
  waitEvent(r.ev_done, 3000);   // inside the sequential commit turn: short naps (the decode threads' waits are milliseconds long and nap 50 us)
  r.perf.nof_harq_combines[0]++;
  for (uint32_t i = 0; i < reqs.size(); i++) { HarqDone& d = harq_cache[reqs[i].key]; d = HarqDone{}; d.req = reqs[i]; }
  for (uint32_t k = 0; k < nd; k++) {
    const Ref& f = refs[k];
    HarqDone& d = harq_cache[reqs[f.req].key];
    const LsnCbRes& cr = r.h_cbres_pinned[k];   // (res_idx = k: the index before sorting)
    d.ok[f.q] = cr.ok ? 1 : 0; d.rem_a[f.q] = cr.rem_a; d.iters[f.q] = cr.iters;
    d.loc[f.q] = HARQ_LOC_SCRATCH | cbs[k].spp_off;
    d.bytes[f.q].assign(r.h_payload_pinned + f.out, r.h_payload_pinned + f.out + cbs[k].out_bytes);
  }
}

// The combined decodes the walk over this chunk will probably ask for, as far as their inputs are known now: the walk's HARQ decisions (commitChunk,
// known-table branch) replayed on COPIES of the process database and of the touched buffers' states, with the tables, jobs and p-a values as they stand at
// the start of the turn.  A retransmission whose result is in harq_cache continues its buffer's chain; one without becomes a request, and the chain of that
// buffer stops for this pass (its later retransmissions need the result first).  Nothing but speed depends on how well this guesses: the walk makes its own
// requests and takes a result only under the key of exactly its inputs.
void Engine::harqScout(Chunk& ch, std::vector<HarqReq>& out, bool first_pass)
{
  out.clear();
  struct View { uint32_t ncb = 0; uint8_t ok[16] = {}; uint32_t rem_a[16] = {}, loc[16] = {}; uint64_t ver = 0; bool pending = false; };
  std::unordered_map<size_t, View> ov;
  auto view = [&](size_t slot) -> View& {
    auto it = ov.find(slot);
    if (it != ov.end()) return it->second;
    View v;
    auto k = harq_keep.find(slot);
    if (k != harq_keep.end()) {
      v.ncb = k->second.ncb; v.ver = k->second.ver;
      for (int q = 0; q < 16; q++) { v.ok[q] = k->second.ok[q]; v.rem_a[q] = k->second.rem_a[q]; v.loc[q] = k->second.loc[q]; }
    }
    return ov.emplace(slot, v).first->second;
  };
  // The transport blocks the walk will put to the process database, in walk order (tables, jobs and p-a values as they stand at the start of the turn: the
  // same in every pass of this turn, so the list is made by the first pass and replayed by the others).  job < 0: the database's 10 s timer.
  if (first_pass) {
    harq_events.clear();
    uint32_t cnt = commit_sf_cnt;
    for (uint32_t sf = 0; sf < ch.nsf; sf++, cnt++) {
      const SubframeCtx& c = ch.ctx[sf];
      if (cnt && (cnt % 10000u) == 0) { HarqEvent ev; ev.job = -1; ev.now = cnt; harq_events.push_back(ev); }
      if (!c.searched) continue;
      for (uint32_t k = ch.cdci_first[sf]; k < ch.cdci_first[sf + 1]; k++) {
        const CommitDci& d = ch.cdci[k];
        const char* name = rnti_name(d.rnti);
        if (name[0] != 'C') continue;
        McsTable table = TABLE_64QAM;
        if (cfg.mcs_tracking_mode == 1) table = (DciFormat)d.format == FORMAT1A ? TABLE_64QAM : mcs_tracking.peek(d.rnti);
        else if (cfg.mcs_tracking_mode == 2) table = TABLE_UNKNOWN;
        if (!(table == TABLE_64QAM || table == TABLE_256QAM)) continue;
        const TableView tv = table_view(table, d.rnti, d.flags & 1, d.flags & 2, d.flags & 4);
        const DlEntry& e = c.dl[d.di];
        if (table == TABLE_64QAM && e.unpack_ok && ((e.grant64.tb[0].enabled && e.grant64.tb[0].mcs_idx > 28) || (e.grant64.tb[1].enabled && e.grant64.tb[1].mcs_idx > 28)))
          continue;   // (a reserved MCS index takes its size from the database at commit and is decoded there)
        const int cur_t = table == TABLE_256QAM ? 1 : 0;
        const bool cur_has = cur_t ? tv.has256 : tv.has64;
        const int32_t cur_tbs0 = cur_has ? (cur_t ? d.tbs0_256 : d.tbs0_64) : 0;
        const bool two_tb = (tv.has64 && (d.flags & 8)) || (tv.has256 && (d.flags & 16));
        if (!(cur_tbs0 > 0 && tv.dci_rnti_ok && !(dlRx() == 1 && two_tb))) continue;
        const int j = d.job[cur_t];
        if (!cur_has || j < 0 || !ch.jres[j].done || ch.jres[j].p_a != mcs_tracking.get_ue_config_rnti(d.rnti).p_a) continue;
        const JobRes& jr = ch.jres[j];
        for (int tb = 0; tb < 2; tb++) {
          if (!jr.enabled[tb]) continue;
          HarqEvent ev;
          ev.job = j; ev.now = cnt; ev.sfn = c.sfn; ev.sf_idx = c.sf_idx; ev.rnti = d.rnti; ev.pid = (uint8_t)e.dci.pid; ev.tb = (uint8_t)tb;
          ev.ndi = e.dci.tb[tb].ndi != 0; ev.rv = (uint8_t)e.dci.tb[tb].rv; ev.tbs = ch.jobs[j].grant.tb[tb].tbs; ev.crc = jr.crc[tb] != 0; ev.n = ch.jobs[j].keep_count[tb];
          harq_events.push_back(ev);
        }
      }
    }
  }
  HarqDatabase db = harq_db;
  for (const HarqEvent& ev : harq_events) {
    if (ev.job < 0) { db.update_database(ev.now); continue; }
    const int j = ev.job, tb = ev.tb;
    int ent = -1;
    const HarqRet hr = db.is_retransmission(ev.rnti, ev.pid, tb, ev.ndi, ev.tbs, ev.sfn, ev.sf_idx, ent);
    const size_t slot = ent < 0 ? 0 : ((size_t)ent * HarqDatabase::NPID + (ev.pid & 7u)) * 2 + (size_t)tb;
    bool crc = ev.crc;
    const uint32_t n = ev.n;
    if (hr == HARQ_NEW_TX) {
      if (!crc) {
        View& v = view(slot);
        v = View{};
        if (n && n <= HARQ_MAX_CB) {
          v.ncb = n; v.ver = harqMix(0x53544F52u, ch.gseq, (uint64_t)(uint32_t)j, (uint64_t)tb);
          for (uint32_t q = 0; q < n; q++) {
            const size_t ki = ch.jobs[j].keep_first[tb] + q;
            const LsnCbRes cr = ki < ch.keep_res.size() ? ch.keep_res[ki] : LsnCbRes{};
            v.ok[q] = cr.ok ? 1 : 0; v.rem_a[q] = cr.rem_a; v.loc[q] = HARQ_LOC_KEEP | ch.keep_cbs[ki].spp_off;
          }
        }
      }
    } else if (hr == HARQ_RE_TX) {
      crc = false;
      if (n && n <= HARQ_MAX_CB) {
        View& v = view(slot);
        if (!v.pending) {
          HarqReq q;
          const bool work = harqRequest(ch, j, tb, slot, n, v.ncb, v.ver, v.ok, v.loc, q);
          if (v.ncb != n) { v = View{}; v.ncb = n; v.ver = q.ver; for (uint32_t b = 0; b < n; b++) v.loc[b] = q.loc[b]; }
          bool have = true;
          if (work) {
            auto it = harq_cache.find(q.key);
            if (it == harq_cache.end()) { out.push_back(q); v.pending = true; have = false; }
            else {
              const HarqDone& dn = it->second;
              for (uint32_t b = 0; b < n; b++)
                if (!v.ok[b]) { v.rem_a[b] = dn.rem_a[b]; v.loc[b] = dn.loc[b]; v.ok[b] = dn.ok[b]; }
              v.ver = q.key;
            }
          }
          if (have) {  // the verdict as far as the scout can tell (every block passed, CRC24A over the blocks; the parity-word and length tests are the walk's)
            bool all_ok = true;
            uint32_t rem = 0, shift = 1;
            for (int b = (int)n - 1; b >= 0; b--) {
              all_ok = all_ok && v.ok[b];
              rem ^= b == (int)n - 1 ? (v.rem_a[b] & 0xFFFFFFu) : crc24a_mulmod(v.rem_a[b], shift);
              if (b > 0) shift = crc24a_mulmod(shift, crc24a_xpow_bytes(ch.keep_cbs[ch.jobs[j].keep_first[tb] + b].out_bytes));
            }
            crc = all_ok && rem == 0;
          }
        }
      }
    } else if (hr == HARQ_DECODED) {
      crc = false;
    }
    if (hr == HARQ_NEW_TX || hr == HARQ_RE_TX) db.update(ent, ev.pid, tb, ev.sfn, ev.sf_idx, crc, ev.ndi, ev.rv, ev.tbs, ev.now);
  }
}

bool Engine::harqCombinedDecode(Chunk& ch, JobRunner& r, int job, int tb, size_t slot, uint32_t& payload_off)
{
  const DecodeJob& j = ch.jobs[job];
  const uint32_t n = j.keep_count[tb];
  if (!n || n > HARQ_MAX_CB) return false;
  HarqKeep& hk = harq_keep[slot];
  HarqReq q;
  const bool work = harqRequest(ch, job, tb, slot, n, hk.ncb, hk.ver, hk.ok, hk.loc, q);
  if (hk.ncb != n) {  // (no first transmission on record for this geometry: nothing passed before)
    hk = HarqKeep{};
    hk.ncb = n; hk.ver = q.ver;
    for (uint32_t b = 0; b < n; b++) hk.loc[b] = q.loc[b];
  }
  for (uint32_t b = 0; b < n; b++) hk.K[b] = ch.keep_cbs[j.keep_first[tb] + b].K;
  const HarqDone* dn = nullptr;
  if (work) {
    auto it = harq_cache.find(q.key);
    if (it != harq_cache.end() && !(it->second.req.ver == q.ver && it->second.req.job == job && it->second.req.tb == tb && it->second.req.n == n && std::memcmp(it->second.req.ok, q.ok, 16) == 0)) it = harq_cache.end();  // (a hash collision)
    if (it == harq_cache.end()) {
      // not foreseen by the scout: decoded now, alone (a round trip inside the turn, as every retransmission was in rounds 4-5)
      size_t need = 0;
      for (uint32_t b = 0; b < n; b++) if (!q.ok[b]) need += LSN_SPP_WORDS(hk.K[b]);
      if (harq_scratch_n + need > harq_scratch_cap) {  // scratch area full: everything goes home first (the unused results of the batches are lost with it)
        harqFlush(ch, r);
        harqRequest(ch, job, tb, slot, n, hk.ncb, hk.ver, hk.ok, hk.loc, q);
        if (need > harq_scratch_cap) { HIP_CHECK(hipStreamSynchronize(r.stream)); grow_dev(d_harq_scratch, harq_scratch_cap, need, r.stream); }
      }
      std::vector<HarqReq> one{q};
      harqRunBatch(ch, r, one);
      r.perf.nof_harq_combines[2]++;
      r.perf.nof_ondemand_decodes++;
      it = harq_cache.find(q.key);
    } else {
      r.perf.nof_harq_combines[1]++;
    }
    it->second.used = true;
    dn = &it->second;
    for (uint32_t b = 0; b < n; b++) {
      if (hk.ok[b]) continue;
      r.perf.nof_turbo_iterations += dn->iters[b];
      hk.rem_a[b] = dn->rem_a[b];
      hk.bytes[b] = dn->bytes[b];
      hk.loc[b] = dn->loc[b];
      // (ok is set below, after the verdict of THIS pass has been taken)
    }
    hk.ver = q.key;
    harq_touched.push_back(slot);
  }
  // transport-block verdict, as in runJobs: every block passed (now or in an earlier transmission), CRC24A over the assembled blocks
  bool all_ok = true;
  uint32_t rem = 0, total = 0, shift = 1;   // shift = x^bits_after mod g, carried along (as in runJobs)
  uint64_t bits_after = 0;
  for (int b = (int)n - 1; b >= 0; b--) {
    const bool okb = hk.ok[b] || (dn && dn->ok[b] != 0);
    all_ok = all_ok && okb;
    rem ^= bits_after ? crc24a_mulmod(hk.rem_a[b], shift) : (hk.rem_a[b] & 0xFFFFFFu);
    const uint32_t nb = ch.keep_cbs[j.keep_first[tb] + b].out_bytes;
    bits_after += 8ull * nb;
    if (b > 0) shift = crc24a_mulmod(shift, crc24a_xpow_bytes(nb));
  }
  for (uint32_t b = 0; b < n; b++) total += (uint32_t)hk.bytes[b].size();
  const int tbs = j.grant.tb[tb].tbs;
  payload_off = (uint32_t)ch.h_payload.size();
  ch.h_payload.resize(ch.h_payload.size() + (((size_t)total + 15) & ~(size_t)15));
  {
    uint8_t* dst = ch.h_payload.data() + payload_off;
    for (uint32_t b = 0; b < n; b++) { std::memcpy(dst, hk.bytes[b].data(), hk.bytes[b].size()); dst += hk.bytes[b].size(); }
  }
  for (uint32_t b = 0; b < n; b++)
    if (!hk.ok[b] && dn && dn->ok[b]) hk.ok[b] = 1;
  const uint8_t* pl = ch.h_payload.data() + payload_off;
  if ((uint64_t)total * 8ull < (uint64_t)tbs + 24ull) return false;
  const uint32_t par = ((uint32_t)pl[tbs / 8] << 16) | ((uint32_t)pl[tbs / 8 + 1] << 8) | pl[tbs / 8 + 2];
  return all_ok && rem == 0 && par != 0 && bits_after == (uint64_t)tbs + 24;
}

// decode threads: each takes the next chunk of the queue, plans and runs its PDSCH decodes on its own streams and hands the chunk
// to the commit thread; commits happen in queue order (PDU order and MCS-table learning stay in TTI order)
void Engine::decodeLoop(int idx)
{
  JobRunner& r = runner_c[idx];
  pinThisThread(nullptr);
  prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
  for (;;) {
    Chunk* ch = nullptr;
    {
      std::unique_lock<std::mutex> lk(mtx);
      cv_work.wait(lk, [&] { return stop || !commit_queue.empty(); });
      if (commit_queue.empty()) return;
      ch = commit_queue.front();
      commit_queue.pop_front();
    }
    std::string err = ch->err;
    if (err.empty()) try {
      (void)hipSetDevice(cfg.device);
      const double t0 = now_ms();
      trace((uint8_t)(2 + idx), TR_DEC_BEGIN, ch->trace_id);
      planJobs(*ch, r);
      trace((uint8_t)(2 + idx), TR_W2_DONE, ch->trace_id);
      r.perf.ms_stage_c += now_ms() - t0;
    } catch (const std::exception& ex) {
      err = ex.what();
    }
    {
      std::unique_lock<std::mutex> lk(mtx);
      decoded[ch->seq] = {ch, err};
    }
    cv_commit.notify_one();
  }
}

void Engine::commitLoop()
{
  JobRunner& r = runner_k;
  pinThisThread(nullptr);
  prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
  for (;;) {
    Chunk* ch = nullptr;
    std::string err;
    {
      std::unique_lock<std::mutex> lk(mtx);
      cv_commit.wait(lk, [&] { return stop || decoded.count(seq_committed); });
      auto it = decoded.find(seq_committed);
      if (it == decoded.end()) return;  // stop
      ch = it->second.first; err = it->second.second;
      decoded.erase(it);
    }
    {
      std::unique_lock<std::mutex> tl(sh->turn_mtx);
      sh->turn_cv.wait(tl, [&] { return sh->commit_turn == ch->gseq || stop; });
    }
    try {
      (void)hipSetDevice(cfg.device);
      const double t1 = now_ms();
      trace(15, TR_COMMIT_BEGIN, ch->trace_id);
      if (err.empty()) { if (cfg.sniffer_mode == 1) commitChunkUl(*ch, r); else commitChunk(*ch, r); }
      trace(15, TR_COMMIT_END, ch->trace_id);
      r.perf.ms_commit += now_ms() - t1;
    } catch (const std::exception& ex) {
      err = ex.what();
      // The queued soft-buffer copies point into this chunk's keep store and never went out: what the slots were about to remember (harq_keep, and the
      // database's "stored" marks behind it) no longer matches the pool.  A retransmission that later combined with such a slot would trust flags and
      // bytes of a transmission whose soft bits never arrived (round-5 advisor finding): the turn failed, so the whole soft-buffer state is dropped -
      // every later block of these processes is a new transmission, which costs combining gain for 8 subframes and nothing else.
      if (cfg.harq_mode) { harq_cache.clear(); harq_touched.clear(); harq_scratch_n = 0; harq_keep.clear(); harq_db = HarqDatabase(); }
    }
    {
      std::unique_lock<std::mutex> tl(sh->turn_mtx);
      sh->commit_turn = ch->gseq + 1;
    }
    sh->turn_cv.notify_all();
    {
      std::unique_lock<std::mutex> lk(mtx);
      ch->err = err;
      seq_committed++;
      write_queue.push_back(ch);
    }
    cv_write.notify_one();
    cv_commit.notify_one();
  }
}

// writer thread: the records of committed chunks go to the PDU sink (pcap writer / callback) in commit order; the chunk's slot is free
// again when its last record is out
void Engine::writerLoop()
{
  pinThisThread(nullptr);
  for (;;) {
    Chunk* ch = nullptr;
    {
      std::unique_lock<std::mutex> lk(mtx);
      cv_write.wait(lk, [&] { return stop || !write_queue.empty(); });
      if (write_queue.empty()) return;
      ch = write_queue.front();
      write_queue.pop_front();
    }
    std::string err = ch->err;
    {
      std::unique_lock<std::mutex> tl(sh->turn_mtx);
      sh->turn_cv.wait(tl, [&] { return sh->write_turn == ch->gseq || stop; });
    }
    if (err.empty() && (sink || api_mode >= 0)) {
      try {
        const uint8_t* base = ch->h_payload.data();
        for (const auto& rec : ch->recs) {
          if (sink) sink(sink_user, &rec.ctx, base + rec.off, rec.len);
          if (api_mode >= 0 && rec.ctx.direction == 1 && (rec.ctx.rnti_type == 1 || rec.ctx.rnti_type == 3)) {  // run_api_dl_mode, DL_Sniffer_PDSCH.cc:804-879
            ApiEvent ev[20];
            int nev = 0;
            const bool keep = api_dl_events(api_mode, rec.ctx.rnti_type == 1 ? 'P' : 'C', base + rec.off, (int)rec.len, rec.ctx.rnti, rec.ctx.tti, ev, 20, &nev);
            for (int i = 0; i < nev && api_sink; i++) {
              lsn_api_event_t e{};
              e.tti = ev[i].tti; e.rnti = ev[i].rnti; e.id_type = ev[i].id_type; e.msg_type = ev[i].msg_type;
              std::memcpy(e.value, ev[i].value, sizeof(e.value));
              api_sink(api_user, &e);
            }
            if (keep && api_pcap_sink) api_pcap_sink(api_pcap, &rec.ctx, base + rec.off, rec.len);
          } else if (api_mode >= 0 && rec.ctx.direction == 0) {  // decode_run's API part, UL_Sniffer_PUSCH.cc:306-372: Msg3 of a RAR grant (modes 0, 3), else SRB messages
            ApiEvent ev[10];
            int nev = 0;
            const bool msg3 = rec.msg3 && (api_mode == 0 || api_mode == 3);
            const bool keep = msg3 ? api_ul_msg3_events(api_mode, base + rec.off, (int)rec.len, rec.ctx.rnti, rec.ctx.tti, ev, 10, &nev)
                                   : api_ul_dcch_events(api_mode, base + rec.off, (int)rec.len, rec.ctx.rnti, rec.ctx.tti, ev, 10, &nev);
            for (int i = 0; i < nev && api_sink; i++) {
              lsn_api_event_t e{};
              e.tti = ev[i].tti; e.rnti = ev[i].rnti; e.id_type = ev[i].id_type; e.msg_type = ev[i].msg_type;
              std::memcpy(e.value, ev[i].value, sizeof(e.value));
              api_sink(api_user, &e);
            }
            if (keep && api_pcap_sink) api_pcap_sink(api_pcap, &rec.ctx, base + rec.off, rec.len);  // write_ul_crnti_api
          }
        }
      } catch (const std::exception& ex) {
        err = ex.what();
      }
    }
    {
      std::unique_lock<std::mutex> tl(sh->turn_mtx);
      sh->write_turn = ch->gseq + 1;
    }
    sh->turn_cv.notify_all();
    {
      std::unique_lock<std::mutex> lk(mtx);
      if (!err.empty() && commit_error.empty()) commit_error = err;
      seq_written++;
      ch->busy = false;
    }
    cv_done.notify_all();
  }
}

// ------------------------------------------------------------------------------------------------ batch driver
// front thread: cuts the submitted blocks into chunks and keeps stage A of up to NSTREAM_A chunks in flight (one stream each), also
// across submits (a new submit does not wait for the previous one's chunks to drain); finished chunks go on to the spec thread
void Engine::frontLoop()
{
  pinThisThread(nullptr);
  prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
  std::deque<Chunk*> inflight;
  // A chunk that failed somewhere keeps travelling with its error text (every later stage skips its work on it and the writer reports it):
  // the turn counters and seq_written then advance exactly as for a good chunk and wait() can always drain the pipeline.
  auto finish_oldest = [&] {
    Chunk* cur = inflight.front();
    inflight.pop_front();
    const double t0 = now_ms();
    if (cur->err.empty()) {
      try { finishStageA(*cur); } catch (const std::exception& ex) { cur->err = ex.what(); }
    }
    trace(1, TR_A_DONE, cur->trace_id);
    perf_front.ms_stage_a += now_ms() - t0;
    {
      std::unique_lock<std::mutex> lk(mtx);
      spec_queue.push_back(cur);
      seq_a_done++;
    }
    cv_spec.notify_one();
    cv_done.notify_all();
  };
  // One poll loop, two duties: (1) launch stage A of the next chunk as soon as a chunk slot is free, a stage-A stream is free and the chunk's
  // input is IN PLACE - the "block ready" event of its submit is waited for HERE, on the host (hipEventQuery), never by the GPU queue: a
  // barrier packet that waits for a copy several blocks ahead stalls every stream that shares the hardware queue (measured: the whole
  // decode stage stood still until the last queued copy had finished, tools/copy_kernel_timeline.py); (2) collect finished stage-A chunks
  // in order and pass them on.  Nothing to do -> nap 20 us (or sleep on the condition variable when the pipeline is empty).
  FrontJob job;
  bool have_job = false, ready_ok = false;
  uint32_t ci = 0, nchunks = 0;
  double t_wait_slot = -1.0;
  for (;;) {
    if (!have_job) {
      std::unique_lock<std::mutex> lk(mtx);
      if (inflight.empty()) cv_front.wait(lk, [&] { return stop || !front_jobs.empty(); });
      if (front_jobs.empty() && inflight.empty()) return;  // stop
      if (!front_jobs.empty()) {
        job = front_jobs.front(); front_jobs.pop_front();
        have_job = true; ready_ok = job.ready == nullptr; ci = 0;
        nchunks = (job.nsf_total + max_batch - 1) / max_batch;
      }
    }
    (void)hipSetDevice(cfg.device);
    bool progress = false;
    // (2) the oldest chunk in flight
    if (!inflight.empty()) {
      Chunk* o = inflight.front();
      // a query that fails for another reason than "not ready" is an error of this chunk, not its completion: the mirrors are not read
      const hipError_t q = o->err.empty() ? hipEventQuery(o->ev_a[16]) : hipSuccess;
      if (q != hipErrorNotReady) {
        if (q != hipSuccess) { o->err = std::string("stage A: ") + hipGetErrorString(q); (void)hipGetLastError(); }
        finish_oldest();
        progress = true;
      }
    }
    // (1) the next chunk of the current block
    if (have_job && inflight.size() < (size_t)NSTREAM_A) {
      if (!ready_ok) {
        const hipError_t q = hipEventQuery(job.ready);   // the block is waited for HERE (a GPU-side wait would be a barrier packet in stage A's hardware queue)
        if (q != hipErrorNotReady) { (void)hipGetLastError(); ready_ok = true; }
      }
      Chunk& ch = chunks[slot_counter % (uint64_t)nslots];  // slots rotate across submits
      bool slot_free = false;
      if (ready_ok) {
        std::unique_lock<std::mutex> lk(mtx);
        if (stop) return;
        slot_free = !ch.busy;
        if (slot_free) ch.busy = true;
      }
      if (ready_ok && !slot_free && t_wait_slot < 0) { t_wait_slot = now_ms(); trace(1, TR_ACQ_BEGIN, ci); }
      if (ready_ok && slot_free) {
        if (t_wait_slot >= 0) { perf_front.ms_wait_slot += now_ms() - t_wait_slot; t_wait_slot = -1.0; } else trace(1, TR_ACQ_BEGIN, ci);
        slot_counter++;
        const uint32_t base = ci * max_batch;
        const size_t sf_stride = (size_t)cfg.nof_rx_antennas * cd.sflen * sizeof(cf32);  // (the cell is set after this thread has started)
        ch.nsf = std::min(max_batch, job.nsf_total - base);
        ch.start_tti = job.start_tti + base;
        ch.update_meta_period = job.update_meta_period;
        ch.force_meta = job.force_meta && ci == 0;
        ch.gseq = job.gseq0 + ci;
        ch.jobs.clear(); ch.jres.clear(); ch.tapjobs.clear(); ch.keep_cbs.clear(); ch.keep_res.clear(); ch.keep_n = 0; ch.cdci.clear(); ch.setup_cfgs.clear(); ch.h_payload.clear(); ch.recs.clear(); ch.err.clear();
        for (uint32_t i = 0; i < ch.nsf; i++) ch.ctx[i].reset(ch.start_tti + i);
        ch.st_a = stream_a[(slot_counter - 1) % NSTREAM_A];
        ch.trace_id = ci;
        trace(1, TR_ACQ_END, ci);
        try {
          if (job.inject_fail == (int)ci) throw std::runtime_error("injected stage-A failure (LSN_INJECT_STAGE_A_ERROR)");
          launchStageA(ch, (const uint8_t*)job.d_iq + (size_t)base * sf_stride);
        } catch (const std::exception& ex) {
          ch.err = ex.what();
          (void)hipStreamSynchronize(ch.st_a);  // kernels of this launch that were queued before it failed still write the chunk's buffers: the slot is released only behind them
          (void)hipGetLastError();
        }
        inflight.push_back(&ch);
        progress = true;
        if (++ci == nchunks) {
          have_job = false;
          if (job.ready) {  // the event can carry another block
            std::unique_lock<std::mutex> lk(mtx);
            ev_pool.push_back(job.ready);
          }
        }
      }
    }
    if (!progress) {
      timespec ts{0, 20000};
      nanosleep(&ts, nullptr);
    }
  }
}

// search thread (stage B): the FALCON decision tree over every subframe of every chunk, strictly in order
void Engine::searchLoop()
{
  pinThisThread(nullptr);
  for (;;) {
    Chunk* cur = nullptr;
    {
      const double tw = now_ms();
      std::unique_lock<std::mutex> lk(mtx);
      cv_search.wait(lk, [&] { return stop || !search_queue.empty(); });
      if (search_queue.empty()) return;
      perf_search.ms_wait_front += now_ms() - tw;
      cur = search_queue.front();
      search_queue.pop_front();
    }
    {  // the sequential search state may be shared with the engines of other GPUs: chunk g is searched when chunks 0 .. g-1 have been
      std::unique_lock<std::mutex> tl(sh->turn_mtx);
      sh->turn_cv.wait(tl, [&] { return sh->search_turn == cur->gseq || stop; });
    }
    if (cur->err.empty()) {
      try {
        (void)hipSetDevice(cfg.device);
        const double t1 = now_ms();
        trace(0, TR_SEARCH_BEGIN, cur->trace_id);
        searchChunk(*cur, cur->update_meta_period);
        trace(0, TR_SEARCH_END, cur->trace_id);
        perf_search.ms_search += now_ms() - t1;
      } catch (const std::exception& ex) {
        cur->err = ex.what();
      }
    }
    {
      std::unique_lock<std::mutex> tl(sh->turn_mtx);
      sh->search_turn = cur->gseq + 1;
    }
    sh->turn_cv.notify_all();
    {
      std::unique_lock<std::mutex> lk(mtx);
      cur->seq = seq_pushed++;
      commit_queue.push_back(cur);
      last_chunk = cur;
    }
    cv_work.notify_one();
  }
}

// speculative RA-RNTI decodes of finished stage-A chunks (a GPU round trip per chunk), off the front thread so that stage A of the
// following chunks keeps being launched and collected meanwhile; chunks reach the search thread in order
void Engine::specLoop()
{
  pinThisThread(nullptr);
  prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
  for (;;) {
    Chunk* cur = nullptr;
    {
      std::unique_lock<std::mutex> lk(mtx);
      cv_spec.wait(lk, [&] { return stop || !spec_queue.empty(); });
      if (spec_queue.empty()) return;
      cur = spec_queue.front();
      spec_queue.pop_front();
    }
    if (cur->err.empty()) {
      try {
        (void)hipSetDevice(cfg.device);
        speculateRar(*cur);
        trace(14, TR_SPEC_DONE, cur->trace_id);
      } catch (const std::exception& ex) {
        cur->err = ex.what();
      }
    }
    {
      std::unique_lock<std::mutex> lk(mtx);
      search_queue.push_back(cur);
    }
    cv_search.notify_one();
  }
}

// process = submit + wait.  submit() only queues the block (the caller keeps the IQ buffer alive until wait() / waitMark()); the front,
// spec, search, decode, commit and writer threads take it from there, so consecutive submits flow through the pipeline without a gap.
int Engine::process(const void* d_iq, uint32_t nsf_total, uint32_t start_tti, uint32_t update_meta_period, hipStream_t stream)
{
  const int r = submit(d_iq, nsf_total, start_tti, update_meta_period, stream);
  const int w = wait();
  return r != LSN_SUCCESS ? r : w;
}

int Engine::submit(const void* d_iq, uint32_t nsf_total, uint32_t start_tti, uint32_t update_meta_period, hipStream_t stream, bool force_meta_first)
{
  if (!cell_set) return LSN_ERROR;
  if (!d_iq && nsf_total) return LSN_ERROR_INVALID_INPUTS;
  try {
    HIP_CHECK(hipSetDevice(cfg.device));
    if (!batch_open) {  // first submit since the last wait: the counters describe one submit ... wait span
      perf = lsn_perf_t{};
      perf_front = lsn_perf_t{};
      perf_search = lsn_perf_t{};
      for (auto& r : runner_c) r.perf = lsn_perf_t{};
      runner_s.perf = lsn_perf_t{};
      runner_f.perf = lsn_perf_t{};
      runner_k.perf = lsn_perf_t{};
      t_batch = now_ms();
      batch_open = true;
    }
    // the caller's stream orders the IQ buffer: stage A of THIS block's chunks starts after everything queued on that stream so far.  The event
    // travels with the block and is waited for by the stage-A stream a chunk is launched on, at launch time (front thread) - a wait inserted
    // here into all stage-A streams would also hold back the chunks of EARLIER blocks that are launched after this call (with host -> device
    // copies queued several blocks ahead that serialised copy and compute: the 27 GB/s ingest of round 2).
    hipEvent_t ev = nullptr;
    {
      std::unique_lock<std::mutex> lk(mtx);
      if (!ev_pool.empty()) { ev = ev_pool.back(); ev_pool.pop_back(); }
    }
    if (!ev) HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(ev, stream));
    const uint32_t nchunks = (nsf_total + max_batch - 1) / max_batch;
    {
      std::unique_lock<std::mutex> lk(mtx);
      front_jobs.push_back({d_iq, nsf_total, start_tti, update_meta_period, sh->next_gseq.fetch_add(nchunks), force_meta_first, ev, inject_stage_a_fail});
      inject_stage_a_fail = -1;  // one shot
      chunks_expected += nchunks;
    }
    cv_front.notify_one();
    return LSN_SUCCESS;
  } catch (const std::exception& ex) {
    fprintf(stderr, "ltesniffer_amd: %s\n", ex.what());
    return LSN_ERROR;
  }
}

int Engine::submitFrom(const void* d_iq, int src_device, uint32_t nsf, uint32_t start_tti, uint32_t update_meta_period, hipStream_t stream)
{
  static const bool force_copy = getenv("LSN_FORCE_PEER_COPY") && atoi(getenv("LSN_FORCE_PEER_COPY"));  // tests: take the staging path on one GPU too
  if (src_device == cfg.device && !force_copy) return submit(d_iq, nsf, start_tti, update_meta_period, stream);
  if (!cell_set || nsf > max_batch) return LSN_ERROR_INVALID_INPUTS;
  try {
    HIP_CHECK(hipSetDevice(cfg.device));
    if (!copy_stream) {
      createCopyStream();
      for (auto& e : copy_done) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const size_t sf_stride = (size_t)cfg.nof_rx_antennas * cd.sflen * sizeof(cf32);
    const uint32_t slot = peer_slot++ % 12u;  // the staging area holds twelve chunks
    if (peer_marks[slot]) waitIqConsumed(peer_marks[slot]);
    uint8_t* dst = (uint8_t*)d_iq_staging + (size_t)slot * max_batch * sf_stride;
    // the copy is ordered behind the caller's stream (the source block) and in front of this engine's stage A.  The caller's stream lives on
    // the SOURCE device: the event that marks "block ready" must be created and recorded there (an event of this engine's device is rejected
    // with hipErrorInvalidHandle); waiting on it from a stream of another device is allowed.
    if (src_device < 0 || src_device >= 16) return LSN_ERROR_INVALID_INPUTS;
    HIP_CHECK(hipSetDevice(src_device));
    if (!peer_ev[src_device]) HIP_CHECK(hipEventCreateWithFlags(&peer_ev[src_device], hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(peer_ev[src_device], stream));
    HIP_CHECK(hipSetDevice(cfg.device));
    HIP_CHECK(hipStreamWaitEvent(copy_stream, peer_ev[src_device], 0));
    HIP_CHECK(hipMemcpyPeerAsync(dst, cfg.device, d_iq, src_device, (size_t)nsf * sf_stride, copy_stream));
    const int rc = submit(dst, nsf, start_tti, update_meta_period, copy_stream);
    peer_marks[slot] = submitMark();
    return rc;
  } catch (const std::exception& ex) {
    fprintf(stderr, "ltesniffer_amd: %s\n", ex.what());
    return LSN_ERROR;
  }
}

// The stream the IQ blocks are copied on.  Round 3 gave it the lowest priority (a class of its own, so that its "copy done" barrier packets would
// hold up nobody else's kernels); since the front thread waits for a block on the HOST (poll loop) no stage-A stream carries such a barrier any
// more, and the low class only delays the copies themselves: measured in round 4 on one box (tools/host_leg_probe.py 12000 400) 85-88 k
// subframes/s at lowest, 96-101 k at default, 90-93 k at highest priority.  Default priority.
void Engine::createCopyStream()
{
  HIP_CHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
}

int Engine::submitHostRows(const void* host_rows, size_t row_pitch, uint32_t nsf, uint32_t start_tti, bool force_meta_first, hipEvent_t copied)
{
  if (!cell_set || !host_rows || nsf == 0 || nsf > max_batch) return LSN_ERROR_INVALID_INPUTS;
  try {
    HIP_CHECK(hipSetDevice(cfg.device));
    if (!copy_stream) {
      createCopyStream();
      for (auto& e : copy_done) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const size_t row_bytes = (size_t)cd.sflen * sizeof(cf32), sf_stride = (size_t)cfg.nof_rx_antennas * row_bytes;
    const uint32_t slot = peer_slot++ % 12u;
    if (peer_marks[slot]) waitIqConsumed(peer_marks[slot]);
    uint8_t* dst = (uint8_t*)d_iq_staging + (size_t)slot * max_batch * sf_stride;
    if (row_pitch == row_bytes)
      HIP_CHECK(hipMemcpyAsync(dst, host_rows, (size_t)nsf * sf_stride, hipMemcpyHostToDevice, copy_stream));
    else
      HIP_CHECK(hipMemcpy2DAsync(dst, row_bytes, host_rows, row_pitch, row_bytes, (size_t)nsf * cfg.nof_rx_antennas, hipMemcpyHostToDevice, copy_stream));
    if (copied) HIP_CHECK(hipEventRecord(copied, copy_stream));
    const int rc = submit(dst, nsf, start_tti, 0u, copy_stream, force_meta_first);
    peer_marks[slot] = submitMark();
    return rc;
  } catch (const std::exception& ex) {
    fprintf(stderr, "ltesniffer_amd: %s\n", ex.what());
    return LSN_ERROR;
  }
}

int Engine::wait()
{
  if (!batch_open) return LSN_SUCCESS;
  try {
    std::string err;
    {
      const double tw = now_ms();
      std::unique_lock<std::mutex> lk(mtx);
      // drain: a failed chunk still passes every stage (skipping the work), so this always completes and the counters stay consistent
      cv_done.wait(lk, [&] { return seq_written == chunks_expected; });
      perf.ms_drain += now_ms() - tw;
      err = commit_error;
      commit_error.clear();
    }
    batch_open = false;
    if (!err.empty()) throw std::runtime_error(err);
    perf.nof_candidates_decoded = search->nof_lookups; search->nof_lookups = 0;
    perf.ms_search_core = 0;
    mergePerf(perf_search);
    perf.ms_wait_front += perf_search.ms_wait_front;
    mergePerf(perf_front);
    for (auto& r : runner_c) mergePerf(r.perf);
    mergePerf(runner_s.perf);
    mergePerf(runner_f.perf);
    mergePerf(runner_k.perf);
    perf.ms_total = now_ms() - t_batch;
    traceDump();
    return LSN_SUCCESS;
  } catch (const std::exception& ex) {
    fprintf(stderr, "ltesniffer_amd: %s\n", ex.what());
    return LSN_ERROR;
  }
}

// Host buffers (the worker pool's pinned staging, or any caller memory): blocks travel over PCIe into a ring of device staging buffers on a
// copy stream of their own while the pipeline works on the blocks before them (submit() only queues).  Caller memory that is not pinned
// yet is registered for the duration of the call so that the copies are real DMA transfers; if the registration is refused the copies
// fall back to the runtime's bounce buffers (still overlapped with the compute of earlier blocks).
// sample_format LSN_FILE_SC16 / LSN_FILE_SC8 (lsn_phy_process_host_int): the caller's buffers hold integer I/Q pairs in the same [subframe][antenna][sample]
// order; a block crosses PCIe as it is (half / a quarter of the bytes) into a raw ring and is converted into its staging block by the file source's kernel
// (k_file_unpack with one "antenna" of nof_rx * sflen samples = a flat conversion) on the copy stream, in front of the submit.
int Engine::processHost(const void* iq, uint32_t nsf_total, uint32_t start_tti, uint32_t update_meta_period, uint32_t fmt, float scale)
{
  if (!cell_set) return LSN_ERROR;
  if ((!iq && nsf_total) || fmt > LSN_FILE_SC8) return LSN_ERROR_INVALID_INPUTS;
  if (fmt != LSN_FILE_CF32 && !(scale >= 0.0f && scale < INFINITY)) return LSN_ERROR_INVALID_INPUTS;
  if (fmt != LSN_FILE_CF32 && scale == 0.0f) scale = fmt == LSN_FILE_SC16 ? 1.0f / 32768.0f : 1.0f / 128.0f;
  bool registered = false;
  try {
    HIP_CHECK(hipSetDevice(cfg.device));
    const size_t sf_stride = (size_t)cfg.nof_rx_antennas * cd.sflen * sizeof(cf32);
    const size_t in_stride = fmt == LSN_FILE_SC16 ? sf_stride / 2 : fmt == LSN_FILE_SC8 ? sf_stride / 4 : sf_stride;  // bytes of one subframe in the caller's memory
    // ring of staging blocks, one pipeline chunk each (max_batch subframes: 393 MB at 20 MHz / 2 rx / 800 subframes - large copies run at the
    // link rate, tools/ubench/h2d_bw.hip).  A block is reusable as soon as stage A of its chunk has consumed the samples (not when the chunk has
    // left the whole pipeline: with decode, commit and write behind stage A that is six or more chunk times later and throttled the copies to
    // three blocks per pipeline latency - the 27 GB/s of round 2).  Copies are queued ahead on their own stream, so the link stays busy.
    const uint32_t blk = max_batch;
    if (!copy_stream) {
      createCopyStream();
      for (auto& e : copy_done) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    {
      hipPointerAttribute_t attr{};
      const bool pinned = hipPointerGetAttributes(&attr, iq) == hipSuccess && attr.type == hipMemoryTypeHost;
      if (!pinned && (size_t)nsf_total * in_stride >= ((size_t)8 << 20))
        registered = hipHostRegister((void*)iq, (size_t)nsf_total * in_stride, hipHostRegisterDefault) == hipSuccess;
      (void)hipGetLastError();
    }
    static const bool host_debug = getenv("LSN_HOST_DEBUG") != nullptr;
    const double t_host0 = now_ms();
    const uint32_t nring = (uint32_t)std::max<size_t>(2, staging_sf / blk);
    if (fmt != LSN_FILE_CF32 && d_iq_raw_bytes < (size_t)nring * blk * in_stride) {
      if (d_iq_raw) { (void)hipFree(d_iq_raw); d_iq_raw = nullptr; d_iq_raw_bytes = 0; }
      HIP_CHECK(hipMalloc(&d_iq_raw, (size_t)nring * blk * in_stride));
      d_iq_raw_bytes = (size_t)nring * blk * in_stride;
    }
    std::vector<uint64_t> marks(nring, 0);
    uint32_t k = 0;
    int rc = LSN_SUCCESS;
    for (uint32_t base = 0; base < nsf_total && rc == LSN_SUCCESS; base += blk, k++) {
      const uint32_t nsf = std::min(blk, nsf_total - base), slot = k % nring;
      if (k >= nring) waitIqConsumed(marks[slot]);
      uint8_t* dst = (uint8_t*)d_iq_staging + (size_t)slot * blk * sf_stride;
      const double tc0 = now_ms();
      if (fmt == LSN_FILE_CF32) {
        HIP_CHECK(hipMemcpyAsync(dst, (const uint8_t*)iq + (size_t)base * sf_stride, (size_t)nsf * sf_stride, hipMemcpyHostToDevice, copy_stream));
      } else {  // (the raw slot is free when the staging slot is: the conversion that read it ran in front of the stage A that `marks` waits for)
        uint8_t* raw = (uint8_t*)d_iq_raw + (size_t)slot * blk * in_stride;
        HIP_CHECK(hipMemcpyAsync(raw, (const uint8_t*)iq + (size_t)base * in_stride, (size_t)nsf * in_stride, hipMemcpyHostToDevice, copy_stream));
        lsn_launch_file_unpack(raw, fmt, scale, nullptr, cfg.nof_rx_antennas * cd.sflen, 1, (cf32*)dst, nsf, copy_stream);
      }
      const double tc1 = now_ms();
      rc = submit(dst, nsf, start_tti + base, update_meta_period, copy_stream);  // stage A of the block waits for the copy on the device
      marks[slot] = submitMark();
      if (host_debug) fprintf(stderr, "process_host: block %u: hipMemcpyAsync call %.3f ms, submit %.3f ms (t = %.3f ms)\n", k, tc1 - tc0, now_ms() - tc1, tc0 - t_host0);
    }
    const int w = wait();
    if (registered) (void)hipHostUnregister((void*)iq);
    return rc != LSN_SUCCESS ? rc : w;
  } catch (const std::exception& ex) {
    (void)wait();  // queued copies still read the caller's memory: drain before the registration goes away
    if (copy_stream) (void)hipStreamSynchronize(copy_stream);
    if (registered) (void)hipHostUnregister((void*)iq);
    fprintf(stderr, "ltesniffer_amd: %s\n", ex.what());
    return LSN_ERROR;
  }
}

// ------------------------------------------------------------------------------------------------ parity taps
// taps address the LAST chunk of the last process call (tests use calls of at most max_batch subframes)
long Engine::tap(int what, uint32_t sf, void* out, size_t cap)
{
  if (!cell_set || !last_chunk) return LSN_ERROR_INVALID_INPUTS;
  const bool by_job = what >= LSN_TAP_PDSCH_JOBS && what <= LSN_TAP_CB_RESULT;  // stage-C taps are indexed by decode job
  if (!by_job && sf >= last_chunk->nsf) return LSN_ERROR_INVALID_INPUTS;
  Chunk& ch = *last_chunk;
  const size_t A = dlRx(), P = cell.nof_ports, nre = cd.nre;
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
    case LSN_TAP_GRID: return d2h(ch.d_grid + (size_t)sf * A * 14 * nre, A * 14 * nre * sizeof(cf32));
    case LSN_TAP_CE: return d2h(ch.d_ce + (size_t)sf * P * A * 14 * nre, P * A * 14 * nre * sizeof(cf32));
    case LSN_TAP_PDCCH_LLR: return d2h(ch.d_llr + (size_t)sf * LSN_LLR_STRIDE, (size_t)cd.nof_cce[ch.ctx[sf].cfi - 1] * 72 * sizeof(float));
    case LSN_TAP_CHEST: {
      // layout of the test-side record: noise[2][W], rsrp[2][W], cepow[2][W] with W = 2 (one or two ports) or 4, then corr_r, corr_i, noise_avg,
      // rsrp_avg, snr_db, cfo_hz, chan_ref: 19 or 31 floats
      float r[31] = {0};
      const LsnChest& h = ch.h_chest[sf];
      const size_t W = P == 4 ? 4 : 2, T = 6 * W;
      for (size_t rx = 0; rx < A; rx++)
        for (size_t p = 0; p < P; p++) {
          r[rx * W + p] = h.noise[rx * P + p]; r[2 * W + rx * W + p] = h.rsrp[rx * P + p]; r[4 * W + rx * W + p] = h.cepow[rx * P + p];
        }
      r[T] = h.corr_r; r[T + 1] = h.corr_i; r[T + 2] = h.noise_avg; r[T + 3] = h.rsrp_avg; r[T + 4] = ch.ctx[sf].snr_db; r[T + 5] = ch.ctx[sf].cfo_hz; r[T + 6] = h.chan_ref;
      return h2h(r, (T + 7) * sizeof(float));
    }
    case LSN_TAP_CFI: return h2h(&ch.ctx[sf].cfi, sizeof(uint32_t));
    case LSN_TAP_CANDIDATES: return h2h(ch.h_cand + (size_t)sf * LSN_MAX_LOC * LSN_MAX_SIZES, (size_t)LSN_MAX_LOC * LSN_MAX_SIZES * sizeof(LsnCand));
    case LSN_TAP_CCE_POWER: return h2h(ch.h_ccepow + (size_t)sf * LSN_CCE_STRIDE, LSN_CCE_STRIDE * sizeof(float));
    case LSN_TAP_ACCEPTED: return h2h(ch.ctx[sf].accepted.data(), ch.ctx[sf].accepted.size() * sizeof(uint32_t));
    case LSN_TAP_RB_POWER: {
      // SubframePower.cc:34-41: dB conversion on the host
      float r[110];
      const float logdiv = 10.0f * log10f(14.0f);
      for (uint32_t i = 0; i < cell.nof_prb; i++) r[i] = 10.0f * log10f(ch.h_rbp[sf * 128 + i]) - logdiv;
      return h2h(r, cell.nof_prb * sizeof(float));
    }
    // ---- stage C (a14): `sf` is the decode job of the chunk; filled only after lsn_phy_set_stage_c_taps(phy, 1)
    case LSN_TAP_PDSCH_JOBS: {
      std::lock_guard<std::mutex> lk(tap_mtx);
      std::vector<lsn_tap_job_t> v(ch.jobs.size());
      for (size_t j = 0; j < ch.jobs.size(); j++) {
        lsn_tap_job_t& o = v[j];
        std::memset(&o, 0, sizeof(o));
        const DecodeJob& dj = ch.jobs[j];
        o.sf = dj.sf; o.tti = ch.ctx[dj.sf].tti; o.rnti = dj.rnti; o.nof_re = dj.grant.nof_re; o.p_a_db = dj.p_a; o.done = dj.done ? 1u : 0u;
        for (int i = 0; i < 2; i++) { o.tbs[i] = dj.grant.tb[i].enabled ? (uint32_t)std::max(0, dj.grant.tb[i].tbs) : 0u; o.crc[i] = dj.crc[i] ? 1u : 0u; }
        if (j < ch.tapjobs.size() && ch.tapjobs[j].have) {
          const TapJob& t = ch.tapjobs[j];
          o.have = 1; o.ncb = (uint32_t)t.cbs.size();
          for (int q = 0; q < 2; q++) { o.qm[q] = t.d.qm[q]; o.llr_len[q] = (uint32_t)t.llr[q].size(); }
        }
      }
      return h2h(v.data(), v.size() * sizeof(lsn_tap_job_t));
    }
    case LSN_TAP_PDSCH_LLR16: case LSN_TAP_RM_WORDS: case LSN_TAP_CB_RESULT: {
      std::lock_guard<std::mutex> lk(tap_mtx);
      if (sf >= ch.tapjobs.size() || !ch.tapjobs[sf].have) return 0;
      const TapJob& t = ch.tapjobs[sf];
      if (what == LSN_TAP_PDSCH_LLR16) {
        const size_t n0 = t.llr[0].size() * 2, n1 = t.llr[1].size() * 2;
        if (n0 + n1 > cap) return LSN_ERROR_INVALID_INPUTS;
        if (n0) std::memcpy(out, t.llr[0].data(), n0);
        if (n1) std::memcpy((uint8_t*)out + n0, t.llr[1].data(), n1);
        return (long)(n0 + n1);
      }
      if (what == LSN_TAP_RM_WORDS) {
        size_t n = 0;
        for (auto& c : t.cbs) n += c.words.size() * 4;
        if (n > cap) return LSN_ERROR_INVALID_INPUTS;
        uint8_t* p = (uint8_t*)out;
        for (auto& c : t.cbs) { std::memcpy(p, c.words.data(), c.words.size() * 4); p += c.words.size() * 4; }
        return (long)n;
      }
      std::vector<lsn_tap_cb_t> v(t.cbs.size());
      for (size_t i = 0; i < t.cbs.size(); i++) {
        const TapCb& c = t.cbs[i];
        const bool skipped = c.cb.dep != LSN_CB_NODEP && c.res.iters == 0;
        v[i] = lsn_tap_cb_t{c.tb, c.cb.K, c.cb.F, c.cb.E, c.cb.rv, c.res.ok, c.res.iters, skipped ? 1u : 0u};
      }
      return h2h(v.data(), v.size() * sizeof(lsn_tap_cb_t));
    }
    default: return LSN_ERROR_INVALID_INPUTS;
  }
}

}  // namespace lsn
