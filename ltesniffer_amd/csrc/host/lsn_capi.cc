// lsn_capi.cc - the C ABI of include/ltesniffer_amd.h: Phy (worker pool + pending queue) and SubframeWorker (pinned IQ
// buffers) on top of the batched GPU engine.
//   Phy::Phy / getAvail / getAvailImmediate / putPending / joinPending   /root/reference/src/src/Phy.cc:5-109
//   SnifferThread::execute_worker                                        /root/reference/src/src/WorkerThread.cc:78-91
//   SubframeBuffer (3 * SF_LEN_PRB(100) samples per antenna)             /root/reference/src/src/SubframeBuffer.cc:25-28
// The reference runs `nof_workers` CPU threads each calling SubframeWorker::work(); here ONE dispatcher thread drains the
// pending queue in FIFO (= TTI) order into GPU batches, which is the reference's sequential (-W 1, file replay) semantic.
// The pool is pipelined: the workers' IQ buffers are rows of ONE pinned slab ([worker][antenna][3 SF_LEN], SubframeBuffer.cc:25), a run
// of pending workers crosses PCIe with a single strided copy straight from that slab (no staging memcpy) and is queued into the engine
// (submit, not process); a worker returns to the avail queue as soon as the copy of its rows has completed - the search / decode /
// commit / write stages of its subframe run on while the caller refills it.  Lossless file mode (blocking getAvail, LTESniffer_Core.cc:441)
// lets batches grow towards half the pool; live mode (getAvailImmediate, :439) dispatches whatever is pending at once.
#include "lsn_engine.h"
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <time.h>

struct lsn_worker {
  lsn_phy* phy = nullptr;
  float* buf[LSN_MAX_RX] = {nullptr, nullptr};
  float* buf_offset[2] = {nullptr, nullptr};  // SubframeBuffer::sf_buffer_offset (allocated on first use)
  uint32_t buf_len = 0;  // complex samples per antenna
  uint32_t index = 0;    // position in the slab
  uint32_t sf_idx = 0, sfn = 0;
  int update_meta = 0;
  lsn_dl_sf_cfg_t sf_cfg{};
};

struct lsn_phy {
  std::unique_ptr<lsn::Engine> engine;
  // lsn_phy_create_multi: further engines (one per additional GPU) that share engine's sequential state; chunk g of a submit goes to engine g mod G
  std::vector<std::unique_ptr<lsn::Engine>> more;
  uint64_t multi_next = 0;  // round-robin position
  lsn::Engine* eng(size_t i) { return i == 0 ? engine.get() : more[i - 1].get(); }
  size_t neng() const { return 1 + more.size(); }
  lsn_phy_cfg_t cfg{};
  std::vector<std::unique_ptr<lsn_worker>> workers;
  std::deque<lsn_worker*> avail, pending;
  std::mutex mtx;
  std::condition_variable cv_avail, cv_pending, cv_idle, cv_release;
  std::thread dispatcher, releaser;
  bool stop = false, busy = false, flush = false, lossless = false;
  float* slab = nullptr;       // pinned [nof_workers][rx][3 * sflen] cf32
  size_t row_floats = 0;       // 3 * sflen * 2
  struct InFlight { hipEvent_t ev = nullptr; std::vector<lsn_worker*> ws; };
  std::deque<InFlight> inflight;   // batches whose rows are crossing PCIe (dispatcher -> releaser)
  std::vector<hipEvent_t> ev_free;
  int last_error = 0;

  void dispatch_loop();
  void release_loop();
  void stop_threads();
};

void lsn_phy::dispatch_loop()
{
  std::vector<lsn_worker*> batch;
  const uint32_t maxb = engine->maxBatch();
  const uint32_t target = std::max<uint32_t>(1, std::min<uint32_t>(maxb, (uint32_t)workers.size() / 2));
  long linger_us = 2000;  // lossless mode: how long a partial batch may wait for more subframes
  if (const char* e = getenv("LSN_POOL_LINGER_US")) linger_us = std::max(0, atoi(e));
  for (;;) {
    hipEvent_t ev = nullptr;
    {
      std::unique_lock<std::mutex> lk(mtx);
      cv_pending.wait(lk, [&] { return stop || !pending.empty(); });
      if (pending.empty() && stop) return;
      if (lossless && linger_us > 0) {
        // file replay: larger batches amortise the per-chunk launches; a partial batch goes out when the caller has run out of workers
        // (it is blocked in getAvail), when joinPending asks for a flush, or after the linger time
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(linger_us);
        cv_pending.wait_until(lk, deadline, [&] { return stop || flush || pending.size() >= target || avail.empty(); });
      }
      batch.clear();
      // consecutive TTIs and consecutive slab rows only: a batch is one contiguous run of subframes (start_tti + i) and one strided copy
      uint32_t prev_tti = 0, prev_idx = 0;
      while (!pending.empty() && batch.size() < maxb) {
        lsn_worker* w = pending.front();
        const uint32_t tti = w->sfn * 10 + w->sf_idx;
        if (!batch.empty() && (tti != (prev_tti + 1) % 10240 || w->update_meta || w->index != prev_idx + 1)) break;
        prev_tti = tti; prev_idx = w->index;
        batch.push_back(w);
        pending.pop_front();
      }
      busy = true;
      if (!ev_free.empty()) { ev = ev_free.back(); ev_free.pop_back(); }
    }
    (void)hipSetDevice(engine->device());
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
    // SubframeWorker::prepare's updateMetaFormats flag (LTESniffer_Core.cc:434) applies to the first subframe of the batch
    const uint32_t tti0 = batch[0]->sfn * 10 + batch[0]->sf_idx;
    const int r = ev ? engine->submitHostRows(batch[0]->buf[0], row_floats * sizeof(float), (uint32_t)batch.size(), tti0, batch[0]->update_meta != 0, ev)
                     : LSN_ERROR;
    {
      std::unique_lock<std::mutex> lk(mtx);
      if (r != LSN_SUCCESS) {
        last_error = r;
        for (auto* w : batch) avail.push_back(w);  // nothing was queued: the rows are free again
        if (ev) ev_free.push_back(ev);
      } else {
        inflight.push_back({ev, batch});
      }
      busy = false;
    }
    cv_release.notify_one();
    cv_avail.notify_all();
    cv_idle.notify_all();
  }
}

// returns the workers of a batch to the pool when the copy of their rows has completed
void lsn_phy::release_loop()
{
  for (;;) {
    InFlight f;
    {
      std::unique_lock<std::mutex> lk(mtx);
      cv_release.wait(lk, [&] { return stop || !inflight.empty(); });
      if (inflight.empty()) return;
      f = inflight.front();
    }
    (void)hipSetDevice(engine->device());
    for (;;) {  // poll + nap (hipEventSynchronize spins on this runtime)
      const hipError_t e = hipEventQuery(f.ev);
      if (e != hipErrorNotReady) { (void)hipGetLastError(); break; }
      timespec ts{0, 20000};
      nanosleep(&ts, nullptr);
    }
    {
      std::unique_lock<std::mutex> lk(mtx);
      inflight.pop_front();
      for (auto* w : f.ws) avail.push_back(w);
      ev_free.push_back(f.ev);
    }
    cv_avail.notify_all();
    cv_idle.notify_all();
    cv_pending.notify_all();
  }
}

void lsn_phy::stop_threads()
{
  {
    std::unique_lock<std::mutex> lk(mtx);
    stop = true;
  }
  cv_pending.notify_all();
  if (dispatcher.joinable()) dispatcher.join();
  cv_release.notify_all();
  if (releaser.joinable()) releaser.join();
  stop = false;
}

extern "C" {

int lsn_phy_create(const lsn_phy_cfg_t* cfg, lsn_phy_t** out)
{
  if (!cfg || !out) return LSN_ERROR_INVALID_INPUTS;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return LSN_ERROR_NO_DEVICE;  // no CPU fallback
  if (cfg->device < 0 || cfg->device >= ndev) return LSN_ERROR_INVALID_INPUTS;
  try {
    std::unique_ptr<lsn_phy> p(new lsn_phy());
    p->cfg = *cfg;
    if (p->cfg.nof_workers == 0) p->cfg.nof_workers = 20;  // Phy.h:19
    p->engine.reset(new lsn::Engine(p->cfg));
    *out = p.release();
    return LSN_SUCCESS;
  } catch (const std::invalid_argument&) {
    return LSN_ERROR_INVALID_INPUTS;
  } catch (const std::exception& ex) {
    fprintf(stderr, "ltesniffer_amd: %s\n", ex.what());
    return LSN_ERROR;
  }
}

int lsn_phy_create_multi(const lsn_phy_cfg_t* cfg, const int* devices, uint32_t n_devices, lsn_phy_t** out)
{
  if (!cfg || !out || !devices || n_devices < 1 || n_devices > 16) return LSN_ERROR_INVALID_INPUTS;
  lsn_phy_cfg_t c0 = *cfg;
  c0.device = devices[0];
  const int r = lsn_phy_create(&c0, out);
  if (r != LSN_SUCCESS) return r;
  int ndev = 0, caller_dev = 0;
  (void)hipGetDeviceCount(&ndev);
  (void)hipGetDevice(&caller_dev);
  struct RestoreDev { int d; ~RestoreDev() { (void)hipSetDevice(d); } } restore{caller_dev};  // the calling thread keeps its current device
  try {
    for (uint32_t i = 1; i < n_devices; i++) {
      if (devices[i] < 0 || devices[i] >= ndev) throw std::invalid_argument("device");
      lsn_phy_cfg_t ci = (*out)->cfg;
      ci.device = devices[i];
      (*out)->more.emplace_back(new lsn::Engine(ci, (*out)->engine->sharedState()));
      if (devices[i] != devices[0]) {  // peer copies of the IQ blocks (xGMI); harmless when already enabled
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devices[i], devices[0]) == hipSuccess && can) {
          (void)hipSetDevice(devices[i]);
          const hipError_t pe = hipDeviceEnablePeerAccess(devices[0], 0);
          if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
            fprintf(stderr, "ltesniffer_amd: peer access %d -> %d refused (%s): blocks resident on device %d are copied through the runtime's staging path\n", devices[i], devices[0], hipGetErrorString(pe), devices[0]);
            // harq_mode = 1: the soft-buffer pool lives on the first device and this engine's combine / copy kernels read and write it in place
            if (ci.harq_mode) throw std::invalid_argument("harq_mode on several GPUs needs peer access to the first device (soft-buffer pool)");
          }
          (void)hipGetLastError();
        } else {
          fprintf(stderr, "ltesniffer_amd: no peer access %d -> %d (hipDeviceCanAccessPeer): device-resident blocks take the runtime's staging path\n", devices[i], devices[0]);
          if (ci.harq_mode) throw std::invalid_argument("harq_mode on several GPUs needs peer access to the first device (soft-buffer pool)");
        }
      }
    }
    return LSN_SUCCESS;
  } catch (const std::exception& ex) {
    fprintf(stderr, "ltesniffer_amd: %s\n", ex.what());
    lsn_phy_destroy(*out);
    *out = nullptr;
    return LSN_ERROR_INVALID_INPUTS;
  }
}
uint32_t lsn_phy_nof_devices(lsn_phy_t* phy) { return phy ? (uint32_t)phy->neng() : 0; }

void lsn_phy_destroy(lsn_phy_t* phy)
{
  if (!phy) return;
  if (phy->dispatcher.joinable()) { (void)lsn_phy_join_pending(phy); phy->stop_threads(); }
  for (auto& w : phy->workers)
    for (auto& b : w->buf_offset) free(b);
  if (phy->slab) (void)hipHostFree(phy->slab);
  for (auto& e : phy->ev_free) (void)hipEventDestroy(e);
  delete phy;
}

int lsn_phy_set_cell(lsn_phy_t* phy, const lsn_cell_t* cell)
{
  if (!phy || !cell) return LSN_ERROR_INVALID_INPUTS;
  if (phy->dispatcher.joinable()) { (void)lsn_phy_join_pending(phy); phy->stop_threads(); }
  const int r = phy->engine->setCell(*cell);
  if (r != LSN_SUCCESS) return r;
  for (auto& e : phy->more) { const int re = e->setCell(*cell); if (re != LSN_SUCCESS) return re; }
  // worker pool: SubframeBuffer allocates 3 * SF_LEN per antenna (SubframeBuffer.cc:25); all of them in one pinned slab
  for (auto& w : phy->workers)
    for (auto& b : w->buf_offset) { free(b); b = nullptr; }
  phy->workers.clear(); phy->avail.clear(); phy->pending.clear();
  if (phy->slab) { (void)hipHostFree(phy->slab); phy->slab = nullptr; }
  const uint32_t sflen = phy->engine->sfLen(), A = phy->engine->nofRx();
  phy->row_floats = (size_t)3 * sflen * 2;
  if (hipHostMalloc((void**)&phy->slab, (size_t)phy->cfg.nof_workers * A * phy->row_floats * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); phy->slab = nullptr; return LSN_ERROR; }
  for (uint32_t i = 0; i < phy->cfg.nof_workers; i++) {
    std::unique_ptr<lsn_worker> w(new lsn_worker());
    w->phy = phy; w->buf_len = 3 * sflen; w->index = i;
    for (uint32_t rx = 0; rx < A; rx++) w->buf[rx] = phy->slab + ((size_t)i * A + rx) * phy->row_floats;
    phy->avail.push_back(w.get());
    phy->workers.push_back(std::move(w));
  }
  phy->lossless = false;
  phy->dispatcher = std::thread([phy] { phy->dispatch_loop(); });
  phy->releaser = std::thread([phy] { phy->release_loop(); });
  return LSN_SUCCESS;
}

lsn_worker_t* lsn_phy_get_avail(lsn_phy_t* phy, int blocking)
{
  if (!phy) return nullptr;
  std::unique_lock<std::mutex> lk(phy->mtx);
  phy->lossless = blocking != 0;  // Phy::getAvail (file replay, lossless) vs getAvailImmediate (live capture, lossy)
  if (blocking && phy->avail.empty()) {
    phy->cv_pending.notify_all();  // the dispatcher sends a partial batch when the caller is out of workers
    phy->cv_avail.wait(lk, [&] { return !phy->avail.empty() || phy->workers.empty(); });
  }
  if (phy->avail.empty()) return nullptr;  // getAvailImmediate: nullptr when none (Phy.cc:84-89)
  lsn_worker* w = phy->avail.front();
  phy->avail.pop_front();
  return w;
}

int lsn_phy_put_pending(lsn_phy_t* phy, lsn_worker_t* w)
{
  if (!phy || !w || w->phy != phy) return LSN_ERROR_INVALID_INPUTS;
  {
    std::unique_lock<std::mutex> lk(phy->mtx);
    phy->pending.push_back(w);
  }
  phy->cv_pending.notify_one();
  return LSN_SUCCESS;
}

int lsn_phy_join_pending(lsn_phy_t* phy)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  {
    std::unique_lock<std::mutex> lk(phy->mtx);
    phy->flush = true;
    phy->cv_pending.notify_all();
    phy->cv_idle.wait(lk, [&] { return phy->pending.empty() && !phy->busy && phy->inflight.empty(); });
    phy->flush = false;
  }
  const int w = phy->engine->wait();  // every queued subframe has been searched, decoded and written (Phy::joinPending, Phy.cc:100-109)
  std::unique_lock<std::mutex> lk(phy->mtx);
  const int r = phy->last_error ? phy->last_error : w;
  phy->last_error = 0;
  return r;
}

int lsn_phy_set_pdu_sink(lsn_phy_t* phy, lsn_pdu_sink_t cb, void* user)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  phy->engine->setSink(cb, user);
  for (auto& e : phy->more) e->setSink(cb, user);  // the writer turns keep the records of all engines in one order
  return LSN_SUCCESS;
}

int lsn_phy_set_pcap_writer(lsn_phy_t* phy, lsn_pcap_t* p)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  phy->engine->setSink(p ? lsn_pcap_sink : nullptr, p);
  for (auto& e : phy->more) e->setSink(p ? lsn_pcap_sink : nullptr, p);
  return LSN_SUCCESS;
}

int lsn_phy_set_api_mode(lsn_phy_t* phy, int api_mode, lsn_api_sink_t cb, void* user, lsn_pcap_t* api_pcap)
{
  if (!phy || api_mode < -1 || api_mode > 3) return LSN_ERROR_INVALID_INPUTS;
  phy->engine->setApi(api_mode, cb, user, api_pcap ? lsn_pcap_sink : nullptr, api_pcap);
  for (auto& e : phy->more) e->setApi(api_mode, cb, user, api_pcap ? lsn_pcap_sink : nullptr, api_pcap);
  return LSN_SUCCESS;
}

int lsn_phy_get_stats(lsn_phy_t* phy, lsn_blind_stats_t* out)
{
  if (!phy || !out) return LSN_ERROR_INVALID_INPUTS;
  phy->engine->getStats(out);
  return LSN_SUCCESS;
}

float lsn_phy_get_est_cfo(lsn_phy_t* phy) { return phy ? phy->engine->estCfo() : 0.0f; }
int lsn_phy_set_cfo_correction(lsn_phy_t* phy, int mode, float cfo_hz, float alpha)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  if (mode == 2 && !phy->more.empty()) return LSN_ERROR_INVALID_INPUTS;  // one loop over one stream of chunks: several engines share a capture chunk by chunk
  int rc = phy->engine->setCfoCorrection(mode, cfo_hz, alpha);
  for (auto& e : phy->more) if (rc == LSN_SUCCESS) rc = e->setCfoCorrection(mode, cfo_hz, alpha);
  return rc;
}
float lsn_phy_get_cfo_correction(lsn_phy_t* phy) { return phy ? phy->engine->cfoCorrection() : 0.0f; }
int lsn_phy_set_candidate_pruning(lsn_phy_t* phy, int mode)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  int rc = phy->engine->setCandidatePruning(mode);
  for (auto& e : phy->more) if (rc == LSN_SUCCESS) rc = e->setCandidatePruning(mode);
  return rc;
}

int lsn_phy_add_evergreen(lsn_phy_t* phy, uint16_t a, uint16_t b, uint32_t f)
{
  if (!phy || f >= lsn::NOF_FORMATS) return LSN_ERROR_INVALID_INPUTS;
  phy->engine->rntiManager().addEvergreen(a, b, f);
  return LSN_SUCCESS;
}
int lsn_phy_add_forbidden(lsn_phy_t* phy, uint16_t a, uint16_t b, uint32_t f)
{
  if (!phy || f >= lsn::NOF_FORMATS) return LSN_ERROR_INVALID_INPUTS;
  phy->engine->rntiManager().addForbidden(a, b, f);
  return LSN_SUCCESS;
}
int lsn_phy_setup_default_rnti_intervals(lsn_phy_t* phy)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  phy->engine->setupDefaultIntervals();
  return LSN_SUCCESS;
}
int lsn_phy_set_shortcut_discovery(lsn_phy_t* phy, int enable)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  phy->engine->searchRef().setShortcutDiscovery(enable != 0);
  return LSN_SUCCESS;
}
int lsn_phy_get_shortcut_discovery(lsn_phy_t* phy) { return phy && phy->engine->searchRef().getShortcutDiscovery() ? 1 : 0; }
int lsn_phy_set_histogram_threshold(lsn_phy_t* phy, uint32_t threshold)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  phy->engine->rntiManager().setHistogramThreshold(threshold);
  return LSN_SUCCESS;
}
int lsn_phy_print_stats(lsn_phy_t* phy, void* file)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  FILE* f = file ? (FILE*)file : stdout;
  lsn_blind_stats_t s;
  phy->engine->getStats(&s);
  const double us = phy->engine->searchTimeUs();
  fprintf(f, "nof_decoded_locations, nof_cce, nof_missed_cce, nof_subframes, nof_subframe_collisions_dw, nof_subframe_collisions_up, time, nof_locations\n");
  fprintf(f, "%u, %u, %u, %u, %u, %u, %ld.%06ld, %u\n", s.nof_decoded_locations, s.nof_cce, s.nof_missed_cce, s.nof_subframes, s.nof_subframe_collisions_dw,
          s.nof_subframe_collisions_up, (long)(us / 1e6), (long)us % 1000000L, s.nof_locations);
  return LSN_SUCCESS;
}
int lsn_phy_set_mcs_update_interval(lsn_phy_t* phy, uint32_t seconds)
{
  if (!phy || seconds > 4000000u) return LSN_ERROR_INVALID_INPUTS;
  phy->engine->setMcsUpdateInterval(seconds);
  return LSN_SUCCESS;
}
int lsn_phy_update_mcs_database(lsn_phy_t* phy)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  phy->engine->updateMcsDatabase();
  return LSN_SUCCESS;
}
uint32_t lsn_phy_nof_tracked_rnti(lsn_phy_t* phy) { return phy ? phy->engine->nofTrackedRnti() : 0; }
int lsn_phy_tracked_ul_modulation(lsn_phy_t* phy, uint16_t rnti) { return phy ? phy->engine->trackedModUl(rnti) : 0; }
uint32_t lsn_phy_nof_active_rnti(lsn_phy_t* phy) { return phy ? phy->engine->rntiManager().nofActive() : 0; }
int lsn_phy_get_meta_formats(lsn_phy_t* phy, uint32_t* primary, uint32_t* nof_primary, uint32_t* secondary, uint32_t* nof_secondary)
{
  if (!phy || !primary || !nof_primary || !secondary || !nof_secondary) return LSN_ERROR_INVALID_INPUTS;
  lsn::DCIMetaFormats& m = phy->engine->searchRef().metaFormats();
  *nof_primary = m.getNofPrimaryMetaFormats(); *nof_secondary = m.getNofSecondaryMetaFormats();
  for (uint32_t i = 0; i < *nof_primary && i < 9; i++) primary[i] = m.getPrimaryMetaFormats()[i]->global_index;
  for (uint32_t i = 0; i < *nof_secondary && i < 9; i++) secondary[i] = m.getSecondaryMetaFormats()[i]->global_index;
  return LSN_SUCCESS;
}
uint32_t lsn_phy_nof_workers(lsn_phy_t* phy) { return phy ? (uint32_t)phy->workers.size() : 0; }
lsn_worker_t* lsn_phy_worker(lsn_phy_t* phy, uint32_t index) { return phy && index < phy->workers.size() ? phy->workers[index].get() : nullptr; }
int lsn_phy_get_ue_config(lsn_phy_t* phy, uint16_t rnti, lsn_ue_config_t* out)
{
  if (!phy || !out) return LSN_ERROR_INVALID_INPUTS;
  const lsn::UeSpecConfig c = phy->engine->ueConfig(rnti);
  out->has_ue_config = c.has_ue_config ? 1u : 0u;
  out->p_a_db = c.p_a;
  out->i_offset_ack = c.i_offset_ack; out->i_offset_cqi = c.i_offset_cqi; out->i_offset_ri = c.i_offset_ri;
  out->cqi_type = c.cqi_type;
  return LSN_SUCCESS;
}

float** lsn_worker_buffers(lsn_worker_t* w) { return w ? w->buf : nullptr; }
float** lsn_worker_buffers_offset(lsn_worker_t* w)
{
  if (!w) return nullptr;
  for (auto& b : w->buf_offset)
    if (!b) b = (float*)calloc((size_t)w->buf_len * 2, sizeof(float));
  return w->buf_offset;
}
uint32_t lsn_worker_buffer_len(lsn_worker_t* w) { return w ? w->buf_len : 0; }
int lsn_worker_prepare(lsn_worker_t* w, uint32_t sf_idx, uint32_t sfn, int update_meta_formats, const lsn_dl_sf_cfg_t* sf)
{
  if (!w || sf_idx > 9) return LSN_ERROR_INVALID_INPUTS;
  w->sf_idx = sf_idx; w->sfn = sfn; w->update_meta = update_meta_formats;
  if (sf) w->sf_cfg = *sf;
  return LSN_SUCCESS;
}
uint32_t lsn_worker_sf_idx(lsn_worker_t* w) { return w ? w->sf_idx : 0; }
uint32_t lsn_worker_sfn(lsn_worker_t* w) { return w ? w->sfn : 0; }

// one capture over several GPUs: the block is cut into chunks of max_batch subframes, chunk g goes to engine g mod G
static int multi_submit(lsn_phy_t* phy, const void* d_iq, uint32_t n, uint32_t start_tti, uint32_t update_meta_period, void* stream)
{
  hipPointerAttribute_t attr{};
  int src = phy->engine->device();
  if (hipPointerGetAttributes(&attr, d_iq) == hipSuccess && attr.type == hipMemoryTypeDevice) src = attr.device;
  (void)hipGetLastError();
  const uint32_t mb = phy->engine->maxBatch();
  const size_t sf_stride = (size_t)phy->engine->nofRx() * phy->engine->sfLen() * 2 * sizeof(float);
  for (uint32_t base = 0; base < n; base += mb) {
    lsn::Engine* e = phy->eng((size_t)(phy->multi_next++ % phy->neng()));
    const int r = e->submitFrom((const uint8_t*)d_iq + (size_t)base * sf_stride, src, std::min(mb, n - base), start_tti + base, update_meta_period, (hipStream_t)stream);
    if (r != LSN_SUCCESS) return r;
  }
  return LSN_SUCCESS;
}
static int multi_wait(lsn_phy_t* phy)
{
  int rc = LSN_SUCCESS;
  for (size_t i = 0; i < phy->neng(); i++) { const int r = phy->eng(i)->wait(); if (r != LSN_SUCCESS) rc = r; }
  return rc;
}

int lsn_phy_process_device(lsn_phy_t* phy, const void* d_iq, uint32_t n, uint32_t start_tti, uint32_t update_meta_period, void* stream)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  if (!phy->more.empty()) { const int r = multi_submit(phy, d_iq, n, start_tti, update_meta_period, stream); const int w = multi_wait(phy); return r != LSN_SUCCESS ? r : w; }
  return phy->engine->process(d_iq, n, start_tti, update_meta_period, (hipStream_t)stream);
}
int lsn_phy_submit_device(lsn_phy_t* phy, const void* d_iq, uint32_t n, uint32_t start_tti, uint32_t update_meta_period, void* stream)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  if (!phy->more.empty()) return multi_submit(phy, d_iq, n, start_tti, update_meta_period, stream);
  return phy->engine->submit(d_iq, n, start_tti, update_meta_period, (hipStream_t)stream);
}
int lsn_phy_wait(lsn_phy_t* phy)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  if (!phy->more.empty()) return multi_wait(phy);
  return phy->engine->wait();
}
int lsn_phy_process_host(lsn_phy_t* phy, const float* iq, uint32_t n, uint32_t start_tti, uint32_t update_meta_period)
{
  if (!phy || (!iq && n)) return LSN_ERROR_INVALID_INPUTS;
  return phy->engine->processHost(iq, n, start_tti, update_meta_period);
}

int lsn_phy_process_host_int(lsn_phy_t* phy, const void* iq, uint32_t sample_format, float sample_scale, uint32_t n, uint32_t start_tti, uint32_t update_meta_period)
{
  if (!phy || (!iq && n) || (sample_format != LSN_FILE_SC16 && sample_format != LSN_FILE_SC8)) return LSN_ERROR_INVALID_INPUTS;
  return phy->engine->processHost(iq, n, start_tti, update_meta_period, sample_format, sample_scale);
}

int lsn_phy_mib_decode(lsn_phy_t* phy, const void* iq, int iq_on_device, lsn_mib_t* out)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  return phy->engine->mibDecode(iq, iq_on_device != 0, out, nullptr);
}
int lsn_phy_mib_decode_llr(lsn_phy_t* phy, const void* iq, int iq_on_device, lsn_mib_t* out, float* llr_raw480)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  return phy->engine->mibDecode(iq, iq_on_device != 0, out, llr_raw480);
}
int lsn_phy_process_file(lsn_phy_t* phy, const char* path, const lsn_file_cfg_t* cfg, uint32_t start_tti, uint64_t max_subframes, uint32_t update_meta_period,
                         uint64_t* subframes_done)
{
  if (!phy || !path || !cfg) return LSN_ERROR_INVALID_INPUTS;
  return phy->engine->processFile(path, *cfg, start_tti, max_subframes, update_meta_period, subframes_done);
}
int lsn_phy_prepare_file(lsn_phy_t* phy, uint32_t nof_antennas)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  return phy->engine->reserveFileBuffers(nof_antennas);
}
int lsn_phy_set_ul_config(lsn_phy_t* phy, const lsn_ul_cfg_t* cfg)
{
  if (!phy || !cfg) return LSN_ERROR_INVALID_INPUTS;
  return phy->engine->setUlConfig(*cfg);
}
static void sib2_to_c(const lsn::Sib2Config& c, lsn_sib2_t* o)
{
  o->n_sb = c.n_sb; o->hopping_mode = c.hopping_mode; o->pusch_hop_offset = c.pusch_hop_offset; o->enable_64qam = c.enable_64qam;
  o->group_hopping_enabled = c.group_hopping_enabled; o->group_assignment_pusch = c.group_assignment_pusch;
  o->sequence_hopping_enabled = c.sequence_hopping_enabled; o->cyclic_shift = c.cyclic_shift;
  o->root_seq_idx = c.root_seq_idx; o->prach_config_idx = c.prach_config_idx; o->high_speed_flag = c.high_speed_flag;
  o->zero_corr_zone = c.zero_corr_zone; o->prach_freq_offset = c.prach_freq_offset;
}
int lsn_phy_get_ul_config(lsn_phy_t* phy, lsn_ul_cfg_t* ul, lsn_sib2_t* sib2, uint32_t* from_sib2)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  lsn::Sib2Config s;
  if (!phy->engine->getUlConfig(ul, nullptr, &s)) return 0;
  if (sib2) sib2_to_c(s, sib2);
  if (from_sib2) *from_sib2 = phy->engine->sib2Learned() ? 1u : 0u;
  return 1;
}
int lsn_sib2_decode(const uint8_t* pdu, uint32_t len, lsn_sib2_t* out)
{
  if (!pdu || !out) return LSN_ERROR_INVALID_INPUTS;
  lsn::Sib2Config s;
  const int r = lsn::sib2_decode(pdu, (int)len, s);
  if (r == 2) sib2_to_c(s, out);
  return r;
}
int lsn_phy_pusch_decode(lsn_phy_t* phy, const void* ul_iq, int iq_on_device, uint32_t n_subframes, uint32_t start_tti,
                         const lsn_pusch_grant_t* grants, uint32_t n_grants, lsn_pusch_result_t* results, uint8_t* payloads, size_t payload_cap)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  return phy->engine->puschDecode(ul_iq, iq_on_device != 0, n_subframes, start_tti, grants, n_grants, results, payloads, payload_cap);
}
long lsn_phy_tap_ul(lsn_phy_t* phy, int what, uint32_t index, void* out, size_t cap)
{
  if (!phy || !out) return LSN_ERROR_INVALID_INPUTS;
  if (what == 2) return phy->engine->tapPrach(index, out, cap);
  return phy->engine->tapUl(what, index, out, cap);
}
int lsn_phy_set_prach_config(lsn_phy_t* phy, const lsn_prach_cfg_t* cfg)
{
  if (!phy || !cfg) return LSN_ERROR_INVALID_INPUTS;
  return phy->engine->setPrachConfig(*cfg);
}
int lsn_phy_prach_detect(lsn_phy_t* phy, const void* ul_iq, int iq_on_device, uint32_t n_subframes, uint32_t start_tti, lsn_prach_det_t* out, uint32_t cap)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  return phy->engine->prachDetect(ul_iq, iq_on_device != 0, n_subframes, start_tti, out, cap);
}
void lsn_phy_set_prach_sink(lsn_phy_t* phy, lsn_prach_sink_t cb, void* user)
{
  if (phy) phy->engine->setPrachSink(cb, user);
}
int lsn_prach_tti_opportunity(uint32_t config_idx, uint32_t tti) { return lsn::prach_tti_opportunity(config_idx, tti) ? 1 : 0; }

int lsn_cell_search(int device, const void* iq, int iq_on_device, uint64_t nof_samples, uint32_t nof_prb, const lsn_cell_search_cfg_t* cfg,
                    lsn_cell_search_t* out, float* corr_out)
{
  if (!iq || !cfg || !out) return LSN_ERROR_INVALID_INPUTS;
  try {
    return lsn::cell_search(device, (const cf32*)iq, iq_on_device != 0, nof_samples, nof_prb, *cfg, *out, corr_out);
  } catch (const std::exception& ex) {
    fprintf(stderr, "ltesniffer_amd: cell search: %s\n", ex.what());
    return LSN_ERROR;
  }
}

long lsn_phy_tap(lsn_phy_t* phy, int what, uint32_t sf, void* out, size_t cap)
{
  if (!phy || !out) return LSN_ERROR_INVALID_INPUTS;
  return phy->engine->tap(what, sf, out, cap);
}
int lsn_phy_set_stage_c_taps(lsn_phy_t* phy, int enable)
{
  if (!phy) return LSN_ERROR_INVALID_INPUTS;
  phy->engine->setStageCTaps(enable != 0);
  for (auto& e : phy->more) e->setStageCTaps(enable != 0);
  return LSN_SUCCESS;
}
int lsn_phy_get_perf(lsn_phy_t* phy, lsn_perf_t* out)
{
  if (!phy || !out) return LSN_ERROR_INVALID_INPUTS;
  phy->engine->getPerf(out);
  for (auto& e : phy->more) {  // counters and times of the other GPUs' engines add up (wall-clock fields: the longest one)
    lsn_perf_t p;
    e->getPerf(&p);
    out->ms_stage_a += p.ms_stage_a; out->ms_search += p.ms_search; out->ms_stage_c += p.ms_stage_c; out->ms_commit += p.ms_commit;
    out->ms_total = std::max(out->ms_total, p.ms_total);
    for (int k = 0; k < 16; k++) { out->kernel_ms[k] += p.kernel_ms[k]; out->kernel_launches[k] += p.kernel_launches[k]; }
    out->algo_bytes += p.algo_bytes; out->turbo_algo_bytes += p.turbo_algo_bytes; out->turbo128_algo_bytes += p.turbo128_algo_bytes;
    out->nof_tb_decodes += p.nof_tb_decodes; out->nof_cb_decodes += p.nof_cb_decodes; out->nof_turbo_iterations += p.nof_turbo_iterations;
    out->nof_candidates_decoded += p.nof_candidates_decoded; out->nof_ondemand_decodes += p.nof_ondemand_decodes; out->nof_pdus += p.nof_pdus;
    out->ms_search_core += p.ms_search_core; out->ms_rar += p.ms_rar; out->turbo_cyc_rm += p.turbo_cyc_rm; out->turbo_cyc_map += p.turbo_cyc_map;
    out->turbo_cyc_out += p.turbo_cyc_out; out->ms_wait_front += p.ms_wait_front; out->ms_wait_slot += p.ms_wait_slot; out->ms_drain = std::max(out->ms_drain, p.ms_drain);
    out->nof_turbo_iterations_run += p.nof_turbo_iterations_run; out->ms_ondemand_commit += p.ms_ondemand_commit;
    for (int k = 0; k < 4; k++) out->nof_ondemand_commit[k] += p.nof_ondemand_commit[k];
    for (int k = 0; k < 4; k++) out->nof_harq_combines[k] += p.nof_harq_combines[k];
    out->nof_candidate_misses += p.nof_candidate_misses;
    for (int k = 0; k < 3; k++) out->ms_harq[k] += p.ms_harq[k];
    out->nof_pusch_on_unverified_dmrs += p.nof_pusch_on_unverified_dmrs; out->nof_tb_on_derived_tbs += p.nof_tb_on_derived_tbs;
    out->nof_decode_jobs += p.nof_decode_jobs; out->nof_decode_jobs_used += p.nof_decode_jobs_used; out->nof_speculative_jobs += p.nof_speculative_jobs;
    for (int k = 0; k < 5; k++) { out->jobs_by_kind[k] += p.jobs_by_kind[k]; out->jobs_unused_by_kind[k] += p.jobs_unused_by_kind[k]; out->iters_by_kind[k] += p.iters_by_kind[k]; out->iters_unused_by_kind[k] += p.iters_unused_by_kind[k]; }
  }
  return LSN_SUCCESS;
}
const char* lsn_kernel_name(int k)
{
  static const char* names[LSN_K_COUNT] = {"k_ofdm", "k_chest", "k_chest_fin", "k_pcfich", "k_pdcch_llr", "k_cce_power", "k_viterbi",
                                           "k_pdsch_prep", "k_pdsch_demod", "k_turbo<64>", "k_rb_power", "k_turbo<128>", "k_rm"};
  return (k >= 0 && k < LSN_K_COUNT) ? names[k] : "";
}
const char* lsn_version(void) { return "ltesniffer_amd 0.1 (gfx950)"; }

}  // extern "C"
