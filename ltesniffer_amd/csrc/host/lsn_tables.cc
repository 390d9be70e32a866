// lsn_tables.cc - cell-constant device tables (twiddles, CRS, REG maps, scrambling, Viterbi gather ranks, PDSCH RE
// masks, Gold-sequence masks, CRC remainder tables) and device buffer allocation for the engine.
#include "lsn_engine.h"
#include "../../../spec/lte_tables.h"
#include "../kernels/lsn_turbo_core.h"
#include <cmath>
#include <cstdio>
#include <stdexcept>

#define HIP_CHECK(x)                                                                                       \
  do {                                                                                                     \
    hipError_t _e = (x);                                                                                   \
    if (_e != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #x); \
  } while (0)

namespace lsn {

// c(n) = x1(n+1600) ^ x2(n+1600), TS 36.211 7.2
void gold_sequence(uint32_t cinit, uint8_t* c, int len)
{
  uint32_t x1 = 1, x2 = cinit & 0x7FFFFFFFu;
  for (int n = 0; n < 1600 + len; n++) {
    if (n >= 1600) c[n - 1600] = (uint8_t)((x1 ^ x2) & 1u);
    x1 = (x1 >> 1) | ((((x1 >> 3) ^ x1) & 1u) << 30);
    x2 = (x2 >> 1) | ((((x2 >> 3) ^ (x2 >> 2) ^ (x2 >> 1) ^ x2) & 1u) << 30);
  }
}

static int fft_size_for(uint32_t nprb)
{
  switch (nprb) { case 6: return 128; case 15: return 256; case 25: return 512; case 50: return 1024; case 75: return 1536; case 100: return 2048; default: return -1; }
}

template <typename T>
static T* upload(std::vector<void*>& allocs, const std::vector<T>& v)
{
  void* d = nullptr;
  HIP_CHECK(hipMalloc(&d, v.size() * sizeof(T) + 16));
  HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  allocs.push_back(d);
  return (T*)d;
}
template <typename T>
static T* dalloc(std::vector<void*>& allocs, size_t n)
{
  void* d = nullptr;
  HIP_CHECK(hipMalloc(&d, n * sizeof(T) + 16));
  HIP_CHECK(hipMemset(d, 0, n * sizeof(T)));
  allocs.push_back(d);
  return (T*)d;
}

void Engine::freeDevice(bool keep_file_buffers)
{
  for (void* p : dev_allocs) (void)hipFree(p);
  dev_allocs.clear();
  for (void* p : host_allocs) (void)hipHostFree(p);
  host_allocs.clear();
  h_prune_ring = nullptr;   // (one of them)
  for (auto& ch : chunks) {
    for (auto& e : ch.ev_a)
      if (e) { (void)hipEventDestroy(e); e = nullptr; }
    ch = Chunk();
  }
  for (auto& r : runner_c) freeRunner(r);
  freeRunner(runner_s);
  freeRunner(runner_f);
  freeRunner(runner_k);
  freeRunner(runner_u);
  {
    auto df = [](auto*& p) { if (p) (void)hipFree(p); p = nullptr; };
    df(ul_d_iq); df(ul_d_grid); df(ul_d_hs); df(ul_d_stat); df(ul_d_grants);
    ul_iq_cap = ul_grid_cap = ul_hs_cap = ul_stat_cap = ul_grants_cap = 0;
    if (ul_h_grants) { (void)hipHostFree(ul_h_grants); ul_h_grants = nullptr; ul_h_grants_cap = 0; }
    if (ul_h_stat) { (void)hipHostFree(ul_h_stat); ul_h_stat = nullptr; ul_h_stat_cap = 0; }
    ul_set = false;
    df(prach.d_W); df(prach.d_V); df(prach.d_D); df(prach.d_Y); df(prach.d_corr); df(prach.d_out); df(prach.d_off);
    prach = Prach();
    df(mib_d_iq); df(mib_d_llr); df(mib_d_cand);
    // the file source's block buffers hold raw bytes, not cell tables: a cell that is set (again) keeps them - page-locking 1.5 GB costs more than replaying
    // 10 000 subframes, and a caller that reserved them (lsn_phy_prepare_file) and then sets its cell must not pay for them a second time inside its first replay
    if (!keep_file_buffers)
      for (auto& fb : file_buf) { if (fb.h_raw) (void)hipHostFree(fb.h_raw); df(fb.d_raw); df(fb.d_iq); fb.h_raw = nullptr; fb.bytes = 0; }
  }
  d_iq_staging = nullptr; staging_sf = 0;
  if (d_iq_raw) { (void)hipFree(d_iq_raw); d_iq_raw = nullptr; d_iq_raw_bytes = 0; }
  if (sh->harq_pool_owner == this) { d_harq_pool = nullptr; sh->harq_pool_owner = nullptr; }   // (freed with this engine's device allocations)
  if (harq_h_store) { (void)hipHostFree(harq_h_store); harq_h_store = nullptr; harq_h_store_cap = 0; }
  if (harq_d_store) { (void)hipFree(harq_d_store); harq_d_store = nullptr; harq_d_store_cap = 0; }
  if (d_harq_scratch) { (void)hipFree(d_harq_scratch); d_harq_scratch = nullptr; harq_scratch_cap = 0; harq_scratch_n = 0; }
  harq_cache.clear(); harq_touched.clear();
  for (auto& ch : chunks) { if (ch.d_keep) (void)hipFree(ch.d_keep); ch.d_keep = nullptr; ch.keep_cap = ch.keep_n = 0; }
  last_chunk = nullptr;
}

void Engine::freeRunner(JobRunner& r)
{
  auto df = [](auto*& p) { if (p) (void)hipFree(p); p = nullptr; };
  auto hf = [](auto*& p) { if (p) (void)hipHostFree(p); p = nullptr; };
  df(r.d_jobs); df(r.d_cbs); df(r.d_cbres); df(r.d_prefix); df(r.d_llr16); df(r.d_payload); df(r.d_spp); df(r.d_items);
  hf(r.h_items_pinned); hf(r.h_payload_pinned); hf(r.h_cbres_pinned); hf(r.h_jobs_pinned); hf(r.h_cbs_pinned);
  r.items_cap = r.h_items_cap = r.spp_cap = r.jobs_cap = r.cbs_cap = r.cbres_cap = r.prefix_cap = r.llr16_cap = r.payload_cap = r.h_payload_cap = r.h_cbres_cap = r.h_jobs_cap = r.h_cbs_cap = 0;
  for (auto& e : r.ev)
    if (e) { (void)hipEventDestroy(e); e = nullptr; }
  if (r.ev_done) { (void)hipEventDestroy(r.ev_done); r.ev_done = nullptr; }
  if (r.stream) { (void)hipStreamDestroy(r.stream); r.stream = nullptr; }
}

void Engine::allocRunner(JobRunner& r)
{
  // the search thread's runner (on-demand RAR decodes) sits on the critical path of the sequential search: high priority
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (&r == &runner_s || &r == &runner_f || &r == &runner_u || &r == &runner_k) {
    HIP_CHECK(hipStreamCreateWithPriority(&r.stream, hipStreamNonBlocking, hi));
  } else {
    // bulk decode streams: lowest priority.  (Round 2 masked a few CUs out of them instead; CU-masked streams stand still while host -> device copies
    // are in flight on this runtime and cost 12 % of the resident rate: DESIGN 3.1, point 5.)
    HIP_CHECK(hipStreamCreateWithPriority(&r.stream, hipStreamNonBlocking, lo));
  }
  for (auto& e : r.ev) HIP_CHECK(hipEventCreate(&e));
  HIP_CHECK(hipEventCreateWithFlags(&r.ev_done, hipEventDisableTiming));  // waited for with Engine's poll-and-sleep waitEvent()
  // The arenas of a bulk decode runner start at the size a full chunk of a loaded cell needs (per subframe at 100 PRB: 16 decode calls, 32 code
  // blocks, 0.5 M soft bits, 128 K packed words, 24 KB of payload), scaled with the bandwidth: a fresh engine otherwise grows each of them several
  // times during its first chunks, every step a stream synchronisation + hipFree (a device-wide wait) - part of why the cold pass of a capture
  // ran 10-15 % below the following ones.  On-demand runners (single decodes by the search, front and commit threads) start at four subframes' worth
  // (12 MB each): they grew from zero until the last session of round 6 - some forty steps spread over the first TWO passes of a capture (a runner that
  // meets its first large grant in the second pass starts its ladder there), which is what kept the second pass of the PCIe-inclusive legs 10-17 %
  // below the third (profiles/r06_exp_int16_chunks.txt; LSN_HOST_DEBUG prints every step).
  bool bulk = false;
  for (int i = 0; i < NDEC; i++) bulk = bulk || &r == &runner_c[i];
  if (!getenv("LSN_NO_PRESIZE")) {
    const double scale = (double)cell.nof_prb / 100.0;
    // The pre-size is a start-up optimisation, never a requirement (round-4 advisor finding): it is bounded by the chunk a runner can meet
    // (max_batch) AND by a share of the memory that is free right now - all bulk runners together take at most a quarter of it - and an
    // allocation that fails anyway leaves the arena to the lazy growth of runJobs (grow_dev / grow_host) instead of failing setCell.
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
    const double per_sf_bytes = (16.0 * (sizeof(LsnGrantDev) + 2.0 * (14 * 100 + 16)) + 32.0 * (sizeof(LsnCbDev) + sizeof(LsnCbRes)) + 512.0 * 1024 * 2 + 24.0 * 1024 + 64 * 4 + 128.0 * 1024 * 4) * scale;
    size_t sfn_ = bulk ? max_batch : std::min<size_t>(max_batch, 4);
    if (free_b) sfn_ = std::min<size_t>(sfn_, (size_t)((double)free_b / 4.0 / (double)NDEC / per_sf_bytes));
    auto dev = [&](auto*& p, size_t& cap, double per_sf) {
      const size_t n = (size_t)(per_sf * scale * (double)sfn_) + 4096;
      if (hipMalloc((void**)&p, n * sizeof(*p)) != hipSuccess) { (void)hipGetLastError(); p = nullptr; cap = 0; return; }
      cap = n;
    };
    auto host = [&](auto*& p, size_t& cap, double per_sf) {
      const size_t n = (size_t)(per_sf * scale * (double)sfn_) + 4096;
      if (hipHostMalloc((void**)&p, n * sizeof(*p), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); p = nullptr; cap = 0; return; }
      cap = n;
    };
    dev(r.d_jobs, r.jobs_cap, 16); dev(r.d_cbs, r.cbs_cap, 32); dev(r.d_cbres, r.cbres_cap, 32);
    dev(r.d_prefix, r.prefix_cap, 16.0 * (14 * 100 + 16)); dev(r.d_llr16, r.llr16_cap, 512.0 * 1024); dev(r.d_payload, r.payload_cap, 24.0 * 1024);
    dev(r.d_items, r.items_cap, 64); dev(r.d_spp, r.spp_cap, 128.0 * 1024);
    host(r.h_cbres_pinned, r.h_cbres_cap, 32); host(r.h_payload_pinned, r.h_payload_cap, 24.0 * 1024); host(r.h_jobs_pinned, r.h_jobs_cap, 16);
    host(r.h_cbs_pinned, r.h_cbs_cap, 32); host(r.h_items_pinned, r.h_items_cap, 64);
  }
}

template <typename T>
static T* halloc(std::vector<void*>& allocs, size_t n)
{
  void* h = nullptr;
  HIP_CHECK(hipHostMalloc(&h, n * sizeof(T) + 16, hipHostMallocCoherent | hipHostMallocMapped));  // fine-grained: GPU stores / loads go straight to host memory (copy kernels, lsn_dev.h)
  allocs.push_back(h);
  return (T*)h;
}

void Engine::allocChunk(Chunk& ch)
{
  const size_t B = max_batch, A = dlRx(), P = cell.nof_ports;
  if (cfg.sniffer_mode == 1) ch.d_ul_grid = dalloc<cf32>(dev_allocs, B * 14 * cd.nre);
  ch.d_grid = dalloc<cf32>(dev_allocs, B * A * 14 * cd.nre);
  ch.d_ce = dalloc<cf32>(dev_allocs, B * P * A * 14 * cd.nre);
  ch.d_chest_raw = dalloc<float>(dev_allocs, B * A * P * 8);
  ch.d_chest = dalloc<LsnChest>(dev_allocs, B);
  ch.d_cfi = dalloc<uint32_t>(dev_allocs, B);
  ch.d_sfidx = dalloc<uint32_t>(dev_allocs, B);
  ch.d_prune_snap = dalloc<uint32_t>(dev_allocs, LSN_PRUNE_SNAP_WORDS);
  ch.d_acc = dalloc<uint8_t>(dev_allocs, B * LSN_MAX_LOC * LSN_MAX_SIZES);   // (zeroed: the slots of sizes the cell does not have are never written and must read 0)
  ch.d_dphi = dalloc<uint32_t>(dev_allocs, B);
  ch.d_pcfich_corr = dalloc<float>(dev_allocs, B * 3);
  ch.d_llr = dalloc<float>(dev_allocs, B * LSN_LLR_STRIDE);
  ch.d_ccepow = dalloc<float>(dev_allocs, B * LSN_CCE_STRIDE);
  ch.d_cand = dalloc<LsnCand>(dev_allocs, B * LSN_MAX_LOC * LSN_MAX_SIZES);
  ch.d_cand4 = dalloc<uint32_t>(dev_allocs, B * LSN_MAX_LOC * LSN_MAX_SIZES);
  ch.d_rbp = dalloc<float>(dev_allocs, B * 128);
  ch.d_rbp_part = dalloc<float>(dev_allocs, B * 14 * 128);   // per-symbol terms of the PRB power, written by k_ofdm (zeroed once: rows 12, 13 of an extended-CP cell stay zero)
  ch.h_cand = halloc<LsnCand>(host_allocs, B * LSN_MAX_LOC * LSN_MAX_SIZES);
  ch.h_cand4 = halloc<uint32_t>(host_allocs, B * LSN_MAX_LOC * LSN_MAX_SIZES);
  ch.h_ccepow = halloc<float>(host_allocs, B * LSN_CCE_STRIDE);
  ch.h_chest = halloc<LsnChest>(host_allocs, B);
  ch.h_cfi = halloc<uint32_t>(host_allocs, B);
  ch.h_rbp = halloc<float>(host_allocs, B * 128);
  ch.h_sfidx = halloc<uint32_t>(host_allocs, B);
  ch.h_dphi = halloc<uint32_t>(host_allocs, B);
  ch.ctx.assign(B, SubframeCtx());
  for (auto& e : ch.ev_a) HIP_CHECK(hipEventCreate(&e));
  (void)hipEventDestroy(ch.ev_a[16]);  // the "stage A results are on the host" marker is only waited for, never timed
  HIP_CHECK(hipEventCreateWithFlags(&ch.ev_a[16], hipEventDisableTiming));
}

void Engine::buildTables()
{
  const uint32_t nprb = cell.nof_prb, P = cell.nof_ports, id = cell.id;
  const int N = fft_size_for(nprb);
  cd = LsnCellDev{};
  cd.nof_prb = nprb; cd.nof_ports = P; cd.id = id; cd.nof_rx = dlRx(); cd.iq_nant = cfg.nof_rx_antennas;
  const int Nsub = N == 1536 ? 512 : N;  // 15 MHz: three interleaved 512-point transforms + a radix-3 combination
  cd.N = (uint32_t)N; cd.nsub = (uint32_t)Nsub; cd.lgN = 0; while ((1 << cd.lgN) < Nsub) cd.lgN++;
  cd.nre = 12 * nprb; cd.nref = 2 * nprb; cd.sflen = 15u * (uint32_t)N;
  cd.cp = cell.cp; cd.nsym = cell.nsym(); cd.nslot = cell.nslot();
  cd.reg_w6 = 1u | (P == 4 ? 2u : 0u) | (cell.cp ? 8u : 0u);
  // FFT twiddles and NCO tables (double -> float, same generation as the oracle's definition)
  std::vector<cf32> tw((size_t)Nsub / 2), coarse(4096), fine(1024);
  for (int k = 0; k < Nsub / 2; k++) { double a = 2.0 * M_PI * k / Nsub; tw[k] = {(float)std::cos(a), (float)(-std::sin(a))}; }
  if (N != Nsub) {
    std::vector<cf32> t3((size_t)N);
    for (int k = 0; k < N; k++) { double a = 2.0 * M_PI * k / N; t3[k] = {(float)std::cos(a), (float)(-std::sin(a))}; }
    cd.twiddle3 = upload(dev_allocs, t3);
  }
  for (int k = 0; k < 4096; k++) { double a = 2.0 * M_PI * k / 4096.0; coarse[k] = {(float)std::cos(a), (float)std::sin(a)}; }
  for (int k = 0; k < 1024; k++) { double a = 2.0 * M_PI * k / 4194304.0; fine[k] = {(float)std::cos(a), (float)std::sin(a)}; }
  cd.twiddle = upload(dev_allocs, tw); cd.nco_coarse = upload(dev_allocs, coarse); cd.nco_fine = upload(dev_allocs, fine);
  // CRS values for every subframe index (36.211 6.10.1.1)
  {
    const float s = 0.70710678118654752440f;
    std::vector<cf32> crs((size_t)10 * P * 4 * cd.nref);
    std::vector<uint8_t> c(440);
    // ports 0 / 1 share pilot symbols 0, 4, 7, 11 (extended CP: 0, 3, 6, 9) and their sequences; ports 2 / 3 share symbols 1, 8 (1, 7): rows 0, 1 of their
    // [4][nref] block.  c_init carries N_CP = 1 (normal) / 0 (extended), 36.211 6.10.1.1
    const uint32_t nsl = cell.nslot();
    for (uint32_t sf = 0; sf < 10; sf++)
      for (uint32_t p0 = 0; p0 < P; p0 += 2)
        for (int q = 0; q < (p0 == 0 ? 4 : 2); q++) {
          const uint32_t l = p0 == 0 ? (uint32_t)(q >> 1) * nsl + ((q & 1) ? nsl - 3 : 0) : (uint32_t)q * nsl + 1, ns = 2 * sf + (l >= nsl ? 1 : 0), lsl = l % nsl;
          gold_sequence(1024u * (7u * (ns + 1) + lsl + 1) * (2u * id + 1) + 2u * id + (cell.cp ? 0u : 1u), c.data(), 440);
          for (uint32_t m = 0; m < cd.nref; m++) {
            const uint32_t mp = m + 110 - nprb;
            cf32 v{c[2 * mp] ? -s : s, c[2 * mp + 1] ? -s : s};
            for (uint32_t p = p0; p < p0 + 2 && p < P; p++) crs[((sf * P + p) * 4 + q) * cd.nref + m] = v;
          }
        }
    cd.crs = upload(dev_allocs, crs);
  }
  // REGs: PCFICH (36.211 6.7.4), PHICH (6.9.3, normal duration), PDCCH quadruplet -> REG map (6.8.5)
  {
    const int nre = (int)cd.nre, n0 = nre / 6;
    std::vector<uint8_t> used0((size_t)n0, 0);
    const int kbar = 6 * (int)(id % (2 * nprb));
    for (int i = 0; i < 4; i++) { int k = (kbar + (i * (int)nprb / 2) * 6) % nre; cd.pcfich_k0[i] = (uint32_t)k; used0[k / 6] = 1; }
    const int ng = (int)((cell.phich_ng_x6 * nprb + 47) / 48);
    std::vector<int> avail;
    for (int i = 0; i < n0; i++) if (!used0[i]) avail.push_back(i);
    const int na = (int)avail.size();
    for (int m = 0; m < ng; m++) for (int i = 0; i < 3; i++) used0[avail[((int)id + m + (i * na) / 3) % na]] = 1;
    std::vector<uint16_t> rk(3 * 800, 0);
    std::vector<uint8_t> rl(3 * 800, 0);
    std::vector<uint16_t> rq(3 * 800, 0xFFFFu);  // natural REG index: symbol by symbol, nre / 6 REGs in a symbol that carries CRS (cd.reg_w6), nre / 4 in the others
    auto wof = [&](int l) { return ((cd.reg_w6 >> l) & 1u) ? 6 : 4; };
    for (int cfi = 1; cfi <= 3; cfi++) {
      const int nsym = cfi + (nprb <= 10 ? 1 : 0);
      std::vector<std::pair<uint16_t, uint8_t>> regs;
      for (int k = 0; k < nre; k++)
        for (int l = 0; l < nsym; l++) {
          const int w = wof(l);
          if (k % w || (l == 0 && used0[k / 6])) continue;
          regs.push_back({(uint16_t)k, (uint8_t)l});
        }
      const int M = (int)regs.size(), R = (M + 31) / 32, ND = 32 * R - M;
      std::vector<int> perm;
      for (int j = 0; j < 32; j++) for (int r = 0; r < R; r++) { int idx = r * 32 + lsn_perm_cc[j]; if (idx >= ND) perm.push_back(idx - ND); }
      cd.nof_regs[cfi - 1] = (uint32_t)M; cd.nof_cce[cfi - 1] = (uint32_t)(M / 9);
      for (int mp = 0; mp < M; mp++) {
        const int q = perm[(mp + (int)id) % M];
        if (q < 800) {
          rk[(cfi - 1) * 800 + q] = regs[mp].first; rl[(cfi - 1) * 800 + q] = regs[mp].second;
          const int l = regs[mp].second;
          int nat = regs[mp].first / wof(l);
          for (int j = 0; j < l; j++) nat += nre / wof(j);
          if (nat < 800) rq[(cfi - 1) * 800 + nat] = (uint16_t)q;
        }
      }
    }
    cd.reg_k0 = upload(dev_allocs, rk); cd.reg_l = upload(dev_allocs, rl); cd.reg_q = upload(dev_allocs, rq);
  }
  // scrambling sequences of the control region
  {
    std::vector<uint8_t> scr((size_t)10 * LSN_LLR_STRIDE), pscr(10 * 32);
    for (uint32_t sf = 0; sf < 10; sf++) {
      gold_sequence(sf * 512u + id, scr.data() + (size_t)sf * LSN_LLR_STRIDE, LSN_LLR_STRIDE);
      gold_sequence((sf + 1) * (2u * id + 1) * 512u + id, pscr.data() + sf * 32, 32);
    }
    cd.pdcch_scr = upload(dev_allocs, scr); cd.pcfich_scr = upload(dev_allocs, pscr);
  }
  // Gaussian smoothing taps: srsran_chest_set_smooth_filter_gauss(order 4, std 1) [srsRAN], SubframeWorker.cc:381-383
  {
    float sum = 0.0f;
    for (int i = 0; i < 5; i++) { float d = (float)(i - 2); cd.taps[i] = expf(-(d * d) / 2.0f); }
    for (int i = 0; i < 5; i++) sum = sum + cd.taps[i];
    const float inv = 1.0f / sum;
    for (int i = 0; i < 5; i++) cd.taps[i] = cd.taps[i] * inv;
  }
  // distinct DCI sizes + the de-rate-matching rank of every Viterbi input position (36.212 5.1.4.2)
  {
    search->setCell(cell, cd.nof_cce);
    std::vector<uint32_t> sizes(search->sizes(), search->sizes() + search->nofSizes());
    cd.nsizes = (uint32_t)sizes.size();
    for (size_t i = 0; i < sizes.size(); i++) cd.sizes[i] = sizes[i];
    std::vector<uint16_t> rank((size_t)LSN_MAX_SIZES * 3 * LSN_MAX_DCI_D, 0);
    for (size_t si = 0; si < sizes.size(); si++) {
      const int D = (int)sizes[si] + 16, R = (D + 31) / 32, KP = 32 * R, ND = KP - D;
      int nonnull = 0;
      for (int s = 0; s < 3; s++)
        for (int col = 0; col < 32; col++)
          for (int r = 0; r < R; r++) {
            const int idx = r * 32 + lsn_perm_cc[col];
            if (idx >= ND) rank[si * 3 * LSN_MAX_DCI_D + 3 * (idx - ND) + s] = (uint16_t)nonnull++;
          }
    }
    cd.rankmap = upload(dev_allocs, rank);
    // CRC16 weights of the payload bits (k_viterbi computes the remainder lane-parallel): x^(n - 1 - i + 16) mod (x^16 + x^12 + x^5 + 1)
    std::vector<uint16_t> cw((size_t)(LSN_MAX_SIZES + 1) * 64, 0);
    auto fill = [&](uint16_t* row, uint32_t n) {
      uint32_t w = 1;                       // x^0
      for (int t = 0; t < 16; t++) { w <<= 1; if (w & 0x10000u) w ^= 0x11021u; }   // x^16 mod g: the weight of the LAST payload bit
      for (int i = (int)n - 1; i >= 0; i--) {
        if (i < 64) row[i] = (uint16_t)w;
        w <<= 1; if (w & 0x10000u) w ^= 0x11021u;
      }
    };
    for (uint32_t k = 0; k < cd.nsizes; k++) fill(cw.data() + (size_t)k * 64, cd.sizes[k]);
    fill(cw.data() + (size_t)LSN_MAX_SIZES * 64, 24);
    cd.crc16_w = upload(dev_allocs, cw);
    std::vector<uint16_t> prank(120, 0);  // PBCH block: 24 + 16 bits
    {
      const int D = 40, R = 2, KP = 64, ND = KP - D;
      int nonnull = 0;
      for (int s = 0; s < 3; s++)
        for (int col = 0; col < 32; col++)
          for (int r = 0; r < R; r++) {
            const int idx = r * 32 + lsn_perm_cc[col];
            if (idx >= ND) prank[3 * (idx - ND) + s] = (uint16_t)nonnull++;
          }
    }
    cd.pbch_rank = upload(dev_allocs, prank);
  }
  // PDSCH-capable RE masks per (subframe class, symbol, PRB)
  {
    std::vector<uint16_t> vm((size_t)3 * 14 * nprb, 0);
    const uint32_t cls_sf[3] = {0, 5, 1};
    for (int cl = 0; cl < 3; cl++)
      for (uint32_t l = 0; l < 14; l++)
        for (uint32_t prb = 0; prb < nprb; prb++) {
          uint16_t m = 0;
          for (uint32_t kk = 0; kk < 12; kk++) if (pdsch_re_usable(cell, cls_sf[cl], l, 12 * prb + kk)) m |= (uint16_t)(1u << kk);
          vm[((size_t)cl * 14 + l) * nprb + prb] = m;
        }
    cd.validmask = upload(dev_allocs, vm);
  }
  // Gold sequence as a table: x1 part + the GF(2)-linear map cinit -> x2(n+1600)
  {
    std::vector<uint8_t> x1v(LSN_GOLD_LEN);
    std::vector<uint32_t> m2(LSN_GOLD_LEN);
    uint32_t x1 = 1;
    uint32_t st[31];
    for (int i = 0; i < 31; i++) st[i] = 1u << i;  // st[i] = mask of cinit bits that make up x2(n+i)
    int head = 0;
    for (int n = 0; n < 1600 + LSN_GOLD_LEN; n++) {
      if (n >= 1600) { x1v[n - 1600] = (uint8_t)(x1 & 1u); m2[n - 1600] = st[head]; }
      x1 = (x1 >> 1) | ((((x1 >> 3) ^ x1) & 1u) << 30);
      const uint32_t nm = st[(head + 3) % 31] ^ st[(head + 2) % 31] ^ st[(head + 1) % 31] ^ st[head];
      st[head] = nm;  // slot of x2(n) becomes x2(n+31)
      head = (head + 1) % 31;
    }
    cd.gold_x1 = upload(dev_allocs, x1v); cd.gold_x2mask = upload(dev_allocs, m2);
  }
  // x^j mod g for the two 24-bit CRCs
  {
    std::vector<uint32_t> ta(6144), tb(6144);
    uint32_t a = 1, b = 1;
    for (int j = 0; j < 6144; j++) {
      ta[j] = a; tb[j] = b;
      a <<= 1; if (a & 0x1000000u) a ^= 0x1864CFBu;
      b <<= 1; if (b & 0x1000000u) b ^= 0x1800063u;
    }
    cd.crc_tab_a = upload(dev_allocs, ta); cd.crc_tab_b = upload(dev_allocs, tb);
  }
  // interleaver address tables of the turbo decoder, one per block size (kernels/lsn_turbo_core.h)
  {
    std::vector<uint32_t> il(turbo_il_offset(0) + 8);
    for (int i = 0; i < LSN_QPP_NSIZES; i++) {
      const int K = lsn_qpp_table[i][0];
      lsn_turbo_il_fill(il.data() + turbo_il_offset(K), K, lsn_qpp_table[i][1], lsn_qpp_table[i][2]);
    }
    cd.turbo_il = upload(dev_allocs, il);
  }
  if (cfg.harq_mode) {  // soft buffers of 300 entities x 8 processes x 2 transport blocks, 16 code blocks of K = 6144 each: 1.9 GB of the 288
    // (several engines on one capture: the first one to get here owns the pool, the others reach it over the peer link)
    if (!sh->harq_pool_owner || sh->harq_pool_owner == this) {
      d_harq_pool = dalloc<uint32_t>(dev_allocs, (size_t)HarqDatabase::NENT * HarqDatabase::NPID * 2 * HARQ_SLOT_WORDS);
      sh->harq_pool_owner = this;
      harq_db = HarqDatabase();
      harq_keep.clear();
    }
  }
  h_prune_ring = halloc<uint32_t>(host_allocs, (size_t)PRUNE_RING * LSN_PRUNE_SNAP_WORDS);
  std::memset(h_prune_ring, 0, (size_t)PRUNE_RING * LSN_PRUNE_SNAP_WORDS * sizeof(uint32_t));
  prune_pub.store(0);
  publishPruneSnapshot();
  // pipeline slots, decode runners, staging
  for (int i = 0; i < nslots; i++) allocChunk(chunks[i]);
  for (int i = 0; i < ndec; i++) allocRunner(runner_c[i]);
  allocRunner(runner_s);
  allocRunner(runner_f);
  allocRunner(runner_k);
  staging_sf = (size_t)max_batch * 12;  // three staging blocks of four chunks each (processHost)
  d_iq_staging = dalloc<cf32>(dev_allocs, staging_sf * cfg.nof_rx_antennas * cd.sflen);
}

}  // namespace lsn
