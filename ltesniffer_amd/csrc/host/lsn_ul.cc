// lsn_ul.cc - uplink half of the engine: cell-constant uplink tables and the batched PUSCH grant decoder
// (srsran_enb_ul_fft + srsran_chest_ul_estimate_pusch + srsran_pusch_decode for a list of grants,
// /root/reference/src/src/UL_Sniffer_PUSCH.cc:250-262,389-392).  The UL-mode orchestration around it (ULSchedule,
// modulation-table trials, Msg3 grants) is host logic that calls this entry point.
// Product code: no CPU fallback, nothing from oracle/ is included or linked.
#include "lsn_engine.h"
#include "../kernels/lsn_rm.h"
#include "../../../spec/lte_tables.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <stdexcept>

#define HIP_CHECK(x)                                                                                       \
  do {                                                                                                     \
    hipError_t _e = (x);                                                                                   \
    if (_e != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #x); \
  } while (0)

namespace lsn {

static int largest_prime_below(int n)
{
  for (int p = n - 1; p >= 2; p--) {
    bool ok = true;
    for (int d = 2; d * d <= p; d++)
      if (p % d == 0) { ok = false; break; }
    if (ok) return p;
  }
  return 2;
}

template <typename T>
static T* upload_vec(std::vector<void*>& allocs, const std::vector<T>& v)
{
  void* d = nullptr;
  HIP_CHECK(hipMalloc(&d, v.size() * sizeof(T) + 16));
  HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  allocs.push_back(d);
  return (T*)d;
}

// 7.5 kHz shift of the SC-FDMA demodulator: needed by the uplink OFDM stage of every chunk, also before SIB2 is known
void Engine::uploadUlStatic()
{
  if (cd.ul_shift) return;
  const int N = (int)cd.N;
  std::vector<cf32> sh((size_t)N);
  for (int n = 0; n < N; n++) { const double a = M_PI * n / N; sh[n] = {(float)std::cos(a), (float)(-std::sin(a))}; }
  cd.ul_shift = upload_vec(dev_allocs, sh);
}

bool Engine::getUlConfig(lsn_ul_cfg_t* u, lsn_prach_cfg_t* p, Sib2Config* sib) const
{
  if (!ul_set) return false;
  if (u) *u = ul_cfg;
  if (p) { *p = prach.cfg; if (!prach.set) p->config_idx = 0xFFFFFFFFu; }
  if (sib) *sib = sib2;
  return true;
}

// srsran_enb_ul_set_cell with the DMRS configuration of SIB2 (SubframeWorker.cc:258-262, ULSchedule.cc:140-158): the configuration becomes the
// shared one (every engine of a multi-GPU capture follows it, syncUlConfig), this engine's device tables are built now
int Engine::setUlConfig(const lsn_ul_cfg_t& u)
{
  if (!cell_set || u.cyclic_shift > 7 || u.delta_ss > 29) return LSN_ERROR_INVALID_INPUTS;
  const int r = buildUlTables(u);
  if (r != LSN_SUCCESS) return r;
  ul_cfg = u;
  ul_set = true;
  sib2_learned = false;
  ul_tables_epoch = ul_cfg_epoch.fetch_add(1, std::memory_order_release) + 1;
  return LSN_SUCCESS;
}

// commit turn of an UL_MODE chunk: another engine of the capture (or the caller, through engine 0) may have (re)configured the uplink since this
// engine built its tables
void Engine::syncUlConfig()
{
  const uint32_t ep = ul_cfg_epoch.load(std::memory_order_acquire);
  if (ul_set && ul_tables_epoch != ep) {
    if (buildUlTables(ul_cfg) != LSN_SUCCESS) throw std::runtime_error("UL_MODE: the shared uplink configuration could not be applied on this device");
    ul_tables_epoch = ep;
  }
  if (sh->prach_cfg_set && prach_tables_epoch != sh->prach_epoch) {
    const uint32_t pe = sh->prach_epoch;
    const lsn_prach_cfg_t pc = sh->prach_cfg;
    if (setPrachConfig(pc) != LSN_SUCCESS) prach.set = false;
    sh->prach_epoch = pe;          // (setPrachConfig counted a new epoch: this was a replay of the shared one)
    prach_tables_epoch = pe;
  }
}

int Engine::buildUlTables(const lsn_ul_cfg_t& u)
{
  if (!cell_set || u.cyclic_shift > 7 || u.delta_ss > 29) return LSN_ERROR_INVALID_INPUTS;
  try {
    HIP_CHECK(hipSetDevice(cfg.device));
    const int N = (int)cd.N;
    (void)N;
    std::vector<cf32> ph(12), base, idft;
    for (int m = 0; m < 12; m++) { const double a = 2.0 * M_PI * m / 12.0; ph[m] = {(float)std::cos(a), (float)std::sin(a)}; }
    const uint32_t fss = ((cell.id % 30u) + u.delta_ss) % 30u;  // f_ss^PUSCH
    // Base sequences r_{u,v}(n) of every allocation size for all 30 sequence groups and both sequence numbers (36.211 5.5.1.1 / 5.5.1.2):
    // variant (2 u + v) of size L starts at (2 u + v) * ul_base_stride + ul_off[L]; v = 1 only exists from 6 PRB on (else a copy of v = 0).
    ul_off.assign(111, -1);
    uint32_t stride = 0;
    for (uint32_t L = 1; L <= cell.nof_prb; L++) {
      if (!ul_valid_prb(L)) continue;
      ul_off[L] = (int)stride; stride += 12 * L;
    }
    ul_base_stride = stride;
    base.resize((size_t)60 * stride);
    for (uint32_t L = 1; L <= cell.nof_prb; L++) {
      if (ul_off[L] < 0) continue;
      const int M = 12 * (int)L, Nzc = largest_prime_below(M);
      for (uint32_t uu = 0; uu < 30; uu++)
        for (uint32_t vv = 0; vv < 2; vv++) {
          cf32* dst = base.data() + (size_t)(2 * uu + vv) * stride + ul_off[L];
          const double qb = (double)Nzc * (double)(uu + 1) / 31.0;
          long long q = (long long)std::floor(qb + 0.5);
          if (vv && M >= 72) q += ((long long)std::floor(2.0 * qb) & 1) ? -1 : 1;
          for (int n = 0; n < M; n++) {
            if (L <= 2) {  // one / two PRB: the tabulated sequences exp(j phi(n) pi / 4) (Tables 5.5.1.2-1 / -2, spec/lte_tables.h)
              const double a = M_PI * (double)(L == 1 ? lsn_dmrs_phi12[uu][n] : lsn_dmrs_phi24[uu][n]) / 4.0;
              dst[n] = {(float)std::cos(a), (float)std::sin(a)};
              continue;
            }
            const long long m = n % Nzc;
            const double a = M_PI * (double)((q * m * (m + 1)) % (2ll * Nzc)) / (double)Nzc;
            dst[n] = {(float)std::cos(a), (float)(-std::sin(a))};
          }
        }
      for (int k = 0; k < M; k++) { const double a = 2.0 * M_PI * k / M; idft.push_back({(float)std::cos(a), (float)std::sin(a)}); }
    }
    // sequence group u(ns) and sequence number v(ns) of the 20 slots (5.5.1.3 / 5.5.1.4)
    {
      std::vector<uint8_t> cg(8 * 20 + 8), cs(32);
      gold_sequence(cell.id / 30u, cg.data(), 8 * 20);
      gold_sequence(((cell.id / 30u) << 5) + fss, cs.data(), 20);
      for (uint32_t ns = 0; ns < 20; ns++) {
        uint32_t fgh = 0;
        if (u.group_hopping_enabled) { for (int i = 0; i < 8; i++) fgh += (uint32_t)cg[8 * ns + i] << i; fgh %= 30u; }
        ul_u[ns] = (uint8_t)((fgh + fss) % 30u);
        ul_v[ns] = (uint8_t)((!u.group_hopping_enabled && u.sequence_hopping_enabled) ? cs[ns] : 0);
      }
    }
    uploadUlStatic();
    cd.ul_ph12 = upload_vec(dev_allocs, ph);
    cd.ul_base = upload_vec(dev_allocs, base);
    cd.ul_idft = upload_vec(dev_allocs, idft);
    // n_PN(ns) of 36.211 5.5.2.1.1
    std::vector<uint8_t> c(8 * 7 * 20 + 8);
    gold_sequence(((cell.id / 30u) << 5) + fss, c.data(), (int)c.size());
    for (uint32_t ns = 0; ns < 20; ns++) {
      uint32_t npn = 0;
      for (int i = 0; i < 8; i++) npn += (uint32_t)c[8 * (int)cell.nslot() * ns + i] << i;   // n_PN(ns) = sum c(8 N_symb^UL ns + i) 2^i (36.211 5.5.2.1.1)
      ul_npn[ns] = npn;
    }
    cell.pusch_hop_offset = u.hopping_offset;  // n_rb_ho of the DCI 0 -> grant conversion from now on (SubframeWorker.cc:271-277)
    search->setPuschHopOffset(u.hopping_offset);
    if (!runner_u.stream) allocRunner(runner_u);
    return LSN_SUCCESS;
  } catch (const std::exception& ex) {
    fprintf(stderr, "ltesniffer_amd: %s\n", ex.what());
    return LSN_ERROR;
  }
}

template <typename T>
static void grow_d(T*& p, size_t& cap, size_t need)
{
  if (need <= cap) return;
  HIP_CHECK(hipDeviceSynchronize());
  if (p) HIP_CHECK(hipFree(p));
  cap = need + need / 2 + 1024;
  HIP_CHECK(hipMalloc((void**)&p, cap * sizeof(T)));
}

int Engine::puschDecode(const void* ul_iq, bool on_device, uint32_t nsf, uint32_t start_tti, const lsn_pusch_grant_t* grants, uint32_t ngrants,
                        lsn_pusch_result_t* results, uint8_t* payloads, size_t payload_cap)
{
  if (!cell_set || !ul_set) return LSN_ERROR;
  if ((!ul_iq && nsf) || (!grants && ngrants) || (!results && ngrants)) return LSN_ERROR_INVALID_INPUTS;
  try {
    HIP_CHECK(hipSetDevice(cfg.device));
    hipStream_t st = runner_u.stream;
    const cf32* d_iq = (const cf32*)ul_iq;
    if (!on_device) {
      grow_d(ul_d_iq, ul_iq_cap, (size_t)nsf * cd.sflen);
      HIP_CHECK(hipMemcpyAsync(ul_d_iq, ul_iq, (size_t)nsf * cd.sflen * sizeof(cf32), hipMemcpyHostToDevice, st));
      d_iq = ul_d_iq;
    }
    grow_d(ul_d_grid, ul_grid_cap, (size_t)nsf * 14 * cd.nre);
    lsn_launch_ul_fft(cd, d_iq, 1, 0, ul_d_grid, nsf, st);
    std::vector<uint8_t> pay;
    puschDecodeGrid(ul_d_grid, nsf, start_tti, grants, ngrants, results, pay);
    if (payloads && pay.size() > payload_cap) return LSN_ERROR_INVALID_INPUTS;  // results[].payload_off would point past the caller's buffer
    if (payloads) std::memcpy(payloads, pay.data(), pay.size());
    return LSN_SUCCESS;
  } catch (const std::exception& ex) {
    fprintf(stderr, "ltesniffer_amd: %s\n", ex.what());
    return LSN_ERROR;
  }
}

// srsran_chest_ul_estimate_pusch + srsran_pusch_decode for a list of grants on an uplink grid that is already on the device
// (stream runner_u).  payload_out: tbs/8 bytes per valid grant at results[i].payload_off.  Throws on HIP errors.
void Engine::puschDecodeGrid(const cf32* d_grid, uint32_t nsf, uint32_t start_tti, const lsn_pusch_grant_t* grants, uint32_t ngrants,
                             lsn_pusch_result_t* results, std::vector<uint8_t>& payload_out)
{
  static const uint32_t n_dmrs1[8] = {0, 2, 3, 4, 6, 8, 9, 10};  // 36.211 Table 5.5.2.1.1-2
  static const uint32_t n_dmrs2[8] = {0, 6, 3, 4, 2, 8, 10, 9};  // Table 5.5.2.1.1-1
  JobRunner& r = runner_u;
  hipStream_t st = r.stream;
  payload_out.clear();
  std::vector<LsnUlGrantDev> gd;
  std::vector<int> gidx;
  r.h_cbs.clear();
  struct TbRef { uint32_t grant, cb_first, cb_count, pay_off; int tbs; };
  std::vector<TbRef> tbs;
  size_t hs_n = 0, llr_n = 0, pay_n = 0;
  for (uint32_t i = 0; i < ngrants; i++) {
    const lsn_pusch_grant_t& g = grants[i];
    results[i] = lsn_pusch_result_t{};
    const uint32_t n_prb2 = g.hop == 1 ? g.n_prb_slot1 : g.n_prb;
    // (no uint32 wrap-around in the range checks: n_prb <= nof_prb first, then L_prb against what is left)
    const bool ok = g.sf < nsf && g.L_prb >= 1 && g.L_prb <= cell.nof_prb && ul_valid_prb(g.L_prb) && g.n_prb <= cell.nof_prb - g.L_prb && n_prb2 <= cell.nof_prb - g.L_prb && g.hop <= 1 &&
                    g.tbs > 0 && (g.tbs % 8) == 0 &&
                    (g.mod == 2 || g.mod == 4 || g.mod == 6 || g.mod == 8) && ul_off[g.L_prb] >= 0 && g.rv >= 0 && g.rv < 4;
    if (!ok) continue;
    const uint32_t M = 12 * g.L_prb, sf_idx = (start_tti + g.sf) % 10;
    const uint32_t C = 2 * (cell.nslot() - 1);  // N_symb^PUSCH: 12 SC-FDMA data symbols per subframe, 10 with the extended CP (= columns of the channel interleaver)
    CbSegm s;
    if (!cbsegm((int)g.tbs, s)) continue;
    // control resources Q' = min(ceil(O M_sc N_symb beta / sum K_r), cap) (36.212 5.2.2.6), beta in eighths
    const long long sumK = (long long)s.Cp * s.Kp + (long long)s.Cm * s.Km;
    auto qprime = [&](uint32_t O, long long beta8, uint32_t cap) -> uint32_t {
      if (!O) return 0u;
      const long long q = ((long long)O * M * C * beta8 + 8 * sumK - 1) / (8 * sumK);
      return (uint32_t)std::min<long long>(q, cap);
    };
    if (g.nof_ack > 2 || g.ri_bits > 2 || g.cqi_bits > 64) continue;
    // beta offsets of the UE (36.213 Tables 8.6.3-1/-2/-3, spec/lte_tables.h), the reference's defaults when the grant does not name them
    const uint32_t ia = g.beta_offset_ack_idx_p1 ? g.beta_offset_ack_idx_p1 - 1 : 10u, ic = g.beta_offset_cqi_idx_p1 ? g.beta_offset_cqi_idx_p1 - 1 : 8u,
                   ir = g.beta_offset_ri_idx_p1 ? g.beta_offset_ri_idx_p1 - 1 : 11u;
    if (ia > 15 || ic > 15 || ir > 15) continue;
    if ((g.nof_ack && !lsn_beta_ack8[ia]) || (g.ri_bits && !lsn_beta_ri8[ir]) || (g.cqi_bits && !lsn_beta_cqi8[ic])) continue;  // reserved index
    const uint32_t q_ack = qprime(g.nof_ack, lsn_beta_ack8[ia], 4 * M), q_ri = qprime(g.ri_bits, lsn_beta_ri8[ir], 4 * M);
    const uint32_t q_cqi = g.cqi_bits ? qprime(g.cqi_bits + (g.cqi_bits > 11 ? 8u : 0u), lsn_beta_cqi8[ic], C * M - q_ri) : 0u;
    if (q_ri + q_cqi >= C * M) continue;
    const int G = (int)((C * M - q_ri - q_cqi) * g.mod);
    LsnUlGrantDev d{};
    d.sf = g.sf; d.n_prb = g.n_prb; d.n_prb2 = n_prb2; d.L_prb = g.L_prb; d.qm = g.mod;
    for (uint32_t sl = 0; sl < 2; sl++) d.ncs[sl] = (n_dmrs1[ul_cfg.cyclic_shift & 7] + n_dmrs2[g.n_dmrs & 7] + ul_npn[2 * sf_idx + sl]) % 12u;
    d.cinit = ((uint32_t)g.rnti << 14) | (sf_idx << 9) | cell.id;
    for (uint32_t sl = 0; sl < 2; sl++) {
      const uint32_t ns = 2 * sf_idx + sl, var = 2u * ul_u[ns] + (g.L_prb >= 6 ? ul_v[ns] : 0u);
      (sl ? d.base_off1 : d.base_off) = var * ul_base_stride + (uint32_t)ul_off[g.L_prb];
    }
    d.idft_off = (uint32_t)ul_off[g.L_prb];
    d.hs_off = (uint32_t)hs_n; hs_n += 2 * M;
    d.llr_off = (uint32_t)llr_n; llr_n += ((size_t)G + 7) & ~(size_t)7;
    d.scale = 1.0f / sqrtf((float)M);
    d.q_ack = q_ack; d.q_ri = q_ri; d.q_cqi = q_cqi;
    // UL-SCH: one transport block, one layer (36.212 5.2.2.6)
    const int Qm = (int)g.mod, Gp = G / Qm, gamma = Gp % s.C;
    TbRef ref{i, (uint32_t)r.h_cbs.size(), (uint32_t)s.C, (uint32_t)pay_n, (int)g.tbs};
    int rp = 0;
    uint32_t wp = 0;
    for (int q = 0; q < s.C; q++) {
      LsnCbDev cb{};
      const int K = q < s.Cm ? s.Km : s.Kp, F = q == 0 ? s.F : 0;
      int E = (q <= s.C - gamma - 1) ? Qm * (Gp / s.C) : Qm * ((Gp + s.C - 1) / s.C);
      if (rp + E > G) E = G - rp;
      cb.e_off = d.llr_off + (uint32_t)rp; cb.E = (uint32_t)E; cb.K = (uint32_t)K; cb.F = (uint32_t)F; cb.rv = (uint32_t)g.rv;
      cb.crc_b = s.C > 1 ? 1u : 0u;
      cb.out_bytes = (uint32_t)(K - F - (s.C > 1 ? 24 : 0)) / 8;
      cb.out_off = (uint32_t)pay_n + wp;
      cb.il_off = turbo_il_offset(K);
      cb.nwin = turbo_nwin(K);
      cb.max_iter = (uint32_t)cfg.max_turbo_iterations;
      cb.dep = LSN_CB_NODEP;  // every code block is decoded: the iteration count of a grant is part of what lsn_phy_pusch_decode reports
      wp += cb.out_bytes;
      rp += E;
      r.h_cbs.push_back(cb);
    }
    pay_n += (wp + 15) & ~15u;
    tbs.push_back(ref);
    gd.push_back(d);
    gidx.push_back((int)i);
  }
  ul_last_gd = gd; ul_last_idx = gidx;
  const uint32_t ng = (uint32_t)gd.size(), ncb = (uint32_t)r.h_cbs.size();
  if (!ng) { HIP_CHECK(hipStreamSynchronize(st)); return; }
  grow_d(ul_d_grants, ul_grants_cap, ng);
  grow_d(ul_d_hs, ul_hs_cap, hs_n);
  grow_d(ul_d_stat, ul_stat_cap, (size_t)2 * ng);
  grow_d(r.d_llr16, r.llr16_cap, llr_n + 8);
  grow_d(r.d_cbs, r.cbs_cap, ncb);
  grow_d(r.d_cbres, r.cbres_cap, ncb);
  grow_d(r.d_payload, r.payload_cap, pay_n + 16);
  std::vector<uint32_t> order(ncb);
  uint32_t n128 = 0, kmax128 = 0, kmax64 = 0;
  for (uint32_t i = 0; i < ncb; i++) { r.h_cbs[i].res_idx = i; order[i] = i; }
  std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
    const uint32_t kx = r.h_cbs[x].K, ky = r.h_cbs[y].K;
    const bool bx = lsn_turbo_two_wave_class((int)kx), by = lsn_turbo_two_wave_class((int)ky);
    if (bx != by) return bx;
    if (kx != ky) return kx > ky;
    return x < y;
  });
  std::vector<LsnCbDev> sorted(ncb);
  size_t spp_n = 0;
  uint32_t emax = 0;
  for (uint32_t i = 0; i < ncb; i++) {
    sorted[i] = r.h_cbs[order[i]];
    sorted[i].spp_off = (uint32_t)spp_n; spp_n += LSN_SPP_WORDS(sorted[i].K);
    emax = std::max(emax, sorted[i].E);
    if (lsn_turbo_two_wave_class((int)sorted[i].K)) { n128++; kmax128 = std::max(kmax128, sorted[i].K); } else kmax64 = std::max(kmax64, sorted[i].K);
  }
  grow_d(r.d_spp, r.spp_cap, spp_n + 16);
  // descriptors go through pinned mirrors and the upload kernel, not through the host -> device copy engine (its FIFO may hold IQ blocks, lsn_dev.h)
  if (ng > ul_h_grants_cap) {
    HIP_CHECK(hipStreamSynchronize(st));
    if (ul_h_grants) HIP_CHECK(hipHostFree(ul_h_grants));
    ul_h_grants_cap = ng + ng / 2 + 64;
    HIP_CHECK(hipHostMalloc((void**)&ul_h_grants, ul_h_grants_cap * sizeof(LsnUlGrantDev), hipHostMallocCoherent | hipHostMallocMapped));
  }
  if (ncb > r.h_cbs_cap) {
    HIP_CHECK(hipStreamSynchronize(st));
    if (r.h_cbs_pinned) HIP_CHECK(hipHostFree(r.h_cbs_pinned));
    r.h_cbs_cap = ncb + ncb / 2 + 1024;
    HIP_CHECK(hipHostMalloc((void**)&r.h_cbs_pinned, r.h_cbs_cap * sizeof(LsnCbDev), hipHostMallocCoherent | hipHostMallocMapped));
  }
  std::memcpy(ul_h_grants, gd.data(), ng * sizeof(LsnUlGrantDev));
  std::memcpy(r.h_cbs_pinned, sorted.data(), ncb * sizeof(LsnCbDev));
  lsn_launch_upload(ul_d_grants, ul_h_grants, ng * sizeof(LsnUlGrantDev), st);
  lsn_launch_upload(r.d_cbs, r.h_cbs_pinned, ncb * sizeof(LsnCbDev), st);
  HIP_CHECK(hipMemsetAsync(r.d_llr16, 0, llr_n * sizeof(int16_t), st));
  lsn_launch_pusch_chest(cd, ul_d_grants, d_grid, ul_d_hs, ul_d_stat, ng, st);
  lsn_launch_pusch_demod(cd, ul_d_grants, d_grid, ul_d_hs, ul_d_stat, r.d_llr16, ng, st);
  lsn_launch_rm(r.d_cbs, r.d_llr16, r.d_spp, ncb, emax, st);
  lsn_launch_turbo(cd, r.d_cbs, r.d_spp, r.d_payload, r.d_cbres, n128, kmax128, ncb - n128, kmax64, st, nullptr);
  // results come back through pinned mirrors written by the copy kernel (lsn_dev.h), not through the copy engine
  auto grow_pinned = [&](auto*& p, size_t& cap, size_t need) {
    if (need <= cap) return;
    HIP_CHECK(hipStreamSynchronize(st));
    if (p) HIP_CHECK(hipHostFree(p));
    cap = need + need / 2 + 1024;
    HIP_CHECK(hipHostMalloc((void**)&p, cap * sizeof(*p), hipHostMallocCoherent | hipHostMallocMapped));
  };
  grow_pinned(r.h_cbres_pinned, r.h_cbres_cap, ncb);
  grow_pinned(r.h_payload_pinned, r.h_payload_cap, pay_n + 16);
  grow_pinned(ul_h_stat, ul_h_stat_cap, (size_t)2 * ng);
  lsn_launch_download(r.h_cbres_pinned, r.d_cbres, ncb * sizeof(LsnCbRes), st);
  lsn_launch_download(r.h_payload_pinned, r.d_payload, pay_n, st);
  lsn_launch_download(ul_h_stat, ul_d_stat, (size_t)2 * ng * sizeof(float), st);
  HIP_CHECK(hipStreamSynchronize(st));
  const LsnCbRes* cbres = r.h_cbres_pinned;
  const uint8_t* pay_base = r.h_payload_pinned;
  const float* stat = ul_h_stat;
  for (size_t t = 0; t < tbs.size(); t++) {
    const TbRef& ref = tbs[t];
    bool all_ok = true;
    uint32_t rem = 0, iters = 0;
    uint64_t bits_after = 0;
    for (int q = (int)ref.cb_count - 1; q >= 0; q--) {
      const LsnCbRes& cr = cbres[ref.cb_first + q];
      all_ok = all_ok && cr.ok != 0;
      iters += cr.iters;
      rem ^= crc24a_mulmod(cr.rem_a, crc24a_xpow(bits_after));
      bits_after += 8ull * r.h_cbs[ref.cb_first + q].out_bytes;
    }
    const uint8_t* pl = pay_base + ref.pay_off;
    const uint32_t par = ((uint32_t)pl[ref.tbs / 8] << 16) | ((uint32_t)pl[ref.tbs / 8 + 1] << 8) | pl[ref.tbs / 8 + 2];
    lsn_pusch_result_t& res = results[ref.grant];
    res.crc_ok = all_ok && rem == 0 && par != 0 && bits_after == (uint64_t)ref.tbs + 24;
    res.iterations = iters;
    res.snr_db = 10.0f * log10f(stat[2 * t + 1] / stat[2 * t]);
    res.payload_off = (uint32_t)payload_out.size();
    payload_out.insert(payload_out.end(), pl, pl + ref.tbs / 8);
  }
}

long Engine::tapUl(int what, uint32_t index, void* out, size_t cap)
{
  if (!ul_set) return LSN_ERROR_INVALID_INPUTS;
  size_t n = 0;
  const void* src = nullptr;
  if (what == 0) { n = (size_t)14 * cd.nre * sizeof(cf32); src = ul_d_grid + (size_t)index * 14 * cd.nre; }  // UL grid of subframe `index`
  else if (what == 1) {  // rate-matched LLRs (UL-SCH order) of grant `index` of the last call
    const LsnUlGrantDev* d = nullptr;
    for (size_t i = 0; i < ul_last_idx.size(); i++)
      if ((uint32_t)ul_last_idx[i] == index) d = &ul_last_gd[i];
    if (!d) return LSN_ERROR_INVALID_INPUTS;
    n = ((size_t)2 * (cell.nslot() - 1) * 12 * d->L_prb - d->q_ri - d->q_cqi) * d->qm * sizeof(int16_t); src = runner_u.d_llr16 + d->llr_off;
  } else return LSN_ERROR_INVALID_INPUTS;
  if (n > cap) return LSN_ERROR_INVALID_INPUTS;
  if (hipMemcpy(out, src, n, hipMemcpyDeviceToHost) != hipSuccess) return LSN_ERROR;
  return (long)n;
}

}  // namespace lsn
