// lsn_prach.cc - PRACH preamble detection on the uplink antenna (PUSCH_Decoder::set_rach_config / work_prach,
// /root/reference/src/src/UL_Sniffer_PUSCH.cc:640-713: srsran_prach_init + srsran_prach_set_cfg +
// srsran_prach_set_detect_factor(60), then per uplink subframe srsran_prach_tti_opportunity and
// srsran_prach_detect_offset on the samples behind the cyclic prefix).  Preamble format 0 (36.211 5.7: T_CP = 3168 Ts,
// T_SEQ = 24576 Ts, N_ZC = 839, K = 12, phi = 7), unrestricted cyclic-shift set.  Tables (cos/sin in double) are host work;
// the DFT bins, the root correlations and the window peaks are the kernels of stage_ul.hip.
// Product code: no CPU fallback, nothing from oracle/ is included or linked.
#include "lsn_engine.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

#define HIP_CHECK(x)                                                                                       \
  do {                                                                                                     \
    hipError_t _e = (x);                                                                                   \
    if (_e != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #x); \
  } while (0)

namespace lsn {

static const int NZC = 839;
static const uint16_t kPrachNcs[16] = {0, 13, 15, 18, 22, 26, 32, 38, 46, 59, 76, 93, 119, 167, 279, 419};  // 36.211 Table 5.7.2-2

// 36.211 Table 5.7.1-2, preamble format 0 = PRACH configuration index 0..15 (srsran_prach_tti_opportunity(p, tti, -1))
bool prach_tti_opportunity(uint32_t config_idx, uint32_t tti)
{
  static const uint16_t sf_mask[16] = {0x002, 0x010, 0x080, 0x002, 0x010, 0x080, 0x042, 0x084, 0x108, 0x092, 0x124, 0x248, 0x155, 0x2AA, 0x3FF, 0x200};
  if (config_idx > 15) return false;
  const uint32_t sfn = tti / 10, sf = tti % 10;
  const bool even_only = config_idx < 3 || config_idx == 15;
  if (even_only && (sfn & 1u)) return false;
  return (sf_mask[config_idx] >> sf) & 1u;
}

template <typename T>
static void regrow(T*& p, size_t& cap, size_t need)
{
  if (need <= cap) return;
  HIP_CHECK(hipDeviceSynchronize());
  if (p) HIP_CHECK(hipFree(p));
  cap = need + need / 2 + 64;
  HIP_CHECK(hipMalloc((void**)&p, cap * sizeof(T)));
}

int Engine::setPrachConfig(const lsn_prach_cfg_t& p)
{
  // one-subframe formats only (work_prach copies SF_LEN samples, UL_Sniffer_PUSCH.cc:680-686); restricted sets are not built
  if (!cell_set || p.config_idx > 15 || p.zero_corr_zone > 15 || p.root_seq_idx > 837 || p.hs_flag != 0 || p.freq_offset + 6 > cell.nof_prb)
    return LSN_ERROR_INVALID_INPUTS;
  try {
    HIP_CHECK(hipSetDevice(cfg.device));
    if (!runner_u.stream) allocRunner(runner_u);
    prach.cfg = p;
    prach.cfg.zc_roots = nullptr;
    prach.factor = p.detect_factor > 0.0f ? p.detect_factor : 60.0f;
    prach.ncs = kPrachNcs[p.zero_corr_zone];
    prach.nwin = prach.ncs ? NZC / prach.ncs : 1;
    prach.nroots = (64 + prach.nwin - 1) / prach.nwin;
    prach.N12 = 12 * (int)cd.N;
    prach.Ncp = 3168 * (int)cd.N / 2048;
    prach.b0 = 7 + 12 * (12 * (int)p.freq_offset - 6 * (int)cell.nof_prb) + 6;  // phi + K (k0 + 1/2)
    std::vector<cf32> W((size_t)prach.N12), V(NZC), D((size_t)prach.nroots * NZC);
    for (int i = 0; i < prach.N12; i++) { const double a = 2.0 * M_PI * (double)i / (double)prach.N12; W[i] = {(float)std::cos(a), (float)(-std::sin(a))}; }
    std::vector<double> vr(NZC), vi(NZC), xr(NZC), xi(NZC);
    for (int m = 0; m < NZC; m++) {
      const double a = 2.0 * M_PI * (double)m / (double)NZC;
      vr[m] = std::cos(a); vi[m] = std::sin(a);
      V[m] = {(float)vr[m], (float)vi[m]};
    }
    for (uint32_t i = 0; i < prach.nroots; i++) {
      const uint32_t lr = (p.root_seq_idx + i) % 838u;
      const uint32_t u = p.zc_roots ? p.zc_roots[lr] : lr + 1;  // 36.211 Table 5.7.2-4 comes from the caller
      if (u < 1 || u > 838) return LSN_ERROR_INVALID_INPUTS;
      for (int n = 0; n < NZC; n++) {  // x_u(n) = exp(-j pi u n (n+1) / N_ZC)
        const long long ph = ((long long)u * n % (2 * NZC)) * (n + 1) % (2 * NZC);
        const double a = M_PI * (double)ph / (double)NZC;
        xr[n] = std::cos(a); xi[n] = -std::sin(a);
      }
      for (int k = 0; k < NZC; k++) {  // its 839-point DFT (sequential sums in double)
        double sr = 0.0, si = 0.0;
        for (int n = 0; n < NZC; n++) {
          const int m = (int)((long long)n * k % NZC);
          sr = sr + (xr[n] * vr[m] + xi[n] * vi[m]);
          si = si + (xi[n] * vr[m] - xr[n] * vi[m]);
        }
        D[(size_t)i * NZC + k] = {(float)sr, (float)si};
      }
    }
    auto up = [&](cf32*& d, const std::vector<cf32>& v) {
      if (d) HIP_CHECK(hipFree(d));
      HIP_CHECK(hipMalloc((void**)&d, v.size() * sizeof(cf32)));
      HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(cf32), hipMemcpyHostToDevice));
    };
    HIP_CHECK(hipDeviceSynchronize());
    up(prach.d_W, W); up(prach.d_V, V); up(prach.d_D, D);
    prach.set = true;
    sh->prach_cfg = prach.cfg; sh->prach_cfg_set = true;   // the other engines of a multi-GPU capture follow at their next commit turn (syncUlConfig)
    prach_tables_epoch = ++sh->prach_epoch;
    return LSN_SUCCESS;
  } catch (const std::exception& ex) {
    fprintf(stderr, "ltesniffer_amd: %s\n", ex.what());
    return LSN_ERROR;
  }
}

// detection on the subframes of d_iq ([sf][nant][sflen], antenna `ant`) that are PRACH occasions; stream runner_u.  Throws on HIP errors.
void Engine::prachDetectDev(const cf32* d_iq, uint32_t nant, uint32_t ant, uint32_t nsf, uint32_t start_tti, std::vector<lsn_prach_det_t>& out, uint32_t first_sf)
{
  out.clear();
  std::vector<uint64_t> off;
  std::vector<uint32_t> occ_sf;
  for (uint32_t s = first_sf; s < nsf; s++)
    if (prach_tti_opportunity(prach.cfg.config_idx, (start_tti + s) % 10240u)) { off.push_back(((uint64_t)s * nant + ant) * cd.sflen); occ_sf.push_back(s); }
  const uint32_t nocc = (uint32_t)off.size();
  prach.last_nocc = nocc;
  if (!nocc) return;
  hipStream_t st = runner_u.stream;
  regrow(prach.d_off, prach.off_cap, nocc);
  regrow(prach.d_Y, prach.y_cap, (size_t)nocc * NZC);
  regrow(prach.d_corr, prach.corr_cap, (size_t)nocc * prach.nroots * NZC);
  regrow(prach.d_out, prach.out_cap, (size_t)nocc * prach.nroots * 130);
  HIP_CHECK(hipMemcpyAsync(prach.d_off, off.data(), nocc * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  lsn_launch_prach(d_iq, prach.d_off, nocc, prach.d_W, prach.d_D, prach.d_V, prach.N12, prach.Ncp, prach.b0, (int)prach.nroots, (int)prach.ncs,
                   (int)prach.nwin, prach.d_Y, prach.d_corr, prach.d_out, st);
  std::vector<float> res((size_t)nocc * prach.nroots * 130);
  HIP_CHECK(hipMemcpyAsync(res.data(), prach.d_out, res.size() * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  // threshold + result list in (occasion, root, window) order (srsran_prach_detect_offset's index order)
  for (uint32_t o = 0; o < nocc; o++)
    for (uint32_t i = 0; i < prach.nroots; i++) {
      const float* r = res.data() + ((size_t)o * prach.nroots + i) * 130;
      const float ave = r[0], thr = prach.factor * ave;
      for (uint32_t j = 0; j < prach.nwin && j < 64; j++) {
        const float peak = r[2 + 2 * j];
        const uint32_t preamble = i * prach.nwin + j;
        if (!(peak > thr) || preamble >= 64) continue;
        lsn_prach_det_t d{};
        d.sf = occ_sf[o]; d.preamble = preamble; d.offset = (uint32_t)r[3 + 2 * j];
        d.offset_sec = (float)d.offset * (float)(24576.0 / 30.72e6) / (float)NZC;
        d.p2avg = peak / ave;
        out.push_back(d);
      }
    }
}

int Engine::prachDetect(const void* ul_iq, bool on_device, uint32_t nsf, uint32_t start_tti, lsn_prach_det_t* out, uint32_t cap)
{
  if (!cell_set || !prach.set) return LSN_ERROR;
  if ((!ul_iq && nsf) || (!out && cap)) return LSN_ERROR_INVALID_INPUTS;
  try {
    HIP_CHECK(hipSetDevice(cfg.device));
    const cf32* d_iq = (const cf32*)ul_iq;
    if (!on_device) {
      regrow(ul_d_iq, ul_iq_cap, (size_t)nsf * cd.sflen);
      HIP_CHECK(hipMemcpyAsync(ul_d_iq, ul_iq, (size_t)nsf * cd.sflen * sizeof(cf32), hipMemcpyHostToDevice, runner_u.stream));
      d_iq = ul_d_iq;
    }
    std::vector<lsn_prach_det_t> det;
    prachDetectDev(d_iq, 1, 0, nsf, start_tti, det);
    const uint32_t n = (uint32_t)std::min<size_t>(det.size(), cap);
    if (n) std::memcpy(out, det.data(), n * sizeof(lsn_prach_det_t));
    return (int)n;
  } catch (const std::exception& ex) {
    fprintf(stderr, "ltesniffer_amd: %s\n", ex.what());
    return LSN_ERROR;
  }
}

// parity tap: correlation power [nroots][839] of occasion `index` of the last detection call
long Engine::tapPrach(uint32_t index, void* out, size_t cap)
{
  if (!prach.set || index >= prach.last_nocc) return LSN_ERROR_INVALID_INPUTS;
  const size_t n = (size_t)prach.nroots * NZC * sizeof(float);
  if (n > cap) return LSN_ERROR_INVALID_INPUTS;
  if (hipMemcpy(out, prach.d_corr + (size_t)index * prach.nroots * NZC, n, hipMemcpyDeviceToHost) != hipSuccess) return LSN_ERROR;
  return (long)n;
}

}  // namespace lsn
