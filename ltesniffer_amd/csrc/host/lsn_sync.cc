// lsn_sync.cc - PSS / SSS cell search on a block of downlink samples: physical cell id, half-frame timing and carrier offset,
// i.e. what the reference has when rf_search_and_decode_mib(&rf, nant, &cell_detect_config, force_N_id_2, &cell, &cfo) returns
// (/root/reference/src/src/LTESniffer_Core.cc:195-204; srsRAN ue_cell_search / sync / pss / sss are not in the tree) and what a
// recording needs before lsn_phy_process_file can be pointed at it (-O offset, -c cell id of the reference's file mode).
// TS 36.211 6.11, FDD, normal cyclic prefix.  Sequences and cos/sin tables are host work (double), the matched filter, the two
// 62-carrier DFTs and the 336 SSS hypotheses are the kernels of stage_sync.hip; argmax / mean / atan2 on the host.
// Needs no Phy (the cell is not known yet).  Product code: no CPU fallback, nothing from oracle/ is included or linked.
#include "../../../include/ltesniffer_amd.h"
#include "../kernels/lsn_dev.h"
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#define HIP_CHECK(x)                                                                                       \
  do {                                                                                                     \
    hipError_t _e = (x);                                                                                   \
    if (_e != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #x); \
  } while (0)

void lsn_launch_pss_corr(const cf32* x, const cf32* p, uint32_t N, uint32_t W5, uint32_t P, uint32_t nroots, float* C, hipStream_t s);
void lsn_launch_sync_fin(const cf32* x, const cf32* p, const cf32* w, const cf32* d, const int8_t* sss, uint32_t N, uint32_t W5, uint32_t P, uint32_t bn,
                         uint32_t cp, void* out, hipStream_t s);

namespace lsn {

struct SyncFin {  // LsnSyncFin of stage_sync.hip
  uint32_t j;
  float y[2][2];
  float hyp[336][2];
};

static uint32_t sync_fft_size(uint32_t nof_prb)
{
  switch (nof_prb) {
    case 6: return 128;
    case 15: return 256;
    case 25: return 512;
    case 50: return 1024;
    case 75: return 1536;
    case 100: return 2048;
    default: return 0;
  }
}

// 36.211 6.11.1.1: d_u(n), the length-63 Zadoff-Chu sequence of root 25 / 29 / 34 without its middle element
static void pss_sequence(uint32_t n_id_2, cf32* d)
{
  static const int kRoot[3] = {25, 29, 34};
  for (int n = 0; n < 62; n++) {
    const int a = n < 31 ? n * (n + 1) : (n + 1) * (n + 2);
    const double ph = -M_PI * (double)kRoot[n_id_2] * (double)(a % 126) / 63.0;
    d[n] = {(float)std::cos(ph), (float)std::sin(ph)};
  }
}
static inline int sync_bin(int m, int N) { return m < 31 ? N - 31 + m : m - 30; }  // carriers -31..-1, +1..+31

// unit-energy time-domain replica
static void pss_replica(uint32_t n_id_2, uint32_t N, cf32* p)
{
  cf32 d[62];
  pss_sequence(n_id_2, d);
  const double sc = 1.0 / std::sqrt(62.0 * (double)N);
  for (uint32_t n = 0; n < N; n++) {
    double ar = 0, ai = 0;
    for (int m = 0; m < 62; m++) {
      const uint32_t idx = (uint32_t)(((uint64_t)sync_bin(m, (int)N) * n) % N);
      const double ph = 2.0 * M_PI * (double)idx / (double)N;
      const double c = std::cos(ph), s = std::sin(ph);
      ar += (double)d[m].r * c - (double)d[m].i * s;
      ai += (double)d[m].r * s + (double)d[m].i * c;
    }
    p[n] = {(float)(ar * sc), (float)(ai * sc)};
  }
}

// 36.211 6.11.2.1: all 168 x 2 SSS sequences of one N_id_2, row h = 2 N_id_1 + (subframe 5)
static void sss_table(uint32_t n_id_2, int8_t* out /* [336][62] */)
{
  int s[31], c[31], z[31];
  for (int i = 0; i < 5; i++) s[i] = c[i] = z[i] = (i == 4);
  for (int i = 0; i < 26; i++) {
    s[i + 5] = (s[i + 2] + s[i]) & 1;
    c[i + 5] = (c[i + 3] + c[i]) & 1;
    z[i + 5] = (z[i + 4] + z[i + 2] + z[i + 1] + z[i]) & 1;
  }
  for (int i = 0; i < 31; i++) { s[i] = 1 - 2 * s[i]; c[i] = 1 - 2 * c[i]; z[i] = 1 - 2 * z[i]; }
  for (uint32_t n1 = 0; n1 < 168; n1++) {
    const uint32_t qp = n1 / 30, q = (n1 + qp * (qp + 1) / 2) / 30, mp = n1 + q * (q + 1) / 2;
    const uint32_t m0 = mp % 31, m1 = (m0 + mp / 31 + 1) % 31;
    int8_t* d0 = out + (size_t)(2 * n1) * 62;
    int8_t* d5 = d0 + 62;
    for (uint32_t n = 0; n < 31; n++) {
      const int s0 = s[(n + m0) % 31], s1 = s[(n + m1) % 31], c0 = c[(n + n_id_2) % 31], c1 = c[(n + n_id_2 + 3) % 31];
      const int z0 = z[(n + m0 % 8) % 31], z1 = z[(n + m1 % 8) % 31];
      d0[2 * n] = (int8_t)(s0 * c0);
      d0[2 * n + 1] = (int8_t)(s1 * c1 * z0);
      d5[2 * n] = (int8_t)(s1 * c0);
      d5[2 * n + 1] = (int8_t)(s0 * c1 * z1);
    }
  }
}

namespace {
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  template <typename T> T* alloc(size_t n) { HIP_CHECK(hipMalloc(&p, n * sizeof(T))); return (T*)p; }
};
}  // namespace

int cell_search(int device, const cf32* iq, bool on_device, uint64_t nsamples, uint32_t nof_prb, const lsn_cell_search_cfg_t& cfg, lsn_cell_search_t& out,
                float* corr_out)
{
  std::memset(&out, 0, sizeof out);
  const uint32_t N = sync_fft_size(nof_prb);
  const uint32_t P = cfg.nof_periods ? cfg.nof_periods : 1;
  if (!iq || !N || P > 16 || cfg.force_n_id_2 > 2 || cfg.force_n_id_2 < -1) return LSN_ERROR_INVALID_INPUTS;
  const uint32_t W5 = 75 * N;
  const uint64_t need = (uint64_t)(P + 1) * W5 + N;
  if (nsamples < need) return LSN_ERROR_INVALID_INPUTS;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return LSN_ERROR_NO_DEVICE;
  HIP_CHECK(hipSetDevice(device));
  hipStream_t st = nullptr;
  HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{st};

  const uint32_t nroots = cfg.force_n_id_2 >= 0 ? 1u : 3u;
  std::vector<cf32> rep((size_t)nroots * N);
  for (uint32_t r = 0; r < nroots; r++) pss_replica(cfg.force_n_id_2 >= 0 ? (uint32_t)cfg.force_n_id_2 : r, N, rep.data() + (size_t)r * N);

  DevBuf bx, bp, bc, bw, bd, bs, bo;
  const cf32* d_x = iq;
  if (!on_device) {
    cf32* dx = bx.alloc<cf32>(need);
    HIP_CHECK(hipMemcpyAsync(dx, iq, need * sizeof(cf32), hipMemcpyHostToDevice, st));
    d_x = dx;
  }
  cf32* d_p = bp.alloc<cf32>(rep.size());
  float* d_c = bc.alloc<float>((size_t)nroots * W5);
  HIP_CHECK(hipMemcpyAsync(d_p, rep.data(), rep.size() * sizeof(cf32), hipMemcpyHostToDevice, st));
  lsn_launch_pss_corr(d_x, d_p, N, W5, P, nroots, d_c, st);
  std::vector<float> C((size_t)nroots * W5);
  HIP_CHECK(hipMemcpyAsync(C.data(), d_c, C.size() * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));

  // first maximum in (root, lag) order; acceptance: peak over the mean of the winning root's correlation power
  float best = -1.0f;
  uint32_t br = 0, bn = 0;
  for (uint32_t r = 0; r < nroots; r++)
    for (uint32_t n = 0; n < W5; n++) {
      const float c = C[(size_t)r * W5 + n];
      if (c > best) { best = c; br = r; bn = n; }
    }
  double mean = 0.0;
  for (uint32_t n = 0; n < W5; n++) mean += (double)C[(size_t)br * W5 + n];
  mean /= (double)W5;
  const uint32_t n_id_2 = cfg.force_n_id_2 >= 0 ? (uint32_t)cfg.force_n_id_2 : br;
  if (corr_out) {
    std::memset(corr_out, 0, (size_t)3 * W5 * sizeof(float));
    for (uint32_t r = 0; r < nroots; r++)
      std::memcpy(corr_out + (size_t)(cfg.force_n_id_2 >= 0 ? n_id_2 : r) * W5, C.data() + (size_t)r * W5, (size_t)W5 * sizeof(float));
  }
  out.n_id_2 = n_id_2;
  out.pss_pos = bn;
  out.pss_peak = best;
  out.pss_p2avg = mean > 0.0 ? (float)((double)best / mean) : 0.0f;
  out.found = out.pss_p2avg >= cfg.threshold ? 1u : 0u;

  // SSS, carrier offset
  std::vector<cf32> w(N);
  for (uint32_t i = 0; i < N; i++) {
    const double ph = -2.0 * M_PI * (double)i / (double)N;
    w[i] = {(float)std::cos(ph), (float)std::sin(ph)};
  }
  cf32 d[62];
  pss_sequence(n_id_2, d);
  std::vector<int8_t> sss((size_t)336 * 62);
  sss_table(n_id_2, sss.data());
  cf32* d_w = bw.alloc<cf32>(N);
  cf32* d_d = bd.alloc<cf32>(62);
  int8_t* d_s = bs.alloc<int8_t>(sss.size());
  SyncFin* d_o = bo.alloc<SyncFin>(2);
  HIP_CHECK(hipMemcpyAsync(d_w, w.data(), N * sizeof(cf32), hipMemcpyHostToDevice, st));
  HIP_CHECK(hipMemcpyAsync(d_d, d, sizeof d, hipMemcpyHostToDevice, st));
  HIP_CHECK(hipMemcpyAsync(d_s, sss.data(), sss.size(), hipMemcpyHostToDevice, st));
  // The cyclic prefix of the cell (the reference's search reports it, LTESniffer_Core.cc:195-204): the SSS symbol sits N + 144 (x N / 2048) samples in front of
  // the PSS symbol with the normal CP, N + 512 with the extended one - both positions are evaluated, the larger best SSS metric decides (normal on a tie)
  const uint32_t cps[2] = {144 * N / 2048, 512 * N / 2048};
  for (int hyp = 0; hyp < 2; hyp++) lsn_launch_sync_fin(d_x, d_p + (size_t)br * N, d_w, d_d, d_s, N, W5, P, bn, cps[hyp], d_o + hyp, st);
  SyncFin fin2[2];
  HIP_CHECK(hipMemcpyAsync(fin2, d_o, sizeof fin2, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  float win_m1 = -1.0f;
  for (uint32_t hyp = 0; hyp < 2; hyp++) {
    const SyncFin& fin = fin2[hyp];
    const uint32_t cp = cps[hyp];
    float m1 = -1.0f, m2 = -1.0f, hr = 0.0f, hi = 0.0f;
    uint32_t bh = 0;
    for (uint32_t h = 0; h < 336; h++) {
      const float ar = fin.hyp[h][0], ai = fin.hyp[h][1];
      const float mt = ar * ar + ai * ai;
      if (mt > m1) { m2 = m1; m1 = mt; bh = h; hr = ar; hi = ai; }
      else if (mt > m2) m2 = mt;
    }
    if (!(m1 > win_m1)) continue;
    win_m1 = m1;
    out.cp = hyp;
    {
      const float cr = fin.y[0][0] * fin.y[1][0] + fin.y[0][1] * fin.y[1][1], ci = fin.y[0][0] * fin.y[1][1] - fin.y[0][1] * fin.y[1][0];  // conj(y0) y1
      out.cfo_coarse_hz = atan2f(ci, cr) / (float)M_PI * 15000.0f;
    }
    out.n_id_1 = bh >> 1;
    out.cell_id = 3 * out.n_id_1 + n_id_2;
    out.sss_metric = m1;
    out.sss_second = m2;
    out.cfo_hz = -atan2f(hi, hr) / (2.0f * (float)M_PI) * (15000.0f * (float)N / (float)(N + cp));
    // subframe timing: the useful part of the PSS symbol starts 160 + 6 (N + 144) [x N / 2048] samples into subframes 0 and 5 (normal CP), 5 (N + 512) + 512 (extended)
    const uint32_t pss_off = hyp ? 5 * (N + cp) + cp : 160 * N / 2048 + 6 * (N + cp);
    const uint32_t sf_used = (bh & 1u) ? 5u : 0u;                      // of the occurrence the SSS was taken from
    const uint32_t sf_bn = (fin.j & 1u) ? (sf_used + 5) % 10 : sf_used;  // of the first occurrence
    if (bn >= pss_off) {
      out.sf_start = bn - pss_off;
      out.sf_idx = sf_bn;
    } else {
      out.sf_start = bn + W5 - pss_off;
      out.sf_idx = (sf_bn + 5) % 10;
    }
  }
  return out.found ? 1 : 0;
}

}  // namespace lsn
