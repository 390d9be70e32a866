// stage_sync.hip - PSS / SSS cell search (gfx950).  Replaces what the reference gets from rf_search_and_decode_mib
// (/root/reference/src/src/LTESniffer_Core.cc:195-204) [srsRAN sync/pss/sss, not in tree]; TS 36.211 6.11, FDD, normal CP.
//
// k_pss_corr   one thread per lag n of the 5 ms search window: full-rate matched filter of up to three PSS replicas,
//              each power divided by the energy of its window, the ratios of P consecutive periods added.  256 lags per workgroup; the 256 + N samples a workgroup needs live in
//              LDS (lane t reads xs[t + k]: consecutive 8-byte words, conflict free), the replica taps are wave-uniform
//              (scalar loads).  Every sum runs in index order with one rounding per operation, as the CPU statement of the same
//              algorithm does, so results are bit-identical (compiled with -ffp-contract=off).
// k_sync_fin   one workgroup: (a) the matched filter in two halves at the winning lag for each of the P + 1 occurrences,
//              (b) pick the strongest occurrence whose SSS symbol is inside the buffer, (c) 62-carrier DFT of its PSS and SSS
//              symbols (twiddle table), channel from the PSS, (d) the 336 SSS hypotheses (168 N_id_1 x subframe 0 / 5).
// The search runs once per capture; it is latency, not throughput, that matters: 0.5 M lags x 2048 taps x 3 roots at 20 MHz.
#include "lsn_dev.h"

#define SYNC_TILE 256

__global__ __launch_bounds__(SYNC_TILE) void k_pss_corr(const cf32* __restrict__ x, const cf32* __restrict__ p /* [nroots][N] */, uint32_t N, uint32_t W5,
                                                        uint32_t P, uint32_t nroots, float* __restrict__ C /* [nroots][W5] */)
{
  __shared__ cf32 xs[SYNC_TILE + 2048];
  const cf32* __restrict__ ps = p;  // uniform index: scalar loads
  const uint32_t t = threadIdx.x, n0 = blockIdx.x * SYNC_TILE, n = n0 + t;
  float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
  for (uint32_t q = 0; q < P; q++) {
    __syncthreads();
    const cf32* xq = x + (size_t)q * W5 + n0;
    for (uint32_t i = t; i < SYNC_TILE + N; i += SYNC_TILE) xs[i] = xq[i];  // the buffer holds (P + 1) W5 + N samples: in range
    __syncthreads();
    float ar0 = 0.0f, ai0 = 0.0f, ar1 = 0.0f, ai1 = 0.0f, ar2 = 0.0f, ai2 = 0.0f, e = 0.0f;
    if (nroots == 3) {
      for (uint32_t k = 0; k < N; k++) {
        const cf32 v = xs[t + k];
        const cf32 a = ps[k], b = ps[N + k], d = ps[2 * N + k];
        e = e + (v.r * v.r + v.i * v.i);
        ar0 = ar0 + (v.r * a.r + v.i * a.i);
        ai0 = ai0 + (v.i * a.r - v.r * a.i);
        ar1 = ar1 + (v.r * b.r + v.i * b.i);
        ai1 = ai1 + (v.i * b.r - v.r * b.i);
        ar2 = ar2 + (v.r * d.r + v.i * d.i);
        ai2 = ai2 + (v.i * d.r - v.r * d.i);
      }
    } else {
      for (uint32_t k = 0; k < N; k++) {
        const cf32 v = xs[t + k];
        const cf32 a = ps[k];
        e = e + (v.r * v.r + v.i * v.i);
        ar0 = ar0 + (v.r * a.r + v.i * a.i);
        ai0 = ai0 + (v.i * a.r - v.r * a.i);
      }
    }
    if (e > 0.0f) {  // power over the energy of the window: in [0, 1], the replicas have unit energy
      c0 = c0 + (ar0 * ar0 + ai0 * ai0) / e;
      c1 = c1 + (ar1 * ar1 + ai1 * ai1) / e;
      c2 = c2 + (ar2 * ar2 + ai2 * ai2) / e;
    }
  }
  if (n < W5) {
    C[n] = c0;
    if (nroots == 3) {
      C[W5 + n] = c1;
      C[2 * (size_t)W5 + n] = c2;
    }
  }
}

struct LsnSyncFin {
  uint32_t j;          // PSS occurrence used (sample bn + j W5)
  float y[2][2];       // its matched filter halves (re, im)
  float hyp[336][2];   // coherent SSS sums per hypothesis h = 2 N_id_1 + (subframe 5)
};

__global__ __launch_bounds__(512) void k_sync_fin(const cf32* __restrict__ x, const cf32* __restrict__ p /* [N] replica of the winning root */,
                                                  const cf32* __restrict__ w /* [N] exp(-2 pi i k / N) */, const cf32* __restrict__ d /* [62] PSS sequence */,
                                                  const int8_t* __restrict__ sss /* [336][62] */, uint32_t N, uint32_t W5, uint32_t P, uint32_t bn,
                                                  uint32_t cp, LsnSyncFin* __restrict__ out)
{
  __shared__ float yh[17][2][3];  // [occurrence][half][re / im / energy]
  __shared__ cf32 Y[2][62];
  __shared__ cf32 z[62];
  __shared__ uint32_t jsel;
  const uint32_t t = threadIdx.x;
  if (t < 2 * (P + 1)) {
    const uint32_t j = t >> 1, h = t & 1;
    const cf32* xj = x + (size_t)j * W5 + bn;
    float ar = 0.0f, ai = 0.0f, e = 0.0f;
    for (uint32_t k = h * (N / 2); k < (h + 1) * (N / 2); k++) {
      const cf32 v = xj[k], a = p[k];
      ar = ar + (v.r * a.r + v.i * a.i);
      ai = ai + (v.i * a.r - v.r * a.i);
      e = e + (v.r * v.r + v.i * v.i);
    }
    yh[j][h][0] = ar;
    yh[j][h][1] = ai;
    yh[j][h][2] = e;
  }
  __syncthreads();
  if (t == 0) {
    uint32_t jb = 0;
    float cjb = -1.0f;
    for (uint32_t j = bn >= N + cp ? 0u : 1u; j <= P; j++) {
      const float sr = yh[j][0][0] + yh[j][1][0], si = yh[j][0][1] + yh[j][1][1], et = yh[j][0][2] + yh[j][1][2];
      const float cj = et > 0.0f ? (sr * sr + si * si) / et : 0.0f;
      if (cj > cjb) { cjb = cj; jb = j; }
    }
    jsel = jb;
    out->j = jb;
    out->y[0][0] = yh[jb][0][0]; out->y[0][1] = yh[jb][0][1];
    out->y[1][0] = yh[jb][1][0]; out->y[1][1] = yh[jb][1][1];
  }
  __syncthreads();
  const uint32_t q0 = bn + jsel * W5;
  if (t < 124) {
    const uint32_t s = t / 62, m = t % 62;
    const uint32_t kb = m < 31 ? N - 31 + m : m - 30;
    const cf32* xx = s ? x + q0 - (N + cp) : x + q0;
    float ar = 0.0f, ai = 0.0f;
    uint32_t idx = 0;  // (kb * n) mod N
    for (uint32_t n = 0; n < N; n++) {
      const cf32 v = xx[n], ww = w[idx];
      ar = ar + (v.r * ww.r - v.i * ww.i);
      ai = ai + (v.r * ww.i + v.i * ww.r);
      idx += kb;
      idx = idx >= N ? idx - N : idx;
    }
    Y[s][m].r = ar;
    Y[s][m].i = ai;
  }
  __syncthreads();
  if (t < 62) {
    const cf32 yp = Y[0][t], ys = Y[1][t], dd = d[t];
    const float hr = yp.r * dd.r + yp.i * dd.i, hi = yp.i * dd.r - yp.r * dd.i;  // H = Ypss conj(d)
    z[t].r = ys.r * hr + ys.i * hi;                                              // z = Ysss conj(H)
    z[t].i = ys.i * hr - ys.r * hi;
  }
  __syncthreads();
  if (t < 336) {
    const int8_t* sq = sss + (size_t)t * 62;
    float ar = 0.0f, ai = 0.0f;
    for (int m = 0; m < 62; m++) {
      const float sg = (float)sq[m];
      ar = ar + z[m].r * sg;
      ai = ai + z[m].i * sg;
    }
    out->hyp[t][0] = ar;
    out->hyp[t][1] = ai;
  }
}

void lsn_launch_pss_corr(const cf32* x, const cf32* p, uint32_t N, uint32_t W5, uint32_t P, uint32_t nroots, float* C, hipStream_t s)
{
  LSN_LAUNCH(k_pss_corr, dim3((W5 + SYNC_TILE - 1) / SYNC_TILE), dim3(SYNC_TILE), 0, s, x, p, N, W5, P, nroots, C);
}
void lsn_launch_sync_fin(const cf32* x, const cf32* p, const cf32* w, const cf32* d, const int8_t* sss, uint32_t N, uint32_t W5, uint32_t P, uint32_t bn,
                         uint32_t cp, void* out, hipStream_t s)
{
  LSN_LAUNCH(k_sync_fin, dim3(1), dim3(512), 0, s, x, p, w, d, sss, N, W5, P, bn, cp, (LsnSyncFin*)out);
}
