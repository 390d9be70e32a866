// stage_ul.hip - gfx950 kernels of the uplink receive chain: SC-FDMA demodulation of the uplink antenna and, per PUSCH
// grant, DMRS channel estimation, 1-tap MMSE equalisation, transform de-precoding, soft demodulation, descrambling and
// channel de-interleaving.  They replace srsran_enb_ul_fft (/root/reference/src/src/UL_Sniffer_PUSCH.cc:392),
// srsran_chest_ul_estimate_pusch (:256) and the front half of srsran_pusch_decode (:262); the back half (rate
// de-matching + turbo + CRC) is k_turbo of stage_c.hip.  Scope: one antenna, no hopping, no SRS, no UCI, L_prb >= 3.
// Float arithmetic: one rounding per operation, fixed summation orders (bit-identical to the tests' CPU oracle).
#include "lsn_dev.h"

#define LLR_Q 180.0f

__device__ __forceinline__ cf32 cmul(cf32 a, cf32 b) { cf32 c; c.r = a.r * b.r - a.i * b.i; c.i = a.r * b.i + a.i * b.r; return c; }
__device__ __forceinline__ cf32 cmulconj(cf32 a, cf32 b) { cf32 c; c.r = a.r * b.r + a.i * b.i; c.i = a.i * b.r - a.r * b.i; return c; }

// ------------------------------------------------------------------------------------------------ SC-FDMA demodulation
template <int R>
__device__ __forceinline__ void ul_fft_pass(cf32* a, const cf32* w, int s, int N, int lgN, int tid)
{
  constexpr int G = 1 << R;
  const int h = 1 << s;
#pragma unroll
  for (int u = 0; u < (8 >> R); u++) {
    int g = tid * (8 >> R) + u;
    if (g >= (N >> R)) break;
    int low = g & (h - 1), high = g >> s, base = (high << (s + R)) | low;
    cf32 e[G];
#pragma unroll
    for (int j = 0; j < G; j++) e[j] = a[base + j * h];
#pragma unroll
    for (int q = 0; q < R; q++) {
#pragma unroll
      for (int j = 0; j < G; j++) {
        if (j & (1 << q)) continue;
        int pos = low + (j & ((1 << q) - 1)) * h;
        cf32 v = cmul(e[j + (1 << q)], w[pos << (lgN - (s + q + 1))]);
        cf32 uu = e[j];
        e[j].r = uu.r + v.r; e[j].i = uu.i + v.i;
        e[j + (1 << q)].r = uu.r - v.r; e[j + (1 << q)].i = uu.i - v.i;
      }
    }
#pragma unroll
    for (int j = 0; j < G; j++) a[base + j * h] = e[j];
  }
}

// one workgroup per (subframe, symbol): CP strip, 7.5 kHz shift, radix-8/4/2 DIT FFT in LDS, carrier extract (no DC gap)
__global__ __launch_bounds__(256) void k_ul_fft(LsnCellDev c, const cf32* __restrict__ iq, uint32_t nant, uint32_t ant, cf32* __restrict__ grid)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int N = (int)c.N, lgN = (int)c.lgN, tid = threadIdx.x;
  cf32* a = (cf32*)smem;
  cf32* w = a + N;
  const int nsym = (int)c.nsym;  // 14, or 12 with the extended cyclic prefix (rows 12, 13 of the grid stay zero)
  const int l = blockIdx.x % nsym, sf = blockIdx.x / nsym;
  const int cp0 = 160 * N / 2048, cp1 = 144 * N / 2048;
  const int slot = l / 7, ls = l % 7;
  const int pos = c.cp ? l * (N + N / 4) + N / 4 : slot * (cp0 + 6 * cp1 + 7 * N) + cp0 + ls * (N + cp1);
  const cf32* in = iq + ((size_t)sf * nant + ant) * c.sflen + pos;
  const int nre = (int)c.nre;
  cf32* out = grid + ((size_t)sf * 14 + l) * nre;
  if (c.twiddle3) {  // 15 MHz, N = 1536 = 3 x 512 (see k_ofdm)
    const int M = (int)c.nsub;
    for (int n = tid; n < M / 2; n += 256) w[n] = c.twiddle[n];
    for (int n = tid; n < N; n += 256) {
      const int m = n / 3, r = n - 3 * m;
      a[r * M + (int)(__brev((unsigned)m) >> (32 - lgN))] = cmul(in[n], c.ul_shift[n]);
    }
    __syncthreads();
    for (int s = 0; s < lgN; s += 3) {
      for (int r = 0; r < 3; r++) ul_fft_pass<3>(a + r * M, w, s, M, lgN, tid);
      __syncthreads();
    }
    const cf32* __restrict__ T = c.twiddle3;
    for (int k = tid; k < nre; k += 256) {
      const int bin = (k < nre / 2) ? (N - nre / 2 + k) : (k - nre / 2), kq = bin & (M - 1);
      int b2 = 2 * bin;
      b2 = b2 >= N ? b2 - N : b2;
      const cf32 t1 = cmul(a[M + kq], T[bin]), t2 = cmul(a[2 * M + kq], T[b2]);
      const float sr = a[kq].r + t1.r, si = a[kq].i + t1.i;
      cf32 X;
      X.r = sr + t2.r;
      X.i = si + t2.i;
      out[k] = X;
    }
    return;
  }
  for (int n = tid; n < N / 2; n += 256) w[n] = c.twiddle[n];
  for (int n = tid; n < N; n += 256) a[__brev((unsigned)n) >> (32 - lgN)] = cmul(in[n], c.ul_shift[n]);
  __syncthreads();
  int s = 0;
  while (s < lgN) {
    int r = lgN - s;
    if (r >= 3) { ul_fft_pass<3>(a, w, s, N, lgN, tid); s += 3; }
    else if (r == 2) { ul_fft_pass<2>(a, w, s, N, lgN, tid); s += 2; }
    else { ul_fft_pass<1>(a, w, s, N, lgN, tid); s += 1; }
    __syncthreads();
  }
  for (int k = tid; k < nre; k += 256) out[k] = a[(k < nre / 2) ? (N - nre / 2 + k) : (k - nre / 2)];
}

void lsn_launch_ul_fft(const LsnCellDev& c, const cf32* iq, uint32_t nant, uint32_t ant, cf32* grid, uint32_t nsf, hipStream_t s)
{
  LSN_LAUNCH(k_ul_fft, dim3(nsf * c.nsym), dim3(256), sizeof(cf32) * (c.N + c.N / 2), s, c, iq, nant, ant, grid);
}

// ------------------------------------------------------------------------------------------------ DMRS estimate
// one workgroup per grant: LS on the reference-signal symbols (3 / 10; extended CP 2 / 8), 3-tap smoothing per slot, noise = mean |smoothed - ls|^2,
// signal = mean |smoothed|^2 (both over the 2 M values in the fixed 256-way order)
__global__ __launch_bounds__(256) void k_pusch_chest(LsnCellDev c, const LsnUlGrantDev* __restrict__ grants, const cf32* __restrict__ grid,
                                                     cf32* __restrict__ hs_out, float* __restrict__ stat)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const LsnUlGrantDev g = grants[blockIdx.x];
  const int M = 12 * (int)g.L_prb, nre = (int)c.nre, tid = threadIdx.x;
  cf32* ls = (cf32*)smem;            // [2 M]
  float* part = (float*)(ls + 2 * M);  // [2][256]
  for (int i = tid; i < 2 * M; i += 256) {
    const int s = i >= M ? 1 : 0, n = i - s * M;
    const cf32 y = grid[((size_t)g.sf * 14 + (c.nslot - 4) + c.nslot * s) * nre + 12 * (s ? g.n_prb2 : g.n_prb) + n];   // reference signal: symbol 3 of each slot (extended CP: 2)
    const cf32 r = cmul(c.ul_base[(s ? g.base_off1 : g.base_off) + n], c.ul_ph12[(g.ncs[s] * (uint32_t)n) % 12u]);
    ls[i] = cmulconj(y, r);
  }
  __syncthreads();
  cf32* hs = hs_out + g.hs_off;
  float p0 = 0.0f, p1 = 0.0f;
  for (int i = tid; i < 2 * M; i += 256) {
    const int s = i >= M ? 1 : 0, n = i - s * M;
    const cf32* q = ls + s * M;
    cf32 a;
    if (n == 0) { a.r = (q[0].r + q[1].r) / 2.0f; a.i = (q[0].i + q[1].i) / 2.0f; }
    else if (n == M - 1) { a.r = (q[n - 1].r + q[n].r) / 2.0f; a.i = (q[n - 1].i + q[n].i) / 2.0f; }
    else { a.r = ((q[n - 1].r + q[n].r) + q[n + 1].r) / 3.0f; a.i = ((q[n - 1].i + q[n].i) + q[n + 1].i) / 3.0f; }
    hs[i] = a;
    const float dr = a.r - q[n].r, di = a.i - q[n].i;
    p0 = p0 + (dr * dr + di * di);
    p1 = p1 + (a.r * a.r + a.i * a.i);
  }
  part[tid] = p0; part[256 + tid] = p1;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { part[tid] = part[tid] + part[tid + s]; part[256 + tid] = part[256 + tid] + part[256 + tid + s]; }
    __syncthreads();
  }
  if (tid == 0) {
    stat[2 * blockIdx.x] = part[0] / (float)(2 * M);
    stat[2 * blockIdx.x + 1] = part[256] / (float)(2 * M);
  }
}

void lsn_launch_pusch_chest(const LsnCellDev& c, const LsnUlGrantDev* g, const cf32* grid, cf32* hs, float* stat, uint32_t ngrants, hipStream_t s)
{
  LSN_LAUNCH(k_pusch_chest, dim3(ngrants), dim3(256), sizeof(cf32) * 2 * 1200 + sizeof(float) * 512, s, c, g, grid, hs, stat);
}

// ------------------------------------------------------------------------------------------------ equalise + IDFT + demod
__device__ __forceinline__ void ul_demod_llr(int Qm, float I, float Q, float* L)
{
  float aI = fabsf(I), aQ = fabsf(Q);
  L[0] = -I; L[1] = -Q;
  if (Qm == 4) {
    const float a = 0.31622776601683794f;
    L[2] = aI - 2.0f * a; L[3] = aQ - 2.0f * a;
  } else if (Qm == 6) {
    const float a = 0.15430334996209191f;
    float tI = aI - 4.0f * a, tQ = aQ - 4.0f * a;
    L[2] = tI; L[3] = tQ; L[4] = fabsf(tI) - 2.0f * a; L[5] = fabsf(tQ) - 2.0f * a;
  } else if (Qm == 8) {
    const float a = 0.07669649888473704f;
    float tI = aI - 8.0f * a, tQ = aQ - 8.0f * a;
    float uI = fabsf(tI) - 4.0f * a, uQ = fabsf(tQ) - 4.0f * a;
    L[2] = tI; L[3] = tQ; L[4] = uI; L[5] = uQ; L[6] = fabsf(uI) - 2.0f * a; L[7] = fabsf(uQ) - 2.0f * a;
  }
}

// one workgroup per (data symbol 0..11 - 0..9 with the extended CP -, grant): equalised carriers, a ping-pong buffer and the IDFT twiddles in LDS; transform de-precoding as
// an autosort (Stockham) decimation-in-frequency IDFT over the radices 4 (while the remaining length divides by 4), 2, 3, 5 - one thread per
// output of a stage, terms added in index order: the operation order of the oracle's o_idft_mixed (M = 1200: 21 complex MACs per output
// instead of 1200); LLRs written in UL-SCH order
__global__ __launch_bounds__(256) void k_pusch_demod(LsnCellDev c, const LsnUlGrantDev* __restrict__ grants, const cf32* __restrict__ grid,
                                                     const cf32* __restrict__ hs_all, const float* __restrict__ stat, int16_t* __restrict__ llr)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const LsnUlGrantDev g = grants[blockIdx.y];
  // data symbol `col` of the subframe (12, or 10 with the extended CP) -> symbol l: the reference-signal symbol of each slot is skipped
  const int nsl = (int)c.nslot, dm = nsl - 4, ncol = 2 * (nsl - 1);
  const int col = blockIdx.x, l = col < dm ? col : (col < nsl - 1 + dm ? col + 1 : col + 2);
  const int M = 12 * (int)g.L_prb, nre = (int)c.nre, Qm = (int)g.qm, tid = threadIdx.x;
  cf32* xa = (cf32*)smem;  // [M]
  cf32* xb = xa + M;       // [M]
  cf32* w = xb + M;        // [M] exp(+2 pi j k / M)
  const cf32* y = grid + ((size_t)g.sf * 14 + l) * nre + 12 * (l >= nsl ? g.n_prb2 : g.n_prb);
  const cf32* h = hs_all + g.hs_off + (l >= nsl ? 1 : 0) * M;
  const cf32* wt = c.ul_idft + g.idft_off;
  const float noise = stat[2 * blockIdx.y];
  for (int n = tid; n < M; n += 256) {
    const cf32 hh = h[n], t = cmulconj(y[n], hh);
    const float den = (hh.r * hh.r + hh.i * hh.i) + noise;
    cf32 v; v.r = t.r / den; v.i = t.i / den;
    xa[n] = v;
    w[n] = wt[n];
  }
  __syncthreads();
  cf32 *in = xa, *out = xb;
  for (int n = M, s = 1; n > 1;) {
    const int r = (n % 4 == 0) ? 4 : (n % 2 == 0) ? 2 : (n % 3 == 0) ? 3 : 5;
    const int m = n / r, wr = M / r;
    for (int o = tid; o < M; o += 256) {
      const int q = o % s, rest = o / s, t = rest % r, p = rest / r;
      cf32 acc = cmul(in[q + s * p], w[0]);
      int wi = 0;
      for (int i = 1; i < r; i++) {
        wi += t * wr; wi = wi >= M ? wi - M : wi;            // (i t M / r) mod M
        const cf32 term = cmul(in[q + s * (p + m * i)], w[wi]);
        acc.r = acc.r + term.r; acc.i = acc.i + term.i;
      }
      out[o] = cmul(acc, w[(int)(((unsigned)p * (unsigned)t * (unsigned)s) % (unsigned)M)]);
    }
    __syncthreads();
    cf32* sw = in; in = out; out = sw;
    n = m; s *= r;
  }
  const float scale = g.scale;  // 1 / sqrt(M), from the host
  int16_t* e = llr + g.llr_off;
  for (int r = tid; r < M; r += 256) {
    const float ar = in[r].r, ai = in[r].i;
    // cell (r, col) of the M x 12 channel-interleaver matrix (36.212 5.2.2.8) in closed form: the i-th RI / HARQ-ACK symbol sits
    // in row M - 1 - i / 4, column set[(-i) mod 4]; CQI then data fill the other cells row by row
    int dcell; bool is_ack = false;
    {
      const int qri = (int)g.q_ri, qack = (int)g.q_ack, qcqi = (int)g.q_cqi;
      auto in_col = [&](int q, int slot) { const int i0 = (4 - slot) & 3; return q > i0 ? (q - 1 - i0) / 4 + 1 : 0; };  // symbols of a column slot
      int before = 0; bool is_ri = false;
#pragma unroll
      for (int slot = 0; slot < 4; slot++) {
        // column sets of 36.212 Tables 5.2.2.8-1 / -2: rank indication 1 4 7 10 (extended CP 0 3 5 8), HARQ-ACK 2 3 8 9 (1 2 6 7)
        const int cc = c.cp ? (slot == 0 ? 0 : (slot == 1 ? 3 : (slot == 2 ? 5 : 8))) : (slot == 0 ? 1 : (slot == 1 ? 4 : (slot == 2 ? 7 : 10))), n = in_col(qri, slot), top = M - n;
        before += r > top ? r - top : 0;             // RI cells of this column in the rows above r
        if (n && r >= top && cc < col) before += 1;  // ... and to the left in row r
        if (n && r >= top && cc == col) is_ri = true;
        const int ca = c.cp ? (slot == 0 ? 1 : (slot == 1 ? 2 : (slot == 2 ? 6 : 7))) : (slot == 0 ? 2 : (slot == 1 ? 3 : (slot == 2 ? 8 : 9)));
        if (ca == col && r >= M - in_col(qack, slot) && in_col(qack, slot)) is_ack = true;
      }
      const int rank = r * ncol + col - before;
      dcell = (is_ri || rank < qcqi) ? -1 : rank - qcqi;
    }
    float Lb[8];
    ul_demod_llr(Qm, ar * scale, ai * scale, Lb);
#pragma unroll
    for (int b = 0; b < 8; b++) {  // fixed trip count keeps Lb in registers (a run-time bound would put it in scratch memory)
      if (b >= Qm) break;
      float v = rintf(Lb[b] * LLR_Q);
      v = v > (float)LSN_LLR_CLIP ? (float)LSN_LLR_CLIP : v;
      v = v < (float)-LSN_LLR_CLIP ? (float)-LSN_LLR_CLIP : v;
      int q = (int)v;
      const uint32_t sidx = ((uint32_t)col * (uint32_t)M + (uint32_t)r) * (uint32_t)Qm + (uint32_t)b;  // scrambling: transmitted order
      const uint32_t cbit = (uint32_t)c.gold_x1[sidx] ^ (uint32_t)(__popc(c.gold_x2mask[sidx] & g.cinit) & 1);
      if (dcell >= 0) e[(uint32_t)dcell * (uint32_t)Qm + (uint32_t)b] = is_ack ? (int16_t)0 : (int16_t)(cbit ? -q : q);  // UL-SCH (row-major) order
    }
  }
}

void lsn_launch_pusch_demod(const LsnCellDev& c, const LsnUlGrantDev* g, const cf32* grid, const cf32* hs, const float* stat, int16_t* llr,
                            uint32_t ngrants, hipStream_t s)
{
  LSN_LAUNCH(k_pusch_demod, dim3(2 * (c.nslot - 1), ngrants), dim3(256), sizeof(cf32) * 3 * 1200, s, c, g, grid, hs, stat, llr);
}

// ------------------------------------------------------------------------------------------------ PRACH detection
// srsran_prach_detect_offset (/root/reference/src/src/UL_Sniffer_PUSCH.cc:690) for preamble format 0, three kernels:
//   k_prach_bins : the 839 PRACH bins of the 12N-point DFT of the samples behind the CP (one workgroup per bin: 256
//                  interleaved partial sums, fixed tree) - only 839 of the 12N outputs are needed, so no full FFT
//   k_prach_corr : bins x conj(root spectrum), 839-point inverse DFT (one workgroup per lag), |.|^2
//   k_prach_peaks: mean of the correlation power + the peak of every cyclic-shift window
// The threshold test and the result list are host work (a few dozen floats per occasion).
#define LSN_NZC 839

// fixed 256-leaf tree of the oracle's o_reduce256 on two arrays at once; result valid in thread 0
__device__ __forceinline__ void tree256_2(float* pr, float* pi, int tid)
{
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { pr[tid] = pr[tid] + pr[tid + s]; pi[tid] = pi[tid] + pi[tid + s]; }
    __syncthreads();
  }
}

// grid (839, nocc).  occ_off[o]: cf32 offset of the uplink subframe of occasion o inside iq; W[i] = exp(-2 pi j i / N12)
__global__ __launch_bounds__(256) void k_prach_bins(const cf32* __restrict__ iq, const uint64_t* __restrict__ occ_off, const cf32* __restrict__ W,
                                                    int N12, int Ncp, int b0, cf32* __restrict__ Y)
{
  __shared__ float pr[256], pi[256];
  const int j = blockIdx.x, o = blockIdx.y, tid = threadIdx.x;
  const cf32* x = iq + occ_off[o] + Ncp;
  const int b = ((b0 + j) % N12 + N12) % N12;
  int idx = (int)((long long)b * tid % N12);
  const int step = (int)((long long)b * 256 % N12);
  float ar = 0.0f, ai = 0.0f;
  for (int n = tid; n < N12; n += 256) {
    const cf32 p = cmul(x[n], W[idx]);
    ar = ar + p.r; ai = ai + p.i;
    idx += step; idx = idx >= N12 ? idx - N12 : idx;
  }
  pr[tid] = ar; pi[tid] = ai;
  tree256_2(pr, pi, tid);
  if (tid == 0) { cf32 y; y.r = pr[0]; y.i = pi[0]; Y[(size_t)o * LSN_NZC + j] = y; }
}

// grid (839 lags, nroots, nocc).  D[root][839]: DFT of the root sequence, V[m] = exp(+2 pi j m / 839)
__global__ __launch_bounds__(256) void k_prach_corr(const cf32* __restrict__ Y, const cf32* __restrict__ D, const cf32* __restrict__ V, int nroots,
                                                    float* __restrict__ corr)
{
  __shared__ float pr[256], pi[256];
  const int k = blockIdx.x, root = blockIdx.y, o = blockIdx.z, tid = threadIdx.x;
  const cf32* y = Y + (size_t)o * LSN_NZC;
  const cf32* d = D + (size_t)root * LSN_NZC;
  float ar = 0.0f, ai = 0.0f;
  for (int j = tid; j < LSN_NZC; j += 256) {
    const cf32 p = cmul(cmulconj(y[j], d[j]), V[(j * k) % LSN_NZC]);
    ar = ar + p.r; ai = ai + p.i;
  }
  pr[tid] = ar; pi[tid] = ai;
  tree256_2(pr, pi, tid);
  if (tid == 0) corr[((size_t)o * nroots + root) * LSN_NZC + k] = pr[0] * pr[0] + pi[0] * pi[0];
}

// grid (nroots, nocc).  out[(o * nroots + root) * 130 + {0: mean, 1: unused, 2 + 2 w: peak of window w, 3 + 2 w: its lag}]
__global__ __launch_bounds__(256) void k_prach_peaks(const float* __restrict__ corr, int nroots, int ncs, int nwin, float* __restrict__ out)
{
  __shared__ float pr[256], pi[256];
  const int root = blockIdx.x, o = blockIdx.y, tid = threadIdx.x;
  const float* c = corr + ((size_t)o * nroots + root) * LSN_NZC;
  float* res = out + ((size_t)o * nroots + root) * 130;
  float p = 0.0f;
  for (int i = tid; i < LSN_NZC; i += 256) p = p + c[i];
  pr[tid] = p; pi[tid] = 0.0f;
  tree256_2(pr, pi, tid);
  if (tid == 0) { res[0] = pr[0] / (float)LSN_NZC; res[1] = 0.0f; }
  if (tid < nwin && tid < 64) {
    const int start = (LSN_NZC - tid * ncs) % LSN_NZC, win = ncs ? ncs : LSN_NZC;
    float peak = 0.0f; int off = 0;
    for (int k = 0; k < win; k++) {
      const float v = c[start + k];
      if (v > peak) { peak = v; off = k; }
    }
    res[2 + 2 * tid] = peak; res[3 + 2 * tid] = (float)off;
  }
}

void lsn_launch_prach(const cf32* iq, const uint64_t* occ_off, uint32_t nocc, const cf32* W, const cf32* D, const cf32* V, int N12, int Ncp, int b0,
                      int nroots, int ncs, int nwin, cf32* Y, float* corr, float* out, hipStream_t s)
{
  if (!nocc) return;
  LSN_LAUNCH(k_prach_bins, dim3(LSN_NZC, nocc), dim3(256), 0, s, iq, occ_off, W, N12, Ncp, b0, Y);
  LSN_LAUNCH(k_prach_corr, dim3(LSN_NZC, nroots, nocc), dim3(256), 0, s, Y, D, V, nroots, corr);
  LSN_LAUNCH(k_prach_peaks, dim3(nroots, nocc), dim3(256), 0, s, corr, nroots, ncs, nwin, out);
}
