// stage_a.hip - gfx950 kernels for the control-plane half of the LTESniffer worker:
//   OFDM demodulation, CRS channel estimation, PCFICH, PDCCH LLR extraction, CCE power, exhaustive PDCCH Viterbi.
// These replace what DCISearch::search obtains from srsran_ue_dl_decode_fft_estimate
// (/root/reference/src/src/DCISearch.cc:562) and every srsran_pdcch_decode_msg_limit_avg_llr_power call of the blind
// search (/root/reference/src/src/DCISearch.cc:133 -> /root/reference/lib/src/phy/falcon_phch/falcon_pdcch.c:110-170).
// Layout: one workgroup per (subframe, antenna, OFDM symbol) for the FFT (LDS-staged radix-2^3 passes, coalesced
// float2 loads of the IQ); one workgroup per (subframe, antenna, port) for the estimator; one wavefront (64 lanes =
// the 64 trellis states) per (subframe, location, DCI size) for the tail-biting Viterbi with __ballot decision words.
// Float arithmetic is written one rounding per operation (compiled with -ffp-contract=off) so that results are
// bit-identical to the CPU oracle used by the tests.
#include "lsn_dev.h"
#include <algorithm>
#include <type_traits>
#include <cstdlib>

#define SQRT2F 1.41421356237309504880f

__device__ __forceinline__ cf32 cmul(cf32 a, cf32 b) { cf32 c; c.r = a.r * b.r - a.i * b.i; c.i = a.r * b.i + a.i * b.r; return c; }
__device__ __forceinline__ cf32 cmulconj(cf32 a, cf32 b) { cf32 c; c.r = a.r * b.r + a.i * b.i; c.i = a.i * b.r - a.r * b.i; return c; }

// ------------------------------------------------------------------------------------------------ OFDM
template <int R>
__device__ __forceinline__ void fft_pass(cf32* a, const cf32* w, int s, int N, int lgN, int tid)
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

// rbp_part (optional): [sf][14][128] - the per-symbol term of SubframePower::computePower (SubframePower.cc:26-30: mean |x|^2 over the 12 REs of every PRB of
// antenna 0), written here from the symbol the workgroup has just produced; k_rb_power then only adds the 14 terms of a PRB in symbol order (rounds 1-4: a
// kernel that read the whole grid a second time, 0.13 MB per subframe)
__device__ __forceinline__ void ofdm_rb_power(const LsnCellDev& c, const cf32* out, float* __restrict__ rbp_part, int sf, int l, int tid)
{
  __syncthreads();  // the row this workgroup stored is read back (L2)
  if (tid < (int)c.nof_prb) {
    float s = 0.0f;
    for (int k = 0; k < 12; k++) { const cf32 x = out[tid * 12 + k]; s = s + (x.r * x.r + x.i * x.i); }
    rbp_part[((size_t)sf * 14 + l) * 128 + tid] = s / 12.0f;
  }
}
__global__ __launch_bounds__(256) void k_ofdm(LsnCellDev c, const cf32* __restrict__ iq, const uint32_t* __restrict__ dphi_sf,
                                              cf32* __restrict__ grid, float* __restrict__ rbp_part)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int N = (int)c.N, lgN = (int)c.lgN, tid = threadIdx.x;
  cf32* a = (cf32*)smem;
  cf32* w = a + N;
  const int nsym = (int)c.nsym;  // 14, or 12 with the extended cyclic prefix (rows 12, 13 of the grid are never written)
  const int blk = blockIdx.x, l = blk % nsym, rx = (blk / nsym) % (int)c.nof_rx, sf = blk / (nsym * (int)c.nof_rx);
  const int cp0 = 160 * N / 2048, cp1 = 144 * N / 2048;
  const int slot = l / 7, ls = l % 7;
  // 36.211 Table 6.12-1: normal CP 160 / 144 samples (at N = 2048), extended CP N / 4 on every symbol
  const int pos = c.cp ? l * (N + N / 4) + N / 4 : slot * (cp0 + 6 * cp1 + 7 * N) + cp0 + ls * (N + cp1);
  const cf32* in = iq + ((size_t)sf * c.iq_nant + rx) * c.sflen + pos;
  const uint32_t dphi = dphi_sf ? dphi_sf[sf] : 0u;
  const int nre = (int)c.nre;
  cf32* out = grid + (((size_t)sf * c.nof_rx + rx) * 14 + l) * nre;
  if (c.twiddle3) {
    // 15 MHz, N = 1536 = 3 x 512: x_r[m] = x[3 m + r] -> three 512-point transforms side by side in LDS, then
    // X[k] = (F_0[k % 512] + F_1[k % 512] T[k]) + F_2[k % 512] T[2 k mod N] for the 900 carriers that are kept
    const int M = (int)c.nsub;
    for (int n = tid; n < M / 2; n += 256) w[n] = c.twiddle[n];
    for (int n = tid; n < N; n += 256) {
      cf32 x = in[n];
      if (dphi != 0u) {
        uint32_t ph = (uint32_t)(pos + n) * dphi;
        cf32 rot = cmul(c.nco_coarse[ph >> 20], c.nco_fine[(ph >> 10) & 1023u]);
        x = cmul(x, rot);
      }
      const int m = n / 3, r = n - 3 * m;
      a[r * M + (int)(__brev((unsigned)m) >> (32 - lgN))] = x;
    }
    __syncthreads();
    for (int s = 0; s < lgN; s += 3) {  // lgN = 9: three radix-8 passes per block
      for (int r = 0; r < 3; r++) fft_pass<3>(a + r * M, w, s, M, lgN, tid);
      __syncthreads();
    }
    const cf32* __restrict__ T = c.twiddle3;
    for (int k = tid; k < nre; k += 256) {
      const int bin = (k < nre / 2) ? (N - nre / 2 + k) : (k - nre / 2 + 1), kq = bin & (M - 1);
      int b2 = 2 * bin;
      b2 = b2 >= N ? b2 - N : b2;
      const cf32 t1 = cmul(a[M + kq], T[bin]), t2 = cmul(a[2 * M + kq], T[b2]);
      const float sr = a[kq].r + t1.r, si = a[kq].i + t1.i;
      cf32 X;
      X.r = sr + t2.r;
      X.i = si + t2.i;
      out[k] = X;
    }
    if (rbp_part && rx == 0) ofdm_rb_power(c, out, rbp_part, sf, l, tid);
    return;
  }
  for (int n = tid; n < N / 2; n += 256) w[n] = c.twiddle[n];
  for (int n = tid; n < N; n += 256) {
    cf32 x = in[n];
    if (dphi != 0u) {
      uint32_t ph = (uint32_t)(pos + n) * dphi;
      cf32 rot = cmul(c.nco_coarse[ph >> 20], c.nco_fine[(ph >> 10) & 1023u]);
      x = cmul(x, rot);
    }
    a[__brev((unsigned)n) >> (32 - lgN)] = x;
  }
  __syncthreads();
  int s = 0;
  while (s < lgN) {
    int r = lgN - s;
    if (r >= 3) { fft_pass<3>(a, w, s, N, lgN, tid); s += 3; }
    else if (r == 2) { fft_pass<2>(a, w, s, N, lgN, tid); s += 2; }
    else { fft_pass<1>(a, w, s, N, lgN, tid); s += 1; }
    __syncthreads();
  }
  for (int k = tid; k < nre; k += 256) {
    int bin = (k < nre / 2) ? (N - nre / 2 + k) : (k - nre / 2 + 1);
    out[k] = a[bin];
  }
  if (rbp_part && rx == 0) ofdm_rb_power(c, out, rbp_part, sf, l, tid);
}

void lsn_launch_ofdm(const LsnCellDev& c, const cf32* iq, const uint32_t* dphi, cf32* grid, uint32_t nsf, hipStream_t s, float* rbp_part)
{
  size_t lds = sizeof(cf32) * (c.N + c.N / 2);
  LSN_LAUNCH(k_ofdm, dim3(nsf * c.nof_rx * c.nsym), dim3(256), lds, s, c, iq, dphi, grid, rbp_part);
}

// ------------------------------------------------------------------------------------------------ channel estimation
__device__ __forceinline__ int crs_koff(const LsnCellDev& c, int port, int s)
{
  // ports 0, 1: pilot symbols 0, 4, 7, 11 (s = 0..3); ports 2, 3: symbols 1, 8 (s = 0, 1) with v = 3 (n_s mod 2) / 3 + 3 (n_s mod 2), 36.211 6.10.1.2
  int v = (port == 0) ? ((s & 1) ? 3 : 0) : (port == 1) ? ((s & 1) ? 0 : 3) : (port == 2) ? 3 * s : 3 + 3 * s;
  return (v + (int)(c.id % 6)) % 6;
}

// raw[(sf*A+rx)*P+p][8] = {noise_sum, ls_r_sum, ls_i_sum, cepow_sum, cfo_r_sum, cfo_i_sum, -, -}
__global__ __launch_bounds__(256) void k_chest(LsnCellDev c, const cf32* __restrict__ grid, const uint32_t* __restrict__ sf_idx_arr,
                                               cf32* __restrict__ ce, float* __restrict__ raw)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nre = (int)c.nre, nref = (int)c.nref, tid = threadIdx.x;
  const int P = (int)c.nof_ports, A = (int)c.nof_rx;
  const int blk = blockIdx.x, p = blk % P, rx = (blk / P) % A, sf = blk / (P * A);
  cf32* ls = (cf32*)smem;           // [4*nref]
  cf32* sm = ls + 4 * nref;         // [4*nref]
  float* part = (float*)(sm + 4 * nref);  // [6][256]
  const int S = p < 2 ? 4 : 2;  // pilot symbols of this port
  // pilot symbols: ports 0, 1 on symbols 0 and N_symb - 3 of both slots (0, 4, 7, 11; extended CP 0, 3, 6, 9), ports 2, 3 on symbol 1 of both slots (1, 8 / 1, 7)
  const int nsl = (int)c.nslot, nsym = (int)c.nsym;
  const int sym[4] = {p < 2 ? 0 : 1, p < 2 ? nsl - 3 : nsl + 1, nsl, 2 * nsl - 3};
  const cf32* g = grid + ((size_t)sf * A + rx) * 14 * nre;
  const cf32* crs = c.crs + ((size_t)sf_idx_arr[sf] * P + p) * 4 * nref;
  const int n4 = S * nref;
  for (int i = tid; i < n4; i += 256) {
    int s = i / nref, m = i - s * nref;
    ls[i] = cmulconj(g[sym[s] * nre + 6 * m + crs_koff(c, p, s)], crs[i]);
  }
  __syncthreads();
  for (int i = tid; i < n4; i += 256) {
    int s = i / nref, m = i - s * nref;
    float ar = 0.0f, ai = 0.0f;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      int q = m + j - 2;
      if (q < 0 || q >= nref) continue;
      ar = ar + c.taps[j] * ls[s * nref + q].r;
      ai = ai + c.taps[j] * ls[s * nref + q].i;
    }
    sm[i].r = ar; sm[i].i = ai;
  }
  __syncthreads();
  // six strided partial sums per thread, then one shared tree (o_reduce256 order)
  float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f, p4 = 0.0f, p5 = 0.0f;
  for (int i = tid; i < n4; i += 256) {
    float dr = sm[i].r - ls[i].r, di = sm[i].i - ls[i].i;
    p0 = p0 + (dr * dr + di * di);
    p1 = p1 + ls[i].r;
    p2 = p2 + ls[i].i;
    p3 = p3 + (sm[i].r * sm[i].r + sm[i].i * sm[i].i);
  }
  if (p < 2)  // the slot-to-slot correlation of the CFO estimate: ports 2, 3 change their subcarriers between the slots and stay out
    for (int i = tid; i < 2 * nref; i += 256) {
      cf32 t = (i < nref) ? cmulconj(ls[2 * nref + i], ls[i]) : cmulconj(ls[3 * nref + (i - nref)], ls[nref + (i - nref)]);
      p4 = p4 + t.r;
      p5 = p5 + t.i;
    }
  part[0 * 256 + tid] = p0; part[1 * 256 + tid] = p1; part[2 * 256 + tid] = p2;
  part[3 * 256 + tid] = p3; part[4 * 256 + tid] = p4; part[5 * 256 + tid] = p5;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
#pragma unroll
      for (int q = 0; q < 6; q++) part[q * 256 + tid] = part[q * 256 + tid] + part[q * 256 + tid + s];
    }
    __syncthreads();
  }
  if (tid < 6) raw[(((size_t)sf * A + rx) * P + p) * 8 + tid] = part[tid * 256];
  // interpolation: frequency (linear, pilot spacing 6, edges extrapolated) then time (0,4,7,11; 12,13 extrapolated)
  cf32* co = ce + (((size_t)sf * P + p) * A + rx) * 14 * nre;
  for (int k = tid; k < nre; k += 256) {
    cf32 row[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
      if (s >= S) break;
      int koff = crs_koff(c, p, s);
      const cf32* pl = sm + s * nref;
      int m = (k - koff) >= 0 ? (k - koff) / 6 : 0;
      if (m > nref - 2) m = nref - 2;
      float dr = (pl[m + 1].r - pl[m].r) / 6.0f, di = (pl[m + 1].i - pl[m].i) / 6.0f;
      float f = (float)(k - (6 * m + koff));
      row[s].r = pl[m].r + dr * f;
      row[s].i = pl[m].i + di * f;
    }
    if (p >= 2) {  // ports 2, 3: one line through their two pilot symbols for the whole subframe
      const cf32 c1 = row[0], c8 = row[1];
      const int la = sym[0], lb = sym[1];
      const float dl = (float)(lb - la);
      const float dr = (c8.r - c1.r) / dl, di = (c8.i - c1.i) / dl;
      co[la * nre + k] = c1; co[lb * nre + k] = c8;
      for (int l = 0; l < nsym; l++) {
        if (l == la || l == lb) continue;
        cf32 v; v.r = c1.r + dr * (float)(l - la); v.i = c1.i + di * (float)(l - la); co[l * nre + k] = v;
      }
      continue;
    }
    cf32 c0 = row[0], c4 = row[1], c7 = row[2], c11 = row[3];
    const int q0 = sym[0], q1 = sym[1], q2 = sym[2], q3 = sym[3];
    const float w01 = (float)(q1 - q0), w12 = (float)(q2 - q1), w23 = (float)(q3 - q2);
    float d01r = (c4.r - c0.r) / w01, d01i = (c4.i - c0.i) / w01;
    float d12r = (c7.r - c4.r) / w12, d12i = (c7.i - c4.i) / w12;
    float d23r = (c11.r - c7.r) / w23, d23i = (c11.i - c7.i) / w23;
    co[q0 * nre + k] = c0; co[q1 * nre + k] = c4; co[q2 * nre + k] = c7; co[q3 * nre + k] = c11;
    for (int l = q0 + 1; l < q1; l++) { cf32 v; v.r = c0.r + d01r * (float)(l - q0); v.i = c0.i + d01i * (float)(l - q0); co[l * nre + k] = v; }
    for (int l = q1 + 1; l < q2; l++) { cf32 v; v.r = c4.r + d12r * (float)(l - q1); v.i = c4.i + d12i * (float)(l - q1); co[l * nre + k] = v; }
    for (int l = q2 + 1; l < q3; l++) { cf32 v; v.r = c7.r + d23r * (float)(l - q2); v.i = c7.i + d23i * (float)(l - q2); co[l * nre + k] = v; }
    for (int l = q3 + 1; l < nsym; l++) { cf32 v; v.r = c11.r + d23r * (float)(l - q3); v.i = c11.i + d23i * (float)(l - q3); co[l * nre + k] = v; }   // behind the last pilot: its slope continues
  }
}

void lsn_launch_chest(const LsnCellDev& c, const cf32* grid, const uint32_t* sf_idx, cf32* ce, float* raw, uint32_t nsf, hipStream_t s)
{
  size_t lds = sizeof(cf32) * 8 * c.nref + sizeof(float) * 6 * 256;
  LSN_LAUNCH(k_chest, dim3(nsf * c.nof_rx * c.nof_ports), dim3(256), lds, s, c, grid, sf_idx, ce, raw);
}

__global__ void k_chest_fin(LsnCellDev c, const float* __restrict__ raw, LsnChest* __restrict__ out, uint32_t nsf)
{
  uint32_t sf = blockIdx.x * blockDim.x + threadIdx.x;
  if (sf >= nsf) return;
  const int A = (int)c.nof_rx, P = (int)c.nof_ports;
  LsnChest o;
  float ns = 0.0f, rs = 0.0f, cp = 0.0f, cr = 0.0f, ci = 0.0f;
  for (int q = 0; q < LSN_MAX_RX * LSN_MAX_PORTS; q++) { o.noise[q] = 0.0f; o.rsrp[q] = 0.0f; o.cepow[q] = 0.0f; }
  for (int rx = 0; rx < A; rx++)
    for (int p = 0; p < P; p++) {
      const float* r = raw + (((size_t)sf * A + rx) * P + p) * 8;
      const float n = (float)((p < 2 ? 4 : 2) * c.nref);  // pilots of the port in one subframe
      float noise = r[0] / n, mr = r[1] / n, mi = r[2] / n, cepow = r[3] / n;
      float rsrp = mr * mr + mi * mi;
      o.noise[rx * P + p] = noise; o.rsrp[rx * P + p] = rsrp; o.cepow[rx * P + p] = cepow;
      ns = ns + noise; rs = rs + rsrp; cp = cp + cepow;
      cr = cr + r[4]; ci = ci + r[5];
    }
  float cnt = (float)(A * P);
  o.noise_avg = ns / cnt; o.rsrp_avg = rs / cnt; o.chan_ref = cp; o.corr_r = cr; o.corr_i = ci;
  o.pad[0] = o.pad[1] = o.pad[2] = 0.0f;
  out[sf] = o;
}
void lsn_launch_chest_fin(const LsnCellDev& c, const float* raw, LsnChest* out, uint32_t nsf, hipStream_t s)
{
  LSN_LAUNCH(k_chest_fin, dim3((nsf + 63) / 64), dim3(64), 0, s, c, raw, out, nsf);
}

// ------------------------------------------------------------------------------------------------ control region
// equalise the 4 data REs of one REG (36.211 6.2.4) -> 4 QPSK symbols; single port: MRC/(|h|^2+noise), two ports: SFBC, four: SFBC-FSTD
// g / ce point at symbol l of antenna 0 (port 0); rs = distance between antennas, ps = distance between ports (in REs), so the
// same arithmetic runs on the global grids (rs = 14 nre, ps = A 14 nre) and on rows staged in LDS (rs = nre, ps = A nre)
__device__ __forceinline__ void reg_equalise(const LsnCellDev& c, const cf32* __restrict__ g, const cf32* __restrict__ ce, float noise,
                                             int l, int k0, cf32* x, int rs, int ps)
{
  const int A = (int)c.nof_rx;
  int kk[4], n = 0;
  if ((c.reg_w6 >> l) & 1u) {  // the REG spans 6 REs, two of them CRS positions (symbol 0; symbol 1: ports 2, 3 of a four-port cell; symbol 3 with the extended CP)
    for (int k = k0; k < k0 + 6; k++)
      if ((k % 3) != (int)(c.id % 3)) { if (n < 4) kk[n] = k; n++; }
  } else {
    for (int k = 0; k < 4; k++) kk[k] = k0 + k;
  }
  if (c.nof_ports == 1) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float nr = 0.0f, ni = 0.0f, den = 0.0f;
      for (int rx = 0; rx < A; rx++) {
        cf32 y = g[rx * rs + kk[i]];
        cf32 h = ce[rx * rs + kk[i]];
        cf32 t = cmulconj(y, h);
        float hp = h.r * h.r + h.i * h.i;
        if (rx == 0) { nr = t.r; ni = t.i; den = hp; } else { nr = nr + t.r; ni = ni + t.i; den = den + hp; }
      }
      den = den + noise;
      x[i].r = nr / den; x[i].i = ni / den;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
      float x0r = 0, x0i = 0, x1r = 0, x1i = 0, hh = 0;
      // two ports: SFBC on ports (0, 1); four ports (SFBC-FSTD, 36.211 6.3.4.3): first pair of the quadruplet on ports (0, 2), second on (1, 3)
      const int pa = (c.nof_ports == 4 && i == 2) ? 1 : 0, pb = c.nof_ports == 4 ? pa + 2 : 1;
      for (int rx = 0; rx < A; rx++) {
        const int bg = rx * rs, b0 = pa * ps + rx * rs, b1 = pb * ps + rx * rs;
        cf32 r0 = g[bg + kk[i]], r1 = g[bg + kk[i + 1]];
        cf32 h00 = ce[b0 + kk[i]], h01 = ce[b0 + kk[i + 1]], h10 = ce[b1 + kk[i]], h11 = ce[b1 + kk[i + 1]];
        float hp = (h00.r * h00.r + h00.i * h00.i) + (h11.r * h11.r + h11.i * h11.i);
        cf32 a = cmulconj(r0, h00), b = cmulconj(h11, r1), cc = cmulconj(h10, r0), d = cmulconj(r1, h01);
        float t0r = a.r + b.r, t0i = a.i + b.i, t1r = d.r - cc.r, t1i = d.i - cc.i;
        if (rx == 0) { x0r = t0r; x0i = t0i; x1r = t1r; x1i = t1i; hh = hp; }
        else { x0r = x0r + t0r; x0i = x0i + t0i; x1r = x1r + t1r; x1i = x1i + t1i; hh = hh + hp; }
      }
      x[i].r = x0r / hh * SQRT2F; x[i].i = x0i / hh * SQRT2F;
      x[i + 1].r = x1r / hh * SQRT2F; x[i + 1].i = x1i / hh * SQRT2F;
    }
  }
}

__global__ __launch_bounds__(64) void k_pcfich(LsnCellDev c, const cf32* __restrict__ grid, const cf32* __restrict__ ce,
                                               const LsnChest* __restrict__ ch, const uint32_t* __restrict__ sf_idx_arr,
                                               uint32_t* __restrict__ cfi_out, float* __restrict__ corr_out)
{
  __shared__ float llr[32];
  const int sf = blockIdx.x, lane = threadIdx.x;
  const cf32* g = grid + (size_t)sf * c.nof_rx * 14 * c.nre;
  const cf32* e = ce + (size_t)sf * c.nof_ports * c.nof_rx * 14 * c.nre;
  const uint8_t* scr = c.pcfich_scr + sf_idx_arr[sf] * 32;
  if (lane < 4) {
    cf32 x[4];
    reg_equalise(c, g, e, ch[sf].noise_avg, 0, (int)c.pcfich_k0[lane], x, 14 * (int)c.nre, (int)c.nof_rx * 14 * (int)c.nre);
    for (int j = 0; j < 4; j++) {
      float a = -(x[j].r * SQRT2F), b = -(x[j].i * SQRT2F);
      llr[8 * lane + 2 * j] = scr[8 * lane + 2 * j] ? -a : a;
      llr[8 * lane + 2 * j + 1] = scr[8 * lane + 2 * j + 1] ? -b : b;
    }
  }
  __syncthreads();
  if (lane == 0) {
    // CFI codewords <0,1,1>, <1,0,1>, <1,1,0> repeated (36.212 5.3.4): bit i of codeword w is (i % 3 != w)
    uint32_t best = 0; float bestc = 0.0f;
    for (int w = 0; w < 3; w++) {
      float acc = 0.0f;
      for (int i = 0; i < 32; i++) acc = acc + (((i % 3) != w) ? llr[i] : -llr[i]);
      corr_out[sf * 3 + w] = acc;
      if (w == 0 || acc > bestc) { bestc = acc; best = (uint32_t)w; }
    }
    cfi_out[sf] = best + 1;
  }
}
void lsn_launch_pcfich(const LsnCellDev& c, const cf32* grid, const cf32* ce, const LsnChest* ch, const uint32_t* sf_idx, uint32_t* cfi, float* corr, uint32_t nsf, hipStream_t s)
{
  LSN_LAUNCH(k_pcfich, dim3(nsf), dim3(64), 0, s, c, grid, ce, ch, sf_idx, cfi, corr);
}

// One thread per REG in (symbol, frequency) order: consecutive lanes read consecutive REGs of the received grid and of the channel
// estimates (32 / 48 contiguous bytes each, so a wavefront reads whole rows segments), equalise and write the 8 LLRs of the quadruplet to
// its place in the PDCCH order (reg_q: inverse of the REG interleaver, 36.211 6.8.5).  Same arithmetic as the gather form.
__global__ __launch_bounds__(256) void k_pdcch_llr(LsnCellDev c, const cf32* __restrict__ grid, const cf32* __restrict__ ce,
                                                   const LsnChest* __restrict__ ch, const uint32_t* __restrict__ sf_idx_arr,
                                                   const uint32_t* __restrict__ cfi_arr, float* __restrict__ llr)
{
  const int sf = blockIdx.y;
  const uint32_t cfi = cfi_arr[sf];
  const int nre = (int)c.nre, A = (int)c.nof_rx, n0 = nre / 6, n1 = nre / 4;
  const int nat = blockIdx.x * 256 + threadIdx.x;
  // natural REG order: symbol by symbol; a symbol that carries CRS (c.reg_w6) has nre / 6 REGs of 6 REs, the others nre / 4 of 4
  int l = 0, k0 = 0, r = nat;
  for (;; l++) {
    const int six = (int)((c.reg_w6 >> l) & 1u), cnt = six ? n0 : n1;
    if (r < cnt || l == 3) { k0 = (six ? 6 : 4) * r; break; }
    r -= cnt;
  }
  if (k0 >= nre) return;
  if ((uint32_t)l >= cfi + (c.nof_prb <= 10 ? 1u : 0u) || nat >= 800) return;  // 36.211 6.7: one more control symbol at <= 10 PRB
  const uint32_t q = c.reg_q[(cfi - 1) * 800 + nat];
  if (q >= c.nof_cce[cfi - 1] * 9) return;  // PCFICH / PHICH REG (0xFFFF) or behind the last whole CCE
  const cf32* g = grid + ((size_t)sf * A * 14 + l) * nre;
  const cf32* e = ce + ((size_t)sf * c.nof_ports * A * 14 + l) * nre;
  const uint8_t* sc = c.pdcch_scr + (size_t)sf_idx_arr[sf] * LSN_LLR_STRIDE + 8 * q;
  cf32 x[4];
  reg_equalise(c, g, e, ch[sf].noise_avg, l, k0, x, 14 * nre, A * 14 * nre);
  float v[8];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    float a = -(x[j].r * SQRT2F), b = -(x[j].i * SQRT2F);
    v[2 * j] = sc[2 * j] ? -a : a;
    v[2 * j + 1] = sc[2 * j + 1] ? -b : b;
  }
  float4 o0, o1;
  o0.x = v[0]; o0.y = v[1]; o0.z = v[2]; o0.w = v[3]; o1.x = v[4]; o1.y = v[5]; o1.z = v[6]; o1.w = v[7];
  float4* o = reinterpret_cast<float4*>(llr + (size_t)sf * LSN_LLR_STRIDE + 8 * q);
  o[0] = o0; o[1] = o1;
}
void lsn_launch_pdcch_llr(const LsnCellDev& c, const cf32* grid, const cf32* ce, const LsnChest* ch, const uint32_t* sf_idx, const uint32_t* cfi, float* llr, uint32_t nsf, hipStream_t s)
{
  const uint32_t nreg = c.nre / 6 + (c.nof_prb <= 10 ? 3u : 2u) * (c.nre / 4);  // REGs of the widest control region (an upper bound: symbols with CRS hold fewer)
  LSN_LAUNCH(k_pdcch_llr, dim3((nreg + 255) / 256, nsf), dim3(256), 0, s, c, grid, ce, ch, sf_idx, cfi, llr);
}

// falcon_pdcch.c:595-620: mean |llr| over the 72 LLRs of each CCE, accumulated in double in index order
__global__ __launch_bounds__(128) void k_cce_power(LsnCellDev c, const float* __restrict__ llr, const uint32_t* __restrict__ cfi_arr,
                                                   float* __restrict__ pw)
{
  const int sf = blockIdx.x, cce = threadIdx.x;
  if (cce >= LSN_CCE_STRIDE) return;
  float out = 0.0f;
  if ((uint32_t)cce < c.nof_cce[cfi_arr[sf] - 1]) {
    const float* l = llr + (size_t)sf * LSN_LLR_STRIDE + 72 * cce;
    double mean = 0.0;
    for (int i = 0; i < 72; i++) mean += (double)fabsf(l[i]);
    out = (float)(mean / 72.0);
  }
  pw[sf * LSN_CCE_STRIDE + cce] = out;
}
void lsn_launch_cce_power(const LsnCellDev& c, const float* llr, const uint32_t* cfi, float* pw, uint32_t nsf, hipStream_t s)
{
  LSN_LAUNCH(k_cce_power, dim3(nsf), dim3(128), 0, s, c, llr, cfi, pw);
}

// ------------------------------------------------------------------------------------------------ search space
// srsran_pdcch_validate_location (falcon_pdcch.c:223-250) in closed form (36.213 9.1.1): 0 invalid, 1 valid but ambiguous
// with aggregation level l-1 at the same CCE, 2 valid.  Same arithmetic as lsn::SearchSpace::validate on the host.
__device__ __forceinline__ uint32_t ss_validate(uint32_t n, uint32_t ncce, uint32_t l, uint32_t nsubframe, uint32_t rnti)
{
  const bool ue = rnti >= 0x000Bu && rnti <= 0xFFF3u;
  if (!ue && !((rnti >= 1u && rnti <= 10u) || rnti >= 0xFFFDu)) return 0;
  uint32_t Yk = rnti;
  if (ue)
    for (uint32_t m = 0; m <= nsubframe; m++) {
      const uint32_t x = 39827u * Yk;
      const int t = (int)(x & 0xFFFFu) - (int)(x >> 16);
      Yk = (uint32_t)(t < 0 ? t + 65537 : t);
    }
  auto member = [&](uint32_t lv) -> bool {
    const uint32_t L = 1u << lv;
    if (n < L) return false;
    if (lv >= 2) {  // common search space: L = 4, 8 inside the first 16 CCEs
      const uint32_t lim = (n < 16u ? n : 16u) / L;
      if ((ncce & (L - 1)) == 0 && ncce / L < lim && ncce + L <= n) return true;
    }
    if (!ue || (ncce & (L - 1))) return false;
    const uint32_t M = n / L, q = ncce / L, nc = lv < 2 ? 6u : 2u;
    if (q >= M) return false;
    return (q + M - Yk % M) % M < nc;
  };
  if (!member(l)) return 0;
  if (l > 0 && member(l - 1)) return 1;
  return 2;
}

// ------------------------------------------------------------------------------------------------ PDCCH Viterbi
// tail-biting Viterbi over D = nbits + 16 steps whose symbols are in symw (LDS, signed-byte triples); one wavefront.
// Returns the nbits decoded bits (bit i at position 63 - i) and, in lane 0, the CRC16 remainder XOR the received parity
// (the RNTI of a DCI / the antenna-port mask of the PBCH).
// da = sw . sa + ca, db = sw . sb + cb over four signed bytes: the non-accumulating VOP3P form (the builtin selects v_dot4c plus a move of
// the constant).  gfx90a+ leaves the DOT -> VALU read hazard (3 wait states) to software and the hazard recogniser does not look into inline
// assembly, hence the s_nop - it stalls this wave only, the SIMD issues from its other waves meanwhile.
__device__ __forceinline__ void lsn_dot4x2(int sw, int sa, int ca, int sb, int cb, int& da, int& db)
{
  asm("v_dot4_i32_i8 %0, %2, %3, %4\n\tv_dot4_i32_i8 %1, %2, %5, %6\n\ts_nop 2" : "=&v"(da), "=&v"(db) : "v"(sw), "v"(sa), "v"(ca), "v"(sb), "v"(cb));
}
// bits = bits * 2 + (x < y): the decision of this lane's state is shifted into the lane's own history word (compare + add-with-carry)
__device__ __forceinline__ void lsn_push_lt(int& bits, int x, int y)
{
  asm("v_cmp_lt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(x), "v"(y) : "vcc");
}
// Add-compare-select on DOUBLED metrics shifted by a per-step constant: with the branch metric of predecessor j >> 1 written as dot + k0 and that of
// (j >> 1) | 32 as 765 - (dot + k0) (all three outputs complemented), the comparison  mpa + dot + k0  <>  mpb + 765 - dot - k0  is the comparison
// 2 mpa + e  <>  2 mpb - e  with e = 2 dot + 2 k0 - 765: ONE v_dot4 (sign word +-2, accumulator 2 k0 - 765) instead of two, and both candidates move
// by the same amount per step, so decisions, the order of the end metrics and their ties are those of the plain recursion (M = 2 m - 765 t).
// |M| <= 2 x 765 x 240 steps: no wrap in 32 bits.
__device__ __forceinline__ int lsn_dot4_acc(int sw, int sg, int cc)
{
  int d;
  asm("v_dot4_i32_i8 %0, %1, %2, %3\n\ts_nop 2" : "=&v"(d) : "v"(sw), "v"(sg), "v"(cc));
  return d;
}
// ... for FOUR steps at once: the three wait states between a DOT and the first VALU read of its result are filled by the other three DOTs for e0 (the
// add-compare-select chain consumes e0 first), one s_nop 2 covers e3 - one wait instruction per four steps instead of one per step
__device__ __forceinline__ void lsn_dot4_acc4(int w0, int w1, int w2, int w3, int sg, int cc, int& e0, int& e1, int& e2, int& e3)
{
  asm("v_dot4_i32_i8 %0, %4, %8, %9\n\tv_dot4_i32_i8 %1, %5, %8, %9\n\tv_dot4_i32_i8 %2, %6, %8, %9\n\tv_dot4_i32_i8 %3, %7, %8, %9\n\ts_nop 2"
      : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3) : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(sg), "v"(cc));
}
// four trace-back steps at the compile-time bit positions P .. P + 3 of a history word (see viterbi_tb): H = 2 H + bit P of the word of state H & 63
template <int P>
__device__ __forceinline__ void lsn_tb_walk4(int hw, unsigned& H)
{
  // (the lane select of v_readlane is the low six bits of its operand.  One asm block: left to the compiler the step costs five instructions - shift, extract, or,
  // and a wait state for a hazard that is not there: the recogniser sees the SGPR the previous v_readlane wrote being used as a lane select after the SALU rewrote it)
  unsigned w;
  asm("v_readlane_b32 %1, %2, %0\n\ts_bitcmp1_b32 %1, %3\n\ts_addc_u32 %0, %0, %0\n\t"
      "v_readlane_b32 %1, %2, %0\n\ts_bitcmp1_b32 %1, %4\n\ts_addc_u32 %0, %0, %0\n\t"
      "v_readlane_b32 %1, %2, %0\n\ts_bitcmp1_b32 %1, %5\n\ts_addc_u32 %0, %0, %0\n\t"
      "v_readlane_b32 %1, %2, %0\n\ts_bitcmp1_b32 %1, %6\n\ts_addc_u32 %0, %0, %0"
      : "+s"(H), "=&s"(w) : "v"(hw), "n"(P), "n"(P + 1), "n"(P + 2), "n"(P + 3) : "scc");
}
__device__ __forceinline__ void viterbi_tb(const int* symw, uint32_t D_, uint32_t nbits, const uint16_t* __restrict__ crcw, int lane, unsigned long long& bits_out, uint32_t& rem_out)
{
  __syncthreads();
  const int D = (int)D_;  // <= 80: the payload is returned in 64 bits
  // lane = new state j: input bit b = j&1, predecessors j>>1 and (j>>1)|32; generator masks on the old state
  const int b = lane & 1, s0 = lane >> 1;
  const int c0 = b ^ (__popc(s0 & 0x36) & 1), c1 = b ^ (__popc(s0 & 0x27) & 1), c2 = b ^ (__popc(s0 & 0x2B) & 1);
  // branch metric from predecessor j>>1: sum of c_i ? 255 - q_i : q_i = dot(sign, q - 128) + k0
  const int signs2 = (c0 ? 0xFE : 0x02) | (c1 ? 0xFE00 : 0x0200) | (c2 ? 0xFE0000 : 0x020000);
  const int k0c = (c0 ? 127 : 128) + (c1 ? 127 : 128) + (c2 ? 127 : 128), kap = 2 * k0c - 765;
  const int pa = s0 << 2, pb = (s0 | 32) << 2;  // ds_bpermute byte addresses of the two predecessors
  int m = 0;
  // pass 1 only warms the path metrics up: no decision is kept
  {
    auto step = [&](int e) {
      const int a0 = __builtin_amdgcn_ds_bpermute(pa, m) + e, a1 = __builtin_amdgcn_ds_bpermute(pb, m) - e;
      m = a1 < a0 ? a1 : a0;
    };
    int k = 0;
    for (; k + 4 <= D; k += 4) {  // four steps share one symbol fetch and one wait for the DOT results
      int e0, e1, e2, e3;
      lsn_dot4_acc4(symw[k], symw[k + 1], symw[k + 2], symw[k + 3], signs2, kap, e0, e1, e2, e3);
      step(e0); step(e1); step(e2); step(e3);
    }
    for (; k < D; k++) step(lsn_dot4_acc(symw[k], signs2, kap));
  }
  // passes 2 and 3: every lane shifts the decisions of ITS state into a history word, 32 steps per register (D <= 80: three per pass) -
  // no ballot, no LDS; the trace-back reads the word of the state it stands on with one v_readlane
  int h2[3] = {0, 0, 0}, h3[3] = {0, 0, 0};
  auto sweep = [&](int* h) {
#pragma unroll
    for (int g = 0; g < 3; g++) {
      const int k1 = D < 32 * (g + 1) ? D : 32 * (g + 1);
      auto step = [&](int e) {
        const int a0 = __builtin_amdgcn_ds_bpermute(pa, m) + e, a1 = __builtin_amdgcn_ds_bpermute(pb, m) - e;
        lsn_push_lt(h[g], a1, a0);
        m = a1 < a0 ? a1 : a0;
      };
      int k = 32 * g;
      for (; k + 4 <= k1; k += 4) {
        int e0, e1, e2, e3;
        lsn_dot4_acc4(symw[k], symw[k + 1], symw[k + 2], symw[k + 3], signs2, kap, e0, e1, e2, e3);
        step(e0); step(e1); step(e2); step(e3);
      }
      for (; k < k1; k++) step(lsn_dot4_acc(symw[k], signs2, kap));
    }
  };
  sweep(h2);
  sweep(h3);
  // best end state: minimum metric, lowest index on ties (the metrics are negative by now: biased into an unsigned key)
  unsigned long long key = ((unsigned long long)((unsigned)m ^ 0x80000000u) << 6) | (unsigned)lane;
  for (int off = 32; off > 0; off >>= 1) {
    unsigned long long o2 = __shfl_xor(key, off);
    key = o2 < key ? o2 : key;
  }
  // Trace-back over pass 3, then pass 2 (whose states are the output), on the scalar unit.  The walk st <- (st >> 1) | (d << 5), d = the decision of state
  // st at this step, is kept BIT-REVERSED: with rs = bitrev6(st) it reads rs <- ((rs << 1) | d) & 63, i.e. the decisions are shifted into ONE history
  // register H whose low six bits are the state - three instructions per step (v_readlane, s_bfe, s_lshl1_add: H = 2 H + d), and the decoded
  // bits need no work of their own: the input bit of step k is the decision read six steps further down the walk, so after the walk they all stand
  // in H.  (Rounds 2-5 walked st itself and placed every decoded bit with a 64-bit shift / or under two branches: 10 scalar instructions per step in
  // the pass that is not emitted, 19 in the one that is - more wave-instructions on the scalar unit than on the vector unit for the whole kernel.)
  // The history words change lanes once so that lane j holds the word of state bitrev6(j): the lane select is H & 63 as it stands.
  const int rl = (int)(__brev((unsigned)lane) >> 26) << 2;
  int p2[3], p3[3];
#pragma unroll
  for (int g = 0; g < 3; g++) { p2[g] = __builtin_amdgcn_ds_bpermute(rl, h2[g]); p3[g] = __builtin_amdgcn_ds_bpermute(rl, h3[g]); }
  unsigned H = (unsigned)__builtin_amdgcn_readfirstlane((int)(__brev((unsigned)(key & 63ull)) >> 26));
  // one group of decisions: step k of the group's cnt steps sits at bit k1 - 1 - k of its word, the walk goes up the bit positions
  auto walk = [&](int hw, int cnt) {
    if (cnt >= 4) { lsn_tb_walk4<0>(hw, H);
    if (cnt >= 8) { lsn_tb_walk4<4>(hw, H);
    if (cnt >= 12) { lsn_tb_walk4<8>(hw, H);
    if (cnt >= 16) { lsn_tb_walk4<12>(hw, H);
    if (cnt >= 20) { lsn_tb_walk4<16>(hw, H);
    if (cnt >= 24) { lsn_tb_walk4<20>(hw, H);
    if (cnt >= 28) { lsn_tb_walk4<24>(hw, H);
    if (cnt >= 32) { lsn_tb_walk4<28>(hw, H); } } } } } } } }
    for (int p = cnt & ~3; p < cnt; p++) {
      H = (H << 1) + (((unsigned)__builtin_amdgcn_readlane(hw, (int)(H & 63u)) >> p) & 1u);
    }
  };
  int cnt[3];
#pragma unroll
  for (int g = 0; g < 3; g++) {
    const int k1 = D < 32 * (g + 1) ? D : 32 * (g + 1);
    cnt[g] = k1 > 32 * g ? k1 - 32 * g : 0;
  }
  walk(p3[2], cnt[2]); walk(p3[1], cnt[1]); walk(p3[0], cnt[0]);
  const unsigned rs0 = H & 63u;   // the state pass 2 ends in (reversed)
  unsigned G[3];
#pragma unroll
  for (int g = 2; g >= 0; g--) {
    walk(p2[g], cnt[g]);
    G[g] = cnt[g] >= 32 ? H : (H & ((1u << cnt[g]) - 1u));   // decision of step 32 g + i at bit i
  }
  // all D decisions as one number, the end state on top: the input bit of step k is bit k + 6 of it (D <= 80)
  unsigned __int128 hb = (unsigned __int128)G[0] | ((unsigned __int128)G[1] << 32) | ((unsigned __int128)G[2] << 64) | ((unsigned __int128)rs0 << D);
  hb >>= 6;
  // decoded bit i of the middle pass at position 63 - i (payload); the 16 CRC bits behind it, first bit on top
  const unsigned long long bits = nbits ? (__brevll((unsigned long long)hb) & (~0ull << (64u - nbits))) : 0ull;
  const unsigned int tailcrc = __brev((unsigned)(hb >> nbits) & 0xFFFFu) >> 16;
  // CRC16 (x^16+x^12+x^5+1) of the payload, all lanes at once: payload bit i weighs x^(nbits - 1 - i + 16) mod g (crcw, built by the host per payload
  // size), the remainder is the XOR of the weights of the set bits (rounds 1-3: a 60-step long division on lane 0 - a third of the kernel's vector
  // instructions, issued for one working lane)
  {
    unsigned int v = (lane < (int)nbits && ((bits >> (63 - lane)) & 1ull)) ? (unsigned int)crcw[lane] : 0u;
    for (int off = 32; off > 0; off >>= 1) v ^= __shfl_xor(v, off);
    rem_out = (tailcrc ^ v) & 0xFFFFu;
  }
  bits_out = bits;
}


// One wavefront per (location, size, subframe).  Rate de-matching is a gather through a host-built rank table;
// u8 quantisation 127.5 + 32*llr (truncated); 32-bit path metrics; 3 passes over the tail-biting block, middle pass kept.
// Issue-cost shaping (VALU instructions are what this kernel is made of, 55 % of the whole path's): the three symbols of
// a step are stored as one word of signed bytes (q - 128), so the branch metric of a lane is ONE v_dot4_i32_i8 with the
// lane's constant sign word (sum of c_i ? 255 - q_i : q_i  ==  dot(sign, q - 128) + sum of c_i ? 127 : 128); the decision
// ballots are stored by a uniform, branch-free LDS write; the trace-back covers only passes 2 and 3, fetches 64 ballot
// words per LDS read (one per lane) and then walks them with v_readlane + scalar shifts instead of one dependent LDS
// round trip per step.
// A two-candidates-per-wavefront variant on packed wrapping 16-bit metrics was built and measured in round 4 (bit-identical candidate tables, half the ds_bpermute
// traffic, 40 % more vector instructions, the same launch time and pipeline rate: DESIGN 5.2, tools/viterbi_packed_design.py); it is not part of the product.
// Round 6: the table is no longer exhaustive by default.  The sequential search reads about 180 of the 960 (location, size) slots of a loaded 20 MHz subframe: a
// DCI it accepts at a location closes everything underneath (DCISearch.cc:371-376: the CCEs become occupied), and the decoder was a third of the chip's
// vector work.  Whether a candidate is accepted depends on the RNTI manager's sequential state, so the kernel cannot know - but it can make a SAFE guess from a
// snapshot of that state (the active RNTIs as a bitmap, the primary / secondary split of the meta formats): "this location holds a candidate the search
// will accept" is only claimed when the candidate passes the search's own stateless tests (format, RNTI ranges, search space: inspect_dci_location_recursively
// :139-214) and its RNTI is evergreen, or active in the snapshot and not forbidden.  The levels are decoded in four launches, 8 -> 4 -> 2 -> 1 CCEs; a wavefront
// whose ancestor made the claim leaves without decoding and marks its slot LSN_CAND_NOT_COMPUTED.  The guess errs on the side of decoding (an RNTI activated since
// the snapshot, shortcut discoveries, histogram validations: all "not claimed"); where it claims too much (an RNTI that expired since) the search finds the mark
// and has the slot decoded on demand (Engine::candidateMiss) - the table the search SEES is the exhaustive one, entry for entry.
// acc[sf][location][size]: what the decoded candidate of that slot lets the search do, one byte: bit 0 / 1 = acceptable in the primary pass with a
// search-space verdict of 1 / 2 (verdict 1: the search still looks into the second half of the location, :288-298), bit 2 / 3 = the same in the secondary pass
// (recorded, not used for closing: see below).
__device__ __forceinline__ bool lsn_in_intervals(const uint32_t* iv, uint32_t n, uint32_t r)
{
  bool hit = false;
  for (uint32_t i = 0; i < n && i < 4u; i++) hit = hit || (r >= (iv[i] & 0xFFFFu) && r <= (iv[i] >> 16));
  return hit;
}
__device__ __forceinline__ uint32_t lsn_cand_claim(const LsnPruneCfg& pc, const uint32_t* __restrict__ snap, uint32_t sz, unsigned long long bits, uint32_t rnti, uint32_t ss)
{
  if (ss == 0u || rnti > 0xFFFFu) return 0u;
  const uint32_t primary = snap[2048];
  const bool active = (snap[rnti >> 5] >> (rnti & 31u)) & 1u;
  uint32_t claim = 0;
  for (uint32_t f = 0; f < 9u; f++) {
    if (pc.fmt_size[f] != sz) continue;
    // falcon_pdcch.c:147-148: a payload of the format 0 / 1A size is a format 0 or a format 1A by its first bit; :163 the meta format must be that format
    if ((f == 0u || f == 2u) && ((bits >> 63) == 0ull) != (f == 0u)) continue;
    if (f == 4u && rnti > 0x000Au && rnti < 0xFFFEu) continue;                         // :174 format 1C carries no C-RNTI
    if (rnti > 0x0001u && rnti < 0x000Au && f != 2u && f != 4u) continue;              // :181-197 RA-RNTIs come in 1A / 1C
    const bool valid = lsn_in_intervals(pc.ever[f], pc.n_ever[f], rnti) || (!lsn_in_intervals(pc.forb[f], pc.n_forb[f], rnti) && active);
    if (!valid) continue;
    claim |= ((primary >> f) & 1u) ? (ss == 1u ? 1u : 2u) : (ss == 1u ? 4u : 8u);
  }
  return claim;
}

// level: 3 .. 0 = the locations of that aggregation level only (blockIdx.x counts them), -1 = every location (the exhaustive table of rounds 1-5, pc.on = 0),
// -2 = decode on demand: every slot that is still marked LSN_CAND_NOT_COMPUTED among the (up to 15) locations of the 8-CCE block one_li of subframe one_sf
// (blockIdx.x: 0 = the block's level-3 location, 1-2 level 2, 3-6 level 1, 7-14 level 0; blockIdx.y = size), whatever their ancestors say
__global__ __launch_bounds__(64) void k_viterbi(LsnCellDev c, const float* __restrict__ llr, const float* __restrict__ pw,
                                                const uint32_t* __restrict__ cfi_arr, const uint32_t* __restrict__ sf_idx_arr,
                                                LsnCand* __restrict__ cand, uint32_t* __restrict__ cand4, LsnPruneCfg pc, const uint32_t* __restrict__ snap, uint8_t* __restrict__ acc, int level,
                                                uint32_t one_sf, uint32_t one_li, uint32_t one_sz)
{
  __shared__ __attribute__((aligned(16))) int symw[LSN_MAX_DCI_D + 4];  // per trellis step: (q0 - 128) | (q1 - 128) << 8 | (q2 - 128) << 16, signed bytes
  const int lane = threadIdx.x, sf = level == -2 ? (int)one_sf : (int)blockIdx.z, sz = (int)blockIdx.y;
  const uint32_t ncce_tot = c.nof_cce[cfi_arr[sf] - 1];
  const uint32_t lim = ncce_tot < LSN_MAX_NUM_OF_CCE ? ncce_tot : LSN_MAX_NUM_OF_CCE;
  // location enumeration of srsran_pdcch_ue_locations_all_map (falcon_pdcch.c:321-356): level 3 first, then 2, 1, 0; the slots behind the last location are cleared
  int li, L = -1; uint32_t ncce = 0;
  if (level >= 0) {
    // the per-level launches have the largest count of their level as grid (84 >> level; level 0: three more): an index behind this subframe's count takes
    // one of the 160 - nloc unused slots, so that every slot of the table is written exactly once per chunk
    const uint32_t cnt = lim >> level, nloc = (lim >> 3) + (lim >> 2) + (lim >> 1) + lim;
    uint32_t off = 0, spare = 0;
    for (int l = 3; l > level; l--) { off += lim >> l; spare += (LSN_MAX_NUM_OF_CCE >> l) - (lim >> l); }
    if (blockIdx.x < cnt) { li = (int)(off + blockIdx.x); L = level; ncce = (blockIdx.x % (ncce_tot >> level)) << level; }
    else { li = (int)(nloc + spare + (blockIdx.x - cnt)); if (li >= LSN_MAX_LOC) return; }
  } else if (level == -2) {
    const int bl = blockIdx.x == 0 ? 3 : (blockIdx.x < 3 ? 2 : (blockIdx.x < 7 ? 1 : 0));      // level of this block-local location
    const uint32_t j = blockIdx.x - ((1u << (3 - bl)) - 1u);                                    // ... and its place among the block's locations of that level
    const uint32_t idx = (one_li << (3 - bl)) + j;                                               // index among ALL locations of the level
    if (idx >= (lim >> bl)) return;
    uint32_t off = 0;
    for (int l = 3; l > bl; l--) off += lim >> l;
    li = (int)(off + idx); L = bl; ncce = idx << bl;
    if (!(cand[((size_t)sf * LSN_MAX_LOC + li) * LSN_MAX_SIZES + sz].flags & LSN_CAND_NOT_COMPUTED)) return;   // computed before (by the level launches or an earlier miss)
  } else {
    li = (int)blockIdx.x;
    int r = li;
    for (int l = 3; l >= 0; l--) {
      int cnt = (int)(lim >> l);
      if (r < cnt) { L = l; ncce = ((uint32_t)r % (ncce_tot >> l)) << l; break; }
      r -= cnt;
    }
  }
  LsnCand* out = cand + ((size_t)sf * LSN_MAX_LOC + li) * LSN_MAX_SIZES + sz;
  uint32_t* out4 = cand4 + ((size_t)sf * LSN_MAX_LOC + li) * LSN_MAX_SIZES + sz;   // the search's view of the slot: LSN_CAND_HOT (lsn_types.h)
  uint8_t* aout = acc + ((size_t)sf * LSN_MAX_LOC + li) * LSN_MAX_SIZES + sz;
  bool ok = L >= 0;
  const uint32_t E = ok ? (72u << L) : 0u;
  if (ok && ncce * 72 + E > ncce_tot * 72) ok = false;
  if (ok) {
    for (uint32_t i = 0; i < (1u << L); i++)
      if (pw[sf * LSN_CCE_STRIDE + ncce + i] < 0.7f) ok = false;  // location->sufficient_power (falcon_pdcch.c:610-614)
  }
  if (!ok) {
    if (lane == 0) { out->bits = 0; out->rnti = 0; out->flags = 0; *out4 = 0; if (pc.on) *aout = 0; }
    return;
  }
  if (pc.on && level >= 0 && L < 3) {
    // Does the search come here for this size?  In its primary pass only if no ancestor is accepted there (or this location lies in the second half of an
    // ancestor accepted with verdict 1); in its secondary pass only if, besides, no ancestor was accepted in the primary pass at all (its CCEs are occupied then)
    const uint32_t primary = snap[2048];
    bool has_p = false, has_s = false;
    for (uint32_t f = 0; f < 9u; f++)
      if (pc.fmt_size[f] == (uint32_t)sz) { if ((primary >> f) & 1u) has_p = true; else has_s = true; }
    bool closed_p = false, closed_s = false;
    uint32_t offA = 0;
    for (int la = 3; la > L; la--) {
      const uint32_t idxA = ncce >> la;
      if (idxA < (lim >> la)) {
        const unsigned long long w = *(const unsigned long long*)(acc + ((size_t)sf * LSN_MAX_LOC + offA + idxA) * LSN_MAX_SIZES);   // the eight size slots of the ancestor
        uint32_t m = (uint32_t)(w | (w >> 32));
        m |= m >> 16; m |= m >> 8;
        const bool first_half = ((ncce >> (la - 1)) & 1u) == 0u;
        if ((m & 3u) && (first_half || !(m & 1u))) closed_p = true;
        // (a candidate of the ancestor that only the SECONDARY pass could accept closes nothing here: that pass does not come to the ancestor at all once any
        // location underneath it was accepted in the primary pass - which the launches of the deeper levels have not told yet)
        if (m & 3u) closed_s = true;
      }
      offA += lim >> la;
    }
    if (!((has_p && !closed_p) || (has_s && !closed_s))) {
      if (lane == 0) { out->bits = 0; out->rnti = 0; out->flags = LSN_CAND_NOT_COMPUTED; *out4 = LSN_CAND_HOT(0ull, 0u, LSN_CAND_NOT_COMPUTED); *aout = 0; }
      return;
    }
  }
  const float* e = llr + (size_t)sf * LSN_LLR_STRIDE + ncce * 72;
  const uint32_t nbits = c.sizes[sz], D = nbits + 16, D3 = 3 * D;
  const uint16_t* rank = c.rankmap + sz * 3 * LSN_MAX_DCI_D;
  bool nz = false;
  for (uint32_t t = lane; t < D; t += 64) {
    uint32_t word = 0;
#pragma unroll
    for (uint32_t j = 0; j < 3; j++) {
      float acc = 0.0f;
      bool first = true;
      for (uint32_t k = rank[3 * t + j]; k < E; k += D3) {
        float v = e[k];
        if (v != 0.0f) nz = true;
        if (first) { acc = v; first = false; } else acc = acc + v;
      }
      float q = 127.5f + 32.0f * acc;
      q = q < 0.0f ? 0.0f : q;
      q = q > 255.0f ? 255.0f : q;
      word |= (((uint32_t)(unsigned char)q - 128u) & 0xFFu) << (8 * j);
    }
    symw[t] = (int)word;
  }
  if (__ballot(nz) == 0ull) {  // mean |llr| == 0: the reference skips the decode (falcon_pdcch.c:141)
    if (lane == 0) { out->bits = 0; out->rnti = 0; out->flags = 0; *out4 = 0; if (pc.on) *aout = 0; }
    return;
  }
  unsigned long long bits; uint32_t rnti;
  viterbi_tb(symw, D, nbits, c.crc16_w + sz * 64, lane, bits, rnti);
  if (lane == 0) {
    const uint32_t ss = ss_validate(ncce_tot, ncce, (uint32_t)L, sf_idx_arr[sf], rnti);
    out->bits = bits;
    out->rnti = rnti;
    out->flags = 1u | (ss << 1);  // bit 0: decoded, bits 1-2: search-space match
    *out4 = LSN_CAND_HOT(bits, rnti, 1u | (ss << 1));
    if (pc.on && level >= 0) *aout = (uint8_t)lsn_cand_claim(pc, snap, (uint32_t)sz, bits, rnti, ss);
  }
}
// ------------------------------------------------------------------------------------------------ PBCH / MIB
// srsran_ue_mib_decode on subframe 0 (LTESniffer_Core.cc:382-395).  k_pbch_llr: one workgroup; the 240 PBCH symbols
// (72 centre carriers of symbols 7-10, CRS positions of four ports left out) are equalised like a REG (MRC or SFBC pairs),
// turned into QPSK soft bits and written once raw and once descrambled for each of the four radio-frame positions of the
// 40 ms BCH period (c_init = N_cell_ID).  k_pbch_viterbi: one wavefront per hypothesis, the DCI decoder's tail-biting
// Viterbi on 40 steps (24 MIB bits + CRC16 whose mask tells the number of CRS ports).
// PBCH resource element i in mapping order: symbols 0-3 of slot 1, 72 centre carriers, the CRS positions of four ports left out of the symbols that carry CRS
// (normal CP: symbols 0, 1 -> 48 + 48 + 72 + 72 = 240 elements; extended CP: symbols 0, 1, 3 -> 48 + 48 + 72 + 48 = 216; 36.211 6.6.4)
__device__ __forceinline__ int pbch_count(const LsnCellDev& c) { return c.cp ? 216 : 240; }
__device__ __forceinline__ void pbch_pos(const LsnCellDev& c, int i, int& l, int& k)
{
  const int k0 = (int)c.nre / 2 - 36, l0 = (int)c.nslot;
  int s, j;
  bool crs;
  if (i < 96) { s = i / 48; j = i % 48; crs = true; }
  else if (i < 168) { s = 2; j = i - 96; crs = false; }
  else { s = 3; j = i - 168; crs = c.cp != 0; }
  l = l0 + s;
  if (crs) {
    const int r = (int)(c.id % 3);
    const int d0 = r == 0 ? 1 : 0, d1 = r == 2 ? 1 : 2;  // the two carriers of a group of three that carry data
    k = k0 + 3 * (j >> 1) + ((j & 1) ? d1 : d0);
  } else {
    k = k0 + j;
  }
}
__global__ __launch_bounds__(256) void k_pbch_llr(LsnCellDev c, const cf32* __restrict__ g, const cf32* __restrict__ ce, const LsnChest* __restrict__ ch,
                                                  float* __restrict__ out /* [5][480]: raw, then 4 descrambled */)
{
  const int tid = threadIdx.x, nre = (int)c.nre, A = (int)c.nof_rx;
  const float noise = ch[0].noise_avg;
  const int np = pbch_count(c), E4 = 2 * np;  // symbols / coded bits of one radio frame's PBCH
  if (tid < 480 - E4) out[E4 + tid] = 0.0f;   // (extended CP: the raw row keeps its 480 entries, the 48 behind the 432 soft bits are zero)
  cf32 x0, x1;
  int i0 = -1;
  if (c.nof_ports == 1) {
    if (tid < np) {
      i0 = tid;
      int l, k; pbch_pos(c, tid, l, k);
      float nr = 0.0f, ni = 0.0f, den = 0.0f;
      for (int rx = 0; rx < A; rx++) {
        const size_t b = ((size_t)rx * 14 + l) * nre + k;
        const cf32 t = cmulconj(g[b], ce[b]);
        const float hp = ce[b].r * ce[b].r + ce[b].i * ce[b].i;
        if (rx == 0) { nr = t.r; ni = t.i; den = hp; } else { nr = nr + t.r; ni = ni + t.i; den = den + hp; }
      }
      den = den + noise;
      x0.r = nr / den; x0.i = ni / den;
    }
  } else if (tid < np / 2) {
    i0 = 2 * tid;
    int l, ka, kb, l2; pbch_pos(c, i0, l, ka); pbch_pos(c, i0 + 1, l2, kb);
    float x0r = 0, x0i = 0, x1r = 0, x1i = 0, hh = 0;
    // four ports (SFBC-FSTD): symbol pairs alternate between the port pairs (0, 2) and (1, 3)
    const size_t pa = (c.nof_ports == 4 && (tid & 1)) ? 1 : 0, pb = c.nof_ports == 4 ? pa + 2 : 1;
    for (int rx = 0; rx < A; rx++) {
      const size_t bg = ((size_t)rx * 14 + l) * nre;
      const size_t b0 = ((pa * (size_t)A + rx) * 14 + l) * nre, b1 = ((pb * (size_t)A + rx) * 14 + l) * nre;
      const cf32 r0 = g[bg + ka], r1 = g[bg + kb];
      const cf32 h00 = ce[b0 + ka], h01 = ce[b0 + kb], h10 = ce[b1 + ka], h11 = ce[b1 + kb];
      const float hp = (h00.r * h00.r + h00.i * h00.i) + (h11.r * h11.r + h11.i * h11.i);
      const cf32 a = cmulconj(r0, h00), b = cmulconj(h11, r1), cc = cmulconj(h10, r0), d = cmulconj(r1, h01);
      const float t0r = a.r + b.r, t0i = a.i + b.i, t1r = d.r - cc.r, t1i = d.i - cc.i;
      if (rx == 0) { x0r = t0r; x0i = t0i; x1r = t1r; x1i = t1i; hh = hp; }
      else { x0r = x0r + t0r; x0i = x0i + t0i; x1r = x1r + t1r; x1i = x1i + t1i; hh = hh + hp; }
    }
    x0.r = x0r / hh * SQRT2F; x0.i = x0i / hh * SQRT2F;
    x1.r = x1r / hh * SQRT2F; x1.i = x1i / hh * SQRT2F;
  }
  if (i0 < 0) return;
  const int nsym = c.nof_ports == 1 ? 1 : 2;
  for (int s = 0; s < nsym; s++) {
    const cf32 x = s ? x1 : x0;
    const float v[2] = {-(x.r * SQRT2F), -(x.i * SQRT2F)};
    for (int j = 0; j < 2; j++) {
      const int n = 2 * (i0 + s) + j;
      out[n] = v[j];
      for (int q = 0; q < 4; q++) {
        const uint32_t m = (uint32_t)E4 * (uint32_t)q + (uint32_t)n;   // the scrambling sequence runs over the 4 x E4 bits of the 40 ms period
        const uint32_t cbit = (uint32_t)c.gold_x1[m] ^ (uint32_t)(__popc(c.gold_x2mask[m] & c.id) & 1);
        out[480 * (q + 1) + n] = cbit ? -v[j] : v[j];
      }
    }
  }
}
__global__ __launch_bounds__(64) void k_pbch_viterbi(LsnCellDev c, const float* __restrict__ llr5, LsnCand* __restrict__ out4)
{
  __shared__ __attribute__((aligned(16))) int symw[LSN_MAX_DCI_D + 4];
  const int lane = threadIdx.x, q = blockIdx.x;
  const float* e = llr5 + 480 * (q + 1);
  const uint32_t D = 40, D3 = 120, E = c.cp ? 432u : 480u;
  for (uint32_t t = lane; t < D; t += 64) {
    uint32_t word = 0;
#pragma unroll
    for (uint32_t j = 0; j < 3; j++) {
      float acc = 0.0f;
      bool first = true;
      // radio frame q of the 40 ms period holds bits [E q, E (q + 1)) of the rate-matched sequence: with the extended CP (E = 432) that piece starts
      // (E q) mod 120 positions into the circular buffer, so the first soft bit of buffer entry r is e[(r - off) mod 120]
      int k0 = (int)c.pbch_rank[3 * t + j] - (int)((E * (uint32_t)q) % D3);
      if (k0 < 0) k0 += (int)D3;
      for (uint32_t k = (uint32_t)k0; k < E; k += D3) {
        const float v = e[k];
        if (first) { acc = v; first = false; } else acc = acc + v;
      }
      float qv = 127.5f + 32.0f * acc;
      qv = qv < 0.0f ? 0.0f : qv;
      qv = qv > 255.0f ? 255.0f : qv;
      word |= (((uint32_t)(unsigned char)qv - 128u) & 0xFFu) << (8 * j);
    }
    symw[t] = (int)word;
  }
  unsigned long long bits; uint32_t rem;
  viterbi_tb(symw, D, 24, c.crc16_w + LSN_MAX_SIZES * 64, lane, bits, rem);
  if (lane == 0) { out4[q].bits = bits; out4[q].rnti = rem; out4[q].flags = 1u; }
}
void lsn_launch_pbch(const LsnCellDev& c, const cf32* grid, const cf32* ce, const LsnChest* ch, float* llr5, LsnCand* out4, hipStream_t s)
{
  LSN_LAUNCH(k_pbch_llr, dim3(1), dim3(256), 0, s, c, grid, ce, ch, llr5);
  LSN_LAUNCH(k_pbch_viterbi, dim3(4), dim3(64), 0, s, c, llr5, out4);
}

void lsn_launch_viterbi(const LsnCellDev& c, const float* llr, const float* pw, const uint32_t* cfi, const uint32_t* sf_idx, LsnCand* cand, uint32_t* cand4, uint32_t nsf,
                        const LsnPruneCfg& pc, const uint32_t* snap, uint8_t* acc, hipStream_t s)
{
  if (!pc.on) {
    LSN_LAUNCH(k_viterbi, dim3(LSN_MAX_LOC, c.nsizes, nsf), dim3(64), 0, s, c, llr, pw, cfi, sf_idx, cand, cand4, pc, snap, acc, -1, 0u, 0u, 0u);
    return;
  }
  static_assert((LSN_MAX_NUM_OF_CCE >> 3) + (LSN_MAX_NUM_OF_CCE >> 2) + (LSN_MAX_NUM_OF_CCE >> 1) + LSN_MAX_NUM_OF_CCE + 3 == LSN_MAX_LOC, "the spare slots are taken by the level-0 launch");
  for (int level = 3; level >= 0; level--)
    LSN_LAUNCH(k_viterbi, dim3((LSN_MAX_NUM_OF_CCE >> level) + (level == 0 ? 3 : 0), c.nsizes, nsf), dim3(64), 0, s, c, llr, pw, cfi, sf_idx, cand, cand4, pc, snap, acc, level, 0u, 0u, 0u);
}
// decode on demand: the slots of one 8-CCE block of one subframe that are still marked LSN_CAND_NOT_COMPUTED (the search found one of them)
void lsn_launch_viterbi_block(const LsnCellDev& c, const float* llr, const float* pw, const uint32_t* cfi, const uint32_t* sf_idx, LsnCand* cand, uint32_t* cand4, uint32_t sf, uint32_t block,
                              const LsnPruneCfg& pc, const uint32_t* snap, uint8_t* acc, hipStream_t s)
{
  LsnPruneCfg off = pc;
  off.on = 0;
  LSN_LAUNCH(k_viterbi, dim3(15, c.nsizes, 1), dim3(64), 0, s, c, llr, pw, cfi, sf_idx, cand, cand4, off, snap, acc, -2, sf, block, 0u);
}

// SubframePower::computePower (SubframePower.cc:18-42) linear part: sum over 14 symbols of mean |x|^2 per PRB (antenna 0).  The per-symbol terms come from
// k_ofdm (rbp_part); this kernel adds them in symbol order (the rows of an extended-CP subframe behind symbol 11 are zero, as in the 14-row grid)
__global__ void k_rb_power(LsnCellDev c, const float* __restrict__ part, float* __restrict__ rbp)
{
  const int sf = blockIdx.x, prb = threadIdx.x;
  if (prb >= (int)c.nof_prb) return;
  float acc = 0.0f;
  for (int j = 0; j < 14; j++) acc = acc + part[((size_t)sf * 14 + j) * 128 + prb];
  rbp[sf * 128 + prb] = acc;
}
void lsn_launch_rb_power(const LsnCellDev& c, const float* part, float* rbp, uint32_t nsf, hipStream_t s)
{
  LSN_LAUNCH(k_rb_power, dim3(nsf), dim3(128), 0, s, c, part, rbp);
}

// ------------------------------------------------------------------------------------------------ IQ capture file source
// srsran_ue_sync_zerocopy in file mode (LTESniffer_Core.cc:365): de-interleave the antennas of a block of raw file samples
// ([subframe][sample][antenna]) into the engine's [subframe][antenna][sample] cf32 layout and, with a frequency offset,
// multiply every subframe by rot[n] = exp(-j 2 pi offset_freq n / fs) (the phase restarts in every subframe).  HBM-bound copy.
// FMT: LSN_FILE_CF32 (the reference's file format), LSN_FILE_SC16 / LSN_FILE_SC8 - integer I/Q pairs as the radio sends them over its
// link (srsRAN's SRSRAN_COMPLEX_SHORT_BIN file type), x = (float)integer * scale (the conversion is exact, the product rounds once):
// half / a quarter of the bytes per subframe cross PCIe.
template <int FMT>
__global__ __launch_bounds__(256) void k_file_unpack(const void* __restrict__ raw, const cf32* __restrict__ rot, uint32_t sflen, uint32_t nant,
                                                     float scale, cf32* __restrict__ out)
{
  const uint32_t n = blockIdx.x * 256 + threadIdx.x, a = blockIdx.y, sf = blockIdx.z;
  if (n >= sflen) return;
  const size_t src = ((size_t)sf * sflen + n) * nant + a;
  cf32 x;
  if (FMT == 1) {
    const short2 q = ((const short2*)raw)[src];
    x.r = (float)q.x * scale; x.i = (float)q.y * scale;
  } else if (FMT == 2) {
    const char2 q = ((const char2*)raw)[src];
    x.r = (float)q.x * scale; x.i = (float)q.y * scale;
  } else {
    x = ((const cf32*)raw)[src];
  }
  if (rot) {
    const cf32 w = rot[n];
    cf32 y; y.r = x.r * w.r - x.i * w.i; y.i = x.r * w.i + x.i * w.r;
    x = y;
  }
  out[((size_t)sf * nant + a) * sflen + n] = x;
}
void lsn_launch_file_unpack(const void* raw, uint32_t fmt, float scale, const cf32* rot, uint32_t sflen, uint32_t nant, cf32* out, uint32_t nsf, hipStream_t s)
{
  if (!nsf) return;
  const dim3 g((sflen + 255) / 256, nant, nsf);
  if (fmt == 1) LSN_LAUNCH(k_file_unpack<1>, g, dim3(256), 0, s, raw, rot, sflen, nant, scale, out);
  else if (fmt == 2) LSN_LAUNCH(k_file_unpack<2>, g, dim3(256), 0, s, raw, rot, sflen, nant, scale, out);
  else LSN_LAUNCH(k_file_unpack<0>, g, dim3(256), 0, s, raw, rot, sflen, nant, scale, out);
}

// ------------------------------------------------------------------------------------------------ descriptor upload (see lsn_dev.h)
__global__ __launch_bounds__(256) void k_upload_words(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t n)
{
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) dst[i] = src[i];
}
void lsn_launch_upload(void* dst_dev, const void* src_pinned, size_t bytes, hipStream_t s)
{
  const uint32_t n = (uint32_t)((bytes + 3) / 4);
  if (!n) return;
  const uint32_t blocks = std::min<uint32_t>((n + 255u) / 256u, 64u);
  LSN_LAUNCH(k_upload_words, dim3(blocks), dim3(256), 0, s, (const uint32_t*)src_pinned, (uint32_t*)dst_dev, n);
}

__global__ __launch_bounds__(256) void k_download(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n16, const uint32_t* __restrict__ src_w,
                                                  uint32_t* __restrict__ dst_w, uint32_t tail_first, uint32_t tail_n)
{
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
  if (blockIdx.x == 0 && threadIdx.x < tail_n) dst_w[tail_first + threadIdx.x] = src_w[tail_first + threadIdx.x];
  __threadfence_system();  // the stores to host memory are performed at system scope before the kernel (and the event behind it) completes
}
// several transfers in ONE launch (the mirrors of a stage-A chunk: five; the descriptors of a decode launch: three; its results: two): every launch of a
// copy kernel occupies a hardware queue slot of the streams that share it, and the pipeline made 55 of them per 1000 subframes (round 3)
__global__ __launch_bounds__(256) void k_copy_multi(LsnCopySegs sg, uint32_t to_host)
{
  for (uint32_t q = 0; q < sg.n; q++) {
    const uint32_t words = sg.words[q];
    const bool al = ((((uintptr_t)sg.src[q]) | ((uintptr_t)sg.dst[q])) & 15u) == 0;
    const uint32_t n16 = al ? words / 4 : 0;
    const uint4* a4 = (const uint4*)sg.src[q];
    uint4* b4 = (uint4*)sg.dst[q];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) b4[i] = a4[i];
    const uint32_t* a = (const uint32_t*)sg.src[q];
    uint32_t* b = (uint32_t*)sg.dst[q];
    for (uint32_t i = n16 * 4 + blockIdx.x * 256u + threadIdx.x; i < words; i += gridDim.x * 256u) b[i] = a[i];
  }
  if (to_host) __threadfence_system();  // the stores to host memory are performed at system scope before the kernel (and the event behind it) completes
}
void lsn_launch_copy_multi(const LsnCopySegs& sg, bool to_host, hipStream_t s)
{
  uint32_t total = 0;
  for (uint32_t q = 0; q < sg.n; q++) total += sg.words[q];
  if (!total) return;
  const uint32_t blocks = std::max<uint32_t>(1u, std::min<uint32_t>((total / 4 + 255u) / 256u, 128u));
  LSN_LAUNCH(k_copy_multi, dim3(blocks), dim3(256), 0, s, sg, to_host ? 1u : 0u);
}

void lsn_launch_download(void* dst_pinned, const void* src_dev, size_t bytes, hipStream_t s)
{
  if (!bytes) return;
  const bool al = (((uintptr_t)dst_pinned | (uintptr_t)src_dev) & 15u) == 0;
  const uint32_t words = (uint32_t)((bytes + 3) / 4), n16 = al ? words / 4 : 0, tail_first = n16 * 4, tail_n = words - tail_first;
  if (!al || tail_n > 256) {  // unaligned buffers: word copy
    const uint32_t blocks = std::min<uint32_t>((words + 255u) / 256u, 128u);
    LSN_LAUNCH(k_upload_words, dim3(blocks), dim3(256), 0, s, (const uint32_t*)src_dev, (uint32_t*)dst_pinned, words);
    return;
  }
  const uint32_t blocks = std::max<uint32_t>(1u, std::min<uint32_t>((n16 + 255u) / 256u, 128u));
  LSN_LAUNCH(k_download, dim3(blocks), dim3(256), 0, s, (const uint4*)src_dev, (uint4*)dst_pinned, n16, (const uint32_t*)src_dev, (uint32_t*)dst_pinned, tail_first,
                     tail_n);
}
