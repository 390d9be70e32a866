/* o_pusch.c - ORACLE (test infrastructure only): uplink receive chain for one PUSCH grant.
 * Restates what the reference obtains from srsran_enb_ul_fft (/root/reference/src/src/UL_Sniffer_PUSCH.cc:392),
 * srsran_chest_ul_estimate_pusch (:256) and srsran_pusch_decode (:262) [srsRAN, not in tree], following TS 36.211
 * 5.6 (SC-FDMA, 7.5 kHz shift), 5.5.1/5.5.2.1 (DMRS: Zadoff-Chu base sequences, cyclic shifts), 5.3 (scrambling,
 * modulation, transform precoding) and TS 36.212 5.2.2 (UL-SCH coding, channel interleaver).
 * Scope of this restatement: one receive antenna (the reference uses antenna 1 for the uplink,
 * UL_Sniffer_PUSCH.cc:391), group and sequence hopping of the reference signal as SIB2 configures them (ULSchedule.cc:143-146), type-1
 * frequency hopping, no SRS; allocations of 1 and 2 PRB (tabulated sequences, spec/lte_tables.h) and >= 3 PRB (Zadoff-Chu).
 * Arithmetic contract as in lsn_oracle.h: one float rounding per operation, fixed summation orders, all cos/sin on the
 * "host" side (tables). */
#include "lsn_oracle.h"
#include "../spec/lte_tables.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LLR_Q 180.0f
#define LLR_CLIP 511

static inline ocf_t cmul(ocf_t a, ocf_t b) { ocf_t c = {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; return c; }
static inline ocf_t cmulconj(ocf_t a, ocf_t b) { ocf_t c = {a.r * b.r + a.i * b.i, a.i * b.r - a.r * b.i}; return c; }

/* UL_Sniffer_PUSCH.cc:3-10: L_prb = 2^a 3^b 5^c */
int o_ul_valid_prb(uint32_t L)
{
  if (L == 0 || L > 110) return 0;
  while (L % 2 == 0) L /= 2;
  while (L % 3 == 0) L /= 3;
  while (L % 5 == 0) L /= 5;
  return L == 1;
}

/* half-subcarrier (7.5 kHz) shift table: exp(-j pi n / N), n = 0..N-1 (applied to each symbol after CP removal) */
void o_ul_shift_table(int N, ocf_t* t)
{
  for (int n = 0; n < N; n++) {
    double a = M_PI * (double)n / (double)N;
    t[n].r = (float)cos(a);
    t[n].i = (float)(-sin(a));
  }
}

/* SC-FDMA demodulation of one subframe, one antenna: CP strip, 7.5 kHz shift, N-point FFT, keep the 12*nprb carriers
 * around DC: grid[l][k], k < 6 nprb -> bin N - 6 nprb + k, else bin k - 6 nprb (no DC gap in the uplink). */
void o_ul_fft(const o_cell_t* cell, const ocf_t* in, ocf_t* grid)
{
  int N = o_fft_size(cell->nof_prb), nre = 12 * (int)cell->nof_prb;
  ocf_t* w = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)o_fft_twiddle_len(N));
  ocf_t* sh = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)N);
  ocf_t* buf = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)N);
  o_fft_twiddles(N, w);
  o_ul_shift_table(N, sh);
  int pos = 0;
  for (int l = 0; l < o_nsym(cell); l++) {
    pos += cell->cp ? 512 * N / 2048 : ((l % 7) == 0 ? 160 : 144) * N / 2048; /* extended CP: N / 4 on every symbol (36.211 Table 5.6-1) */
    for (int n = 0; n < N; n++) buf[n] = cmul(in[pos + n], sh[n]);
    o_fft(N, w, buf);
    for (int k = 0; k < nre; k++) grid[l * nre + k] = buf[(k < nre / 2) ? (N - nre / 2 + k) : (k - nre / 2)];
    pos += N;
  }
  free(w); free(sh); free(buf);
}

static int largest_prime_below(int n)
{
  for (int p = n - 1; p >= 2; p--) {
    int ok = 1;
    for (int d = 2; d * d <= p; d++)
      if (p % d == 0) { ok = 0; break; }
    if (ok) return p;
  }
  return 2;
}

/* base sequence r_{u,v}(n), n < M_sc = 12 L (36.211 5.5.1.1 / 5.5.1.2); v = 1 exists from 6 PRB on */
int o_dmrs_base(uint32_t u, uint32_t v, int M_sc, ocf_t* r)
{
  if (M_sc == 12 || M_sc == 24) { /* one / two PRB: the tabulated sequences exp(j phi(n) pi / 4), Tables 5.5.1.2-1 / -2 */
    for (int n = 0; n < M_sc; n++) {
      double a = M_PI * (double)(M_sc == 12 ? lsn_dmrs_phi12[u % 30][n] : lsn_dmrs_phi24[u % 30][n]) / 4.0;
      r[n].r = (float)cos(a);
      r[n].i = (float)sin(a);
    }
    return 0;
  }
  if (M_sc < 36) return -1;
  int Nzc = largest_prime_below(M_sc);
  double qb = (double)Nzc * (double)(u + 1) / 31.0;
  int q = (int)floor(qb + 0.5);
  if (v && M_sc >= 72) q += ((int)floor(2.0 * qb) & 1) ? -1 : 1; /* q = floor(qb + 1/2) + v (-1)^floor(2 qb) */
  for (int n = 0; n < M_sc; n++) {
    long long m = n % Nzc;
    long long ph = ((long long)q * m * (m + 1)) % (2ll * Nzc); /* exp(-j pi q m(m+1)/Nzc) has period 2 Nzc in the product */
    double a = M_PI * (double)ph / (double)Nzc;
    r[n].r = (float)cos(a);
    r[n].i = (float)(-sin(a));
  }
  return 0;
}

/* sequence group u and base sequence number v of slot ns (36.211 5.5.1.3, 5.5.1.4):
 *   u = (f_gh(ns) + f_ss^PUSCH) mod 30, f_gh = sum_i c(8 ns + i) 2^i mod 30 with c_init = floor(N_ID / 30) when group hopping is on, else 0;
 *   v = c(ns) with c_init = floor(N_ID / 30) 2^5 + f_ss^PUSCH when sequence hopping is on, group hopping off and M_sc >= 6 PRB, else 0 */
void o_dmrs_uv(const o_cell_t* cell, const o_ul_cfg_t* ul, uint32_t ns, int M_sc, uint32_t* u, uint32_t* v)
{
  const uint32_t fss = ((cell->id % 30u) + ul->delta_ss) % 30u;
  uint32_t fgh = 0;
  uint8_t c[8 * 20 + 8];
  if (ul->group_hopping_enabled) {
    o_gold(cell->id / 30u, c, 8 * 20);
    for (int i = 0; i < 8; i++) fgh += (uint32_t)c[8 * ns + (uint32_t)i] << i;
    fgh %= 30u;
  }
  *u = (fgh + fss) % 30u;
  *v = 0;
  if (!ul->group_hopping_enabled && ul->sequence_hopping_enabled && M_sc >= 72) {
    o_gold(((cell->id / 30u) << 5) + fss, c, 20);
    *v = c[ns];
  }
}

static const uint32_t n_dmrs1_tab[8] = {0, 2, 3, 4, 6, 8, 9, 10}; /* 36.211 Table 5.5.2.1.1-2 (cyclicShift of SIB2) */
static const uint32_t n_dmrs2_tab[8] = {0, 6, 3, 4, 2, 8, 10, 9}; /* Table 5.5.2.1.1-1 (cyclic shift field of DCI 0) */

/* n_cs of slot ns (36.211 5.5.2.1.1) */
uint32_t o_dmrs_ncs(const o_cell_t* cell, const o_ul_cfg_t* ul, uint32_t ns, uint32_t n_dmrs_dci)
{
  uint32_t fss = ((cell->id % 30u) + ul->delta_ss) % 30u;
  uint8_t c[8 * 7 * 20 + 8];
  o_gold(((cell->id / 30u) << 5) + fss, c, 8 * 7 * 20 + 8);
  uint32_t npn = 0;
  for (int i = 0; i < 8; i++) npn += (uint32_t)c[8u * (uint32_t)o_nslot(cell) * ns + (uint32_t)i] << i; /* n_PN(ns) = sum c(8 N_symb^UL ns + i) 2^i */
  return (n_dmrs1_tab[ul->cyclic_shift & 7] + n_dmrs2_tab[n_dmrs_dci & 7] + npn) % 12u;
}

/* r_PUSCH of slot ns: base sequence of the slot's (u, v) times the cyclic shift exp(j 2 pi n_cs n / 12) (36.211 5.5.2.1.1) */
int o_dmrs_pusch(const o_cell_t* cell, const o_ul_cfg_t* ul, uint32_t ns, uint32_t n_dmrs_dci, int M_sc, ocf_t* r)
{
  uint32_t u, v, ncs = o_dmrs_ncs(cell, ul, ns, n_dmrs_dci);
  o_dmrs_uv(cell, ul, ns, M_sc, &u, &v);
  if (o_dmrs_base(u, v, M_sc, r)) return -1;
  for (int n = 0; n < M_sc; n++) {
    double a = 2.0 * M_PI * (double)((ncs * (uint32_t)n) % 12u) / 12.0;
    ocf_t p = {(float)cos(a), (float)sin(a)};
    r[n] = cmul(r[n], p);
  }
  return 0;
}

static void demod_llr(int Qm, float I, float Q, float* L)
{
  float aI = fabsf(I), aQ = fabsf(Q);
  L[0] = -I;
  L[1] = -Q;
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

/* IDFT twiddles exp(+2 pi j k / M), k < M */
void o_idft_table(int M, ocf_t* w)
{
  for (int k = 0; k < M; k++) {
    double a = 2.0 * M_PI * (double)k / (double)M;
    w[k].r = (float)cos(a);
    w[k].i = (float)sin(a);
  }
}

/* Transform de-precoding: z[n] = sum_k x[k] exp(+2 pi j n k / M), M = 12 L = 2^a 3^b 5^c, as an autosort (Stockham) decimation-in-frequency
 * transform.  Radices: 4 while the remaining length divides by 4, then 2, 3, 5.  A stage of radix r on remaining length n (m = n / r, s = M / n):
 *   y[q + s (r p + t)] = W_n^(p t) * ( ((x_0 W_r^0) + x_1 W_r^t) + ... + x_(r-1) W_r^((r-1) t) ),  x_i = x[q + s (p + m i)],  0 <= p < m, 0 <= q < s, 0 <= t < r
 * with both twiddles read from the size-M table w (W_n^(p t) = w[(p t s) mod M], W_r^(i t) = w[(i t M / r) mod M]) and the sum taken in the order
 * written: one complex multiply per term, terms added left to right, one final complex multiply.  This operation order IS the definition the
 * HIP kernel (k_pusch_demod) reproduces bit for bit.  x is overwritten (ping-pong with tmp); the result is returned in x. */
void o_idft_mixed(int M, const ocf_t* w, ocf_t* x, ocf_t* tmp)
{
  int n = M, s = 1;
  ocf_t *in = x, *out = tmp;
  while (n > 1) {
    int r = (n % 4 == 0) ? 4 : (n % 2 == 0) ? 2 : (n % 3 == 0) ? 3 : 5;
    int m = n / r;
    for (int o = 0; o < M; o++) {
      int q = o % s, rest = o / s, t = rest % r, p = rest / r;
      ocf_t acc = cmul(in[q + s * p], w[0]);
      for (int i = 1; i < r; i++) {
        ocf_t term = cmul(in[q + s * (p + m * i)], w[(int)(((long long)i * t * (M / r)) % M)]);
        acc.r = acc.r + term.r;
        acc.i = acc.i + term.i;
      }
      out[o] = cmul(acc, w[(int)(((long long)p * t * s) % M)]);
    }
    ocf_t* sw = in; in = out; out = sw;
    n = m; s *= r;
  }
  if (in != x) memcpy(x, in, sizeof(ocf_t) * (size_t)M);
}

/* ---- UCI multiplexed into the PUSCH (36.212 5.2.2.6-5.2.2.8), as far as the data decoder needs it ----
 * What the reference configures (UL_Sniffer_PUSCH.cc:429-450, defaults MCSTracking.cc:1534-1538): HARQ-ACK bits of the
 * downlink grants seen together with the DCI 0 (nof_ack 0/1/2, SubframeWorker.cc:318-336), and for an aperiodic CSI
 * request a higher-layer sub-band CQI report of 4 + 2 N bits plus one RI bit; offsets I_ack = 10, I_cqi = 8, I_ri = 11
 * (beta = 20, 2.25, 15.875).  The control information itself is not decoded here; its resource elements are located so
 * that the UL-SCH bits are de-multiplexed correctly: RI cells are skipped, the first Q'_CQI cells of the row-major
 * fill are CQI, HARQ-ACK cells puncture the data and come back as erasures. */
int o_uci_cqi_bits(uint32_t nof_prb) /* srsran_cqi_size for SRSRAN_CQI_TYPE_SUBBAND_HL, one codeword, no PMI */
{
  int k = nof_prb <= 7 ? 0 : (nof_prb <= 26 ? 4 : (nof_prb <= 63 ? 6 : 8)); /* ul_cqi_hl_get_subband_size, dl_sniffer_pdsch.c:276-292 */
  if (!k) return 0;
  return 4 + 2 * (((int)nof_prb + k - 1) / k);
}

/* srsran_cqi_size [srsRAN lib/src/phy/phch/cqi.c] for the three report types the reference configures (DL_Sniffer_PDSCH.cc:150-163),
 * one codeword, no PMI, one-bit sub-band label: wideband 4 bits, UE-selected sub-band 4 + 1, higher-layer sub-band 4 + 2 N */
int o_uci_cqi_bits_type(uint32_t nof_prb, uint32_t cqi_type)
{
  if (cqi_type == 0) return 4;
  if (cqi_type == 1) return 5;
  return o_uci_cqi_bits(nof_prb);
}

static int uci_qprime(int O, int M, int nsymb, int beta8, int sumK, int cap)
{
  if (O <= 0) return 0;
  long long num = (long long)O * M * nsymb * beta8, den = 8ll * sumK;
  int q = (int)((num + den - 1) / den);
  return q < cap ? q : cap;
}

/* cls[r * C + c]: 0 data, 1 CQI, 2 RI, 3 HARQ-ACK; didx: UL-SCH symbol index of a data / ACK-punctured cell; C = N_symb^PUSCH = 12 columns (normal CP) or 10
 * (extended).  Built by running the standard's procedure literally. Returns the number of UL-SCH symbols (G / Qm), -1 on error. */
int o_uci_layout_cp(int M, int tbs, const o_uci_t* uci, int cp, uint8_t* cls, int* didx, int* q_ack, int* q_ri, int* q_cqi)
{
  o_cbsegm_t sg;
  int sumK = 0;
  const int C = cp ? 10 : 12;
  if (o_cbsegm(&sg, tbs)) return -1;
  sumK = sg.Cp * sg.Kp + sg.Cm * sg.Km;
  /* beta offsets of 36.213 Tables 8.6.3-1/-2/-3 in eighths (spec/lte_tables.h); a reserved index makes the grant undecodable */
  const int ia = uci && uci->i_ack_p1 ? (int)uci->i_ack_p1 - 1 : 10, ic = uci && uci->i_cqi_p1 ? (int)uci->i_cqi_p1 - 1 : 8,
            ir = uci && uci->i_ri_p1 ? (int)uci->i_ri_p1 - 1 : 11;
  if (ia > 15 || ic > 15 || ir > 15) return -1;
  int Oc = uci ? (int)uci->cqi_bits : 0;
  if (uci && ((uci->nof_ack && !lsn_beta_ack8[ia]) || (uci->ri_bits && !lsn_beta_ri8[ir]) || (Oc && !lsn_beta_cqi8[ic]))) return -1;
  const int Qa = uci ? uci_qprime((int)uci->nof_ack, M, C, lsn_beta_ack8[ia], sumK, 4 * M) : 0;
  const int Qr = uci ? uci_qprime((int)uci->ri_bits, M, C, lsn_beta_ri8[ir], sumK, 4 * M) : 0;
  const int Qc = Oc ? uci_qprime(Oc + (Oc > 11 ? 8 : 0), M, C, lsn_beta_cqi8[ic], sumK, C * M - Qr) : 0;
  /* 36.212 Tables 5.2.2.8-1 / -2: column sets of the rank indication and of the HARQ-ACK */
  static const int ri_cols_n[4] = {1, 4, 7, 10}, ack_cols_n[4] = {2, 3, 8, 9}, ri_cols_e[4] = {0, 3, 5, 8}, ack_cols_e[4] = {1, 2, 6, 7};
  const int* ri_cols = cp ? ri_cols_e : ri_cols_n;
  const int* ack_cols = cp ? ack_cols_e : ack_cols_n;
  memset(cls, 0, (size_t)(C * M));
  for (int i = 0, j = 0, r = M - 1; i < Qr;) { /* 5.2.2.8: rank indication first, bottom row upwards */
    cls[r * C + ri_cols[j]] = 2;
    i++; r = M - 1 - i / 4; j = (j + 3) % 4;
  }
  int k = 0; /* then CQI followed by data, row by row, skipping the RI cells */
  for (int r = 0; r < M; r++)
    for (int c = 0; c < C; c++) {
      if (cls[r * C + c] == 2) { didx[r * C + c] = -1; continue; }
      if (k < Qc) { cls[r * C + c] = 1; didx[r * C + c] = -1; }
      else didx[r * C + c] = k - Qc;
      k++;
    }
  for (int i = 0, j = 0, r = M - 1; i < Qa;) { /* HARQ-ACK overwrites */
    cls[r * C + ack_cols[j]] = 3;
    i++; r = M - 1 - i / 4; j = (j + 3) % 4;
  }
  if (q_ack) *q_ack = Qa;
  if (q_ri) *q_ri = Qr;
  if (q_cqi) *q_cqi = Qc;
  return C * M - Qr - Qc;
}
int o_uci_layout(int M, int tbs, const o_uci_t* uci, uint8_t* cls, int* didx, int* q_ack, int* q_ri, int* q_cqi)
{
  return o_uci_layout_cp(M, tbs, uci, 0, cls, didx, q_ack, q_ri, q_cqi);
}

/* One grant: DMRS channel estimate (LS on symbols 3 and 10, 3-tap frequency smoothing, one estimate per slot), 1-tap
 * MMSE equaliser, transform de-precoding (mixed-radix IDFT of size M_sc = 2^a 3^b 5^c, o_idft_mixed, scale
 * 1/sqrt(M_sc)), soft demodulation, descrambling and channel de-interleaving -> e[nof_re * Qm] int16 (UL-SCH order).
 * noise_out / sigpow_out: scalar estimates (mean |ls - smoothed|^2, mean |smoothed|^2). Returns 0 or -1. */
int o_pusch_demod_uci(const o_cell_t* cell, const o_ul_cfg_t* ul, uint32_t sf_idx, uint16_t rnti, const o_pusch_grant_t* g, uint32_t n_dmrs_dci,
                      const o_uci_t* uci, const ocf_t* grid, int16_t* e, float* noise_out, float* sigpow_out);
int o_pusch_demod(const o_cell_t* cell, const o_ul_cfg_t* ul, uint32_t sf_idx, uint16_t rnti, const o_pusch_grant_t* g, uint32_t n_dmrs_dci,
                  const ocf_t* grid, int16_t* e, float* noise_out, float* sigpow_out)
{
  return o_pusch_demod_uci(cell, ul, sf_idx, rnti, g, n_dmrs_dci, NULL, grid, e, noise_out, sigpow_out);
}
/* the same with multiplexed control information: e holds G = Qm * (12 M - Q'_RI - Q'_CQI) values, HARQ-ACK cells as zeros */
int o_pusch_demod_uci(const o_cell_t* cell, const o_ul_cfg_t* ul, uint32_t sf_idx, uint16_t rnti, const o_pusch_grant_t* g, uint32_t n_dmrs_dci,
                      const o_uci_t* uci, const ocf_t* grid, int16_t* e, float* noise_out, float* sigpow_out)
{
  int L = (int)g->L_prb, M = 12 * L, nre = 12 * (int)cell->nof_prb, Qm = g->mod;
  if (L < 1 || !o_ul_valid_prb(g->L_prb) || g->n_prb + g->L_prb > cell->nof_prb || Qm <= 0 || g->hop > 1) return -1;
  if (g->hop == 1 && g->n_prb2 + g->L_prb > cell->nof_prb) return -1;
  const int k0s[2] = {12 * (int)g->n_prb, 12 * (int)(g->hop == 1 ? g->n_prb2 : g->n_prb)}; /* first carrier per slot (type-1 hopping: two places) */
  ocf_t* base = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)M);
  ocf_t* ls = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)(2 * M));
  ocf_t* hs = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)(2 * M));
  float* tmp = (float*)malloc(sizeof(float) * (size_t)(2 * M));
  ocf_t* x = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)M);
  ocf_t* xt = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)M);
  ocf_t* w = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)M);
  const int C = cell->cp ? 10 : 12, nsl = o_nslot(cell), dm = nsl - 4; /* PUSCH symbols per subframe; the reference signal sits on symbol 3 (extended CP: 2) of each slot */
  uint8_t* c = (uint8_t*)malloc((size_t)(12 * M * Qm));
  uint8_t* cls = (uint8_t*)malloc((size_t)(12 * M));
  int* didx = (int*)malloc(sizeof(int) * (size_t)(12 * M));
  if (o_uci_layout_cp(M, g->tbs > 0 ? g->tbs : 16, uci, (int)cell->cp, cls, didx, NULL, NULL, NULL) < 0) { free(cls); free(didx); free(c); free(base); free(ls); free(hs); free(tmp); free(x); free(xt); free(w); return -1; }
  o_idft_table(M, w);
  o_gold(((uint32_t)rnti << 14) | (sf_idx << 9) | cell->id, c, C * M * Qm);
  /* cyclic-shift phasors exp(j 2 pi m / 12) */
  ocf_t ph12[12];
  for (int m = 0; m < 12; m++) {
    double a = 2.0 * M_PI * (double)m / 12.0;
    ph12[m].r = (float)cos(a);
    ph12[m].i = (float)sin(a);
  }
  for (int s = 0; s < 2; s++) {
    uint32_t ncs = o_dmrs_ncs(cell, ul, 2 * sf_idx + (uint32_t)s, n_dmrs_dci), u, v;
    o_dmrs_uv(cell, ul, 2 * sf_idx + (uint32_t)s, M, &u, &v);
    o_dmrs_base(u, v, M, base);
    const ocf_t* y = grid + (size_t)(dm + nsl * s) * (size_t)nre + (size_t)k0s[s];
    for (int n = 0; n < M; n++) {
      ocf_t r = cmul(base[n], ph12[(ncs * (uint32_t)n) % 12u]);
      ls[s * M + n] = cmulconj(y[n], r);
    }
    for (int n = 0; n < M; n++) {
      ocf_t a;
      if (n == 0) { a.r = (ls[s * M].r + ls[s * M + 1].r) / 2.0f; a.i = (ls[s * M].i + ls[s * M + 1].i) / 2.0f; }
      else if (n == M - 1) { a.r = (ls[s * M + n - 1].r + ls[s * M + n].r) / 2.0f; a.i = (ls[s * M + n - 1].i + ls[s * M + n].i) / 2.0f; }
      else { a.r = ((ls[s * M + n - 1].r + ls[s * M + n].r) + ls[s * M + n + 1].r) / 3.0f; a.i = ((ls[s * M + n - 1].i + ls[s * M + n].i) + ls[s * M + n + 1].i) / 3.0f; }
      hs[s * M + n] = a;
    }
  }
  for (int i = 0; i < 2 * M; i++) { float dr = hs[i].r - ls[i].r, di = hs[i].i - ls[i].i; tmp[i] = dr * dr + di * di; }
  float noise = o_reduce256(tmp, 2 * M) / (float)(2 * M);
  for (int i = 0; i < 2 * M; i++) tmp[i] = hs[i].r * hs[i].r + hs[i].i * hs[i].i;
  float sigpow = o_reduce256(tmp, 2 * M) / (float)(2 * M);
  if (noise_out) *noise_out = noise;
  if (sigpow_out) *sigpow_out = sigpow;
  const float scale = 1.0f / sqrtf((float)M);
  int col = 0;
  for (int l = 0; l < 2 * nsl; l++) {
    if (l == dm || l == nsl + dm) continue;
    const ocf_t* y = grid + (size_t)l * (size_t)nre + (size_t)k0s[l / nsl];
    const ocf_t* h = hs + (l / nsl) * M;
    for (int n = 0; n < M; n++) {
      ocf_t t = cmulconj(y[n], h[n]);
      float den = (h[n].r * h[n].r + h[n].i * h[n].i) + noise;
      x[n].r = t.r / den;
      x[n].i = t.i / den;
    }
    o_idft_mixed(M, w, x, xt); /* z = IDFT_M(x), mixed-radix with the operation order defined above */
    for (int r = 0; r < M; r++) {
      const float ar = x[r].r, ai = x[r].i;
      float Lb[8];
      demod_llr(Qm, ar * scale, ai * scale, Lb);
      for (int b = 0; b < Qm; b++) {
        float v = rintf(Lb[b] * LLR_Q);
        if (v > (float)LLR_CLIP) v = (float)LLR_CLIP;
        if (v < (float)-LLR_CLIP) v = (float)-LLR_CLIP;
        int16_t q = (int16_t)v;
        /* scrambling runs over the transmitted (column-major) order, the decoder wants the row-major UL-SCH order
         * (36.212 5.2.2.8: R_mux x 12 matrix written row by row, read column by column) */
        if (c[((size_t)col * (size_t)M + (size_t)r) * (size_t)Qm + (size_t)b]) q = (int16_t)-q;
        const int cell_cls = cls[r * C + col];
        if (cell_cls == 1 || cell_cls == 2 || didx[r * C + col] < 0) continue; /* CQI / RI (also under a HARQ-ACK symbol that overwrote a CQI cell): not part of the UL-SCH stream */
        e[(size_t)didx[r * C + col] * (size_t)Qm + (size_t)b] = cell_cls == 3 ? (int16_t)0 : q; /* HARQ-ACK punctures */
      }
    }
    col++;
  }
  free(base); free(ls); free(hs); free(tmp); free(x); free(xt); free(w); free(c); free(cls); free(didx);
  return 0;
}

/* srsran_chest_ul_estimate_pusch + srsran_pusch_decode for one grant: returns 1 when the transport block CRC passes */
int o_pusch_decode(const o_cell_t* cell, const o_ul_cfg_t* ul, uint32_t sf_idx, uint16_t rnti, const o_pusch_grant_t* g, uint32_t n_dmrs_dci,
                   const ocf_t* grid, int max_iter, uint8_t* payload, int* iters, float* snr_db)
{
  return o_pusch_decode_uci(cell, ul, sf_idx, rnti, g, n_dmrs_dci, NULL, grid, max_iter, payload, iters, snr_db);
}
int o_pusch_decode_uci(const o_cell_t* cell, const o_ul_cfg_t* ul, uint32_t sf_idx, uint16_t rnti, const o_pusch_grant_t* g, uint32_t n_dmrs_dci,
                       const o_uci_t* uci, const ocf_t* grid, int max_iter, uint8_t* payload, int* iters, float* snr_db)
{
  if (g->tbs <= 0) return 0;
  int M = 12 * (int)g->L_prb, G = (cell->cp ? 10 : 12) * M * g->mod;
  if (uci) {
    uint8_t* cls = (uint8_t*)malloc((size_t)(12 * M));
    int* didx = (int*)malloc(sizeof(int) * (size_t)(12 * M));
    int nsym = o_uci_layout_cp(M, g->tbs, uci, (int)cell->cp, cls, didx, NULL, NULL, NULL);
    free(cls); free(didx);
    if (nsym <= 0) return 0;
    G = nsym * g->mod;
  }
  int16_t* e = (int16_t*)malloc(sizeof(int16_t) * (size_t)(12 * M * g->mod > 0 ? 12 * M * g->mod : 1));
  float noise = 0, sig = 0;
  int ok = 0;
  if (o_pusch_demod_uci(cell, ul, sf_idx, rnti, g, n_dmrs_dci, uci, grid, e, &noise, &sig) == 0) {
    ok = o_pdsch_decode_tb(e, G, g->tbs, g->mod, 1, g->rv, max_iter, payload, iters);
    if (snr_db) *snr_db = 10.0f * log10f(sig / noise);
  }
  free(e);
  return ok;
}
