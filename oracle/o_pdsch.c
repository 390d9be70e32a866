/* o_pdsch.c - ORACLE (test infrastructure only): PDSCH receive chain for one grant.
 * Restates srsran_ue_dl_decode_pdsch [srsRAN, not in tree] as called at
 * /root/reference/src/src/DL_Sniffer_PDSCH.cc:997,1110,1207 with the configuration of
 * /root/reference/src/src/SubframeWorker.cc:362-371 (12 turbo iterations, MMSE MIMO decoder, CSI weighting,
 * power scaling), following TS 36.211 6.3 (layer mapping / precoding: single port, SFBC, large-delay CDD,
 * closed-loop spatial multiplexing), 7.1 (QPSK..256QAM), 6.3.1 (scrambling) and TS 36.212 5.1.2-5.1.4
 * (segmentation, turbo code with QPP interleaver, rate matching with RV).
 * Arithmetic contract: float32 equaliser with one rounding per operation, LLRs quantised to int (|x|<=511),
 * integer max-log-MAP on P=o_turbo_nwin(K) parallel windows with next-iteration boundary initialisation. */
#include "lsn_oracle.h"
#include "../spec/lte_tables.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SQRT1_2F 0.70710678118654752440f
#define SQRT2F 1.41421356237309504880f
#define LLR_Q 180.0f
#define LLR_CLIP 511
#define EXT_CLIP 2047
#define NEG_METRIC (-12000)

static inline ocf_t cmulconj(ocf_t a, ocf_t b)
{
  ocf_t c;
  c.r = a.r * b.r + a.i * b.i;
  c.i = a.i * b.r - a.r * b.i;
  return c;
}
static inline float cabs2(ocf_t a) { return a.r * a.r + a.i * a.i; }

int o_qpp_find(int K, int* f1, int* f2)
{
  for (int i = 0; i < LSN_QPP_NSIZES; i++)
    if (lsn_qpp_table[i][0] == K) {
      *f1 = lsn_qpp_table[i][1];
      *f2 = lsn_qpp_table[i][2];
      return i;
    }
  return -1;
}

/* 36.212 5.1.2 */
int o_cbsegm(o_cbsegm_t* s, int tbs)
{
  int B = tbs + 24, Bp;
  memset(s, 0, sizeof(*s));
  s->tbs = tbs;
  if (tbs <= 0) return -1;
  if (B <= 6144) {
    s->C = 1;
    Bp = B;
  } else {
    s->C = (B + 6119) / 6120;
    Bp = B + 24 * s->C;
  }
  int idx = -1;
  for (int i = 0; i < LSN_QPP_NSIZES; i++)
    if (s->C * (int)lsn_qpp_table[i][0] >= Bp) {
      idx = i;
      break;
    }
  if (idx < 0) return -1;
  s->Kp = lsn_qpp_table[idx][0];
  if (s->C == 1) {
    s->Cp = 1;
    s->Km = 0;
    s->Cm = 0;
  } else {
    if (idx == 0) return -1;
    s->Km = lsn_qpp_table[idx - 1][0];
    int dK = s->Kp - s->Km;
    s->Cm = (s->C * s->Kp - Bp) / dK;
    s->Cp = s->C - s->Cm;
  }
  s->F = s->Cp * s->Kp + s->Cm * s->Km - Bp;
  return 0;
}

/* ---- soft demodulation (approximate max-log LLR, positive = bit 1) ---- */
static void demod_llr(int Qm, float I, float Q, float* L)
{
  float aI = fabsf(I), aQ = fabsf(Q);
  L[0] = -I;
  L[1] = -Q;
  if (Qm == 4) {
    const float a = 0.31622776601683794f;
    L[2] = aI - 2.0f * a;
    L[3] = aQ - 2.0f * a;
  } else if (Qm == 6) {
    const float a = 0.15430334996209191f;
    float tI = aI - 4.0f * a, tQ = aQ - 4.0f * a;
    L[2] = tI;
    L[3] = tQ;
    L[4] = fabsf(tI) - 2.0f * a;
    L[5] = fabsf(tQ) - 2.0f * a;
  } else if (Qm == 8) {
    const float a = 0.07669649888473704f;
    float tI = aI - 8.0f * a, tQ = aQ - 8.0f * a;
    float uI = fabsf(tI) - 4.0f * a, uQ = fabsf(tQ) - 4.0f * a;
    L[2] = tI;
    L[3] = tQ;
    L[4] = uI;
    L[5] = uQ;
    L[6] = fabsf(uI) - 2.0f * a;
    L[7] = fabsf(uQ) - 2.0f * a;
  }
}

/* soft-bit clip of the demodulator: 511 (10-bit soft values, the production contract); the second-opinion chain (o_second.c) lifts it */
static int g_llr_clip = LLR_CLIP;
void o_pdsch_set_llr_clip(int clip) { g_llr_clip = clip > 0 && clip < 32768 ? clip : LLR_CLIP; }
static inline int16_t quant_llr(float v)
{
  float r = rintf(v);
  if (r > (float)g_llr_clip) r = (float)g_llr_clip;
  if (r < (float)-g_llr_clip) r = (float)-g_llr_clip;
  return (int16_t)r;
}

static void emit(int Qm, ocf_t x, float w, float inv_amp, const uint8_t* c, int16_t* out)
{
  float L[8];
  float wq = w * LLR_Q;
  demod_llr(Qm, x.r * inv_amp, x.i * inv_amp, L);
  for (int b = 0; b < Qm; b++) {
    int16_t q = quant_llr(L[b] * wq);
    out[b] = c[b] ? (int16_t)-q : q;
  }
}

int o_pdsch_demod(const o_cell_t* cell, uint32_t nof_rx, uint32_t sf_idx, uint32_t cfi, uint16_t rnti,
                  const o_pdsch_grant_t* g, const ocf_t* grid, const ocf_t* ce, float noise, float chan_ref,
                  float rho_a_db, int16_t* llr_cw0, int16_t* llr_cw1)
{
  uint32_t nprb = cell->nof_prb, nre = 12 * nprb, nofre = g->nof_re;
  if (nofre == 0) return -1;
  uint16_t* rl = (uint16_t*)malloc(sizeof(uint16_t) * nofre);
  uint16_t* rk = (uint16_t*)malloc(sizeof(uint16_t) * nofre);
  uint32_t n = 0, l0 = cfi + (nprb <= 10 ? 1u : 0u);
  for (uint32_t l = l0; l < (uint32_t)o_nsym(cell); l++)
    for (uint32_t prb = 0; prb < nprb; prb++)
      if (g->prb_idx[l / (uint32_t)o_nslot(cell)][prb])
        for (uint32_t k = 12 * prb; k < 12 * prb + 12; k++)
          if (o_pdsch_re_ok(cell, sf_idx, l, k) && n < nofre) {
            rl[n] = (uint16_t)l;
            rk[n] = (uint16_t)k;
            n++;
          }
  /* modulation per codeword */
  int qm_cw[2] = {0, 0};
  for (int i = 0; i < 2; i++)
    if (g->tb[i].enabled) qm_cw[g->tb[i].cw_idx & 1] = g->tb[i].mod;
  int16_t* out[2] = {llr_cw0, llr_cw1};
  /* scrambling sequences (36.211 6.3.1) */
  uint8_t* c[2] = {NULL, NULL};
  for (int q = 0; q < 2; q++)
    if (qm_cw[q]) {
      c[q] = (uint8_t*)malloc((size_t)nofre * (size_t)qm_cw[q]);
      o_gold(((uint32_t)rnti << 14) | ((uint32_t)q << 13) | (sf_idx << 9) | cell->id, c[q], (int)nofre * qm_cw[q]);
    }
  /* power allocation 36.213 5.2, p_b = 1 (SubframeWorker.cc:372): rho_B/rho_A = 4/5 (1 port), 1 (2, 4 ports) */
  float rho_a = powf(10.0f, rho_a_db / 20.0f);
  float rho_b = (cell->nof_ports == 1) ? rho_a * sqrtf(0.8f) : rho_a;
  float inv_amp_a = 1.0f / rho_a, inv_amp_b = 1.0f / rho_b;
#define GRID(rx, i) grid[((size_t)(rx) * 14 + rl[i]) * nre + rk[i]]
#define CE(p, rx, i) ce[(((size_t)(p) * nof_rx + (rx)) * 14 + rl[i]) * nre + rk[i]]
#define INVAMP(i) (o_is_crs_sym01(cell, (int)rl[i]) ? inv_amp_b : inv_amp_a)   /* rho_B on the symbols that carry the CRS of ports 0, 1 (36.213 Table 5.2-2) */
  switch (g->tx_scheme) {
    case O_TX_PORT0: {
      int Qm = qm_cw[0];
      for (uint32_t i = 0; i < nofre; i++) {
        float nr = 0, ni = 0, den = 0;
        for (uint32_t rx = 0; rx < nof_rx; rx++) {
          ocf_t h = CE(0, rx, i);
          ocf_t t = cmulconj(GRID(rx, i), h);
          float hp = cabs2(h);
          if (rx == 0) { nr = t.r; ni = t.i; den = hp; }
          else { nr = nr + t.r; ni = ni + t.i; den = den + hp; }
        }
        float dn = den + noise;
        ocf_t x = {nr / dn, ni / dn};
        emit(Qm, x, den / chan_ref, INVAMP(i), c[0] + (size_t)i * (size_t)Qm, out[0] + (size_t)i * (size_t)Qm);
      }
      break;
    }
    case O_TX_DIVERSITY: {
      int Qm = qm_cw[0];
      /* four ports (SFBC-FSTD, 36.211 6.3.4.3): symbol pairs alternate between the port pairs (0, 2) and (1, 3); each pair sees half of the
       * ports chan_ref sums over, so its weight doubles */
      const int fstd = cell->nof_ports == 4;
      const float wscale = fstd ? 2.0f : 1.0f;
      for (uint32_t i = 0; i + 1 < nofre; i += 2) {
        float x0r = 0, x0i = 0, x1r = 0, x1i = 0, hh = 0;
        const uint32_t pa = (fstd && (i & 2u)) ? 1u : 0u, pb = fstd ? pa + 2u : 1u;
        for (uint32_t rx = 0; rx < nof_rx; rx++) {
          ocf_t r0 = GRID(rx, i), r1 = GRID(rx, i + 1);
          ocf_t h00 = CE(pa, rx, i), h01 = CE(pa, rx, i + 1), h10 = CE(pb, rx, i), h11 = CE(pb, rx, i + 1);
          float hp = cabs2(h00) + cabs2(h11);
          ocf_t a = cmulconj(r0, h00), b = cmulconj(h11, r1), cc = cmulconj(h10, r0), d = cmulconj(r1, h01);
          float t0r = a.r + b.r, t0i = a.i + b.i, t1r = d.r - cc.r, t1i = d.i - cc.i;
          if (rx == 0) { x0r = t0r; x0i = t0i; x1r = t1r; x1i = t1i; hh = hp; }
          else { x0r = x0r + t0r; x0i = x0i + t0i; x1r = x1r + t1r; x1i = x1i + t1i; hh = hh + hp; }
        }
        ocf_t x0 = {x0r / hh * SQRT2F, x0i / hh * SQRT2F}, x1 = {x1r / hh * SQRT2F, x1i / hh * SQRT2F};
        float w = hh * wscale / chan_ref;
        emit(Qm, x0, w, INVAMP(i), c[0] + (size_t)i * (size_t)Qm, out[0] + (size_t)i * (size_t)Qm);
        emit(Qm, x1, w, INVAMP(i + 1), c[0] + (size_t)(i + 1) * (size_t)Qm, out[0] + (size_t)(i + 1) * (size_t)Qm);
      }
      break;
    }
    case O_TX_SPATIALMUX:
    case O_TX_CDD: {
      if (cell->nof_ports != 2) { n = 0; break; } /* one port: no such transmission; four ports: not decodable (the reference's srsRAN precodes four ports for transmit diversity only) */
      if (g->nof_layers == 1) { /* closed-loop rank 1: w = [1, q]/sqrt2, q = 1,-1,j,-j */
        int Qm = qm_cw[0];
        for (uint32_t i = 0; i < nofre; i++) {
          float nr = 0, ni = 0, den = 0;
          for (uint32_t rx = 0; rx < nof_rx; rx++) {
            ocf_t h0 = CE(0, rx, i), h1 = CE(1, rx, i), qh;
            switch (g->pmi) {
              case 0: qh = h1; break;
              case 1: qh.r = -h1.r; qh.i = -h1.i; break;
              case 2: qh.r = -h1.i; qh.i = h1.r; break;
              default: qh.r = h1.i; qh.i = -h1.r; break;
            }
            ocf_t he = {(h0.r + qh.r) * SQRT1_2F, (h0.i + qh.i) * SQRT1_2F};
            ocf_t t = cmulconj(GRID(rx, i), he);
            float hp = cabs2(he);
            if (rx == 0) { nr = t.r; ni = t.i; den = hp; }
            else { nr = nr + t.r; ni = ni + t.i; den = den + hp; }
          }
          float dn = den + noise;
          ocf_t x = {nr / dn, ni / dn};
          emit(Qm, x, den * 2.0f / chan_ref, INVAMP(i), c[0] + (size_t)i * (size_t)Qm, out[0] + (size_t)i * (size_t)Qm);
        }
      } else { /* two layers, MMSE: CDD (W(i)D(i)U) or closed-loop codebook index pmi+1 */
        if (nof_rx < 2) { n = 0; break; }
        for (uint32_t i = 0; i < nofre; i++) {
          float a = 0, d = 0, br = 0, bi = 0, z0r = 0, z0i = 0, z1r = 0, z1i = 0;
          for (uint32_t rx = 0; rx < nof_rx; rx++) {
            ocf_t h0 = CE(0, rx, i), h1 = CE(1, rx, i), qh, y = GRID(rx, i);
            if (g->tx_scheme == O_TX_CDD) {
              if (i & 1) { qh.r = -h1.r; qh.i = -h1.i; } else qh = h1;
            } else if (g->pmi == 0) {
              qh = h1;
            } else {
              qh.r = -h1.i; qh.i = h1.r;
            }
            ocf_t e0 = {(h0.r + qh.r) * 0.5f, (h0.i + qh.i) * 0.5f};
            ocf_t e1 = {(h0.r - qh.r) * 0.5f, (h0.i - qh.i) * 0.5f};
            ocf_t b = cmulconj(e1, e0); /* conj(e0) e1 */
            ocf_t t0 = cmulconj(y, e0), t1 = cmulconj(y, e1);
            float p0 = cabs2(e0), p1 = cabs2(e1);
            if (rx == 0) { a = p0; d = p1; br = b.r; bi = b.i; z0r = t0.r; z0i = t0.i; z1r = t1.r; z1i = t1.i; }
            else { a = a + p0; d = d + p1; br = br + b.r; bi = bi + b.i; z0r = z0r + t0.r; z0i = z0i + t0.i; z1r = z1r + t1.r; z1i = z1i + t1.i; }
          }
          a = a + noise;
          d = d + noise;
          float det = a * d - (br * br + bi * bi);
          /* x0 = (d z0 - b z1)/det ; x1 = (a z1 - conj(b) z0)/det */
          ocf_t x0, x1;
          x0.r = (d * z0r - (br * z1r - bi * z1i)) / det;
          x0.i = (d * z0i - (br * z1i + bi * z1r)) / det;
          x1.r = (a * z1r - (br * z0r + bi * z0i)) / det;
          x1.i = (a * z1i - (br * z0i - bi * z0r)) / det;
          float w0 = det / d * 4.0f / chan_ref, w1 = det / a * 4.0f / chan_ref;
          float ia = INVAMP(i);
          if (qm_cw[0]) emit(qm_cw[0], x0, w0, ia, c[0] + (size_t)i * (size_t)qm_cw[0], out[0] + (size_t)i * (size_t)qm_cw[0]);
          if (qm_cw[1]) emit(qm_cw[1], x1, w1, ia, c[1] + (size_t)i * (size_t)qm_cw[1], out[1] + (size_t)i * (size_t)qm_cw[1]);
        }
      }
      break;
    }
    default: n = 0; break;
  }
  free(rl);
  free(rk);
  free(c[0]);
  free(c[1]);
  return n ? 0 : -1;
}

/* ---- turbo rate de-matching, 36.212 5.1.4.1 ---- */
void o_rm_turbo_rx_cb(const int16_t* e, int E, int K, int F, int rv, int16_t* d3)
{
  int D = K + 4, R = (D + 31) / 32, KP = 32 * R, ND = KP - D, Ncb = 3 * KP;
  int* map = (int*)malloc(sizeof(int) * (size_t)Ncb);
  int32_t* acc = (int32_t*)calloc((size_t)(3 * D), sizeof(int32_t));
  for (int k = 0; k < KP; k++) {
    int col = k / R, row = k % R;
    int y = row * 32 + lsn_perm_tc[col];
    int i01 = y - ND;
    map[k] = (i01 >= 0 && i01 >= F) ? i01 : -1;                    /* d0 (filler = NULL) */
    map[KP + 2 * k] = (i01 >= 0 && i01 >= F) ? D + i01 : -1;       /* d1 */
    int pi = (lsn_perm_tc[col] + 32 * row + 1) % KP;
    map[KP + 2 * k + 1] = (pi - ND >= 0) ? 2 * D + pi - ND : -1;   /* d2 */
  }
  int k0 = R * (2 * ((Ncb + 8 * R - 1) / (8 * R)) * rv + 2);
  int k = 0, j = 0;
  while (k < E) {
    int o = map[(k0 + j) % Ncb];
    if (o >= 0) {
      acc[o] += e[k];
      k++;
    }
    j++;
  }
  for (int i = 0; i < 3 * D; i++) {
    int32_t v = acc[i];
    if (v > LLR_CLIP) v = LLR_CLIP;
    if (v < -LLR_CLIP) v = -LLR_CLIP;
    d3[i] = (int16_t)v;
  }
  for (int i = 0; i < F; i++) { /* filler bits are known zeros */
    d3[i] = -LLR_CLIP;
    d3[D + i] = -LLR_CLIP;
  }
  free(map);
  free(acc);
}

/* ---- windowed max-log-MAP turbo decoder ---- */
int o_turbo_nwin(int K)
{
  /* largest divisor of K with windows of >= 32 steps that fills one 64-lane group, or two groups when a divisor in
   * 96..128 exists (design parameter of this restatement: srsRAN's SIMD decoders use 8..32 windows) */
  int p1 = 1, p2 = 1;
  for (int P = (K / 32 < 128 ? K / 32 : 128); P >= 1; P--)
    if (K % P == 0) { p2 = P; break; }
  for (int P = (K / 32 < 64 ? K / 32 : 64); P >= 1; P--)
    if (K % P == 0) { p1 = P; break; }
  return p2 >= 96 ? p2 : p1;
}

static uint8_t tr_next[8][2], tr_par[8][2];
static int tr_init = 0;
static void trellis_init(void)
{
  for (int S = 0; S < 8; S++)
    for (int u = 0; u < 2; u++) {
      int s1 = (S >> 2) & 1, s2 = (S >> 1) & 1, s3 = S & 1;
      int a = u ^ s2 ^ s3;
      tr_par[S][u] = (uint8_t)(a ^ s1 ^ s3);
      tr_next[S][u] = (uint8_t)((a << 2) | (s1 << 1) | s2);
    }
  tr_init = 1;
}

static inline int ext_scale(int x)
{
  int a = x < 0 ? -x : x;
  a = (a * 3) >> 2;
  if (a > EXT_CLIP) a = EXT_CLIP;
  return x < 0 ? -a : a;
}

#ifdef O_TURBO_STATS
/* instrumentation for tools/turbo_metric_ranges.py (a separate build of this file, never the test / bench library): the largest magnitudes the
 * recursions see, i.e. the word length a fixed-point variant of the kernel would need.  [0] |alpha| (normalised to state 0), [1] |beta| (normalised),
 * [2] |beta + gamma|, [3] |alpha + beta + gamma|, [4] |gamma| */
long long o_turbo_stat_max[5];
#define STAT(i, v) do { long long _a = (v) < 0 ? -(long long)(v) : (long long)(v); if (_a > o_turbo_stat_max[i]) o_turbo_stat_max[i] = _a; } while (0)
#else
#define STAT(i, v) do { } while (0)
#endif

/* one constituent decoder over all windows. idx[t] = position in natural order of trellis step t (identity for
 * DEC1, QPP for DEC2); par[t] in trellis order. a_init/b_init: [P][8] boundary metrics (in: previous iteration,
 * out: this iteration). beta_tail: [8] exact termination metrics for the last window. */
static void map_decode(int K, int P, const int16_t* sys, const int16_t* par, const int* idx, int16_t* ext,
                       int32_t (*a_init)[8], int32_t (*b_init)[8], const int32_t* beta_tail, int32_t* llr_out)
{
  int W = K / P;
  int32_t(*alpha)[8] = (int32_t(*)[8])malloc(sizeof(int32_t[8]) * (size_t)(W + 1));
  int32_t(*a_new)[8] = (int32_t(*)[8])malloc(sizeof(int32_t[8]) * (size_t)P);
  int32_t(*b_new)[8] = (int32_t(*)[8])malloc(sizeof(int32_t[8]) * (size_t)P);
  for (int p = 0; p < P; p++) {
    int t0 = p * W;
    if (p == 0) {
      for (int S = 0; S < 8; S++) alpha[0][S] = S == 0 ? 0 : NEG_METRIC;
    } else {
      memcpy(alpha[0], a_init[p], sizeof(int32_t[8]));
    }
    for (int t = 0; t < W; t++) {
      int pos = idx[t0 + t];
      int lsa = sys[pos] + ext[pos], lp = par[t0 + t];
      int32_t nx[8];
      for (int S = 0; S < 8; S++) nx[S] = -(1 << 30);
      for (int S = 0; S < 8; S++)
        for (int u = 0; u < 2; u++) {
          int32_t m = alpha[t][S] + (u ? lsa : 0) + (tr_par[S][u] ? lp : 0);
          int Sn = tr_next[S][u];
          if (m > nx[Sn]) nx[Sn] = m;
        }
      int32_t n0 = nx[0];
      for (int S = 0; S < 8; S++) { alpha[t + 1][S] = nx[S] - n0; STAT(0, alpha[t + 1][S]); }
      STAT(4, lsa + lp); STAT(4, lsa); STAT(4, lp);
    }
    memcpy(a_new[p], alpha[W], sizeof(int32_t[8]));
    int32_t beta[8], bn[8];
    if (p == P - 1)
      memcpy(beta, beta_tail, sizeof(beta));
    else
      memcpy(beta, b_init[p], sizeof(beta));
    for (int t = W - 1; t >= 0; t--) {
      int pos = idx[t0 + t];
      int lsa = sys[pos] + ext[pos], lp = par[t0 + t];
      int32_t m1 = -(1 << 30), m0 = -(1 << 30);
      for (int S = 0; S < 8; S++) {
        int32_t g0 = (tr_par[S][0] ? lp : 0), g1 = lsa + (tr_par[S][1] ? lp : 0);
        int32_t b0 = beta[tr_next[S][0]] + g0, b1 = beta[tr_next[S][1]] + g1;
        int32_t v0 = alpha[t][S] + b0, v1 = alpha[t][S] + b1;
        STAT(2, b0); STAT(2, b1); STAT(3, v0); STAT(3, v1);
        if (v0 > m0) m0 = v0;
        if (v1 > m1) m1 = v1;
        bn[S] = b0 > b1 ? b0 : b1;
      }
      int32_t L = m1 - m0;
      if (llr_out) llr_out[t0 + t] = L;
      ext[pos] = (int16_t)ext_scale(L - lsa);
      int32_t n0 = bn[0];
      for (int S = 0; S < 8; S++) { beta[S] = bn[S] - n0; STAT(1, beta[S]); }
    }
    memcpy(b_new[p], beta, sizeof(beta));
  }
  /* publish boundaries for the next iteration: window p starts from the end of window p-1 and ends at the start
   * of window p+1 */
  for (int p = 1; p < P; p++) memcpy(a_init[p], a_new[p - 1], sizeof(int32_t[8]));
  for (int p = 0; p < P - 1; p++) memcpy(b_init[p], b_new[p + 1], sizeof(int32_t[8]));
  free(alpha);
  free(a_new);
  free(b_new);
}

static void tail_beta(const int16_t* ts, const int16_t* tp, int32_t* beta)
{
  int32_t b[8], bn[8];
  for (int S = 0; S < 8; S++) b[S] = S == 0 ? 0 : NEG_METRIC;
  for (int t = 2; t >= 0; t--) {
    for (int S = 0; S < 8; S++) {
      int s1 = (S >> 2) & 1, s2 = (S >> 1) & 1, s3 = S & 1;
      int u = s2 ^ s3, z = s1 ^ s3, Sn = (s1 << 1) | s2;
      bn[S] = b[Sn] + (u ? ts[t] : 0) + (z ? tp[t] : 0);
    }
    memcpy(b, bn, sizeof(b));
  }
  int32_t n0 = b[0];
  for (int S = 0; S < 8; S++) beta[S] = b[S] - n0;
}

int o_turbo_decode_cb(const int16_t* d3, int K, int max_iter, uint32_t crc_poly, uint8_t* bits, int* crc_ok)
{
  if (!tr_init) trellis_init();
  int D = K + 4, f1, f2;
  if (o_qpp_find(K, &f1, &f2) < 0) return -1;
  int P = o_turbo_nwin(K);
  const int16_t *d0 = d3, *d1 = d3 + D, *d2 = d3 + 2 * D;
  int* pi = (int*)malloc(sizeof(int) * (size_t)K);
  int* id = (int*)malloc(sizeof(int) * (size_t)K);
  int16_t* ext = (int16_t*)calloc((size_t)K, sizeof(int16_t));
  int32_t* llr2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)K);
  for (int i = 0; i < K; i++) {
    pi[i] = (int)(((long long)f1 * i + (long long)f2 * i * i) % K);
    id[i] = i;
  }
  /* tails: enc1 sys d0[K],d2[K],d1[K+1] par d1[K],d0[K+1],d2[K+1]; enc2 sys d0[K+2],d2[K+2],d1[K+3] par
   * d1[K+2],d0[K+3],d2[K+3] (36.212 5.1.3.2.2) */
  int16_t ts1[3] = {d0[K], d2[K], d1[K + 1]}, tp1[3] = {d1[K], d0[K + 1], d2[K + 1]};
  int16_t ts2[3] = {d0[K + 2], d2[K + 2], d1[K + 3]}, tp2[3] = {d1[K + 2], d0[K + 3], d2[K + 3]};
  int32_t bt1[8], bt2[8];
  tail_beta(ts1, tp1, bt1);
  tail_beta(ts2, tp2, bt2);
  int32_t(*a1)[8] = (int32_t(*)[8])calloc((size_t)P, sizeof(int32_t[8]));
  int32_t(*b1)[8] = (int32_t(*)[8])calloc((size_t)P, sizeof(int32_t[8]));
  int32_t(*a2)[8] = (int32_t(*)[8])calloc((size_t)P, sizeof(int32_t[8]));
  int32_t(*b2)[8] = (int32_t(*)[8])calloc((size_t)P, sizeof(int32_t[8]));
  int it = 0, ok = 0;
  while (it < max_iter && !ok) {
    map_decode(K, P, d0, d1, id, ext, a1, b1, bt1, NULL);
    map_decode(K, P, d0, d2, pi, ext, a2, b2, bt2, llr2);
    for (int i = 0; i < K; i++) bits[pi[i]] = llr2[i] > 0 ? 1 : 0;
    it++;
    ok = (o_crc_bits(crc_poly, 24, bits, K) == 0); /* data||parity divides g(x) */
  }
  if (crc_ok) *crc_ok = ok;
  free(pi); free(id); free(ext); free(llr2); free(a1); free(b1); free(a2); free(b2);
  return it;
}

/* one transport block: e[G] (one codeword) -> payload[tbs/8]; returns 1 if TB CRC24A ok (and not all-zero) */
int o_pdsch_decode_tb(const int16_t* e, int G, int tbs, int Qm, int NL, int rv, int max_iter, uint8_t* payload,
                      int* iters_total)
{
  return o_pdsch_decode_tb_harq(e, G, tbs, Qm, NL, rv, max_iter, payload, iters_total, NULL, 0, NULL);
}

/* ... with a HARQ soft buffer (the softbuffer_rx of srsran_ue_dl_decode_pdsch, HARQ.cc:71-190 / DL_Sniffer_PDSCH.cc:955-990): acc holds, per code
 * block, the de-rate-matched streams d0|d1|d2 of the transmissions so far (O_HARQ_CB_STRIDE int16 each).  combine = 0: a new transmission, the
 * buffer is overwritten (srsran_softbuffer_rx_reset_tbs); combine = 1: a retransmission, the streams of this redundancy version are added to the
 * buffer, the sum clipped to the decoder's 10-bit soft values, and decoder and buffer both see the combined streams.  Design parameter of this
 * restatement (srsRAN keeps int16 circular buffers; parity unpinned): the buffer holds clipped 10-bit values. */
/* keep (round 5, advisor finding): srsran_softbuffer_rx_t also holds, per code block, cb_crc and the decoded bits of a block whose CRC passed
 * (softbuffer.h / sch.c decode_tb_cb [srsRAN]): such a block is neither combined nor decoded again by a retransmission - its bits come from the buffer.
 * keep[r * O_HARQ_KEEP_STRIDE] = 1 when block r passed in an earlier transmission of this transport block, followed by its K decoded bits; a new
 * transmission clears the flags (srsran_softbuffer_rx_reset_tbs). */
int o_pdsch_decode_tb_harq(const int16_t* e, int G, int tbs, int Qm, int NL, int rv, int max_iter, uint8_t* payload,
                           int* iters_total, int16_t* acc, int combine, uint8_t* keep)
{
  o_cbsegm_t s;
  if (o_cbsegm(&s, tbs) || Qm <= 0 || G <= 0) return 0;
  int Gp = G / (NL * Qm), gamma = Gp % s.C;
  uint8_t* tbbits = (uint8_t*)malloc((size_t)(tbs + 24 + 64));
  int16_t* d3 = (int16_t*)malloc(sizeof(int16_t) * 3 * (6144 + 4));
  uint8_t* cb = (uint8_t*)malloc(6144);
  int rp = 0, wp = 0, all_ok = 1, its = 0;
  for (int r = 0; r < s.C; r++) {
    int K = r < s.Cm ? s.Km : s.Kp;
    int F = r == 0 ? s.F : 0;
    int E = (r <= s.C - gamma - 1) ? NL * Qm * (Gp / s.C) : NL * Qm * ((Gp + s.C - 1) / s.C);
    int ok = 0;
    if (rp + E > G) E = G - rp;
    uint8_t* kp = keep ? keep + (size_t)r * O_HARQ_KEEP_STRIDE : NULL;
    if (kp && !combine) kp[0] = 0;
    if (kp && combine && kp[0]) { /* passed before: taken from the buffer, not decoded again */
      int take_k = K - F - (s.C > 1 ? 24 : 0);
      memcpy(tbbits + wp, kp + 1 + F, (size_t)take_k);
      wp += take_k;
      rp += E;
      continue;
    }
    o_rm_turbo_rx_cb(e + rp, E, K, F, rv, d3);
    if (acc) {
      int16_t* a = acc + (size_t)r * O_HARQ_CB_STRIDE;
      if (combine)
        for (int i = 0; i < 3 * (K + 4); i++) {
          int v = (int)a[i] + (int)d3[i];
          d3[i] = (int16_t)(v > LLR_CLIP ? LLR_CLIP : (v < -LLR_CLIP ? -LLR_CLIP : v));
        }
      memcpy(a, d3, sizeof(int16_t) * 3 * (size_t)(K + 4));
    }
    int n = o_turbo_decode_cb(d3, K, max_iter, s.C > 1 ? O_CRC24B : O_CRC24A, cb, &ok);
    its += n > 0 ? n : 0;
    if (o_trace_enabled()) o_trace_cb(K, F, E, rv, d3, n, ok);
    if (!ok) all_ok = 0;
    if (kp && ok) { kp[0] = 1; memcpy(kp + 1, cb, (size_t)K); }
    int take_n = K - F - (s.C > 1 ? 24 : 0);
    memcpy(tbbits + wp, cb + F, (size_t)take_n);
    wp += take_n;
    rp += E;
  }
  if (iters_total) *iters_total += its;
  int crc_ok = 0;
  if (wp == tbs + 24) {
    uint32_t par = 0;
    for (int i = 0; i < 24; i++) par = (par << 1) | tbbits[tbs + i];
    uint32_t crc = o_crc_bits(O_CRC24A, 24, tbbits, tbs);
    crc_ok = (crc == par) && par != 0 && all_ok;
  }
  o_pack_bits(tbbits, payload, tbs);
  free(tbbits); free(d3); free(cb);
  return crc_ok;
}
