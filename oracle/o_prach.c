/* o_prach.c - ORACLE (test infrastructure only): PRACH preamble detection on one uplink subframe.
 * Restates what the reference obtains from srsran_prach_init / srsran_prach_set_cfg / srsran_prach_set_detect_factor(60) /
 * srsran_prach_tti_opportunity / srsran_prach_detect_offset (/root/reference/src/src/UL_Sniffer_PUSCH.cc:640-713, config
 * fields from SIB2: /root/reference/src/src/ULSchedule.cc:149-154) [srsRAN lib/src/phy/phch/prach.c, not in tree],
 * following TS 36.211 5.7: preamble format 0 (T_CP = 3168 Ts, T_SEQ = 24576 Ts - the only FDD format that fits the one
 * subframe work_prach hands over), Zadoff-Chu sequences of length 839, unrestricted cyclic-shift set (highSpeedFlag = 0),
 * K = 12, phi = 7.  Detector (the published srsRAN algorithm): DFT of the T_SEQ samples behind the CP, the 839 PRACH
 * bins times the conjugate root spectrum, 839-point inverse DFT, |.|^2, one window of N_CS lags per cyclic shift;
 * a preamble is reported when its window peak exceeds detect_factor x the mean of the correlation power.
 * The logical -> physical root table (36.211 Table 5.7.2-4) is an input (zc_roots, 838 entries); without it the root
 * numbers are taken as physical.  Preamble indices >= 64 of the last root are not reported.
 * Parity unpinned against srsRAN itself (library not vendored); arithmetic contract as in lsn_oracle.h: one float
 * rounding per operation, fixed summation orders (o_reduce256), all cos/sin on the "host" side in double. */
#include "lsn_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NZC 839

static inline ocf_t cmul(ocf_t a, ocf_t b) { ocf_t c = {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; return c; }
static inline ocf_t cmulconj(ocf_t a, ocf_t b) { ocf_t c = {a.r * b.r + a.i * b.i, a.i * b.r - a.r * b.i}; return c; }

/* 36.211 Table 5.7.2-2, unrestricted set */
static const uint16_t prach_ncs[16] = {0, 13, 15, 18, 22, 26, 32, 38, 46, 59, 76, 93, 119, 167, 279, 419};

uint32_t o_prach_ncs(uint32_t zero_corr_zone) { return zero_corr_zone < 16 ? prach_ncs[zero_corr_zone] : 0; }

/* 36.211 Table 5.7.1-2, preamble format 0 (configuration index 0..15): srsran_prach_tti_opportunity(p, tti, -1) */
int o_prach_tti_opportunity(uint32_t config_idx, uint32_t tti)
{
  static const uint16_t sf_mask[16] = {0x002, 0x010, 0x080, 0x002, 0x010, 0x080, 0x042, 0x084, 0x108, 0x092, 0x124, 0x248, 0x155, 0x2AA, 0x3FF, 0x200};
  if (config_idx > 15) return 0;
  uint32_t sfn = tti / 10, sf = tti % 10;
  int even_only = config_idx < 3 || config_idx == 15;
  if (even_only && (sfn & 1)) return 0;
  return (sf_mask[config_idx] >> sf) & 1;
}

/* first PRACH bin (1.25 kHz units relative to the carrier centre): phi + K (k0 + 1/2), k0 = n_PRB^RA N_sc^RB - N_RB^UL N_sc^RB / 2 */
int o_prach_first_bin(uint32_t nof_prb, uint32_t freq_offset) { return 7 + 12 * (12 * (int)freq_offset - 6 * (int)nof_prb) + 6; }

/* DFT_839 of the root sequence x_u(n) = exp(-j pi u n (n+1) / 839), in double, sequential sums, rounded to float */
void o_prach_root_spectrum(uint32_t u, ocf_t* D)
{
  static double vr[NZC], vi[NZC];
  static int init = 0;
  if (!init) {
    for (int m = 0; m < NZC; m++) { double a = 2.0 * M_PI * (double)m / (double)NZC; vr[m] = cos(a); vi[m] = sin(a); }
    init = 1;
  }
  double xr[NZC], xi[NZC];
  for (int n = 0; n < NZC; n++) {
    long long ph = ((long long)u * n % (2 * NZC)) * (n + 1) % (2 * NZC);
    double a = M_PI * (double)ph / (double)NZC;
    xr[n] = cos(a); xi[n] = -sin(a);
  }
  for (int k = 0; k < NZC; k++) {
    double sr = 0.0, si = 0.0;
    for (int n = 0; n < NZC; n++) {
      int m = (int)((long long)n * k % NZC); /* exp(-j 2 pi n k / 839) = conj(v[m]) */
      sr = sr + (xr[n] * vr[m] + xi[n] * vi[m]);
      si = si + (xi[n] * vr[m] - xr[n] * vi[m]);
    }
    D[k].r = (float)sr; D[k].i = (float)si;
  }
}

uint32_t o_prach_nof_roots(uint32_t zero_corr_zone)
{
  uint32_t ncs = o_prach_ncs(zero_corr_zone), nwin = ncs ? NZC / ncs : 1;
  return (64 + nwin - 1) / nwin;
}

/* samples: one uplink subframe (15 N cf32).  corr_out (optional): [nof_roots][839] correlation power.
 * Returns the number of detections written to out (at most cap), in (root, window) order like srsran_prach_detect_offset. */
int o_prach_detect(const o_cell_t* cell, const o_prach_cfg_t* cfg, const ocf_t* samples, o_prach_det_t* out, int cap, float* corr_out)
{
  if (cfg->config_idx > 15 || cfg->zero_corr_zone > 15 || cfg->hs_flag || cfg->root_seq_idx > 837) return -1;
  const int Nsym = o_fft_size(cell->nof_prb), N = 12 * Nsym, Ncp = 3168 * Nsym / 2048;
  const uint32_t ncs = o_prach_ncs(cfg->zero_corr_zone), nwin = ncs ? NZC / ncs : 1, win = ncs ? ncs : NZC;
  const uint32_t nroots = o_prach_nof_roots(cfg->zero_corr_zone);
  const float factor = cfg->detect_factor > 0.0f ? cfg->detect_factor : 60.0f;
  const int b0 = o_prach_first_bin(cell->nof_prb, cfg->freq_offset);
  ocf_t* W = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)N);
  ocf_t V[NZC], Y[NZC], D[NZC];
  for (int i = 0; i < N; i++) { double a = 2.0 * M_PI * (double)i / (double)N; W[i].r = (float)cos(a); W[i].i = (float)(-sin(a)); }
  for (int m = 0; m < NZC; m++) { double a = 2.0 * M_PI * (double)m / (double)NZC; V[m].r = (float)cos(a); V[m].i = (float)sin(a); }
  const ocf_t* x = samples + Ncp;
  float pr[256], pi[256];
  /* the 839 PRACH bins of the N-point DFT: 256 interleaved partial sums per bin, then the fixed tree */
  for (int j = 0; j < NZC; j++) {
    const int b = ((b0 + j) % N + N) % N;
    for (int t = 0; t < 256; t++) {
      float ar = 0.0f, ai = 0.0f;
      int idx = (int)((long long)b * t % N);
      const int step = (int)((long long)b * 256 % N);
      for (int n = t; n < N; n += 256) {
        ocf_t p = cmul(x[n], W[idx]);
        ar = ar + p.r; ai = ai + p.i;
        idx += step; if (idx >= N) idx -= N;
      }
      pr[t] = ar; pi[t] = ai;
    }
    Y[j].r = o_reduce256(pr, 256); Y[j].i = o_reduce256(pi, 256);
  }
  int ndet = 0;
  float* corr = (float*)malloc(sizeof(float) * NZC);
  for (uint32_t i = 0; i < nroots; i++) {
    uint32_t lr = (cfg->root_seq_idx + i) % 838u;
    uint32_t u = cfg->zc_roots ? cfg->zc_roots[lr] : lr + 1;
    o_prach_root_spectrum(u, D);
    for (int k = 0; k < NZC; k++) {
      for (int t = 0; t < 256; t++) {
        float ar = 0.0f, ai = 0.0f;
        for (int j = t; j < NZC; j += 256) {
          ocf_t p = cmul(cmulconj(Y[j], D[j]), V[(int)((long long)j * k % NZC)]);
          ar = ar + p.r; ai = ai + p.i;
        }
        pr[t] = ar; pi[t] = ai;
      }
      float cr = o_reduce256(pr, 256), ci = o_reduce256(pi, 256);
      corr[k] = cr * cr + ci * ci;
    }
    if (corr_out) memcpy(corr_out + (size_t)i * NZC, corr, sizeof(float) * NZC);
    const float ave = o_reduce256(corr, NZC) / (float)NZC;
    const float thr = factor * ave;
    for (uint32_t j = 0; j < nwin; j++) {
      const uint32_t start = (NZC - j * ncs) % NZC;
      float peak = 0.0f; uint32_t off = 0;
      for (uint32_t k = 0; k < win; k++)
        if (corr[start + k] > peak) { peak = corr[start + k]; off = k; }
      const uint32_t preamble = i * nwin + j;
      if (peak > thr && preamble < 64 && ndet < cap) {
        out[ndet].preamble = preamble; out[ndet].offset = off;
        out[ndet].offset_sec = (float)off * (float)(24576.0 / 30.72e6) / (float)NZC;
        out[ndet].p2avg = peak / ave;
        ndet++;
      }
    }
  }
  free(corr); free(W);
  return ndet;
}
