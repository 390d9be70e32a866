/* o_sync.c - ORACLE (test infrastructure only): PSS/SSS cell search on a block of downlink samples.
 * Restates what the reference obtains from rf_search_and_decode_mib(&rf, nant, &cell_detect_config, force_N_id_2, &cell,
 * &cfo) (/root/reference/src/src/LTESniffer_Core.cc:195-204, configuration :108-112) before it creates the Phy, i.e. the
 * physical cell id, the half-frame timing and the carrier offset [srsRAN lib/src/phy/sync/{pss,sss,sync}.c and
 * lib/src/phy/ue/ue_cell_search.c, not in tree], following TS 36.211 6.11 (FDD, normal cyclic prefix - the only prefix the
 * rest of the path handles):
 *   PSS  d_u(n), u = 25/29/34, 62 carriers around DC, last symbol of slots 0 and 10;
 *   SSS  two interleaved length-31 m-sequences (m0, m1 from N_id_1), scrambled by c0/c1 (N_id_2) and z1, one symbol earlier,
 *        different in subframes 0 and 5.
 * Detector: full-rate matched filter of the three time-domain PSS replicas over one 5 ms period, each output power divided by
 * the energy of its N-sample window (a burst of strong samples - AGC settling, junk in front of a recording - cannot win),
 * the ratios of `nof_periods` consecutive periods added (srsRAN averages the peak over nof_valid_pss_frames), argmax over
 * (root, lag); accepted when peak / mean of the winning root's metric >= threshold.  The channel on the 62 carriers is taken from the PSS
 * symbol, the SSS symbol is matched against all 168 x 2 sequences (|.|^2 of the coherent sum, so that a common phase turn
 * from the carrier offset does not matter) - an exhaustive search instead of srsRAN's m0/m1 partial correlations, which is
 * what one launch on a GPU does anyway.  CFO: coarse from the two halves of the PSS
 * symbol (srsran_pss_cfo_compute), fine from the phase turn between the PSS and SSS symbols.
 * Parity unpinned against srsRAN itself (library not vendored); the sequences are pinned by the 36.211 tables
 * (tests/test_sync_oracle.py).  Arithmetic contract as in lsn_oracle.h: one float rounding per operation, sequential
 * sums in index order, all cos/sin/atan2 on the "host" side. */
#include "lsn_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* 36.211 6.11.1.1 */
void o_pss_seq(uint32_t n_id_2, ocf_t* d /* 62 */)
{
  static const int root[3] = {25, 29, 34};
  const int u = root[n_id_2 % 3];
  for (int n = 0; n < 62; n++) {
    const int a = n < 31 ? n * (n + 1) : (n + 1) * (n + 2);
    const double ph = -M_PI * (double)u * (double)(a % 126) / 63.0; /* exp(-j pi u a / 63) has period 126 in a */
    d[n].r = (float)cos(ph);
    d[n].i = (float)sin(ph);
  }
}

/* 36.211 6.11.2.1: m0, m1 of N_id_1 (Table 6.11.2.1-1 is this formula) */
void o_sss_m0m1(uint32_t n_id_1, uint32_t* m0, uint32_t* m1)
{
  const uint32_t qp = n_id_1 / 30;
  const uint32_t q = (n_id_1 + qp * (qp + 1) / 2) / 30;
  const uint32_t mp = n_id_1 + q * (q + 1) / 2;
  *m0 = mp % 31;
  *m1 = (*m0 + mp / 31 + 1) % 31;
}

/* 36.211 6.11.2.1: d(0..61) in {-1,+1} for subframe 0 (sf5 = 0) or 5 */
void o_sss_seq(uint32_t n_id_1, uint32_t n_id_2, int sf5, int8_t* d /* 62 */)
{
  int xs[31], xc[31], xz[31];
  for (int i = 0; i < 5; i++) xs[i] = xc[i] = xz[i] = (i == 4);
  for (int i = 0; i < 26; i++) {
    xs[i + 5] = (xs[i + 2] + xs[i]) & 1;
    xc[i + 5] = (xc[i + 3] + xc[i]) & 1;
    xz[i + 5] = (xz[i + 4] + xz[i + 2] + xz[i + 1] + xz[i]) & 1;
  }
  uint32_t m0, m1;
  o_sss_m0m1(n_id_1, &m0, &m1);
  for (uint32_t n = 0; n < 31; n++) {
    const int s0 = 1 - 2 * xs[(n + m0) % 31], s1 = 1 - 2 * xs[(n + m1) % 31];
    const int c0 = 1 - 2 * xc[(n + n_id_2) % 31], c1 = 1 - 2 * xc[(n + n_id_2 + 3) % 31];
    const int z0 = 1 - 2 * xz[(n + (m0 % 8)) % 31], z1 = 1 - 2 * xz[(n + (m1 % 8)) % 31];
    if (!sf5) {
      d[2 * n] = (int8_t)(s0 * c0);
      d[2 * n + 1] = (int8_t)(s1 * c1 * z0);
    } else {
      d[2 * n] = (int8_t)(s1 * c0);
      d[2 * n + 1] = (int8_t)(s0 * c1 * z1);
    }
  }
}

/* FFT bin of synchronisation-signal carrier m (0..61): carriers -31..-1, +1..+31 around DC */
static inline int sync_bin(int m, int N) { return m < 31 ? N - 31 + m : m - 30; }

/* time-domain replica of PSS root n_id_2, N samples, unit energy: (1/sqrt(62 N)) sum_m d(m) exp(+j 2 pi k_m n / N) */
void o_pss_time(uint32_t n_id_2, uint32_t N, ocf_t* p)
{
  ocf_t d[62];
  o_pss_seq(n_id_2, d);
  const double sc = 1.0 / sqrt(62.0 * (double)N);
  for (uint32_t n = 0; n < N; n++) {
    double ar = 0, ai = 0;
    for (int m = 0; m < 62; m++) {
      const uint32_t idx = (uint32_t)(((uint64_t)sync_bin(m, (int)N) * n) % N);
      const double ph = 2.0 * M_PI * (double)idx / (double)N;
      const double c = cos(ph), s = sin(ph);
      ar += (double)d[m].r * c - (double)d[m].i * s;
      ai += (double)d[m].r * s + (double)d[m].i * c;
    }
    p[n].r = (float)(ar * sc);
    p[n].i = (float)(ai * sc);
  }
}

uint32_t o_sync_min_samples(uint32_t nof_prb, uint32_t nof_periods)
{
  const uint32_t N = (uint32_t)o_fft_size(nof_prb), W5 = 5 * 15 * N;
  return (nof_periods + 1) * W5 + N;
}

/* corr_out (optional): [3][5 * sflen] accumulated correlation powers (rows of roots not searched stay 0) */
int o_cell_search(const ocf_t* x, uint64_t nsamples, uint32_t nof_prb, const o_sync_cfg_t* cfg, o_sync_t* out, float* corr_out)
{
  memset(out, 0, sizeof *out);
  const uint32_t N = (uint32_t)o_fft_size(nof_prb);
  if (!N || (int)N < 0) return -1;
  const uint32_t sflen = 15 * N, W5 = 5 * sflen;
  const uint32_t P = cfg->nof_periods ? cfg->nof_periods : 1;
  if (cfg->force_n_id_2 > 2) return -1;
  if (nsamples < (uint64_t)(P + 1) * W5 + N) return -1;
  float* C = (float*)calloc((size_t)3 * W5, sizeof(float));
  ocf_t* p = (ocf_t*)malloc((size_t)N * sizeof(ocf_t));
  ocf_t* pbest = (ocf_t*)malloc((size_t)N * sizeof(ocf_t));
  float best = -1.0f;
  uint32_t bu = 0, bn = 0;
  for (uint32_t u = 0; u < 3; u++) {
    if (cfg->force_n_id_2 >= 0 && (uint32_t)cfg->force_n_id_2 != u) continue;
    o_pss_time(u, N, p);
    for (uint32_t n = 0; n < W5; n++) {
      float c = 0.0f;
      for (uint32_t q = 0; q < P; q++) {
        const ocf_t* xs = x + (size_t)q * W5 + n;
        float ar = 0.0f, ai = 0.0f, e = 0.0f;
        for (uint32_t k = 0; k < N; k++) { /* x * conj(p) and the energy of the window, one rounding per operation */
          const float t1 = xs[k].r * p[k].r, t2 = xs[k].i * p[k].i, t3 = xs[k].i * p[k].r, t4 = xs[k].r * p[k].i;
          ar = ar + (t1 + t2);
          ai = ai + (t3 - t4);
          e = e + (xs[k].r * xs[k].r + xs[k].i * xs[k].i);
        }
        if (e > 0.0f) c = c + (ar * ar + ai * ai) / e; /* in [0, 1]: the replica has unit energy */
      }
      C[(size_t)u * W5 + n] = c;
      if (c > best) { best = c; bu = u; bn = n; } /* first maximum in (root, lag) order */
    }
  }
  o_pss_time(bu, N, pbest);
  if (corr_out) memcpy(corr_out, C, (size_t)3 * W5 * sizeof(float));
  double mean = 0.0;
  for (uint32_t n = 0; n < W5; n++) mean += (double)C[(size_t)bu * W5 + n];
  mean /= (double)W5;
  free(C);
  out->n_id_2 = bu;
  out->pss_pos = bn;
  out->pss_peak = best;
  out->pss_p2avg = mean > 0.0 ? (float)((double)best / mean) : 0.0f;
  out->found = out->pss_p2avg >= cfg->threshold;

  /* The cyclic prefix of the cell (srsran_sync_detect_cp [srsRAN]: the search reports it, LTESniffer_Core.cc:195-204 hands it on): the SSS symbol sits
   * one symbol in front of the PSS symbol, N + 144 (x N / 2048) samples earlier with the normal CP, N + 512 with the extended one.  Both positions are
   * tried; the larger best SSS metric decides (the normal CP on a tie). */
  ocf_t* w = (ocf_t*)malloc((size_t)N * sizeof(ocf_t));
  for (uint32_t i = 0; i < N; i++) {
    const double ph = -2.0 * M_PI * (double)i / (double)N;
    w[i].r = (float)cos(ph);
    w[i].i = (float)sin(ph);
  }
  ocf_t d[62];
  o_pss_seq(bu, d);
  float win_m1 = -1.0f;
  for (uint32_t hyp = 0; hyp < 2; hyp++) {
    const uint32_t cpl = (hyp ? 512u : 144u) * N / 2048;
    /* Which of the P + 1 PSS occurrences at this lag to take the SSS and the carrier offset from: the strongest one (a
     * capture may start with dead samples) whose SSS symbol lies inside the buffer.  Per occurrence: the matched filter in
     * two halves (srsran_pss_cfo_compute: the phase turn between the halves is the carrier offset). */
    uint32_t jb = 0;
    float cjb = -1.0f;
    ocf_t yb[2] = {{0, 0}, {0, 0}};
    for (uint32_t j = bn >= N + cpl ? 0u : 1u; j <= P; j++) {
      const ocf_t* xj = x + (size_t)j * W5 + bn;
      ocf_t y[2];
      float eh[2];
      for (int h = 0; h < 2; h++) {
        float ar = 0.0f, ai = 0.0f, e = 0.0f;
        for (uint32_t k = h * (N / 2); k < (h + 1) * (N / 2); k++) {
          const float t1 = xj[k].r * pbest[k].r, t2 = xj[k].i * pbest[k].i, t3 = xj[k].i * pbest[k].r, t4 = xj[k].r * pbest[k].i;
          ar = ar + (t1 + t2);
          ai = ai + (t3 - t4);
          e = e + (xj[k].r * xj[k].r + xj[k].i * xj[k].i);
        }
        y[h].r = ar;
        y[h].i = ai;
        eh[h] = e;
      }
      const float sr = y[0].r + y[1].r, si = y[0].i + y[1].i, et = eh[0] + eh[1];
      const float cj = et > 0.0f ? (sr * sr + si * si) / et : 0.0f;
      if (cj > cjb) { cjb = cj; jb = j; yb[0] = y[0]; yb[1] = y[1]; }
    }
    const uint32_t q0 = bn + jb * W5;
    const int flipped = (int)(jb & 1u);
    const ocf_t* xp = x + q0;
    const ocf_t* xs = x + q0 - (N + cpl);
    /* 62 carriers of the PSS and SSS symbols: direct DFT, twiddles from a table */
    ocf_t z[62];
    for (int m = 0; m < 62; m++) {
      const uint32_t kb = (uint32_t)sync_bin(m, (int)N);
      ocf_t Y[2];
      for (int sy = 0; sy < 2; sy++) {
        const ocf_t* xx = sy ? xs : xp;
        float ar = 0.0f, ai = 0.0f;
        for (uint32_t n = 0; n < N; n++) {
          const ocf_t ww = w[(uint32_t)(((uint64_t)kb * n) % N)];
          const float t1 = xx[n].r * ww.r, t2 = xx[n].i * ww.i, t3 = xx[n].r * ww.i, t4 = xx[n].i * ww.r;
          ar = ar + (t1 - t2);
          ai = ai + (t3 + t4);
        }
        Y[sy].r = ar;
        Y[sy].i = ai;
      }
      /* H = Ypss conj(d); z = Ysss conj(H) */
      const float hr = Y[0].r * d[m].r + Y[0].i * d[m].i, hi = Y[0].i * d[m].r - Y[0].r * d[m].i;
      z[m].r = Y[1].r * hr + Y[1].i * hi;
      z[m].i = Y[1].i * hr - Y[1].r * hi;
    }
    float m1 = -1.0f, m2 = -1.0f, br = 0.0f, bi = 0.0f;
    uint32_t bh = 0;
    for (uint32_t h = 0; h < 336; h++) { /* h = 2 N_id_1 + (subframe 5) */
      int8_t sq[62];
      o_sss_seq(h >> 1, bu, (int)(h & 1), sq);
      float ar = 0.0f, ai = 0.0f;
      for (int m = 0; m < 62; m++) {
        const float sg = (float)sq[m];
        ar = ar + z[m].r * sg;
        ai = ai + z[m].i * sg;
      }
      const float mt = ar * ar + ai * ai;
      if (mt > m1) { m2 = m1; m1 = mt; bh = h; br = ar; bi = ai; }
      else if (mt > m2) m2 = mt;
    }
    if (!(m1 > win_m1)) continue; /* the other hypothesis stays */
    win_m1 = m1;
    out->cp = hyp;
    {
      const float cr = yb[0].r * yb[1].r + yb[0].i * yb[1].i, ci = yb[0].r * yb[1].i - yb[0].i * yb[1].r; /* conj(y0) y1 */
      out->cfo_coarse_hz = atan2f(ci, cr) / (float)M_PI * 15000.0f; /* +-15 kHz, disturbed by the other carriers of a loaded cell */
    }
    out->n_id_1 = bh >> 1;
    out->cell_id = 3 * out->n_id_1 + bu;
    out->sss_metric = m1;
    out->sss_second = m2;
    /* fine carrier offset: the SSS symbol, equalised with the channel seen by the PSS symbol one symbol (N + cp samples)
     * later, is turned by -2 pi f (N + cp) / fs; unambiguous within +-7 kHz */
    out->cfo_hz = -atan2f(bi, br) / (2.0f * (float)M_PI) * (15000.0f * (float)N / (float)(N + cpl));
    /* subframe timing: the PSS symbol's useful part starts 160 + 6 (N + 144) [x N/2048] samples into subframe 0 / 5 (normal CP: the seventh symbol),
     * 5 (N + 512) + 512 with the extended CP (the sixth symbol) */
    const uint32_t pss_off = hyp ? 5 * (N + cpl) + cpl : 160 * N / 2048 + 6 * (N + cpl);
    uint32_t sf_of_q0 = (bh & 1) ? 5u : 0u;
    uint32_t sf_of_bn = flipped ? (sf_of_q0 + 5) % 10 : sf_of_q0;
    if (bn >= pss_off) {
      out->sf_start = bn - pss_off;
      out->sf_idx = sf_of_bn;
    } else {
      out->sf_start = bn + W5 - pss_off;
      out->sf_idx = (sf_of_bn + 5) % 10;
    }
  }
  free(w);
  free(p);
  free(pbest);
  return out->found ? 1 : 0;
}
