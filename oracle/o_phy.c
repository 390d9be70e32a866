/* o_phy.c - ORACLE (test infrastructure only): OFDM demodulation, CRS channel estimation, PCFICH,
 * PDCCH REG de-mapping + LLR extraction, per-PRB power.
 *
 * Restates what the reference obtains from srsran_ue_dl_decode_fft_estimate
 * (/root/reference/src/src/DCISearch.cc:562; configured at /root/reference/src/src/SubframeWorker.cc:362-400:
 * Gaussian smoothing coef (4,1), NOISE_ALG_REFS, ESTIMATOR_ALG_INTERPOLATE, CFO estimate on) following
 * TS 36.211 6.12 (OFDM), 6.10.1 (CRS), 6.7 (PCFICH), 6.8/6.2.4 (PDCCH, REGs), 6.9.3 (PHICH REGs), and
 * SubframePower::computePower (/root/reference/src/src/SubframePower.cc:18-42).
 * srsRAN itself is NOT in the tree: parity with its soft values is unpinned (see lsn_oracle.h). */
#include "lsn_oracle.h"
#include "../spec/lte_tables.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SQRT1_2F 0.70710678118654752440f
#define SQRT2F 1.41421356237309504880f

int o_fft_size(uint32_t nof_prb)
{
  switch (nof_prb) {
    case 6: return 128;
    case 15: return 256;
    case 25: return 512;
    case 50: return 1024;
    case 100: return 2048;
    case 75: return 1536;
    default: return -1;
  }
}

/* table length for o_fft_twiddles: N/2 for the power-of-two sizes; 1536 = 3 x 512: the 256 twiddles of the 512-point
 * transform followed by the full circle exp(-2 pi i k / 1536), k < 1536, of the radix-3 combination */
int o_fft_twiddle_len(int N) { return N == 1536 ? 256 + 1536 : N / 2; }

void o_fft_twiddles(int N, ocf_t* w)
{
  const int M = N == 1536 ? 512 : N;
  for (int k = 0; k < M / 2; k++) {
    double a = 2.0 * M_PI * (double)k / (double)M;
    w[k].r = (float)cos(a);
    w[k].i = (float)(-sin(a));
  }
  if (N == 1536)
    for (int k = 0; k < 1536; k++) {
      double a = 2.0 * M_PI * (double)k / 1536.0;
      w[256 + k].r = (float)cos(a);
      w[256 + k].i = (float)(-sin(a));
    }
}

static inline ocf_t cmul(ocf_t a, ocf_t b)
{
  ocf_t c;
  c.r = a.r * b.r - a.i * b.i;
  c.i = a.r * b.i + a.i * b.r;
  return c;
}
static inline ocf_t cmulconj(ocf_t a, ocf_t b) /* a * conj(b) */
{
  ocf_t c;
  c.r = a.r * b.r + a.i * b.i;
  c.i = a.i * b.r - a.r * b.i;
  return c;
}

/* Radix-2 decimation-in-time, bit-reversed load, twiddle table W[k]=exp(-2 pi i k/N).  Every butterfly is
 * v = b*w (4 mul, 1 sub, 1 add), a' = a+v, b' = a-v, including the trivial twiddles. */
void o_fft(int N, const ocf_t* w, ocf_t* a)
{
  if (N == 1536) {
    /* 15 MHz: decimation in time by three, x_r[m] = x[3 m + r] -> F_r = FFT512(x_r), then
     * X[k] = (F_0[k % 512] + F_1[k % 512] T[k]) + F_2[k % 512] T[2 k mod 1536], T[k] = exp(-2 pi i k / 1536) */
    static _Thread_local ocf_t sub[3][512];
    const ocf_t* T = w + 256;
    for (int r = 0; r < 3; r++) {
      for (int m = 0; m < 512; m++) sub[r][m] = a[3 * m + r];
      o_fft(512, w, sub[r]);
    }
    for (int k = 0; k < 1536; k++) {
      const int kq = k % 512;
      const ocf_t t1 = cmul(sub[1][kq], T[k]), t2 = cmul(sub[2][kq], T[(2 * k) % 1536]);
      const float sr = sub[0][kq].r + t1.r, si = sub[0][kq].i + t1.i;
      a[k].r = sr + t2.r;
      a[k].i = si + t2.i;
    }
    return;
  }
  int lg = 0;
  while ((1 << lg) < N) lg++;
  for (int i = 0; i < N; i++) {
    int j = 0;
    for (int b = 0; b < lg; b++)
      if (i & (1 << b)) j |= 1 << (lg - 1 - b);
    if (j > i) {
      ocf_t t = a[i];
      a[i] = a[j];
      a[j] = t;
    }
  }
  for (int len = 2; len <= N; len <<= 1) {
    int half = len >> 1, step = N / len;
    for (int i = 0; i < N; i += len)
      for (int j = 0; j < half; j++) {
        ocf_t u = a[i + j];
        ocf_t v = cmul(a[i + j + half], w[j * step]);
        a[i + j].r = u.r + v.r;
        a[i + j].i = u.i + v.i;
        a[i + j + half].r = u.r - v.r;
        a[i + j + half].i = u.i - v.i;
      }
  }
}

void o_nco_tables(ocf_t* coarse, ocf_t* fine)
{
  for (int k = 0; k < 4096; k++) {
    double a = 2.0 * M_PI * (double)k / 4096.0;
    coarse[k].r = (float)cos(a);
    coarse[k].i = (float)sin(a);
  }
  for (int k = 0; k < 1024; k++) {
    double a = 2.0 * M_PI * (double)k / 4194304.0;
    fine[k].r = (float)cos(a);
    fine[k].i = (float)sin(a);
  }
}

/* phase increment per sample (2^32 = one turn) that REMOVES a carrier offset of cfo_hz */
uint32_t o_nco_dphi(float cfo_hz, int fft_size)
{
  double fs = 15000.0 * (double)fft_size;
  double turns = -(double)cfo_hz / fs;
  long long v = llrint(turns * 4294967296.0);
  return (uint32_t)(int32_t)v;
}

/* One subframe, one antenna: strip CP (160/144 scaled), optional NCO de-rotation, N-point FFT, keep the
 * 12*nprb carriers around DC (DC bin dropped): grid[l][k], k<6nprb -> bin N-6nprb+k, else bin k-6nprb+1. */
void o_ofdm_rx(const o_cell_t* cell, const ocf_t* in, uint32_t dphi, ocf_t* grid)
{
  int N = o_fft_size(cell->nof_prb);
  int nre = 12 * (int)cell->nof_prb;
  ocf_t* w = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)o_fft_twiddle_len(N));
  ocf_t* buf = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)N);
  static ocf_t coarse[4096], fine[1024];
  static int nco_init = 0;
  if (!nco_init) {
    o_nco_tables(coarse, fine);
    nco_init = 1;
  }
  o_fft_twiddles(N, w);
  int pos = 0;
  for (int l = 0; l < o_nsym(cell); l++) {
    int cp = cell->cp ? 512 * N / 2048 : ((l % 7) == 0 ? 160 : 144) * N / 2048; /* 36.211 Table 6.12-1: extended CP = N/4 on every symbol, 12 x 1.25 N = 15 N */
    pos += cp;
    for (int n = 0; n < N; n++) {
      ocf_t x = in[pos + n];
      if (dphi != 0) {
        uint32_t ph = (uint32_t)(pos + n) * dphi;
        ocf_t rot = cmul(coarse[ph >> 20], fine[(ph >> 10) & 1023u]);
        x = cmul(x, rot);
      }
      buf[n] = x;
    }
    o_fft(N, w, buf);
    for (int k = 0; k < nre; k++) {
      int bin = (k < nre / 2) ? (N - nre / 2 + k) : (k - nre / 2 + 1);
      grid[l * nre + k] = buf[bin];
    }
    pos += N;
  }
  free(w);
  free(buf);
}

/* ---- CRS (36.211 6.10.1) ---- */
/* ports 0, 1: symbols 0 and N_symb - 3 of both slots (normal CP: 0, 4, 7, 11; extended: 0, 3, 6, 9); ports 2, 3: symbol 1 of both slots (1, 8 / 1, 7) */
static int crs_nsym(int port) { return port < 2 ? 4 : 2; }
static int crs_l(const o_cell_t* cell, int port, int s) { return port < 2 ? o_crs_sym01(cell, s) : o_crs_sym23(cell, s); }

static int crs_koff(const o_cell_t* cell, int port, int s)
{
  int v;
  switch (port) {
    case 0: v = (s & 1) ? 3 : 0; break;
    case 1: v = (s & 1) ? 0 : 3; break;
    case 2: v = 3 * s; break;         /* 3 (n_s mod 2) */
    default: v = 3 + 3 * s; break;    /* 3 + 3 (n_s mod 2) */
  }
  return (v + (int)(cell->id % 6)) % 6;
}

void o_crs_table(const o_cell_t* cell, uint32_t sf_idx, ocf_t* crs)
{
  int nref = 2 * (int)cell->nof_prb;
  uint8_t c[2 * 220];
  for (uint32_t p = 0; p < cell->nof_ports; p += 2) /* ports 0/1 share their symbols and sequences, and so do ports 2/3 */
    for (int s = 0; s < crs_nsym((int)p); s++) {
      int l = crs_l(cell, (int)p, s);
      uint32_t ns = 2 * sf_idx + (l >= o_nslot(cell) ? 1u : 0u);
      uint32_t lslot = (uint32_t)(l % o_nslot(cell));
      uint32_t cinit = 1024u * (7u * (ns + 1u) + lslot + 1u) * (2u * cell->id + 1u) + 2u * cell->id + (cell->cp ? 0u : 1u); /* N_CP = 1 normal, 0 extended */
      o_gold(cinit, c, 2 * 220);
      for (int m = 0; m < nref; m++) {
        int mp = m + 110 - (int)cell->nof_prb;
        ocf_t r;
        r.r = c[2 * mp] ? -SQRT1_2F : SQRT1_2F;
        r.i = c[2 * mp + 1] ? -SQRT1_2F : SQRT1_2F;
        for (uint32_t q = p; q < p + 2 && q < cell->nof_ports; q++) crs[(q * 4 + (uint32_t)s) * (uint32_t)nref + (uint32_t)m] = r;
      }
    }
}

static void gauss_taps(float* t)
{
  /* srsran_chest_set_smooth_filter_gauss(order 4, std 1) [srsRAN]: 5 taps, unit sum */
  float sum = 0.0f;
  for (int i = 0; i < 5; i++) {
    float d = (float)(i - 2);
    t[i] = expf(-(d * d) / 2.0f);
  }
  for (int i = 0; i < 5; i++) sum = sum + t[i];
  float inv = 1.0f / sum;
  for (int i = 0; i < 5; i++) t[i] = t[i] * inv;
}

void o_chest(const o_cell_t* cell, uint32_t nof_rx, uint32_t sf_idx, const ocf_t* grid, ocf_t* ce, o_chest_res_t* res)
{
  int nprb = (int)cell->nof_prb, nre = 12 * nprb, nref = 2 * nprb;
  int P = (int)cell->nof_ports;
  ocf_t* crs = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)(P * 4 * nref));
  ocf_t* ls = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)(4 * nref));
  ocf_t* sm = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)(4 * nref));
  float* tmp = (float*)malloc(sizeof(float) * (size_t)(4 * nref));
  float taps[5];
  gauss_taps(taps);
  o_crs_table(cell, sf_idx, crs);
  memset(res, 0, sizeof(*res));
  ocf_t corr_tot = {0.0f, 0.0f};

  for (uint32_t rx = 0; rx < nof_rx; rx++) {
    const ocf_t* g = grid + (size_t)rx * 14u * (size_t)nre;
    for (int p = 0; p < P; p++) {
      const int S = crs_nsym(p), np = S * nref; /* pilot symbols of this port, pilots in the subframe */
      /* least squares at the pilots */
      for (int s = 0; s < S; s++) {
        int koff = crs_koff(cell, p, s);
        for (int m = 0; m < nref; m++)
          ls[s * nref + m] = cmulconj(g[crs_l(cell, p, s) * nre + 6 * m + koff], crs[(p * 4 + s) * nref + m]);
      }
      /* Gaussian smoothing across frequency, zero-padded edges (conv "same") */
      for (int s = 0; s < S; s++)
        for (int m = 0; m < nref; m++) {
          float ar = 0.0f, ai = 0.0f;
          for (int j = 0; j < 5; j++) {
            int q = m + j - 2;
            if (q < 0 || q >= nref) continue;
            ar = ar + taps[j] * ls[s * nref + q].r;
            ai = ai + taps[j] * ls[s * nref + q].i;
          }
          sm[s * nref + m].r = ar;
          sm[s * nref + m].i = ai;
        }
      /* noise (NOISE_ALG_REFS): mean |smoothed - ls|^2 */
      for (int i = 0; i < np; i++) {
        float dr = sm[i].r - ls[i].r, di = sm[i].i - ls[i].i;
        tmp[i] = dr * dr + di * di;
      }
      res->noise[rx][p] = o_reduce256(tmp, np) / (float)np;
      /* RSRP: |mean ls|^2 */
      for (int i = 0; i < np; i++) tmp[i] = ls[i].r;
      float mr = o_reduce256(tmp, np) / (float)np;
      for (int i = 0; i < np; i++) tmp[i] = ls[i].i;
      float mi = o_reduce256(tmp, np) / (float)np;
      res->rsrp[rx][p] = mr * mr + mi * mi;
      for (int i = 0; i < np; i++) tmp[i] = sm[i].r * sm[i].r + sm[i].i * sm[i].i;
      res->cepow[rx][p] = o_reduce256(tmp, np) / (float)np;
      if (p < 2) {
        /* CFO: correlate pilots one slot (0.5 ms) apart: symbols (7 vs 0) and (11 vs 4); ports 2, 3 change their subcarriers between the slots and stay out */
        for (int m = 0; m < nref; m++) {
          tmp[m] = cmulconj(ls[2 * nref + m], ls[0 * nref + m]).r;
          tmp[nref + m] = cmulconj(ls[3 * nref + m], ls[1 * nref + m]).r;
        }
        float cr = o_reduce256(tmp, 2 * nref);
        for (int m = 0; m < nref; m++) {
          tmp[m] = cmulconj(ls[2 * nref + m], ls[0 * nref + m]).i;
          tmp[nref + m] = cmulconj(ls[3 * nref + m], ls[1 * nref + m]).i;
        }
        float ci = o_reduce256(tmp, 2 * nref);
        corr_tot.r = corr_tot.r + cr;
        corr_tot.i = corr_tot.i + ci;
      }

      /* frequency interpolation (linear, pilot spacing 6, edge extrapolation) on the pilot symbols */
      ocf_t* c = ce + ((size_t)p * nof_rx + rx) * 14u * (size_t)nre;
      for (int s = 0; s < S; s++) {
        int koff = crs_koff(cell, p, s);
        ocf_t* row = c + crs_l(cell, p, s) * nre;
        const ocf_t* pl = sm + s * nref;
        for (int k = 0; k < nre; k++) {
          int m = (k - koff) >= 0 ? (k - koff) / 6 : 0;
          if (m > nref - 2) m = nref - 2;
          float dr = (pl[m + 1].r - pl[m].r) / 6.0f;
          float di = (pl[m + 1].i - pl[m].i) / 6.0f;
          float f = (float)(k - (6 * m + koff));
          row[k].r = pl[m].r + dr * f;
          row[k].i = pl[m].i + di * f;
        }
      }
      const int nsym = o_nsym(cell);
      if (p >= 2) {
        /* ports 2, 3 [srsRAN chest_dl.c interpolates its two pilot symbols the same way]: one line through symbols 1 and 8 (extended CP: 1 and 7) for the whole subframe */
        const int la = crs_l(cell, p, 0), lb = crs_l(cell, p, 1);
        const float dl_ = (float)(lb - la);
        for (int k = 0; k < nre; k++) {
          ocf_t c1 = c[la * nre + k], c8 = c[lb * nre + k];
          float dr = (c8.r - c1.r) / dl_, di = (c8.i - c1.i) / dl_;
          for (int l = 0; l < nsym; l++) {
            if (l == la || l == lb) continue;
            c[l * nre + k].r = c1.r + dr * (float)(l - la);
            c[l * nre + k].i = c1.i + di * (float)(l - la);
          }
        }
        continue;
      }
      /* time interpolation between the pilot symbols (normal CP 0, 4, 7, 11; extended 0, 3, 6, 9); the symbols behind the last pilot continue its slope */
      const int q0 = crs_l(cell, p, 0), q1 = crs_l(cell, p, 1), q2 = crs_l(cell, p, 2), q3 = crs_l(cell, p, 3);
      const float w01 = (float)(q1 - q0), w12 = (float)(q2 - q1), w23 = (float)(q3 - q2);
      for (int k = 0; k < nre; k++) {
        ocf_t c0 = c[q0 * nre + k], c4 = c[q1 * nre + k], c7 = c[q2 * nre + k], c11 = c[q3 * nre + k];
        float d01r = (c4.r - c0.r) / w01, d01i = (c4.i - c0.i) / w01;
        float d12r = (c7.r - c4.r) / w12, d12i = (c7.i - c4.i) / w12;
        float d23r = (c11.r - c7.r) / w23, d23i = (c11.i - c7.i) / w23;
        for (int l = q0 + 1; l < q1; l++) {
          c[l * nre + k].r = c0.r + d01r * (float)(l - q0);
          c[l * nre + k].i = c0.i + d01i * (float)(l - q0);
        }
        for (int l = q1 + 1; l < q2; l++) {
          c[l * nre + k].r = c4.r + d12r * (float)(l - q1);
          c[l * nre + k].i = c4.i + d12i * (float)(l - q1);
        }
        for (int l = q2 + 1; l < q3; l++) {
          c[l * nre + k].r = c7.r + d23r * (float)(l - q2);
          c[l * nre + k].i = c7.i + d23i * (float)(l - q2);
        }
        for (int l = q3 + 1; l < nsym; l++) {
          c[l * nre + k].r = c11.r + d23r * (float)(l - q3);
          c[l * nre + k].i = c11.i + d23i * (float)(l - q3);
        }
      }
    }
  }
  /* host-side scalars (product computes these on the host as well) */
  float ns = 0.0f, rs = 0.0f, cp = 0.0f;
  for (uint32_t rx = 0; rx < nof_rx; rx++)
    for (int p = 0; p < P; p++) {
      ns = ns + res->noise[rx][p];
      rs = rs + res->rsrp[rx][p];
      cp = cp + res->cepow[rx][p];
    }
  res->chan_ref = cp;
  float cnt = (float)(nof_rx * (uint32_t)P);
  res->noise_avg = ns / cnt;
  res->rsrp_avg = rs / cnt;
  res->snr_db = 10.0f * log10f(res->rsrp_avg / res->noise_avg);
  res->cfo_corr = corr_tot;
  res->cfo_hz = atan2f(corr_tot.i, corr_tot.r) / (2.0f * (float)M_PI * 0.0005f);
  free(crs);
  free(ls);
  free(sm);
  free(tmp);
}

/* ---- REGs (36.211 6.2.4), PCFICH (6.7.4), PHICH (6.9.3), PDCCH mapping (6.8.5) ---- */
static int reg_width(const o_cell_t* cell, int l)
{
  /* symbol 0 always leaves the CRS positions of two ports out; with four ports symbol 1 carries the CRS of ports 2, 3; with the extended CP symbol 3
   * (the fourth control symbol of a cell of at most 10 PRB) is the slot's second CRS symbol (36.211 6.2.4) */
  return (l == 0 || (l == 1 && cell->nof_ports == 4) || (l == 3 && cell->cp)) ? 6 : 4;
}

void o_regs_init(const o_cell_t* cell, o_regs_t* regs)
{
  int nprb = (int)cell->nof_prb, nre = 12 * nprb;
  memset(regs, 0, sizeof(*regs));
  int n0 = nre / 6;
  uint8_t* used0 = (uint8_t*)calloc((size_t)n0, 1);
  /* PCFICH */
  int kbar = 6 * (int)(cell->id % (2u * (uint32_t)nprb));
  for (int i = 0; i < 4; i++) {
    int k = (kbar + (i * nprb / 2) * 6) % nre;
    regs->pcfich_k0[i] = (uint16_t)k;
    used0[k / 6] = 1;
  }
  /* PHICH, normal duration: all in symbol 0; Ng/6 */
  int ng = (int)((cell->phich_ng_x6 * (uint32_t)nprb + 47u) / 48u); /* ceil(Ng * nprb / 8) with Ng = x/6 */
  regs->ngroups_phich = (uint32_t)(cell->cp ? 2 * ng : ng); /* extended CP: twice the groups, two of them per mapping unit (36.211 6.9, 6.9.3: m' = m / 2) - the REGs are the same */
  int navail = 0;
  int* avail = (int*)malloc(sizeof(int) * (size_t)n0);
  for (int i = 0; i < n0; i++)
    if (!used0[i]) avail[navail++] = i;
  for (int mp = 0; mp < ng; mp++)
    for (int i = 0; i < 3; i++) {
      int ni = ((int)cell->id + mp + (i * navail) / 3) % navail;
      used0[avail[ni]] = 1;
    }
  for (int cfi = 1; cfi <= 3; cfi++) {
    int nsym = cfi + (nprb <= 10 ? 1 : 0);
    int M = 0;
    static uint16_t tk[1200];
    static uint8_t tl[1200];
    for (int k = 0; k < nre; k++)
      for (int l = 0; l < nsym; l++) {
        int w = reg_width(cell, l);
        if (k % w) continue;
        if (l == 0 && used0[k / 6]) continue;
        tk[M] = (uint16_t)k;
        tl[M] = (uint8_t)l;
        M++;
      }
    regs->nof_regs[cfi - 1] = (uint32_t)M;
    regs->nof_cce[cfi - 1] = (uint32_t)(M / 9);
    /* sub-block interleaver on M quadruplets (36.212 5.1.4.2.1 pattern) + cyclic shift by N_ID:
     * REG m' carries quadruplet perm[(m' + N_ID) % M] */
    int R = (M + 31) / 32, ND = 32 * R - M;
    int* perm = (int*)malloc(sizeof(int) * (size_t)M);
    int n = 0;
    for (int j = 0; j < 32; j++)
      for (int r = 0; r < R; r++) {
        int idx = r * 32 + lsn_perm_cc[j];
        if (idx >= ND) perm[n++] = idx - ND;
      }
    for (int mprime = 0; mprime < M; mprime++) {
      int q = perm[(mprime + (int)cell->id) % M];
      if (q < 800) {
        regs->pdcch_reg_k0[cfi - 1][q] = tk[mprime];
        regs->pdcch_reg_l[cfi - 1][q] = tl[mprime];
      }
    }
    free(perm);
  }
  free(avail);
  free(used0);
}

/* equalise the 4 data REs of one REG -> 4 QPSK symbols (x[0..3]) */
static void reg_equalise(const o_cell_t* cell, uint32_t nof_rx, const ocf_t* grid, const ocf_t* ce, float noise,
                         int l, int k0, ocf_t* x)
{
  int nre = 12 * (int)cell->nof_prb;
  int kk[4], n = 0;
  if (reg_width(cell, l) == 6) {
    for (int k = k0; k < k0 + 6; k++)
      if ((k % 3) != (int)(cell->id % 3)) kk[n++] = k;
  } else {
    for (int k = k0; k < k0 + 4; k++) kk[n++] = k;
  }
  if (cell->nof_ports == 1) {
    for (int i = 0; i < 4; i++) {
      float nr = 0.0f, ni = 0.0f, den = 0.0f;
      for (uint32_t rx = 0; rx < nof_rx; rx++) {
        ocf_t y = grid[((size_t)rx * 14 + (size_t)l) * (size_t)nre + (size_t)kk[i]];
        ocf_t h = ce[((size_t)rx * 14 + (size_t)l) * (size_t)nre + (size_t)kk[i]]; /* port 0 */
        ocf_t t = cmulconj(y, h);
        float hp = h.r * h.r + h.i * h.i;
        if (rx == 0) {
          nr = t.r; ni = t.i; den = hp;
        } else {
          nr = nr + t.r; ni = ni + t.i; den = den + hp;
        }
      }
      den = den + noise;
      x[i].r = nr / den;
      x[i].i = ni / den;
    }
  } else {
    for (int i = 0; i < 4; i += 2) {
      float x0r = 0, x0i = 0, x1r = 0, x1i = 0, hh = 0;
      /* two ports: SFBC on ports (0, 1); four ports (SFBC-FSTD, 36.211 6.3.4.3): the first pair of the quadruplet on ports (0, 2), the second on (1, 3) */
      const size_t pa = (cell->nof_ports == 4 && i == 2) ? 1 : 0, pb = cell->nof_ports == 4 ? pa + 2 : 1;
      for (uint32_t rx = 0; rx < nof_rx; rx++) {
        size_t b0 = ((pa * (size_t)nof_rx + rx) * 14 + (size_t)l) * (size_t)nre;
        size_t b1 = ((pb * (size_t)nof_rx + rx) * 14 + (size_t)l) * (size_t)nre;
        size_t bg = ((size_t)rx * 14 + (size_t)l) * (size_t)nre;
        ocf_t r0 = grid[bg + (size_t)kk[i]], r1 = grid[bg + (size_t)kk[i + 1]];
        ocf_t h00 = ce[b0 + (size_t)kk[i]], h01 = ce[b0 + (size_t)kk[i + 1]];
        ocf_t h10 = ce[b1 + (size_t)kk[i]], h11 = ce[b1 + (size_t)kk[i + 1]];
        float hp = (h00.r * h00.r + h00.i * h00.i) + (h11.r * h11.r + h11.i * h11.i);
        ocf_t a = cmulconj(r0, h00);     /* conj(h00) r0 */
        ocf_t b = cmulconj(h11, r1);     /* h11 conj(r1) */
        ocf_t c = cmulconj(h10, r0);     /* h10 conj(r0) */
        ocf_t d = cmulconj(r1, h01);     /* conj(h01) r1 */
        float t0r = a.r + b.r, t0i = a.i + b.i, t1r = d.r - c.r, t1i = d.i - c.i;
        if (rx == 0) {
          x0r = t0r; x0i = t0i; x1r = t1r; x1i = t1i; hh = hp;
        } else {
          x0r = x0r + t0r; x0i = x0i + t0i; x1r = x1r + t1r; x1i = x1i + t1i; hh = hh + hp;
        }
      }
      x[i].r = x0r / hh * SQRT2F;
      x[i].i = x0i / hh * SQRT2F;
      x[i + 1].r = x1r / hh * SQRT2F;
      x[i + 1].i = x1i / hh * SQRT2F;
    }
  }
}

static const char* cfi_cw[3] = {"01101101101101101101101101101101", "10110110110110110110110110110110",
                                "11011011011011011011011011011011"};

uint32_t o_pcfich_decode(const o_cell_t* cell, const o_regs_t* regs, uint32_t nof_rx, uint32_t sf_idx,
                         const ocf_t* grid, const ocf_t* ce, float noise, float* corr3)
{
  float llr[32];
  uint8_t c[32];
  uint32_t cinit = (sf_idx + 1u) * (2u * cell->id + 1u) * 512u + cell->id; /* floor(ns/2)=sf_idx */
  o_gold(cinit, c, 32);
  for (int i = 0; i < 4; i++) {
    ocf_t x[4];
    reg_equalise(cell, nof_rx, grid, ce, noise, 0, regs->pcfich_k0[i], x);
    for (int j = 0; j < 4; j++) {
      llr[8 * i + 2 * j] = -(x[j].r * SQRT2F);
      llr[8 * i + 2 * j + 1] = -(x[j].i * SQRT2F);
    }
  }
  for (int i = 0; i < 32; i++)
    if (c[i]) llr[i] = -llr[i];
  uint32_t best = 0;
  float bestc = 0.0f;
  for (int w = 0; w < 3; w++) {
    float acc = 0.0f;
    for (int i = 0; i < 32; i++) acc = acc + (cfi_cw[w][i] == '1' ? llr[i] : -llr[i]);
    if (corr3) corr3[w] = acc;
    if (w == 0 || acc > bestc) {
      bestc = acc;
      best = (uint32_t)w;
    }
  }
  return best + 1;
}

void o_pdcch_llr(const o_cell_t* cell, const o_regs_t* regs, uint32_t nof_rx, uint32_t sf_idx, uint32_t cfi,
                 const ocf_t* grid, const ocf_t* ce, float noise, float* llr)
{
  uint32_t ncce = regs->nof_cce[cfi - 1];
  uint32_t nbits = 8u * regs->nof_regs[cfi - 1];
  uint8_t* c = (uint8_t*)malloc(nbits);
  o_gold(sf_idx * 512u + cell->id, c, (int)nbits);
  for (uint32_t q = 0; q < ncce * 9u; q++) {
    ocf_t x[4];
    reg_equalise(cell, nof_rx, grid, ce, noise, regs->pdcch_reg_l[cfi - 1][q], regs->pdcch_reg_k0[cfi - 1][q], x);
    for (int j = 0; j < 4; j++) {
      float a = -(x[j].r * SQRT2F), b = -(x[j].i * SQRT2F);
      llr[8 * q + 2 * (uint32_t)j] = c[8 * q + 2 * (uint32_t)j] ? -a : a;
      llr[8 * q + 2 * (uint32_t)j + 1] = c[8 * q + 2 * (uint32_t)j + 1] ? -b : b;
    }
  }
  free(c);
}

/* SubframePower.cc:18-42: sum over 14 symbols of the mean |x|^2 over the PRB's 12 REs, then dB - 10log10(14) */
void o_subframe_power(const o_cell_t* cell, const ocf_t* g, float* rb_power_db, float* pmin, float* pmax)
{
  int nprb = (int)cell->nof_prb, nre = 12 * nprb;
  float mx = -3.4e38f, mn = 3.4e38f;
  const float logdiv = 10.0f * log10f(14.0f);
  for (int i = 0; i < nprb; i++) {
    float acc = 0.0f;
    for (int j = 0; j < 14; j++) {
      float s = 0.0f;
      for (int k = 0; k < 12; k++) {
        ocf_t x = g[j * nre + i * 12 + k];
        s = s + (x.r * x.r + x.i * x.i);
      }
      acc = acc + s / 12.0f;
    }
    rb_power_db[i] = 10.0f * log10f(acc) - logdiv;
    if (rb_power_db[i] > mx) mx = rb_power_db[i];
    if (rb_power_db[i] < mn) mn = rb_power_db[i];
  }
  if (pmin) *pmin = mn;
  if (pmax) *pmax = mx;
}
