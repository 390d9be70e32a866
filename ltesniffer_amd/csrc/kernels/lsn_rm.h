// lsn_rm.h - closed-form index arithmetic of the LTE turbo rate matcher (TS 36.212 5.1.4.1), usable on host and
// device.  Instead of walking the circular buffer, every destination soft bit computes the rank it has among the
// non-<NULL> entries after k0, so that de-rate-matching becomes a gather: d = sum_m e[rank + m*nn].
#pragma once
#include <stdint.h>
#ifndef LSN_HD
#ifdef __HIPCC__
#define LSN_HD __host__ __device__ __forceinline__
#else
#define LSN_HD static inline
#endif
#endif

struct LsnRmGeom {
  int K, D, R, KP, ND, F, Ncb, nn, k0, cum_k0;
  int pre01[33];  // pre01[c] = number of <NULL> entries of v0 (== v1) in columns < c
  int pre2[33];   // same for v2
  uint8_t cnt01[32];
};

// 36.212 Table 5.1.4-1 inter-column permutation and its inverse
LSN_HD int lsn_perm_tc_f(int c)
{
  const uint8_t p[32] = {0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30, 1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31};
  return p[c];
}
LSN_HD int lsn_perm_tc_inv(int v)
{
  // P is the 5-bit bit reversal, which is an involution
  return lsn_perm_tc_f(v);
}

// number of <NULL>s of v0/v1 at positions < k (k = col*R + row)
LSN_HD int lsn_rm_nb01(const LsnRmGeom& g, int k)
{
  int col = k / g.R, row = k - col * g.R;
  if (col >= 32) return g.pre01[32];
  int c = g.cnt01[col];
  return g.pre01[col] + (row < c ? row : c);
}
LSN_HD int lsn_rm_isnull01(const LsnRmGeom& g, int k)
{
  int col = k / g.R, row = k - col * g.R;
  return row * 32 + lsn_perm_tc_f(col) < g.ND + g.F;
}
// number of <NULL>s of v2 at positions < k
LSN_HD int lsn_rm_nb2(const LsnRmGeom& g, int k)
{
  int col = k / g.R, row = k - col * g.R;
  if (col >= 32) return g.pre2[32];
  int first = (lsn_perm_tc_f(col) + 1 < g.ND) ? 1 : 0;  // the row-0 entry of this column is <NULL>
  return g.pre2[col] + ((row > 0) ? first : 0);
}
// non-<NULL> entries of the circular buffer at indices < j
LSN_HD int lsn_rm_cum(const LsnRmGeom& g, int j)
{
  if (j < g.KP) return j - lsn_rm_nb01(g, j);
  int jp = j - g.KP, k = jp >> 1, odd = jp & 1;
  int nn0 = g.KP - g.pre01[32];
  int v = nn0 + (k - lsn_rm_nb01(g, k)) + (k - lsn_rm_nb2(g, k));
  if (odd) v += lsn_rm_isnull01(g, k) ? 0 : 1;
  return v;
}

LSN_HD void lsn_rm_geom(LsnRmGeom& g, int K, int F, int rv)
{
  g.K = K; g.D = K + 4; g.R = (g.D + 31) / 32; g.KP = 32 * g.R; g.ND = g.KP - g.D; g.F = F; g.Ncb = 3 * g.KP;
  int T01 = g.ND + F, a01 = 0, a2 = 0;
  for (int c = 0; c < 32; c++) {
    int p = lsn_perm_tc_f(c);
    int c01 = (T01 > p) ? (T01 - p + 31) / 32 : 0;
    g.cnt01[c] = (uint8_t)c01;
    g.pre01[c] = a01; a01 += c01;
    int c2 = ((p + 1 < g.ND) ? 1 : 0) + ((p == 31 && g.ND > 0) ? 1 : 0);
    g.pre2[c] = a2; a2 += c2;
  }
  g.pre01[32] = a01; g.pre2[32] = a2;
  g.nn = 3 * g.KP - 2 * a01 - a2;
  g.k0 = g.R * (2 * ((g.Ncb + 8 * g.R - 1) / (8 * g.R)) * rv + 2);
  g.cum_k0 = lsn_rm_cum(g, g.k0);
}

// rank (first e index) of destination (stream s in 0..2, index i in 0..D-1); -1 if the destination is <NULL>/filler
LSN_HD int lsn_rm_rank(const LsnRmGeom& g, int s, int i)
{
  int j;
  if (s < 2) {
    if (i < g.F) return -1;
    int y = i + g.ND, row = y >> 5, col = lsn_perm_tc_inv(y & 31), k = col * g.R + row;
    j = (s == 0) ? k : g.KP + 2 * k;
  } else {
    int y2 = i + g.ND;
    int z = y2 - 1; if (z < 0) z += g.KP;
    int row = z >> 5, col = lsn_perm_tc_inv(z & 31), k = col * g.R + row;
    j = g.KP + 2 * k + 1;
  }
  int r = lsn_rm_cum(g, j) - g.cum_k0;
  if (r < 0) r += g.nn;
  return r;
}
