// lsn_rm.h - closed-form index arithmetic of the LTE turbo rate matcher (TS 36.212 5.1.4.1), usable on host and
// device.  The circular buffer w = v0 | interleave(v1, v2) is never materialised: for every buffer position (sub-block
// column, row) the number of non-<NULL> entries in front of it is a closed form, so de-rate-matching is a gather
// d[dest(j)] = sum_m e[eidx(j) + m*nn] with coalesced reads of e.
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
  int K, D, R, KP, ND, F, Ncb, nn, nn0, k0, cum_k0;
  int pre01[33];  // pre01[c] = number of <NULL> entries of v0 (== v1) in columns < c
  int pre2[33];   // same for v2
  uint8_t cnt01[32];
  uint8_t first2[32];
};

// 36.212 Table 5.1.4-1 inter-column permutation = 5-bit bit reversal (an involution)
LSN_HD int lsn_perm_tc_f(int c) { return ((c & 1) << 4) | ((c & 2) << 2) | (c & 4) | ((c & 8) >> 2) | ((c & 16) >> 4); }

// <NULL>s of v0/v1 at buffer positions before (col,row)
LSN_HD int lsn_rm_nb01_cr(const LsnRmGeom& g, int col, int row)
{
  if (col >= 32) return g.pre01[32];
  const int c = g.cnt01[col];
  return g.pre01[col] + (row < c ? row : c);
}
LSN_HD int lsn_rm_isnull01_cr(const LsnRmGeom& g, int col, int row) { return row < (int)g.cnt01[col]; }
// <NULL>s of v2 before (col,row)
LSN_HD int lsn_rm_nb2_cr(const LsnRmGeom& g, int col, int row)
{
  if (col >= 32) return g.pre2[32];
  return g.pre2[col] + ((row > 0) ? (int)g.first2[col] : 0);
}
// non-<NULL> entries in front of v0[k], v1[k], v2[k] in the circular buffer, k = col*R + row
LSN_HD int lsn_rm_cum_v0(const LsnRmGeom& g, int col, int row) { return col * g.R + row - lsn_rm_nb01_cr(g, col, row); }
LSN_HD int lsn_rm_cum_v1(const LsnRmGeom& g, int col, int row)
{
  const int k = col * g.R + row;
  return g.nn0 + (k - lsn_rm_nb01_cr(g, col, row)) + (k - lsn_rm_nb2_cr(g, col, row));
}
LSN_HD int lsn_rm_cum_v2(const LsnRmGeom& g, int col, int row) { return lsn_rm_cum_v1(g, col, row) + (lsn_rm_isnull01_cr(g, col, row) ? 0 : 1); }

// The geometry in two parts, so that the de-rate-matching kernel fills it with 32 lanes (one buffer column each, the prefix sums by a shuffle scan) while
// the host fills it in a loop: the <NULL> counts of one column ...
LSN_HD void lsn_rm_geom_col(int ND, int F, int c, int* cnt01, int* first2, int* cnt2)
{
  const int p = lsn_perm_tc_f(c), T01 = ND + F;
  *cnt01 = (T01 > p) ? (T01 - p + 31) / 32 : 0;
  *first2 = (p + 1 < ND) ? 1 : 0;  // the row-0 entry of this column of v2 is <NULL>
  *cnt2 = *first2 + ((p == 31 && ND > 0) ? 1 : 0);
}
// ... and the scalars, once cnt01 / first2 / pre01[0..32] / pre2[0..32] stand
LSN_HD void lsn_rm_geom_finish(LsnRmGeom& g, int K, int F, int rv)
{
  g.K = K; g.D = K + 4; g.R = (g.D + 31) / 32; g.KP = 32 * g.R; g.ND = g.KP - g.D; g.F = F; g.Ncb = 3 * g.KP;
  g.nn0 = g.KP - g.pre01[32];
  g.nn = 3 * g.KP - 2 * g.pre01[32] - g.pre2[32];
  const int c0 = 2 * ((g.Ncb + 8 * g.R - 1) / (8 * g.R)) * rv + 2;  // k0 = R * c0: always at row 0 of a column
  g.k0 = g.R * c0;
  if (c0 < 32) {
    g.cum_k0 = lsn_rm_cum_v0(g, c0, 0);
  } else {
    const int jp = g.k0 - g.KP, k = jp >> 1, col = k / g.R, row = k - col * g.R;
    g.cum_k0 = (col >= 32) ? g.nn : ((jp & 1) ? lsn_rm_cum_v2(g, col, row) : lsn_rm_cum_v1(g, col, row));
  }
}
LSN_HD void lsn_rm_geom(LsnRmGeom& g, int K, int F, int rv)
{
  const int D = K + 4, R = (D + 31) / 32, ND = 32 * R - D;
  g.R = R;
  int a01 = 0, a2 = 0;
  for (int c = 0; c < 32; c++) {
    int c01, f2, c2;
    lsn_rm_geom_col(ND, F, c, &c01, &f2, &c2);
    g.cnt01[c] = (uint8_t)c01; g.first2[c] = (uint8_t)f2;
    g.pre01[c] = a01; a01 += c01;
    g.pre2[c] = a2; a2 += c2;
  }
  g.pre01[32] = a01; g.pre2[32] = a2;
  lsn_rm_geom_finish(g, K, F, rv);
}

// first e index that lands on buffer position with `cum` non-<NULL> predecessors
LSN_HD int lsn_rm_eidx(const LsnRmGeom& g, int cum)
{
  int r = cum - g.cum_k0;
  return r < 0 ? r + g.nn : r;
}

// rank (first e index) of destination (stream s in 0..2, index i in 0..D-1); -1 if the destination is <NULL>/filler
LSN_HD int lsn_rm_rank(const LsnRmGeom& g, int s, int i)
{
  if (s < 2) {
    if (i < g.F) return -1;
    const int y = i + g.ND, row = y >> 5, col = lsn_perm_tc_f(y & 31);
    return lsn_rm_eidx(g, s == 0 ? lsn_rm_cum_v0(g, col, row) : lsn_rm_cum_v1(g, col, row));
  }
  int z = i + g.ND - 1;
  if (z < 0) z += g.KP;
  const int row = z >> 5, col = lsn_perm_tc_f(z & 31);
  return lsn_rm_eidx(g, lsn_rm_cum_v2(g, col, row));
}

// The same ranks from ONE 8-byte table entry per buffer column (k_rm: the entry is an LDS read, everything else registers):
//   a = (col R - pre01[col] - cum_k0) * 256 + cnt01[col],   b = (nn0 + 2 col R - pre01[col] - pre2[col] - cum_k0) * 256 + first2[col]
// so that with m = min(row, cnt01)  rank(v0) = (a >> 8) + row - m,  rank(v1) = (b >> 8) + 2 row - m - (row > 0 ? first2 : 0),
// rank(v2) = rank(v1) + (row >= cnt01), each + nn when negative (lsn_rm_eidx).  y = i + ND for the streams 0 / 1, z = i + ND - 1 for stream 2
// (ND is 4, 12, 20 or 28 for the block sizes of 36.212 Table 5.1.3-3 - K is a multiple of 8 -, so z never wraps).
struct LsnRmCol { int32_t a, b; };
LSN_HD LsnRmCol lsn_rm_fast_col(const LsnRmGeom& g, int c)
{
  LsnRmCol o;
  o.a = (c * g.R - g.pre01[c] - g.cum_k0) * 256 + (int)g.cnt01[c];
  o.b = (g.nn0 + 2 * c * g.R - g.pre01[c] - g.pre2[c] - g.cum_k0) * 256 + (int)g.first2[c];
  return o;
}
LSN_HD int lsn_rm_col_of(int y)
{
#ifdef __HIP_DEVICE_COMPILE__
  return (int)(__builtin_bitreverse32((uint32_t)y) >> 27);
#else
  return lsn_perm_tc_f(y & 31);
#endif
}
// filler positions (i < F) are the caller's business: the ranks returned for them are meaningless
LSN_HD void lsn_rm_rank01_fast(LsnRmCol e, int nn, int y, int* r0, int* r1)
{
  const int row = y >> 5, cnt = e.a & 255, m = row < cnt ? row : cnt;
  int a = (e.a >> 8) + row - m;
  int b = (e.b >> 8) + 2 * row - m - (row > 0 ? (e.b & 1) : 0);
  *r0 = a < 0 ? a + nn : a;
  *r1 = b < 0 ? b + nn : b;
}
LSN_HD int lsn_rm_rank2_fast(LsnRmCol e, int nn, int z)
{
  const int row = z >> 5, cnt = e.a & 255, m = row < cnt ? row : cnt;
  const int b = (e.b >> 8) + 2 * row - m - (row > 0 ? (e.b & 1) : 0) + (row < cnt ? 0 : 1);
  return b < 0 ? b + nn : b;
}

// Number of trellis windows the turbo decoder cuts a code block of K bits into (windows of W = K / P >= 32 steps,
// decoded in parallel with next-iteration boundary initialisation): the largest divisor of K that fills one wavefront
// (P <= 64), or two wavefronts when a divisor in 96..128 exists.
LSN_HD int lsn_turbo_nwin(int K)
{
  int p1 = 1, p2 = 1;
  for (int P = (K / 32 < 128 ? K / 32 : 128); P >= 1; P--)
    if (K % P == 0) { p2 = P; break; }
  for (int P = (K / 32 < 64 ? K / 32 : 64); P >= 1; P--)
    if (K % P == 0) { p1 = P; break; }
  return p2 >= 96 ? p2 : p1;
}
// Blocks of at most 64 windows whose K exceeds this bound are decoded by the two-wavefront kernel too (its second wavefront leaves at once):
// their 40 KiB of LDS would otherwise set the LDS size - and with it the occupancy - of every one-wavefront launch they are part of.
#define LSN_TURBO_ONE_WAVE_KMAX 3072
// words of the turbo decoder's interleaver address table of block size K (two trellis steps per word, lsn_turbo_core.h)
LSN_HD int lsn_turbo_il_words(int K) { const int P = lsn_turbo_nwin(K), W = K / P; return ((W + 1) / 2) * P; }
LSN_HD bool lsn_turbo_two_wave_class(int K) { return lsn_turbo_nwin(K) > 64 || K > LSN_TURBO_ONE_WAVE_KMAX; }
