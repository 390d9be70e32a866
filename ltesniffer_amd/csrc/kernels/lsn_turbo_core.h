// lsn_turbo_core.h - the per-lane arithmetic of the turbo decoder kernel (stage_c.hip: k_turbo), written so that the SAME text compiles for
// gfx950 and for the host: tests/native/test_turbo_core.cc runs it lane by lane over a plain-memory stand-in of the LDS and compares every
// decision with the CPU oracle's decoder - the packed arithmetic below is checked without a GPU (and, in that build, every packed add is
// range-checked against 32-bit arithmetic).
//
// One lane = one trellis window (see stage_c.hip for the schedule).  Round 3, second half: the recursions run on PACKED int16 pairs.
// With one 381-register wavefront per SIMD a wave issues one VALU instruction per 4 cycles whatever its class (profiles/r03_valu_peak_isa.txt),
// so v_pk_add_i16 / v_pk_max_i16 do two state updates for the price of one: 98 instead of 140 instructions per trellis step.
//
// Layout C: four registers hold the eight state metrics as (m[k] | m[k+4] << 16), k = 0..3.
//  * forward: butterfly k reads states 2k, 2k+1 and writes k, k+4 - with the operands taken as half-broadcasts (op_sel) layout C maps to
//    itself: 8 v_pk_add + 4 v_pk_max per step.
//  * backward: the pairs that meet alpha (k, k+4) need beta pairs (0,2) (4,6) (1,3) (5,7): four v_perm per step, then 8 v_pk_add
//    (branch + beta), 8 v_pk_add (+ alpha), 6 + 4 v_pk_max; the result is layout C again.
// Word length (int16, no wrap-around anywhere - v_pk_max_i16 compares values, not residues):
//   |gamma| <= |sys| + |ext| + |par| <= 512 + 2047 + 512 = 3071 per step.  Any state is reached from any state in 3 steps, so a metric
//   vector normalised to state 0 spreads at most 3 * 3071 = 9213 once it is 3 steps away from its initialisation; the initialisations are
//   (0, -12000 x 7), the termination metrics (<= 3 * 1022) and boundary metrics of the previous iteration (<= 9213), i.e. <= 12000 + 2 * 3071
//   = 18142 in the first two steps of the first window and <= 15355 elsewhere.  Alpha is normalised every step where it is stored (every
//   second step in the forward sweep: + 3071), beta every step, and a window is at least 32 steps long, so the two "young" ends never meet:
//   |alpha + beta + gamma| <= max(18142 + 9213, 9213 + 15355) + 3071 = 30426 < 32768.  (tools/turbo_metric_ranges.py measures <= 19920.)
//   L = m1 - m0 is formed in 32 bits.
#pragma once
#include <stdint.h>
#include <type_traits>
#include "lsn_rm.h"
#ifndef LSN_NEG_METRIC
#define LSN_NEG_METRIC (-12000)
#define LSN_EXT_CLIP 2047
#endif

typedef short lsn_s2 __attribute__((ext_vector_type(2)));
typedef lsn_s2 s2;

#ifdef LSN_TURBO_RANGE_CHECK  // host test build only
#include <cstdio>
#include <cstdlib>
static inline s2 pk_chk(int x, int y, const char* what)
{
  if (x < -32768 || x > 32767 || y < -32768 || y > 32767) { std::fprintf(stderr, "int16 range exceeded in %s: %d %d\n", what, x, y); std::abort(); }
  return s2{(short)x, (short)y};
}
static inline s2 pka(s2 a, s2 b) { return pk_chk((int)a.x + (int)b.x, (int)a.y + (int)b.y, "add"); }
static inline s2 pks(s2 a, s2 b) { return pk_chk((int)a.x - (int)b.x, (int)a.y - (int)b.y, "sub"); }
#else
LSN_HD s2 pka(s2 a, s2 b) { return a + b; }
LSN_HD s2 pks(s2 a, s2 b) { return a - b; }
#endif
LSN_HD s2 pkmax(s2 a, s2 b) { return __builtin_elementwise_max(a, b); }
LSN_HD uint32_t pk_u32(s2 v) { return __builtin_bit_cast(uint32_t, v); }
LSN_HD s2 pk_s2(uint32_t w) { return __builtin_bit_cast(s2, w); }

// sign(x) * min(floor(3 |x| / 4), LSN_EXT_CLIP) without a select: (3x + (3x < 0 ? 3 : 0)) >> 2 truncates towards zero, v_med3 clips
LSN_HD int lsn_ext_scale(int x)
{
  const int t = x + (x << 1);
  const int r = (t + (int)((uint32_t)t >> 30)) >> 2;  // |t| < 2^30: the two top bits are 11 exactly when t < 0
  return r < -LSN_EXT_CLIP ? -LSN_EXT_CLIP : (r > LSN_EXT_CLIP ? LSN_EXT_CLIP : r);
}

// forward step, layout C in place; q = (lsa | lp << 16): the four branch metrics are 0, g01 = lp, g10 = lsa, g11 = lsa + lp
template <bool NORM>
LSN_HD void lsn_step_fwd_pk(s2* a, s2 q)
{
  const s2 gg = pka(q, q.yx);    // (g11, g11)
  const s2 p = s2{0, gg.y};      // (0, g11)
  const s2 x0 = pka(a[0].xx, p), y0 = pka(a[1].xx, p.yx);     // -> states 0, 4 from 0, 1
  const s2 x1 = pka(a[2].xx, q), y1 = pka(a[3].xx, q.yx);     // -> 1, 5 from 2, 3
  const s2 x2 = pka(a[0].yy, q.yx), y2 = pka(a[1].yy, q);     // -> 2, 6 from 4, 5
  const s2 x3 = pka(a[2].yy, p.yx), y3 = pka(a[3].yy, p);     // -> 3, 7 from 6, 7
  a[0] = pkmax(x0, y0); a[1] = pkmax(x1, y1); a[2] = pkmax(x2, y2); a[3] = pkmax(x3, y3);
  if (NORM) {
    const s2 n = a[0].xx;
    a[0] = pks(a[0], n); a[1] = pks(a[1], n); a[2] = pks(a[2], n); a[3] = pks(a[3], n);
  }
}

// backward step: b (layout C, normalised) becomes the beta vector one step earlier (normalised); A = the alphas in front of this step (layout C,
// normalised); returns L = max over branches with input 1 - max over branches with input 0
LSN_HD int lsn_step_bwd_pk(s2* b, const s2* A, s2 q)
{
  const s2 G0 = __builtin_shufflevector(b[0], b[2], 0, 2), G1 = __builtin_shufflevector(b[0], b[2], 1, 3);  // (b0, b2) (b4, b6)
  const s2 G2 = __builtin_shufflevector(b[1], b[3], 0, 2), G3 = __builtin_shufflevector(b[1], b[3], 1, 3);  // (b1, b3) (b5, b7)
  const s2 gg = pka(q, q.yx);
  const s2 S = s2{0, q.y};       // (0, g01)
  const s2 T = s2{gg.x, q.x};    // (g11, g10)
  // successor metric + branch metric, paired like alpha: (state k, state k + 4); input 0 and input 1
  const s2 u00 = pka(G0, S), u01 = pka(G1, S), u02 = pka(G3, S.yx), u03 = pka(G2, S.yx);
  const s2 u10 = pka(G1, T), u11 = pka(G0, T), u12 = pka(G2, T.yx), u13 = pka(G3, T.yx);
  const s2 M0 = pkmax(pkmax(pka(A[0], u00), pka(A[1], u01)), pkmax(pka(A[2], u02), pka(A[3], u03)));
  const s2 M1 = pkmax(pkmax(pka(A[0], u10), pka(A[1], u11)), pkmax(pka(A[2], u12), pka(A[3], u13)));
  const s2 m0 = pkmax(M0, M0.yx), m1 = pkmax(M1, M1.yx);
  b[0] = pkmax(u00, u10); b[1] = pkmax(u01, u11); b[2] = pkmax(u02, u12); b[3] = pkmax(u03, u13);
  const s2 n = b[0].xx;
  b[0] = pks(b[0], n); b[1] = pks(b[1], n); b[2] = pks(b[2], n); b[3] = pks(b[3], n);
  return (int)m1.x - (int)m0.x;
}

// eight 32-bit metrics (state order) -> layout C
LSN_HD void lsn_pack_c(const int* m, s2* c)
{
  for (int k = 0; k < 4; k++) c[k] = s2{(short)m[k], (short)m[k + 4]};
}

struct TurboLds {
  uint32_t* spp;   // [K] sys | p1 << 10 | p2 << 20 (10-bit two's complement fields), transposed
  int16_t* ext;    // [K] extrinsic * 2 + hard bit, transposed
  uint8_t* ckpt;   // check-point slots of 14 * NT bytes each: [3][NT] words (states 1|5, 2|6, 3|7) + [NT] halves (state 4); state 0 is 0
};
template <int NT>
LSN_HD void lsn_ckpt_store(uint8_t* area, int slot, int lane, const s2* a)
{
  uint32_t* w = (uint32_t*)(area + (size_t)slot * 14 * NT);
  w[lane] = pk_u32(a[1]); w[NT + lane] = pk_u32(a[2]); w[2 * NT + lane] = pk_u32(a[3]);
  ((int16_t*)(w + 3 * NT))[lane] = a[0].y;
}
template <int NT>
LSN_HD void lsn_ckpt_load(const uint8_t* area, int slot, int lane, s2* a)
{
  const uint32_t* w = (const uint32_t*)(area + (size_t)slot * 14 * NT);
  a[1] = pk_s2(w[lane]); a[2] = pk_s2(w[NT + lane]); a[3] = pk_s2(w[2 * NT + lane]);
  a[0] = s2{0, ((const int16_t*)(w + 3 * NT))[lane]};
}

#ifdef __HIP_DEVICE_COMPILE__
#define LSN_UMUL24(a, b) __umul24(a, b)
#define LSN_MUL24(a, b) __mul24(a, b)
#else
#define LSN_UMUL24(a, b) ((uint32_t)(a) * (uint32_t)(b))
#define LSN_MUL24(a, b) ((int)(a) * (int)(b))
#endif
// x -> (x % W) * P + x / W with 24-bit multiplies; magicW = ceil(2^20 / W) is exact for x < 6144, W <= 96 (error x / 2^20 < 1 / W)
LSN_HD int tr_idx(int x, int W, int P, uint32_t magicW)
{
  const int q = (int)(LSN_UMUL24((uint32_t)x, magicW) >> 20);
  return LSN_MUL24(x - LSN_MUL24(q, W), P) + q;
}
LSN_HD int fld0(uint32_t w) { return (int)(w << 22) >> 22; }
LSN_HD int fld1(uint32_t w) { return (int)(w << 12) >> 22; }
LSN_HD int fld2(uint32_t w) { return (int)(w << 2) >> 22; }

// Sub-block length (steps whose operands are fetched in one burst and whose alphas are kept in registers); even: the forward sweep
// normalises every second step
#ifndef TB_S
#define TB_S 16
#endif
// check-point slots: sub-blocks 1 .. nsb-2; 64 threads: W <= 96, 128 threads: W <= 64.  The area also carries the window-boundary exchange
// (2 slots) and, before the first iteration, the 12 termination values (48 bytes at offset 2048).
#define TB_CKPT_SLOTS(NT) ((((NT) == 64 ? 96 : 64) + TB_S - 1) / TB_S - 2)
#define TB_CKPT_BYTES(NT) ((size_t)((TB_CKPT_SLOTS(NT) < 2 ? 2 : TB_CKPT_SLOTS(NT)) * 14 * (NT)) < 2200 ? (size_t)2200 : (size_t)((TB_CKPT_SLOTS(NT) < 2 ? 2 : TB_CKPT_SLOTS(NT)) * 14 * (NT)))

// One constituent decoder, the part of one lane (= window `wl`; idle lanes shadow window 0 and write their soft output to the spare slot
// ext[K]).  nii_a / nii_b: boundary metrics of the previous iteration (layout C); beta_tail: termination metrics (layout C);
// a_end / b_out: this window's metrics at its end / start, for the exchange between the lanes (the caller's business).
template <bool IL, int NT>
LSN_HD void lsn_map_pass_lane(const TurboLds& m, int lane, bool active, int K, int P, int W, uint32_t magicW, int f1, int f2,
                              const s2* nii_a, const s2* nii_b, const s2* beta_tail, s2* a_end, s2* b_out)
{
  const int wl = active ? lane : 0;
  const int t0 = wl * W;
  const int nsb = (W + TB_S - 1) / TB_S;
  s2 a[4], b[4], a0[4];
  if (wl == 0) {
    a[0] = s2{0, (short)LSN_NEG_METRIC};
    a[1] = a[2] = a[3] = s2{(short)LSN_NEG_METRIC, (short)LSN_NEG_METRIC};
  } else {
    for (int k = 0; k < 4; k++) a[k] = nii_a[k];
  }
  for (int k = 0; k < 4; k++) a0[k] = a[k];
  int pi = t0, gq = 0;
  const int twof2 = (2 * f2) % K;
  if (IL) {
    pi = (int)(((long long)f1 * t0 + (long long)f2 * t0 % K * t0) % K);
    gq = (int)(((long long)f1 + f2 + 2ll * f2 % K * t0) % K);
  }
  uint32_t g[TB_S];  // operands of one sub-block: lsa (low half) | lp << 16
  // ---- forward sweep over sub-blocks 0 .. nsb-2 (the last one is covered by the recompute below) ----
  for (int sb = 0; sb + 1 < nsb; sb++) {
    if (sb >= 1) lsn_ckpt_store<NT>(m.ckpt, sb - 1, lane, a);
    const int tb = sb * TB_S;
#pragma unroll
    for (int u = 0; u < TB_S; u++) {
      const int nat = (tb + u) * P + wl;
      if (IL) {
        const int idx = tr_idx(pi, W, P, magicW);
        pi += gq; pi = pi >= K ? pi - K : pi; gq += twof2; gq = gq >= K ? gq - K : gq;
        g[u] = ((uint32_t)(fld0(m.spp[idx]) + ((int)m.ext[idx] >> 1)) & 0xFFFFu) | ((uint32_t)fld2(m.spp[nat]) << 16);
      } else {
        const uint32_t w = m.spp[nat];
        g[u] = ((uint32_t)(fld0(w) + ((int)m.ext[nat] >> 1)) & 0xFFFFu) | ((uint32_t)fld1(w) << 16);
      }
    }
#pragma unroll
    for (int u = 0; u < TB_S; u += 2) {
      lsn_step_fwd_pk<false>(a, pk_s2(g[u]));
      lsn_step_fwd_pk<true>(a, pk_s2(g[u + 1]));
    }
  }
  if (IL) {  // interleaver state -> end of the window
    for (int t = (nsb - 1) * TB_S; t < W; t++) { pi += gq; pi = pi >= K ? pi - K : pi; gq += twof2; gq = gq >= K ? gq - K : gq; }
  }
  if (wl == P - 1) {
    for (int k = 0; k < 4; k++) b[k] = beta_tail[k];
  } else {
    for (int k = 0; k < 4; k++) b[k] = nii_b[k];
  }
  // ---- backward, sub-block by sub-block ----
  int ix[TB_S];     // LDS index of the systematic / extrinsic value of each step
  s2 A[TB_S][4];    // alphas in front of each step of the sub-block
  for (int sb = nsb - 1; sb >= 0; sb--) {
    const int tb = sb * TB_S, n = (tb + TB_S < W) ? TB_S : W - tb;
    if (sb + 1 < nsb) {
      if (sb == 0) {
        for (int k = 0; k < 4; k++) a[k] = a0[k];
      } else {
        lsn_ckpt_load<NT>(m.ckpt, sb - 1, lane, a);
      }
    }
    // only the last sub-block of a window can be shorter than TB_S: the full-length variant carries no per-step guards
    auto subblock = [&](auto fullc) {
      constexpr bool FULL = decltype(fullc)::value;
      // operand burst, last step first (the QPP recursion runs in reverse)
#pragma unroll
      for (int u = TB_S - 1; u >= 0; u--) {
        if (FULL || u < n) {
          const int nat = (tb + u) * P + wl;
          if (IL) {
            gq -= twof2; gq = gq < 0 ? gq + K : gq; pi -= gq; pi = pi < 0 ? pi + K : pi;
            const int idx = tr_idx(pi, W, P, magicW);
            ix[u] = active ? idx : K;
            g[u] = ((uint32_t)(fld0(m.spp[idx]) + ((int)m.ext[idx] >> 1)) & 0xFFFFu) | ((uint32_t)fld2(m.spp[nat]) << 16);
          } else {
            const uint32_t w = m.spp[nat];
            ix[u] = active ? nat : K;
            g[u] = ((uint32_t)(fld0(w) + ((int)m.ext[nat] >> 1)) & 0xFFFFu) | ((uint32_t)fld1(w) << 16);
          }
        }
      }
      // recompute the alphas of this sub-block into registers
#pragma unroll
      for (int u = 0; u < TB_S; u++) {
        if (FULL || u < n) {
          A[u][0] = a[0]; A[u][1] = a[1]; A[u][2] = a[2]; A[u][3] = a[3];
          lsn_step_fwd_pk<true>(a, pk_s2(g[u]));
        }
      }
      if (sb == nsb - 1) {
        for (int k = 0; k < 4; k++) a_end[k] = a[k];
      }
      // beta recursion + LLR + extrinsic
#pragma unroll
      for (int u = TB_S - 1; u >= 0; u--) {
        if (FULL || u < n) {
          const s2 q = pk_s2(g[u]);
          const int L = lsn_step_bwd_pk(b, A[u], q);
          const int hard = L < 0 ? 0 : (L > 1 ? 1 : L);  // v_med3_i32(L, 0, 1)
          m.ext[ix[u]] = (int16_t)((lsn_ext_scale(L - (int)q.x) << 1) | hard);  // idle lanes: spare slot ext[K]
        }
      }
    };
    if (n == TB_S) subblock(std::true_type{}); else subblock(std::false_type{});
  }
  for (int k = 0; k < 4; k++) b_out[k] = b[k];
}
