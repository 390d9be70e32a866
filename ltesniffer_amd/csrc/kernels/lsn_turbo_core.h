// lsn_turbo_core.h - the per-lane arithmetic of the turbo decoder kernel (stage_c.hip: k_turbo), written so that the SAME text compiles for
// gfx950 and for the host: tests/native/test_turbo_core.cc runs it lane by lane over a plain-memory stand-in of the LDS and compares every
// decision with the CPU oracle's decoder - the packed arithmetic below is checked without a GPU (and, in that build, every packed add is
// range-checked against 32-bit arithmetic).
//
// One lane = one trellis window (see stage_c.hip for the schedule).  Round 3, second half: the recursions run on PACKED int16 pairs.
// A wavefront alone on a SIMD issues one VALU instruction per 4 cycles whatever its class (profiles/r03_valu_peak_isa.txt),
// and packed forms occupy the VALU port for 4 cycles at any occupancy, so v_pk_add_i16 / v_pk_max_i16 do two state updates for the price of one:
// 106 instead of 140 vector instructions per trellis step in the first constituent decoder, 107 instead of 171 in the interleaved one.
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
// packed add / subtract with a free choice of the source halves (VOP3P op_sel): result.lo = a.{AL} +- b.{BL}, result.hi = a.{AH} +- b.{BH}
// (0 = low half, 1 = high half).  The compiler folds broadcasts into op_sel by itself but spends a v_alignbit on every swapped operand; the
// device build therefore states the modifiers itself.
#ifdef __HIP_DEVICE_COMPILE__
template <int AL, int AH, int BL, int BH>
__device__ __forceinline__ s2 pka_sel(s2 a, s2 b)
{
  s2 d;
  asm("v_pk_add_u16 %0, %1, %2 op_sel:[%3,%4] op_sel_hi:[%5,%6]" : "=v"(d) : "v"(a), "v"(b), "n"(AL), "n"(BL), "n"(AH), "n"(BH));
  return d;
}
template <int AL, int AH, int BL, int BH>
__device__ __forceinline__ s2 pks_sel(s2 a, s2 b)
{
  s2 d;
  asm("v_pk_sub_i16 %0, %1, %2 op_sel:[%3,%4] op_sel_hi:[%5,%6]" : "=v"(d) : "v"(a), "v"(b), "n"(AL), "n"(BL), "n"(AH), "n"(BH));
  return d;
}
#else
template <int AL, int AH, int BL, int BH>
static inline s2 pka_sel(s2 a, s2 b) { return pka(s2{AL ? a.y : a.x, AH ? a.y : a.x}, s2{BL ? b.y : b.x, BH ? b.y : b.x}); }
template <int AL, int AH, int BL, int BH>
static inline s2 pks_sel(s2 a, s2 b) { return pks(s2{AL ? a.y : a.x, AH ? a.y : a.x}, s2{BL ? b.y : b.x, BH ? b.y : b.x}); }
#endif

// sign(x) * min(floor(3 |x| / 4), LSN_EXT_CLIP) without a select: (3x + (3x < 0 ? 3 : 0)) >> 2 truncates towards zero, v_med3 clips
LSN_HD int lsn_ext_scale(int x)
{
  const int t = 3 * x;
  const int r = (t + (int)((uint32_t)t >> 30)) >> 2;  // |t| < 2^30: the two top bits are 11 exactly when t < 0
  return r < -LSN_EXT_CLIP ? -LSN_EXT_CLIP : (r > LSN_EXT_CLIP ? LSN_EXT_CLIP : r);
}

// forward step, layout C in place; q = (lsa | lp << 16): the four branch metrics are 0, g01 = lp, g10 = lsa, g11 = lsa + lp
template <bool NORM>
LSN_HD void lsn_step_fwd_pk(s2* a, s2 q)
{
  const s2 gg = pka_sel<0, 1, 1, 0>(q, q);                      // (g11, g11)
  const s2 p = pk_s2(pk_u32(gg) & 0xFFFF0000u);                 // (0, g11)
  const s2 x0 = pka_sel<0, 0, 0, 1>(a[0], p), y0 = pka_sel<0, 0, 1, 0>(a[1], p);  // -> states 0, 4 from 0 (+ 0, g11), 1 (+ g11, 0)
  const s2 x1 = pka_sel<0, 0, 0, 1>(a[2], q), y1 = pka_sel<0, 0, 1, 0>(a[3], q);  // -> 1, 5 from 2 (+ g10, g01), 3 (+ g01, g10)
  const s2 x2 = pka_sel<1, 1, 1, 0>(a[0], q), y2 = pka_sel<1, 1, 0, 1>(a[1], q);  // -> 2, 6 from 4 (+ g01, g10), 5 (+ g10, g01)
  const s2 x3 = pka_sel<1, 1, 1, 0>(a[2], p), y3 = pka_sel<1, 1, 0, 1>(a[3], p);  // -> 3, 7 from 6 (+ g11, 0), 7 (+ 0, g11)
  a[0] = pkmax(x0, y0); a[1] = pkmax(x1, y1); a[2] = pkmax(x2, y2); a[3] = pkmax(x3, y3);
  if (NORM) {
    const s2 n = a[0];
    a[0] = pks_sel<0, 1, 0, 0>(a[0], n); a[1] = pks_sel<0, 1, 0, 0>(a[1], n); a[2] = pks_sel<0, 1, 0, 0>(a[2], n); a[3] = pks_sel<0, 1, 0, 0>(a[3], n);
  }
}

// backward step: b (layout C, normalised) becomes the beta vector one step earlier (normalised); A = the alphas in front of this step (layout C,
// normalised); returns L = max over branches with input 1 - max over branches with input 0
LSN_HD int lsn_step_bwd_pk(s2* b, const s2* A, s2 q)
{
  const s2 G0 = __builtin_shufflevector(b[0], b[2], 0, 2), G1 = __builtin_shufflevector(b[0], b[2], 1, 3);  // (b0, b2) (b4, b6)
  const s2 G2 = __builtin_shufflevector(b[1], b[3], 0, 2), G3 = __builtin_shufflevector(b[1], b[3], 1, 3);  // (b1, b3) (b5, b7)
  const s2 gg = pka_sel<0, 1, 1, 0>(q, q);
  const s2 S = pk_s2(pk_u32(q) & 0xFFFF0000u);   // (0, g01)
  const s2 T = s2{gg.x, q.x};                    // (g11, g10)
  // successor metric + branch metric, paired like alpha: (state k, state k + 4); input 0 and input 1
  const s2 u00 = pka(G0, S), u01 = pka(G1, S), u02 = pka_sel<0, 1, 1, 0>(G3, S), u03 = pka_sel<0, 1, 1, 0>(G2, S);
  const s2 u10 = pka(G1, T), u11 = pka(G0, T), u12 = pka_sel<0, 1, 1, 0>(G2, T), u13 = pka_sel<0, 1, 1, 0>(G3, T);
  const s2 M0 = pkmax(pkmax(pka(A[0], u00), pka(A[1], u01)), pkmax(pka(A[2], u02), pka(A[3], u03)));
  const s2 M1 = pkmax(pkmax(pka(A[0], u10), pka(A[1], u11)), pkmax(pka(A[2], u12), pka(A[3], u13)));
  const s2 m0 = pkmax(M0, M0.yx), m1 = pkmax(M1, M1.yx);
  b[0] = pkmax(u00, u10); b[1] = pkmax(u01, u11); b[2] = pkmax(u02, u12); b[3] = pkmax(u03, u13);
  const s2 n = b[0];
  b[0] = pks_sel<0, 1, 0, 0>(b[0], n); b[1] = pks_sel<0, 1, 0, 0>(b[1], n); b[2] = pks_sel<0, 1, 0, 0>(b[2], n); b[3] = pks_sel<0, 1, 0, 0>(b[3], n);
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
  uint8_t* ckpt;   // check-point slots of 14 * nt bytes each: [3][nt] words (states 1|5, 2|6, 3|7) + [nt] halves (state 4); state 0 is 0
  // A workgroup that decodes TWO blocks (k_turbo: one wavefront each) keeps the three pointers the same for both - compile-time LDS offsets in every
  // access, as in a one-block workgroup - and moves the second block through its INDICES: `bias` is added to every spp / ext index (the arrays of the two
  // blocks lie side by side: spp0 spp1 ext0 ext1, kmax + 8 entries each), cw / ch to the lane index of the check-point words / halves (ckpt0 ckpt1).  A
  // run-time LDS base per wavefront instead cost 60 more spilled registers (round 5).
  int bias = 0, cw = 0, ch = 0;
};
// nt = threads that work on the block (64, or 128 for the blocks with more than 64 windows)
LSN_HD void lsn_ckpt_store(uint8_t* area, int nt, int slot, int lane, const s2* a, int cw = 0, int ch = 0)
{
  uint32_t* w = (uint32_t*)(area + (size_t)(slot * 14 * nt));
  const int lw = lane + cw;
  w[lw] = pk_u32(a[1]); w[nt + lw] = pk_u32(a[2]); w[2 * nt + lw] = pk_u32(a[3]);
  ((int16_t*)(w + 3 * nt))[lane + ch] = a[0].y;
}
LSN_HD void lsn_ckpt_load(const uint8_t* area, int nt, int slot, int lane, s2* a, int cw = 0, int ch = 0)
{
  const uint32_t* w = (const uint32_t*)(area + (size_t)(slot * 14 * nt));
  const int lw = lane + cw;
  a[1] = pk_s2(w[lw]); a[2] = pk_s2(w[nt + lw]); a[3] = pk_s2(w[2 * nt + lw]);
  a[0] = s2{0, ((const int16_t*)(w + 3 * nt))[lane + ch]};
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
// check-point slots: sub-blocks 1 .. nsb-2.  64 working threads: W <= 95 -> 6 sub-blocks, 4 slots; 128: W <= 52 -> 4 sub-blocks, 2 slots - 3584 bytes
// either way.  The area also carries the window-boundary exchange (2 slots) and, before the first iteration, the 12 termination values
// (48 bytes at offset 2048).
#define TB_CKPT_BYTES ((size_t)3584)
static_assert(TB_S == 16, "the check-point area is sized for sub-blocks of 16 steps");

// interleaver address table of one block size, two trellis steps per word: dst[(t / 2) * P + w] = a(t) | a(t + 1) << 16 for even t, with
// a(t) = transposed address of pi(w * W + t), pi(x) = (f1 x + f2 x^2) mod K (36.212 5.1.3.2.3); lsn_turbo_il_words(K) words (lsn_rm.h)
LSN_HD void lsn_turbo_il_fill(uint32_t* dst, int K, int f1, int f2)
{
  const int P = lsn_turbo_nwin(K), W = K / P;
  for (int i = 0; i < ((W + 1) / 2) * P; i++) dst[i] = 0;
  for (int w = 0; w < P; w++)
    for (int t = 0; t < W; t++) {
      const long long x = (long long)w * W + t;
      const int pi = (int)(((long long)f1 * x + (long long)f2 * x % K * x) % K);
      dst[(t >> 1) * P + w] |= (uint32_t)((pi % W) * P + pi / W) << (16 * (t & 1));
    }
}

// One constituent decoder, the part of one lane (= window `wl`; idle lanes shadow window 0 and write their soft output to the spare slot
// ext[K]).  nii_a / nii_b: boundary metrics of the previous iteration (layout C); beta_tail: termination metrics (layout C);
// a_end / b_out: this window's metrics at its end / start, for the exchange between the lanes (the caller's business).
// il (second decoder only): il[(t / 2) * P + w] = transposed LDS addresses of the positions the QPP interleaver gives steps t, t + 1 of
// window w, 16 bits each (lsn_turbo_il_fill above; one table per block size, 1.1 MB for all 188 sizes, L2 resident).  The addresses of a sub-block are fetched one
// sub-block ahead of their use: the L2 latency hides behind the recursion of the sub-block in hand.  (Rounds 1-2 stepped the QPP recursion
// pi += g, g += 2 f2 per lane and divided by W with a multiply: 14 instructions per step and direction instead of one load.)
template <bool IL>
LSN_HD void lsn_map_pass_lane(const TurboLds& m, const uint32_t* il, int nt, int lane, bool active, int K, int P, int W,
                              const s2* nii_a, const s2* nii_b, const s2* beta_tail, s2* a_end, s2* b_out)
{
  const int wl = active ? lane : 0;
  const int wlb = wl + m.bias;   // index of this window's column in the (possibly second) block's arrays
  const uint32_t bias2 = (uint32_t)m.bias * 0x10001u;  // (addresses stay below 2^16: K + bias <= 2 * 2760)
  const int nsb = (W + TB_S - 1) / TB_S;
  s2 a[4], b[4], a0[4];
  if (wl == 0) {
    a[0] = s2{0, (short)LSN_NEG_METRIC};
    a[1] = a[2] = a[3] = s2{(short)LSN_NEG_METRIC, (short)LSN_NEG_METRIC};
  } else {
    for (int k = 0; k < 4; k++) a[k] = nii_a[k];
  }
  for (int k = 0; k < 4; k++) a0[k] = a[k];
  uint32_t nx[TB_S / 2];   // interleaver addresses (two steps per word) of the sub-block that comes next
  uint32_t cur[TB_S / 2];  // ... of the sub-block in hand
  const int wlast = (W - 1) >> 1;
  auto il_load = [&](int sb) {
#pragma unroll
    for (int u = 0; u < TB_S / 2; u++) {
      int t2 = sb * (TB_S / 2) + u;
      t2 = t2 < wlast ? t2 : wlast;
      nx[u] = (il + (uint32_t)(t2 * P))[wl] + bias2;  // uniform row address + lane offset; both 16-bit addresses of the word move by the block's index bias
    }
  };
  if (IL) il_load(0);
  uint32_t g[TB_S];  // operands of one sub-block: lsa (low half) | lp << 16
  // ---- forward sweep over sub-blocks 0 .. nsb-2 (the last one is covered by the recompute below) ----
  for (int sb = 0; sb + 1 < nsb; sb++) {
    if (sb >= 1) lsn_ckpt_store(m.ckpt, nt, sb - 1, lane, a, m.cw, m.ch);
    const int tb = sb * TB_S;
    if (IL) {
#pragma unroll
      for (int u = 0; u < TB_S / 2; u++) cur[u] = nx[u];
      il_load(sb + 1);
    }
#pragma unroll
    for (int u = 0; u < TB_S; u++) {
      const int nat = (tb + u) * P + wlb;
      if (IL) {
        const int idx = (int)((u & 1) ? cur[u >> 1] >> 16 : cur[u >> 1] & 0xFFFFu);
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
        lsn_ckpt_load(m.ckpt, nt, sb - 1, lane, a, m.cw, m.ch);
      }
    }
    if (IL) {
#pragma unroll
      for (int u = 0; u < TB_S / 2; u++) cur[u] = nx[u];
      if (sb > 0) il_load(sb - 1);
    }
    // only the last sub-block of a window can be shorter than TB_S: the full-length variant carries no per-step guards
    auto subblock = [&](auto fullc) {
      constexpr bool FULL = decltype(fullc)::value;
      // operand burst
#pragma unroll
      for (int u = TB_S - 1; u >= 0; u--) {
        if (FULL || u < n) {
          const int nat = (tb + u) * P + wlb;
          if (IL) {
            const int idx = (int)((u & 1) ? cur[u >> 1] >> 16 : cur[u >> 1] & 0xFFFFu);
            ix[u] = active ? idx : K + m.bias;
            g[u] = ((uint32_t)(fld0(m.spp[idx]) + ((int)m.ext[idx] >> 1)) & 0xFFFFu) | ((uint32_t)fld2(m.spp[nat]) << 16);
          } else {
            const uint32_t w = m.spp[nat];
            ix[u] = active ? nat : K + m.bias;
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
          m.ext[ix[u]] = (int16_t)(lsn_ext_scale(L - (int)q.x) * 2 + hard);  // idle lanes: spare slot ext[K]
        }
      }
    };
    if (n == TB_S) subblock(std::true_type{}); else subblock(std::false_type{});
  }
  for (int k = 0; k < 4; k++) b_out[k] = b[k];
}
