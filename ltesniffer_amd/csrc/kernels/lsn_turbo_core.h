// lsn_turbo_core.h - the per-lane arithmetic of the turbo decoder kernel (stage_c.hip: k_turbo), written so that the SAME text compiles for
// gfx950 and for the host: tests/native/test_turbo_core.cc runs it lane by lane over a plain-memory stand-in of the LDS and compares every
// decision with the CPU oracle's decoder - the packed arithmetic below is checked without a GPU (and, in that build, every packed add is
// range-checked against 32-bit arithmetic).
//
// One lane = one trellis window (see stage_c.hip for the schedule).  Round 3, second half: the recursions run on PACKED int16 pairs.
// A wavefront alone on a SIMD issues one VALU instruction per 4 cycles whatever its class (profiles/r03_valu_peak_isa.txt),
// and packed forms occupy the VALU port for 4 cycles at any occupancy, so v_pk_add_i16 / v_pk_max_i16 do two state updates for the price of one.
// Vector instructions per trellis step in the full-length sub-blocks (counted in the ISA; first / interleaved constituent decoder):
//                                  rounds 3-4      round 5 (second half)   round 6 (last session)
//   forward sweep                  25.4 / 27.8     23.4 / 26.8             23.4 / 26.8
//   alpha recompute                27.3 / 28.3     23.3 / 25.3             23.6 / 25.6
//   beta + soft output             50.9 / 49.9     44.9 / 44.5             37.7 / 37.3
//   per step of a 64-step window   (forward sweep over 48 of the 64 steps)  85.8 / 89.9  ->  78.9 / 83.0
//   round 5: operands of two steps built together (lsn_operands2), alpha normalised every second step in the recompute too, soft outputs of two steps on packed
//   halves (lsn_ext_two), (g10, g11) with one add, lanes without a window masked instead of redirected per step;
//   round 6: no shuffles inside a full-length sub-block (the cycle of seven register layouts, lsn_turbo_cyc.h: - 3.75 per step), beta normalised behind every
//   second step (- 2), the sign mask of the soft output from one shift (- 0.5).
//
// Layout C: four registers hold the eight state metrics as (m[k] | m[k+4] << 16), k = 0..3.
//  * forward: butterfly k reads states 2k, 2k+1 and writes k, k+4 - with the operands taken as half-broadcasts (op_sel) layout C maps to
//    itself: 8 v_pk_add + 4 v_pk_max per step.
//  * backward: the pairs that meet alpha (k, k+4) need beta pairs (0,2) (4,6) (1,3) (5,7): four v_perm per step, then 8 v_pk_add
//    (branch + beta), 8 v_pk_add (+ alpha), 6 + 4 v_pk_max; the result is layout C again.  (The short last sub-block of a window; the full-length ones
//    walk through the seven layouts of lsn_turbo_cyc.h, in which neither direction shuffles.)
// Word length (int16, no wrap-around anywhere - v_pk_max_i16 compares values, not residues):
//   |gamma| <= |sys| + |ext| + |par| <= 512 + 2047 + 512 = 3071 per step.  Any state is reached from any state in 3 steps, so a metric
//   vector normalised to state 0 spreads at most 3 * 3071 = 9213 once it is 3 steps away from its initialisation; the initialisations are
//   (0, -12000 x 7), the termination metrics (<= 3 * 1022) and boundary metrics of the previous iteration (<= 9213).
//   Alpha is normalised behind every ODD step (forward sweep and recompute; sub-blocks start at even steps).  In front of an even step t it is normalised:
//   <= 9213 at t = 0 (boundary metrics), <= 15355 at t = 2 (two steps of growth), <= 9213 from t = 4 on; in front of an odd step it carries the growth of
//   one more step: <= 12284 at t = 1, <= 18426 at t = 3, <= 12284 from t = 5 on.  The first window starts from (0, -12000 x 7) instead: 12000 at t = 0,
//   <= 15071 at t = 1 and <= 18142 at t = 2 - on the NEGATIVE side only (the states that cannot be reached yet still carry the initial -12000, the reached
//   ones are within 2 * 3071) -, and as every other window from t = 3 on.
//   Beta (round 6, last session) is normalised behind every EVEN step of a full-length sub-block and behind every step of the short last one.  Let N(t) bound
//   the NORMALISED vector that enters step t: N(W-1) = 9213 (the initialisation), N(W-2) = 12284, N(W-3) = 15355, N(t) = 9213 below (three steps away).
//   A vector that enters an odd step is normalised (<= N(t)); one that enters an even step is at worst the raw result of the odd step behind it,
//   <= N(t+1) + 3071 (<= 12284 in general, 18426 at t = W-4, 15355 at t = W-3, 12284 at t = W-2).  A window has at least 32 steps, so the young ends never meet:
//     odd t:   |alpha| + |beta| + |gamma| <= max(12284 + 15355, 18426 + 9213) + 3071 = 30710
//     even t:  <= max(9213 + 18426, 15355 + 12284) + 3071 = 30710;  first window, t = 0: 12000 + 12284 + 3071;  t = 2: 18142 + 12284 + 3071 = 33497 - beyond int16, but on the
//              negative side only and only in sums over states that cannot be reached yet.  The alpha + (beta + gamma) adds SATURATE (pka_sat: v_pk_add_i16 clamp):
//              a saturated sum is below -32768 in exact arithmetic while the maximum it competes in is at least the state-0 term >= -(12284 + 3071), so it
//              loses either way and every maximum equals the exact one.  beta + gamma alone stays within 18426 + 3071.
//   (tools/turbo_metric_ranges.py measured <= 19920 with both recursions normalised every step; the host build of this file - tests/native/test_turbo_core.cc -
//   checks every packed add and subtract, and aborts when a saturating add leaves the range on the positive side.)  Check-points and the window-boundary
//   exchange store normalised vectors only (state 0 = 0 is not stored).
//   L = m1 - m0 is formed in 32 bits, or in 16 bits with saturation where that is proven equal (lsn_ext_two).
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
// alpha + (beta + branch): saturating (v_pk_add_i16 clamp - the same issue slot as the wrapping add).  A sum can leave the int16 range on the NEGATIVE side
// only, and only at step 2 of the first window, where the states that cannot be reached yet still carry the initial -12000 (word-length argument above):
// such a sum loses every maximum it takes part in whether it saturates or not.  The host build checks exactly that: a positive overflow aborts.
#ifdef LSN_TURBO_RANGE_CHECK
static inline s2 pka_sat(s2 a, s2 b)
{
  const int x = (int)a.x + (int)b.x, y = (int)a.y + (int)b.y;
  if (x > 32767 || y > 32767) { std::fprintf(stderr, "int16 range exceeded on the positive side in alpha + beta: %d %d\n", x, y); std::abort(); }
  return s2{(short)(x < -32768 ? -32768 : x), (short)(y < -32768 ? -32768 : y)};
}
#elif defined(__HIP_DEVICE_COMPILE__)
LSN_HD s2 pka_sat(s2 a, s2 b) { return __builtin_elementwise_add_sat(a, b); }
#else
static inline s2 pka_sat(s2 a, s2 b)
{
  const int x = (int)a.x + (int)b.x, y = (int)a.y + (int)b.y;
  return s2{(short)(x < -32768 ? -32768 : (x > 32767 ? 32767 : x)), (short)(y < -32768 ? -32768 : (y > 32767 ? 32767 : y))};
}
#endif
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

// backward step: b (layout C) becomes the beta vector one step earlier; A = the alphas in front of this step (layout C);
// M0 / M1 = max over the branches with input 0 / 1 of alpha + branch + beta, still split over the two halves (max(M.x, M.y) is the maximum).
// NORM: the new vector is normalised to its state 0.  The full-length sub-blocks normalise behind the EVEN steps only (round 6, last session; as the alpha
// recursion does behind the odd ones): a beta vector that enters an odd step is normalised, one that enters an even step carries the growth of one step - and
// meets an alpha vector that is normalised (word-length argument at the top).
template <bool NORM = true>
LSN_HD void lsn_step_bwd_pk(s2* b, const s2* A, s2 q, s2* M0o, s2* M1o)
{
  const s2 G0 = __builtin_shufflevector(b[0], b[2], 0, 2), G1 = __builtin_shufflevector(b[0], b[2], 1, 3);  // (b0, b2) (b4, b6)
  const s2 G2 = __builtin_shufflevector(b[1], b[3], 0, 2), G3 = __builtin_shufflevector(b[1], b[3], 1, 3);  // (b1, b3) (b5, b7)
  const s2 S = pk_s2(pk_u32(q) & 0xFFFF0000u);   // (0, g01)
  const s2 T = pka_sel<0, 0, 0, 1>(q, S);        // (g10, g11): one add (round 5; rounds 3-4 built (g11, g10) with an add and a v_perm)
  // successor metric + branch metric, paired like alpha: (state k, state k + 4); input 0 and input 1
  const s2 u00 = pka(G0, S), u01 = pka(G1, S), u02 = pka_sel<0, 1, 1, 0>(G3, S), u03 = pka_sel<0, 1, 1, 0>(G2, S);
  const s2 u10 = pka_sel<0, 1, 1, 0>(G1, T), u11 = pka_sel<0, 1, 1, 0>(G0, T), u12 = pka(G2, T), u13 = pka(G3, T);
  *M0o = pkmax(pkmax(pka_sat(A[0], u00), pka_sat(A[1], u01)), pkmax(pka_sat(A[2], u02), pka_sat(A[3], u03)));
  *M1o = pkmax(pkmax(pka_sat(A[0], u10), pka_sat(A[1], u11)), pkmax(pka_sat(A[2], u12), pka_sat(A[3], u13)));
  b[0] = pkmax(u00, u10); b[1] = pkmax(u01, u11); b[2] = pkmax(u02, u12); b[3] = pkmax(u03, u13);
  if (NORM) {
    const s2 n = b[0];
    b[0] = pks_sel<0, 1, 0, 0>(b[0], n); b[1] = pks_sel<0, 1, 0, 0>(b[1], n); b[2] = pks_sel<0, 1, 0, 0>(b[2], n); b[3] = pks_sel<0, 1, 0, 0>(b[3], n);
  }
}
// soft output of one step: extrinsic * 2 + hard decision (L = m1 - m0 in 32 bits)
LSN_HD int16_t lsn_ext_one(s2 M0, s2 M1, s2 q)
{
  const s2 m0 = pkmax(M0, M0.yx), m1 = pkmax(M1, M1.yx);
  const int L = (int)m1.x - (int)m0.x;
  const int hard = L < 0 ? 0 : (L > 1 ? 1 : L);  // v_med3_i32(L, 0, 1)
  return (int16_t)(lsn_ext_scale(L - (int)q.x) * 2 + hard);
}
// ... of TWO steps at once on packed halves (step a in the low half, step b in the high half): 8 instead of 12 vector instructions per step.
// Equal to lsn_ext_one although L is formed in 16 bits WITH SATURATION (v_pk_sub_i16 clamp): the extrinsic value is
// sign(x) min(floor(3 |x| / 4), 2047) with x = L - lsa, |lsa| <= 2558, and every |x| >= 2730 gives 2047 (3 * 2730 / 4 = 2047.5) - so x may be clamped to
// +- 2730 first, a saturated L (|L| >= 32767) keeps |L - lsa| >= 30209 on the same side, and 3 * 2730 fits 16 bits; the hard decision is the
// sign of L, which saturation keeps.
LSN_HD s2 lsn_sat_sub(s2 a, s2 b)
{
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_elementwise_sub_sat(a, b);
#else
  const int x = (int)a.x - (int)b.x, y = (int)a.y - (int)b.y;
  return s2{(short)(x < -32768 ? -32768 : (x > 32767 ? 32767 : x)), (short)(y < -32768 ? -32768 : (y > 32767 ? 32767 : y))};
#endif
}
LSN_HD s2 lsn_ext_two(s2 M0a, s2 M1a, s2 qa, s2 M0b, s2 M1b, s2 qb)
{
#ifdef __HIP_DEVICE_COMPILE__
  // Stated instruction by instruction, in ONE asm block.  Left to itself the compiler turns the sign mask and the hard decision into compare / select
  // pairs per half and re-packs them with v_perm (21 instructions instead of 15); the maxima over the two halves of M land directly in the half of
  // their step (SDWA).  gfx940-class parts need a wait state between a VALU write with dst_sel != DWORD and a VALU read of that register, and the
  // compiler's hazard recogniser does not look inside inline asm: the order below keeps one instruction between every such pair.
  s2 r0, r1, r2, r3, out;
  asm("v_max_i16_sdwa %0, %5, %5 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"         // r0 = m0 (step a)
      "v_max_i16_sdwa %1, %6, %6 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"         // r1 = m1 (step a)
      "v_max_i16_sdwa %0, %7, %7 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1\n\t"    //      m0 (step b) into the high half
      "v_max_i16_sdwa %1, %8, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1\n\t"
      "v_perm_b32 %3, %10, %9, %11\n\t"                 // r3 = (lsa a, lsa b)
      "v_pk_sub_i16 %2, %1, %0 clamp\n\t"               // r2 = L = m1 - m0, saturated
      "v_pk_sub_i16 %3, %2, %3 clamp\n\t"               // r3 = x = L - lsa
      "v_pk_max_i16 %3, %3, %12\n\t"                    //      clamped to +- 2730
      "v_pk_min_i16 %3, %3, %13\n\t"
      "v_pk_lshrrev_b16 %0, 14, %3 op_sel_hi:[0,1]\n\t" // r0 = 3 where x < 0, else 0: the two top bits of a value in +- 2730 are its sign
      "v_pk_mad_u16 %0, %3, 3, %0 op_sel_hi:[1,0,1]\n\t" // r0 = t = 3 x + (x < 0 ? 3 : 0);  t >> 2 truncates 3 x / 4 towards zero
      "v_pk_max_i16 %1, %2, 0\n\t"                      // r1 = hard decision: min(max(L, 0), 1)
      "v_pk_min_i16 %1, %1, 1 op_sel_hi:[1,0]\n\t"
      "v_pk_ashrrev_i16 %0, 1, %0 op_sel_hi:[0,1]\n\t"  // (t >> 2) * 2 = (t >> 1) & ~1
      "v_and_or_b32 %4, %0, %14, %1"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(out)
      : "v"(M0a), "v"(M1a), "v"(M0b), "v"(M1b), "v"(qa), "v"(qb), "s"(0x05040100u), "s"(0xF556F556u), "s"(0x0AAA0AAAu), "s"(0xFFFEFFFEu));
  return out;
#else
  const s2 m0a = pkmax(M0a, M0a.yx), m1a = pkmax(M1a, M1a.yx), m0b = pkmax(M0b, M0b.yx), m1b = pkmax(M1b, M1b.yx);
  const s2 m0 = s2{m0a.x, m0b.x}, m1 = s2{m1a.x, m1b.x};
  const s2 L = lsn_sat_sub(m1, m0);
  s2 x = lsn_sat_sub(L, s2{qa.x, qb.x});
  const s2 lim = s2{2730, 2730};
  x = __builtin_elementwise_min(pkmax(x, -lim), lim);
  const s2 neg = x >> 15;                                 // -1 where x < 0
  const s2 r = (x * (short)3 + (neg & (short)3)) >> 2;     // truncation towards zero, as lsn_ext_scale
  const s2 hard = __builtin_elementwise_min(pkmax(L, s2{0, 0}), s2{1, 1});
  return r * (short)2 + hard;
#endif
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

// Branch-metric operands of two consecutive trellis steps, (lsa | lp << 16) each: lsa = systematic field of ws + extrinsic value (es = ext * 2 + hard bit,
// 16 bits as loaded), lp = the 10-bit field at bit PF of wp.  The sums of both steps are formed in one packed register (v_perm to pair the halves, packed
// shifts and add), each parity field costs one v_bfe_i32 and one v_perm that also picks the step's half of the sums: 10 vector instructions per two
// steps (rounds 3-4: 14 - shift / shift / and / or per parity field).
LSN_HD uint32_t lsn_perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_amdgcn_perm(hi, lo, sel);
#else
  const uint64_t v = ((uint64_t)hi << 32) | lo;
  uint32_t d = 0;
  for (int k = 0; k < 4; k++) d |= (uint32_t)((v >> (8 * ((sel >> (8 * k)) & 7u))) & 0xFFu) << (8 * k);
  return d;
#endif
}
template <int PF>
LSN_HD int lsn_par_field(uint32_t w)
{
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_amdgcn_sbfe((int)w, PF, 10);
#else
  return (int)(w << (22 - PF)) >> 22;
#endif
}
template <int PF>
LSN_HD void lsn_operands2(uint32_t ws0, uint32_t ws1, uint32_t es0, uint32_t es1, uint32_t wp0, uint32_t wp1, uint32_t* g0, uint32_t* g1)
{
  const s2 sys = (pk_s2(lsn_perm(ws1, ws0, 0x05040100u)) << 6) >> 6;
  const s2 apr = pk_s2(lsn_perm(es1, es0, 0x05040100u)) >> 1;
  const uint32_t sum = pk_u32(pka(sys, apr));
  *g0 = lsn_perm((uint32_t)lsn_par_field<PF>(wp0), sum, 0x05040100u);
  *g1 = lsn_perm((uint32_t)lsn_par_field<PF>(wp1), sum, 0x05040302u);
}

// the same steps over a cycle of seven register layouts (no shuffles inside a full-length sub-block): generated, see tools/turbo_layouts.py
#include "lsn_turbo_cyc.h"

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

// One constituent decoder, the part of one lane (= window `wl`; only lanes with a window run it - the caller masks the others, they meet the working
// lanes again at the caller's barriers).  nii_a / nii_b: boundary metrics of the previous iteration (layout C); beta_tail: termination metrics (layout C);
// a_end / b_out: this window's metrics at its end / start, for the exchange between the lanes (the caller's business).
// il (second decoder only): il[(t / 2) * P + w] = transposed LDS addresses of the positions the QPP interleaver gives steps t, t + 1 of
// window w, 16 bits each (lsn_turbo_il_fill above; one table per block size, 1.1 MB for all 188 sizes, L2 resident).  The addresses of a sub-block are fetched one
// sub-block ahead of their use: the L2 latency hides behind the recursion of the sub-block in hand.  (Rounds 1-2 stepped the QPP recursion
// pi += g, g += 2 f2 per lane and divided by W with a multiply: 14 instructions per step and direction instead of one load.)
template <bool IL>
LSN_HD void lsn_map_pass_lane(const TurboLds& m, const uint32_t* il, int nt, int wl, int K, int P, int W,
                              const s2* nii_a, const s2* nii_b, const s2* beta_tail, s2* a_end, s2* b_out)
{
  const int lane = wl;
  int wlb = wl + m.bias;   // index of this window's column in the (possibly second) block's arrays
  // The LDS addresses of a sub-block are lane + a multiple of P.  The sub-block the backward pass starts with is the same in every pass, and the compiler computes
  // its sixteen-odd address vectors once per kernel and keeps them alive across all iterations - beyond the 256 registers: 14 of them went to scratch memory and
  // came back once per pass (5 MB of scratch write-back per launch).  Declaring the lane term opaque at the top of every sub-block keeps the addresses local
  // to their sub-block (a few adds per sub-block instead of a spill).
#ifdef __HIP_DEVICE_COMPILE__
#define LSN_LOCAL_ADDR(x) asm volatile("" : "+v"(x))
#else
#define LSN_LOCAL_ADDR(x) (void)(x)
#endif
  const uint32_t bias2 = (uint32_t)m.bias * 0x10001u;  // (addresses stay below 2^16: K + bias <= 2 * 2760)
  const int nsb = (W + TB_S - 1) / TB_S;
  s2 a[4], b[4];
  // the window's first alpha vector: wanted again when the backward pass reaches sub-block 0 (read from the caller's registers a second time rather than kept)
  auto a_start = [&]() {
    if (wl == 0) {
      a[0] = s2{0, (short)LSN_NEG_METRIC};
      a[1] = a[2] = a[3] = s2{(short)LSN_NEG_METRIC, (short)LSN_NEG_METRIC};
    } else {
      for (int k = 0; k < 4; k++) a[k] = nii_a[k];
    }
  };
  a_start();
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
  // operands of steps tb + u, tb + u + 1 (u even); a step past the end of the window (odd W, last pair) reads the spare entries behind the block
  auto build2 = [&](int tb, int u) {
    const int nat0 = (tb + u) * P + wlb, nat1 = nat0 + P;
    if (IL) {
      const int i0 = (int)(cur[u >> 1] & 0xFFFFu), i1 = (int)(cur[u >> 1] >> 16);
      lsn_operands2<20>(m.spp[i0], m.spp[i1], (uint16_t)m.ext[i0], (uint16_t)m.ext[i1], m.spp[nat0], m.spp[nat1], &g[u], &g[u + 1]);
    } else {
      const uint32_t w0 = m.spp[nat0], w1 = m.spp[nat1];
      lsn_operands2<10>(w0, w1, (uint16_t)m.ext[nat0], (uint16_t)m.ext[nat1], w0, w1, &g[u], &g[u + 1]);
    }
  };
  // ---- forward sweep over sub-blocks 0 .. nsb-2 (the last one is covered by the recompute below) ----
  for (int sb = 0; sb + 1 < nsb; sb++) {
    LSN_LOCAL_ADDR(wlb);
    if (sb >= 1) lsn_ckpt_store(m.ckpt, nt, sb - 1, lane, a, m.cw, m.ch);
    const int tb = sb * TB_S;
    if (IL) {
#pragma unroll
      for (int u = 0; u < TB_S / 2; u++) cur[u] = nx[u];
      il_load(sb + 1);
    }
#pragma unroll
    for (int u = 0; u < TB_S; u += 2) build2(tb, u);
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
  s2 A[TB_S][4];    // alphas in front of each step of the sub-block
  // LDS index of the systematic / extrinsic value of step tb + u
  auto ext_index = [&](int tb, int u) { return IL ? (int)((u & 1) ? cur[u >> 1] >> 16 : cur[u >> 1] & 0xFFFFu) : (tb + u) * P + wlb; };
  for (int sb = nsb - 1; sb >= 0; sb--) {
    LSN_LOCAL_ADDR(wlb);
    const int tb = sb * TB_S, n = (tb + TB_S < W) ? TB_S : W - tb;
    if (sb + 1 < nsb) {
      if (sb == 0) {
        a_start();
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
      for (int u = TB_S - 2; u >= 0; u -= 2)
        if (FULL || u < n) build2(tb, u);
      // recompute the alphas of this sub-block into registers.  Normalised every SECOND step (as in the forward sweep): the alphas in front of the
      // odd steps carry the growth of one step (word-length argument at the top).
      // Full-length sub-block (round 6, last session): the alphas in front of step u are kept in register layout u mod 7 (lsn_turbo_cyc.h) - the forward step
      // from layout L to L + 1 and the backward step from L + 1 to L need no shuffle -, the beta vector enters through one conversion from layout C and
      // leaves step 0 in layout C again, normalised (check-points, window boundaries and the short last sub-block stay in layout C).
      if constexpr (FULL) {
        auto fwd1 = [&](auto uc) {
          constexpr int u = decltype(uc)::value;
          A[u][0] = a[0]; A[u][1] = a[1]; A[u][2] = a[2]; A[u][3] = a[3];
          lsn_cyc_fwd<u % 7>(a, pk_s2(g[u]));
          if (u & 1) {
            const s2 n0 = a[0];
            a[0] = pks_sel<0, 1, 0, 0>(a[0], n0); a[1] = pks_sel<0, 1, 0, 0>(a[1], n0); a[2] = pks_sel<0, 1, 0, 0>(a[2], n0); a[3] = pks_sel<0, 1, 0, 0>(a[3], n0);
          }
        };
#define LSN_IC(n) std::integral_constant<int, n>{}
        fwd1(LSN_IC(0)); fwd1(LSN_IC(1)); fwd1(LSN_IC(2)); fwd1(LSN_IC(3)); fwd1(LSN_IC(4)); fwd1(LSN_IC(5)); fwd1(LSN_IC(6)); fwd1(LSN_IC(7));
        fwd1(LSN_IC(8)); fwd1(LSN_IC(9)); fwd1(LSN_IC(10)); fwd1(LSN_IC(11)); fwd1(LSN_IC(12)); fwd1(LSN_IC(13)); fwd1(LSN_IC(14)); fwd1(LSN_IC(15));
        if (sb == nsb - 1) {
          // (the caller wants a normalised vector in layout C: the next iteration starts a window from it)
          lsn_cyc_to_c<TB_S % 7>(a);
          const s2 nn = a[0];
          for (int k = 0; k < 4; k++) a_end[k] = pks_sel<0, 1, 0, 0>(a[k], nn);
        }
        // beta recursion + LLR + extrinsic, two steps at a time: normalised behind the even step only
        lsn_cyc_from_c<TB_S % 7>(b);
        auto bwd2 = [&](auto uc) {
          constexpr int u = decltype(uc)::value;   // odd
          s2 M0a, M1a, M0b, M1b;
          lsn_cyc_bwd<u % 7, false>(b, A[u], pk_s2(g[u]), &M0a, &M1a);
          lsn_cyc_bwd<(u - 1) % 7, true>(b, A[u - 1], pk_s2(g[u - 1]), &M0b, &M1b);
          const s2 e = lsn_ext_two(M0a, M1a, pk_s2(g[u]), M0b, M1b, pk_s2(g[u - 1]));
          m.ext[ext_index(tb, u)] = e.x;
          m.ext[ext_index(tb, u - 1)] = e.y;
        };
        bwd2(LSN_IC(15)); bwd2(LSN_IC(13)); bwd2(LSN_IC(11)); bwd2(LSN_IC(9)); bwd2(LSN_IC(7)); bwd2(LSN_IC(5)); bwd2(LSN_IC(3)); bwd2(LSN_IC(1));
#undef LSN_IC
      } else {
#pragma unroll
        for (int u = 0; u < TB_S; u++) {
          if (u < n) {
            A[u][0] = a[0]; A[u][1] = a[1]; A[u][2] = a[2]; A[u][3] = a[3];
            if (u & 1) lsn_step_fwd_pk<true>(a, pk_s2(g[u])); else lsn_step_fwd_pk<false>(a, pk_s2(g[u]));
          }
        }
        if (sb == nsb - 1) {
          // (the caller wants a normalised vector: the next iteration starts a window from it)
          const s2 nn = a[0];
          for (int k = 0; k < 4; k++) a_end[k] = pks_sel<0, 1, 0, 0>(a[k], nn);
        }
        // beta recursion + LLR + extrinsic
#pragma unroll
        for (int u = TB_S - 1; u >= 0; u--) {
          if (u < n) {
            s2 M0, M1;
            lsn_step_bwd_pk(b, A[u], pk_s2(g[u]), &M0, &M1);
            m.ext[ext_index(tb, u)] = lsn_ext_one(M0, M1, pk_s2(g[u]));
          }
        }
      }
    };
    if (n == TB_S) subblock(std::true_type{}); else subblock(std::false_type{});
  }
  for (int k = 0; k < 4; k++) b_out[k] = b[k];
}
