#!/usr/bin/env python3
"""Generates ltesniffer_amd/csrc/kernels/lsn_turbo_cyc.h: the packed max-log-MAP steps of the turbo decoder kernel over a CYCLE OF SEVEN register layouts.

The decoder keeps the eight state metrics of a trellis step in four registers of two int16 halves.  With a fixed layout C = (0,4) (1,5) (2,6) (3,7) the forward
step maps C to itself through half-broadcasts, but the backward step needs the successor metrics paired differently and pays four v_perm per step
(lsn_turbo_core.h: lsn_step_bwd_pk).  A packed add takes each of its two halves from either half of ONE source register (VOP3P op_sel), so a backward step from
layout X to layout Y needs no shuffle exactly when, for every pair (p, q) of Y and each input bit i, the successors succ_i(p), succ_i(q) share a register of X.
For the LTE constituent code (36.212 5.1.3.2.1: next = (i ^ s2 ^ s3, s1, s2)) that relation is a permutation of the pairings with one cycle of length seven:
    C -> G -> H -> I -> J -> K -> L -> C      (layout of step t -> layout of step t + 1)
The forward step between two neighbours of the cycle needs no shuffle either.  The full-length sub-blocks of 16 steps start from C (the check-points and the
window-boundary exchange stay in C): the alphas of step u are recomputed into layout u mod 7, the beta vector enters the sub-block through one conversion
C -> layout 2 (four v_perm per 16 steps instead of four per step) and comes out of step 0 in C.

This script derives the layouts, finds the operand selections, checks every generated step against an eight-state reference on random metrics and writes the
header.  Run it from the repo root: python tools/turbo_layouts.py
"""
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(s):
    return (s >> 2) & 1, (s >> 1) & 1, s & 1


def succ(s, i):
    s1, s2, s3 = bits(s)
    a = i ^ s2 ^ s3
    return (a << 2) | (s >> 1)


def gamma_half(s, i):
    """half of S = (0, g01) (i = 0) or T = (g10, g11) (i = 1) that is the branch metric of state s under input i: the parity bit is i ^ s1 ^ s2"""
    s1, s2, _ = bits(s)
    z = i ^ s1 ^ s2
    return z  # i = 0: z = 1 -> g01 (high half of S); i = 1: z = 1 -> g11 (high half of T)


def gamma_val(s, i, lsa, lp):
    return i * lsa + gamma_half(s, i) * lp


def need(Y):
    """pairing the beta vector of step t + 1 must have for the beta vector of step t to come out in pairing Y"""
    X = []
    for (p, q) in Y:
        for i in (0, 1):
            pr = frozenset((succ(p, i), succ(q, i)))
            if pr not in X:
                X.append(pr)
    assert len(X) == 4, X
    return X


def ordered(X):
    regs = [tuple(sorted(pr)) for pr in X]
    regs.sort()
    assert regs[0][0] == 0  # state 0 in the low half of register 0: the normalisation subtracts that half in every layout
    return regs


C = [(0, 4), (1, 5), (2, 6), (3, 7)]
T = [C]
while True:
    nxt = ordered(need(T[-1]))
    if nxt == C:
        break
    T.append(nxt)
assert len(T) == 7, len(T)
NL = 7


def where(layout, s):
    for r, (lo, hi) in enumerate(layout):
        if lo == s:
            return r, 0
        if hi == s:
            return r, 1
    raise KeyError(s)


def bwd_spec(L):
    """beta: layout L+1 -> layout L.  For output register r and input i: (source register, half for the low state, half for the high state, gamma halves)"""
    Y, X = T[L], T[(L + 1) % NL]
    spec = []
    for (p, q) in Y:
        per_i = []
        for i in (0, 1):
            rp, hp = where(X, succ(p, i))
            rq, hq = where(X, succ(q, i))
            assert rp == rq
            per_i.append((rp, hp, hq, gamma_half(p, i), gamma_half(q, i)))
        spec.append(per_i)
    return spec


def fwd_spec(L):
    """alpha: layout L -> layout L+1.  For output register r two candidates: (source register, halves, input bit, gamma halves)"""
    Y, X = T[L], T[(L + 1) % NL]
    spec = []
    for (p2, q2) in X:
        cands = []
        for (ra, (ylo, yhi)) in enumerate(Y):
            for (sp, hp) in ((ylo, 0), (yhi, 1)):
                for (sq, hq) in ((ylo, 0), (yhi, 1)):
                    for i in (0, 1):
                        if succ(sp, i) == p2 and succ(sq, i) == q2:
                            cands.append((ra, hp, hq, i, gamma_half(sp, i), gamma_half(sq, i)))
        assert len(cands) == 2, (L, p2, q2, cands)
        spec.append(cands)
    return spec


def conv_spec(src, dst):
    """dst register r = (half of src register a, half of src register b) -> v_perm selector with src1 = a (bytes 0-3), src0 = b (bytes 4-7)"""
    out = []
    for (lo, hi) in dst:
        ra, ha = where(src, lo)
        rb, hb = where(src, hi)
        sel = (2 * ha) | ((2 * ha + 1) << 8) | ((4 + 2 * hb) << 16) | ((4 + 2 * hb + 1) << 24)
        out.append((ra, rb, sel))
    return out


# ---------------------------------------------------------------------------------------------------------------- check against an eight-state reference
def ref_fwd(al, lsa, lp):
    out = [None] * 8
    for s in range(8):
        for i in (0, 1):
            d = succ(s, i)
            v = al[s] + gamma_val(s, i, lsa, lp)
            out[d] = v if out[d] is None else max(out[d], v)
    return out


def ref_bwd(be, al, lsa, lp):
    nb, M = [None] * 8, [None, None]
    for s in range(8):
        for i in (0, 1):
            u = be[succ(s, i)] + gamma_val(s, i, lsa, lp)
            nb[s] = u if nb[s] is None else max(nb[s], u)
            f = al[s] + u
            M[i] = f if M[i] is None else max(M[i], f)
    return nb, M


def pack(layout, v):
    return [(v[lo], v[hi]) for (lo, hi) in layout]


def check():
    rnd = random.Random(5)
    for L in range(NL):
        fs, bs = fwd_spec(L), bwd_spec(L)
        for _ in range(200):
            al = [rnd.randint(-9000, 9000) for _ in range(8)]
            be = [rnd.randint(-9000, 9000) for _ in range(8)]
            lsa, lp = rnd.randint(-2500, 2500), rnd.randint(-511, 511)
            S, Tt = (0, lp), (lsa, lsa + lp)
            a = pack(T[L], al)
            out = []
            for cands in fs:
                vals = []
                for (ra, hp, hq, i, gp, gq) in cands:
                    G = Tt if i else S
                    vals.append((a[ra][hp] + G[gp], a[ra][hq] + G[gq]))
                out.append((max(vals[0][0], vals[1][0]), max(vals[0][1], vals[1][1])))
            assert out == pack(T[(L + 1) % NL], ref_fwd(al, lsa, lp)), ("fwd", L)
            b = pack(T[(L + 1) % NL], be)
            nb, M = [], [[], []]
            for r, per_i in enumerate(bs):
                u = []
                for i, (rs, hp, hq, gp, gq) in enumerate(per_i):
                    G = Tt if i else S
                    u.append((b[rs][hp] + G[gp], b[rs][hq] + G[gq]))
                    M[i].append((a[r][0] + u[i][0], a[r][1] + u[i][1]))
                nb.append((max(u[0][0], u[1][0]), max(u[0][1], u[1][1])))
            rb, rM = ref_bwd(be, al, lsa, lp)
            assert nb == pack(T[L], rb), ("bwd", L)
            for i in (0, 1):
                assert max(max(x) for x in M[i]) == rM[i], ("M", L, i)
    for L in range(NL):
        v = list(range(100, 108))
        for (src, dst) in ((C, T[L]), (T[L], C)):
            a = pack(src, v)
            got = []
            for (ra, rb, sel) in conv_spec(src, dst):
                by = [a[ra][0], a[ra][1], a[rb][0], a[rb][1]]  # halves as the perm sees them: src1 = a (selector bytes 0-3), src0 = b (4-7)
                got.append((by[(sel & 0xFF) // 2], by[((sel >> 16) & 0xFF) // 2]))
            assert got == pack(dst, v)


# ---------------------------------------------------------------------------------------------------------------- emit
def emit():
    o = []
    w = o.append
    w("// lsn_turbo_cyc.h - GENERATED by tools/turbo_layouts.py (which derives and checks it): do not edit.")
    w("// The packed max-log-MAP steps of lsn_turbo_core.h over a cycle of seven register layouts, so that neither recursion needs a shuffle inside a full-length")
    w("// sub-block.  Layout L, register r = (state in the low half, state in the high half):")
    for L in range(NL):
        w("//   %d: %s" % (L, "  ".join("(%d,%d)" % p for p in T[L])))
    w("// alpha: lsn_cyc_fwd<L> takes layout L to layout L + 1 (mod 7);  beta: lsn_cyc_bwd<L> takes layout L + 1 to layout L and meets the alphas in layout L.")
    w("// S = (0, g01), T = (g10, g11) are the branch metrics of the step (lsn_cyc_st).  State 0 sits in the low half of register 0 in every layout.")
    w("#pragma once")
    w("#ifdef __HIPCC__")
    w("#define LSN_HDS __host__ __device__ __forceinline__")
    w("#else")
    w("#define LSN_HDS inline   // (an explicit specialisation takes no storage class)")
    w("#endif")
    w("LSN_HD void lsn_cyc_st(s2 q, s2* S, s2* T)")
    w("{")
    w("  *S = pk_s2(pk_u32(q) & 0xFFFF0000u);")
    w("  *T = pka_sel<0, 0, 0, 1>(q, *S);")
    w("}")
    w("template <int L> LSN_HD void lsn_cyc_fwd(s2* a, s2 q);")
    w("template <int L, bool NORM> LSN_HD void lsn_cyc_bwd(s2* b, const s2* A, s2 q, s2* M0o, s2* M1o);")
    w("template <int L> LSN_HD void lsn_cyc_from_c(s2* v);")
    w("template <int L> LSN_HD void lsn_cyc_to_c(s2* v);")
    for L in range(NL):
        fs, bs = fwd_spec(L), bwd_spec(L)
        w("template <> LSN_HDS void lsn_cyc_fwd<%d>(s2* a, s2 q)" % L)
        w("{")
        w("  s2 S, T;")
        w("  lsn_cyc_st(q, &S, &T);")
        for r, cands in enumerate(fs):
            for j, (ra, hp, hq, i, gp, gq) in enumerate(cands):
                w("  const s2 %s%d = pka_sel<%d, %d, %d, %d>(a[%d], %s);" % ("xy"[j], r, hp, hq, gp, gq, ra, "T" if i else "S"))
        w("  " + " ".join("a[%d] = pkmax(x%d, y%d);" % (r, r, r) for r in range(4)))
        w("}")
        w("template <> LSN_HDS void lsn_cyc_bwd<%d, true>(s2* b, const s2* A, s2 q, s2* M0o, s2* M1o);" % L)
        for norm in ("false", "true"):
            w("template <> LSN_HDS void lsn_cyc_bwd<%d, %s>(s2* b, const s2* A, s2 q, s2* M0o, s2* M1o)" % (L, norm))
            w("{")
            w("  s2 S, T;")
            w("  lsn_cyc_st(q, &S, &T);")
            for r, per_i in enumerate(bs):
                for i, (rs, hp, hq, gp, gq) in enumerate(per_i):
                    w("  const s2 u%d%d = pka_sel<%d, %d, %d, %d>(b[%d], %s);" % (i, r, hp, hq, gp, gq, rs, "T" if i else "S"))
            for i in (0, 1):
                w("  *M%do = pkmax(pkmax(pka_sat(A[0], u%d0), pka_sat(A[1], u%d1)), pkmax(pka_sat(A[2], u%d2), pka_sat(A[3], u%d3)));" % (i, i, i, i, i))
            w("  " + " ".join("b[%d] = pkmax(u0%d, u1%d);" % (r, r, r) for r in range(4)))
            if norm == "true":
                w("  const s2 n = b[0];")
                w("  " + " ".join("b[%d] = pks_sel<0, 1, 0, 0>(b[%d], n);" % (r, r) for r in range(4)))
            w("}")
        for name, src, dst in (("from_c", C, T[L]), ("to_c", T[L], C)):
            w("template <> LSN_HDS void lsn_cyc_%s<%d>(s2* v)" % (name, L))
            w("{")
            if src == dst:
                w("  (void)v;")
            else:
                for r, (ra, rb, sel) in enumerate(conv_spec(src, dst)):
                    w("  const s2 t%d = pk_s2(lsn_perm(pk_u32(v[%d]), pk_u32(v[%d]), 0x%08Xu));" % (r, rb, ra, sel))
                w("  v[0] = t0; v[1] = t1; v[2] = t2; v[3] = t3;")
            w("}")
    return "\n".join(o) + "\n"


if __name__ == "__main__":
    check()
    path = os.path.join(ROOT, "ltesniffer_amd", "csrc", "kernels", "lsn_turbo_cyc.h")
    open(path, "w").write(emit())
    print("layouts:")
    for L in range(NL):
        print("  %d: %s" % (L, T[L]))
    print("wrote", path)
