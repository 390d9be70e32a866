#!/usr/bin/env python3
"""tools/viterbi_rot_design.py - numpy check of the ROTATING-LABEL add-compare-select of k_viterbi (round 5, stage_a.hip: viterbi_tb) against the plain
one-lane-per-state recursion it replaces (two ds_bpermute per trellis step).

Plain: lane j = new state j = ((old << 1) | b) & 63; predecessors j >> 1 (x = 0) and (j >> 1) | 32 (x = 1); candidates 2 m(x0) + e(j) and 2 m(x1) - e(j) on
doubled metrics, decision dd = [cand(x1) < cand(x0)] (tie: x0), trace-back st = (st >> 1) | (dd << 5), decoded bit = st & 1.

Rotating labels: the butterfly (old o, o + 32) -> (new 2 o, 2 o + 1) is computed IN PLACE - the lane that held o keeps 2 o, the lane that held o + 32 keeps
2 o + 1 - so after t steps lane p holds state rotl^t(p) (6-bit rotation) and the two lanes of a butterfly differ in lane bit beta(t) = (5 - t) mod 6: the
exchange is lane XOR 1 << beta(t) - DPP quad_perm (1, 2), two DPP moves (4), row_ror:8 (8), ds_swizzle (16), ds_bpermute (32) - ONE move per step instead of
two ds_bpermute.  With e' = (own old state is the x0 one ? e : -e) both lanes of a butterfly hold the SAME e' (e(n ^ 1) = - e(n)), so
    a_own = m + e',  a_par = (m - e')(partner) = m(partner) - e',  m' = min(a_own, a_par),  raw = [a_par < a_own + cls]   (cls = 1 on the x1 lane: ties go to x0)
and raw means "the survivor came from the partner lane": the trace-back walks LANES, p ^= raw << beta(t), and the decoded bit of step t is bit beta(t) of p.
This script runs both on noise, saturated noise and noisy code words for every DCI length and checks: decoded bits, end state, every surviving path."""
import numpy as np


def rotl(v, r):
    r %= 6
    return ((v << r) | (v >> (6 - r))) & 63 if r else v


def gen_tables():
    j = np.arange(64)
    b, s0 = j & 1, j >> 1
    pc = lambda x: np.array([bin(int(v)).count("1") & 1 for v in x])
    c = [b ^ pc(s0 & 0x36), b ^ pc(s0 & 0x27), b ^ pc(s0 & 0x2B)]
    return c


C = gen_tables()


def e_of_state(q):
    """e(j) = 2 * sum_i (c_i ? 255 - q_i : q_i) - 765 for all 64 new states j; q = three bytes"""
    bm = sum(np.where(C[i] == 1, 255 - int(q[i]), int(q[i])) for i in range(3))
    return 2 * bm - 765


def plain(sym, nbits):
    D = len(sym)
    m = np.zeros(64, dtype=np.int64)
    j = np.arange(64)
    pa, pb = j >> 1, (j >> 1) | 32
    hist = []
    for p in range(3):
        for k in range(D):
            e = e_of_state(sym[k])
            a0, a1 = m[pa] + e, m[pb] - e
            dd = (a1 < a0).astype(np.int64)
            m = np.minimum(a0, a1)
            if p:
                hist.append(dd)
    st = int(np.lexsort((j, m))[0])  # minimum metric, lowest index on ties
    end = st
    bits = []
    for t in range(2 * D - 1, -1, -1):
        k = t % D
        if t < D:
            bits.append((k, st & 1))
        st = (st >> 1) | (int(hist[t][st]) << 5)
    return end, sorted(bits), st


def rotating(sym, nbits):
    D = len(sym)
    lane = np.arange(64)
    m = np.zeros(64, dtype=np.int64)
    # per phase: e' selector (new label of every lane, class)
    phase = []
    for ph in range(6):
        n = np.array([rotl(int(p), ph + 1) for p in lane])
        cls = n & 1
        phase.append((n, cls))
    hist = []
    t = 0
    for p in range(3):
        for k in range(D):
            n, cls = phase[t % 6]
            beta = (5 - t) % 6
            e = e_of_state(sym[k])[n]
            ep = np.where(cls == 1, -e, e)
            part = lane ^ (1 << beta)
            assert np.array_equal(ep, ep[part])            # both lanes of a butterfly hold the same e'
            a_own, a_par = m + ep, (m - ep)[part]
            raw = (a_par < a_own + cls).astype(np.int64)
            m = np.minimum(a_own, a_par)
            if p:
                hist.append(raw)
            t += 1
    T = 3 * D
    label = np.array([rotl(int(p), T) for p in lane])
    pos = int(np.lexsort((label, m))[0])
    end = int(label[pos])
    bits = []
    for t in range(3 * D - 1, D - 1, -1):
        k, beta = t % D, (5 - t) % 6
        if t < 2 * D:
            bits.append((k, (pos >> beta) & 1))
        pos ^= int(hist[t - D][pos]) << beta
    return end, sorted(bits), rotl(pos, D)  # label of the lane the walk ends on (state in front of pass 2)


def main():
    g = np.random.Generator(np.random.PCG64(5))
    n = 0
    for nbits in list(range(8, 65)):
        D = nbits + 16
        for kind in range(4):
            if kind == 0:
                sym = g.integers(0, 256, size=(D, 3))
            elif kind == 1:
                sym = g.choice([0, 255], size=(D, 3))          # saturated: many ties
            elif kind == 2:
                sym = np.full((D, 3), 128)                      # all ties
                sym[g.integers(0, D)] = [0, 255, 0]
            else:                                               # a tail-biting code word + noise
                u = g.integers(0, 2, size=D)
                st = 0
                for x in u[-6:]:
                    st = ((st << 1) | int(x)) & 63
                sym = np.zeros((D, 3), dtype=np.int64)
                for k in range(D):
                    j = ((st << 1) | int(u[k])) & 63
                    for i in range(3):
                        sym[k, i] = 255 if C[i][j] else 0
                    st = j
                sym = np.clip(sym + g.normal(0, 90, size=sym.shape), 0, 255).astype(np.int64)
            a, b = plain(sym, nbits), rotating(sym, nbits)
            assert a == b, (nbits, kind, a[0], b[0])
            n += 1
    print("rotating-label add-compare-select == plain recursion on %d blocks (every DCI length 8..64, noise / saturated / all-ties / noisy code words): "
          "decoded bits, end state and the state in front of the kept pass" % n)


if __name__ == "__main__":
    main()
