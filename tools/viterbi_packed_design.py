#!/usr/bin/env python3
"""Design check for the next step on k_viterbi: two PDCCH candidates per wavefront on packed 16-bit path metrics.

The kernel keeps 32-bit metrics that grow without normalisation (<= 765 per step).  With two candidates sharing a wavefront each lane would hold
(metric of candidate A | metric of candidate B << 16) and one ds_bpermute would move both.  16-bit metrics WRAP, so every comparison has to be made on
differences: d = (a1 - a0) mod 2^16 read as int16, new metric = a0 + min(d, 0).  That is exact as long as |a1 - a0| < 2^15, i.e. as long as the spread of
the metric vector plus one branch metric stays below 32768 (any state is reached from any state in 6 steps: spread <= 6 * 765 = 4590).

This script runs the tail-biting decoder both ways on random soft inputs (noise, saturated noise, noisy code words) and checks that every decision word,
the best end state and the decoded bits are identical, and reports the largest spread seen.  Numpy model of the kernel's arithmetic, no GPU."""
import sys
import numpy as np

LANE = np.arange(64)
B, S0 = LANE & 1, LANE >> 1


def _par(x):
    return np.array([bin(int(v)).count("1") & 1 for v in x])


C = [B ^ _par(S0 & 0x36), B ^ _par(S0 & 0x27), B ^ _par(S0 & 0x2B)]
PA, PB = S0, S0 | 32


def branch(q):
    """(D, 3) u8 symbols -> (D, 64) metric of the branch from predecessor j >> 1; the one from (j >> 1) | 32 is 765 minus that"""
    return sum(np.where(C[i][None, :] == 1, 255 - q[:, i:i + 1], q[:, i:i + 1]) for i in range(3)).astype(np.int64)


def decode32(q):
    D = len(q)
    bm = branch(q)
    m = np.zeros(64, np.int64)
    dec, spread = [], 0
    for p in range(3):
        for t in range(D):
            a0, a1 = m[PA] + bm[t], m[PB] + 765 - bm[t]
            d = a1 < a0
            m = np.where(d, a1, a0)
            spread = max(spread, int(m.max() - m.min()))
            if p:
                dec.append(d.copy())
    return np.array(dec), int(np.argmin(m)), spread  # argmin: lowest index on ties, as the kernel's key = metric << 6 | lane


def decode16_pair(qa, qb):
    """two candidates of the same size in one 'wavefront': uint32 word = metric A | metric B << 16, wrapping arithmetic on the halves"""
    D = len(qa)
    bma, bmb = branch(qa), branch(qb)
    m = np.zeros(64, np.uint32)
    dec_a, dec_b = [], []
    M = np.uint32(0xFFFF)

    def pk(lo, hi):
        return (lo.astype(np.uint32) & M) | ((hi.astype(np.uint32) & M) << np.uint32(16))

    def pk_add(x, y):  # v_pk_add_u16
        return pk((x & M) + (y & M), (x >> np.uint32(16)) + (y >> np.uint32(16)))

    def pk_sub(x, y):  # v_pk_sub_u16
        return pk((x & M) - (y & M), (x >> np.uint32(16)) - (y >> np.uint32(16)))

    def halves_i16(x):
        return (x & M).astype(np.uint16).view(np.int16), (x >> np.uint32(16)).astype(np.uint16).view(np.int16)

    for p in range(3):
        for t in range(D):
            g0 = pk(bma[t], bmb[t])
            g1 = pk(765 - bma[t], 765 - bmb[t])
            a0, a1 = pk_add(m[PA], g0), pk_add(m[PB], g1)   # one bpermute per predecessor moves both candidates
            da, db = halves_i16(pk_sub(a1, a0))             # differences as int16
            mn = pk(np.minimum(da, 0).view(np.uint16), np.minimum(db, 0).view(np.uint16))  # v_pk_min_i16(d, 0)
            m = pk_add(a0, mn)
            if p:
                dec_a.append(da < 0); dec_b.append(db < 0)
    la, lb = halves_i16(pk_sub(m, np.full(64, m[0], np.uint32)))   # metrics relative to lane 0, then the usual arg-min
    return (np.array(dec_a), int(np.argmin(la))), (np.array(dec_b), int(np.argmin(lb)))


def encode(bits):
    D = len(bits)
    st = 0
    for k in range(6):
        st = ((st << 1) | int(bits[D - 6 + k])) & 63
    out = np.zeros((D, 3), int)
    for t in range(D):
        j = ((st << 1) | int(bits[t])) & 63
        s0 = j >> 1
        for i, mk in enumerate((0x36, 0x27, 0x2B)):
            out[t, i] = (int(bits[t]) ^ (bin(s0 & mk).count("1") & 1)) ^ (st >> 5)
        st = j
    return out


def main():
    rng = np.random.default_rng(7)
    n, worst = 0, 0
    for D in (27 + 16, 31 + 16, 43 + 16, 57 + 16, 64 + 16):
        for kind in range(4):
            for _ in range(40 if len(sys.argv) < 2 else int(sys.argv[1])):
                qs = []
                for _c in range(2):
                    if kind == 0:
                        x = 127.5 + rng.normal(0, 40, (D, 3))
                    elif kind == 1:
                        x = np.where(rng.integers(0, 2, (D, 3)), 255.0, 0.0)
                    elif kind == 2:
                        x = 127.5 - 90.0 * (2 * encode(rng.integers(0, 2, D)) - 1) + rng.normal(0, 30, (D, 3))
                    else:
                        x = 127.5 - 40.0 * (2 * encode(rng.integers(0, 2, D)) - 1) + rng.normal(0, 60, (D, 3))
                    qs.append(np.clip(x, 0, 255).astype(np.int64))
                ra, rb = decode32(qs[0]), decode32(qs[1])
                pa, pb = decode16_pair(qs[0], qs[1])
                assert (ra[0] == pa[0]).all() and ra[1] == pa[1], (D, kind)
                assert (rb[0] == pb[0]).all() and rb[1] == pb[1], (D, kind)
                worst = max(worst, ra[2], rb[2])
                n += 2
    print("%d candidates: decisions and end states of the packed 16-bit decoder identical to the 32-bit one; largest metric spread %d (bound 6 x 765 = 4590; "
          "needed: spread + 765 < 32768)" % (n, worst))


if __name__ == "__main__":
    main()
