"""A second opinion on the receiver front end, written from TS 36.211 alone in float64 numpy (no line of it follows the oracle or the reference): OFDM demodulation
(section 6.12), the cell-specific reference signals (6.10.1: Gold sequence 7.2, mapping 6.10.1.2), least-squares channel samples on them and a plain estimator
(linear in frequency between the pilots of a symbol, linear in time between pilot symbols - no smoothing).  It serves tests/test_frontend_truth.py: the oracle's
resource grid against this FFT, and the oracle's channel estimate / noise / CFO figures against GROUND TRUTH (the same capture rendered without noise gives the
true channel on every pilot).  Test infrastructure only."""
import numpy as np


def symbol_starts(nfft, cp=0):
    """sample index of the first useful sample of each of the 14 (normal CP) / 12 (extended) symbols of a subframe (36.211 Table 6.12-1, scaled from 2048)"""
    out, pos = [], 0
    for slot in range(2):
        for l in range(7 if cp == 0 else 6):
            ncp = (160 if l == 0 else 144) * nfft // 2048 if cp == 0 else 512 * nfft // 2048
            pos += ncp
            out.append(pos)
            pos += nfft
    return out, pos


def ofdm_demod(x, nof_prb, cp=0):
    """x: complex time samples of one subframe at the cell's rate -> grid[nsym, 12 * nof_prb], unnormalised FFT, DC removed (36.211 6.12: k(-) = k + floor(N/2) below DC, k(+) above)"""
    nfft = len(x) // 15  # a subframe is 15 FFT lengths long at any sampling rate (1 ms x 15 kHz)
    starts, sflen = symbol_starts(nfft, cp)
    assert len(x) == sflen and nfft > 12 * nof_prb, (len(x), sflen)
    nre = 12 * nof_prb
    g = np.zeros((len(starts), nre), dtype=np.complex128)
    for i, s in enumerate(starts):
        X = np.fft.fft(np.asarray(x[s:s + nfft], dtype=np.complex128))
        g[i, :nre // 2] = X[nfft - nre // 2:]
        g[i, nre // 2:] = X[1:nre // 2 + 1]
    return g


def gold(c_init, n):
    """36.211 7.2: length-31 Gold sequence, Nc = 1600"""
    nc = 1600
    x1 = np.zeros(nc + n + 31, dtype=np.uint8)
    x2 = np.zeros(nc + n + 31, dtype=np.uint8)
    x1[0] = 1
    for i in range(31):
        x2[i] = (c_init >> i) & 1
    for i in range(nc + n):
        x1[i + 31] = x1[i + 3] ^ x1[i]
        x2[i + 31] = x2[i + 3] ^ x2[i + 2] ^ x2[i + 1] ^ x2[i]
    return x1[nc:nc + n] ^ x2[nc:nc + n]


def crs(cell_id, nof_prb, port, sf_idx, cp=0):
    """-> list of (symbol index in the subframe, k[2 * nof_prb], r[2 * nof_prb]) for one antenna port (36.211 6.10.1.1 / 6.10.1.2)"""
    nsymb = 7 if cp == 0 else 6
    out = []
    for slot in range(2):
        ns = 2 * sf_idx + slot
        for l in ((0, nsymb - 3) if port < 2 else (1,)):
            c_init = (1 << 10) * (7 * (ns + 1) + l + 1) * (2 * cell_id + 1) + 2 * cell_id + (1 if cp == 0 else 0)
            c = gold(c_init, 4 * 110).astype(np.float64)
            r = ((1 - 2 * c[0::2]) + 1j * (1 - 2 * c[1::2])) / np.sqrt(2.0)
            if port == 0:
                v = 0 if l == 0 else 3
            elif port == 1:
                v = 3 if l == 0 else 0
            elif port == 2:
                v = 3 * (ns % 2)
            else:
                v = 3 + 3 * (ns % 2)
            m = np.arange(2 * nof_prb)
            k = 6 * m + (v + cell_id % 6) % 6
            out.append((slot * nsymb + l, k, r[m + 110 - nof_prb]))
    return out


def ls_pilots(grid, cell_id, nof_prb, port, sf_idx, cp=0):
    """least-squares channel samples on the pilots of one port: list of (symbol, k, h_ls)"""
    return [(l, k, grid[l, k] * np.conj(r)) for l, k, r in crs(cell_id, nof_prb, port, sf_idx, cp)]


def interpolate(pil, nsym, nre):
    """the plain estimator: per pilot symbol linear interpolation over the subcarriers (held flat outside the outermost pilots), then linear in time between pilot
    symbols (extrapolated with the nearest pair's slope outside)"""
    ls = sorted(set(l for l, _, _ in pil))
    rows = {}
    kk = np.arange(nre)
    for l, k, h in pil:
        rows[l] = np.interp(kk, k, h.real) + 1j * np.interp(kk, k, h.imag)
    ce = np.zeros((nsym, nre), dtype=np.complex128)
    for s in range(nsym):
        if len(ls) == 1:
            ce[s] = rows[ls[0]]
            continue
        hi = next((i for i, l in enumerate(ls) if l >= s), len(ls) - 1)
        hi = max(hi, 1)
        a, b = ls[hi - 1], ls[hi]
        w = (s - a) / (b - a)
        ce[s] = (1 - w) * rows[a] + w * rows[b]
    return ce
