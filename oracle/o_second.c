/* o_second.c - ORACLE (test infrastructure only): SECOND-OPINION decoders.
 *
 * The production oracle (o_pdsch.c, o_conv.c) restates the decoders the way the HIP kernels run them: trellis windows with
 * next-iteration initialisation, 10-bit soft values, extrinsic scaled by 3/4, u8 Viterbi symbols - design parameters of this repo,
 * not of srsRAN (whose sources are absent; /root/reference/src/src/DL_Sniffer_PDSCH.cc:997 calls srsran_ue_dl_decode_pdsch,
 * falcon_pdcch.c:142 calls srsran_pdcch_dci_decode).  Nothing in the reference pins those numerics, so this file holds decoders
 * that share NONE of those parameters and are as close to the text-book algorithm as possible:
 *   - turbo: max-log-MAP over the FULL trellis of the code block (no windows; exact start state and termination), 16-bit soft
 *     input (no 10-bit clip), 32-bit metrics, no extrinsic scaling or clipping, early stop on the code-block CRC;
 *   - rate de-matching without clipping of the combined values;
 *   - DCI: tail-biting Viterbi on FLOAT branch metrics (no 8-bit quantisation), five trips around the circle, decisions of the middle one.
 * The worker can be switched to them (o_worker_set_second_opinion); tests/test_second_opinion.py and tools/second_opinion_sweep.py count,
 * per SNR point, how often a CRC verdict or a payload differs from the production oracle: the measured distance between "our" numerics
 * and an unwindowed, unclipped decoder. */
#include "lsn_oracle.h"
#include "../spec/lte_tables.h"
#include <stdlib.h>
#include <string.h>

/* ---- turbo rate de-matching without clipping (36.212 5.1.4.1.2): int32 sums ---- */
static void rm_rx_unclipped(const int16_t* e, int E, int K, int F, int rv, int32_t* d3)
{
  int D = K + 4, R = (D + 31) / 32, KP = 32 * R, ND = KP - D, Ncb = 3 * KP;
  int* map = (int*)malloc(sizeof(int) * (size_t)Ncb);
  memset(d3, 0, sizeof(int32_t) * (size_t)(3 * D));
  for (int k = 0; k < KP; k++) {
    int col = k / R, row = k % R;
    int y = row * 32 + lsn_perm_tc[col];
    int i01 = y - ND;
    map[k] = (i01 >= 0 && i01 >= F) ? i01 : -1;
    map[KP + 2 * k] = (i01 >= 0 && i01 >= F) ? D + i01 : -1;
    int pi = (lsn_perm_tc[col] + 32 * row + 1) % KP;
    map[KP + 2 * k + 1] = (pi - ND >= 0) ? 2 * D + pi - ND : -1;
  }
  int k0 = R * (2 * ((Ncb + 8 * R - 1) / (8 * R)) * rv + 2);
  int k = 0, j = 0;
  while (k < E) {
    int o = map[(k0 + j) % Ncb];
    if (o >= 0) { d3[o] += e[k]; k++; }
    j++;
  }
  for (int i = 0; i < F; i++) { d3[i] = -(1 << 20); d3[D + i] = -(1 << 20); } /* filler bits: known zeros, very reliable */
  free(map);
}

static uint8_t nx[8][2], pr[8][2];
static int tinit = 0;
static void tr_init(void)
{
  for (int S = 0; S < 8; S++)
    for (int u = 0; u < 2; u++) {
      int s1 = (S >> 2) & 1, s2 = (S >> 1) & 1, s3 = S & 1;
      int a = u ^ s2 ^ s3;
      pr[S][u] = (uint8_t)(a ^ s1 ^ s3);
      nx[S][u] = (uint8_t)((a << 2) | (s1 << 1) | s2);
    }
  tinit = 1;
}
#define NEG (-(1LL << 40))

/* one constituent decoder over the whole block: K information steps + 3 termination steps with their own systematic / parity values */
static void map_full(int K, const int32_t* sys, const int64_t* ext_in, const int32_t* par, const int* idx, const int32_t* ts, const int32_t* tp,
                     int64_t* ext_out, int64_t* llr)
{
  int64_t(*alpha)[8] = (int64_t(*)[8])malloc(sizeof(int64_t[8]) * (size_t)(K + 1));
  for (int S = 0; S < 8; S++) alpha[0][S] = S == 0 ? 0 : NEG;
  for (int t = 0; t < K; t++) {
    int64_t lsa = (int64_t)sys[idx[t]] + ext_in[idx[t]], lp = par[t], n[8];
    for (int S = 0; S < 8; S++) n[S] = NEG * 2;
    for (int S = 0; S < 8; S++)
      for (int u = 0; u < 2; u++) {
        int64_t m = alpha[t][S] + (u ? lsa : 0) + (pr[S][u] ? lp : 0);
        if (m > n[nx[S][u]]) n[nx[S][u]] = m;
      }
    int64_t mx = n[0];
    for (int S = 1; S < 8; S++) if (n[S] > mx) mx = n[S];
    for (int S = 0; S < 8; S++) alpha[t + 1][S] = n[S] - mx;
  }
  /* termination: three steps that drive the encoder to state 0 (36.212 5.1.3.2.2): input u = s2 ^ s3, parity z = s1 ^ s3 */
  int64_t beta[8], bn[8];
  for (int S = 0; S < 8; S++) beta[S] = S == 0 ? 0 : NEG;
  for (int t = 2; t >= 0; t--) {
    for (int S = 0; S < 8; S++) {
      int s1 = (S >> 2) & 1, s2 = (S >> 1) & 1, s3 = S & 1;
      int u = s2 ^ s3, z = s1 ^ s3, Sn = (s1 << 1) | s2;
      bn[S] = beta[Sn] + (u ? ts[t] : 0) + (z ? tp[t] : 0);
    }
    memcpy(beta, bn, sizeof(beta));
  }
  for (int t = K - 1; t >= 0; t--) {
    int64_t lsa = (int64_t)sys[idx[t]] + ext_in[idx[t]], lp = par[t], m0 = NEG * 4, m1 = NEG * 4;
    for (int S = 0; S < 8; S++) {
      int64_t b0 = beta[nx[S][0]] + (pr[S][0] ? lp : 0), b1 = beta[nx[S][1]] + lsa + (pr[S][1] ? lp : 0);
      int64_t v0 = alpha[t][S] + b0, v1 = alpha[t][S] + b1;
      if (v0 > m0) m0 = v0;
      if (v1 > m1) m1 = v1;
      bn[S] = b0 > b1 ? b0 : b1;
    }
    int64_t L = m1 - m0;
    llr[t] = L;
    ext_out[idx[t]] = L - lsa;
    int64_t mx = bn[0];
    for (int S = 1; S < 8; S++) if (bn[S] > mx) mx = bn[S];
    for (int S = 0; S < 8; S++) beta[S] = bn[S] - mx;
  }
  free(alpha);
}

int o_turbo_decode_cb_second(const int32_t* d3, int K, int max_iter, uint32_t crc_poly, uint8_t* bits, int* crc_ok)
{
  if (!tinit) tr_init();
  int D = K + 4, f1, f2;
  if (o_qpp_find(K, &f1, &f2) < 0) return -1;
  const int32_t *d0 = d3, *d1 = d3 + D, *d2 = d3 + 2 * D;
  int* pi = (int*)malloc(sizeof(int) * (size_t)K);
  int* id = (int*)malloc(sizeof(int) * (size_t)K);
  int64_t* e12 = (int64_t*)calloc((size_t)K, sizeof(int64_t));
  int64_t* e21 = (int64_t*)calloc((size_t)K, sizeof(int64_t));
  int64_t* llr = (int64_t*)malloc(sizeof(int64_t) * (size_t)K);
  for (int i = 0; i < K; i++) { pi[i] = (int)(((long long)f1 * i + (long long)f2 * i * i) % K); id[i] = i; }
  int32_t ts1[3] = {d0[K], d2[K], d1[K + 1]}, tp1[3] = {d1[K], d0[K + 1], d2[K + 1]};
  int32_t ts2[3] = {d0[K + 2], d2[K + 2], d1[K + 3]}, tp2[3] = {d1[K + 2], d0[K + 3], d2[K + 3]};
  int it = 0, ok = 0;
  while (it < max_iter && !ok) {
    map_full(K, d0, e21, d1, id, ts1, tp1, e12, llr);
    map_full(K, d0, e12, d2, pi, ts2, tp2, e21, llr);
    for (int i = 0; i < K; i++) bits[pi[i]] = llr[i] > 0 ? 1 : 0;
    it++;
    ok = (o_crc_bits(crc_poly, 24, bits, K) == 0);
  }
  if (crc_ok) *crc_ok = ok;
  free(pi); free(id); free(e12); free(e21); free(llr);
  return it;
}

/* one transport block with the second-opinion chain; same contract as o_pdsch_decode_tb */
int o_pdsch_decode_tb_second(const int16_t* e, int G, int tbs, int Qm, int NL, int rv, int max_iter, uint8_t* payload, int* iters_total)
{
  o_cbsegm_t s;
  if (o_cbsegm(&s, tbs) || Qm <= 0 || G <= 0) return 0;
  int Gp = G / (NL * Qm), gamma = Gp % s.C;
  uint8_t* tbbits = (uint8_t*)malloc((size_t)(tbs + 24 + 64));
  int32_t* d3 = (int32_t*)malloc(sizeof(int32_t) * 3 * (6144 + 4));
  uint8_t* cb = (uint8_t*)malloc(6144);
  int rp = 0, wp = 0, all_ok = 1, its = 0;
  for (int r = 0; r < s.C; r++) {
    int K = r < s.Cm ? s.Km : s.Kp, F = r == 0 ? s.F : 0;
    int E = (r <= s.C - gamma - 1) ? NL * Qm * (Gp / s.C) : NL * Qm * ((Gp + s.C - 1) / s.C);
    int ok = 0;
    if (rp + E > G) E = G - rp;
    rm_rx_unclipped(e + rp, E, K, F, rv, d3);
    int n = o_turbo_decode_cb_second(d3, K, max_iter, s.C > 1 ? O_CRC24B : O_CRC24A, cb, &ok);
    its += n > 0 ? n : 0;
    if (!ok) all_ok = 0;
    int take_n = K - F - (s.C > 1 ? 24 : 0);
    memcpy(tbbits + wp, cb + F, (size_t)take_n);
    wp += take_n;
    rp += E;
  }
  if (iters_total) *iters_total += its;
  int crc_ok = 0;
  if (wp == tbs + 24) {
    uint32_t par = 0;
    for (int i = 0; i < 24; i++) par = (par << 1) | tbbits[tbs + i];
    crc_ok = (o_crc_bits(O_CRC24A, 24, tbbits, tbs) == par) && par != 0 && all_ok;
  }
  o_pack_bits(tbbits, payload, tbs);
  free(tbbits); free(d3); free(cb);
  return crc_ok;
}

/* ---- DCI candidate: float tail-biting Viterbi ---- */
static inline int par6(unsigned x) { x ^= x >> 4; x ^= x >> 2; x ^= x >> 1; return (int)(x & 1u); }
uint16_t o_dci_decode_second(const float* llr, int E, int nof_bits, uint8_t* payload)
{
  int D = nof_bits + 16;
  float rm[3 * (O_DCI_MAX_BITS + 16)];
  static uint64_t dec[5 * (O_DCI_MAX_BITS + 16)];
  uint8_t bits[O_DCI_MAX_BITS + 16];
  o_rm_conv_rx(llr, E, rm, 3 * D);
  double m[64], mn[64];
  for (int s = 0; s < 64; s++) m[s] = 0.0;
  const int TRIPS = 5, T = TRIPS * D;
  for (int t = 0; t < T; t++) {
    const float* q = rm + 3 * (t % D);  /* llr > 0 <=> bit 1 more likely; cost of hypothesising code bit c: c ? -q : +q (halved, constant dropped) */
    uint64_t dw = 0;
    for (int j = 0; j < 64; j++) {
      int b = j & 1, s0 = j >> 1, s1 = s0 | 32;
      int c0 = b ^ par6((unsigned)s0 & 0x36u), c1 = b ^ par6((unsigned)s0 & 0x27u), c2 = b ^ par6((unsigned)s0 & 0x2Bu);
      double bm0 = (c0 ? -q[0] : q[0]) + (c1 ? -q[1] : q[1]) + (c2 ? -q[2] : q[2]);
      double a0 = m[s0] + bm0, a1 = m[s1] - bm0;  /* the branch from s1 carries the complementary code bits */
      if (a1 < a0) { mn[j] = a1; dw |= (uint64_t)1 << j; } else mn[j] = a0;
    }
    dec[t] = dw;
    double mnm = mn[0];
    for (int s = 1; s < 64; s++) if (mn[s] < mnm) mnm = mn[s];
    for (int s = 0; s < 64; s++) m[s] = mn[s] - mnm;
  }
  int best = 0;
  for (int s = 1; s < 64; s++) if (m[s] < m[best]) best = s;
  int st = best;
  for (int t = T - 1; t >= 0; t--) {
    if (t >= 2 * D && t < 3 * D) bits[t - 2 * D] = (uint8_t)(st & 1);
    int d = (int)((dec[t] >> st) & 1u);
    st = (st >> 1) | (d << 5);
  }
  memcpy(payload, bits, (size_t)nof_bits);
  uint32_t p = 0;
  for (int i = 0; i < 16; i++) p = (p << 1) | bits[nof_bits + i];
  uint32_t crc = o_crc_bits(O_CRC16, 16, bits, nof_bits);
  return (uint16_t)((p ^ crc) & 0xFFFFu);
}
