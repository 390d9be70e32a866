// CPU check of the turbo kernel's per-lane arithmetic (ltesniffer_amd/csrc/kernels/lsn_turbo_core.h, the text k_turbo is compiled from):
// the lanes of a workgroup are run one after the other over plain memory standing in for the LDS, the window-boundary exchange and the
// early-stop CRC are done the way the kernel does them, and iteration count, verdict and every decided bit are compared with the CPU
// oracle's decoder (o_turbo_decode_cb) - on noisy code words of every interleaver size class and on pure noise (12 iterations, where any
// arithmetic difference snowballs).  Built with LSN_TURBO_RANGE_CHECK: every packed add / subtract is verified to stay inside int16.
// Needs clang (ext_vector_type); test infrastructure - links the oracle.
#define LSN_TURBO_RANGE_CHECK 1
#include "../../ltesniffer_amd/csrc/kernels/lsn_turbo_core.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
extern "C" {
int o_turbo_decode_cb(const int16_t* d3, int K, int max_iter, uint32_t crc_poly, uint8_t* bits, int* crc_ok);
int o_qpp_find(int K, int* f1, int* f2);
int o_turbo_nwin(int K);
uint32_t o_crc_bits(uint32_t poly, int order, const uint8_t* bits, int n);
}
static uint64_t rng_s = 88172645463325252ull;
static uint32_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (uint32_t)(rng_s >> 16); }
static double gauss() { double s = 0; for (int i = 0; i < 12; i++) s += (rnd() & 0xFFFF) / 65536.0; return s - 6.0; }

static void tail_beta(const int* ts, const int* tp, int* beta)  // as in stage_c.hip
{
  int b[8], bn[8];
  for (int S = 0; S < 8; S++) b[S] = S == 0 ? 0 : LSN_NEG_METRIC;
  for (int t = 2; t >= 0; t--) {
    for (int S = 0; S < 8; S++) {
      int s1 = (S >> 2) & 1, s2_ = (S >> 1) & 1, s3 = S & 1;
      int u = s2_ ^ s3, z = s1 ^ s3, Sn = (s1 << 1) | s2_;
      bn[S] = b[Sn] + (u ? ts[t] : 0) + (z ? tp[t] : 0);
    }
    for (int S = 0; S < 8; S++) b[S] = bn[S];
  }
  for (int S = 7; S >= 0; S--) beta[S] = b[S] - b[0];
}

// the kernel, lane by lane
template <int NT>
static int decode_like_kernel(const int16_t* d3, int K, int max_iter, uint32_t poly, uint8_t* bits, int* ok_out)
{
  const int D = K + 4, P = lsn_turbo_nwin(K), W = K / P;
  int f1, f2;
  o_qpp_find(K, &f1, &f2);
  const uint32_t magicW = ((1u << 20) + (uint32_t)W - 1u) / (uint32_t)W;
  std::vector<uint32_t> il(lsn_turbo_il_words(K));
  lsn_turbo_il_fill(il.data(), K, f1, f2);
  std::vector<uint32_t> spp(K + 256, 0);   // (an odd window length reads one step past the block: lsn_map_pass_lane)
  std::vector<int16_t> ext(K + 256, 0);
  std::vector<uint8_t> ckpt(TB_CKPT_BYTES + 64);
  const int16_t *d0 = d3, *d1 = d3 + D, *d2 = d3 + 2 * D;
  for (int t = 0; t < K; t++) {  // k_rm's output order: slot t holds position x = (t % P) * W + t / P
    const int x = (t % P) * W + t / P;
    spp[t] = ((uint32_t)d0[x] & 0x3FFu) | (((uint32_t)d1[x] & 0x3FFu) << 10) | (((uint32_t)d2[x] & 0x3FFu) << 20);
  }
  int tail[12];
  for (int s = 0; s < 3; s++) for (int j = 0; j < 4; j++) tail[s * 4 + j] = (s == 0 ? d0 : s == 1 ? d1 : d2)[K + j];
  int bt1i[8], bt2i[8];
  {
    const int *s4 = tail, *q1 = tail + 4, *q2 = tail + 8;
    int ts1[3] = {s4[0], q2[0], q1[1]}, tp1[3] = {q1[0], s4[1], q2[1]};
    int ts2[3] = {s4[2], q2[2], q1[3]}, tp2[3] = {q1[2], s4[3], q2[3]};
    tail_beta(ts1, tp1, bt1i);
    tail_beta(ts2, tp2, bt2i);
  }
  s2 bt1[4], bt2[4];
  lsn_pack_c(bt1i, bt1); lsn_pack_c(bt2i, bt2);
  TurboLds m;
  m.spp = spp.data(); m.ext = ext.data(); m.ckpt = ckpt.data();
  std::vector<s2> na1(4 * NT, s2{0, 0}), nb1(4 * NT, s2{0, 0}), na2(4 * NT, s2{0, 0}), nb2(4 * NT, s2{0, 0}), ae(4 * NT), bo(4 * NT);
  auto pass = [&](bool second, std::vector<s2>& na, std::vector<s2>& nb, const s2* bt) {
    for (int lane = 0; lane < NT; lane++) {
      for (int k = 0; k < 4; k++) ae[4 * lane + k] = bo[4 * lane + k] = s2{0, 0};
      if (lane >= P) continue;  // lanes without a window sit the pass out (stage_c.hip: map_pass)
      if (second) lsn_map_pass_lane<true>(m, il.data(), NT, lane, K, P, W, &na[4 * lane], &nb[4 * lane], bt, &ae[4 * lane], &bo[4 * lane]);
      else lsn_map_pass_lane<false>(m, il.data(), NT, lane, K, P, W, &na[4 * lane], &nb[4 * lane], bt, &ae[4 * lane], &bo[4 * lane]);
    }
    // exchange through the check-point area, as the kernel does it (7 halves per lane and direction)
    for (int lane = 0; lane < NT; lane++) { lsn_ckpt_store(m.ckpt, NT, 0, lane, &ae[4 * lane]); lsn_ckpt_store(m.ckpt, NT, 1, lane, &bo[4 * lane]); }
    for (int lane = 0; lane < NT; lane++) {
      const int lm = lane > 0 ? lane - 1 : 0, lq = lane + 1 < NT ? lane + 1 : lane;
      lsn_ckpt_load(m.ckpt, NT, 0, lm, &na[4 * lane]);
      lsn_ckpt_load(m.ckpt, NT, 1, lq, &nb[4 * lane]);
    }
  };
  int it = 0, ok = 0;
  while (it < max_iter && !ok) {
    pass(false, na1, nb1, bt1);
    pass(true, na2, nb2, bt2);
    it++;
    for (int x = 0; x < K; x++) bits[x] = (uint8_t)(ext[tr_idx(x, W, P, magicW)] & 1);
    ok = o_crc_bits(poly, 24, bits, K) == 0;
  }
  *ok_out = ok;
  return it;
}

// 36.212 5.1.3.2 encoder (for code words that do decode)
static void encode(const uint8_t* c, int K, int f1, int f2, int16_t* d3, double amp, double sigma)
{
  const int D = K + 4;
  std::vector<uint8_t> x(D), z(D), zp(D), xp(4);
  auto rsc = [&](auto in, uint8_t* par, uint8_t* tail_sys, uint8_t* tail_par) {
    int r1 = 0, r2 = 0, r3 = 0;
    for (int i = 0; i < K; i++) {
      const int a = in(i) ^ r2 ^ r3;
      par[i] = (uint8_t)(a ^ r1 ^ r3);
      r3 = r2; r2 = r1; r1 = a;
    }
    for (int i = 0; i < 3; i++) {
      const int u = r2 ^ r3;
      tail_sys[i] = (uint8_t)u;
      tail_par[i] = (uint8_t)(r1 ^ r3);
      r3 = r2; r2 = r1; r1 = 0;
    }
  };
  std::vector<uint8_t> p1(K), p2(K);
  uint8_t ts1[3], tp1[3], ts2[3], tp2[3];
  rsc([&](int i) { return (int)c[i]; }, p1.data(), ts1, tp1);
  rsc([&](int i) { return (int)c[(int)(((long long)f1 * i + (long long)f2 * i * i) % K)]; }, p2.data(), ts2, tp2);
  std::vector<uint8_t> b0(D), b1(D), b2(D);
  for (int i = 0; i < K; i++) { b0[i] = c[i]; b1[i] = p1[i]; b2[i] = p2[i]; }
  b0[K] = ts1[0]; b1[K] = tp1[0]; b2[K] = ts1[1];
  b0[K + 1] = tp1[1]; b1[K + 1] = ts1[2]; b2[K + 1] = tp1[2];
  b0[K + 2] = ts2[0]; b1[K + 2] = tp2[0]; b2[K + 2] = ts2[1];
  b0[K + 3] = tp2[1]; b1[K + 3] = ts2[2]; b2[K + 3] = tp2[2];
  for (int s = 0; s < 3; s++)
    for (int i = 0; i < D; i++) {
      const uint8_t bit = (s == 0 ? b0 : s == 1 ? b1 : b2)[i];
      double v = (bit ? amp : -amp) + sigma * gauss();
      int q = (int)(v < 0 ? v - 0.5 : v + 0.5);
      q = q > 511 ? 511 : (q < -511 ? -511 : q);
      d3[s * D + i] = (int16_t)q;
    }
}

template <int NT>
static int run_case(int K, int kind, double amp, double sigma, int max_iter, long* iters_sum)
{
  const int D = K + 4;
  int f1, f2;
  if (o_qpp_find(K, &f1, &f2) < 0) return 0;
  std::vector<int16_t> d3(3 * D);
  std::vector<uint8_t> c(K), ba(K), bb(K);
  const uint32_t poly = (kind & 1) ? 0x1800063u : 0x1864CFBu;
  if (kind < 2) {  // a code word whose last 24 bits are the CRC of the rest
    for (int i = 0; i < K - 24; i++) c[i] = (uint8_t)(rnd() & 1);
    std::vector<uint8_t> tmp(K, 0);
    memcpy(tmp.data(), c.data(), K - 24);
    const uint32_t r = o_crc_bits(poly, 24, tmp.data(), K);
    for (int i = 0; i < 24; i++) c[K - 24 + i] = (uint8_t)((r >> (23 - i)) & 1);
    encode(c.data(), K, f1, f2, d3.data(), amp, sigma);
  } else if (kind == 2) {
    for (auto& v : d3) v = (int16_t)((int)(rnd() % 1023) - 511);  // uniform noise
  } else if (kind == 3) {
    for (auto& v : d3) v = (int16_t)((rnd() & 1) ? 511 : -511);   // saturated noise
  } else if (kind == 4) {
    for (auto& v : d3) v = -511;
  } else {
    for (auto& v : d3) v = 511;
  }
  int oka = 0, okb = 0;
  const int ia = o_turbo_decode_cb(d3.data(), K, max_iter, poly, ba.data(), &oka);
  const int ib = decode_like_kernel<NT>(d3.data(), K, max_iter, poly, bb.data(), &okb);
  *iters_sum += ib;
  if (ia != ib || oka != okb || memcmp(ba.data(), bb.data(), K) != 0) {
    int nd = 0;
    for (int i = 0; i < K; i++) nd += ba[i] != bb[i];
    std::fprintf(stderr, "MISMATCH K=%d kind=%d: oracle it=%d ok=%d, kernel text it=%d ok=%d, %d bits differ\n", K, kind, ia, oka, ib, okb, nd);
    return 1;
  }
  return 0;
}

int main(int argc, char** argv)
{
  const int quick = argc > 1 && !strcmp(argv[1], "quick");
  std::vector<int> Ks;
  for (int K = 40; K <= 512; K += 8) Ks.push_back(K);
  for (int K = 528; K <= 1024; K += 16) Ks.push_back(K);
  for (int K = 1056; K <= 2048; K += 32) Ks.push_back(K);
  for (int K = 2112; K <= 6144; K += 64) Ks.push_back(K);
  int bad = 0, n = 0;
  long iters = 0;
  for (size_t i = 0; i < Ks.size(); i++) {
    const int K = Ks[i];
    if (quick && (i % 7) != 0 && K != 6144 && K != 40) continue;
    const bool two = lsn_turbo_nwin(K) > 64;
    for (int kind = 0; kind < 6; kind++) {
      if (quick && kind >= 3 && (i % 21) != 0) continue;
      const double sigma = kind == 0 ? 60.0 + (rnd() % 60) : 140.0 + (rnd() % 80);  // easy / marginal
      const int mi = kind < 2 ? 12 : (quick ? 4 : 12);
      bad += two ? run_case<128>(K, kind, 64.0, sigma, mi, &iters) : run_case<64>(K, kind, 64.0, sigma, mi, &iters);
      n++;
    }
  }
  std::printf("%d cases, %ld iterations, %d mismatches\n", n, iters, bad);
  return bad ? 1 : 0;
}
