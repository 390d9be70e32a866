// txgen.cc - synthetic LTE eNB downlink transmitter + channel (TEST TOOLING, not product, not oracle).
// Encoder-side implementation of TS 36.211/36.212 written independently of the receiver code so that
// transmitter -> receiver loop-backs are meaningful known-answer tests.
// Produces pre-aligned subframes: IQ[rx][15*N] cf32 + the ground-truth list of MAC PDUs.
#include "../../spec/lte_tables.h"
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <memory>
#include <thread>
#include <atomic>

typedef std::complex<float> cf;
typedef std::vector<uint8_t> bits_t;

namespace {

// ---------------- rng (xorshift64*) ----------------
struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) { next(); next(); }
  uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1Dull; }
  uint32_t u32() { return (uint32_t)(next() >> 32); }
  uint32_t below(uint32_t n) { return n ? (uint32_t)(((uint64_t)u32() * n) >> 32) : 0; }
  double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  void gauss2(double& a, double& b) { double u1 = uni(), u2 = uni(); if (u1 < 1e-300) u1 = 1e-300; const double r = std::sqrt(-2.0 * std::log(u1)), ph = 6.283185307179586 * u2; a = r * std::cos(ph); b = r * std::sin(ph); }
  double gauss() { double u1 = uni(), u2 = uni(); if (u1 < 1e-300) u1 = 1e-300; return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2); }
};

// ---------------- bit primitives ----------------
uint32_t crc_calc(const uint8_t* b, int n, uint32_t poly, int order) {
  uint32_t reg = 0, top = 1u << order;
  for (int i = 0; i < n + order; i++) { reg = (reg << 1) | (i < n ? b[i] : 0); if (reg & top) reg ^= poly; }
  return reg & (top - 1);
}
void crc_attach(bits_t& b, uint32_t poly, int order, uint32_t xormask = 0) {
  uint32_t c = crc_calc(b.data(), (int)b.size(), poly, order) ^ xormask;
  for (int i = order - 1; i >= 0; i--) b.push_back((c >> i) & 1);
}
bits_t gold(uint32_t cinit, int len) {
  bits_t c(len);
  uint32_t x1 = 1, x2 = cinit & 0x7FFFFFFF;
  for (int n = 0; n < 1600 + len; n++) {
    if (n >= 1600) c[n - 1600] = (x1 ^ x2) & 1;
    uint32_t n1 = ((x1 >> 3) ^ x1) & 1, n2 = ((x2 >> 3) ^ (x2 >> 2) ^ (x2 >> 1) ^ x2) & 1;
    x1 = (x1 >> 1) | (n1 << 30); x2 = (x2 >> 1) | (n2 << 30);
  }
  return c;
}
int par6(unsigned x) { x ^= x >> 4; x ^= x >> 2; x ^= x >> 1; return x & 1; }

// tail-biting conv code 36.212 5.1.3.1: out[3*D], streams interleaved per bit (d0,d1,d2)
std::vector<bits_t> conv_encode(const bits_t& c) {
  int D = (int)c.size();
  std::vector<bits_t> d(3, bits_t(D));
  unsigned s = 0;
  for (int i = 0; i < 6; i++) s |= (unsigned)c[D - 1 - i] << i;  // bit0 = most recent
  for (int k = 0; k < D; k++) {
    int b = c[k];
    d[0][k] = b ^ par6(s & 0x36); d[1][k] = b ^ par6(s & 0x27); d[2][k] = b ^ par6(s & 0x2B);
    s = ((s << 1) | b) & 63;
  }
  return d;
}
bits_t rm_conv_tx(const std::vector<bits_t>& d, int E) {
  int D = (int)d[0].size(), R = (D + 31) / 32, KP = 32 * R, ND = KP - D;
  std::vector<int> w(3 * KP);  // -1 = NULL, else bit
  for (int s = 0; s < 3; s++)
    for (int col = 0; col < 32; col++)
      for (int r = 0; r < R; r++) { int idx = r * 32 + lsn_perm_cc[col]; w[s * KP + col * R + r] = idx >= ND ? d[s][idx - ND] : -1; }
  bits_t e(E);
  int k = 0, j = 0;
  while (k < E) { if (w[j] >= 0) e[k++] = (uint8_t)w[j]; j = (j + 1) % (3 * KP); }
  return e;
}

// turbo encoder 36.212 5.1.3.2 -> d0,d1,d2 each K+4
void rsc(const bits_t& in, bits_t& par, uint8_t* tail_x, uint8_t* tail_z) {
  int s1 = 0, s2 = 0, s3 = 0, K = (int)in.size();
  par.resize(K);
  for (int k = 0; k < K; k++) { int a = in[k] ^ s2 ^ s3; par[k] = a ^ s1 ^ s3; s3 = s2; s2 = s1; s1 = a; }
  for (int t = 0; t < 3; t++) { int u = s2 ^ s3; tail_x[t] = u; tail_z[t] = s1 ^ s3; s3 = s2; s2 = s1; s1 = 0; }
}
bool qpp(int K, int& f1, int& f2) {
  for (int i = 0; i < LSN_QPP_NSIZES; i++) if (lsn_qpp_table[i][0] == K) { f1 = lsn_qpp_table[i][1]; f2 = lsn_qpp_table[i][2]; return true; }
  return false;
}
std::vector<bits_t> turbo_encode(const bits_t& c) {
  int K = (int)c.size(), f1, f2;
  if (!qpp(K, f1, f2)) { fprintf(stderr, "txgen: bad K %d\n", K); abort(); }
  bits_t ci(K), z, zp; uint8_t x[3], zz[3], xp[3], zzp[3];
  for (int i = 0; i < K; i++) ci[i] = c[(int)(((long long)f1 * i + (long long)f2 * i * i) % K)];
  rsc(c, z, x, zz); rsc(ci, zp, xp, zzp);
  std::vector<bits_t> d(3, bits_t(K + 4));
  for (int k = 0; k < K; k++) { d[0][k] = c[k]; d[1][k] = z[k]; d[2][k] = zp[k]; }
  d[0][K] = x[0]; d[1][K] = zz[0]; d[2][K] = x[1];
  d[0][K + 1] = zz[1]; d[1][K + 1] = x[2]; d[2][K + 1] = zz[2];
  d[0][K + 2] = xp[0]; d[1][K + 2] = zzp[0]; d[2][K + 2] = xp[1];
  d[0][K + 3] = zzp[1]; d[1][K + 3] = xp[2]; d[2][K + 3] = zzp[2];
  return d;
}
bits_t rm_turbo_tx(const std::vector<bits_t>& d, int F, int rv, int E) {
  int D = (int)d[0].size(), R = (D + 31) / 32, KP = 32 * R, ND = KP - D, Ncb = 3 * KP;
  std::vector<int> w(Ncb);
  for (int k = 0; k < KP; k++) {
    int col = k / R, row = k % R, y = row * 32 + lsn_perm_tc[col], i01 = y - ND;
    w[k] = (i01 >= 0 && i01 >= F) ? d[0][i01] : -1;
    w[KP + 2 * k] = (i01 >= 0 && i01 >= F) ? d[1][i01] : -1;
    int pi = (lsn_perm_tc[col] + 32 * row + 1) % KP;
    w[KP + 2 * k + 1] = pi - ND >= 0 ? d[2][pi - ND] : -1;
  }
  int k0 = R * (2 * ((Ncb + 8 * R - 1) / (8 * R)) * rv + 2);
  bits_t e(E);
  int k = 0, j = 0;
  while (k < E) { int v = w[(k0 + j) % Ncb]; if (v >= 0) e[k++] = (uint8_t)v; j++; }
  return e;
}

struct Segm { int C, Cp, Cm, Kp, Km, F; };
bool cbsegm(int tbs, Segm& s) {
  int B = tbs + 24, Bp;
  if (B <= 6144) { s.C = 1; Bp = B; } else { s.C = (B + 6119) / 6120; Bp = B + 24 * s.C; }
  int idx = -1;
  for (int i = 0; i < LSN_QPP_NSIZES; i++) if (s.C * (int)lsn_qpp_table[i][0] >= Bp) { idx = i; break; }
  if (idx < 0) return false;
  s.Kp = lsn_qpp_table[idx][0];
  if (s.C == 1) { s.Cp = 1; s.Km = 0; s.Cm = 0; }
  else { s.Km = lsn_qpp_table[idx - 1][0]; s.Cm = (s.C * s.Kp - Bp) / (s.Kp - s.Km); s.Cp = s.C - s.Cm; }
  s.F = s.Cp * s.Kp + s.Cm * s.Km - Bp;
  return true;
}
// DL-SCH: payload bytes -> G coded bits
bits_t dlsch_encode(const uint8_t* payload, int tbs, int G, int Qm, int NL, int rv) {
  bits_t a(tbs);
  for (int i = 0; i < tbs; i++) a[i] = (payload[i >> 3] >> (7 - (i & 7))) & 1;
  crc_attach(a, 0x1864CFB, 24);
  Segm s; cbsegm(tbs, s);
  bits_t out; out.reserve(G);
  int Gp = G / (NL * Qm), gamma = Gp % s.C, rp = 0;
  for (int r = 0; r < s.C; r++) {
    int K = r < s.Cm ? s.Km : s.Kp, F = r == 0 ? s.F : 0;
    bits_t cb(K, 0);
    int n = K - F - (s.C > 1 ? 24 : 0);
    for (int i = 0; i < n; i++) cb[F + i] = a[rp + i];
    rp += n;
    if (s.C > 1) { bits_t t(cb.begin(), cb.begin() + K - 24); crc_attach(t, 0x1800063, 24); cb = t; }
    auto d = turbo_encode(cb);
    int E = (r <= s.C - gamma - 1) ? NL * Qm * (Gp / s.C) : NL * Qm * ((Gp + s.C - 1) / s.C);
    bits_t e = rm_turbo_tx(d, F, rv, E);
    out.insert(out.end(), e.begin(), e.end());
  }
  out.resize(G, 0);
  return out;
}

// ---------------- modulation 36.211 7.1 ----------------
void modulate(const bits_t& b, int Qm, std::vector<cf>& out) {
  int n = (int)b.size() / Qm;
  out.resize(n);
  static const int a64[4] = {3, 1, 5, 7};
  static const int a256[8] = {5, 7, 3, 1, 11, 9, 13, 15};
  for (int i = 0; i < n; i++) {
    const uint8_t* p = &b[(size_t)i * Qm];
    float I, Q;
    switch (Qm) {
      case 2: I = (1 - 2 * p[0]) * 0.70710678f; Q = (1 - 2 * p[1]) * 0.70710678f; break;
      case 4: I = (1 - 2 * p[0]) * (p[2] ? 3 : 1) * 0.31622777f; Q = (1 - 2 * p[1]) * (p[3] ? 3 : 1) * 0.31622777f; break;
      case 6: I = (1 - 2 * p[0]) * a64[p[2] * 2 + p[4]] * 0.15430335f; Q = (1 - 2 * p[1]) * a64[p[3] * 2 + p[5]] * 0.15430335f; break;
      default: I = (1 - 2 * p[0]) * a256[p[2] * 4 + p[4] * 2 + p[6]] * 0.076696499f; Q = (1 - 2 * p[1]) * a256[p[3] * 4 + p[5] * 2 + p[7]] * 0.076696499f; break;
    }
    out[i] = cf(I, Q);
  }
}

// ---------------- fft (double) ----------------
void fft_d(std::vector<std::complex<double>>& a, bool inv) {
  if (a.size() % 3 == 0) {  // 1536 = 3 x 512 (15 MHz): three interleaved power-of-two transforms, then the radix-3 combination
    const int N = (int)a.size(), M = N / 3;
    std::vector<std::complex<double>> f[3];
    for (int r = 0; r < 3; r++) { f[r].resize(M); for (int m = 0; m < M; m++) f[r][m] = a[3 * m + r]; fft_d(f[r], inv); }
    const double ang = 2 * M_PI / N * (inv ? 1 : -1);
    for (int k = 0; k < N; k++) {
      std::complex<double> w1(std::cos(ang * k), std::sin(ang * k)), w2(std::cos(ang * 2 * k), std::sin(ang * 2 * k));
      a[k] = f[0][k % M] + w1 * f[1][k % M] + w2 * f[2][k % M];
    }
    return;
  }
  int N = (int)a.size(), lg = 0;
  while ((1 << lg) < N) lg++;
  for (int i = 0; i < N; i++) { int j = 0; for (int b = 0; b < lg; b++) if (i & (1 << b)) j |= 1 << (lg - 1 - b); if (j > i) std::swap(a[i], a[j]); }
  for (int len = 2; len <= N; len <<= 1) {
    double ang = 2 * M_PI / len * (inv ? 1 : -1);
    for (int i = 0; i < N; i += len)
      for (int j = 0; j < len / 2; j++) {
        std::complex<double> w(std::cos(ang * j), std::sin(ang * j)), u = a[i + j], v = a[i + j + len / 2] * w;
        a[i + j] = u + v; a[i + j + len / 2] = u - v;
      }
  }
}

uint32_t log2ceil(uint32_t x) { uint32_t n = 0; while ((1u << n) < x) n++; return n; }
uint32_t riv_nbits(uint32_t n) { return log2ceil(n * (n + 1) / 2); }
uint32_t ra_P(uint32_t n) { return n <= 10 ? 1 : n <= 26 ? 2 : n <= 63 ? 3 : 4; }
bool amb(uint32_t n) { static const uint32_t a[10] = {12, 14, 16, 20, 24, 26, 32, 40, 44, 56}; for (auto v : a) if (v == n) return true; return false; }
void put(bits_t& b, uint32_t v, uint32_t n) { for (int i = (int)n - 1; i >= 0; i--) b.push_back((v >> i) & 1); }

}  // namespace

// =====================================================================================================
extern "C" {

enum { TXG_FMT0 = 0, TXG_FMT1 = 1, TXG_FMT1A = 2, TXG_FMT1C = 4, TXG_FMT2 = 6, TXG_FMT2A = 7 };

typedef struct {
  uint32_t nof_prb, nof_ports, cell_id, phich_ng_x6, nof_rx;
  float snr_db, cfo_hz;
  uint32_t delay_samples;  // second rx path extra delay (0 = flat)
  uint64_t seed;
  // scenario
  uint32_t n_rnti, dl_min, dl_max, ul_min, ul_max;
  uint32_t cfi;           // 1..3 fixed; 0 = random per subframe
  uint32_t mix_tm3_pct, mix_tm4_pct;  // rest: TM1/TM2 (format 1 / 1A)
  uint32_t pct_256qam;    // share of UEs on the 256QAM table
  uint32_t mcs_min, mcs_max;
  uint32_t sib_period;    // SI-RNTI DCI1A on sf 5 of even frames if != 0
  uint32_t rar_period;    // one RAR every n subframes (0 = none)
  uint32_t paging_period;
  uint32_t start_tti;
  uint32_t fixed_L;       // 0 = random aggregation level
  uint32_t pct_rv;        // share of C-RNTI transport blocks sent with a random redundancy version (else rv 0)
  uint32_t pct_cqi_req;   // share of DCI 0 that request an aperiodic CSI report
  uint32_t pct_hop;       // share of subframes whose (then only) DCI 0 uses type-1 PUSCH frequency hopping (36.213 8.4.1)
  uint32_t pusch_hop_offset;  // SIB2 pusch-HoppingOffset of the cell
  uint32_t msg4_period;   // every n subframes one scheduled UE gets a contention-resolution PDU with an RRCConnectionSetup (0 = never)
  uint32_t msg4_p_a_idx;  // its pdsch-ConfigDedicated.p-a (0..7 = dB-6 .. dB3), 8 = random; the UE's PDSCH is sent with that power offset from the next subframe on
  // BCCH-DL-SCH messages sent with the SI-RNTI (sib_period != 0): message (sfn / 2) % 2 when its length is not 0, else random bytes
  uint32_t si_len[2];
  uint8_t si_msg[2][96];
  uint32_t pg_len;        // PCCH message sent with the P-RNTI (paging_period != 0) when not 0, else random bytes
  uint8_t pg_msg[96];
  uint32_t pct_harq;      // share of C-RNTI downlink grants that are sent again 8 subframes later (same HARQ process, NDI not toggled, next redundancy version of 0 2 3 1, same payload)
  uint32_t cp;            // 0 = normal cyclic prefix, 1 = extended (6 symbols per slot, CP = N / 4; 36.211 Table 6.2.3-1 / 6.12-1)
  // multipath fading (round 6): 0 = the static flat channel above; 1 / 2 / 3 = the tapped-delay-line profiles EPA / EVA / ETU of TS 36.101 Annex B.2.1 with an
  // independent Rayleigh process per (rx antenna, CRS port, tap) - "low correlation" - of maximum Doppler frequency doppler_hz (B.2.2: 5 / 70 / 300 Hz)
  uint32_t chan_model;
  float doppler_hz;
  float timing_offset_samples;  // a fractional sampling-time offset of the receiver, in samples of the cell's own rate (applied as extra delay of every tap)
  float cfo_drift_hz_per_s;     // the carrier offset moves: cfo_hz + cfo_drift_hz_per_s * t (a warming oscillator), phase continuous
} txg_cfg_t;

typedef struct { uint16_t rnti; uint8_t format, L; uint16_t ncce; uint32_t tti; uint32_t nbytes; uint32_t offset; uint8_t tb, mod, table256, is_ul; uint32_t nof_prb; uint32_t mcs; uint32_t cqi_req; uint32_t hop_bits_plus1; /* DCI 0: 0 = no hopping, else 1 + hopping bits */ } txg_pdu_t;

struct txg;
typedef struct txg txg_t;
txg_t* txg_new(const txg_cfg_t* cfg);
void txg_free(txg_t*);
int txg_next(txg_t*, float* iq, txg_pdu_t* pdus, int max_pdus, uint8_t* payload_buf, int payload_cap);
uint32_t txg_sf_len(const txg_t*);
uint32_t txg_tti(const txg_t*);
}

struct Ue { uint16_t rnti; int tm; bool t256; float p_a_db = 0.0f; };
struct RegInfo { std::vector<uint16_t> k0[3]; std::vector<uint8_t> l[3]; uint32_t nregs[3], ncce[3]; uint16_t pcfich_k0[4]; };

struct Grant;
struct Retx { uint32_t due; int ui; int nr; int rvk; std::vector<uint8_t> payload[2]; std::shared_ptr<Grant> proto; };
struct txg {
  txg_cfg_t c;
  std::vector<Retx> retx;  // HARQ retransmissions waiting for their subframe (pct_harq)
  int N, nre;
  Rng rng;
  std::vector<Ue> ues;
  RegInfo regs;
  uint32_t tti;
  uint64_t count = 0;      // subframes produced so far: seeds the noise of each subframe (independent of the scheduling stream)
  bool plan_only = false;  // txg_generate workers: walk the scheduler (every draw of `rng`) without synthesising the waveform
  cf h[2][4];
  // fading channel (chan_model != 0): per tap its delay in samples and amplitude; per (rx, port, tap) a sum of NSIN sinusoids (Jakes / Clarke: arrival angles
  // evenly spread with a random offset, random phases) evaluated on ABSOLUTE time, so that a capture rendered in parallel blocks is the one a sequential run renders
  static constexpr int NSIN = 16, FIR_HALF = 8;
  std::vector<double> tap_delay, tap_amp;
  std::vector<double> jk_w, jk_ph;   // [rx][port][tap][NSIN]: angular Doppler frequency (rad / s) and phase
  explicit txg(const txg_cfg_t& cfg) : c(cfg), rng(cfg.seed) {}
};

static int fft_size(uint32_t nprb) { switch (nprb) { case 6: return 128; case 15: return 256; case 25: return 512; case 50: return 1024; case 75: return 1536; case 100: return 2048; default: return -1; } }

// symbols per slot / subframe and the CRS symbols of ports 0, 1 (0 and N_symb - 3 of both slots) for the cell's cyclic prefix
static inline int nslot_of(const txg_cfg_t& c) { return c.cp ? 6 : 7; }
static inline int nsym_of(const txg_cfg_t& c) { return c.cp ? 12 : 14; }
static inline bool is_crs01(const txg_cfg_t& c, int l) { const int q = l % nslot_of(c); return q == 0 || q == nslot_of(c) - 3; }

static void build_regs(txg* g) {
  int nprb = g->c.nof_prb, nre = 12 * nprb, n0 = nre / 6, id = g->c.cell_id;
  std::vector<uint8_t> used0(n0, 0);
  int kbar = 6 * (id % (2 * nprb));
  for (int i = 0; i < 4; i++) { int k = (kbar + (i * nprb / 2) * 6) % nre; g->regs.pcfich_k0[i] = (uint16_t)k; used0[k / 6] = 1; }
  int ng = (g->c.phich_ng_x6 * nprb + 47) / 48;
  std::vector<int> avail;
  for (int i = 0; i < n0; i++) if (!used0[i]) avail.push_back(i);
  int na = (int)avail.size();
  for (int m = 0; m < ng; m++) for (int i = 0; i < 3; i++) used0[avail[(id + m + (i * na) / 3) % na]] = 1;
  for (int cfi = 1; cfi <= 3; cfi++) {
    int nsym = cfi + (nprb <= 10 ? 1 : 0);
    std::vector<uint16_t> tk; std::vector<uint8_t> tl;
    for (int k = 0; k < nre; k++) for (int l = 0; l < nsym; l++) {
      int w = (l == 0 || (l == 1 && g->c.nof_ports == 4) || (l == 3 && g->c.cp)) ? 6 : 4;  // symbol 1 carries the CRS of ports 2, 3; extended CP: symbol 3 is a CRS symbol
      if (k % w) continue;
      if (l == 0 && used0[k / 6]) continue;
      tk.push_back((uint16_t)k); tl.push_back((uint8_t)l);
    }
    int M = (int)tk.size(), R = (M + 31) / 32, ND = 32 * R - M;
    std::vector<int> perm;
    for (int j = 0; j < 32; j++) for (int r = 0; r < R; r++) { int idx = r * 32 + lsn_perm_cc[j]; if (idx >= ND) perm.push_back(idx - ND); }
    g->regs.nregs[cfi - 1] = M; g->regs.ncce[cfi - 1] = M / 9;
    g->regs.k0[cfi - 1].assign(M, 0); g->regs.l[cfi - 1].assign(M, 0);
    for (int mp = 0; mp < M; mp++) { int q = perm[(mp + id) % M]; g->regs.k0[cfi - 1][q] = tk[mp]; g->regs.l[cfi - 1][q] = tl[mp]; }
  }
}

extern "C" txg_t* txg_new(const txg_cfg_t* cfg) {
  if (fft_size(cfg->nof_prb) < 0) return nullptr;
  txg* g = new txg(*cfg);
  g->N = fft_size(cfg->nof_prb); g->nre = 12 * cfg->nof_prb; g->tti = cfg->start_tti;
  build_regs(g);
  // static channel: unit-ish gains with random phases, mild imbalance
  for (int r = 0; r < 2; r++) for (int p = 0; p < 2; p++) {
    double ph = g->rng.uni() * 2 * M_PI, mag = 0.7 + 0.5 * g->rng.uni();
    g->h[r][p] = cf((float)(mag * std::cos(ph)), (float)(mag * std::sin(ph)));
  }
  if (cfg->nof_ports == 4)  // drawn after the gains of ports 0, 1, so that captures with one or two ports do not change
    for (int r = 0; r < 2; r++) for (int p = 2; p < 4; p++) {
      double ph = g->rng.uni() * 2 * M_PI, mag = 0.7 + 0.5 * g->rng.uni();
      g->h[r][p] = cf((float)(mag * std::cos(ph)), (float)(mag * std::sin(ph)));
    }
  if (cfg->chan_model) {  // TS 36.101 Table B.2.1-2 / -3 / -4: excess tap delay (ns), relative power (dB)
    static const double epa[][2] = {{0, 0.0}, {30, -1.0}, {70, -2.0}, {90, -3.0}, {110, -8.0}, {190, -17.2}, {410, -20.8}};
    static const double eva[][2] = {{0, 0.0}, {30, -1.5}, {150, -1.4}, {310, -3.6}, {370, -0.6}, {710, -9.1}, {1090, -7.0}, {1730, -12.0}, {2510, -16.9}};
    static const double etu[][2] = {{0, -1.0}, {50, -1.0}, {120, -1.0}, {200, 0.0}, {230, 0.0}, {500, 0.0}, {1600, -3.0}, {2300, -5.0}, {5000, -7.0}};
    const double (*tab)[2] = cfg->chan_model == 1 ? epa : cfg->chan_model == 2 ? eva : etu;
    const int nt = cfg->chan_model == 1 ? 7 : 9;
    const double fs = 15000.0 * g->N;
    double tot = 0;
    for (int i = 0; i < nt; i++) tot += std::pow(10.0, tab[i][1] / 10.0);
    for (int i = 0; i < nt; i++) { g->tap_delay.push_back(tab[i][0] * 1e-9 * fs + (double)cfg->timing_offset_samples); g->tap_amp.push_back(std::sqrt(std::pow(10.0, tab[i][1] / 10.0) / tot)); }
    Rng fr((cfg->seed << 8) ^ 0xFAD1C0DEull);  // a stream of its own: the scheduling draws of a capture do not change with the channel model
    const double wd = 2 * M_PI * (double)cfg->doppler_hz;
    for (int r = 0; r < 2; r++) for (int p = 0; p < 4; p++) for (int i = 0; i < nt; i++) {
      const double a0 = fr.uni() * 2 * M_PI;
      for (int m = 0; m < txg::NSIN; m++) { g->jk_w.push_back(wd * std::cos(a0 + 2 * M_PI * (m + 0.5) / txg::NSIN + (fr.uni() - 0.5) * 2 * M_PI / txg::NSIN)); g->jk_ph.push_back(fr.uni() * 2 * M_PI); }
    }
  }
  for (uint32_t i = 0; i < cfg->n_rnti; i++) {
    Ue u; u.rnti = (uint16_t)(0x0100 + g->rng.below(0xFFF3 - 0x0100));
    bool dup = false; for (auto& o : g->ues) if (o.rnti == u.rnti) dup = true;
    if (dup) { i--; continue; }
    uint32_t r = g->rng.below(100);
    u.tm = (cfg->nof_ports < 2 || cfg->nof_rx < 2) ? 1 : r < cfg->mix_tm3_pct ? 3 : r < cfg->mix_tm3_pct + cfg->mix_tm4_pct ? 4 : 2;
    u.t256 = g->rng.below(100) < cfg->pct_256qam;
    g->ues.push_back(u);
  }
  return g;
}
extern "C" void txg_free(txg_t* g) { delete g; }
extern "C" uint32_t txg_sf_len(const txg_t* g) { return 15u * g->N; }
extern "C" uint32_t txg_tti(const txg_t* g) { return g->tti; }

// ---------------- DCI sizes / packing ----------------
static uint32_t f0raw(uint32_t n) { return 1 + 1 + riv_nbits(n) + 5 + 1 + 2 + 3 + 1; }
static uint32_t f1a_sz(uint32_t n) { uint32_t s = 1 + 1 + riv_nbits(n) + 5 + 3 + 1 + 2 + 2; while (s < f0raw(n)) s++; if (amb(s)) s++; return s; }
static uint32_t f0_sz(uint32_t n) { uint32_t s = f0raw(n); while (s < f1a_sz(n)) s++; return s; }
static uint32_t alloc_bits(uint32_t n) { return (n > 10 ? 1 : 0) + (n + ra_P(n) - 1) / ra_P(n); }
static uint32_t f1_sz(uint32_t n) { uint32_t s = alloc_bits(n) + 5 + 3 + 1 + 2 + 2; while (s == f0_sz(n) || s == f1a_sz(n) || amb(s)) s++; return s; }
static uint32_t f2_sz(uint32_t n, uint32_t ports) { uint32_t s = alloc_bits(n) + 2 + 3 + 1 + 16 + (ports == 2 ? 3 : ports == 4 ? 6 : 0); while (amb(s)) s++; return s; }
static uint32_t f2a_sz(uint32_t n, uint32_t ports) { uint32_t s = alloc_bits(n) + 2 + 3 + 1 + 16 + (ports == 4 ? 2 : 0); while (amb(s)) s++; return s; }

// RRCConnectionSetup (TS 36.331 6.2.2, UPER) as an eNB sends it in message 4: SRB1 with default RLC / logical channel configuration,
// explicit mac-MainConfig (optionally with the Rel-9 sr-ProhibitTimer extension group) and physicalConfigDedicated
static std::vector<uint8_t> rrc_conn_setup(Rng& rng, uint32_t p_a_idx, uint32_t b_ack, uint32_t b_ri, uint32_t b_cqi, int aper_mode /* -1 none */) {
  bits_t b;
  put(b, 0, 1); put(b, 3, 2);            // DL-CCCH-Message: c1, rrcConnectionSetup
  put(b, rng.below(4), 2);               // rrc-TransactionIdentifier
  put(b, 0, 1); put(b, 0, 3);            // criticalExtensions c1, rrcConnectionSetup-r8
  put(b, 0, 1);                          // no nonCriticalExtension
  put(b, 0, 1);                          // RadioResourceConfigDedicated: no extension
  put(b, 1, 1); put(b, 0, 1); put(b, 0, 1); put(b, 1, 1); put(b, 0, 1); put(b, 1, 1);  // srb list, -, -, mac-MainConfig, -, physicalConfigDedicated
  put(b, 0, 1);                          // one SRB
  const bool explicit_rlc = rng.below(2);
  put(b, 0, 1); put(b, 1, 1); put(b, 1, 1); put(b, 0, 1);  // SRB-ToAddMod: no ext, rlc-Config, logicalChannelConfig, srb-Identity 1
  if (explicit_rlc) { put(b, 0, 1); put(b, 0, 1); put(b, 0, 2); put(b, rng.below(55), 6); put(b, rng.below(8), 3); put(b, rng.below(15), 4); put(b, rng.below(8), 3); put(b, rng.below(31), 5); put(b, rng.below(56), 6); }
  else put(b, 1, 1);
  if (rng.below(2)) { put(b, 0, 1); put(b, 0, 1); put(b, 1, 1); put(b, 1, 1); put(b, rng.below(16), 4); put(b, rng.below(8), 4); put(b, rng.below(6), 3); put(b, rng.below(4), 2); }  // explicit: ul-SpecificParameters with group
  else put(b, 1, 1);
  const bool mac_ext = rng.below(2), drx = rng.below(2);
  put(b, 0, 1);                          // mac-MainConfig explicitValue
  put(b, mac_ext, 1); put(b, 1, 1); put(b, drx, 1); put(b, 1, 1);  // ext, ul-SCH-Config, drx-Config, phr-Config
  put(b, 1, 1); put(b, 1, 1); put(b, rng.below(14), 4); put(b, rng.below(14), 4); put(b, rng.below(6), 3); put(b, 0, 1);
  if (drx) { put(b, 1, 1); put(b, 1, 1); put(b, rng.below(16), 4); put(b, rng.below(22), 5); put(b, rng.below(8), 3); put(b, 7, 4); put(b, rng.below(160), 8); put(b, rng.below(16), 4); put(b, rng.below(16), 4); }  // setup with shortDRX, longDRX sf160
  put(b, rng.below(8), 3);               // timeAlignmentTimerDedicated
  put(b, 1, 1); put(b, rng.below(8), 3); put(b, rng.below(8), 3); put(b, rng.below(4), 2);  // phr-Config setup
  if (mac_ext) { put(b, 0, 1); put(b, 0, 6); put(b, 1, 1); put(b, 1, 8); put(b, 1, 1); put(b, rng.below(8), 3); put(b, 0, 4); }  // one addition group: [[ sr-ProhibitTimer-r9 ]] in one octet
  put(b, 0, 1);                          // PhysicalConfigDedicated: no extension
  const bool tpc = rng.below(2), srs = rng.below(2);
  put(b, 1, 1); put(b, 1, 1); put(b, 1, 1); put(b, 1, 1); put(b, tpc, 1); put(b, 0, 1); put(b, 1, 1); put(b, srs, 1); put(b, 1, 1); put(b, 1, 1);
  put(b, p_a_idx, 3);                    // pdsch-ConfigDedicated
  put(b, 0, 1); put(b, 0, 1);            // pucch-ConfigDedicated: no tdd mode, ackNackRepetition release
  put(b, b_ack, 4); put(b, b_ri, 4); put(b, b_cqi, 4);  // pusch-ConfigDedicated
  put(b, 0, 1); put(b, 8, 4); put(b, 0, 1); put(b, 1, 1); put(b, 8, 4); put(b, 3, 4);  // uplinkPowerControlDedicated, filterCoefficient default
  if (tpc) { put(b, 1, 1); put(b, rng.below(65536), 16); put(b, 0, 1); put(b, rng.below(15), 4); }  // tpc-PDCCH-ConfigPUCCH setup, indexOfFormat3
  put(b, aper_mode >= 0, 1); put(b, 1, 1);  // cqi-ReportConfig: aperiodic mode?, periodic
  if (aper_mode >= 0) put(b, (uint32_t)aper_mode, 3);
  put(b, 1, 3);                          // nomPDSCH-RS-EPRE-Offset 0
  put(b, 1, 1); put(b, 0, 1); put(b, rng.below(1186), 11); put(b, 38, 10); put(b, 0, 1); put(b, 1, 1);  // periodic setup
  if (srs) { put(b, 1, 1); put(b, rng.below(4), 2); put(b, rng.below(4), 2); put(b, rng.below(24), 5); put(b, 1, 1); put(b, rng.below(1024), 10); put(b, rng.below(2), 1); put(b, rng.below(8), 3); }
  put(b, 0, 1); put(b, 0, 1); put(b, 1, 3); put(b, 0, 1);  // antennaInfo explicit: tm2, no codebook restriction, antenna selection release
  put(b, 1, 1); put(b, rng.below(2048), 11); put(b, 15, 8); put(b, 4, 3);  // schedulingRequestConfig setup
  while (b.size() % 8) b.push_back(0);
  std::vector<uint8_t> out(b.size() / 8, 0);
  for (size_t i = 0; i < b.size(); i++) out[i / 8] |= (uint8_t)(b[i] << (7 - i % 8));
  return out;
}

struct Grant {
  uint16_t rnti; int format; int L; int ncce;
  bool type0; uint32_t rbg_mask; uint32_t riv; std::vector<int> prbs;
  int ntb; uint32_t mcs[2]; int rv[2]; uint32_t ndi[2]; uint32_t pid; uint32_t pinfo; uint32_t swap;
  bool t256; bool nprb1a_is2; int tbs[2]; int qm[2];
  int scheme;  // 0 port0, 1 div, 2 SM, 3 CDD
  int pmi, nlayers;
  bool is_ul;
  uint32_t cqi_req = 0;
  int hop_bits = -1;  // DCI 0: hopping bits (36.213 Table 8.4-2), -1 = no hopping
};

static bits_t dci_pack(const txg* g, const Grant& gr) {
  uint32_t n = g->c.nof_prb;
  bits_t b;
  bool user = gr.rnti >= 0x000B && gr.rnti <= 0xFFF3;
  switch (gr.format) {
    case TXG_FMT0:
      put(b, 0, 1);
      if (gr.hop_bits < 0) { put(b, 0, 1); put(b, gr.riv, riv_nbits(n)); }
      else { const uint32_t nh = n < 50 ? 1 : 2; put(b, 1, 1); put(b, (uint32_t)gr.hop_bits, nh); put(b, gr.riv, riv_nbits(n) - nh); }
      put(b, gr.mcs[0], 5); put(b, gr.ndi[0], 1); put(b, 1, 2); put(b, 0, 3); put(b, gr.cqi_req, 1);
      while (b.size() < f0_sz(n)) b.push_back(0);
      break;
    case TXG_FMT1A:
      put(b, 1, 1); put(b, 0, 1); put(b, gr.riv, riv_nbits(n)); put(b, gr.mcs[0], 5); put(b, gr.pid, 3);
      put(b, user ? gr.ndi[0] : 0, 1); put(b, (uint32_t)gr.rv[0], 2);
      if (user) put(b, 1, 2); else { put(b, 0, 1); put(b, gr.nprb1a_is2 ? 0 : 1, 1); }
      while (b.size() < f1a_sz(n)) b.push_back(0);
      break;
    case TXG_FMT1:
      if (n > 10) put(b, 0, 1);
      put(b, gr.rbg_mask, (n + ra_P(n) - 1) / ra_P(n)); put(b, gr.mcs[0], 5); put(b, gr.pid, 3); put(b, gr.ndi[0], 1); put(b, (uint32_t)gr.rv[0], 2); put(b, 1, 2);
      while (b.size() < f1_sz(n)) b.push_back(0);
      break;
    case TXG_FMT2:
    case TXG_FMT2A:
      if (n > 10) put(b, 0, 1);
      put(b, gr.rbg_mask, (n + ra_P(n) - 1) / ra_P(n)); put(b, 1, 2); put(b, gr.pid, 3); put(b, gr.swap, 1);
      for (int i = 0; i < 2; i++) {
        if (i < gr.ntb) { put(b, gr.mcs[i], 5); put(b, gr.ndi[i], 1); put(b, (uint32_t)gr.rv[i], 2); }
        else { put(b, 0, 5); put(b, 0, 1); put(b, 1, 2); }  // disabled TB: mcs 0, rv 1
      }
      if (gr.format == TXG_FMT2) put(b, gr.pinfo, g->c.nof_ports == 2 ? 3 : 6);
      else if (g->c.nof_ports == 4) put(b, gr.pinfo, 2);
      while (b.size() < (gr.format == TXG_FMT2 ? f2_sz(n, g->c.nof_ports) : f2a_sz(n, g->c.nof_ports))) b.push_back(0);
      break;
    default: break;
  }
  return b;
}

// ---------------- search space ----------------
static void ss_candidates(uint32_t ncce, uint32_t sf, uint16_t rnti, int l, bool common, std::vector<int>& out) {
  out.clear();
  uint32_t L = 1u << l;
  if (ncce < L) return;
  if (common) { for (uint32_t i = 0; i < std::min<uint32_t>(ncce, 16) / L; i++) out.push_back((int)(L * (i % (ncce / L)))); return; }
  static const uint32_t nc[4] = {6, 6, 2, 2};
  uint32_t Yk = rnti;
  for (uint32_t m = 0; m < sf + 1; m++) Yk = (39827u * Yk) % 65537u;
  for (uint32_t i = 0; i < nc[l]; i++) { uint32_t n = L * ((Yk + i) % (ncce / L)); if (n + L <= ncce) out.push_back((int)n); }
}

static bool pdsch_re_ok(const txg* g, uint32_t sf, int l, int k) {
  int nprb = g->c.nof_prb, id = g->c.cell_id;
  const int nsl = nslot_of(g->c), lq = l % nsl;
  if (g->c.nof_ports == 4 && lq == 1 && k % 3 == id % 3) return false;
  if (is_crs01(g->c, l)) {
    if (g->c.nof_ports >= 2) { if (k % 3 == id % 3) return false; }
    else { int v = lq == 0 ? 0 : 3; if (k % 6 == (v + id % 6) % 6) return false; }
  }
  int kc0 = 6 * nprb - 36;
  if (k >= kc0 && k < kc0 + 72) {
    if ((sf == 0 || sf == 5) && (l == nsl - 2 || l == nsl - 1)) return false;
    if (sf == 0 && l >= nsl && l <= nsl + 3) return false;
  }
  return true;
}

static void sfbc_pair(cf x0, cf x1, cf* p0, cf* p1) {
  const float s = 0.70710678f;
  p0[0] = x0 * s; p0[1] = x1 * s; p1[0] = -std::conj(x1) * s; p1[1] = std::conj(x0) * s;
}

extern "C" int txg_next(txg_t* g, float* iq, txg_pdu_t* pdus, int max_pdus, uint8_t* pbuf, int pcap) {
  const txg_cfg_t& c = g->c;
  int nprb = c.nof_prb, nre = g->nre, N = g->N, P = c.nof_ports, id = c.cell_id;
  uint32_t tti = g->tti, sf = tti % 10, sfn = (tti / 10) % 1024;
  uint32_t cfi = c.cfi ? c.cfi : 1 + g->rng.below(3);
  uint32_t ncce = g->regs.ncce[cfi - 1];
  std::vector<std::vector<cf>> grid(P, std::vector<cf>(14 * nre, cf(0, 0)));
  int npdu = 0, poff = 0;

  // ---- CRS ----
  const int nsl = nslot_of(c), nsym = nsym_of(c);
  const uint32_t ncp = c.cp ? 0u : 1u;  // N_CP of the CRS sequence initialisation (36.211 6.10.1.1)
  for (int s = 0; s < 4; s++) {
    int l = (s >> 1) * nsl + ((s & 1) ? nsl - 3 : 0);
    uint32_t ns = 2 * sf + (l >= nsl), lsl = l % nsl;
    bits_t cc = gold(1024u * (7u * (ns + 1) + lsl + 1) * (2u * id + 1) + 2u * id + ncp, 440);
    for (int p = 0; p < std::min(P, 2); p++) {
      int v = p == 0 ? ((s & 1) ? 3 : 0) : ((s & 1) ? 0 : 3), koff = (v + id % 6) % 6;
      for (int m = 0; m < 2 * nprb; m++) {
        int mp = m + 110 - nprb;
        grid[p][l * nre + 6 * m + koff] = cf((1 - 2 * cc[2 * mp]) * 0.70710678f, (1 - 2 * cc[2 * mp + 1]) * 0.70710678f);
      }
    }
  }
  if (P == 4)  // ports 2, 3: symbol 1 of both slots, v = 3 (n_s mod 2) / 3 + 3 (n_s mod 2)
    for (int s = 0; s < 2; s++) {
      int l = s * nsl + 1;
      uint32_t ns = 2 * sf + (uint32_t)s;
      bits_t cc = gold(1024u * (7u * (ns + 1) + 1 + 1) * (2u * id + 1) + 2u * id + ncp, 440);
      for (int p = 2; p < 4; p++) {
        int v = (p == 2 ? 0 : 3) + 3 * s, koff = (v + id % 6) % 6;
        for (int m = 0; m < 2 * nprb; m++) {
          int mp = m + 110 - nprb;
          grid[p][l * nre + 6 * m + koff] = cf((1 - 2 * cc[2 * mp]) * 0.70710678f, (1 - 2 * cc[2 * mp + 1]) * 0.70710678f);
        }
      }
    }
  // transmit diversity of one symbol pair (the i-th of its channel) on two resource elements: SFBC on ports (0, 1); with four ports SFBC-FSTD
  // (36.211 6.3.4.3): even pairs on ports (0, 2), odd pairs on ports (1, 3)
  auto put_pair = [&](int i, cf x0, cf x1, int la, int ka, int lb, int kb, float amp) {
    cf p0[2], p1[2]; sfbc_pair(x0, x1, p0, p1);
    const int pa = (P == 4 && (i & 1)) ? 1 : 0, pb = P == 4 ? pa + 2 : 1;
    grid[pa][la * nre + ka] = p0[0] * amp; grid[pa][lb * nre + kb] = p0[1] * amp;
    grid[pb][la * nre + ka] = p1[0] * amp; grid[pb][lb * nre + kb] = p1[1] * amp;
  };
  auto map_quad = [&](int k0, int l, const cf* x) {  // 4 symbols onto the data REs of a REG
    int kk[4], n = 0;
    if (l == 0 || (l == 1 && P == 4) || (l == 3 && c.cp)) { for (int k = k0; k < k0 + 6; k++) if (k % 3 != id % 3) kk[n++] = k; }
    else for (int k = k0; k < k0 + 4; k++) kk[n++] = k;
    if (P == 1) { for (int i = 0; i < 4; i++) grid[0][l * nre + kk[i]] = x[i]; }
    else for (int i = 0; i < 4; i += 2) put_pair(i / 2, x[i], x[i + 1], l, kk[i], l, kk[i + 1], 1.0f);
  };
  // ---- PCFICH ----
  {
    static const char* cw[3] = {"01101101101101101101101101101101", "10110110110110110110110110110110", "11011011011011011011011011011011"};
    bits_t sc = gold((sf + 1) * (2u * id + 1) * 512u + id, 32), b(32);
    for (int i = 0; i < 32; i++) b[i] = (cw[cfi - 1][i] == '1') ^ sc[i];
    std::vector<cf> sy; modulate(b, 2, sy);
    for (int i = 0; i < 4; i++) map_quad(g->regs.pcfich_k0[i], 0, &sy[4 * i]);
  }

  // ---- PBCH (36.211 6.6, 36.212 5.3.1): MIB of this radio frame's 40 ms period, quarter sfn % 4, on subframe 0 ----
  if (sf == 0) {
    bits_t mib;
    static const int bwidx[6] = {6, 15, 25, 50, 75, 100};
    int bi = 0; for (int i = 0; i < 6; i++) if (bwidx[i] == nprb) bi = i;
    put(mib, (uint32_t)bi, 3);
    put(mib, 0, 1);  // phich-Duration normal
    put(mib, c.phich_ng_x6 == 1 ? 0u : c.phich_ng_x6 == 3 ? 1u : c.phich_ng_x6 == 6 ? 2u : 3u, 2);
    put(mib, (sfn >> 2) & 0xFF, 8);
    put(mib, 0, 10);
    crc_attach(mib, 0x11021, 16, P == 1 ? 0x0000u : (P == 2 ? 0xFFFFu : 0x5555u));
    // normal CP: 240 symbols per radio frame (CRS positions of four ports left out of symbols 0, 1 of slot 1), E = 1920 bits per 40 ms;
    // extended CP: symbol 3 of the slot carries CRS as well -> 216 symbols, E = 1728 (36.211 6.6.4, 36.212 5.3.1.3)
    std::vector<std::pair<int, int>> pos;
    for (int l = nsl; l <= nsl + 3; l++)
      for (int k = 6 * nprb - 36; k < 6 * nprb + 36; k++)
        if (!((l <= nsl + 1 || (c.cp && l == nsl + 3)) && k % 3 == id % 3)) pos.push_back({l, k});
    const int nps = (int)pos.size(), E4 = 2 * nps;
    bits_t e = rm_conv_tx(conv_encode(mib), 4 * E4), scr = gold((uint32_t)id, 4 * E4), q(E4);
    for (int i = 0; i < E4; i++) q[i] = e[E4 * (sfn & 3) + i] ^ scr[E4 * (sfn & 3) + i];
    std::vector<cf> sy; modulate(q, 2, sy);
    if (P == 1) { for (int i = 0; i < nps; i++) grid[0][pos[i].first * nre + pos[i].second] = sy[i]; }
    else for (int i = 0; i < nps; i += 2) put_pair(i / 2, sy[i], sy[i + 1], pos[i].first, pos[i].second, pos[i + 1].first, pos[i + 1].second, 1.0f);
  }

  // ---- PSS / SSS (36.211 6.11, FDD): symbols 6 / 5 of subframes 0 and 5, 62 carriers around DC, antenna port 0 ----
  if (sf == 0 || sf == 5) {
    static const int root[3] = {25, 29, 34};
    const int n2 = id % 3, n1 = id / 3, u = root[n2];
    int xs[31], xc[31], xz[31];
    for (int i = 0; i < 5; i++) xs[i] = xc[i] = xz[i] = (i == 4);
    for (int i = 0; i < 26; i++) {
      xs[i + 5] = (xs[i + 2] + xs[i]) & 1;
      xc[i + 5] = (xc[i + 3] + xc[i]) & 1;
      xz[i + 5] = (xz[i + 4] + xz[i + 2] + xz[i + 1] + xz[i]) & 1;
    }
    const int qp = n1 / 30, q = (n1 + qp * (qp + 1) / 2) / 30, mp = n1 + q * (q + 1) / 2;
    const int m0 = mp % 31, m1 = (m0 + mp / 31 + 1) % 31;
    for (int n = 0; n < 62; n++) {
      const int k = 6 * nprb - 31 + n;
      const int a = n < 31 ? n * (n + 1) : (n + 1) * (n + 2);
      const double ph = -M_PI * u * (double)(a % 126) / 63.0;
      grid[0][(nsl - 1) * nre + k] = cf((float)std::cos(ph), (float)std::sin(ph));   // PSS: last symbol of slots 0 and 10
      const int i = n / 2;
      const int s0 = 1 - 2 * xs[(i + m0) % 31], s1 = 1 - 2 * xs[(i + m1) % 31];
      const int c0 = 1 - 2 * xc[(i + n2) % 31], c1 = 1 - 2 * xc[(i + n2 + 3) % 31];
      const int z0 = 1 - 2 * xz[(i + m0 % 8) % 31], z1 = 1 - 2 * xz[(i + m1 % 8) % 31];
      int d;
      if (n % 2 == 0) d = (sf == 0 ? s0 : s1) * c0;
      else d = sf == 0 ? s1 * c1 * z0 : s0 * c1 * z1;
      grid[0][(nsl - 2) * nre + k] = cf((float)d, 0.0f);   // SSS: the symbol in front of it
    }
  }

  // ---- schedule ----
  std::vector<Grant> grants;
  std::vector<uint8_t> cce_used(ncce, 0);
  auto place = [&](Grant& gr, bool common) -> bool {
    std::vector<int> cand;
    for (int attempt = 0; attempt < 4; attempt++) {
      int l = gr.L;
      if (common && l < 2) l = 2;
      ss_candidates(ncce, sf, gr.rnti, l, common, cand);
      for (int nc : cand) {
        bool freec = true;
        for (int i = 0; i < (1 << l); i++) if (cce_used[nc + i]) freec = false;
        if (freec) { for (int i = 0; i < (1 << l); i++) cce_used[nc + i] = 1; gr.ncce = nc; gr.L = l; return true; }
      }
      gr.L = (gr.L + 3) % 4;  // try another level
    }
    return false;
  };
  auto pickL = [&]() -> int { if (c.fixed_L) return (int)c.fixed_L - 1; uint32_t r = g->rng.below(10); return r < 4 ? 0 : r < 7 ? 1 : r < 9 ? 2 : 3; };
  int l0 = cfi + (nprb <= 10 ? 1 : 0);
  uint32_t Prbg = ra_P(nprb), nrbg = (nprb + Prbg - 1) / Prbg;
  int next_rbg = 0;  // RBGs handed out left to right

  auto count_re = [&](const std::vector<int>& prbs) { int n = 0; for (int l = l0; l < nsym; l++) for (int p : prbs) for (int k = 12 * p; k < 12 * p + 12; k++) n += pdsch_re_ok(g, sf, l, k); return n; };
  auto set_tbs = [&](Grant& gr, int nre_g) {
    for (int i = 0; i < gr.ntb; i++) {
      int mcs = (int)gr.mcs[i];
      for (;; mcs--) {
        const int8_t(*t)[2] = gr.t256 ? lsn_mcs_dl_256qam : lsn_mcs_dl_64qam;
        int itbs = t[mcs][1]; gr.qm[i] = t[mcs][0];
        gr.tbs[i] = lsn_tbs_table[itbs][gr.prbs.size() - 1];
        double rate = (gr.tbs[i] + 24.0) / ((double)nre_g * gr.qm[i]);
        if (rate <= 0.88 || mcs == 0) break;
      }
      gr.mcs[i] = (uint32_t)mcs;
    }
  };
  // broadcast-type grants first (common search space, DCI 1A, 64QAM table, QPSK)
  auto add_common = [&](uint16_t rnti, int nbytes_hint, const uint8_t* fixed_payload, int fixed_len) {
    if (next_rbg + 3 > (int)nrbg) return;
    Grant gr{}; gr.rnti = rnti; gr.format = TXG_FMT1A; gr.L = 2; gr.type0 = false; gr.ntb = 1; gr.is_ul = false;
    int start = next_rbg * (int)Prbg, Lcrb = 3 * (int)Prbg;
    if (start + Lcrb > nprb) return;
    gr.riv = (uint32_t)(nprb * (Lcrb - 1) + start);
    if (Lcrb - 1 > nprb / 2) gr.riv = (uint32_t)(nprb * (nprb - Lcrb + 1) + (nprb - 1 - start));
    for (int i = 0; i < Lcrb; i++) gr.prbs.push_back(start + i);
    gr.nprb1a_is2 = false; gr.mcs[0] = 2 + g->rng.below(6); gr.rv[0] = 0; gr.t256 = false;
    if (fixed_payload)  // the block must hold the whole message
      while (gr.mcs[0] < 26 && lsn_tbs_table[gr.mcs[0]][2] < 8 * fixed_len) gr.mcs[0]++;
    gr.tbs[0] = lsn_tbs_table[gr.mcs[0]][2]; gr.qm[0] = 2; gr.scheme = P == 1 ? 0 : 1; gr.nlayers = P;
    (void)nbytes_hint;
    if (!place(gr, true)) return;
    next_rbg += 3;
    grants.push_back(gr);
    // payload
    int nb = gr.tbs[0] / 8;
    if (npdu < max_pdus && poff + nb <= pcap) {
      for (int i = 0; i < nb; i++) pbuf[poff + i] = (uint8_t)g->rng.u32();
      if (fixed_payload) { memset(pbuf + poff, 0, nb); memcpy(pbuf + poff, fixed_payload, std::min(nb, fixed_len)); }
      txg_pdu_t& pd = pdus[npdu++];
      pd = txg_pdu_t{rnti, (uint8_t)gr.format, (uint8_t)gr.L, (uint16_t)gr.ncce, tti, (uint32_t)nb, (uint32_t)poff, 0, 2, 0, 0, (uint32_t)gr.prbs.size(), gr.mcs[0]};
      poff += nb;
    }
  };
  if (c.sib_period && sf == 5 && (sfn % 2) == 0) {
    const uint32_t mi = (sfn / 2) % 2;
    if (c.si_len[mi]) add_common(0xFFFF, 0, c.si_msg[mi], (int)std::min<uint32_t>(c.si_len[mi], 96));
    else add_common(0xFFFF, 0, nullptr, 0);
  }
  if (c.paging_period && (tti % c.paging_period) == 3) {
    if (c.pg_len) add_common(0xFFFE, 0, c.pg_msg, (int)std::min<uint32_t>(c.pg_len, 96));
    else add_common(0xFFFE, 0, nullptr, 0);
  }
  if (c.rar_period && (tti % c.rar_period) == 7) {
    // MAC RAR PDU: one RAPID subheader + one RAR with a fresh temporary C-RNTI
    uint16_t t_crnti = (uint16_t)(0x0100 + g->rng.below(0xFFF3 - 0x0100));
    uint8_t rar[7] = {(uint8_t)(0x40 | g->rng.below(64)), 0x00, 0x10, 0x0c, 0x00, (uint8_t)(t_crnti >> 8), (uint8_t)t_crnti};
    add_common((uint16_t)(2 + g->rng.below(8)), 0, rar, 7);
    if (!g->ues.empty()) {  // the new UE replaces a random old one and is scheduled from now on
      Ue u = g->ues[g->rng.below((uint32_t)g->ues.size())]; u.rnti = t_crnti;
      g->ues[g->rng.below((uint32_t)g->ues.size())] = u;
    }
  }
  // unicast DL
  int msg4_ue = -1;
  float msg4_p_a = 0.0f;
  uint32_t kdl = c.dl_min + g->rng.below(c.dl_max - c.dl_min + 1);
  std::vector<int> picked;
  static const int rv_seq[4] = {0, 2, 3, 1};
  // HARQ retransmissions due now (pct_harq): the grant of 8 subframes ago again - same process, NDI, MCS and number of resource block groups, next
  // redundancy version, same transport blocks - placed in front of the new grants
  auto maybe_retx = [&](const Grant& gr, int ui, int nr, int rvk, const uint8_t* p0, const uint8_t* p1) {
    if (!c.pct_harq || rvk >= 3 || !(gr.rnti >= 0x000B && gr.rnti <= 0xFFF3) || g->rng.below(100) >= c.pct_harq) return;
    Retx r; r.due = (tti + 8) % 10240; r.ui = ui; r.nr = nr; r.rvk = rvk + 1; r.proto = std::make_shared<Grant>(gr);
    if (p0) r.payload[0].assign(p0, p0 + gr.tbs[0] / 8);
    if (p1 && gr.ntb > 1) r.payload[1].assign(p1, p1 + gr.tbs[1] / 8);
    g->retx.push_back(std::move(r));
  };
  for (size_t ri = 0; ri < g->retx.size();) {
    if (g->retx[ri].due != tti) { ri++; continue; }
    Retx r = g->retx[ri];
    g->retx.erase(g->retx.begin() + (long)ri);
    if (next_rbg + r.nr > (int)nrbg || r.ui >= (int)g->ues.size() || g->ues[r.ui].rnti != r.proto->rnti) continue;  // no room / the UE has left
    Grant gr = *r.proto;
    gr.L = pickL(); gr.prbs.clear();
    const int r0 = next_rbg;
    for (int q = r0; q < r0 + r.nr; q++) for (int p = q * (int)Prbg; p < (q + 1) * (int)Prbg && p < nprb; p++) gr.prbs.push_back(p);
    gr.rbg_mask = 0;
    for (int q = r0; q < r0 + r.nr; q++) gr.rbg_mask |= 1u << (nrbg - 1 - q);
    if (!gr.type0) { int start = gr.prbs.front(), Lc = (int)gr.prbs.size(); gr.riv = (Lc - 1 <= nprb / 2) ? (uint32_t)(nprb * (Lc - 1) + start) : (uint32_t)(nprb * (nprb - Lc + 1) + (nprb - 1 - start)); }
    if ((int)gr.prbs.size() != (int)r.proto->prbs.size()) continue;  // the last group of the band is shorter: the block size would change
    for (int i = 0; i < gr.ntb; i++) gr.rv[i] = rv_seq[r.rvk];
    const int nre_g = count_re(gr.prbs);
    if (nre_g < 24) continue;
    set_tbs(gr, nre_g);
    if (gr.tbs[0] != r.proto->tbs[0] || (gr.ntb > 1 && gr.tbs[1] != r.proto->tbs[1])) continue;
    if (!place(gr, false)) continue;
    next_rbg += r.nr;
    picked.push_back(r.ui);
    grants.push_back(gr);
    const uint8_t* pp[2] = {nullptr, nullptr};
    for (int i = 0; i < gr.ntb; i++) {
      const int nb = gr.tbs[i] / 8;
      if (npdu < max_pdus && poff + nb <= pcap && (int)r.payload[i].size() == nb) {
        memcpy(pbuf + poff, r.payload[i].data(), (size_t)nb);
        pp[i] = pbuf + poff;
        txg_pdu_t& pd = pdus[npdu++];
        pd = txg_pdu_t{gr.rnti, (uint8_t)gr.format, (uint8_t)gr.L, (uint16_t)gr.ncce, tti, (uint32_t)nb, (uint32_t)poff, (uint8_t)i, (uint8_t)gr.qm[i], (uint8_t)gr.t256, 0, (uint32_t)gr.prbs.size(), gr.mcs[i]};
        poff += nb;
      }
    }
    maybe_retx(gr, r.ui, r.nr, r.rvk, pp[0], pp[1]);
  }
  int rbg_left = (int)nrbg - next_rbg;
  if ((int)kdl > rbg_left) kdl = (uint32_t)std::max(rbg_left, 0);
  std::vector<int> share(kdl, 1);
  for (int i = 0; i < rbg_left - (int)kdl; i++) share[g->rng.below(kdl)]++;
  for (uint32_t q = 0; q < kdl && !g->ues.empty(); q++) {
    int ui; bool dupl;
    int tries = 0;
    do { ui = (int)g->rng.below((uint32_t)g->ues.size()); dupl = false; for (int p : picked) if (p == ui) dupl = true; } while (dupl && ++tries < 20);
    if (dupl) continue;
    picked.push_back(ui);
    const Ue& u = g->ues[ui];
    Grant gr{}; gr.rnti = u.rnti; gr.L = pickL(); gr.is_ul = false; gr.t256 = u.t256; gr.pid = g->rng.below(8);
    int r0 = next_rbg, nr = share[q];
    next_rbg += nr;
    for (int r = r0; r < r0 + nr; r++) for (int p = r * (int)Prbg; p < (r + 1) * (int)Prbg && p < nprb; p++) gr.prbs.push_back(p);
    gr.type0 = true; gr.rbg_mask = 0;
    for (int r = r0; r < r0 + nr; r++) gr.rbg_mask |= 1u << (nrbg - 1 - r);
    uint32_t mlo = c.mcs_min, mhi = std::min<uint32_t>(c.mcs_max, u.t256 ? 27 : 28);
    if (mlo > mhi) mlo = mhi;
    if (u.tm == 3 || u.tm == 4) {
      gr.format = u.tm == 3 ? TXG_FMT2A : TXG_FMT2; gr.ntb = 2; gr.swap = 0;
      if (u.tm == 3) { gr.scheme = 3; gr.nlayers = 2; gr.pinfo = 0; }
      else { gr.scheme = 2; gr.nlayers = 2; gr.pinfo = g->rng.below(2); gr.pmi = (int)gr.pinfo; }
      // four ports: the receivers under test (like the reference's srsRAN) decode transmit diversity only; such a grant still goes out on the
      // PDCCH (format 2 / 2A at their four-port sizes) with a diversity waveform of the first block on its PRBs, and no receiver decodes it
      if (P == 4) gr.scheme = 4;
      for (int i = 0; i < 2; i++) {
        gr.mcs[i] = mlo + g->rng.below(mhi - mlo + 1); gr.rv[i] = 0; gr.ndi[i] = g->rng.below(2);
        if (c.pct_rv && g->rng.below(100) < c.pct_rv) { gr.rv[i] = (int)g->rng.below(4); if (gr.mcs[i] == 0 && gr.rv[i] == 1) gr.rv[i] = 2; }  // (mcs 0, rv 1) = TB disabled
      }
    } else {
      bool f1a = g->rng.below(4) == 0;
      gr.format = f1a ? TXG_FMT1A : TXG_FMT1; gr.ntb = 1; gr.scheme = P == 1 ? 0 : 1; gr.nlayers = P;
      gr.mcs[0] = mlo + g->rng.below(mhi - mlo + 1); gr.rv[0] = 0; gr.ndi[0] = g->rng.below(2);
      if (c.pct_rv && g->rng.below(100) < c.pct_rv) gr.rv[0] = (int)g->rng.below(4);
      if (f1a) {
        gr.type0 = false; gr.t256 = false;  // 1A always uses the 64QAM table
        if (gr.mcs[0] > 28) gr.mcs[0] = 28;
        int start = gr.prbs.front(), Lc = (int)gr.prbs.size();
        gr.riv = (Lc - 1 <= nprb / 2) ? (uint32_t)(nprb * (Lc - 1) + start) : (uint32_t)(nprb * (nprb - Lc + 1) + (nprb - 1 - start));
      }
    }
    int nre_g = count_re(gr.prbs);
    if (nre_g < 24) continue;
    set_tbs(gr, nre_g);
    if (!place(gr, false)) continue;
    grants.push_back(gr);
    const uint8_t* new_pp[2] = {nullptr, nullptr};
    for (int i = 0; i < gr.ntb; i++) {
      int nb = gr.tbs[i] / 8;
      if (npdu < max_pdus && poff + nb <= pcap) {
        new_pp[i] = pbuf + poff;
        for (int b = 0; b < nb; b++) pbuf[poff + b] = (uint8_t)g->rng.u32();
        pbuf[poff] = 0x03;  // a well-formed MAC PDU: one subheader (E = 0, LCID 3 = a DTCH), the SDU takes the rest; never an all-zero TB
        if (i == 0 && q == 0 && c.msg4_period && (tti % c.msg4_period) == 5 % c.msg4_period && msg4_ue < 0) {
          // message 4: UE contention resolution identity + CCCH SDU (RRCConnectionSetup) + padding (TS 36.321 6.1.2), sent with the
          // power offset in force so far; the new p-a applies to this UE from the next subframe on
          static const float p_a_db[8] = {-6.0f, -4.77f, -3.0f, -1.77f, 0.0f, 1.0f, 2.0f, 3.0f};
          const uint32_t pi = c.msg4_p_a_idx < 8 ? c.msg4_p_a_idx : g->rng.below(8);
          const int am = g->rng.below(3) == 0 ? -1 : (int)g->rng.below(5);
          std::vector<uint8_t> rrc = rrc_conn_setup(g->rng, pi, g->rng.below(15), g->rng.below(13), 2 + g->rng.below(14), am);
          if ((int)rrc.size() + 10 <= nb && rrc.size() < 128) {
            memset(pbuf + poff, 0, (size_t)nb);
            pbuf[poff] = 0x3C; pbuf[poff + 1] = 0x20; pbuf[poff + 2] = (uint8_t)rrc.size(); pbuf[poff + 3] = 0x1F;
            for (int b = 0; b < 6; b++) pbuf[poff + 4 + b] = (uint8_t)g->rng.u32();
            memcpy(pbuf + poff + 10, rrc.data(), rrc.size());
            msg4_ue = ui; msg4_p_a = p_a_db[pi];
          }
        }
        txg_pdu_t& pd = pdus[npdu++];
        pd = txg_pdu_t{gr.rnti, (uint8_t)gr.format, (uint8_t)gr.L, (uint16_t)gr.ncce, tti, (uint32_t)nb, (uint32_t)poff, (uint8_t)i, (uint8_t)gr.qm[i], (uint8_t)gr.t256, 0, (uint32_t)gr.prbs.size(), gr.mcs[i]};
        poff += nb;
      }
    }
    if (c.pct_harq && new_pp[0] && (gr.ntb < 2 || new_pp[1])) maybe_retx(gr, ui, nr, 0, new_pp[0], new_pp[1]);
  }
  // UL grants (DCI 0 only; PUSCH itself is not generated)
  uint32_t kul = c.ul_min + g->rng.below(c.ul_max - c.ul_min + 1);
  int ul_next = 0;
  for (uint32_t q = 0; q < kul && !g->ues.empty(); q++) {
    const Ue& u = g->ues[g->rng.below((uint32_t)g->ues.size())];
    bool dupl = false; for (auto& o : grants) if (o.rnti == u.rnti && o.is_ul) dupl = true;
    if (dupl) continue;
    Grant gr{}; gr.rnti = u.rnti; gr.format = TXG_FMT0; gr.L = pickL(); gr.is_ul = true; gr.ntb = 0;
    int Lc = 2 + (int)g->rng.below(8), start = ul_next; ul_next += Lc;
    if (q == 0 && c.pct_hop && g->rng.below(100) < c.pct_hop) {
      // a hopping grant: the only uplink grant of this subframe, allocation in the lower quarter so that both slots fit and the
      // shortened RIV field holds it; hopping bits: 1-bit field 0 = +N/2, 2-bit field 0 / 2 = +N/4 / +N/2 (type 1)
      const uint32_t nh = nprb < 50 ? 1 : 2;
      Lc = 3 + (int)g->rng.below(4);
      start = (int)((c.pusch_hop_offset + 1) / 2) + (int)g->rng.below(4);
      gr.hop_bits = nh == 1 ? 0 : (g->rng.below(2) ? 2 : 0);
      kul = 1;
      if ((uint32_t)(nprb * (Lc - 1) + start) >= (1u << (riv_nbits((uint32_t)nprb) - nh))) gr.hop_bits = -1;
    }
    if (start + Lc > nprb) break;
    gr.riv = (Lc - 1 <= nprb / 2) ? (uint32_t)(nprb * (Lc - 1) + start) : (uint32_t)(nprb * (nprb - Lc + 1) + (nprb - 1 - start));
    gr.mcs[0] = g->rng.below(25); gr.ndi[0] = g->rng.below(2);
    if (c.pct_cqi_req) gr.cqi_req = g->rng.below(100) < c.pct_cqi_req ? 1u : 0u;
    if (!place(gr, false)) continue;
    grants.push_back(gr);
    if (npdu < max_pdus) { txg_pdu_t& pd = pdus[npdu++]; pd = txg_pdu_t{gr.rnti, 0, (uint8_t)gr.L, (uint16_t)gr.ncce, tti, 0, (uint32_t)start /* UL grants: offset = first PRB */, 0, 0, 0, 1, (uint32_t)Lc, gr.mcs[0], gr.cqi_req, (uint32_t)(gr.hop_bits + 1)}; }
  }

  if (g->plan_only) {  // scheduler state advanced exactly as a rendering call would (the waveform below draws nothing from `rng`)
    if (msg4_ue >= 0) g->ues[msg4_ue].p_a_db = msg4_p_a;
    g->tti = (g->tti + 1) % 10240; g->count++;
    return npdu;
  }

  // ---- PDCCH ----
  {
    uint32_t nbits = 8 * g->regs.nregs[cfi - 1];
    std::vector<int> ctl(nbits, -1);
    for (auto& gr : grants) {
      bits_t b = dci_pack(g, gr);
      crc_attach(b, 0x11021, 16, gr.rnti);
      bits_t e = rm_conv_tx(conv_encode(b), 72 << gr.L);
      for (size_t i = 0; i < e.size(); i++) ctl[gr.ncce * 72 + i] = e[i];
    }
    bits_t sc = gold(sf * 512u + id, (int)nbits);
    for (uint32_t q = 0; q < ncce * 9; q++) {
      cf x[4]; bool any = false;
      for (int j = 0; j < 4; j++) {
        int b0 = ctl[8 * q + 2 * j], b1 = ctl[8 * q + 2 * j + 1];
        if (b0 < 0) { x[j] = cf(0, 0); continue; }
        any = true;
        b0 ^= sc[8 * q + 2 * j]; b1 ^= sc[8 * q + 2 * j + 1];
        x[j] = cf((1 - 2 * b0) * 0.70710678f, (1 - 2 * b1) * 0.70710678f);
      }
      if (any) map_quad(g->regs.k0[cfi - 1][q], g->regs.l[cfi - 1][q], x);
    }
  }

  // ---- PDSCH ----
  int pdu_i = 0;
  const float rho_b = P == 1 ? std::sqrt(0.8f) : 1.0f;
  for (auto& gr : grants) {
    if (gr.is_ul) continue;
    while (pdu_i < npdu && !(pdus[pdu_i].rnti == gr.rnti && !pdus[pdu_i].is_ul)) pdu_i++;
    std::vector<std::pair<int, int>> res;
    for (int l = l0; l < nsym; l++) for (int p : gr.prbs) for (int k = 12 * p; k < 12 * p + 12; k++) if (pdsch_re_ok(g, sf, l, k)) res.push_back({l, k});
    int nre_g = (int)res.size();
    float rho_a = 1.0f;
    for (auto& u : g->ues) if (u.rnti == gr.rnti) rho_a = std::pow(10.0f, u.p_a_db / 20.0f);
    std::vector<cf> sym[2];
    for (int i = 0; i < gr.ntb; i++) {
      if (pdu_i + i >= npdu) break;
      const txg_pdu_t& pd = pdus[pdu_i + i];
      int G = nre_g * gr.qm[i], NL = gr.scheme == 1 ? 2 : 1;
      bits_t e = dlsch_encode(pbuf + pd.offset, gr.tbs[i], G, gr.qm[i], NL, gr.rv[i]);
      bits_t sc = gold(((uint32_t)gr.rnti << 14) | ((uint32_t)i << 13) | (sf << 9) | (uint32_t)id, G);
      for (int b = 0; b < G; b++) e[b] ^= sc[b];
      modulate(e, gr.qm[i], sym[i]);
    }
    pdu_i += gr.ntb;
    for (int i = 0; i < nre_g; i++) {
      int l = res[i].first, k = res[i].second;
      float amp = rho_a * (is_crs01(c, l) ? rho_b : 1.0f);  // 36.213 5.2: rho_A from the UE's p-a, rho_B / rho_A from p-b = 1
      cf* g0 = &grid[0][l * nre + k];
      cf* g1 = P > 1 ? &grid[1][l * nre + k] : nullptr;
      switch (gr.scheme) {
        case 0: *g0 = sym[0][i] * amp; break;
        case 1:
          if ((i & 1) == 0 && i + 1 < nre_g) put_pair(i / 2, sym[0][i], sym[0][i + 1], l, k, res[i + 1].first, res[i + 1].second, amp);
          break;
        case 4:
          if ((i & 1) == 0 && i + 1 < nre_g) put_pair(i / 2, sym[0][i], sym[0][i + 1], l, k, res[i + 1].first, res[i + 1].second, amp);
          break;
        case 3: { cf x0 = sym[0][i], x1 = sym[1][i]; float s = (i & 1) ? -1.f : 1.f; *g0 = (x0 + x1) * 0.5f * amp; *g1 = (x0 - x1) * (0.5f * s) * amp; break; }
        case 2:
          if (gr.nlayers == 2) { cf x0 = sym[0][i], x1 = sym[1][i], d = (x0 - x1) * 0.5f; *g0 = (x0 + x1) * 0.5f * amp; *g1 = (gr.pmi == 0 ? d : cf(-d.imag(), d.real())) * amp; }
          break;
      }
    }
  }

  if (msg4_ue >= 0) g->ues[msg4_ue].p_a_db = msg4_p_a;

  // ---- OFDM + channel ----
  int sflen = 15 * N;
  std::vector<std::vector<cf>> tx(P, std::vector<cf>(sflen));
  std::vector<std::complex<double>> buf(N);
  for (int p = 0; p < P; p++) {
    int pos = 0;
    for (int l = 0; l < nsym; l++) {
      int cp = c.cp ? 512 * N / 2048 : ((l % 7) == 0 ? 160 : 144) * N / 2048;
      for (auto& v : buf) v = 0;
      for (int k = 0; k < nre; k++) { int bin = k < nre / 2 ? N - nre / 2 + k : k - nre / 2 + 1; buf[bin] = grid[p][l * nre + k]; }
      fft_d(buf, true);
      for (int n = 0; n < N; n++) tx[p][pos + cp + n] = cf((float)(buf[n].real() / N), (float)(buf[n].imag() / N));
      for (int n = 0; n < cp; n++) tx[p][pos + n] = tx[p][pos + cp + N - cp + n];
      pos += cp + N;
    }
  }
  double sigma = std::sqrt(std::pow(10.0, -c.snr_db / 10.0) / (2.0 * N));  // per real dimension, time domain
  double fs = 15000.0 * N;
  Rng noise((c.seed << 20) ^ ((g->count + 1) * 0xD6E8FEB86659FD93ull));  // per-subframe stream: captures can be rendered in parallel (txg_generate)
  // fading: per (rx, port) and per block of 64 samples ONE complex FIR = sum over the taps of (Rayleigh gain at the block's time) x (Hann-windowed sinc at the tap's
  // fractional delay); the samples in front of the subframe count as zero (the last taps of a subframe's very first samples fall into the cyclic prefix of symbol 0)
  std::vector<std::vector<std::complex<double>>> faded;
  if (c.chan_model) {
    const int nt = (int)g->tap_delay.size(), H = txg::FIR_HALF, BL = 64;
    const int Lh = (int)std::ceil(g->tap_delay[nt - 1]) + 2 * H + 2;
    faded.assign((size_t)c.nof_rx, std::vector<std::complex<double>>((size_t)sflen));
    std::vector<std::vector<double>> sinc((size_t)nt, std::vector<double>((size_t)(2 * H + 1)));
    std::vector<int> d0((size_t)nt);
    for (int i = 0; i < nt; i++) {
      d0[i] = (int)std::floor(g->tap_delay[i]);
      const double fr = g->tap_delay[i] - d0[i];
      for (int k = -H; k <= H; k++) { const double x = k - fr, w = 0.5 * (1 + std::cos(M_PI * x / (H + 1))); sinc[i][k + H] = (std::fabs(x) < 1e-12 ? 1.0 : std::sin(M_PI * x) / (M_PI * x)) * (std::fabs(x) <= H + 1 ? w : 0.0); }
    }
    std::vector<std::complex<double>> fir((size_t)Lh);
    for (uint32_t r = 0; r < c.nof_rx; r++) for (int p = 0; p < P; p++)
      for (int b0 = 0; b0 < sflen; b0 += BL) {
        const double t = ((double)g->count * sflen + b0 + BL / 2) / fs;
        for (auto& v : fir) v = 0;
        for (int i = 0; i < nt; i++) {
          const size_t base = (((size_t)r * 4 + (size_t)p) * (size_t)nt + (size_t)i) * txg::NSIN;
          std::complex<double> gn(0, 0);
          for (int m = 0; m < txg::NSIN; m++) { const double a = g->jk_w[base + m] * t + g->jk_ph[base + m]; gn += std::complex<double>(std::cos(a), std::sin(a)); }
          gn *= g->tap_amp[i] / std::sqrt((double)txg::NSIN);
          for (int k = -H; k <= H; k++) { const int idx = d0[i] + k + H; if (idx >= 0 && idx < Lh) fir[(size_t)idx] += gn * sinc[i][k + H]; }
        }
        for (int n = b0; n < b0 + BL && n < sflen; n++) {
          std::complex<double> acc(0, 0);
          for (int j = 0; j < Lh; j++) { const int m = n - j + H; if (m >= 0 && m < sflen) acc += fir[(size_t)j] * std::complex<double>(tx[p][m]); }
          faded[r][n] += acc;
        }
      }
  }
  for (uint32_t r = 0; r < c.nof_rx; r++) {
    float* out = iq + (size_t)r * sflen * 2;
    int dly = (r == 1) ? (int)c.delay_samples : 0;
    for (int n = 0; n < sflen; n++) {
      std::complex<double> y(0, 0);
      if (c.chan_model) y = faded[r][n];
      else for (int p = 0; p < P; p++) { int m = n - dly; cf v = m >= 0 ? tx[p][m] : cf(0, 0); y += std::complex<double>(g->h[r][p]) * std::complex<double>(v); }
      if (c.cfo_hz != 0 || c.cfo_drift_hz_per_s != 0) {
        const double t = ((double)n + (double)g->count * sflen) / fs;  // (count = subframes since the start, also across the 10240-subframe wrap of the TTI)
        const double ph = c.cfo_drift_hz_per_s != 0 ? 2 * M_PI * ((double)c.cfo_hz * t + 0.5 * (double)c.cfo_drift_hz_per_s * t * t)
                                                    : 2 * M_PI * c.cfo_hz * ((double)n + (double)(tti - c.start_tti) * sflen) / fs;
        y *= std::complex<double>(std::cos(ph), std::sin(ph));
      }
      double nr, ni; noise.gauss2(nr, ni);
      out[2 * n] = (float)(y.real() + sigma * nr);
      out[2 * n + 1] = (float)(y.imag() + sigma * ni);
    }
  }
  g->tti = (g->tti + 1) % 10240; g->count++;
  return npdu;
}

extern "C" void txg_set_plan_only(txg_t* g, int on) { g->plan_only = on != 0; }

// A whole capture [nsf][rx][15 N] cf32 rendered by `nthreads` workers: every worker walks the complete schedule (cheap) and synthesises the
// subframes of its blocks; the result is the one nsf consecutive txg_next calls produce.  Returns the TTI of the first subframe, or -1.
extern "C" int txg_generate(const txg_cfg_t* cfg, uint32_t nsf, float* iq, int nthreads) {
  if (fft_size(cfg->nof_prb) < 0 || !iq) return -1;
  const int T = std::max(1, nthreads);
  const uint32_t blk = 25;
  const size_t sf_floats = (size_t)cfg->nof_rx * 15u * (size_t)fft_size(cfg->nof_prb) * 2u;
  std::vector<std::thread> th;
  std::atomic<int> bad{0};
  for (int t = 0; t < T; t++) th.emplace_back([&, t] {
    txg_t* g = txg_new(cfg);
    if (!g) { bad++; return; }
    std::vector<txg_pdu_t> pdus(128);
    std::vector<uint8_t> pbuf(1u << 19);
    for (uint32_t i = 0; i < nsf; i++) {
      g->plan_only = ((i / blk) % (uint32_t)T) != (uint32_t)t;
      txg_next(g, iq + (size_t)i * sf_floats, pdus.data(), 128, pbuf.data(), (int)pbuf.size());
    }
    txg_free(g);
  });
  for (auto& x : th) x.join();
  return bad.load() ? -1 : (int)cfg->start_tti;
}

// =====================================================================================================
// Uplink: SC-FDMA transmitters of several UEs summed at the sniffer's uplink antenna (test tooling).
// TS 36.212 5.2.2 (UL-SCH: CRC, segmentation, turbo code, rate matching, control multiplexing with random control bits, channel interleaver),
// TS 36.211 5.3 (scrambling, modulation, transform precoding), 5.5 (DMRS), 5.6 (7.5 kHz shifted SC-FDMA).
typedef struct { uint32_t nof_prb, cell_id, cyclic_shift, delta_ss, group_hopping, sequence_hopping; /* SIB2 ul-ReferenceSignalsPUSCH */ uint32_t cp; /* 1: extended cyclic prefix (6 symbols per slot, reference signal on symbol 2) */ } txg_ul_cell_t;
typedef struct { uint16_t rnti; uint16_t n_dmrs; uint32_t n_prb, L_prb, mod, tbs, rv; float gain_db, phase_rad, ta_samples;
                 uint32_t nof_ack, cqi_bits, ri_bits; /* UCI multiplexed into the PUSCH (36.212 5.2.2.6-8): HARQ-ACK bits, CQI report size, RI bits */
                 uint32_t hop, n_prb2; /* hop = 1: slot 1 is sent on n_prb2 .. n_prb2 + L_prb - 1 (type-1 frequency hopping) */
                 uint32_t i_ack_p1, i_cqi_p1, i_ri_p1; /* 1 + betaOffset-ACK / -CQI / -RI-Index the UE was configured with, 0 = 10 / 8 / 11 */ } txg_ul_grant_t;

static int ul_largest_prime_below(int n) { for (int p = n - 1; p >= 2; p--) { bool ok = true; for (int d = 2; d * d <= p; d++) if (p % d == 0) { ok = false; break; } if (ok) return p; } return 2; }

// iq: 15*N cf32 (one antenna); payloads: concatenated tbs/8 bytes per grant at payload_off[i]; returns bytes used
extern "C" int txg_ul_make(const txg_ul_cell_t* c, uint32_t tti, const txg_ul_grant_t* gr, int ngr, float snr_db, uint64_t seed, float* iq,
                uint8_t* payloads, uint32_t* payload_off)
{
  Rng rng(seed * 7919ull + tti);
  const int nprb = (int)c->nof_prb, nre = 12 * nprb;
  int N = 0;
  switch (nprb) { case 6: N = 128; break; case 15: N = 256; break; case 25: N = 512; break; case 50: N = 1024; break; case 75: N = 1536; break; case 100: N = 2048; break; default: return -1; }
  const uint32_t sf = tti % 10;
  std::vector<std::complex<double>> grid((size_t)14 * nre, 0.0);
  const int nsl = c->cp ? 6 : 7, C = 2 * (nsl - 1), dm = nsl - 4;  // symbols per slot, PUSCH symbols per subframe (channel-interleaver columns), reference-signal symbol of a slot
  static const uint32_t d1[8] = {0, 2, 3, 4, 6, 8, 9, 10}, d2[8] = {0, 6, 3, 4, 2, 8, 10, 9};
  const uint32_t fss = ((c->cell_id % 30) + c->delta_ss) % 30;
  bits_t cpn = gold(((c->cell_id / 30) << 5) + fss, 8 * 7 * 20 + 8);
  uint32_t used = 0;
  for (int gi = 0; gi < ngr; gi++) {
    const txg_ul_grant_t& g = gr[gi];
    const int M = 12 * (int)g.L_prb, Qm = (int)g.mod, H = C * M * Qm;
    const int k0s[2] = {12 * (int)g.n_prb, 12 * (int)(g.hop == 1 ? g.n_prb2 : g.n_prb)};
    payload_off[gi] = used;
    uint8_t* pl = payloads + used;
    for (uint32_t i = 0; i < g.tbs / 8; i++) pl[i] = (uint8_t)rng.below(256);
    used += g.tbs / 8;
    // control resources, 36.212 5.2.2.6: Q' = min(ceil(O M_sc N_symb beta / sum K_r), cap), beta from the UE's betaOffset indices
    Segm sg; cbsegm((int)g.tbs, sg);
    const double sumK = (double)sg.Cp * sg.Kp + (double)sg.Cm * sg.Km;
    auto qprime = [&](int O, double beta, int cap) { if (O <= 0) return 0; int q = (int)std::ceil((double)O * M * (double)C * beta / sumK - 1e-9); return q < cap ? q : cap; };
    // 36.213 Tables 8.6.3-1/-2/-3
    static const double b_ack[16] = {2.0, 2.5, 3.125, 4.0, 5.0, 6.25, 8.0, 10.0, 12.625, 15.875, 20.0, 31.0, 50.0, 80.0, 126.0, 0.0};
    static const double b_ri[16] = {1.25, 1.625, 2.0, 2.5, 3.125, 4.0, 5.0, 6.25, 8.0, 10.0, 12.625, 15.875, 20.0, 0.0, 0.0, 0.0};
    static const double b_cqi[16] = {0.0, 0.0, 1.125, 1.25, 1.375, 1.625, 1.75, 2.0, 2.25, 2.5, 2.875, 3.125, 3.5, 4.0, 5.0, 6.25};
    const double beta_ack = b_ack[g.i_ack_p1 ? (g.i_ack_p1 - 1) & 15 : 10], beta_ri = b_ri[g.i_ri_p1 ? (g.i_ri_p1 - 1) & 15 : 11],
                 beta_cqi = b_cqi[g.i_cqi_p1 ? (g.i_cqi_p1 - 1) & 15 : 8];
    const int Qa = qprime((int)g.nof_ack, beta_ack, 4 * M), Qr = qprime((int)g.ri_bits, beta_ri, 4 * M);
    const int Oc = (int)g.cqi_bits, Qc = Oc ? qprime(Oc + (Oc > 11 ? 8 : 0), beta_cqi, C * M - Qr) : 0;
    const int G = (C * M - Qr - Qc) * Qm;
    bits_t f = dlsch_encode(pl, (int)g.tbs, G, Qm, 1, (int)g.rv);
    // channel interleaver 5.2.2.8: M rows x 12 columns of Qm-bit cells; RI first (bottom rows, columns 1,4,7,10), then CQI + data row by
    // row, then HARQ-ACK overwriting (bottom rows, columns 2,3,8,9); control bits are random here
    std::vector<int> owner((size_t)C * M, 0);  // 0 free, 2 RI
    bits_t mat((size_t)H);
    static const int ri_cols_n[4] = {1, 4, 7, 10}, ack_cols_n[4] = {2, 3, 8, 9}, ri_cols_e[4] = {0, 3, 5, 8}, ack_cols_e[4] = {1, 2, 6, 7};  // 36.212 Tables 5.2.2.8-1 / -2
    const int* ri_cols = c->cp ? ri_cols_e : ri_cols_n;
    const int* ack_cols = c->cp ? ack_cols_e : ack_cols_n;
    for (int i = 0, j = 0, r = M - 1; i < Qr; i++, r = M - 1 - i / 4, j = (j + 3) % 4) {
      owner[(size_t)r * C + ri_cols[j]] = 2;
      for (int b = 0; b < Qm; b++) mat[((size_t)r * C + ri_cols[j]) * Qm + b] = (uint8_t)rng.below(2);
    }
    {
      int k = 0;
      for (int r = 0; r < M; r++)
        for (int cc = 0; cc < C; cc++) {
          if (owner[(size_t)r * C + cc] == 2) continue;
          for (int b = 0; b < Qm; b++) mat[((size_t)r * C + cc) * Qm + b] = k < Qc ? (uint8_t)rng.below(2) : f[(size_t)(k - Qc) * Qm + b];
          k++;
        }
    }
    for (int i = 0, j = 0, r = M - 1; i < Qa; i++, r = M - 1 - i / 4, j = (j + 3) % 4)
      for (int b = 0; b < Qm; b++) mat[((size_t)r * C + ack_cols[j]) * Qm + b] = (uint8_t)rng.below(2);
    bits_t h((size_t)H);
    for (int col = 0; col < C; col++)
      for (int r = 0; r < M; r++)
        for (int b = 0; b < Qm; b++) h[((size_t)col * M + r) * Qm + b] = mat[((size_t)r * C + col) * Qm + b];
    bits_t scr = gold(((uint32_t)g.rnti << 14) | (sf << 9) | c->cell_id, H);
    for (int i = 0; i < H; i++) h[i] ^= scr[i];
    std::vector<cf> sym;
    modulate(h, Qm, sym);
    const std::complex<double> chan = std::polar(std::pow(10.0, g.gain_db / 20.0), (double)g.phase_rad);
    // base sequence: group u and number v of the slot (36.211 5.5.1.3 / 5.5.1.4), Zadoff-Chu root q from (u, v)
    const int Nzc = ul_largest_prime_below(M);
    auto slot_q = [&](uint32_t ns, uint32_t& u_out) -> long long {
      uint32_t fgh = 0, vv = 0;
      if (c->group_hopping) { bits_t cg = gold(c->cell_id / 30, 8 * 20); for (int i = 0; i < 8; i++) fgh += (uint32_t)cg[8 * ns + i] << i; fgh %= 30; }
      else if (c->sequence_hopping && M >= 72) { bits_t cs = gold(((c->cell_id / 30) << 5) + fss, 20); vv = cs[ns]; }
      u_out = (fgh + fss) % 30;
      const double qb = (double)Nzc * (double)(u_out + 1) / 31.0;
      long long q = (long long)std::floor(qb + 0.5);
      if (vv) q += ((long long)std::floor(2.0 * qb) & 1) ? -1 : 1;
      return q;
    };
    int col = 0;
    for (int l = 0; l < 2 * nsl; l++) {
      std::vector<std::complex<double>> v((size_t)M);
      if (l == dm || l == nsl + dm) {
        const uint32_t ns = 2 * sf + (l >= nsl ? 1 : 0);
        uint32_t npn = 0;
        for (int i = 0; i < 8; i++) npn += (uint32_t)cpn[8 * nsl * ns + i] << i;
        const uint32_t ncs = (d1[c->cyclic_shift & 7] + d2[g.n_dmrs & 7] + npn) % 12;
        uint32_t u = 0;
        const long long q = slot_q(ns, u);
        for (int n = 0; n < M; n++) {
          const long long m = n % Nzc;
          const double base = M == 12 ? M_PI * (double)lsn_dmrs_phi12[u][n] / 4.0  // one / two PRB: 36.211 Tables 5.5.1.2-1 / -2
                              : M == 24 ? M_PI * (double)lsn_dmrs_phi24[u][n] / 4.0
                                        : -M_PI * (double)((q * m * (m + 1)) % (2ll * Nzc)) / (double)Nzc;
          const double a = base + 2.0 * M_PI * (double)((ncs * (uint32_t)n) % 12) / 12.0;
          v[n] = std::complex<double>(std::cos(a), std::sin(a));
        }
      } else {
        for (int k = 0; k < M; k++) {  // transform precoding: (1/sqrt M) sum_r d[r] exp(-2 pi j r k / M)
          std::complex<double> acc(0, 0);
          for (int r = 0; r < M; r++) {
            const double a = -2.0 * M_PI * (double)(((long long)r * k) % M) / (double)M;
            acc += std::complex<double>(sym[(size_t)col * M + r]) * std::complex<double>(std::cos(a), std::sin(a));
          }
          v[k] = acc / std::sqrt((double)M);
        }
        col++;
      }
      const int k0 = k0s[l / nsl];
      for (int k = 0; k < M; k++) {
        // timing advance error = linear phase over the carriers (carrier k sits at (k - nre/2 + 1/2) * 15 kHz)
        const double fk = (double)(k0 + k) - nre / 2.0 + 0.5;
        const double a = -2.0 * M_PI * fk * (double)g.ta_samples / (double)N;
        grid[(size_t)l * nre + k0 + k] += chan * v[k] * std::complex<double>(std::cos(a), std::sin(a));
      }
    }
  }
  // SC-FDMA modulation with the half-subcarrier shift
  const int sflen = 15 * N;
  std::vector<std::complex<double>> buf(N);
  const double sigma = std::sqrt(std::pow(10.0, -snr_db / 10.0) / (2.0 * N));
  int pos = 0;
  for (int l = 0; l < 2 * nsl; l++) {
    const int cp = c->cp ? 512 * N / 2048 : ((l % 7) == 0 ? 160 : 144) * N / 2048;
    for (auto& v : buf) v = 0;
    for (int k = 0; k < nre; k++) buf[k < nre / 2 ? N - nre / 2 + k : k - nre / 2] = grid[(size_t)l * nre + k];
    fft_d(buf, true);
    for (int n = -cp; n < N; n++) {
      const double a = M_PI * (double)n / (double)N;
      const std::complex<double> y = buf[(size_t)((n + N) % N)] / (double)N * std::complex<double>(std::cos(a), std::sin(a));
      iq[2 * (pos + cp + n)] = (float)(y.real() + sigma * rng.gauss());
      iq[2 * (pos + cp + n) + 1] = (float)(y.imag() + sigma * rng.gauss());
    }
    pos += cp + N;
  }
  (void)sflen;
  return (int)used;
}
