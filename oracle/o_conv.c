/* o_conv.c - ORACLE (test infrastructure only): PDCCH candidate decode = convolutional rate de-matching
 * (TS 36.212 5.1.4.2) + tail-biting Viterbi (K=7, r=1/3, g = 133,171,165 oct; 36.212 5.1.3.1) + CRC16
 * XOR -> RNTI.  Restates srsran_pdcch_dci_decode / srsran_rm_conv_rx / srsran_viterbi_decode_f [srsRAN, not
 * in tree] as called from /root/reference/lib/src/phy/falcon_phch/falcon_pdcch.c:142 (same bit handling as
 * the in-tree legacy variant falcon_pdcch.c:387-402). Quantisation: u8 = clamp(127.5 + 32*llr) truncated (scale and offset as the reference's own re-encoding check states them, falcon_pdcch.c:432),
 * 32-bit path metrics, three passes over the block (TB_ITER=3 of the SIMD srsRAN builds), middle pass output. */
#include "lsn_oracle.h"
#include "../spec/lte_tables.h"
#include <string.h>

void o_rm_conv_rx(const float* e, int E, float* out, int D3) { o_rm_conv_rx_off(e, E, out, D3, 0); }
/* ... for a piece of the rate-matched sequence that starts `skip` (non-<NULL>) positions into the circular buffer: the PBCH of one radio frame is bits
 * [E q, E (q + 1)) of the 4 E bits of a 40 ms period, and with the extended CP (E = 432) that is not a whole number of turns of the 120-bit buffer
 * (srsRAN de-rate-matches the whole period with the other three quarters marked absent - the same positions) */
void o_rm_conv_rx_off(const float* e, int E, float* out, int D3, int skip)
{
  int D = D3 / 3;
  int R = (D + 31) / 32, KP = 32 * R, ND = KP - D;
  /* map circular-buffer position j -> output index (or -1 for <NULL>) */
  static int map[3 * 32 * 8];
  for (int s = 0; s < 3; s++)
    for (int col = 0; col < 32; col++)
      for (int r = 0; r < R; r++) {
        int idx = r * 32 + lsn_perm_cc[col];
        map[s * KP + col * R + r] = (idx >= ND) ? 3 * (idx - ND) + s : -1;
      }
  static uint8_t seen[3 * O_DCI_MAX_BITS + 64];
  memset(seen, 0, (size_t)D3);
  for (int i = 0; i < D3; i++) out[i] = 0.0f;
  int j = 0, k = -skip;
  while (k < E) {
    int o = map[j];
    if (o >= 0) {
      if (k < 0) {
        /* in front of the piece */
      } else if (!seen[o]) {
        out[o] = e[k];
        seen[o] = 1;
      } else {
        out[o] = out[o] + e[k];
      }
      k++;
    }
    j++;
    if (j == 3 * KP) j = 0;
  }
}

static inline int parity6(unsigned x)
{
  x ^= x >> 4;
  x ^= x >> 2;
  x ^= x >> 1;
  return (int)(x & 1u);
}

void o_viterbi_tb(const uint8_t* sym, int D, uint8_t* bits)
{
  static uint64_t dec[3 * (O_DCI_MAX_BITS + 16)];
  int32_t m[64], mn[64];
  for (int s = 0; s < 64; s++) m[s] = 0;
  int T = 3 * D;
  for (int t = 0; t < T; t++) {
    const uint8_t* q = sym + 3 * (t % D);
    uint64_t dw = 0;
    for (int j = 0; j < 64; j++) {
      int b = j & 1, s0 = j >> 1, s1 = s0 | 32;
      int c0 = b ^ parity6((unsigned)s0 & 0x36u), c1 = b ^ parity6((unsigned)s0 & 0x27u), c2 = b ^ parity6((unsigned)s0 & 0x2Bu);
      int bm0 = (c0 ? 255 - q[0] : q[0]) + (c1 ? 255 - q[1] : q[1]) + (c2 ? 255 - q[2] : q[2]);
      int32_t a0 = m[s0] + bm0, a1 = m[s1] + (765 - bm0);
      if (a1 < a0) {
        mn[j] = a1;
        dw |= (uint64_t)1 << j;
      } else {
        mn[j] = a0;
      }
    }
    dec[t] = dw;
    memcpy(m, mn, sizeof(m));
  }
  int best = 0;
  for (int s = 1; s < 64; s++)
    if (m[s] < m[best]) best = s;
  int st = best;
  for (int t = T - 1; t >= 0; t--) {
    if (t >= D && t < 2 * D) bits[t - D] = (uint8_t)(st & 1);
    int d = (int)((dec[t] >> st) & 1u);
    st = (st >> 1) | (d << 5);
  }
}

uint16_t o_dci_decode(const float* llr, int E, int nof_bits, uint8_t* payload) { return o_dci_decode_off(llr, E, nof_bits, payload, 0); }
uint16_t o_dci_decode_off(const float* llr, int E, int nof_bits, uint8_t* payload, int skip)
{
  int D = nof_bits + 16;
  float rm[3 * (O_DCI_MAX_BITS + 16)];
  uint8_t q[3 * (O_DCI_MAX_BITS + 16)];
  uint8_t bits[O_DCI_MAX_BITS + 16];
  o_rm_conv_rx_off(llr, E, rm, 3 * D, skip);
  for (int i = 0; i < 3 * D; i++) {
    float v = 127.5f + 32.0f * rm[i];
    if (v < 0.0f) v = 0.0f;
    if (v > 255.0f) v = 255.0f;
    q[i] = (uint8_t)v;
  }
  o_viterbi_tb(q, D, bits);
  memcpy(payload, bits, (size_t)nof_bits);
  uint32_t p = 0;
  for (int i = 0; i < 16; i++) p = (p << 1) | bits[nof_bits + i];
  uint32_t crc = o_crc_bits(O_CRC16, 16, bits, nof_bits);
  return (uint16_t)((p ^ crc) & 0xFFFFu);
}
