/* o_bits.c - ORACLE (test infrastructure only): CRC, Gold sequence, bit packing, fixed-order reduction.
 * Follows TS 36.212 5.1.1 (CRC) and TS 36.211 7.2 (pseudo-random sequence); the reference reaches these
 * through srsran_crc_checksum (/root/reference/lib/src/phy/falcon_phch/falcon_pdcch.c:401) and
 * srsran_sequence_* inside srsran_ue_dl_decode_fft_estimate (/root/reference/src/src/DCISearch.cc:562). */
#include "lsn_oracle.h"
#include <string.h>

/* Bitwise long division, zero-augmented message, initial register 0 (36.212 5.1.1). */
uint32_t o_crc_bits(uint32_t poly, int order, const uint8_t* bits, int n)
{
  uint32_t reg = 0, top = 1u << order;
  for (int i = 0; i < n + order; i++) {
    uint32_t b = (i < n) ? (bits[i] & 1u) : 0u;
    reg = (reg << 1) | b;
    if (reg & top) reg ^= poly;
  }
  return reg & (top - 1);
}

/* c(n) = x1(n+1600) ^ x2(n+1600); x1(0)=1, x2 = cinit (36.211 7.2). Registers hold x(n)..x(n+30), bit0 = x(n). */
void o_gold(uint32_t cinit, uint8_t* c, int len)
{
  uint32_t x1 = 1, x2 = cinit & 0x7FFFFFFFu;
  for (int n = 0; n < 1600 + len; n++) {
    if (n >= 1600) c[n - 1600] = (uint8_t)((x1 ^ x2) & 1u);
    uint32_t n1 = ((x1 >> 3) ^ x1) & 1u;
    uint32_t n2 = ((x2 >> 3) ^ (x2 >> 2) ^ (x2 >> 1) ^ x2) & 1u;
    x1 = (x1 >> 1) | (n1 << 30);
    x2 = (x2 >> 1) | (n2 << 30);
  }
}

void o_unpack_bytes(const uint8_t* bytes, uint8_t* bits, int nbits)
{
  for (int i = 0; i < nbits; i++) bits[i] = (bytes[i >> 3] >> (7 - (i & 7))) & 1u;
}

void o_pack_bits(const uint8_t* bits, uint8_t* bytes, int nbits)
{
  memset(bytes, 0, (size_t)(nbits + 7) / 8);
  for (int i = 0; i < nbits; i++)
    if (bits[i]) bytes[i >> 3] |= (uint8_t)(0x80u >> (i & 7));
}

/* The ONE reduction order used by every float sum that feeds a decision: 256 strided partial sums
 * (thread t adds v[t], v[t+256], ... in increasing index order starting from +0.0f) followed by a
 * binary tree partial[t] += partial[t+s], s = 128..1.  A 256-thread workgroup reproduces it exactly. */
float o_reduce256(const float* v, int n)
{
  float partial[256];
  for (int t = 0; t < 256; t++) {
    float p = 0.0f;
    for (int i = t; i < n; i += 256) p = p + v[i];
    partial[t] = p;
  }
  for (int s = 128; s > 0; s >>= 1)
    for (int t = 0; t < s; t++) partial[t] = partial[t] + partial[t + s];
  return partial[0];
}
