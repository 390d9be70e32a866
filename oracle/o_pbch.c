/* o_pbch.c - ORACLE (test infrastructure only): PBCH / MIB decode on subframe 0.
 * Restates what the reference obtains from srsran_ue_mib_decode + srsran_pbch_mib_unpack in its DECODE_MIB state
 * (/root/reference/src/src/LTESniffer_Core.cc:382-395: the SFN of the stream = MIB SFN + the position of the radio frame
 * inside the 40 ms BCH period) [srsRAN ue_mib.c / pbch.c, not in tree], following TS 36.211 6.6 (PBCH: scrambling with
 * c_init = N_cell_ID re-started every 40 ms, QPSK, single port or SFBC, 4 x 240 symbols on the 72 centre carriers of
 * symbols 0-3 of slot 1, the CRS positions of four ports left out) and TS 36.212 5.3.1 (MIB 24 bits + CRC16 masked with
 * 0x0000 / 0xFFFF / 0x5555 for 1 / 2 / 4 ports, tail-biting convolutional code, rate matching to 1920 bits).
 * The worker's OFDM + CRS estimate run first with the configured number of ports (1, 2 or 4, like the rest of this restatement);
 * every quarter of the BCH period holds four complete copies of the 120 coded bits, so one subframe 0 is decoded under the
 * four hypotheses "this is radio frame q of the period" and the CRC (and its port mask) picks the right one.
 * Parity unpinned against srsRAN itself; arithmetic contract as in lsn_oracle.h. */
#include "lsn_oracle.h"
#include <stdlib.h>
#include <string.h>

#define SQRT2F 1.41421356237309504880f
static inline ocf_t cmulconj(ocf_t a, ocf_t b) { ocf_t c = {a.r * b.r + a.i * b.i, a.i * b.r - a.r * b.i}; return c; }

/* the PBCH resource elements of subframe 0 in mapping order (k fastest): l[i], k[i]; returns their number - 240 with the normal CP (symbols 0-3 of slot 1, the CRS
 * positions of four ports left out of symbols 0, 1), 216 with the extended CP (symbol 3 of the slot carries CRS as well; 36.211 6.6.4) */
int o_pbch_positions(const o_cell_t* cell, uint8_t* l, uint16_t* k)
{
  int nre = 12 * (int)cell->nof_prb, k0 = nre / 2 - 36, n = 0, nsl = o_nslot(cell);
  for (int s = nsl; s <= nsl + 3; s++)
    for (int c = k0; c < k0 + 72; c++) {
      if ((s <= nsl + 1 || (cell->cp && s == nsl + 3)) && (c % 3) == (int)(cell->id % 3)) continue; /* CRS of ports 0..3 */
      l[n] = (uint8_t)s; k[n] = (uint16_t)c; n++;
    }
  return n;
}

/* llr[480]: QPSK soft bits of the subframe's PBCH symbols, NOT descrambled (sign: positive = bit 1, like o_pdcch_llr) */
void o_pbch_llr(const o_cell_t* cell, uint32_t nof_rx, const ocf_t* grid, const ocf_t* ce, float noise, float* llr)
{
  int nre = 12 * (int)cell->nof_prb;
  uint8_t pl[240]; uint16_t pk[240];
  const int np = o_pbch_positions(cell, pl, pk);
  ocf_t x[240];
  if (cell->nof_ports == 1) {
    for (int i = 0; i < np; i++) {
      float nr = 0.0f, ni = 0.0f, den = 0.0f;
      for (uint32_t rx = 0; rx < nof_rx; rx++) {
        size_t b = ((size_t)rx * 14 + pl[i]) * (size_t)nre + pk[i];
        ocf_t t = cmulconj(grid[b], ce[b]);
        float hp = ce[b].r * ce[b].r + ce[b].i * ce[b].i;
        if (rx == 0) { nr = t.r; ni = t.i; den = hp; } else { nr = nr + t.r; ni = ni + t.i; den = den + hp; }
      }
      den = den + noise;
      x[i].r = nr / den; x[i].i = ni / den;
    }
  } else {
    for (int i = 0; i < np; i += 2) {
      float x0r = 0, x0i = 0, x1r = 0, x1i = 0, hh = 0;
      /* four ports (SFBC-FSTD): symbol pairs alternate between the port pairs (0, 2) and (1, 3) */
      const size_t pa = (cell->nof_ports == 4 && (i & 2)) ? 1 : 0, pb = cell->nof_ports == 4 ? pa + 2 : 1;
      for (uint32_t rx = 0; rx < nof_rx; rx++) {
        size_t bg = ((size_t)rx * 14 + pl[i]) * (size_t)nre;
        size_t b0 = ((pa * (size_t)nof_rx + rx) * 14 + pl[i]) * (size_t)nre, b1 = ((pb * (size_t)nof_rx + rx) * 14 + pl[i]) * (size_t)nre;
        ocf_t r0 = grid[bg + pk[i]], r1 = grid[bg + pk[i + 1]];
        ocf_t h00 = ce[b0 + pk[i]], h01 = ce[b0 + pk[i + 1]], h10 = ce[b1 + pk[i]], h11 = ce[b1 + pk[i + 1]];
        float hp = (h00.r * h00.r + h00.i * h00.i) + (h11.r * h11.r + h11.i * h11.i);
        ocf_t a = cmulconj(r0, h00), b = cmulconj(h11, r1), c = cmulconj(h10, r0), d = cmulconj(r1, h01);
        float t0r = a.r + b.r, t0i = a.i + b.i, t1r = d.r - c.r, t1i = d.i - c.i;
        if (rx == 0) { x0r = t0r; x0i = t0i; x1r = t1r; x1i = t1i; hh = hp; }
        else { x0r = x0r + t0r; x0i = x0i + t0i; x1r = x1r + t1r; x1i = x1i + t1i; hh = hh + hp; }
      }
      x[i].r = x0r / hh * SQRT2F; x[i].i = x0i / hh * SQRT2F;
      x[i + 1].r = x1r / hh * SQRT2F; x[i + 1].i = x1i / hh * SQRT2F;
    }
  }
  for (int i = 0; i < np; i++) { llr[2 * i] = -(x[i].r * SQRT2F); llr[2 * i + 1] = -(x[i].i * SQRT2F); }
  for (int i = 2 * np; i < 480; i++) llr[i] = 0.0f;
}

/* 36.331 MasterInformationBlock: dl-Bandwidth(3) phich-Duration(1) phich-Resource(2) systemFrameNumber(8) spare(10) */
static int mib_unpack(const uint8_t* b, o_mib_t* m)
{
  static const uint32_t bw[6] = {6, 15, 25, 50, 75, 100};
  static const uint32_t ng6[4] = {1, 3, 6, 12};
  uint32_t v = (uint32_t)(b[0] << 2 | b[1] << 1 | b[2]);
  if (v > 5) return 0;
  m->nof_prb = bw[v];
  m->phich_length = b[3];
  m->phich_ng_x6 = ng6[b[4] << 1 | b[5]];
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s = (s << 1) | b[6 + i];
  m->sfn = s << 2;
  return 1;
}

/* tries the four radio-frame positions; returns 1 and fills out when the CRC matches one of the port masks */
int o_pbch_decode(const o_cell_t* cell, uint32_t nof_rx, const ocf_t* grid, const ocf_t* ce, float noise, o_mib_t* out, float* llr_out)
{
  float llr[480], d[480];
  uint8_t c[1920], bits[24];
  const int E4 = cell->cp ? 432 : 480; /* coded bits per radio frame: 4 x E4 = 1920 / 1728 per 40 ms (36.212 5.3.1.3) */
  o_pbch_llr(cell, nof_rx, grid, ce, noise, llr);
  if (llr_out) memcpy(llr_out, llr, sizeof(llr));
  o_gold(cell->id, c, 4 * E4);
  memset(out, 0, sizeof(*out));
  for (uint32_t q = 0; q < 4; q++) {
    for (int i = 0; i < E4; i++) d[i] = c[(uint32_t)E4 * q + (uint32_t)i] ? -llr[i] : llr[i];
    uint16_t mask = o_dci_decode_off(d, E4, 24, bits, (int)(((uint32_t)E4 * q) % 120u)); /* frame q of the period starts E4 q bits into the rate-matched sequence */
    uint32_t ports = mask == 0x0000 ? 1u : (mask == 0xFFFF ? 2u : (mask == 0x5555 ? 4u : 0u));
    if (!ports) continue;
    int nz = 0;
    for (int i = 0; i < 24; i++) nz |= bits[i];
    if (!nz) continue; /* an all-zero block passes the CRC trivially */
    if (!mib_unpack(bits, out)) continue;
    out->found = 1; out->sfn_offset = q; out->nof_ports = ports; out->sfn = (out->sfn + q) % 1024;
    out->mib_bits = 0;
    for (int i = 0; i < 24; i++) out->mib_bits = (out->mib_bits << 1) | bits[i];
    return 1;
  }
  return 0;
}

/* srsran_ue_mib_decode on one subframe of samples iq[nof_rx][15 N] that is subframe 0 of some radio frame */
int o_mib_decode_subframe(const o_cell_t* cell, uint32_t nof_rx, const ocf_t* iq, o_mib_t* out, float* llr_out)
{
  const int N = o_fft_size(cell->nof_prb), sflen = 15 * N, nre = 12 * (int)cell->nof_prb;
  ocf_t* grid = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)nof_rx * 14 * (size_t)nre);
  ocf_t* ce = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)cell->nof_ports * nof_rx * 14 * (size_t)nre);
  o_chest_res_t ch;
  for (uint32_t rx = 0; rx < nof_rx; rx++) o_ofdm_rx(cell, iq + (size_t)rx * sflen, 0, grid + (size_t)rx * 14u * (size_t)nre);
  o_chest(cell, nof_rx, 0, grid, ce, &ch);
  int r = o_pbch_decode(cell, nof_rx, grid, ce, ch.noise_avg, out, llr_out);
  free(grid); free(ce);
  return r;
}
