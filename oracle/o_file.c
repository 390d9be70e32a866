/* o_file.c - ORACLE (test infrastructure only): the IQ capture file source in front of the worker.
 * Restates srsran_ue_sync_init_file_multi + srsran_ue_sync_zerocopy in file mode
 * (/root/reference/src/src/LTESniffer_Core.cc:252-258,365; -O / -o options ArgManager.cc:144-149) [srsRAN ue_sync.c /
 * filesource.c, not in tree]: the file holds complex float32 samples, the antennas interleaved sample by sample; the
 * first `offset_time` samples (per antenna) are skipped once; every call delivers one subframe (15 N samples per antenna),
 * no PSS tracking in file mode - the file is taken as subframe aligned, the subframe counter just increments; with a
 * non-zero frequency offset every subframe is multiplied by exp(-j 2 pi offset_freq n / fs), n restarting at 0 in each
 * subframe (srsran_cfo_correct is called per subframe with freq = -offset_freq / 15000 / N).
 * Parity unpinned against srsRAN itself; arithmetic contract as in lsn_oracle.h (table in double on the "host" side, one
 * float rounding per operation). */
#include "lsn_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

/* fmt 0: complex float32 (the reference's file type).  fmt 1 / 2: int16 / int8 I/Q pairs (srsRAN's SRSRAN_COMPLEX_SHORT_BIN file type, the
 * radio's link format; not a file type the reference's ue_sync opens - an extension of the product, restated here so that it has a checker):
 * sample = (float)integer * scale, scale 0 = full scale +-1.
 * out: [nsf][nant][15 N]; returns the number of complete subframes delivered (<= nsf), or -1 */
long o_file_read_fmt(const char* path, uint32_t nof_prb, uint32_t nant, long offset_time, float offset_freq, uint32_t first_sf, uint32_t nsf,
                     uint32_t fmt, float scale, ocf_t* out)
{
  FILE* f = fopen(path, "rb");
  if (!f || nant == 0 || fmt > 2) { if (f) fclose(f); return -1; }
  const int N = o_fft_size(nof_prb), sflen = 15 * N;
  const double fs = 15000.0 * (double)N;
  const long smp = fmt == 1 ? 4 : fmt == 2 ? 2 : (long)sizeof(ocf_t);
  if (scale == 0.0f) scale = fmt == 1 ? 1.0f / 32768.0f : 1.0f / 128.0f;
  if (fseek(f, ((long)offset_time + (long)first_sf * sflen) * (long)nant * smp, SEEK_SET)) { fclose(f); return -1; }
  ocf_t* raw = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)sflen * nant);
  void* rawi = malloc((size_t)smp * (size_t)sflen * nant);
  ocf_t* rot = NULL;
  if (offset_freq != 0.0f) {
    rot = (ocf_t*)malloc(sizeof(ocf_t) * (size_t)sflen);
    for (int n = 0; n < sflen; n++) {
      double a = -2.0 * M_PI * (double)offset_freq * (double)n / fs;
      rot[n].r = (float)cos(a); rot[n].i = (float)sin(a);
    }
  }
  long done = 0;
  for (uint32_t s = 0; s < nsf; s++) {
    if (fread(rawi, (size_t)smp, (size_t)sflen * nant, f) != (size_t)sflen * nant) break;
    for (size_t k = 0; k < (size_t)sflen * nant; k++) {
      if (fmt == 1) { raw[k].r = (float)((const int16_t*)rawi)[2 * k] * scale; raw[k].i = (float)((const int16_t*)rawi)[2 * k + 1] * scale; }
      else if (fmt == 2) { raw[k].r = (float)((const int8_t*)rawi)[2 * k] * scale; raw[k].i = (float)((const int8_t*)rawi)[2 * k + 1] * scale; }
      else raw[k] = ((const ocf_t*)rawi)[k];
    }
    for (uint32_t a = 0; a < nant; a++)
      for (int n = 0; n < sflen; n++) {
        ocf_t x = raw[(size_t)n * nant + a];
        if (rot) { ocf_t y = {x.r * rot[n].r - x.i * rot[n].i, x.r * rot[n].i + x.i * rot[n].r}; x = y; }
        out[((size_t)s * nant + a) * sflen + n] = x;
      }
    done++;
  }
  free(raw); free(rawi); free(rot); fclose(f);
  return done;
}

long o_file_read(const char* path, uint32_t nof_prb, uint32_t nant, long offset_time, float offset_freq, uint32_t first_sf, uint32_t nsf, ocf_t* out)
{
  return o_file_read_fmt(path, nof_prb, nant, offset_time, offset_freq, first_sf, nsf, 0, 0.0f, out);
}
