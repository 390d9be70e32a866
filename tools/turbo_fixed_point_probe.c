/* tools/turbo_fixed_point_probe.py builds this against the oracle: can a failing code block be stopped before its 12th iteration EXACTLY, i.e. does
 * the decoder state (extrinsics + window boundary metrics) repeat?  Test tooling over oracle/ (never part of the product). */
#include "../oracle/o_pdsch.c"
/* does the decoder state of a failing block repeat? state = ext[K] + the four boundary-metric arrays; returns the first iteration whose end state
 * equals an earlier one (1-based), 0 if none within max_iter; *period = distance */
int fp_probe(const int16_t* d3, int K, int max_iter, uint32_t crc_poly, int* period, int* ok_out, int* hard_same_from)
{
  if (!tr_init) trellis_init();
  int D = K + 4, f1, f2;
  if (o_qpp_find(K, &f1, &f2) < 0) return -1;
  int P = o_turbo_nwin(K);
  const int16_t *d0 = d3, *d1 = d3 + D, *d2 = d3 + 2 * D;
  int* pi = (int*)malloc(sizeof(int) * (size_t)K);
  int* id = (int*)malloc(sizeof(int) * (size_t)K);
  int16_t* ext = (int16_t*)calloc((size_t)K, sizeof(int16_t));
  int32_t* llr2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)K);
  uint8_t* bits = (uint8_t*)malloc((size_t)K);
  uint8_t* prevbits = (uint8_t*)calloc((size_t)K, 1);
  for (int i = 0; i < K; i++) { pi[i] = (int)(((long long)f1 * i + (long long)f2 * i * i) % K); id[i] = i; }
  int16_t ts1[3] = {d0[K], d2[K], d1[K + 1]}, tp1[3] = {d1[K], d0[K + 1], d2[K + 1]};
  int16_t ts2[3] = {d0[K + 2], d2[K + 2], d1[K + 3]}, tp2[3] = {d1[K + 2], d0[K + 3], d2[K + 3]};
  int32_t bt1[8], bt2[8];
  tail_beta(ts1, tp1, bt1); tail_beta(ts2, tp2, bt2);
  size_t nb = sizeof(int32_t[8]) * (size_t)P;
  int32_t(*a1)[8] = calloc(P, sizeof(int32_t[8])); int32_t(*b1)[8] = calloc(P, sizeof(int32_t[8]));
  int32_t(*a2)[8] = calloc(P, sizeof(int32_t[8])); int32_t(*b2)[8] = calloc(P, sizeof(int32_t[8]));
  size_t ssz = (size_t)K * 2 + 4 * nb;
  uint8_t* states = malloc(ssz * (size_t)max_iter);
  int it = 0, ok = 0, found = 0; *period = 0; *hard_same_from = 0;
  while (it < max_iter && !ok) {
    map_decode(K, P, d0, d1, id, ext, a1, b1, bt1, NULL);
    map_decode(K, P, d0, d2, pi, ext, a2, b2, bt2, llr2);
    for (int i = 0; i < K; i++) bits[pi[i]] = llr2[i] > 0 ? 1 : 0;
    uint8_t* s = states + ssz * (size_t)it;
    memcpy(s, ext, (size_t)K * 2); memcpy(s + K * 2, a1, nb); memcpy(s + K * 2 + nb, b1, nb); memcpy(s + K * 2 + 2 * nb, a2, nb); memcpy(s + K * 2 + 3 * nb, b2, nb);
    if (!found) for (int j = it - 1; j >= 0; j--) if (!memcmp(states + ssz * (size_t)j, s, ssz)) { found = it + 1; *period = it - j; break; }
    if (it > 0 && !memcmp(bits, prevbits, (size_t)K)) { if (!*hard_same_from) *hard_same_from = it + 1; } else *hard_same_from = 0;
    memcpy(prevbits, bits, (size_t)K);
    it++;
    ok = (o_crc_bits(crc_poly, 24, bits, K) == 0);
  }
  *ok_out = ok;
  free(pi); free(id); free(ext); free(llr2); free(bits); free(prevbits); free(a1); free(b1); free(a2); free(b2); free(states);
  return found;
}
