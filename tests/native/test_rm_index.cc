// Host unit test for the closed-form rate-matcher index arithmetic (ltesniffer_amd/csrc/kernels/lsn_rm.h):
// compares lsn_rm_rank() with a brute-force walk of the circular buffer (TS 36.212 5.1.4.1.2) for every
// turbo block size, several filler counts and all redundancy versions.
#include "../../ltesniffer_amd/csrc/kernels/lsn_rm.h"
#include "../../spec/lte_tables.h"
#include <cstdio>
#include <vector>
int main()
{
  long checked = 0;
  for (int qi = 0; qi < LSN_QPP_NSIZES; qi++) {
    int K = lsn_qpp_table[qi][0];
    int Fs[4] = {0, 8, 24, 56};
    for (int fi = 0; fi < 4; fi++) {
      int F = Fs[fi];
      if (F >= K - 24) continue;
      for (int rv = 0; rv < 4; rv++) {
        LsnRmGeom g;
        lsn_rm_geom(g, K, F, rv);
        int D = K + 4, R = (D + 31) / 32, KP = 32 * R, ND = KP - D, Ncb = 3 * KP;
        std::vector<int> map(Ncb);
        for (int k = 0; k < KP; k++) {
          int col = k / R, row = k % R, y = row * 32 + lsn_perm_tc[col], i01 = y - ND;
          map[k] = (i01 >= 0 && i01 >= F) ? i01 : -1;
          map[KP + 2 * k] = (i01 >= 0 && i01 >= F) ? D + i01 : -1;
          int pi = (lsn_perm_tc[col] + 32 * row + 1) % KP;
          map[KP + 2 * k + 1] = (pi - ND >= 0) ? 2 * D + pi - ND : -1;
        }
        int k0 = R * (2 * ((Ncb + 8 * R - 1) / (8 * R)) * rv + 2);
        std::vector<int> rank(3 * D, -1);
        int k = 0;
        for (int j = 0; j < Ncb; j++) {
          int o = map[(k0 + j) % Ncb];
          if (o >= 0) rank[o] = k++;
        }
        LsnRmCol tab[32];
        for (int c = 0; c < 32; c++) tab[c] = lsn_rm_fast_col(g, c);
        if (ND != 4 && ND != 12 && ND != 20 && ND != 28) { printf("ND = %d\n", ND); return 1; }
        if (k != g.nn) { printf("nn mismatch K=%d F=%d rv=%d: %d vs %d\n", K, F, rv, k, g.nn); return 1; }
        for (int s = 0; s < 3; s++)
          for (int i = 0; i < D; i++) {
            int r = lsn_rm_rank(g, s, i);
            if (r != rank[s * D + i]) { printf("rank mismatch K=%d F=%d rv=%d s=%d i=%d: %d vs %d\n", K, F, rv, s, i, r, rank[s * D + i]); return 1; }
            // the one-entry-per-column form the de-rate-matching kernel uses (k_rm)
            if (!(s < 2 && i < F)) {
              int f;
              if (s < 2) {
                int r0, r1;
                lsn_rm_rank01_fast(tab[lsn_rm_col_of(i + ND)], g.nn, i + ND, &r0, &r1);
                f = s == 0 ? r0 : r1;
              } else {
                f = lsn_rm_rank2_fast(tab[lsn_rm_col_of(i + ND - 1)], g.nn, i + ND - 1);
              }
              if (r >= 0 && f != r) { printf("fast rank mismatch K=%d F=%d rv=%d s=%d i=%d: %d vs %d\n", K, F, rv, s, i, f, r); return 1; }
            }
            checked++;
          }
      }
    }
  }
  printf("OK %ld\n", checked);
  return 0;
}
