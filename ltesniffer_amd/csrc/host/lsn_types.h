// lsn_types.h - plain data shared by the HIP kernels, the host engine and the HIP-free host logic (no HIP includes).
#pragma once
#include <stdint.h>

#define LSN_MAX_LOC 160       // falcon_ue_dl.h:39 MAX_CANDIDATES_BLIND
#define LSN_MAX_SIZES 8       // distinct DCI payload sizes per cell
#define LSN_MAX_NUM_OF_CCE 84 // falcon_pdcch.h:36
#define LSN_CCE_STRIDE 96

// one blind-decode result: (location, DCI size) of one subframe
struct LsnCand {
  unsigned long long bits;  // payload bit i at position 63-i
  uint32_t rnti;            // CRC remainder = RNTI (falcon_pdcch.c:399-402)
  uint32_t flags;           // bit 0: decoded (0 = skipped: location out of range / insufficient power / all-zero LLRs);
                            // bits 1-2: search-space verdict of (location, rnti): 0 invalid, 1 ambiguous with L-1, 2 valid
};
