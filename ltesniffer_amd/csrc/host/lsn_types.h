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
                            // bits 1-2: search-space verdict of (location, rnti): 0 invalid, 1 ambiguous with L-1, 2 valid;
                            // LSN_CAND_NOT_COMPUTED: the slot was left out (an ancestor location holds a candidate the search was predicted to accept) -
                            // a search that comes here all the same has it decoded on demand
};
#define LSN_CAND_NOT_COMPUTED 0x80u
// The sequential search decides on RNTI, search-space verdict and the first payload bit (format 0 or 1A); the payload itself is only wanted of the dozen DCIs
// it accepts.  Its view of a slot is therefore one word - 5 KB per subframe to pull into the search thread's cache instead of 20 (that thread bounds a cell, and a
// third of its time was waiting for the table): bits 0-15 the CRC remainder, 16 decoded, 17-18 the verdict, 19 payload bit 0, 23 LSN_CAND_NOT_COMPUTED
#define LSN_CAND_HOT(bits, rnti, flags) (((uint32_t)(rnti) & 0xFFFFu) | (((uint32_t)(flags) & 0x87u) << 16) | ((uint32_t)((unsigned long long)(bits) >> 63) << 19))
// Candidate pruning (k_viterbi, stage_a.hip): the stateless part of the prediction - the RNTI manager's format table as the kernel needs it.  Format f =
// index in falcon_ue_all_formats = the RNTI manager's format index = DciFormat.  Intervals: first | last << 16, at most four per format and kind (more: pruning off).
// The stateful part (active RNTIs as 2048 words of bits, then the primary-format mask) travels per chunk: LSN_PRUNE_SNAP_WORDS words.
struct LsnPruneCfg {
  uint32_t on;             // 0: exhaustive table (every slot decoded, as rounds 1-5)
  uint32_t fmt_size[9];    // size index of format f
  uint32_t n_ever[9], n_forb[9];
  uint32_t ever[9][4], forb[9][4];
};
#define LSN_PRUNE_SNAP_WORDS 2052
