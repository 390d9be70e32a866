// lsn_dev.h - device-side data layout shared by the HIP kernels and the host engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <stdexcept>
#include <string>
#include "../host/lsn_types.h"

// A launch the runtime rejects (LDS request above the function's limit on this device, an empty or oversized grid, no code object for the
// device) does not fail at the call site: the kernel never runs and the error sits in the thread until some later runtime call reports it -
// stale results at the next event wait.  Every launcher therefore asks right behind its launch and throws; the engine's stage wrappers turn
// the text into the chunk's error (round-4 review, weak 9).  The thread's error state is drained FIRST: the engine tolerates non-success codes elsewhere
// (hipEventQuery's NotReady while polling, a best-effort hipMalloc), and a code left behind by one of those must not be blamed on this kernel
// (round-5 advisor finding).
#define LSN_LAUNCH(kernel, grid, block, lds, stream, ...)                                                                      \
  do {                                                                                                                         \
    (void)hipGetLastError();                                                                                                   \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                                         \
    const hipError_t _le = hipGetLastError();                                                                                  \
    if (_le != hipSuccess) throw std::runtime_error(std::string("launch of " #kernel " failed: ") + hipGetErrorString(_le));   \
  } while (0)
// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the function ON ONE DEVICE: an engine on a second GPU of the process needs its own call
// (rounds 1-4 set it once per process).  `done` = one bit per device ordinal.
inline void lsn_func_max_lds(const void* fn, int bytes, std::atomic<uint64_t>& done, const char* name)
{
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) throw std::runtime_error("hipGetDevice failed");
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) throw std::runtime_error(std::string("hipFuncSetAttribute(") + name + ") failed: " + hipGetErrorString(e));
  done.fetch_or(bit, std::memory_order_release);
}

struct cf32 { float r, i; };

#define LSN_MAX_RX 2
#define LSN_MAX_PORTS 4
#define LSN_LLR_STRIDE 6400   // PDCCH LLR floats reserved per subframe (>= 8*787)
#define LSN_MAX_DCI_D 144     // payload + 16
#define LSN_NEG_METRIC (-12000)
#define LSN_LLR_CLIP 511
#define LSN_EXT_CLIP 2047

// per-subframe channel estimation scalars produced on the device (the host adds snr_db / cfo_hz)
struct LsnChest {
  float noise_avg, rsrp_avg, chan_ref, corr_r, corr_i;
  float noise[LSN_MAX_RX * LSN_MAX_PORTS], rsrp[LSN_MAX_RX * LSN_MAX_PORTS], cepow[LSN_MAX_RX * LSN_MAX_PORTS];  // [rx*nof_ports + port]
  float pad[3];
};

// cell-constant tables, passed to kernels by value
struct LsnCellDev {
  uint32_t nof_prb, nof_ports, id, nof_rx, N, lgN, nre, nref, sflen;
  uint32_t iq_nant;         // antennas interleaved in the IQ buffer ([sf][antenna][sflen]); nof_rx of them carry the downlink
  uint32_t nsub;            // power-of-two transform length: N, or 512 when N = 1536 = 3 x 512 (15 MHz); lgN = log2(nsub)
  uint32_t cp, nsym, nslot; // cyclic prefix (0 normal, 1 extended), symbols per subframe (14 / 12) and per slot (7 / 6).  Grids keep 14 rows per antenna;
                            // an extended-CP subframe fills rows 0 .. 11, rows 12, 13 stay zero (cleared once in setCell)
  uint32_t reg_w6;          // bit l set: the REGs of control symbol l span 6 REs (CRS in the symbol): symbol 0, symbol 1 of a four-port cell, symbol 3 with the extended CP
  const cf32* twiddle;      // [nsub/2] exp(-2 pi i k/nsub)
  const cf32* twiddle3;     // N = 1536 only: [1536] exp(-2 pi i k/1536) of the radix-3 combination, else null
  const cf32* nco_coarse;   // [4096]
  const cf32* nco_fine;     // [1024]
  const cf32* crs;          // [10][ports][4][nref]; ports 2, 3: rows 0, 1 = symbols 1, 8
  const uint16_t* reg_k0;   // [3][800] quadruplet -> first RE of its REG
  const uint8_t* reg_l;     // [3][800]
  const uint16_t* reg_q;    // [3][800] inverse map: REG in (symbol, frequency) order -> quadruplet of the PDCCH order, 0xFFFF = PCFICH / PHICH
  const uint8_t* pdcch_scr; // [10][LSN_LLR_STRIDE] scrambling bits
  const uint8_t* pcfich_scr;// [10][32]
  uint32_t nof_cce[3], nof_regs[3];
  uint32_t pcfich_k0[4];
  float taps[5];
  uint32_t nsizes;
  uint32_t sizes[LSN_MAX_SIZES];   // distinct DCI payload sizes, ascending
  const uint16_t* pbch_rank;       // [120] the same for the PBCH block (D = 40)
  const uint16_t* rankmap;         // [LSN_MAX_SIZES][3*LSN_MAX_DCI_D]: output position -> rank in the circular buffer
  const uint16_t* crc16_w;         // [LSN_MAX_SIZES + 1][64]: weight x^(n - 1 - i + 16) mod g_CRC16 of payload bit i for each DCI payload size n; last row: the MIB (24 bits)
  // PDSCH
  const uint16_t* validmask;       // [3][14][nof_prb]: 12-bit mask of PDSCH-capable REs (class 0: sf0, 1: sf5, 2: other)
  const uint8_t* gold_x1;          // [LSN_GOLD_LEN] x1(n+1600)
  const uint32_t* gold_x2mask;     // [LSN_GOLD_LEN] x2(n+1600) = parity(mask & cinit)
  const uint32_t* crc_tab_a;       // [6144] x^j mod gCRC24A
  const uint32_t* crc_tab_b;       // [6144] x^j mod gCRC24B
  const uint32_t* turbo_il;        // interleaver address tables of the turbo decoder (two steps per word), all 188 block sizes back to back (lsn_turbo_il_fill; LsnCbDev::il_off)
  // uplink (valid after lsn_phy_set_ul_config)
  const cf32* ul_shift;            // [N] exp(-j pi n / N): 7.5 kHz shift
  const cf32* ul_base;             // DMRS base sequences r_{u,0}(n) of every valid allocation size, concatenated
  const cf32* ul_idft;             // exp(+2 pi j k / M) of every valid allocation size, concatenated
  const cf32* ul_ph12;             // [12] exp(j 2 pi m / 12): cyclic-shift phasors
};
#define LSN_GOLD_LEN 115200

// one PDSCH decode job = one (accepted DCI, MCS table) pair
struct LsnGrantDev {
  uint32_t sf;          // subframe index inside the batch
  uint32_t sf_idx;      // 0..9
  uint32_t l0;          // first PDSCH symbol
  uint32_t prb_mask[2][4];
  uint32_t nof_re;
  uint32_t tx_scheme, pmi, nof_layers;
  uint32_t qm[2];       // per codeword, 0 = unused
  uint32_t cinit[2];
  uint32_t llr_off[2];  // int16 element offsets into the LLR arena
  uint32_t prefix_off;  // u16 element offset into the prefix arena: [14][nof_prb] then [16] symbol offsets
  float inv_amp_a, inv_amp_b;
};

// one turbo code block
struct LsnCbDev {
  uint32_t e_off;     // int16 element offset of this code block's rate-matched LLRs
  uint32_t E;
  uint32_t K, F, rv;
  uint32_t crc_b;     // 1: CRC24B (C>1), 0: CRC24A
  uint32_t out_off;   // byte offset in the payload arena
  uint32_t out_bytes; // (K - F - 24*crc_b)/8
  uint32_t il_off;    // word offset of this block size's table in LsnCellDev::turbo_il (turbo_il_offset(K))
  uint32_t reserved;
  uint32_t max_iter;
  uint32_t res_idx;   // slot of this block's LsnCbRes (launch order is sorted by size, results are not)
  uint32_t dep;       // res_idx of the FIRST code block of the same transport block when this one may be skipped once that one has failed
                      // (a transport block fails as soon as any of its code blocks fails); 0xFFFFFFFF: always decode
  uint32_t spp_off;   // u32 word offset (multiple of 4) of the block's de-rate-matched soft data: K packed words + 12 termination values (k_rm -> k_turbo)
  uint32_t nwin;      // lsn_turbo_nwin(K), from the host's table (0: the kernel works it out itself)
};
#define LSN_SPP_WORDS(K) (((K) + 12u + 3u) & ~3u)
// one PUSCH grant to decode
struct LsnUlGrantDev {
  uint32_t sf;          // subframe index inside the batch
  uint32_t n_prb, L_prb, qm;
  uint32_t n_prb2;      // first PRB in slot 1 (== n_prb unless the grant hops, 36.213 8.4.1)
  uint32_t ncs[2];      // DMRS cyclic shift n_cs of the two slots
  uint32_t cinit;       // scrambling: rnti << 14 | sf_idx << 9 | cell id
  uint32_t base_off, idft_off;  // offsets into ul_base (reference signal of slot 0: sequence group / number of that slot) / ul_idft
  uint32_t base_off1;   // ul_base offset of slot 1 (differs from base_off under group / sequence hopping)
  uint32_t hs_off;      // cf32 offset of the 2 M smoothed channel estimates
  uint32_t llr_off;     // int16 offset of the 12 M Qm LLRs (UL-SCH order)
  float scale;          // 1 / sqrt(M)
  uint32_t q_ack, q_ri, q_cqi;  // control symbols multiplexed into the allocation (36.212 5.2.2.6): HARQ-ACK punctures, RI / CQI are skipped
};

struct LsnCbRes { uint32_t ok, iters, rem_a, iters_run; uint32_t cyc_rm, cyc_map, cyc_out, cyc_all; };  // cyc_*: shader cycles per phase (s_memtime)

// Descriptor upload without the SDMA queue: a few workgroups read `bytes` (rounded up to words; both buffers are allocated with slack) from PINNED host
// memory and store them to device memory.  hipMemcpyAsync host -> device is served by one FIFO copy engine: a 3 KB descriptor upload queued behind the
// 393 MB IQ blocks of the ingest path waits for all of them (tools/copy_kernel_timeline.py) - this does not.
void lsn_launch_upload(void* dst_dev, const void* src_pinned, size_t bytes, hipStream_t s);
// The way back (stage-A mirrors, code-block verdicts, payloads): device -> PINNED host memory by posted PCIe writes of a copy kernel.  The copy
// engine serves both directions from one queue, so a hipMemcpyAsync device -> host would also wait behind the IQ blocks queued ahead.
void lsn_launch_download(void* dst_pinned, const void* src_dev, size_t bytes, hipStream_t s);
// up to six of either in one launch (word counts; buffers are allocated with slack to a whole word)
struct LsnCopySegs {
  const void* src[6]; void* dst[6]; uint32_t words[6]; uint32_t n = 0;
  void add(void* d, const void* s_, size_t bytes) { if (bytes && n < 6) { src[n] = s_; dst[n] = d; words[n] = (uint32_t)((bytes + 3) / 4); n++; } }
};
void lsn_launch_copy_multi(const LsnCopySegs& sg, bool to_host, hipStream_t s);
void lsn_launch_pdsch_prep_up(const LsnCellDev& c, const LsnGrantDev* jobs_host, LsnGrantDev* jobs_dev, uint32_t njobs, const LsnCopySegs& sg, uint16_t* prefix, hipStream_t s);
// launchers (stage_a.hip / stage_c.hip)
void lsn_launch_ofdm(const LsnCellDev& c, const cf32* iq, const uint32_t* dphi, cf32* grid, uint32_t nsf, hipStream_t s, float* rbp_part = nullptr /* [sf][14][128]: per-symbol PRB power terms of antenna 0 */);
void lsn_launch_chest(const LsnCellDev& c, const cf32* grid, const uint32_t* sf_idx, cf32* ce, float* raw, uint32_t nsf, hipStream_t s);
void lsn_launch_chest_fin(const LsnCellDev& c, const float* raw, LsnChest* out, uint32_t nsf, hipStream_t s);
void lsn_launch_pcfich(const LsnCellDev& c, const cf32* grid, const cf32* ce, const LsnChest* ch, const uint32_t* sf_idx, uint32_t* cfi, float* corr, uint32_t nsf, hipStream_t s);
void lsn_launch_pdcch_llr(const LsnCellDev& c, const cf32* grid, const cf32* ce, const LsnChest* ch, const uint32_t* sf_idx, const uint32_t* cfi, float* llr, uint32_t nsf, hipStream_t s);
void lsn_launch_cce_power(const LsnCellDev& c, const float* llr, const uint32_t* cfi, float* pw, uint32_t nsf, hipStream_t s);
void lsn_launch_viterbi(const LsnCellDev& c, const float* llr, const float* pw, const uint32_t* cfi, const uint32_t* sf_idx, LsnCand* cand, uint32_t* cand4, uint32_t nsf,
                        const LsnPruneCfg& pc, const uint32_t* snap, uint8_t* acc, hipStream_t s);
void lsn_launch_viterbi_block(const LsnCellDev& c, const float* llr, const float* pw, const uint32_t* cfi, const uint32_t* sf_idx, LsnCand* cand, uint32_t* cand4, uint32_t sf, uint32_t block,
                              const LsnPruneCfg& pc, const uint32_t* snap, uint8_t* acc, hipStream_t s);
void lsn_launch_rb_power(const LsnCellDev& c, const float* rbp_part, float* rbp, uint32_t nsf, hipStream_t s);
void lsn_launch_ul_fft(const LsnCellDev& c, const cf32* iq, uint32_t nant, uint32_t ant, cf32* grid, uint32_t nsf, hipStream_t s);
void lsn_launch_pbch(const LsnCellDev& c, const cf32* grid, const cf32* ce, const LsnChest* ch, float* llr5, LsnCand* out4, hipStream_t s);
void lsn_launch_file_unpack(const void* raw, uint32_t fmt /* LSN_FILE_* */, float scale, const cf32* rot, uint32_t sflen, uint32_t nant, cf32* out, uint32_t nsf, hipStream_t s);
void lsn_launch_prach(const cf32* iq, const uint64_t* occ_off, uint32_t nocc, const cf32* W, const cf32* D, const cf32* V, int N12, int Ncp, int b0,
                      int nroots, int ncs, int nwin, cf32* Y, float* corr, float* out, hipStream_t s);
void lsn_launch_pusch_chest(const LsnCellDev& c, const LsnUlGrantDev* g, const cf32* grid, cf32* hs, float* stat, uint32_t ngrants, hipStream_t s);
void lsn_launch_pusch_demod(const LsnCellDev& c, const LsnUlGrantDev* g, const cf32* grid, const cf32* hs, const float* stat, int16_t* llr,
                            uint32_t ngrants, hipStream_t s);
void lsn_launch_pdsch_prep(const LsnCellDev& c, const LsnGrantDev* g, uint16_t* prefix, uint32_t njobs, hipStream_t s);
void lsn_launch_pdsch_demod(const LsnCellDev& c, const LsnGrantDev* g, const uint32_t* items, uint32_t nitems, const uint16_t* prefix, const cf32* grid, const cf32* ce,
                            const LsnChest* ch, int16_t* llr, hipStream_t s);
void lsn_launch_rm(const LsnCbDev* cb, const int16_t* llr, uint32_t* spp, uint32_t ncb, uint32_t emax, hipStream_t s);
void lsn_launch_harq_combine(const LsnCbDev* cbs, uint32_t ncb, const uint32_t* keep, uint32_t* pool, uint32_t* scratch, bool copy, hipStream_t s);
#define LSN_CB_NODEP 0xFFFFFFFFu
void lsn_launch_turbo(const LsnCellDev& c, const LsnCbDev* cb, const uint32_t* spp, uint8_t* payload, LsnCbRes* res, uint32_t n128, uint32_t kmax128,
                      uint32_t n64, uint32_t kmax64, hipStream_t s, hipEvent_t between);
// largest code block two of which share one decoder workgroup (one wavefront and half of the LDS slot each): 2 x (6 K + 16 + 3584) <= 40 960 = a quarter of the CU's LDS
#define LSN_TURBO_PAIR_KMAX 2752u
void lsn_launch_turbo_packed(const LsnCellDev& c, const LsnCbDev* cb, const uint32_t* spp, uint8_t* payload, LsnCbRes* res, uint32_t nsolo, uint32_t kmax_solo,
                             uint32_t npair, uint32_t kmax_pair, hipStream_t s);
