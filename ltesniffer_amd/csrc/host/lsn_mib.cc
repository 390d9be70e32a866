// lsn_mib.cc - PBCH / MIB decode of one subframe 0 (srsran_ue_mib_decode + srsran_pbch_mib_unpack in the DECODE_MIB state of
// the reference, /root/reference/src/src/LTESniffer_Core.cc:382-395): the stream's SFN = MIB SFN + position of the radio frame
// in the 40 ms BCH period.  Runs the worker's own OFDM + CRS estimate (k_ofdm, k_chest, k_chest_fin) on the subframe and then
// k_pbch_llr + k_pbch_viterbi (stage_a.hip); the four candidates come back as four (bits, CRC mask) pairs.
// Product code: no CPU fallback, nothing from oracle/ is included or linked.
#include "lsn_engine.h"
#include <cstdio>
#include <cstring>
#include <stdexcept>

#define HIP_CHECK(x)                                                                                       \
  do {                                                                                                     \
    hipError_t _e = (x);                                                                                   \
    if (_e != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #x); \
  } while (0)

namespace lsn {

// iq: one subframe [iq_nant][15 N] cf32; returns 1 (found), 0 (no MIB in this subframe) or a negative error
int Engine::mibDecode(const void* iq, bool on_device, lsn_mib_t* out, float* llr_raw480)
{
  if (!cell_set) return LSN_ERROR;
  if (!iq || !out) return LSN_ERROR_INVALID_INPUTS;
  if (batch_open) return LSN_ERROR;  // borrows a chunk slot: not while submitted blocks are in flight (call lsn_phy_wait first)
  std::memset(out, 0, sizeof(*out));
  try {
    HIP_CHECK(hipSetDevice(cfg.device));
    Chunk& ch = chunks[0];  // idle between process calls
    hipStream_t st = stream_a[0];
    const size_t sf_bytes = (size_t)cd.iq_nant * cd.sflen * sizeof(cf32);
    if (!mib_d_iq) {
      HIP_CHECK(hipMalloc((void**)&mib_d_iq, sf_bytes));
      HIP_CHECK(hipMalloc((void**)&mib_d_llr, 5 * 480 * sizeof(float)));
      HIP_CHECK(hipMalloc((void**)&mib_d_cand, 4 * sizeof(LsnCand)));
    }
    const cf32* d_iq = (const cf32*)iq;
    if (!on_device) {
      HIP_CHECK(hipMemcpyAsync(mib_d_iq, iq, sf_bytes, hipMemcpyHostToDevice, st));
      d_iq = mib_d_iq;
    }
    ch.h_sfidx[0] = 0;
    HIP_CHECK(hipMemcpyAsync(ch.d_sfidx, ch.h_sfidx, sizeof(uint32_t), hipMemcpyHostToDevice, st));
    lsn_launch_ofdm(cd, d_iq, nullptr, ch.d_grid, 1, st);
    lsn_launch_chest(cd, ch.d_grid, ch.d_sfidx, ch.d_ce, ch.d_chest_raw, 1, st);
    lsn_launch_chest_fin(cd, ch.d_chest_raw, ch.d_chest, 1, st);
    lsn_launch_pbch(cd, ch.d_grid, ch.d_ce, ch.d_chest, mib_d_llr, mib_d_cand, st);
    LsnCand cand[4];
    HIP_CHECK(hipMemcpyAsync(cand, mib_d_cand, sizeof(cand), hipMemcpyDeviceToHost, st));
    if (llr_raw480) HIP_CHECK(hipMemcpyAsync(llr_raw480, mib_d_llr, 480 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    static const uint32_t bw[6] = {6, 15, 25, 50, 75, 100};
    static const uint32_t ng6[4] = {1, 3, 6, 12};
    for (uint32_t q = 0; q < 4; q++) {
      const uint32_t mask = cand[q].rnti & 0xFFFFu;
      const uint32_t ports = mask == 0x0000u ? 1u : (mask == 0xFFFFu ? 2u : (mask == 0x5555u ? 4u : 0u));
      const uint32_t mib = (uint32_t)(cand[q].bits >> 40);  // 24 bits, first bit = MSB
      if (!ports || !mib) continue;  // an all-zero block passes the CRC trivially
      const uint32_t bwi = mib >> 21;
      if (bwi > 5) continue;
      out->found = 1; out->sfn_offset = q; out->nof_ports = ports; out->mib_bits = mib;
      out->nof_prb = bw[bwi]; out->phich_length = (mib >> 20) & 1u; out->phich_resources_x6 = ng6[(mib >> 18) & 3u];
      out->sfn = ((((mib >> 10) & 0xFFu) << 2) + q) % 1024u;
      return 1;
    }
    return 0;
  } catch (const std::exception& ex) {
    fprintf(stderr, "ltesniffer_amd: %s\n", ex.what());
    return LSN_ERROR;
  }
}

}  // namespace lsn
