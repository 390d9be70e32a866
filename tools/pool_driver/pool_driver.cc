// tools/pool_driver/pool_driver.cc - drives the Phy worker pool of the C ABI the way LTESniffer_Core::run does in file mode
// (/root/reference/src/src/LTESniffer_Core.cc:434-451: getAvail() -> receive one subframe into getBuffers() -> prepare(sf_idx, sfn,
// updateMetaFormats every 500 subframes, sf_cfg) -> putPending(); joinPending() at the end).  Bench / test tooling, not product: it only
// calls functions of include/ltesniffer_amd.h.  The "receive" is a memcpy from a capture in host memory (what srsran_ue_sync_zerocopy
// amounts to in file mode); fill_threads > 1 lets several threads do those copies for a group of workers that is then queued in TTI order
// (a caller that can produce samples faster than one core copies them).
#include "../../include/ltesniffer_amd.h"
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

extern "C" int pool_drive(lsn_phy_t* phy, const float* iq, uint32_t nsf, uint32_t nof_rx, uint32_t sflen, uint32_t start_tti, uint32_t meta_period,
                          int fill_threads, double* seconds)
{
  if (!phy || !iq || !nsf) return LSN_ERROR_INVALID_INPUTS;
  const size_t row = (size_t)sflen * 2;  // floats per antenna and subframe
  const auto t0 = std::chrono::steady_clock::now();
  auto fill = [&](lsn_worker_t* w, uint32_t i) {
    float** b = lsn_worker_buffers(w);
    for (uint32_t rx = 0; rx < nof_rx; rx++) std::memcpy(b[rx], iq + ((size_t)i * nof_rx + rx) * row, row * sizeof(float));
    const uint32_t tti = (start_tti + i) % 10240;
    lsn_dl_sf_cfg_t sfc{};
    sfc.tti = tti;
    lsn_worker_prepare(w, tti % 10, tti / 10, meta_period && (i % meta_period) == 0, &sfc);
  };
  int rc = LSN_SUCCESS;
  if (fill_threads <= 1) {
    for (uint32_t i = 0; i < nsf && rc == LSN_SUCCESS; i++) {
      lsn_worker_t* w = lsn_phy_get_avail(phy, 1);
      if (!w) { rc = LSN_ERROR; break; }
      fill(w, i);
      rc = lsn_phy_put_pending(phy, w);
    }
  } else {
    const uint32_t group = (uint32_t)fill_threads * 4;
    std::vector<lsn_worker_t*> ws(group);
    for (uint32_t base = 0; base < nsf && rc == LSN_SUCCESS; base += group) {
      const uint32_t n = std::min(group, nsf - base);
      for (uint32_t k = 0; k < n; k++) { ws[k] = lsn_phy_get_avail(phy, 1); if (!ws[k]) rc = LSN_ERROR; }
      if (rc != LSN_SUCCESS) break;
      std::atomic<uint32_t> next{0};
      std::vector<std::thread> th;
      for (int t = 0; t < fill_threads; t++) th.emplace_back([&] { for (uint32_t k; (k = next.fetch_add(1)) < n;) fill(ws[k], base + k); });
      for (auto& t : th) t.join();
      for (uint32_t k = 0; k < n && rc == LSN_SUCCESS; k++) rc = lsn_phy_put_pending(phy, ws[k]);
    }
  }
  const int j = lsn_phy_join_pending(phy);
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rc != LSN_SUCCESS ? rc : j;
}
