// Compiles and links against the drop-in C++ mirror (include/ltesniffer_amd.hpp) the way a maintainer's LTESniffer_Core would.
// Without a HIP device the constructor must throw (the library has no CPU path); with one, a short worker-pool round trip runs.
#include "../../include/ltesniffer_amd.hpp"
#include <cstdio>
#include <cstring>

int main()
{
  lsn_cell_search_t cs;
  if (lsn_amd::cellSearch(nullptr, 0, 6, cs) != LSN_ERROR_INVALID_INPUTS) { printf("cellSearch accepted a null buffer\n"); return 4; }
  try {
    lsn_amd::Phy phy(/*nof_rx_antennas*/ 1, /*nof_workers*/ 4, /*skipSecondaryMetaFormats*/ false, /*metaFormatSplitRatio*/ 0.99, /*histogramThreshold*/ 5,
                     /*pcapwriter*/ nullptr);
    lsn_cell_t c{25, 1, 7, 0, 0, 0, 0};
    if (!phy.setCell(c)) { printf("setCell failed\n"); return 2; }
    auto w = phy.getAvail();
    if (!w) { printf("no worker\n"); return 2; }
    lsn_amd::cf_t** buf = w->getBuffers();
    std::memset(buf[0], 0, sizeof(lsn_amd::cf_t) * w->getBufferLen());
    lsn_dl_sf_cfg_t sf{0, 0, 0};
    w->prepare(0, 0, true, sf);
    phy.putPending(w);
    phy.joinPending();
    lsn_blind_stats_t st = phy.getStats();
    printf("device path ok: %u subframes\n", st.nof_subframes);
    return 0;
  } catch (const std::exception& ex) {
    printf("exception: %s\n", ex.what());
    return std::strstr(ex.what(), "no HIP device") ? 10 : 3;
  }
}
