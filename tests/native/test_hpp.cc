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
    // the reference's own constructor call (LTESniffer_Core.cc:63-86), argument for argument
    lsn_amd::MCSTracking mcs_tracking;
    lsn_amd::Phy phy(/*nof_rx_antennas*/ 1, /*nof_workers*/ 4, /*dciFileName*/ "", /*statsFileName*/ "", /*skipSecondaryMetaFormats*/ false,
                     /*metaFormatSplitRatio*/ 0.99, /*histogramThreshold*/ 5, /*pcapwriter*/ nullptr, &mcs_tracking, /*harq*/ nullptr, /*mcs_tracking_mode*/ 1,
                     /*harq_mode*/ 0, /*ulsche*/ nullptr);
    phy.getCommon().setShortcutDiscovery(true);                  // LTESniffer_Core.cc:87
    {  // LTESniffer_Core.cc:88-98: the DCI consumer list, in the reference's spelling (namespace aside)
      using namespace lsn_amd;
      using std::static_pointer_cast;
      Phy* phy_ = &phy;
      std::shared_ptr<DCIConsumerList> cons(new DCIConsumerList());
      cons->addConsumer(static_pointer_cast<SubframeInfoConsumer>(std::shared_ptr<DCIToFile>(new DCIToFile(phy_->getCommon().getDCIFile()))));
      phy_->getCommon().setDCIConsumer(cons);                    // :603-605
      phy_->getCommon().resetDCIConsumer();                      // :607-609
      if (phy_->getCommon().getDCIFile() != stdout) return 6;    // an empty dciFileName means stdout (PhyCommon.cc:19-24)
    }
    phy.getCommon().getRNTIManager().setHistogramThreshold(5);   // :620
    if (mcs_tracking.get_interval() != 5.0) return 5;
    lsn_cell_t c{25, 1, 7, 0, 0, 0, 0};
    if (!phy.setCell(c)) { printf("setCell failed\n"); return 2; }
    if (phy.getWorkers().size() != 4 || phy.getWorkers()[0]->getBufferLen() == 0) return 7;            // Phy.h:46 (the pool exists once the cell is set, Phy.cc:111-130)
    if (phy.getMetaFormats().getNofPrimaryMetaFormats() + phy.getMetaFormats().getNofSecondaryMetaFormats() != 9) return 8;   // Phy.h:45: nine formats
    auto w = phy.getAvail();
    if (!w) { printf("no worker\n"); return 2; }
    lsn_amd::cf_t** buf = w->getBuffers();
    std::memset(buf[0], 0, sizeof(lsn_amd::cf_t) * w->getBufferLen());
    lsn_dl_sf_cfg_t sf{0, 0, 0};
    w->prepare(0, 0, true, sf);
    phy.putPending(w);
    phy.joinPending();
    mcs_tracking.update_database_dl();                           // :473-499
    phy.getCommon().printStats();                                // :561
    lsn_blind_stats_t st = phy.getStats();
    printf("device path ok: %u subframes\n", st.nof_subframes);
    return 0;
  } catch (const std::exception& ex) {
    printf("exception: %s\n", ex.what());
    return std::strstr(ex.what(), "no HIP device") ? 10 : 3;
  }
}
