// ltesniffer_amd.hpp - header-only C++ mirror of the reference's Phy / SubframeWorker classes on top of the C ABI
// (ltesniffer_amd.h).  It keeps the names, argument order and meaning of
//   class Phy            /root/reference/src/include/Phy.h:22-66
//   class SubframeWorker /root/reference/src/include/SubframeWorker.h:16-86
// and every member LTESniffer_Core.cc calls on them (constructor :63-90, loop :292-299,361-451, shutdown :547-562, :603-621): tests/native/test_hpp.cc
// repeats those call sites, in the reference's spelling, against this header.  What does NOT carry over unchanged: the srsRAN PODs (re-declared in the C
// header), the singletons this library owns itself (RNTIManager, MCSTracking, DCIMetaFormats live behind lsn_phy_t and are reached through the facades
// below - same member names, other class names).  Since round 6 the reference's own LTESniffer_Core.cc IS compiled against this header in the CPU suite
// (tests/test_reference_caller.py, through ltesniffer_amd_compat.hpp: its class header with the worker-side includes swapped, nothing else touched;
// boost::program_options, srsue and the srsRAN radio / synchronisation API - absent from this image and outside the path - are DECLARED by
// tests/native/core_shim, so the check is of syntax and types, not a link).
#pragma once
#include "ltesniffer_amd.h"
#include <complex>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace lsn_amd {

#ifdef LSN_AMD_SRSRAN_CF_T
using ::cf_t;                      // the caller's own srsRAN headers are in scope: their cf_t (float _Complex), so that getBuffers() goes into srsran_ue_sync_zerocopy unchanged
#else
typedef std::complex<float> cf_t;  // srsRAN cf_t = float _Complex, same layout
#endif

class Phy;

class SubframeWorker {
public:
  cf_t** getBuffers() { return reinterpret_cast<cf_t**>(lsn_worker_buffers(h)); }   // SubframeWorker.h:36
  cf_t** getBuffers_offset() { return reinterpret_cast<cf_t**>(lsn_worker_buffers_offset(h)); }  // SubframeWorker.h:37
  uint32_t getBufferLen() const { return lsn_worker_buffer_len(h); }                // 3 * SF_LEN samples per antenna
  void prepare(uint32_t sf_idx, uint32_t sfn, bool updateMetaFormats, const lsn_dl_sf_cfg_t& dl_sf)  // SubframeWorker.h:32
  {
    if (lsn_worker_prepare(h, sf_idx, sfn, updateMetaFormats ? 1 : 0, &dl_sf) != LSN_SUCCESS) throw std::invalid_argument("prepare");
  }
  // the caller's srsran_dl_sf_cfg_t (SubframeWorker.h:32; LTESniffer_Core.cc:429-434): tti and sf_type cross the boundary, the CFI is decoded per subframe
  template <class DlSfCfg> void prepare(uint32_t sf_idx, uint32_t sfn, bool updateMetaFormats, const DlSfCfg& dl_sf)
  {
    prepare(sf_idx, sfn, updateMetaFormats, lsn_dl_sf_cfg_t{(uint32_t)dl_sf.tti, 0u, (uint32_t)dl_sf.sf_type});
  }
  uint32_t getSfidx() const { return lsn_worker_sf_idx(h); }
  uint32_t getSfn() const { return lsn_worker_sfn(h); }
  // work() is not exposed: the engine runs it for whole batches (SnifferThread::execute_worker, WorkerThread.cc:78-91)
private:
  friend class Phy;
  explicit SubframeWorker(lsn_worker_t* w) : h(w) {}
  lsn_worker_t* h;
};

// Facades for the objects LTESniffer_Core still talks to after the swap.  Their state lives behind lsn_phy_t now; the facades forward the
// calls the unchanged caller makes (LTESniffer_Core.cc:87,422-426,473-499,535-540,561,616,620).
class RNTIManagerFacade {   // phy->getCommon().getRNTIManager()
public:
  void setHistogramThreshold(uint32_t threshold) { lsn_phy_set_histogram_threshold(h, threshold); }        // RNTIManager.cc:442-444
  void addEvergreen(uint16_t a, uint16_t b, uint32_t formatIdx) { lsn_phy_add_evergreen(h, a, b, formatIdx); }
  void addForbidden(uint16_t a, uint16_t b, uint32_t formatIdx) { lsn_phy_add_forbidden(h, a, b, formatIdx); }
  uint32_t nofActive() { return lsn_phy_nof_active_rnti(h); }
private:
  friend class PhyCommonFacade;
  friend class Phy;
  lsn_phy_t* h = nullptr;
};
// PhyCommon's DCI consumer hook (PhyCommon.h:57-62, SubframeInfoConsumer.h).  LTESniffer_Core builds a DCIConsumerList and hands it over
// (LTESniffer_Core.cc:88-98, 603-609); the one call that would feed it - common.consumeDCICollection(subframeInfo) - is a comment in the reference's
// worker (SubframeWorker.cc:205), so the consumer is stored and never invoked there.  Same here.
class SubframeInfo;
class SubframeInfoConsumer {
public:
  virtual ~SubframeInfoConsumer() {}
  virtual void consumeDCICollection(const SubframeInfo& subframeInfo) = 0;
};
class DCIToFile : public SubframeInfoConsumer {   // SubframeInfoConsumer.h: the default consumer, a FILE* holder
public:
  DCIToFile() : dci_file(stdout) {}
  explicit DCIToFile(FILE* f) : dci_file(f) {}
  void setFile(FILE* f) { dci_file = f; }
  FILE* getFile() { return dci_file; }
  void consumeDCICollection(const SubframeInfo&) override {}
private:
  FILE* dci_file;
};
class DCIConsumerList : public SubframeInfoConsumer {   // SubframeInfoConsumer.cc:10-19
public:
  void consumeDCICollection(const SubframeInfo& s) override { for (auto& c : consumers) c->consumeDCICollection(s); }
  void addConsumer(std::shared_ptr<SubframeInfoConsumer> consumer) { consumers.push_back(consumer); }
private:
  std::vector<std::shared_ptr<SubframeInfoConsumer>> consumers;
};
// phy->getMetaFormats() (Phy.h:45): the primary / secondary split of the nine DCI formats, read from the library
class DCIMetaFormatsFacade {
public:
  uint32_t getNofPrimaryMetaFormats() { refresh(); return np; }
  uint32_t getNofSecondaryMetaFormats() { refresh(); return ns; }
  const uint32_t* getPrimaryMetaFormats() { refresh(); return prim; }       // indices into falcon_ue_all_formats (= srsran_dci_format_t 0..8)
  const uint32_t* getSecondaryMetaFormats() { refresh(); return sec; }
private:
  friend class Phy;
  void refresh() { if (h) lsn_phy_get_meta_formats(h, prim, &np, sec, &ns); }
  lsn_phy_t* h = nullptr;
  uint32_t prim[9] = {0}, sec[9] = {0}, np = 0, ns = 0;
};
class PhyCommonFacade {     // phy->getCommon()
public:
  FILE* getDCIFile() { return dci_file; }                                                                  // PhyCommon.cc:48-50 (opened like PhyCommon.cc:19-24)
  void setDCIConsumer(std::shared_ptr<SubframeInfoConsumer> consumer) { dciConsumer = consumer; }          // PhyCommon.cc:77-79
  void resetDCIConsumer() { dciConsumer = defaultDCIConsumer; }                                            // PhyCommon.cc:81-83
  void setShortcutDiscovery(bool enable) { lsn_phy_set_shortcut_discovery(h, enable ? 1 : 0); }           // PhyCommon.cc:69-71
  bool getShortcutDiscovery() const { return lsn_phy_get_shortcut_discovery(h) != 0; }
  void printStats() { lsn_phy_print_stats(h, stats_file); }                                                // PhyCommon.cc:65-67
  lsn_blind_stats_t getStats() { lsn_blind_stats_t s{}; lsn_phy_get_stats(h, &s); return s; }
  RNTIManagerFacade& getRNTIManager() { return rm; }
  FILE* getStatsFile() { return stats_file; }
private:
  friend class Phy;
  lsn_phy_t* h = nullptr;
  FILE* stats_file = nullptr;
  FILE* dci_file = stdout;
  std::shared_ptr<DCIToFile> defaultDCIConsumer{new DCIToFile()};
  std::shared_ptr<SubframeInfoConsumer> dciConsumer{defaultDCIConsumer};
  RNTIManagerFacade rm;
};
// The caller's MCSTracking object (LTESniffer_Core.cc:422-426,473-499): the database it used to own is inside the library; what the main
// loop calls on it keeps working.  The library ages the database by itself on the subframe count (lsn_phy_set_mcs_update_interval), so
// update_database_dl() from the caller's wall-clock timer is an extra, idempotent update.
class MCSTracking {
public:
  void attach(lsn_phy_t* phy) { h = phy; }
  double get_interval() const { return interval; }
  void set_interval(double seconds) { interval = seconds; if (h) lsn_phy_set_mcs_update_interval(h, (uint32_t)seconds); }
  void update_database_dl() { if (h) lsn_phy_update_mcs_database(h); }
  void update_database_ul() { if (h) lsn_phy_update_mcs_database(h); }   // UL_MODE: the uplink database (same call, the mode decides)
  uint32_t nof_RNTI_member_ul() { return h ? lsn_phy_nof_tracked_rnti(h) : 0; }
  uint32_t nof_RNTI_member_dl() { return h ? lsn_phy_nof_tracked_rnti(h) : 0; }
  lsn_ue_config_t get_ue_config_rnti(uint16_t rnti) { lsn_ue_config_t c{}; if (h) lsn_phy_get_ue_config(h, rnti, &c); return c; }
private:
  lsn_phy_t* h = nullptr;
  double interval = 5.0;                               // MCSTracking.h:162
};
class HARQ;        // HARQ.h: the database lives inside the library (harq_mode 1: soft combining, DL mode; 0: the reference's only reachable value, ArgManager.cc:50); the pointer is accepted and ignored
class ULSchedule;  // ULSchedule.h: the schedule lives inside the library; the SIB2 values are learned there (decode_SIB) or given through Phy::setUlConfig

class Phy {
public:
  // Phy.h:24-36 - the reference's argument list, in its order.  dciFileName is not consumed (the DCI text log belongs to DCIToFile on the
  // caller's side of the boundary), statsFileName receives printStats(); `pcapwriter` is this library's writer (lsn_pcap_open) or nullptr
  // when the original LTESniffer_pcap_writer stays in use through setPduSink().  Trailing arguments are this library's own.
  Phy(uint32_t nof_rx_antennas, uint32_t nof_workers, const std::string& dciFileName, const std::string& statsFileName, bool skipSecondaryMetaFormats,
      double metaFormatSplitRatio, uint32_t histogramThreshold, lsn_pcap_t* pcapwriter, MCSTracking* mcs_tracking, HARQ* harq, int mcs_tracking_mode,
      int harq_mode, ULSchedule* ulsche, int device = 0, int sniffer_mode = 0 /* DL_MODE; 1 = UL_MODE */, uint32_t max_batch = 0)
      : nof_rx_antennas(nof_rx_antennas), nof_workers(nof_workers)
  {
    (void)harq; (void)ulsche;
    init(skipSecondaryMetaFormats, metaFormatSplitRatio, histogramThreshold, pcapwriter, mcs_tracking_mode, harq_mode, device, sniffer_mode, max_batch);
    if (!dciFileName.empty()) { common.dci_file = fopen(dciFileName.c_str(), "w"); if (!common.dci_file) common.dci_file = stdout; }   // PhyCommon.cc:19-24
    common.defaultDCIConsumer->setFile(common.dci_file);                                                                               // PhyCommon.cc:33
    if (!statsFileName.empty()) common.stats_file = fopen(statsFileName.c_str(), "w");
    if (mcs_tracking) mcs_tracking->attach(h);
  }
  // short form (tests, new callers)
  Phy(uint32_t nof_rx_antennas, uint32_t nof_workers, bool skipSecondaryMetaFormats, double metaFormatSplitRatio, uint32_t histogramThreshold,
      lsn_pcap_t* pcapwriter, int mcs_tracking_mode = 1, int harq_mode = 0, int device = 0, int sniffer_mode = 0)
      : nof_rx_antennas(nof_rx_antennas), nof_workers(nof_workers)
  {
    init(skipSecondaryMetaFormats, metaFormatSplitRatio, histogramThreshold, pcapwriter, mcs_tracking_mode, harq_mode, device, sniffer_mode, 0);
  }
  ~Phy() { lsn_phy_destroy(h); if (common.stats_file) fclose(common.stats_file); if (common.dci_file && common.dci_file != stdout) fclose(common.dci_file); }
  Phy(const Phy&) = delete;
  Phy& operator=(const Phy&) = delete;
  bool setCell(const lsn_cell_t& cell) { return lsn_phy_set_cell(h, &cell) == LSN_SUCCESS; }               // Phy.cc:111
  // the caller's srsran_cell_t (LTESniffer_Core.cc:292): its enums count like lsn_cell_t's fields (SRSRAN_CP_NORM 0 / EXT 1, SRSRAN_PHICH_NORM 0, SRSRAN_PHICH_R_1_6 0 ... R_2 3)
  template <class SrsranCell> bool setCell(const SrsranCell& c)
  {
    return setCell(lsn_cell_t{(uint32_t)c.nof_prb, (uint32_t)c.nof_ports, (uint32_t)c.id, (uint32_t)c.cp, (uint32_t)c.phich_length, (uint32_t)c.phich_resources, 0u});
  }
  std::shared_ptr<SubframeWorker> getAvail() { return wrap(lsn_phy_get_avail(h, 1)); }                      // Phy.cc:79 (blocking)
  std::shared_ptr<SubframeWorker> getAvailImmediate() { return wrap(lsn_phy_get_avail(h, 0)); }             // Phy.cc:84
  void putPending(std::shared_ptr<SubframeWorker> w) { lsn_phy_put_pending(h, w->h); }                      // Phy.cc:95
  void joinPending() { lsn_phy_join_pending(h); }                                                           // Phy.cc:100
  void setPduSink(lsn_pdu_sink_t cb, void* user) { lsn_phy_set_pdu_sink(h, cb, user); }
  lsn_blind_stats_t getStats() { lsn_blind_stats_t s{}; lsn_phy_get_stats(h, &s); return s; }              // PhyCommon::getStats
  PhyCommonFacade& getCommon() { return common; }                                                           // Phy.h:44
  DCIMetaFormatsFacade& getMetaFormats() { return metaFormats; }                                            // Phy.h:45
  std::vector<std::shared_ptr<SubframeWorker>>& getWorkers()                                                // Phy.h:46: every worker of the pool, by index
  {
    if (workers.empty())
      for (uint32_t i = 0; i < lsn_phy_nof_workers(h); i++) workers.push_back(wrap(lsn_phy_worker(h, i)));
    return workers;
  }
  void printStats() { common.printStats(); }                                                               // Phy.cc:144
  void setChestCFOEstimateEnable(bool, uint32_t) {}   // Phy.h:49-50: the estimator of this path always reports its CFO (getEstCfo)
  void setChestAverageSubframe(bool) {}
  void setRNTI(uint16_t) {}                           // Phy.h:48 (single-RNTI filter of the legacy path; unused by LTESniffer_Core)
  uint32_t nof_rx_antennas, nof_workers;              // Phy.h:54-55
  float getEstCfo() { return lsn_phy_get_est_cfo(h); }                                                      // SubframeWorker.cc:203
  // what ue_sync's CFO tracking does ahead of the reference's workers (LTESniffer_Core.cc:312-316,344), here inside the OFDM kernel
  bool setCfoCorrection(int mode, float cfoHz = 0.0f, float alpha = 0.25f) { return lsn_phy_set_cfo_correction(h, mode, cfoHz, alpha) == LSN_SUCCESS; }
  float getCfoCorrection() { return lsn_phy_get_cfo_correction(h); }                                       // srsran_ue_sync_get_cfo
  // which slots of the blind-decode table are computed ahead of the search (no counterpart in the reference, whose search decodes as it goes: DCISearch.cc:102-447): 0 all, 1 default, 2 test
  bool setCandidatePruning(int mode) { return lsn_phy_set_candidate_pruning(h, mode) == LSN_SUCCESS; }
  // UL_MODE: what ULSchedule::set_config hands to the workers once SIB2 is known (ULSchedule.cc:140-158)
  bool setUlConfig(uint32_t cyclicShift, uint32_t groupAssignmentPUSCH, uint32_t puschHoppingOffset = 0)    // SubframeWorker.cc:258-277
  {
    lsn_ul_cfg_t u{cyclicShift, groupAssignmentPUSCH, puschHoppingOffset};
    return lsn_phy_set_ul_config(h, &u) == LSN_SUCCESS;
  }
  // -a api_mode of LTESniffer_Core (run_api_dl_mode): identities of decoded downlink blocks go to cb, their blocks to apiPcap
  bool setApiMode(int apiMode, lsn_api_sink_t cb, void* user, lsn_pcap_t* apiPcap = nullptr) { return lsn_phy_set_api_mode(h, apiMode, cb, user, apiPcap) == LSN_SUCCESS; }
  // ULSchedule::get_config + getSIB2: true once a configuration is in use (given, or learned from the first SIB2 in UL_MODE)
  bool getUlConfig(lsn_ul_cfg_t* ul = nullptr, lsn_sib2_t* sib2 = nullptr, bool* fromSib2 = nullptr)
  {
    uint32_t f = 0;
    const bool ok = lsn_phy_get_ul_config(h, ul, sib2, &f) == 1;
    if (fromSib2) *fromSib2 = f != 0;
    return ok;
  }
  bool setRachConfig(const lsn_prach_cfg_t& p) { return lsn_phy_set_prach_config(h, &p) == LSN_SUCCESS; }    // PUSCH_Decoder::set_rach_config
  void setPrachSink(lsn_prach_sink_t cb, void* user) { lsn_phy_set_prach_sink(h, cb, user); }               // work_prach's report
  // srsran_ue_mib_decode + srsran_pbch_mib_unpack on one subframe 0 (LTESniffer_Core.cc:386-391); true when a MIB was found
  bool mibDecode(const cf_t* subframe_iq /* [nof_rx_antennas][SF_LEN] */, lsn_mib_t& mib) { return lsn_phy_mib_decode(h, subframe_iq, 0, &mib) == 1; }
  lsn_ue_config_t getUeConfig(uint16_t rnti) { lsn_ue_config_t c{}; lsn_phy_get_ue_config(h, rnti, &c); return c; }  // MCSTracking::get_ue_config_rnti
  lsn_phy_t* handle() { return h; }
private:
  void init(bool skipSecondaryMetaFormats, double metaFormatSplitRatio, uint32_t histogramThreshold, lsn_pcap_t* pcapwriter, int mcs_tracking_mode, int harq_mode,
            int device, int sniffer_mode, uint32_t max_batch)
  {
    lsn_phy_cfg_t cfg{};
    cfg.nof_rx_antennas = nof_rx_antennas; cfg.nof_workers = nof_workers; cfg.skip_secondary_meta_formats = skipSecondaryMetaFormats;
    cfg.meta_format_split_ratio = metaFormatSplitRatio; cfg.histogram_threshold = histogramThreshold; cfg.max_batch = max_batch;
    cfg.mcs_tracking_mode = mcs_tracking_mode; cfg.harq_mode = harq_mode; cfg.device = device; cfg.sniffer_mode = sniffer_mode;
    const int r = lsn_phy_create(&cfg, &h);
    if (r == LSN_ERROR_NO_DEVICE) throw std::runtime_error("ltesniffer_amd: no HIP device (this library has no CPU path)");
    if (r != LSN_SUCCESS) throw std::runtime_error("lsn_phy_create failed");
    lsn_phy_setup_default_rnti_intervals(h);  // LTESniffer_Core.cc:398-417
    if (pcapwriter) lsn_phy_set_pcap_writer(h, pcapwriter);
    common.h = h; common.rm.h = h; metaFormats.h = h;
  }
  PhyCommonFacade common;
  DCIMetaFormatsFacade metaFormats;
  std::vector<std::shared_ptr<SubframeWorker>> workers;
  static std::shared_ptr<SubframeWorker> wrap(lsn_worker_t* w) { return w ? std::shared_ptr<SubframeWorker>(new SubframeWorker(w)) : nullptr; }
  lsn_phy_t* h = nullptr;
};

// rf_search_and_decode_mib's cell search on a block of samples of one antenna (LTESniffer_Core.cc:195-204): 1 found, 0 not, < 0 error
inline int cellSearch(const cf_t* iq, uint64_t nof_samples, uint32_t nof_prb, lsn_cell_search_t& out, int force_N_id_2 = -1,
                      uint32_t nof_periods = 2, float threshold = 20.0f, int device = 0)
{
  lsn_cell_search_cfg_t c{nof_periods, force_N_id_2, threshold};
  return lsn_cell_search(device, iq, 0, nof_samples, nof_prb, &c, &out, nullptr);
}

}  // namespace lsn_amd
