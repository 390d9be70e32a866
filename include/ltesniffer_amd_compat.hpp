// ltesniffer_amd_compat.hpp - the reference's worker-side class names in the global namespace, for an UNCHANGED LTESniffer_Core.
//
// What a maintainer does: in src/include/LTESniffer_Core.h replace
//     #include "include/SubframeWorker.h"   "include/WorkerThread.h"   "MCSTracking.h"   "ULSchedule.h"   "Phy.h"   "PcapWriter.h"   "HARQ.h"
// by  #include "ltesniffer_amd_compat.hpp"
// and link libltesniffer_amd.so instead of the worker-side objects.  LTESniffer_Core.cc itself stays as it is: its constructor (:63-90), main loop (:292-299,
// 361-451), timers (:473-500), shutdown (:531-562) and forwarding members (:603-621) compile against the classes below - tests/test_reference_caller.py does
// exactly that with the reference's file in the CPU suite.  The objects the core still OWNS (LTESniffer_pcap_writer, MCSTracking, HARQ, UL_HARQ, ULSchedule;
// LTESniffer_Core.h:99-110) keep their constructors and the members the core calls; their state lives inside the library behind the Phy they are handed to.
#pragma once
#define LSN_AMD_SRSRAN_CF_T 1   // the core's srsRAN headers are in scope (cf_t = float _Complex): getBuffers() goes into srsran_ue_sync_zerocopy as it is
#include "ltesniffer_amd.hpp"
#include <atomic>
#include <iomanip>    // (LTESniffer_Core.cc uses std::setw and got the header through the swapped ones)
#include <iostream>

// PcapWriter.h:39-51 - open() of the core's constructor (:68) creates the two capture files the reference writes; close() at shutdown (:582)
class LTESniffer_pcap_writer {
public:
  void open(const std::string& filename, const std::string& api_filename, uint32_t /*ue_id*/ = 0)
  {
    h = lsn_pcap_open(filename.c_str());
    api = lsn_pcap_open(api_filename.c_str());
  }
  void close() { if (h) lsn_pcap_close(h); if (api) lsn_pcap_close(api); h = api = nullptr; }
  lsn_pcap_t* handle() { return h; }
  lsn_pcap_t* apiHandle() { return api; }
private:
  lsn_pcap_t *h = nullptr, *api = nullptr;
};

// MCSTracking.h:95-170 as LTESniffer_Core uses it: constructor (:39), the API-header counter (:422-426), the database timers (:473-500) and the final report
// (:531-541).  The tracking tables are the library's (lsn_lte.cc); the print calls report the number of tracked RNTIs, the merge calls have nothing to merge
// (one database per Phy, no per-thread copies).
class MCSTracking : public lsn_amd::MCSTracking {
public:
  MCSTracking(int tracking_mode, uint16_t target_rnti, bool en_debug, int sniffer_mode, int api_mode, std::atomic<float>& est_cfo)
      : tracking_mode(tracking_mode), target_rnti(target_rnti), en_debug(en_debug), sniffer_mode(sniffer_mode), api_mode(api_mode), est_cfo(est_cfo) {}
  int get_nof_api_msg() const { return nof_api_msg; }
  void increase_nof_api_msg() { nof_api_msg++; }
  void reset_nof_api_msg() { nof_api_msg = 0; }
  uint16_t get_target_rnti() const { return target_rnti; }
  bool get_debug_mode() const { return en_debug; }
  int get_sniffer_mode() const { return sniffer_mode; }
  int get_api_mode() const { return api_mode; }
  std::atomic<float>& get_est_cfo() { return est_cfo; }
  void print_database_dl() { printf("[MCS tracking] %u RNTIs in the downlink table database\n", nof_RNTI_member_dl()); }
  void print_database_ul() { printf("[MCS tracking] %u RNTIs in the uplink table database\n", nof_RNTI_member_ul()); }
  void merge_all_database_dl() {}
  void merge_all_database_ul() {}
  void print_all_database_dl() { print_database_dl(); }
  void print_all_database_ul() { print_database_ul(); }
private:
  int tracking_mode; uint16_t target_rnti; bool en_debug; int sniffer_mode, api_mode, nof_api_msg = 0;
  std::atomic<float>& est_cfo;
};

// HARQ.h / ULSchedule.h: constructed and configured by the core (:43,70,72,494), handed to Phy, state inside the library
class UL_HARQ {};
class HARQ {
public:
  void init_HARQ(int mode) { harq_mode = mode; }
  void updateHARQDatabase() {}   // :494 - the library ages its HARQ entities on the subframe count (HarqDatabase, lsn_lte.cc)
  int mode() const { return harq_mode; }
private:
  int harq_mode = 0;
};
class ULSchedule {
public:
  ULSchedule(uint16_t target_rnti, UL_HARQ* ul_harq, bool en_debug) : target_rnti(target_rnti), ul_harq(ul_harq), en_debug(en_debug) {}
  void set_multi_offset(int sniffer_mode) { multi_offset = sniffer_mode; }   // ULSchedule.h: 1 in UL_MODE
private:
  uint16_t target_rnti; UL_HARQ* ul_harq; bool en_debug; int multi_offset = 0;
};

// Phy.h:22-66 with the reference's own pointer types in the constructor (LTESniffer_Core.cc:74-86)
class Phy : public lsn_amd::Phy {
public:
  Phy(uint32_t nof_rx_antennas, uint32_t nof_workers, const std::string& dciFileName, const std::string& statsFileName, bool skipSecondaryMetaFormats,
      double metaFormatSplitRatio, uint32_t histogramThreshold, LTESniffer_pcap_writer* pcapwriter, MCSTracking* mcs_tracking, HARQ* harq, int mcs_tracking_mode,
      int harq_mode, ULSchedule* ulsche)
      : lsn_amd::Phy(nof_rx_antennas, nof_workers, dciFileName, statsFileName, skipSecondaryMetaFormats, metaFormatSplitRatio, histogramThreshold,
                     pcapwriter ? pcapwriter->handle() : nullptr, mcs_tracking, nullptr, mcs_tracking_mode, harq_mode, nullptr, /*device*/ 0,
                     mcs_tracking ? mcs_tracking->get_sniffer_mode() : 0)
  {
    (void)harq; (void)ulsche;
    if (mcs_tracking && mcs_tracking->get_api_mode() >= 0)   // -z api_mode: identities go to the API capture file the writer opened (PcapWriter.cc:120-145,177-190)
      setApiMode(mcs_tracking->get_api_mode(), nullptr, nullptr, pcapwriter ? pcapwriter->apiHandle() : nullptr);
  }
};

using lsn_amd::SubframeWorker;
using lsn_amd::SubframeInfoConsumer;
using lsn_amd::DCIToFile;
using lsn_amd::DCIConsumerList;
typedef lsn_amd::RNTIManagerFacade RNTIManager;   // LTESniffer_Core::getRNTIManager() returns phy->getCommon().getRNTIManager() (:610-613)
