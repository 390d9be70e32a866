/* oracle/_ref harness around the REFERENCE'S OWN MCS-tracking database (test infrastructure, NOT product; see srsran/standin.h).
 *
 * Compiled verbatim from /root/reference by oracle/Makefile.ref into _ref/libref_falcon_mcs.so:
 *   src/src/MCSTracking.cc          the 250-entry MCS-table database of DL_MODE with its ageing, the UE-specific configuration store (SURVEY 8 row a13:
 *                                   MCSTracking; the uplink twin of the database is the same code on another map and is not driven here)
 *   src/src/Sniffer_dependency.cc   constructors of the DCI containers update_statistic_dl takes
 * MCSTracking.cc calls no srsRAN function.  It reads the processor clock (clock()) for its ageing; the library is linked with -Bsymbolic-functions so
 * that its calls bind to the clock() below, which the test sets: 1 ms of it per subframe - the time base oracle and product use.
 * The class keeps its databases private; this file alone is compiled with the access specifiers opened to PEEK at an entry without touching it
 * (find_tracking_info_RNTI_dl refreshes the entry's time).  The reference's translation units are compiled unchanged. */
#include <time.h>
#include <unistd.h>
#include <atomic>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <sstream>
#include <string>
#include <vector>
#define private public
#include "include/MCSTracking.h"
#undef private

static clock_t g_now = 0;
extern "C" clock_t clock(void) __THROW { return g_now; }

struct ref_mcs_t { std::atomic<float> cfo{0.f}; MCSTracking* m; };

extern "C" {

void ref_mcs_set_now_ms(uint64_t ms) { g_now = (clock_t)(ms * (CLOCKS_PER_SEC / 1000)); }

ref_mcs_t* ref_mcs_new(int sniffer_mode)
{
  ref_mcs_t* h = new ref_mcs_t();
  char cwd[4096];
  char* here = getcwd(cwd, sizeof(cwd));
  if (chdir("/tmp") != 0) here = nullptr; /* the constructor opens mcs_statistic.csv in the working directory */
  h->m = new MCSTracking(DL_SNIFFER_MCS_MODE_ON, 0, false, sniffer_mode, -1, h->cfo);
  if (here && chdir(here) != 0) { }
  return h;
}
void ref_mcs_free(ref_mcs_t* h) { delete h->m; delete h; }

int ref_mcs_find(ref_mcs_t* h, uint16_t rnti) { return (int)h->m->find_tracking_info_RNTI_dl(rnti); }
void ref_mcs_update(ref_mcs_t* h, uint16_t rnti, int table) { h->m->update_RNTI_dl(rnti, (dl_sniffer_mcs_table_t)table); }
void ref_mcs_rar(ref_mcs_t* h, uint16_t rnti) { h->m->update_rar_time_crnti(rnti, clock()); }
void ref_mcs_stat(ref_mcs_t* h, uint16_t rnti, int format, int table, int en0, int en1, int ok0, int ok1, int mimo_ret, uint32_t mcs0, uint32_t mcs1)
{
  bool en[2] = {en0 != 0, en1 != 0}, ok[2] = {ok0 != 0, ok1 != 0};
  int ttype[2] = {DL_SNIFFER_NEW_TX, DL_SNIFFER_NEW_TX}; /* harq_mode 0: every transmission is a new one (DL_Sniffer_PDSCH.cc) */
  DL_Sniffer_DCI_DL mem;
  mem.format = (srsran_dci_format_t)format;
  mem.mcs_table = (dl_sniffer_mcs_table_t)table;
  srsran_pdsch_grant_t g;
  memset(&g, 0, sizeof(g));
  g.tb[0].mcs_idx = mcs0; g.tb[1].mcs_idx = mcs1;
  h->m->update_statistic_dl(rnti, en, ttype, ok, mem, mimo_ret, &g);
}
void ref_mcs_update_database(ref_mcs_t* h) { h->m->update_database_dl(); }
uint32_t ref_mcs_count(ref_mcs_t* h) { return (uint32_t)h->m->nof_RNTI_member_dl(); }
/* table of an entry without refreshing it, -1 = no entry; out (may be NULL): nof_active, nof_success_mgs, nof_msg_after_rar, has_rar, nof_unsupport_mimo + nof_pinfo + nof_other_mimo */
int ref_mcs_peek(ref_mcs_t* h, uint16_t rnti, uint32_t* out5)
{
  auto it = h->m->tracking_database_dl_mode.find(rnti);
  if (it == h->m->tracking_database_dl_mode.end()) return -1;
  if (out5) {
    out5[0] = it->second.nof_active; out5[1] = it->second.nof_success_mgs; out5[2] = it->second.nof_msg_after_rar; out5[3] = it->second.has_rar;
    out5[4] = it->second.nof_unsupport_mimo + it->second.nof_pinfo + it->second.nof_other_mimo;
  }
  return (int)it->second.mcs_table;
}
uint32_t ref_mcs_all_database_size(ref_mcs_t* h) { return (uint32_t)h->m->all_database_dl_mode.size(); }

/* UE-specific configuration store (update_ue_config_rnti / get_ue_config_rnti / the default of the default) */
void ref_mcs_set_ue_config(ref_mcs_t* h, uint16_t rnti, float p_a, uint32_t ack, uint32_t cqi, uint32_t ri, int cqi_type)
{
  ltesniffer_ue_spec_config_t c;
  c.has_ue_config = true; c.p_a = p_a;
  c.uci_config.I_offset_ack = ack; c.uci_config.I_offset_cqi = cqi; c.uci_config.I_offset_ri = ri;
  c.cqi_config.type = (srsran_cqi_type_t)cqi_type;
  h->m->update_ue_config_rnti(rnti, c);
}
void ref_mcs_get_ue_config(ref_mcs_t* h, uint16_t rnti, uint32_t* out6)
{
  ltesniffer_ue_spec_config_t c = h->m->get_ue_config_rnti(rnti);
  memcpy(&out6[0], &c.p_a, 4);
  out6[1] = c.uci_config.I_offset_ack; out6[2] = c.uci_config.I_offset_cqi; out6[3] = c.uci_config.I_offset_ri; out6[4] = (uint32_t)c.cqi_config.type;
  out6[5] = c.has_ue_config;
}

} /* extern "C" */
