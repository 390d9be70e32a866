/* oracle/_ref harness around the REFERENCE'S OWN downlink HARQ database (test infrastructure, NOT product; see srsran/standin.h).
 *
 * Compiled verbatim from /root/reference by oracle/Makefile.ref into _ref/libref_falcon_harq.so:
 *   src/src/HARQ.cc     HARQ::is_retransmission / updateHARQRNTI / updateHARQDatabase / getlastTbs - the verdict "new transmission, retransmission to combine,
 *                       already decoded, database full" the PDSCH decoder asks for per transport block (DL_Sniffer_PDSCH.cc:943-1020)
 * The three srsRAN soft-buffer calls (allocate / free / reset the LLR store) are no-ops here: the verdicts do not read the buffers.  clock() is bound
 * (-Bsymbolic-functions) to the settable clock below, 1 ms per subframe, as in mcs_glue.cc.  The grant records are filled as DL_Sniffer_PDSCH.cc:947-953 /
 * 1008-1014 fill them. */
#include <time.h>
#include "include/HARQ.h"

static clock_t g_now = 0;
extern "C" clock_t clock(void) __THROW { return g_now; }

extern "C" {
int srsran_softbuffer_rx_init(srsran_softbuffer_rx_t*, uint32_t) { return SRSRAN_SUCCESS; }
void srsran_softbuffer_rx_free(srsran_softbuffer_rx_t*) {}
void srsran_softbuffer_rx_reset(srsran_softbuffer_rx_t*) {}

void ref_harq_set_now_ms(uint64_t ms) { g_now = (clock_t)(ms * (CLOCKS_PER_SEC / 1000)); }
HARQ* ref_harq_new(void) { HARQ* h = new HARQ(); h->init_HARQ(DL_SNIFFER_HARQ_MODE_ON); return h; }
void ref_harq_free(HARQ* h) { delete h; }
uint32_t ref_harq_size(HARQ* h) { return (uint32_t)h->harqBufferSize(); }
static dl_sniffer_harq_grant_t mk(bool last_decoded, bool ndi, int rv, int tbs)
{
  dl_sniffer_harq_grant_t g = {};
  g.last_decoded = last_decoded; g.ndi = ndi; g.ndi_present = true; g.rv = rv; g.tbs = tbs; g.is_first_transmission = false;
  return g;
}
int ref_harq_is_retransmission(HARQ* h, uint16_t rnti, int pid, int tid, int ndi, int rv, int tbs, uint32_t sfn, uint32_t sf_idx)
{
  return h->is_retransmission(rnti, pid, tid, mk(false, ndi != 0, rv, tbs), sfn, sf_idx);
}
void ref_harq_update(HARQ* h, uint16_t rnti, int pid, int tid, uint32_t sfn, uint32_t sf_idx, int decoded, int ndi, int rv, int tbs)
{
  h->updateHARQRNTI(rnti, pid, tid, sfn, sf_idx, mk(decoded != 0, ndi != 0, rv, tbs));
}
void ref_harq_update_database(HARQ* h) { h->updateHARQDatabase(); }
int ref_harq_last_tbs(HARQ* h, uint16_t rnti, int pid, int tid) { return h->getlastTbs(rnti, pid, tid); }
}
