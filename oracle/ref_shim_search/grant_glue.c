/* oracle/_ref harness around the REFERENCE'S OWN grant conversions (test infrastructure, NOT product; see srsran/standin.h).
 *
 * Compiled verbatim from /root/reference by oracle/Makefile.ref into _ref/libref_falcon_grant.so:
 *   lib/src/phy/falcon_phch/ul_sniffer_pusch.c   DCI 0 -> PUSCH grant: PRB allocation incl. type-1 hopping, both uplink MCS tables, row 32A
 *   lib/src/phy/falcon_phch/dl_sniffer_pdsch.c   transport-block enabling, TBS of SI / P / RA-RNTI grants (formats 1A / 1C), MIMO configuration (a12)
 * The srsRAN functions those files call (absent dependency):
 *   srsran_ra_tbs_from_idx             BOUND TO THE ORACLE's TBS table (o_tbs_from_idx) by the test
 *   srsran_ra_type2_from_riv           TS 36.213 8.1.1 (resource indication value -> length, start), written here
 *   srsran_ra_ul_compute_nof_re        symbols x PRBs x 12 (normal / extended CP, no SRS), written here
 *   srsran_ra_dl_grant_to_grant_prb_allocation, srsran_dl_fill_ra_mcs, srsran_ra_dl_compute_nof_re, srsran_dci_dl_info
 *                                      srsRAN results the DL conversion of a C-RNTI grant needs: NOT supplied - the harness never calls
 *                                      dl_sniffer_ra_dl_dci_to_grant, and dl_sniffer_compute_tb only with SI / P / RA-RNTIs (the branch that does not reach them)
 * Flat C interface for ctypes below. */
#include "falcon/phy/falcon_phch/ul_sniffer_pusch.h"
#include "falcon/phy/falcon_phch/dl_sniffer_pdsch.h"

typedef int (*tbs_fn_t)(int i_tbs, uint32_t n_prb);
static tbs_fn_t g_tbs_fn;
void ref_grant_bind(void* tbs_fn) { g_tbs_fn = (tbs_fn_t)tbs_fn; }

int srsran_ra_tbs_from_idx(uint32_t tbs_idx, uint32_t n_prb) { return g_tbs_fn((int)tbs_idx, n_prb); }
void srsran_ra_type2_from_riv(uint32_t riv, uint32_t* L_crb, uint32_t* RB_start, uint32_t nof_prb, uint32_t nof_vrb)
{
  *L_crb = riv / nof_prb + 1;
  *RB_start = riv % nof_prb;
  if (*L_crb > nof_vrb - *RB_start) {
    *L_crb = nof_prb - riv / nof_prb + 1;
    *RB_start = nof_prb - riv % nof_prb - 1;
  }
}
static uint32_t mod_bits(srsran_mod_t m) { static const uint32_t b[5] = {1, 2, 4, 6, 8}; return b[m]; }
void srsran_ra_ul_compute_nof_re(srsran_pusch_grant_t* grant, srsran_cp_t cp, uint32_t N_srs)
{
  grant->nof_symb = 2 * ((cp == SRSRAN_CP_NORM ? 7 : 6) - 1) - N_srs;
  grant->nof_re = grant->nof_symb * grant->L_prb * 12;
  grant->tb.nof_bits = grant->nof_re * mod_bits(grant->tb.mod);
}
static void unreachable(const char* what) { fprintf(stderr, "ref grant harness: %s is not supplied\n", what); abort(); }
int srsran_ra_dl_grant_to_grant_prb_allocation(const srsran_dci_dl_t* dci, srsran_pdsch_grant_t* grant, uint32_t nof_prb) { unreachable("srsran_ra_dl_grant_to_grant_prb_allocation"); return -1; }
int srsran_dl_fill_ra_mcs(srsran_ra_tb_t* tb, int last_tbs, uint32_t nprb, bool pdsch_use_tbs_index_alt) { unreachable("srsran_dl_fill_ra_mcs"); return -1; }
void srsran_ra_dl_compute_nof_re(const srsran_cell_t* cell, srsran_dl_sf_cfg_t* sf, srsran_pdsch_grant_t* grant) { unreachable("srsran_ra_dl_compute_nof_re"); }
uint32_t srsran_dci_dl_info(const srsran_dci_dl_t* dci, char* str, uint32_t len) { if (len) str[0] = 0; return 0; }

/* DCI 0 -> grant.  freq_hop_fl: -1 no hopping, 0 / 1 / 2 = type 1 (+N/4, -N/4, +N/2), 3 = type 2.  out: rc-independent fields
 * {L_prb, n_prb[0], n_prb[1], freq_hopping, modulation bits, tbs, rv, mcs_idx, nof_re}.  Returns the reference's return code. */
int ref_ul_grant(uint32_t nof_prb, uint32_t cp, uint32_t n_rb_ho, int table_256, uint32_t riv, int freq_hop_fl, uint32_t mcs_idx, int rv, int cqi_request,
                 int32_t* out9)
{
  srsran_cell_t cell; memset(&cell, 0, sizeof(cell)); cell.nof_prb = nof_prb; cell.cp = (srsran_cp_t)cp;
  srsran_ul_sf_cfg_t sf; memset(&sf, 0, sizeof(sf));
  srsran_pusch_hopping_cfg_t hop; memset(&hop, 0, sizeof(hop)); hop.n_rb_ho = n_rb_ho;
  srsran_dci_ul_t dci; memset(&dci, 0, sizeof(dci));
  dci.type2_alloc.riv = riv; dci.freq_hop_fl = freq_hop_fl; dci.tb.mcs_idx = mcs_idx; dci.tb.rv = rv; dci.cqi_request = cqi_request != 0;
  srsran_pusch_grant_t g; memset(&g, 0, sizeof(g));
  int rc = table_256 ? ulsniffer_ra_ul_dci_to_grant_256(&cell, &sf, &hop, &dci, &g) : ul_sniffer_ra_ul_dci_to_grant(&cell, &sf, &hop, &dci, &g);
  out9[0] = (int32_t)g.L_prb; out9[1] = (int32_t)g.n_prb[0]; out9[2] = (int32_t)g.n_prb[1]; out9[3] = (int32_t)g.freq_hopping;
  out9[4] = (int32_t)mod_bits(g.tb.mod); out9[5] = g.tb.tbs; out9[6] = g.tb.rv; out9[7] = (int32_t)g.tb.mcs_idx; out9[8] = (int32_t)g.nof_re;
  return rc;
}

/* MIMO configuration of a downlink grant: dl_sniffer_config_mimo.  out: {tx_scheme, pmi, nof_layers} */
int ref_config_mimo(uint32_t nof_ports, int format, uint32_t pinfo, uint32_t nof_tb, int32_t* out3)
{
  srsran_cell_t cell; memset(&cell, 0, sizeof(cell)); cell.nof_ports = nof_ports;
  srsran_dci_dl_t dci; memset(&dci, 0, sizeof(dci)); dci.pinfo = pinfo;
  srsran_pdsch_grant_t g; memset(&g, 0, sizeof(g)); g.nof_tb = nof_tb;
  g.tb[0].enabled = nof_tb >= 1; g.tb[1].enabled = nof_tb >= 2;
  int rc = dl_sniffer_config_mimo(&cell, (srsran_dci_format_t)format, &dci, &g);
  out3[0] = (int32_t)g.tx_scheme; out3[1] = (int32_t)g.pmi; out3[2] = (int32_t)g.nof_layers;
  return rc;
}

/* transport blocks of a grant: dl_sniffer_compute_tb.  User RNTIs: only the enabling (the TBS needs srsRAN's MCS table: call with tbs_too = 0 and the
 * function stops before it); SI / P / RA-RNTI: format 1A (n_prb1a -> column 2 or 3 of the TBS table) and format 1C (the reference's literal table).
 * out: {nof_tb, enabled0, enabled1, tbs0, mod0 bits} */
int ref_compute_tb_common(int format, uint16_t rnti, uint32_t mcs_idx, int nprb1a_is_2, int32_t* out5)
{
  srsran_dci_dl_t dci; memset(&dci, 0, sizeof(dci));
  dci.format = (srsran_dci_format_t)format; dci.rnti = rnti; dci.tb[0].mcs_idx = mcs_idx;
  dci.tb[1].mcs_idx = 0; dci.tb[1].rv = 1; /* disabled second block */
  dci.type2_alloc.n_prb1a = nprb1a_is_2 ? SRSRAN_RA_TYPE2_NPRB1A_2 : SRSRAN_RA_TYPE2_NPRB1A_3;
  srsran_pdsch_grant_t g; memset(&g, 0, sizeof(g));
  int rc = dl_sniffer_compute_tb(false, &dci, &g);
  out5[0] = (int32_t)g.nof_tb; out5[1] = g.tb[0].enabled; out5[2] = g.tb[1].enabled; out5[3] = g.tb[0].tbs; out5[4] = (int32_t)mod_bits(g.tb[0].mod);
  return rc;
}
