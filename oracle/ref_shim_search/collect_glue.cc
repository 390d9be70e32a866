/* oracle/_ref harness around the REFERENCE'S OWN DCI collection: DCI bits -> unpacked DCI -> grants of both MCS tables -> the per-subframe
 * lists the PDSCH / PUSCH decoders walk (test infrastructure, NOT product; see srsran/standin.h).
 *
 * Compiled verbatim from /root/reference by oracle/Makefile.ref into _ref/libref_falcon_collect.so:
 *   src/src/DCICollection.cc                      addCandidate: MCS table of the RNTI, hopping configuration, RB collision maps, last TBS of a
 *                                                 reserved MCS from the HARQ database, transport blocks without bits disabled   
 *   lib/src/phy/falcon_phch/falcon_dci.c          srsran_dci_msg_to_trace_timestamp: which conversions run for which table, what a failure
 *                                                 does to the entry (RNTI 0), convert_*; the RAR grant -> DCI 0 functions (:636-683)   (a11)
 *   lib/src/phy/falcon_phch/dl_sniffer_pdsch.c    dl_sniffer_ra_dl_dci_to_grant / dl_sniffer_compute_tb INCLUDING the C-RNTI branch   (a12)
 *   lib/src/phy/falcon_phch/ul_sniffer_pusch.c    both uplink conversions (a11 / a15)
 *   src/src/ULSchedule.cc                         DCI 0 -> PUSCH 4 subframes later, RAR grant 6 (a15; DCICollection reads its SIB2)
 *   src/src/MCSTracking.cc, HARQ.cc, Sniffer_dependency.cc, DCIPrint.cc   what DCICollection consults
 *
 * What those files call in srsRAN (absent dependency) is supplied here, in two kinds:
 *   BOUND TO THE ORACLE by function pointer (ref_collect_bind): bit unpacking of a DCI payload into fields (srsran_dci_msg_unpack_pdsch / _pusch:
 *     the oracle's o_dci_unpack_dl / _ul, translated field by field below) and the TBS table (srsran_ra_tbs_from_idx).
 *   WRITTEN HERE FROM THE STANDARD, independently of the oracle's text (a second statement the oracle's o_dci.c is held against):
 *     srsran_ra_dl_grant_to_grant_prb_allocation   TS 36.213 7.1.6.1-7.1.6.3 (type 0 bitmap, type 1 RBG subsets by enumeration, type 2 localized and
 *                                                   distributed - the interleaver matrix of TS 36.211 6.2.3.2 is BUILT and read out, no closed form)
 *     srsran_dl_fill_ra_mcs                        TS 36.213 Tables 7.1.7.1-1 / 7.1.7.1-1A typed out here
 *     srsran_ra_dl_compute_nof_re                  data REs of the allocation by inclusion / exclusion per (symbol, PRB): 12 - CRS - centre band + both
 *     srsran_ra_ul_dci_to_grant                    the reference's own restatement of it (ul_sniffer_ra_ul_dci_to_grant, ul_sniffer_pusch.c:89-136)
 *   Error conventions (what an impossible allocation returns) are srsRAN's and not in the tree: this file follows the oracle's (no PRB -> error).
 * clock() is bound to a settable clock as in mcs_glue.cc. */
#include <time.h>
#include <unistd.h>
#include <atomic>
#include <vector>
#include "include/DCICollection.h"
#include "falcon/phy/falcon_phch/falcon_dci.h"

static clock_t g_now = 0;
extern "C" clock_t clock(void) __THROW { return g_now; }

/* ---- the oracle's unpacked DCI, as lsn_oracle.h lays it out (repeated here so that this file needs no oracle header) ---- */
struct o_dci_tb_m { uint32_t mcs_idx; int rv; uint32_t ndi; uint32_t cw_idx; };
struct o_dci_dl_m {
  uint16_t rnti; int format; uint32_t L, ncce; int alloc_type; uint32_t rbg_bitmask; uint32_t t1_vrb_bitmask, t1_rbg_subset, t1_shift;
  uint32_t riv; int t2_dist; int t2_ngap2; int t2_nprb1a_is2; uint32_t pid; o_dci_tb_m tb[2]; uint32_t tb_cw_swap; uint32_t pinfo; uint32_t tpc; int is_ra_order;
};
struct o_dci_ul_m { uint16_t rnti; uint32_t L, ncce; uint32_t freq_hop_fl, riv, mcs_idx; int rv; uint32_t ndi, tpc, n_dmrs, cqi_req; int hop_type; };
struct o_cell_m { uint32_t nof_prb, nof_ports, id, phich_ng_x6, pusch_hop_offset, cp; };
typedef int (*unpack_dl_fn_t)(const o_cell_m*, const uint8_t*, uint32_t, int, uint16_t, o_dci_dl_m*);
typedef int (*unpack_ul_fn_t)(const o_cell_m*, const uint8_t*, uint32_t, uint16_t, o_dci_ul_m*);
typedef int (*tbs_fn_t)(int, uint32_t);
static unpack_dl_fn_t g_unpack_dl;
static unpack_ul_fn_t g_unpack_ul;
static tbs_fn_t g_tbs;

static o_cell_m ocell_of(const srsran_cell_t* c)
{
  o_cell_m o; memset(&o, 0, sizeof(o));
  o.nof_prb = c->nof_prb; o.nof_ports = c->nof_ports; o.id = c->id; o.phich_ng_x6 = 1; o.cp = (uint32_t)c->cp;
  return o;
}

/* ---------------- TS 36.213 / 36.211 pieces, written for this harness ---------------- */
static uint32_t rbg_size(uint32_t n) { return n <= 10 ? 1 : n <= 26 ? 2 : n <= 63 ? 3 : 4; } /* Table 7.1.6.1-1 */
static uint32_t gap_of(uint32_t n, bool second) /* TS 36.211 Table 6.2.3.2-1 */
{
  if (n <= 10) return (n + 1) / 2;
  if (n == 11) return 4;
  if (n <= 19) return 8;
  if (n <= 26) return 12;
  if (n <= 44) return 18;
  if (n <= 49) return 27;
  if (n <= 63) return second ? 9 : 27;
  if (n <= 79) return second ? 16 : 32;
  return second ? 16 : 48;
}
static uint32_t nvrb_dist(uint32_t n, bool second)
{
  const uint32_t g = gap_of(n, second);
  return second ? (n / (2 * g)) * 2 * g : 2 * (g < n - g ? g : n - g);
}
/* distributed VRB -> PRB of both slots by building the interleaver of 6.2.3.2: N~_VRB numbers written row by row into 4 columns, nulls in the last
 * N_null / 2 rows of columns 2 and 4, read column by column; a VRB's PRB index is its position in the read-out */
static void dvrb_tables(uint32_t n, bool second, std::vector<uint32_t>& even, std::vector<uint32_t>& odd)
{
  const uint32_t g = gap_of(n, second), nv = nvrb_dist(n, second), unit = second ? 2 * g : nv, P = rbg_size(n);
  even.assign(nv, 0); odd.assign(nv, 0);
  if (unit == 0) return;
  const uint32_t rows = ((unit + 4 * P - 1) / (4 * P)) * P, nnull = 4 * rows - unit;
  std::vector<int> m(rows * 4, -1);
  uint32_t v = 0;
  for (uint32_t r = 0; r < rows; r++)
    for (uint32_t c = 0; c < 4; c++) {
      const bool is_null = (c == 1 || c == 3) && r >= rows - nnull / 2;
      if (!is_null && v < unit) m[r * 4 + c] = (int)v++;
    }
  std::vector<uint32_t> pos(unit, 0);
  uint32_t k = 0;
  for (uint32_t c = 0; c < 4; c++)
    for (uint32_t r = 0; r < rows; r++)
      if (m[r * 4 + c] >= 0) pos[(uint32_t)m[r * 4 + c]] = k++;
  for (uint32_t i = 0; i < nv; i++) {
    const uint32_t blk = i / unit, e = pos[i % unit], o = (e + unit / 2) % unit;
    even[i] = (e < unit / 2 ? e : e + g - unit / 2) + unit * blk;
    odd[i] = (o < unit / 2 ? o : o + g - unit / 2) + unit * blk;
  }
}
static void riv_decode(uint32_t riv, uint32_t n, uint32_t* len, uint32_t* start) /* 36.213 7.1.6.3: RIV = n (L - 1) + S or n (n - L + 1) + (n - 1 - S) */
{
  uint32_t L = riv / n + 1, S = riv % n;
  if (L + S > n) { L = n - L + 2; S = n - 1 - S; }
  *len = L; *start = S;
}

extern "C" {

uint32_t srsran_mod_bits_x_symbol(srsran_mod_t mod)
{
  switch (mod) { case SRSRAN_MOD_BPSK: return 1; case SRSRAN_MOD_QPSK: return 2; case SRSRAN_MOD_16QAM: return 4; case SRSRAN_MOD_64QAM: return 6; case SRSRAN_MOD_256QAM: return 8; }
  return 0;
}
uint32_t srsran_bit_pack(uint8_t** bits, int nof_bits)
{
  uint32_t v = 0;
  for (int i = 0; i < nof_bits; i++) v |= (uint32_t)((*bits)[i] & 1) << (nof_bits - i - 1);
  *bits += nof_bits;
  return v;
}
void srsran_bit_unpack(uint32_t value, uint8_t** bits, int nof_bits)
{
  for (int i = 0; i < nof_bits; i++) (*bits)[i] = (value >> (nof_bits - i - 1)) & 1;
  *bits += nof_bits;
}
int srsran_ra_tbs_from_idx(uint32_t tbs_idx, uint32_t n_prb) { return g_tbs((int)tbs_idx, n_prb); }
void srsran_ra_type2_from_riv(uint32_t riv, uint32_t* L_crb, uint32_t* RB_start, uint32_t nof_prb, uint32_t nof_vrb)
{
  (void)nof_vrb;
  riv_decode(riv, nof_prb, L_crb, RB_start);
}
static uint32_t mod_bits(srsran_mod_t m) { return srsran_mod_bits_x_symbol(m); }
void srsran_ra_ul_compute_nof_re(srsran_pusch_grant_t* grant, srsran_cp_t cp, uint32_t N_srs)
{
  grant->nof_symb = 2 * ((cp == SRSRAN_CP_NORM ? 7 : 6) - 1) - N_srs;
  grant->nof_re = grant->nof_symb * grant->L_prb * 12;
  grant->tb.nof_bits = grant->nof_re * mod_bits(grant->tb.mod);
}
uint32_t srsran_dci_dl_info(const srsran_dci_dl_t*, char* str, uint32_t len) { if (len) str[0] = 0; return 0; }
int srsran_softbuffer_rx_init(srsran_softbuffer_rx_t*, uint32_t) { return SRSRAN_SUCCESS; }
void srsran_softbuffer_rx_free(srsran_softbuffer_rx_t*) {}
void srsran_softbuffer_rx_reset(srsran_softbuffer_rx_t*) {}
const char* srsran_dci_format_string(srsran_dci_format_t) { return "?"; }

/* srsRAN's DCI 0 -> grant for the 64QAM-capable uplink table: the reference carries its own statement of it next to the 256QAM one */
int srsran_ra_ul_dci_to_grant(srsran_cell_t* cell, srsran_ul_sf_cfg_t* sf, srsran_pusch_hopping_cfg_t* hopping_cfg, srsran_dci_ul_t* dci, srsran_pusch_grant_t* grant)
{
  return ul_sniffer_ra_ul_dci_to_grant(cell, sf, hopping_cfg, dci, grant);
}

/* ---- DCI bits -> fields: the oracle's unpack, translated member by member ---- */
int srsran_dci_msg_unpack_pdsch(srsran_cell_t* cell, srsran_dl_sf_cfg_t*, srsran_dci_cfg_t*, srsran_dci_msg_t* msg, srsran_dci_dl_t* dci)
{
  o_cell_m oc = ocell_of(cell);
  o_dci_dl_m d; memset(&d, 0, sizeof(d));
  d.L = msg->location.L; d.ncce = msg->location.ncce;
  memset(dci, 0, sizeof(*dci));
  if (g_unpack_dl(&oc, msg->payload, msg->nof_bits, (int)msg->format, msg->rnti, &d)) return SRSRAN_ERROR;
  dci->rnti = msg->rnti; dci->format = msg->format; dci->location = msg->location;
  dci->alloc_type = (srsran_ra_type_t)d.alloc_type;
  dci->type0_alloc.rbg_bitmask = d.rbg_bitmask;
  dci->type1_alloc.vrb_bitmask = d.t1_vrb_bitmask; dci->type1_alloc.rbg_subset = d.t1_rbg_subset; dci->type1_alloc.shift = d.t1_shift != 0;
  dci->type2_alloc.riv = d.riv; dci->type2_alloc.mode = d.t2_dist; dci->type2_alloc.n_gap = d.t2_ngap2;
  dci->type2_alloc.n_prb1a = d.t2_nprb1a_is2 ? SRSRAN_RA_TYPE2_NPRB1A_2 : SRSRAN_RA_TYPE2_NPRB1A_3;
  for (int i = 0; i < 2; i++) { dci->tb[i].mcs_idx = d.tb[i].mcs_idx; dci->tb[i].rv = d.tb[i].rv; dci->tb[i].ndi = d.tb[i].ndi != 0; dci->tb[i].cw_idx = d.tb[i].cw_idx; }
  dci->tb_cw_swap = d.tb_cw_swap != 0; dci->pinfo = d.pinfo; dci->pid = d.pid; dci->tpc_pucch = (uint8_t)d.tpc; dci->is_ra_order = d.is_ra_order != 0;
  return SRSRAN_SUCCESS;
}
int srsran_dci_msg_unpack_pusch(srsran_cell_t* cell, srsran_dl_sf_cfg_t*, srsran_dci_cfg_t*, srsran_dci_msg_t* msg, srsran_dci_ul_t* dci)
{
  o_cell_m oc = ocell_of(cell);
  o_dci_ul_m d; memset(&d, 0, sizeof(d));
  d.L = msg->location.L; d.ncce = msg->location.ncce;
  memset(dci, 0, sizeof(*dci));
  if (g_unpack_ul(&oc, msg->payload, msg->nof_bits, msg->rnti, &d)) return SRSRAN_ERROR;
  dci->rnti = msg->rnti; dci->format = SRSRAN_DCI_FORMAT0; dci->location = msg->location;
  dci->type2_alloc.riv = d.riv; dci->freq_hop_fl = d.hop_type;
  dci->tb.mcs_idx = d.mcs_idx; dci->tb.rv = d.rv; dci->tb.ndi = d.ndi != 0;
  dci->n_dmrs = d.n_dmrs; dci->cqi_request = d.cqi_req != 0; dci->tpc_pusch = (uint8_t)d.tpc;
  return SRSRAN_SUCCESS;
}

/* ---- TS 36.213 7.1.6: resource allocation of a downlink DCI ---- */
int srsran_ra_dl_grant_to_grant_prb_allocation(const srsran_dci_dl_t* dci, srsran_pdsch_grant_t* grant, uint32_t nof_prb)
{
  const uint32_t n = nof_prb, P = rbg_size(n), nrbg = (n + P - 1) / P;
  memset(grant->prb_idx, 0, sizeof(grant->prb_idx));
  grant->nof_prb = 0;
  if (dci->alloc_type == SRSRAN_RA_ALLOC_TYPE0) { /* one bit per RBG, RBG 0 in the most significant bit */
    for (uint32_t prb = 0; prb < n; prb++) {
      const uint32_t rbg = prb / P;
      if ((dci->type0_alloc.rbg_bitmask >> (nrbg - 1 - rbg)) & 1u) { grant->prb_idx[0][prb] = grant->prb_idx[1][prb] = true; grant->nof_prb++; }
    }
  } else if (dci->alloc_type == SRSRAN_RA_ALLOC_TYPE1) { /* the PRBs of RBG subset p = RBGs p, p + P, p + 2 P, ... in increasing order */
    uint32_t logp = 0;
    while ((1u << logp) < P) logp++;
    const uint32_t nbits = nrbg - logp - 1, p = dci->type1_alloc.rbg_subset;
    if (p < P) {
      std::vector<uint32_t> subset;
      for (uint32_t rbg = p; rbg < nrbg; rbg += P)
        for (uint32_t prb = rbg * P; prb < rbg * P + P && prb < n; prb++) subset.push_back(prb);
      const uint32_t shift = dci->type1_alloc.shift ? (uint32_t)subset.size() - nbits : 0;
      for (uint32_t i = 0; i < nbits; i++)
        if ((dci->type1_alloc.vrb_bitmask >> (nbits - 1 - i)) & 1u)
          if (i + shift < subset.size()) { const uint32_t prb = subset[i + shift]; grant->prb_idx[0][prb] = grant->prb_idx[1][prb] = true; grant->nof_prb++; }
    } else {
      /* ceil(log2 P) bits can name a subset p >= P (P = 3: p = 3) that 7.1.6.2 does not define; only a false DCI carries it.  An implementation that evaluates
       * the section's closed forms anyway (the oracle does, and so - as far as can be told without its text - does srsRAN) gets this: */
      const uint32_t sub = ((n - 1) / (P * P)) * P, shift = dci->type1_alloc.shift ? sub - nbits : 0;
      for (uint32_t i = 0; i < nbits; i++)
        if ((dci->type1_alloc.vrb_bitmask >> (nbits - 1 - i)) & 1u) {
          const uint32_t prb = ((i + shift) / P) * P * P + p * P + (i + shift) % P;
          if (prb < n) { grant->prb_idx[0][prb] = grant->prb_idx[1][prb] = true; grant->nof_prb++; }
        }
    }
  } else if (dci->alloc_type == SRSRAN_RA_ALLOC_TYPE2) {
    const bool second = dci->type2_alloc.n_gap != 0;
    uint32_t len, start;
    if (dci->format == SRSRAN_DCI_FORMAT1C) { /* in steps of N_RB^step over floor(N_VRB / step) positions */
      const uint32_t step = n < 50 ? 2 : 4, q = nvrb_dist(n, second) / step;
      if (q == 0) return SRSRAN_ERROR;
      riv_decode(dci->type2_alloc.riv, q, &len, &start);
      len *= step; start *= step;
    } else {
      riv_decode(dci->type2_alloc.riv, n, &len, &start);
    }
    if (dci->type2_alloc.mode == 0) {
      for (uint32_t prb = start; prb < start + len && prb < n; prb++) { grant->prb_idx[0][prb] = grant->prb_idx[1][prb] = true; grant->nof_prb++; }
    } else {
      std::vector<uint32_t> even, odd;
      dvrb_tables(n, second, even, odd);
      if (even.empty()) return SRSRAN_ERROR;
      for (uint32_t v = start; v < start + len; v++)
        if (v < even.size() && even[v] < n && odd[v] < n) { grant->prb_idx[0][even[v]] = true; grant->prb_idx[1][odd[v]] = true; grant->nof_prb++; }
    }
  } else {
    return SRSRAN_ERROR;
  }
  return grant->nof_prb > 0 ? SRSRAN_SUCCESS : SRSRAN_ERROR;
}

/* ---- TS 36.213 Table 7.1.7.1-1 (a) and 7.1.7.1-1A (b): I_MCS -> modulation order, I_TBS (-1: reserved, size of the previous transmission) ---- */
static const int8_t mcs_a[32][2] = {{2, 0}, {2, 1}, {2, 2}, {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7}, {2, 8}, {2, 9}, {4, 9}, {4, 10}, {4, 11}, {4, 12}, {4, 13}, {4, 14},
                                    {4, 15}, {6, 15}, {6, 16}, {6, 17}, {6, 18}, {6, 19}, {6, 20}, {6, 21}, {6, 22}, {6, 23}, {6, 24}, {6, 25}, {6, 26}, {2, -1}, {4, -1}, {6, -1}};
static const int8_t mcs_b[32][2] = {{2, 0}, {2, 2}, {2, 4}, {2, 6}, {2, 8}, {4, 10}, {4, 11}, {4, 12}, {4, 13}, {4, 14}, {4, 15}, {6, 16}, {6, 17}, {6, 18}, {6, 19}, {6, 20},
                                    {6, 21}, {6, 22}, {6, 23}, {6, 24}, {8, 25}, {8, 27}, {8, 28}, {8, 29}, {8, 30}, {8, 31}, {8, 32}, {8, 33}, {2, -1}, {4, -1}, {6, -1}, {8, -1}};
int srsran_dl_fill_ra_mcs(srsran_ra_tb_t* tb, int last_tbs, uint32_t nprb, bool pdsch_use_tbs_index_alt)
{
  const int8_t* row = (pdsch_use_tbs_index_alt ? mcs_b : mcs_a)[tb->mcs_idx & 31];
  tb->mod = row[0] == 2 ? SRSRAN_MOD_QPSK : row[0] == 4 ? SRSRAN_MOD_16QAM : row[0] == 6 ? SRSRAN_MOD_64QAM : SRSRAN_MOD_256QAM;
  int tbs = 0;
  if (row[1] >= 0) { tbs = g_tbs(row[1], nprb); tb->tbs = tbs; } else { tb->tbs = last_tbs; }
  return tbs;
}

/* ---- data REs of an allocation (normal / extended CP, FDD): per symbol and PRB, 12 minus the CRS REs minus the centre 72 carriers on PSS / SSS / PBCH
 * symbols plus what was taken off twice ---- */
void srsran_ra_dl_compute_nof_re(const srsran_cell_t* cell, srsran_dl_sf_cfg_t* sf, srsran_pdsch_grant_t* grant)
{
  const uint32_t n = cell->nof_prb, nsl = cell->cp == SRSRAN_CP_NORM ? 7 : 6, sf_idx = sf->tti % 10;
  const uint32_t first = sf->cfi + (n <= 10 ? 1 : 0);
  const int c0 = (int)(6 * n) - 36, c1 = c0 + 72;
  uint32_t total = 0;
  for (uint32_t l = first; l < 2 * nsl; l++) {
    const uint32_t slot = l / nsl, q = l % nsl;
    int crs_step = 0, crs_off = 0; /* CRS REs of this symbol: every crs_step-th carrier from crs_off */
    if (q == 0 || q == nsl - 3) {
      if (cell->nof_ports >= 2) { crs_step = 3; crs_off = cell->id % 3; } else { crs_step = 6; crs_off = ((q == 0 ? 0 : 3) + cell->id % 6) % 6; }
    } else if (q == 1 && cell->nof_ports == 4) { crs_step = 3; crs_off = cell->id % 3; }
    bool centre = false;
    if ((sf_idx == 0 || sf_idx == 5) && slot == 0 && (q == nsl - 1 || q == nsl - 2)) centre = true; /* PSS, SSS */
    if (sf_idx == 0 && slot == 1 && q <= 3) centre = true;                                           /* PBCH */
    for (uint32_t prb = 0; prb < n; prb++) {
      if (!grant->prb_idx[slot][prb]) continue;
      const int k0 = (int)(12 * prb), k1 = k0 + 12;
      int crs = 0, taken = 0, both = 0;
      for (int k = k0; k < k1; k++) {
        const bool is_crs = crs_step && (k % crs_step) == crs_off, in_c = centre && k >= c0 && k < c1;
        crs += is_crs; taken += in_c; both += is_crs && in_c;
      }
      total += (uint32_t)(12 - crs - taken + both);
    }
  }
  grant->nof_re = total;
  for (int i = 0; i < SRSRAN_MAX_CODEWORDS; i++)
    if (grant->tb[i].enabled) grant->tb[i].nof_bits = grant->nof_re * mod_bits(grant->tb[i].mod);
}

} /* extern "C" */

/* ---------------- the harness ---------------- */
struct ref_collect_t {
  srsran_cell_t cell;
  std::atomic<float> cfo{0.f};
  MCSTracking* mcs;
  HARQ* harq;
  ULSchedule* ulsche;
  DCICollection* coll;
  srsran_dl_sf_cfg_t sf;
  srsran_dci_cfg_t dci_cfg;
  int mcs_tracking_mode, harq_mode;
};

#define DL_WORDS 64
#define UL_WORDS 32

static void put_mask(uint32_t* o, const bool* prb, uint32_t n)
{
  o[0] = o[1] = o[2] = o[3] = 0;
  for (uint32_t i = 0; i < n && i < 128; i++)
    if (prb[i]) o[i >> 5] |= 1u << (i & 31);
}
static void put_dl_grant(uint32_t* o, const srsran_pdsch_grant_t* g, uint32_t nof_prb)
{
  o[0] = g->nof_prb; o[1] = g->nof_re; o[2] = g->nof_tb;
  put_mask(o + 3, g->prb_idx[0], nof_prb);
  put_mask(o + 7, g->prb_idx[1], nof_prb);
  for (int i = 0; i < 2; i++) {
    uint32_t* t = o + 11 + 7 * i;
    t[0] = g->tb[i].enabled; t[1] = g->tb[i].enabled ? mod_bits(g->tb[i].mod) : 0; t[2] = (uint32_t)g->tb[i].tbs; t[3] = g->tb[i].nof_bits;
    t[4] = (uint32_t)g->tb[i].rv; t[5] = g->tb[i].mcs_idx; t[6] = g->tb[i].cw_idx;
  }
}
static void put_ul_grant(uint32_t* o, const srsran_pusch_grant_t* g)
{
  o[0] = g->L_prb; o[1] = g->n_prb[0]; o[2] = g->n_prb[1]; o[3] = g->freq_hopping; o[4] = g->L_prb ? mod_bits(g->tb.mod) : 0; o[5] = (uint32_t)g->tb.tbs;
  o[6] = (uint32_t)g->tb.rv; o[7] = g->tb.mcs_idx; o[8] = g->nof_re;
}

extern "C" {

void ref_collect_bind(void* unpack_dl, void* unpack_ul, void* tbs_fn)
{
  g_unpack_dl = (unpack_dl_fn_t)unpack_dl; g_unpack_ul = (unpack_ul_fn_t)unpack_ul; g_tbs = (tbs_fn_t)tbs_fn;
}
void ref_collect_set_now_ms(uint64_t ms) { g_now = (clock_t)(ms * (CLOCKS_PER_SEC / 1000)); }

ref_collect_t* ref_collect_new(uint32_t nof_prb, uint32_t nof_ports, uint32_t cell_id, uint32_t cp, int mcs_tracking_mode, int harq_mode, int sniffer_mode)
{
  ref_collect_t* h = new ref_collect_t();
  memset(&h->cell, 0, sizeof(h->cell));
  h->cell.nof_prb = nof_prb; h->cell.nof_ports = nof_ports; h->cell.id = cell_id; h->cell.cp = (srsran_cp_t)cp;
  char cwd[4096];
  char* here = getcwd(cwd, sizeof(cwd));
  if (chdir("/tmp") != 0) here = nullptr; /* MCSTracking's constructor opens mcs_statistic.csv in the working directory */
  h->mcs = new MCSTracking(mcs_tracking_mode, 0, false, sniffer_mode, -1, h->cfo);
  if (here && chdir(here) != 0) { }
  h->harq = new HARQ();
  h->harq->init_HARQ(harq_mode);
  h->ulsche = new ULSchedule(0, nullptr, false);
  h->coll = nullptr;
  memset(&h->sf, 0, sizeof(h->sf));
  memset(&h->dci_cfg, 0, sizeof(h->dci_cfg));
  h->mcs_tracking_mode = mcs_tracking_mode; h->harq_mode = harq_mode;
  return h;
}
void ref_collect_free(ref_collect_t* h)
{
  if (!h) return;
  delete h->coll; delete h->ulsche; delete h->harq; delete h->mcs; delete h;
}
/* SIB2 known: ULSchedule::set_SIB2 + set_config (SubframeWorker.cc:249-253); DCICollection reads the hopping offset from it */
void ref_collect_set_sib2(ref_collect_t* h, uint32_t pusch_hop_offset, uint32_t n_sb)
{
  asn1::rrc::sib_type2_s s;
  s.rr_cfg_common.pusch_cfg_common.pusch_cfg_basic.pusch_hop_offset = pusch_hop_offset;
  s.rr_cfg_common.pusch_cfg_common.pusch_cfg_basic.n_sb = n_sb;
  h->ulsche->set_SIB2(&s);
  h->ulsche->set_config();
}
/* what the decoders feed back between subframes */
void ref_collect_mcs_update(ref_collect_t* h, uint16_t rnti, int table) { h->mcs->update_RNTI_dl(rnti, (dl_sniffer_mcs_table_t)table); }
void ref_collect_harq_update(ref_collect_t* h, uint16_t rnti, int pid, int tid, uint32_t sfn, uint32_t sf_idx, int decoded, int ndi, int rv, int tbs)
{
  dl_sniffer_harq_grant_t g = {};
  g.last_decoded = false; g.ndi = ndi != 0; g.ndi_present = true; g.rv = rv; g.tbs = tbs; g.is_first_transmission = false;
  const int verdict = h->harq->is_retransmission(rnti, pid, tid, g, sfn, sf_idx); /* DL_Sniffer_PDSCH.cc:954: finds or takes an entity, locks the block */
  g.last_decoded = decoded != 0;
  if (verdict == DL_SNIFFER_NEW_TX || verdict == DL_SNIFFER_RE_TX) h->harq->updateHARQRNTI(rnti, pid, tid, sfn, sf_idx, g); /* :1008-1014 */
}

/* one subframe: SubframeInfo's DCICollection + setSubframe (DCISearch.cc:575) */
void ref_collect_begin(ref_collect_t* h, uint32_t sfn, uint32_t sf_idx, uint32_t cfi)
{
  delete h->coll;
  h->coll = new DCICollection(h->cell, h->mcs_tracking_mode, h->mcs, h->harq_mode, h->harq, h->ulsche);
  h->coll->setSubframe(sfn, sf_idx, cfi);
  memset(&h->sf, 0, sizeof(h->sf));
  h->sf.tti = sfn * 10 + sf_idx; h->sf.cfi = cfi;
}
void ref_collect_add(ref_collect_t* h, uint16_t rnti, int format, uint32_t L, uint32_t ncce, uint32_t histval, const uint8_t* payload, uint32_t nof_bits)
{
  dci_candidate_t cand;
  memset(&cand, 0, sizeof(cand));
  cand.rnti = rnti;
  cand.dci_msg.format = (srsran_dci_format_t)format; cand.dci_msg.nof_bits = nof_bits; cand.dci_msg.rnti = rnti;
  cand.dci_msg.location.L = L; cand.dci_msg.location.ncce = ncce;
  memcpy(cand.dci_msg.payload, payload, nof_bits);
  srsran_dci_location_t loc; loc.L = L; loc.ncce = ncce;
  h->coll->addCandidate(cand, loc, histval, &h->sf, &h->dci_cfg);
}
/* the lists the decoders walk.  dl: DL_WORDS per entry, ul: UL_WORDS per entry, maps: nof_prb RNTIs each.  flags: bit 0 DL collision, bit 1 UL collision.
 * A grant the reference did not compute for the entry's table (falcon_dci.c:284-310) is uninitialised memory there and written as zeros here. */
uint32_t ref_collect_end(ref_collect_t* h, uint32_t* dl, uint32_t dl_cap, uint32_t* ul, uint32_t ul_cap, uint16_t* map_dl, uint16_t* map_ul, uint32_t* counts2)
{
  std::vector<DL_Sniffer_DCI_DL>& d = h->coll->getDLSnifferDCI_DL();
  std::vector<DCI_UL>& u = h->coll->getULSnifferDCI_UL();
  const uint32_t n = h->cell.nof_prb;
  for (uint32_t i = 0; i < d.size() && i < dl_cap; i++) {
    uint32_t* o = dl + (size_t)i * DL_WORDS;
    memset(o, 0, sizeof(uint32_t) * DL_WORDS);
    const DL_Sniffer_DCI_DL& e = d[i];
    const srsran_dci_dl_t* r = e.ran_dci_dl.get();
    o[0] = e.rnti; o[1] = (uint32_t)e.format; o[2] = (uint32_t)e.mcs_table; o[3] = r->rnti; o[4] = r->pid; o[5] = r->pinfo; o[6] = r->tb_cw_swap;
    for (int t = 0; t < 2; t++) { o[7 + 3 * t] = r->tb[t].mcs_idx; o[8 + 3 * t] = (uint32_t)r->tb[t].rv; o[9 + 3 * t] = r->tb[t].ndi; }
    o[13] = e.check ? 1 : 0;
    const bool has64 = e.mcs_table != DL_SNIFFER_256QAM_TABLE, has256 = e.mcs_table != DL_SNIFFER_64QAM_TABLE;
    if (has64) put_dl_grant(o + 14, e.ran_pdsch_grant.get(), n);
    if (has256) put_dl_grant(o + 39, e.ran_pdsch_grant_256.get(), n);
  }
  for (uint32_t i = 0; i < u.size() && i < ul_cap; i++) {
    uint32_t* o = ul + (size_t)i * UL_WORDS;
    memset(o, 0, sizeof(uint32_t) * UL_WORDS);
    const DCI_UL& e = u[i];
    const srsran_dci_ul_t* r = e.ran_ul_dci.get();
    o[0] = e.rnti; o[1] = r->rnti; o[2] = r->n_dmrs; o[3] = r->cqi_request; o[4] = r->tb.ndi; o[5] = r->tpc_pusch; o[6] = (uint32_t)r->freq_hop_fl; o[7] = r->type2_alloc.riv;
    o[8] = r->tb.mcs_idx; o[9] = (uint32_t)r->tb.rv;
    put_ul_grant(o + 10, e.ran_ul_grant.get());
    put_ul_grant(o + 19, e.ran_ul_grant_256.get());
    o[28] = e.ul_grant->L_prb; o[29] = e.ul_grant->n_prb[0];
  }
  for (uint32_t i = 0; i < n; i++) { map_dl[i] = h->coll->getRBMapDL()[i]; map_ul[i] = h->coll->getRBMapUL()[i]; }
  counts2[0] = (uint32_t)d.size(); counts2[1] = (uint32_t)u.size();
  return (h->coll->hasCollisionDL() ? 1u : 0u) | (h->coll->hasCollisionUL() ? 2u : 0u);
}

/* ---- the RAR grant -> DCI 0 -> PUSCH grant chain of the uplink mode (falcon_dci.c:636-683, DL_Sniffer_PDSCH.cc:632-671) ---- */
int ref_collect_rar_grant(uint32_t nof_prb, uint32_t cp, uint32_t n_rb_ho, const uint8_t* grant20, int32_t* out /* 6 rar fields + 9 grant words */)
{
  srsran_cell_t cell; memset(&cell, 0, sizeof(cell)); cell.nof_prb = nof_prb; cell.cp = (srsran_cp_t)cp;
  uint8_t bits[SRSRAN_RAR_GRANT_LEN];
  memcpy(bits, grant20, SRSRAN_RAR_GRANT_LEN);
  srsran_dci_rar_grant_t rar; memset(&rar, 0, sizeof(rar));
  ul_sniffer_dci_rar_unpack(bits, &rar);
  out[0] = rar.hopping_flag; out[1] = (int32_t)rar.rba; out[2] = (int32_t)rar.trunc_mcs; out[3] = rar.tpc_pusch; out[4] = rar.ul_delay; out[5] = rar.cqi_request;
  srsran_dci_ul_t dci;
  ul_sniffer_dci_rar_to_ul_dci(&cell, &rar, &dci);
  srsran_ul_sf_cfg_t sf; memset(&sf, 0, sizeof(sf));
  srsran_pusch_hopping_cfg_t hop; memset(&hop, 0, sizeof(hop)); hop.n_rb_ho = n_rb_ho;
  srsran_pusch_grant_t g; memset(&g, 0, sizeof(g));
  int rc = srsran_ra_ul_dci_to_grant(&cell, &sf, &hop, &dci, &g);
  uint32_t w[9];
  put_ul_grant(w, &g);
  for (int i = 0; i < 9; i++) out[6 + i] = (int32_t)w[i];
  out[15] = dci.freq_hop_fl; out[16] = (int32_t)dci.type2_alloc.riv; out[17] = (int32_t)dci.tb.mcs_idx; out[18] = dci.tb.rv;
  return rc;
}

/* ---- ULSchedule (ULSchedule.cc:11-138): script interface.  A pushed list is n entries whose RNTIs are given; a get returns the RNTIs found ---- */
void ref_ulsche_push(ref_collect_t* h, uint32_t tti, const uint16_t* rntis, uint32_t n, int rar)
{
  std::vector<DCI_UL> v;
  for (uint32_t i = 0; i < n; i++) { DCI_UL d; d.rnti = rntis[i]; v.push_back(d); }
  if (rar) h->ulsche->push_rar_ULSche(tti, v); else h->ulsche->pushULSche(tti, v);
}
int ref_ulsche_get(ref_collect_t* h, uint32_t tti, uint16_t* rntis, uint32_t cap, int rar)
{
  std::vector<DCI_UL>* v = rar ? h->ulsche->get_rar_ULSche(tti) : h->ulsche->getULSche(tti);
  if (!v) return -1;
  for (uint32_t i = 0; i < v->size() && i < cap; i++) rntis[i] = (*v)[i].rnti;
  return (int)v->size();
}
void ref_ulsche_delete(ref_collect_t* h, uint32_t tti, int rar) { if (rar) h->ulsche->delete_rar_ULSche(tti); else h->ulsche->deleteULSche(tti); }
int ref_ulsche_ul_tti(ref_collect_t* h, uint32_t tti, int rar) { return rar ? h->ulsche->get_rar_ul_tti(tti) : h->ulsche->get_ul_tti(tti); }

} /* extern "C" */
