/* o_dci.c - ORACLE (test infrastructure only): DCI format sizes and unpacking (TS 36.212 5.3.3.1), resource
 * allocation types 0/1/2 incl. distributed VRBs (TS 36.213 7.1.6, TS 36.211 6.2.3.2), MCS -> modulation/TBS
 * (TS 36.213 7.1.7) and MIMO configuration.
 * Reference, in tree: dl_sniffer_compute_tb / dl_sniffer_ra_dl_dci_to_grant / dl_sniffer_config_mimo*
 * (/root/reference/lib/src/phy/falcon_phch/dl_sniffer_pdsch.c:14-276), srsran_dci_msg_to_trace_timestamp
 * (/root/reference/lib/src/phy/falcon_phch/falcon_dci.c:148-352).
 * Reference, out of tree [srsRAN]: srsran_dci_format_sizeof (falcon_pdcch.c:133), srsran_dci_msg_unpack_pdsch/
 * _pusch (falcon_dci.c:271,208), srsran_ra_dl_grant_to_grant_prb_allocation, srsran_dl_fill_ra_mcs,
 * srsran_ra_dl_compute_nof_re, srsran_ra_tbs_from_idx (dl_sniffer_pdsch.c:104,79,110,47). */
#include "lsn_oracle.h"
#include "../spec/lte_tables.h"
#include <string.h>

static uint32_t log2ceil(uint32_t x)
{
  uint32_t n = 0;
  while ((1u << n) < x) n++;
  return n;
}
static uint32_t riv_nbits(uint32_t nprb) { return log2ceil(nprb * (nprb + 1) / 2); }
static uint32_t ra_type0_P(uint32_t nprb) { return nprb <= 10 ? 1 : nprb <= 26 ? 2 : nprb <= 63 ? 3 : 4; }
static int is_ambiguous(uint32_t n)
{
  static const uint32_t a[10] = {12, 14, 16, 20, 24, 26, 32, 40, 44, 56};
  for (int i = 0; i < 10; i++)
    if (a[i] == n) return 1;
  return 0;
}
static uint32_t n_gap(uint32_t nprb, int gap2)
{
  if (!gap2) {
    if (nprb <= 10) return (nprb + 1) / 2;
    if (nprb == 11) return 4;
    if (nprb <= 19) return 8;
    if (nprb <= 26) return 12;
    if (nprb <= 44) return 18;
    if (nprb <= 63) return 27;
    if (nprb <= 79) return 32;
    return 48;
  }
  if (nprb < 50) return 0;
  return nprb <= 63 ? 9 : 16;
}
static uint32_t n_vrb_dl(uint32_t nprb, int gap2)
{
  uint32_t g = n_gap(nprb, gap2);
  if (!gap2) return 2 * (g < nprb - g ? g : nprb - g);
  return g ? (nprb / (2 * g)) * 2 * g : 0;
}
static uint32_t n_step(uint32_t nprb) { return nprb < 50 ? 2 : 4; }
static uint32_t pbits_f2(uint32_t ports) { return ports == 2 ? 3 : ports == 4 ? 6 : 0; }
static uint32_t pbits_f2a(uint32_t ports) { return ports == 4 ? 2 : 0; }
static uint32_t tpmi_bits(uint32_t ports) { return ports == 4 ? 4 : 2; }

static uint32_t f0_raw(uint32_t n) { return 1 + 1 + riv_nbits(n) + 5 + 1 + 2 + 3 + 1; }
static uint32_t f1a_size(uint32_t n)
{
  uint32_t s = 1 + 1 + riv_nbits(n) + 5 + 3 + 1 + 2 + 2;
  while (s < f0_raw(n)) s++;
  if (is_ambiguous(s)) s++;
  return s;
}
static uint32_t f0_size(uint32_t n)
{
  uint32_t s = f0_raw(n);
  while (s < f1a_size(n)) s++;
  return s;
}
static uint32_t alloc_bits(uint32_t n) { return (n > 10 ? 1u : 0u) + (n + ra_type0_P(n) - 1) / ra_type0_P(n); }

uint32_t o_dci_format_sizeof(const o_cell_t* cell, int format)
{
  uint32_t n = cell->nof_prb, s;
  switch (format) {
    case O_FMT0: return f0_size(n);
    case O_FMT1A: return f1a_size(n);
    case O_FMT1:
      s = alloc_bits(n) + 5 + 3 + 1 + 2 + 2;
      while (s == f0_size(n) || s == f1a_size(n) || is_ambiguous(s)) s++;
      return s;
    case O_FMT1B:
      s = 1 + riv_nbits(n) + 5 + 3 + 1 + 2 + 2 + tpmi_bits(cell->nof_ports) + 1;
      while (is_ambiguous(s)) s++;
      return s;
    case O_FMT1C: {
      uint32_t q = n_vrb_dl(n, 0) / n_step(n);
      return (n < 50 ? 0u : 1u) + log2ceil(q * (q + 1) / 2) + 5;
    }
    case O_FMT1D:
      s = 1 + riv_nbits(n) + 5 + 3 + 1 + 2 + 2 + tpmi_bits(cell->nof_ports) + 1;
      while (is_ambiguous(s)) s++;
      return s;
    case O_FMT2:
      s = alloc_bits(n) + 2 + 3 + 1 + 16 + pbits_f2(cell->nof_ports);
      while (is_ambiguous(s)) s++;
      return s;
    case O_FMT2A:
      s = alloc_bits(n) + 2 + 3 + 1 + 16 + pbits_f2a(cell->nof_ports);
      while (is_ambiguous(s)) s++;
      return s;
    case O_FMT2B:
      s = alloc_bits(n) + 2 + 3 + 1 + 16;
      while (is_ambiguous(s)) s++;
      return s;
    default: return 0;
  }
}

static uint32_t take(const uint8_t** y, uint32_t n)
{
  uint32_t v = 0;
  for (uint32_t i = 0; i < n; i++) v = (v << 1) | (*(*y)++ & 1u);
  return v;
}

static void unpack_alloc01(const o_cell_t* cell, const uint8_t** y, o_dci_dl_t* d)
{
  uint32_t n = cell->nof_prb, P = ra_type0_P(n), asz = (n + P - 1) / P;
  d->alloc_type = 0;
  if (n > 10) d->alloc_type = (int)take(y, 1);
  if (d->alloc_type == 0) {
    d->rbg_bitmask = take(y, asz);
  } else {
    uint32_t lb = log2ceil(P);
    d->t1_rbg_subset = take(y, lb);
    d->t1_shift = take(y, 1);
    d->t1_vrb_bitmask = take(y, asz - lb - 1);
  }
}

int o_dci_unpack_dl(const o_cell_t* cell, const uint8_t* payload, uint32_t nof_bits, int format, uint16_t rnti, o_dci_dl_t* d)
{
  const uint8_t* y = payload;
  uint32_t n = cell->nof_prb;
  uint16_t L = (uint16_t)d->L, nc = (uint16_t)d->ncce;
  memset(d, 0, sizeof(*d));
  d->L = L;
  d->ncce = nc;
  d->rnti = rnti;
  d->format = format;
  if (nof_bits != o_dci_format_sizeof(cell, format)) return -1;
  switch (format) {
    case O_FMT1:
      unpack_alloc01(cell, &y, d);
      d->tb[0].mcs_idx = take(&y, 5);
      d->pid = take(&y, 3);
      d->tb[0].ndi = take(&y, 1);
      d->tb[0].rv = (int)take(&y, 2);
      d->tpc = take(&y, 2);
      break;
    case O_FMT1A:
    case O_FMT1B:
    case O_FMT1D: {
      if (format == O_FMT1A && take(&y, 1) != 1) return -1;
      d->alloc_type = 2;
      d->t2_dist = (int)take(&y, 1);
      uint32_t nb_gap = 0;
      if (O_RNTI_ISUSER(rnti) && d->t2_dist && n >= 50) {
        d->t2_ngap2 = (int)take(&y, 1);
        nb_gap = 1;
      }
      d->riv = take(&y, riv_nbits(n) - nb_gap);
      d->tb[0].mcs_idx = take(&y, 5);
      d->pid = take(&y, 3);
      if (!O_RNTI_ISUSER(rnti) && format == O_FMT1A) {
        uint32_t b = take(&y, 1);
        if (n >= 50 && d->t2_dist) d->t2_ngap2 = (int)b;
      } else {
        d->tb[0].ndi = take(&y, 1);
      }
      d->tb[0].rv = (int)take(&y, 2);
      if (O_RNTI_ISUSER(rnti) || format != O_FMT1A) {
        d->tpc = take(&y, 2);
      } else {
        (void)take(&y, 1);
        d->t2_nprb1a_is2 = take(&y, 1) ? 0 : 1; /* LSB of TPC: 0 -> N_PRB^1A = 2, 1 -> 3 (36.213 7.1.7) */
      }
      if (format != O_FMT1A) d->pinfo = take(&y, tpmi_bits(cell->nof_ports));
      break;
    }
    case O_FMT1C: {
      if (n >= 50) d->t2_ngap2 = (int)take(&y, 1);
      uint32_t q = n_vrb_dl(n, d->t2_ngap2) / n_step(n);
      d->alloc_type = 2;
      d->t2_dist = 1;
      d->riv = take(&y, log2ceil(q * (q + 1) / 2));
      d->tb[0].mcs_idx = take(&y, 5);
      d->tb[0].rv = -1; /* set later (DL_Sniffer_PDSCH.cc:891-897) */
      break;
    }
    case O_FMT2:
    case O_FMT2A:
    case O_FMT2B:
      unpack_alloc01(cell, &y, d);
      d->tpc = take(&y, 2);
      d->pid = take(&y, 3);
      d->tb_cw_swap = take(&y, 1); /* 2B: scrambling identity */
      for (int i = 0; i < 2; i++) {
        d->tb[i].mcs_idx = take(&y, 5);
        d->tb[i].ndi = take(&y, 1);
        d->tb[i].rv = (int)take(&y, 2);
      }
      if (format == O_FMT2) d->pinfo = take(&y, pbits_f2(cell->nof_ports));
      if (format == O_FMT2A) d->pinfo = take(&y, pbits_f2a(cell->nof_ports));
      {
        int en0 = !(d->tb[0].mcs_idx == 0 && d->tb[0].rv == 1), en1 = !(d->tb[1].mcs_idx == 0 && d->tb[1].rv == 1);
        if (en0 && en1) {
          d->tb[0].cw_idx = (format != O_FMT2B && d->tb_cw_swap) ? 1 : 0;
          d->tb[1].cw_idx = (format != O_FMT2B && d->tb_cw_swap) ? 0 : 1;
        } else {
          d->tb[0].cw_idx = 0;
          d->tb[1].cw_idx = 0;
        }
      }
      break;
    default: return -1;
  }
  return 0;
}

int o_dci_unpack_ul(const o_cell_t* cell, const uint8_t* payload, uint32_t nof_bits, uint16_t rnti, o_dci_ul_t* d)
{
  const uint8_t* y = payload;
  uint32_t L = d->L, nc = d->ncce;
  memset(d, 0, sizeof(*d));
  d->L = L;
  d->ncce = nc;
  d->rnti = rnti;
  if (nof_bits != f0_size(cell->nof_prb)) return -1;
  if (take(&y, 1) != 0) return -1;
  d->freq_hop_fl = take(&y, 1);
  d->hop_type = -1;
  if (d->freq_hop_fl) { /* 36.213 8.4: the N_UL_hop most significant bits of the allocation field are the hopping bits (Table 8.4-1 / 8.4-2) */
    if (cell->nof_prb < 50) {
      d->hop_type = take(&y, 1) ? 3 : 2;
      d->riv = take(&y, riv_nbits(cell->nof_prb) - 1);
    } else {
      d->hop_type = (int)take(&y, 2);
      d->riv = take(&y, riv_nbits(cell->nof_prb) - 2);
    }
  } else {
    d->riv = take(&y, riv_nbits(cell->nof_prb));
  }
  d->mcs_idx = take(&y, 5);
  d->ndi = take(&y, 1);
  d->tpc = take(&y, 2);
  d->n_dmrs = take(&y, 3);
  d->cqi_req = take(&y, 1);
  return 0;
}

static void type2_from_riv(uint32_t riv, uint32_t* L, uint32_t* start, uint32_t nprb, uint32_t nvrb)
{
  *L = riv / nprb + 1;
  *start = riv % nprb;
  if (*L > nvrb - *start) {
    *L = nprb - *L + 2;
    *start = nprb - 1 - *start;
  }
}

/* distributed VRB -> PRB for both slots (36.211 6.2.3.2) */
static void vrb_dist_to_prb(uint32_t nprb, int gap2, uint32_t nvrb_idx, uint32_t* prb0, uint32_t* prb1)
{
  uint32_t P = ra_type0_P(nprb), G = n_gap(nprb, gap2);
  uint32_t Nvrb = n_vrb_dl(nprb, gap2);
  uint32_t Nt = gap2 ? 2 * G : Nvrb; /* N~_VRB */
  uint32_t Nrow = ((Nt + 4 * P - 1) / (4 * P)) * P;
  uint32_t Nnull = 4 * Nrow - Nt;
  uint32_t nt = nvrb_idx % Nt, blk = nvrb_idx / Nt;
  uint32_t p1 = 2 * Nrow * (nt % 2) + nt / 2 + Nt * blk;
  uint32_t p2 = Nrow * (nt % 4) + nt / 4 + Nt * blk;
  uint32_t e;
  if (Nnull != 0 && nt >= Nt - Nnull && (nt % 2) == 1)
    e = p1 - Nrow;
  else if (Nnull != 0 && nt >= Nt - Nnull && (nt % 2) == 0)
    e = p1 - Nrow + Nnull / 2;
  else if (Nnull != 0 && nt < Nt - Nnull && (nt % 4) >= 2)
    e = p2 - Nnull / 2;
  else
    e = p2;
  uint32_t o = (e % Nt + Nt / 2) % Nt + Nt * blk; /* odd slot */
  uint32_t eb = e % Nt, ob = o % Nt;
  *prb0 = (eb < Nt / 2 ? eb : eb + G - Nt / 2) + Nt * blk;
  *prb1 = (ob < Nt / 2 ? ob : ob + G - Nt / 2) + Nt * blk;
  (void)Nvrb;
}

int o_tbs_from_idx(int i_tbs, uint32_t n_prb)
{
  if (i_tbs < 0 || i_tbs >= LSN_TBS_NROWS || n_prb < 1 || n_prb > 110) return -1;
  return lsn_tbs_table[i_tbs][n_prb - 1];
}

int o_pdsch_re_ok(const o_cell_t* cell, uint32_t sf_idx, uint32_t l, uint32_t k)
{
  uint32_t nprb = cell->nof_prb;
  const uint32_t nsl = (uint32_t)o_nslot(cell), lq = l % nsl;
  if (cell->nof_ports == 4 && lq == 1) { /* CRS of ports 2, 3 */
    if ((k % 3) == (cell->id % 3)) return 0;
  }
  if (o_is_crs_sym01(cell, (int)l)) {
    if (cell->nof_ports >= 2) {
      if ((k % 3) == (cell->id % 3)) return 0;
    } else {
      uint32_t v = lq == 0 ? 0u : 3u;
      if ((k % 6) == (v + cell->id % 6) % 6) return 0;
    }
  }
  uint32_t kc0 = 6 * nprb - 36;
  if (k >= kc0 && k < kc0 + 72) {
    if ((sf_idx == 0 || sf_idx == 5) && (l == nsl - 2 || l == nsl - 1)) return 0; /* SSS, PSS: the last two symbols of slots 0 and 10 */
    if (sf_idx == 0 && l >= nsl && l <= nsl + 3) return 0;                          /* PBCH: symbols 0-3 of slot 1 */
  }
  return 1;
}

uint32_t o_ra_nof_re(const o_cell_t* cell, uint32_t sf_idx, uint32_t cfi, const o_pdsch_grant_t* g)
{
  uint32_t n = 0, l0 = cfi + (cell->nof_prb <= 10 ? 1u : 0u);
  for (uint32_t l = l0; l < (uint32_t)o_nsym(cell); l++)
    for (uint32_t prb = 0; prb < cell->nof_prb; prb++)
      if (g->prb_idx[l / (uint32_t)o_nslot(cell)][prb])
        for (uint32_t k = 12 * prb; k < 12 * prb + 12; k++) n += (uint32_t)o_pdsch_re_ok(cell, sf_idx, l, k);
  return n;
}

static int prb_allocation(const o_cell_t* cell, const o_dci_dl_t* d, o_pdsch_grant_t* g)
{
  uint32_t n = cell->nof_prb, P = ra_type0_P(n);
  memset(g->prb_idx, 0, sizeof(g->prb_idx));
  g->nof_prb = 0;
  switch (d->alloc_type) {
    case 0: {
      uint32_t nb = (n + P - 1) / P;
      for (uint32_t i = 0; i < nb; i++)
        if (d->rbg_bitmask & (1u << (nb - i - 1)))
          for (uint32_t j = 0; j < P && i * P + j < n; j++) {
            g->prb_idx[0][i * P + j] = 1;
            g->nof_prb++;
          }
      memcpy(g->prb_idx[1], g->prb_idx[0], sizeof(g->prb_idx[0]));
      break;
    }
    case 1: {
      uint32_t lb = log2ceil(P), n1 = (n + P - 1) / P - lb - 1;
      uint32_t p = d->t1_rbg_subset, sub;
      if (p < ((n - 1) / P) % P)
        sub = ((n - 1) / (P * P)) * P + P;
      else if (p == ((n - 1) / P) % P)
        sub = ((n - 1) / (P * P)) * P + ((n - 1) % P) + 1;
      else
        sub = ((n - 1) / (P * P)) * P;
      uint32_t shift = d->t1_shift ? sub - n1 : 0;
      for (uint32_t i = 0; i < n1; i++)
        if (d->t1_vrb_bitmask & (1u << (n1 - i - 1))) {
          uint32_t idx = ((i + shift) / P) * P * P + p * P + (i + shift) % P;
          if (idx < n) {
            g->prb_idx[0][idx] = 1;
            g->nof_prb++;
          }
        }
      memcpy(g->prb_idx[1], g->prb_idx[0], sizeof(g->prb_idx[0]));
      break;
    }
    case 2: {
      uint32_t L, start;
      if (d->format == O_FMT1C) {
        uint32_t st = n_step(n), nv = n_vrb_dl(n, d->t2_ngap2) / st;
        if (nv == 0) return -1;
        type2_from_riv(d->riv, &L, &start, nv, nv);
        L *= st;
        start *= st;
      } else {
        type2_from_riv(d->riv, &L, &start, n, n);
      }
      if (!d->t2_dist) {
        for (uint32_t i = 0; i < L; i++)
          if (start + i < n) {
            g->prb_idx[0][start + i] = 1;
            g->nof_prb++;
          }
        memcpy(g->prb_idx[1], g->prb_idx[0], sizeof(g->prb_idx[0]));
      } else {
        uint32_t nv = n_vrb_dl(n, d->t2_ngap2);
        if (nv == 0) return -1;
        for (uint32_t i = 0; i < L; i++) {
          uint32_t v = start + i, p0, p1;
          if (v >= nv) continue;
          vrb_dist_to_prb(n, d->t2_ngap2, v, &p0, &p1);
          if (p0 < n && p1 < n) {
            g->prb_idx[0][p0] = 1;
            g->prb_idx[1][p1] = 1;
            g->nof_prb++;
          }
        }
      }
      break;
    }
    default: return -1;
  }
  return g->nof_prb > 0 ? 0 : -1;
}

/* srsran_dl_fill_ra_mcs [srsRAN] as used at dl_sniffer_pdsch.c:79 */
static int fill_ra_mcs(o_tb_t* tb, int last_tbs, uint32_t nprb, int alt)
{
  const int8_t(*t)[2] = alt ? lsn_mcs_dl_256qam : lsn_mcs_dl_64qam;
  tb->mod = t[tb->mcs_idx & 31][0];
  int i_tbs = t[tb->mcs_idx & 31][1];
  int tbs = 0;
  if (i_tbs >= 0) {
    tbs = o_tbs_from_idx(i_tbs, nprb);
    tb->tbs = tbs;
  } else {
    tb->tbs = last_tbs;
  }
  return tbs;
}

/* dl_sniffer_compute_tb, dl_sniffer_pdsch.c:14-92 */
static int compute_tb(int alt, const o_dci_dl_t* d, o_pdsch_grant_t* g)
{
  for (int i = 0; i < 2; i++) {
    g->tb[i].mcs_idx = d->tb[i].mcs_idx;
    g->tb[i].rv = d->tb[i].rv;
    g->tb[i].cw_idx = d->tb[i].cw_idx;
    int en = !(d->tb[i].mcs_idx == 0 && d->tb[i].rv == 1);
    if ((en && d->format >= O_FMT2) || (d->format < O_FMT2 && i == 0)) {
      g->tb[i].enabled = 1;
      g->nof_tb++;
    } else {
      g->tb[i].enabled = 0;
    }
  }
  if (d->format == O_FMT1A || !O_RNTI_ISUSER(d->rnti)) alt = 0;
  if (!O_RNTI_ISUSER(d->rnti)) {
    int tbs;
    if (d->format == O_FMT1A) {
      uint32_t np = d->t2_nprb1a_is2 ? 2 : 3;
      tbs = o_tbs_from_idx((int)d->tb[0].mcs_idx, np);
      if (tbs < 0) return -1;
    } else if (d->format == O_FMT1C) {
      if (d->tb[0].mcs_idx < 32)
        tbs = lsn_tbs_format1c_table[d->tb[0].mcs_idx];
      else
        return -1;
    } else {
      return -1;
    }
    g->tb[0].mod = O_MOD_QPSK;
    g->tb[0].tbs = tbs;
  } else {
    for (int i = 0; i < 2; i++) {
      if (g->tb[i].enabled) {
        g->tb[i].tbs = fill_ra_mcs(&g->tb[i], 0, g->nof_prb, alt);
        if (g->tb[i].tbs < 0) return -1;
      } else {
        g->tb[i].tbs = 0;
      }
    }
  }
  return 0;
}

/* dl_sniffer_ra_dl_dci_to_grant, dl_sniffer_pdsch.c:95-132 */
int o_ra_dl_dci_to_grant(const o_cell_t* cell, uint32_t sf_idx, uint32_t cfi, int alt, const o_dci_dl_t* d, o_pdsch_grant_t* g)
{
  memset(g, 0, sizeof(*g));
  if (prb_allocation(cell, d, g)) return -1;
  if (compute_tb(alt, d, g)) return -1;
  g->nof_re = o_ra_nof_re(cell, sf_idx, cfi, g);
  for (int i = 0; i < 2; i++) g->tb[i].nof_bits = g->tb[i].enabled ? (int)g->nof_re * g->tb[i].mod : 0;
  if (d->format == O_FMT1C && (O_RNTI_ISRAR(d->rnti) || d->rnti == O_PRNTI))
    for (int i = 0; i < 2; i++) g->tb[i].rv = 0;
  return 0;
}

/* ul_sniffer_ra_ul_grant_to_grant_prb_allocation (ul_sniffer_pusch.c:19-87) + the 64QAM MCS table */
int o_ra_ul_dci_to_grant(const o_cell_t* cell, const o_dci_ul_t* d, o_pusch_grant_t* g)
{
  memset(g, 0, sizeof(*g));
  uint32_t L, start, nprb = cell->nof_prb;
  type2_from_riv(d->riv, &L, &start, nprb, nprb);
  if (L == 0 || start + L > nprb) return -1;
  g->L_prb = L;
  g->n_prb = start;
  g->n_prb2 = start;
  const int hop = d->freq_hop_fl ? d->hop_type : -1;
  if (hop == 3) {
    g->hop = 2; /* type 2: same PRBs in the grant, mirrored / hopped during resource mapping (not decoded here) */
  } else if (hop >= 0) { /* type 1, 36.213 8.4.1: fixed offset between the slots */
    uint32_t n_rb_ho = cell->pusch_hop_offset;
    if (n_rb_ho % 2) n_rb_ho++;
    if (n_rb_ho + (nprb % 2) >= nprb) return -1;
    const uint32_t n_rb_pusch = nprb - n_rb_ho - (nprb % 2);
    if (start < n_rb_ho / 2) return -1;
    if (hop == 0) g->n_prb2 = (n_rb_pusch / 4 + start) % n_rb_pusch;
    else if (hop == 1) g->n_prb2 = start < n_rb_pusch / 4 ? n_rb_pusch + start - n_rb_pusch / 4 : start - n_rb_pusch / 4;
    else g->n_prb2 = (n_rb_pusch / 2 + start) % n_rb_pusch;
    g->hop = 1;
    if (g->n_prb2 + L > nprb) return -1;
  }
  g->mcs_idx = d->mcs_idx;
  int itbs = lsn_mcs_ul_64qam[d->mcs_idx & 31][1];
  g->mod = lsn_mcs_ul_64qam[d->mcs_idx & 31][0];
  if (itbs >= 0) {
    g->tbs = o_tbs_from_idx(itbs, L);
    g->rv = 0;
  } else {
    g->tbs = 0;
    g->rv = (int)d->mcs_idx - 28;
  }
  return 0;
}

/* dl_sniffer_config_mimo, dl_sniffer_pdsch.c:134-276. Return: 0 ok, else error class */
int o_config_mimo(const o_cell_t* cell, int format, const o_dci_dl_t* d, o_pdsch_grant_t* g)
{
  switch (format) {
    case O_FMT1:
    case O_FMT1A:
    case O_FMT1C: g->tx_scheme = cell->nof_ports == 1 ? O_TX_PORT0 : O_TX_DIVERSITY; break;
    case O_FMT2: g->tx_scheme = (g->nof_tb == 1 && d->pinfo == 0) ? O_TX_DIVERSITY : O_TX_SPATIALMUX; break;
    case O_FMT2A: g->tx_scheme = (g->nof_tb == 1 && d->pinfo == 0) ? O_TX_DIVERSITY : O_TX_CDD; break;
    default: return 1; /* DL_SNIFFER_MIMO_NOT_SUPPORT */
  }
  if (g->tx_scheme == O_TX_SPATIALMUX) {
    if (g->nof_tb == 1) {
      if (d->pinfo > 0 && d->pinfo < 5)
        g->pmi = d->pinfo - 1;
      else
        return 2;
    } else {
      if (d->pinfo >= 2) return 2;
      g->pmi = d->pinfo % 2;
    }
  }
  switch (g->tx_scheme) {
    case O_TX_PORT0:
      if (g->nof_tb != 1) return 3;
      g->nof_layers = 1;
      break;
    case O_TX_DIVERSITY:
      if (g->nof_tb != 1) return 3;
      g->nof_layers = cell->nof_ports;
      break;
    case O_TX_SPATIALMUX:
      if (g->nof_tb == 1)
        g->nof_layers = 1;
      else if (g->nof_tb == 2)
        g->nof_layers = 2;
      else
        return 3;
      break;
    case O_TX_CDD:
      if (g->nof_tb != 2) return 3;
      g->nof_layers = 2;
      break;
  }
  return 0;
}

/* ul_fill_ra_mcs_256 (ul_sniffer_pusch.c:91-136, Table 8.6.1-3 of 36.213 incl. the 32A row) on the grant of
 * o_ra_ul_dci_to_grant: same PRB allocation, modulation / TBS of the 256QAM-capable table */
int o_ra_ul_dci_to_grant_256(const o_cell_t* cell, const o_dci_ul_t* d, o_pusch_grant_t* g)
{
  if (o_ra_ul_dci_to_grant(cell, d, g)) return -1;
  uint32_t m = d->mcs_idx, L = g->L_prb;
  if (m <= 28) {
    g->rv = 0;
    if (m < 6) { g->mod = 2; g->tbs = o_tbs_from_idx((int)m * 2, L); }
    else if (m < 14) { g->mod = 4; g->tbs = o_tbs_from_idx((int)m + (m < 10 ? 5 : 6), L); }
    else if (m < 23) { g->mod = 6; g->tbs = o_tbs_from_idx((int)m + (m < 19 ? 6 : 7), L); }
    else {
      g->mod = 8;
      if (m < 26) g->tbs = o_tbs_from_idx((int)m + 7, L);
      else if (m == 26) g->tbs = (L > 0 && L < 111) ? lsn_tbs_table_32A[L - 1] : 0;
      else g->tbs = o_tbs_from_idx((int)m + 6, L);
    }
  } else {
    g->mod = 0; g->tbs = 0; g->rv = (int)m - 28; /* last_tb is empty without HARQ state */
  }
  return 0;
}
