// lsn_lte.cc - see lsn_lte.h.  Product code: must not include anything from oracle/.
#include <atomic>
#include "lsn_lte.h"
#ifdef __HIPCC__
#include <hip/hip_runtime.h>  // lsn_rm.h marks its helpers __host__ __device__ under hipcc
#endif
#include "../kernels/lsn_rm.h"
#include "../../../spec/lte_tables.h"
#include <algorithm>

namespace lsn {

const DciFormat falcon_ue_all_formats[NOF_FORMATS] = {FORMAT0, FORMAT1, FORMAT1A, FORMAT1B, FORMAT1C, FORMAT1D, FORMAT2, FORMAT2A, FORMAT2B};

// ---------------------------------------------------------------------------------------------- small helpers
static uint32_t ceil_log2(uint32_t x) { uint32_t n = 0; while ((1u << n) < x) n++; return n; }
static uint32_t riv_nbits(uint32_t nprb) { return ceil_log2(nprb * (nprb + 1) / 2); }
static uint32_t ra_type0_P(uint32_t nprb) { return nprb <= 10 ? 1 : (nprb <= 26 ? 2 : (nprb <= 63 ? 3 : 4)); }
static bool is_ambiguous_size(uint32_t n)
{
  for (uint32_t a : {12u, 14u, 16u, 20u, 24u, 26u, 32u, 40u, 44u, 56u}) if (a == n) return true;
  return false;
}
static uint32_t ra_type2_ngap(uint32_t nprb, bool gap2)
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
  return nprb < 50 ? 0 : (nprb <= 63 ? 9 : 16);
}
static uint32_t ra_type2_n_vrb_dl(uint32_t nprb, bool gap2)
{
  uint32_t g = ra_type2_ngap(nprb, gap2);
  if (!gap2) return 2 * std::min(g, nprb - g);
  return g ? (nprb / (2 * g)) * 2 * g : 0;
}
static uint32_t ra_type2_n_rb_step(uint32_t nprb) { return nprb < 50 ? 2 : 4; }

// ---------------------------------------------------------------------------------------------- DCI sizes (36.212 5.3.3.1)
namespace {
struct Sizer {
  uint32_t n, ports;
  uint32_t alloc() const { return (n > 10 ? 1u : 0u) + (n + ra_type0_P(n) - 1) / ra_type0_P(n); }
  uint32_t f0_raw() const { return 2 + riv_nbits(n) + 5 + 1 + 2 + 3 + 1; }
  uint32_t f1a() const { uint32_t s = 2 + riv_nbits(n) + 5 + 3 + 1 + 2 + 2; s = std::max(s, f0_raw()); return is_ambiguous_size(s) ? s + 1 : s; }
  uint32_t f0() const { return std::max(f0_raw(), f1a()); }
  uint32_t f1() const { uint32_t s = alloc() + 13; while (s == f0() || s == f1a() || is_ambiguous_size(s)) s++; return s; }
  uint32_t compact(uint32_t extra) const { uint32_t s = 1 + riv_nbits(n) + 13 + extra; while (is_ambiguous_size(s)) s++; return s; }
  uint32_t f1c() const { uint32_t q = ra_type2_n_vrb_dl(n, false) / ra_type2_n_rb_step(n); return (n < 50 ? 0u : 1u) + ceil_log2(q * (q + 1) / 2) + 5; }
  uint32_t f2x(uint32_t pbits) const { uint32_t s = alloc() + 2 + 3 + 1 + 16 + pbits; while (is_ambiguous_size(s)) s++; return s; }
};
}  // namespace

uint32_t dci_format_sizeof(const Cell& cell, DciFormat f)
{
  if ((unsigned)f < 9u && cell.fmt_size[f]) return cell.fmt_size[f];
  Sizer z{cell.nof_prb, cell.nof_ports};
  const uint32_t tpmi = cell.nof_ports == 4 ? 4 : 2;
  switch (f) {
    case FORMAT0: return z.f0();
    case FORMAT1: return z.f1();
    case FORMAT1A: return z.f1a();
    case FORMAT1B: return z.compact(tpmi + 1);
    case FORMAT1C: return z.f1c();
    case FORMAT1D: return z.compact(tpmi + 1);
    case FORMAT2: return z.f2x(cell.nof_ports == 2 ? 3 : (cell.nof_ports == 4 ? 6 : 0));
    case FORMAT2A: return z.f2x(cell.nof_ports == 4 ? 2 : 0);
    case FORMAT2B: return z.f2x(0);
    default: return 0;
  }
}

// ---------------------------------------------------------------------------------------------- DCI unpack
namespace {
struct BitReader {
  const uint8_t* p;
  uint32_t get(uint32_t n) { uint32_t v = 0; while (n--) v = (v << 1) | (*p++ & 1u); return v; }
};
void read_type01(const Cell& cell, BitReader& br, DciDl& d)
{
  const uint32_t n = cell.nof_prb, P = ra_type0_P(n), nbits = (n + P - 1) / P;
  d.alloc_type = n > 10 ? (int)br.get(1) : 0;
  if (d.alloc_type == 0) {
    d.rbg_bitmask = br.get(nbits);
  } else {
    const uint32_t lb = ceil_log2(P);
    d.rbg_subset = br.get(lb);
    d.shift = br.get(1);
    d.vrb_bitmask = br.get(nbits - lb - 1);
  }
}
}  // namespace

bool dci_msg_unpack_pdsch(const Cell& cell, const uint8_t* bits, uint32_t nof_bits, DciFormat f, uint16_t rnti, DciDl& d)
{
  const uint32_t keepL = d.L, keepN = d.ncce;
  d = DciDl();
  d.L = keepL; d.ncce = keepN; d.rnti = rnti; d.format = f;
  if (nof_bits != dci_format_sizeof(cell, f)) return false;
  BitReader br{bits};
  const uint32_t n = cell.nof_prb;
  const bool user = rnti_isuser(rnti);
  switch (f) {
    case FORMAT1:
      read_type01(cell, br, d);
      d.tb[0].mcs_idx = br.get(5); d.pid = br.get(3); d.tb[0].ndi = br.get(1); d.tb[0].rv = (int)br.get(2); d.tpc = br.get(2);
      return true;
    case FORMAT1A: case FORMAT1B: case FORMAT1D: {
      if (f == FORMAT1A && br.get(1) != 1) return false;
      d.alloc_type = 2;
      d.distributed = br.get(1) != 0;
      uint32_t gapbit = 0;
      if (user && d.distributed && n >= 50) { d.ngap2 = br.get(1) != 0; gapbit = 1; }
      d.riv = br.get(riv_nbits(n) - gapbit);
      d.tb[0].mcs_idx = br.get(5);
      d.pid = br.get(3);
      if (!user && f == FORMAT1A) { uint32_t b = br.get(1); if (n >= 50 && d.distributed) d.ngap2 = b != 0; }
      else d.tb[0].ndi = br.get(1);
      d.tb[0].rv = (int)br.get(2);
      if (user || f != FORMAT1A) d.tpc = br.get(2);
      else { br.get(1); d.nprb1a_is2 = br.get(1) == 0; }
      if (f != FORMAT1A) d.pinfo = br.get(cell.nof_ports == 4 ? 4 : 2);
      return true;
    }
    case FORMAT1C: {
      if (n >= 50) d.ngap2 = br.get(1) != 0;
      const uint32_t q = ra_type2_n_vrb_dl(n, d.ngap2) / ra_type2_n_rb_step(n);
      d.alloc_type = 2; d.distributed = true;
      d.riv = br.get(ceil_log2(q * (q + 1) / 2));
      d.tb[0].mcs_idx = br.get(5);
      d.tb[0].rv = -1;  // resolved by the caller (DL_Sniffer_PDSCH.cc:891-897, dl_sniffer_pdsch.c:112-119)
      return true;
    }
    case FORMAT2: case FORMAT2A: case FORMAT2B: {
      read_type01(cell, br, d);
      d.tpc = br.get(2); d.pid = br.get(3); d.tb_cw_swap = br.get(1);
      for (auto& tb : d.tb) { tb.mcs_idx = br.get(5); tb.ndi = br.get(1); tb.rv = (int)br.get(2); }
      if (f == FORMAT2) d.pinfo = br.get(cell.nof_ports == 2 ? 3 : (cell.nof_ports == 4 ? 6 : 0));
      if (f == FORMAT2A) d.pinfo = br.get(cell.nof_ports == 4 ? 2 : 0);
      const bool en0 = !(d.tb[0].mcs_idx == 0 && d.tb[0].rv == 1), en1 = !(d.tb[1].mcs_idx == 0 && d.tb[1].rv == 1);
      const bool swap = f != FORMAT2B && d.tb_cw_swap;
      d.tb[0].cw_idx = (en0 && en1 && swap) ? 1 : 0;
      d.tb[1].cw_idx = (en0 && en1 && !swap) ? 1 : 0;
      return true;
    }
    default: return false;
  }
}

bool dci_msg_unpack_pusch(const Cell& cell, const uint8_t* bits, uint32_t nof_bits, uint16_t rnti, DciUl& d)
{
  const uint32_t keepL = d.L, keepN = d.ncce;
  d = DciUl();
  d.L = keepL; d.ncce = keepN; d.rnti = rnti;
  if (nof_bits != dci_format_sizeof(cell, FORMAT0)) return false;
  BitReader br{bits};
  if (br.get(1) != 0) return false;
  d.hopping = br.get(1);
  if (d.hopping) {  // 36.213 8.4: the N_UL_hop most significant bits of the allocation field are the hopping bits (Tables 8.4-1, 8.4-2)
    if (cell.nof_prb < 50) { d.hop_type = br.get(1) ? 3 : 2; d.riv = br.get(riv_nbits(cell.nof_prb) - 1); }
    else { d.hop_type = (int)br.get(2); d.riv = br.get(riv_nbits(cell.nof_prb) - 2); }
  } else {
    d.riv = br.get(riv_nbits(cell.nof_prb));
  }
  d.mcs_idx = br.get(5); d.ndi = br.get(1);
  d.tpc = br.get(2); d.n_dmrs = br.get(3); d.cqi_req = br.get(1);
  return true;
}

// ---------------------------------------------------------------------------------------------- resource allocation
static void ra_type2_from_riv(uint32_t riv, uint32_t& L, uint32_t& start, uint32_t nprb, uint32_t nvrb)
{
  L = riv / nprb + 1;
  start = riv % nprb;
  if (L > nvrb - start) { L = nprb - L + 2; start = nprb - 1 - start; }
}

// 36.211 6.2.3.2 virtual resource blocks of distributed type
static void distributed_vrb_to_prb(uint32_t nprb, bool gap2, uint32_t vrb, uint32_t& prb_even, uint32_t& prb_odd)
{
  const uint32_t P = ra_type0_P(nprb), G = ra_type2_ngap(nprb, gap2);
  const uint32_t Nt = gap2 ? 2 * G : ra_type2_n_vrb_dl(nprb, false);
  const uint32_t Nrow = ((Nt + 4 * P - 1) / (4 * P)) * P, Nnull = 4 * Nrow - Nt;
  const uint32_t nt = vrb % Nt, blk = vrb / Nt;
  const uint32_t p1 = 2 * Nrow * (nt % 2) + nt / 2 + Nt * blk, p2 = Nrow * (nt % 4) + nt / 4 + Nt * blk;
  uint32_t e;
  if (Nnull && nt >= Nt - Nnull) e = (nt % 2) ? p1 - Nrow : p1 - Nrow + Nnull / 2;
  else if (Nnull && (nt % 4) >= 2) e = p2 - Nnull / 2;
  else e = p2;
  const uint32_t eb = e % Nt, ob = (eb + Nt / 2) % Nt;
  prb_even = (eb < Nt / 2 ? eb : eb + G - Nt / 2) + Nt * blk;
  prb_odd = (ob < Nt / 2 ? ob : ob + G - Nt / 2) + Nt * blk;
}

bool tbs_from_derived_rows(int tbs, uint32_t nof_prb)
{
  if (nof_prb < 1 || nof_prb > 110 || tbs <= 0) return false;
  bool derived = false;
  for (int r = 27; r < 34; r++) derived = derived || lsn_tbs_table[r][nof_prb - 1] == tbs;
  if (!derived) return false;
  for (int r = 0; r < 27; r++) if (lsn_tbs_table[r][nof_prb - 1] == tbs) return false;
  return true;
}

bool ra_dl_grant_to_grant_prb_allocation(const Cell& cell, const DciDl& d, PdschGrant& g)
{
  const uint32_t n = cell.nof_prb, P = ra_type0_P(n);
  auto mark = [&](uint32_t prb) { if (prb < g.prb_lo) g.prb_lo = prb; if (prb > g.prb_hi) g.prb_hi = prb; };
  auto set_both = [&](uint32_t prb) { if (prb < n && !g.prb_idx[0][prb]) { g.prb_idx[0][prb] = g.prb_idx[1][prb] = true; g.nof_prb++; mark(prb); } };
  if (d.alloc_type == 0) {
    const uint32_t nb = (n + P - 1) / P;
    for (uint32_t i = 0; i < nb; i++)
      if (d.rbg_bitmask & (1u << (nb - 1 - i)))
        for (uint32_t j = 0; j < P; j++) set_both(i * P + j);
  } else if (d.alloc_type == 1) {
    const uint32_t lb = ceil_log2(P), n1 = (n + P - 1) / P - lb - 1, p = d.rbg_subset;
    const uint32_t base = ((n - 1) / (P * P)) * P, edge = ((n - 1) / P) % P;
    const uint32_t subset = p < edge ? base + P : (p == edge ? base + ((n - 1) % P) + 1 : base);
    const uint32_t shift = d.shift ? subset - n1 : 0;
    for (uint32_t i = 0; i < n1; i++)
      if (d.vrb_bitmask & (1u << (n1 - 1 - i))) set_both(((i + shift) / P) * P * P + p * P + (i + shift) % P);
  } else if (d.alloc_type == 2) {
    uint32_t L, start;
    if (d.format == FORMAT1C) {
      const uint32_t st = ra_type2_n_rb_step(n), nv = ra_type2_n_vrb_dl(n, d.ngap2) / st;
      if (!nv) return false;
      ra_type2_from_riv(d.riv, L, start, nv, nv);
      L *= st; start *= st;
    } else {
      ra_type2_from_riv(d.riv, L, start, n, n);
    }
    if (!d.distributed) {
      for (uint32_t i = 0; i < L; i++) set_both(start + i);
    } else {
      const uint32_t nv = ra_type2_n_vrb_dl(n, d.ngap2);
      if (!nv) return false;
      for (uint32_t i = 0; i < L; i++) {
        uint32_t v = start + i, pe, po;
        if (v >= nv) continue;
        distributed_vrb_to_prb(n, d.ngap2, v, pe, po);
        if (pe < n && po < n) { g.prb_idx[0][pe] = true; g.prb_idx[1][po] = true; g.nof_prb++; mark(pe); mark(po); }
      }
    }
  } else {
    return false;
  }
  return g.nof_prb > 0;
}

int ra_tbs_from_idx(int i_tbs, uint32_t n_prb)
{
  if (i_tbs < 0 || i_tbs >= LSN_TBS_NROWS || n_prb < 1 || n_prb > 110) return -1;
  return lsn_tbs_table[i_tbs][n_prb - 1];
}

bool pdsch_re_usable(const Cell& cell, uint32_t sf_idx, uint32_t l, uint32_t k)
{
  const uint32_t nsl = cell.nslot(), lq = l % nsl;
  if (l >= cell.nsym()) return false;  // extended CP: rows 12, 13 of the 14-row grids do not exist
  const bool crs_symbol = cell.crs_symbol01(l);
  if (cell.nof_ports == 4 && lq == 1 && k % 3 == cell.id % 3) return false;  // CRS of ports 2, 3
  if (crs_symbol) {
    if (cell.nof_ports >= 2) { if (k % 3 == cell.id % 3) return false; }
    else if (k % 6 == ((lq == 0 ? 0u : 3u) + cell.id % 6) % 6) return false;
  }
  const uint32_t lo = 6 * cell.nof_prb - 36;
  if (k >= lo && k < lo + 72) {
    if ((sf_idx == 0 || sf_idx == 5) && (l == nsl - 2 || l == nsl - 1)) return false;  // SSS, PSS: the last two symbols of slots 0 and 10
    if (sf_idx == 0 && l >= nsl && l <= nsl + 3) return false;                           // PBCH: symbols 0-3 of slot 1
  }
  return true;
}

void cell_build_re_tables(Cell& cell)
{
  const uint32_t n = cell.nof_prb;
  auto t = std::make_shared<std::vector<uint16_t>>((size_t)3 * 5 * 2 * n, (uint16_t)0);
  const uint32_t cls_sf[3] = {0, 5, 1};
  Cell plain = cell;
  plain.re_count.reset();
  for (uint32_t cl = 0; cl < 3; cl++)
    for (uint32_t l0 = 0; l0 < 5; l0++)
      for (uint32_t l = l0; l < cell.nsym(); l++)
        for (uint32_t prb = 0; prb < n; prb++) {
          uint16_t c = 0;
          for (uint32_t k = 12 * prb; k < 12 * prb + 12; k++) c += pdsch_re_usable(plain, cls_sf[cl], l, k) ? 1 : 0;
          (*t)[((cl * 5 + l0) * 2 + l / cell.nslot()) * n + prb] += c;
        }
  cell.re_count = t;
  for (int f = 0; f < 9; f++) cell.fmt_size[f] = 0;
  for (int f = 0; f < 9; f++) cell.fmt_size[f] = dci_format_sizeof(plain, (DciFormat)f);
}

static uint32_t ra_dl_compute_nof_re(const Cell& cell, uint32_t sf_idx, uint32_t cfi, const PdschGrant& g)
{
  const uint32_t l0 = cfi + (cell.nof_prb <= 10 ? 1 : 0);
  uint32_t n = 0;
  if (cell.re_count && l0 < 5) {
    const uint32_t cl = sf_idx == 0 ? 0 : (sf_idx == 5 ? 1 : 2);
    const uint16_t* t0 = cell.re_count->data() + ((size_t)(cl * 5 + l0) * 2) * cell.nof_prb;
    const uint16_t* t1 = t0 + cell.nof_prb;
    for (uint32_t prb = g.prb_lo; prb <= g.prb_hi && prb < cell.nof_prb; prb++) n += (g.prb_idx[0][prb] ? t0[prb] : 0u) + (g.prb_idx[1][prb] ? t1[prb] : 0u);
    return n;
  }
  for (uint32_t l = l0; l < cell.nsym(); l++)
    for (uint32_t prb = 0; prb < cell.nof_prb; prb++)
      if (g.prb_idx[l / cell.nslot()][prb])
        for (uint32_t k = 12 * prb; k < 12 * prb + 12; k++) n += pdsch_re_usable(cell, sf_idx, l, k) ? 1 : 0;
  return n;
}

// dl_sniffer_compute_tb, dl_sniffer_pdsch.c:14-92
static bool dl_sniffer_compute_tb(bool use_alt, const DciDl& d, PdschGrant& g)
{
  for (int i = 0; i < 2; i++) {
    GrantTb& tb = g.tb[i];
    tb.mcs_idx = d.tb[i].mcs_idx; tb.rv = d.tb[i].rv; tb.cw_idx = d.tb[i].cw_idx;
    const bool tb_en = !(d.tb[i].mcs_idx == 0 && d.tb[i].rv == 1);
    tb.enabled = (tb_en && d.format >= FORMAT2) || (d.format < FORMAT2 && i == 0);
    if (tb.enabled) g.nof_tb++;
  }
  if (d.format == FORMAT1A || !rnti_isuser(d.rnti)) use_alt = false;
  if (!rnti_isuser(d.rnti)) {
    int tbs;
    if (d.format == FORMAT1A) {
      tbs = ra_tbs_from_idx((int)d.tb[0].mcs_idx, d.nprb1a_is2 ? 2 : 3);
      if (tbs < 0) return false;
    } else if (d.format == FORMAT1C) {
      if (d.tb[0].mcs_idx >= 32) return false;
      tbs = lsn_tbs_format1c_table[d.tb[0].mcs_idx];
    } else {
      return false;
    }
    g.tb[0].mod = 2;
    g.tb[0].tbs = tbs;
    return true;
  }
  const int8_t(*table)[2] = use_alt ? lsn_mcs_dl_256qam : lsn_mcs_dl_64qam;
  for (auto& tb : g.tb) {
    if (!tb.enabled) { tb.tbs = 0; continue; }
    tb.mod = table[tb.mcs_idx & 31][0];
    const int i_tbs = table[tb.mcs_idx & 31][1];
    tb.tbs = i_tbs >= 0 ? ra_tbs_from_idx(i_tbs, g.nof_prb) : 0;  // reserved MCS: last_tbs is 0 without HARQ
    if (tb.tbs < 0) return false;
  }
  return true;
}

bool dl_sniffer_ra_dl_dci_to_grant(const Cell& cell, uint32_t sf_idx, uint32_t cfi, bool use_alt, const DciDl& dci, PdschGrant& g)
{
  g = PdschGrant();
  if (!ra_dl_grant_to_grant_prb_allocation(cell, dci, g)) return false;
  if (!dl_sniffer_compute_tb(use_alt, dci, g)) return false;
  g.nof_re = ra_dl_compute_nof_re(cell, sf_idx, cfi, g);
  for (auto& tb : g.tb) tb.nof_bits = tb.enabled ? (int)g.nof_re * tb.mod : 0;
  if (dci.format == FORMAT1C && (rnti_israr(dci.rnti) || dci.rnti == PRNTI))
    for (auto& tb : g.tb) tb.rv = 0;
  return true;
}

// second half of dl_sniffer_ra_dl_dci_to_grant_both: g64 holds the PRB allocation (ra_dl_grant_to_grant_prb_allocation succeeded);
// fills the MCS / TBS / RE-count dependent fields for both MCS tables
void dl_sniffer_grant_finish_both(const Cell& cell, uint32_t sf_idx, uint32_t cfi, const DciDl& dci, PdschGrant& g64, bool& ok64, PdschGrant& g256,
                                  bool& ok256)
{
  g256 = g64;  // the PRB set does not depend on the MCS table
  ok64 = dl_sniffer_compute_tb(false, dci, g64);
  ok256 = dl_sniffer_compute_tb(true, dci, g256);
  if (!ok64 && !ok256) return;
  const uint32_t nof_re = ra_dl_compute_nof_re(cell, sf_idx, cfi, g64);
  const bool rv0 = dci.format == FORMAT1C && (rnti_israr(dci.rnti) || dci.rnti == PRNTI);
  for (int t = 0; t < 2; t++) {
    PdschGrant& g = t ? g256 : g64;
    if (!(t ? ok256 : ok64)) continue;
    g.nof_re = nof_re;
    for (auto& tb : g.tb) { tb.nof_bits = tb.enabled ? (int)nof_re * tb.mod : 0; if (rv0) tb.rv = 0; }
  }
}

void dl_sniffer_ra_dl_dci_to_grant_both(const Cell& cell, uint32_t sf_idx, uint32_t cfi, const DciDl& dci, PdschGrant& g64, bool& ok64,
                                        PdschGrant& g256, bool& ok256)
{
  g64 = PdschGrant();
  ok64 = ok256 = false;
  if (!ra_dl_grant_to_grant_prb_allocation(cell, dci, g64)) { g256 = g64; return; }
  dl_sniffer_grant_finish_both(cell, sf_idx, cfi, dci, g64, ok64, g256, ok256);
}

// ul_sniffer_ra_ul_grant_to_grant_prb_allocation (ul_sniffer_pusch.c:19-87) + the 64QAM MCS table
bool ra_ul_dci_to_grant(const Cell& cell, const DciUl& d, PuschGrant& g)
{
  g = PuschGrant();
  uint32_t L, start;
  const uint32_t nprb = cell.nof_prb;
  ra_type2_from_riv(d.riv, L, start, nprb, nprb);
  if (L == 0 || start + L > nprb) return false;
  g.L_prb = L; g.n_prb = start; g.n_prb2 = start;
  const int hop = d.hopping ? d.hop_type : -1;
  if (hop == 3) {
    g.hop = 2;  // type 2: same PRBs in the grant, hopped / mirrored at resource mapping (not decoded)
  } else if (hop >= 0) {  // type 1 (36.213 8.4.1): fixed offset between the two slots
    uint32_t n_rb_ho = cell.pusch_hop_offset;
    if (n_rb_ho % 2) n_rb_ho++;
    if (n_rb_ho + (nprb % 2) >= nprb) return false;
    const uint32_t n_rb_pusch = nprb - n_rb_ho - (nprb % 2);
    if (start < n_rb_ho / 2) return false;
    if (hop == 0) g.n_prb2 = (n_rb_pusch / 4 + start) % n_rb_pusch;
    else if (hop == 1) g.n_prb2 = start < n_rb_pusch / 4 ? n_rb_pusch + start - n_rb_pusch / 4 : start - n_rb_pusch / 4;
    else g.n_prb2 = (n_rb_pusch / 2 + start) % n_rb_pusch;
    g.hop = 1;
    if (g.n_prb2 + L > nprb) return false;
  }
  g.mcs_idx = d.mcs_idx;
  g.mod = lsn_mcs_ul_64qam[d.mcs_idx & 31][0];
  const int i_tbs = lsn_mcs_ul_64qam[d.mcs_idx & 31][1];
  if (i_tbs >= 0) g.tbs = ra_tbs_from_idx(i_tbs, L); else g.rv = (int)d.mcs_idx - 28;
  return true;
}

int rar_parse(const Cell& cell, const uint8_t* p, int len, RarEntry* out, int cap)
{
  int nsub = 0, pos = 0, n = 0;
  bool is_rapid[32];
  uint8_t rapid[32];
  while (pos < len && nsub < 32) {
    const uint8_t b = p[pos++];
    is_rapid[nsub] = (b & 0x40) != 0;
    rapid[nsub] = b & 0x3F;
    nsub++;
    if (!(b & 0x80)) break;
  }
  for (int i = 0; i < nsub && n < cap; i++) {
    RarEntry r;
    uint32_t grant20 = 0;
    if (is_rapid[i]) {
      if (pos + 6 > len) break;
      r.rapid = rapid[i];
      r.ta = ((uint32_t)(p[pos] & 0x7F) << 4) | (p[pos + 1] >> 4);
      grant20 = ((uint32_t)(p[pos + 1] & 0x0F) << 16) | ((uint32_t)p[pos + 2] << 8) | p[pos + 3];
      r.t_crnti = (uint16_t)((p[pos + 4] << 8) | p[pos + 5]);
      pos += 6;
    }
    DciUl d;
    d.rnti = r.t_crnti; d.hopping = (grant20 >> 19) & 1u; d.riv = (grant20 >> 9) & 0x3FFu; d.mcs_idx = (grant20 >> 5) & 0xFu;
    d.hop_type = d.hopping ? 1 : -1;  // ul_sniffer_dci_rar_to_ul_dci, falcon_dci.c:665-670: "freq_hop_fl = 1" = the -N/4 type-1 pattern on the full RIV
    r.hopping = d.hopping; r.riv = d.riv; r.mcs = d.mcs_idx; r.tpc = (grant20 >> 2) & 7u; r.ul_delay = (grant20 >> 1) & 1u; r.csi_req = grant20 & 1u;
    r.grant_ok = ra_ul_dci_to_grant(cell, d, r.grant);
    if (!r.grant_ok) r.grant = PuschGrant();
    out[n++] = r;
  }
  return n;
}

// ul_fill_ra_mcs_256 (ul_sniffer_pusch.c:91-136): Table 8.6.1-3 of 36.213 incl. the 32A row
bool ra_ul_dci_to_grant_256(const Cell& cell, const DciUl& d, PuschGrant& g)
{
  if (!ra_ul_dci_to_grant(cell, d, g)) return false;
  const uint32_t m = d.mcs_idx, L = g.L_prb;
  if (m <= 28) {
    g.rv = 0;
    if (m < 6) { g.mod = 2; g.tbs = ra_tbs_from_idx((int)m * 2, L); }
    else if (m < 14) { g.mod = 4; g.tbs = ra_tbs_from_idx((int)m + (m < 10 ? 5 : 6), L); }
    else if (m < 23) { g.mod = 6; g.tbs = ra_tbs_from_idx((int)m + (m < 19 ? 6 : 7), L); }
    else {
      g.mod = 8;
      if (m < 26) g.tbs = ra_tbs_from_idx((int)m + 7, L);
      else if (m == 26) g.tbs = (L > 0 && L < 111) ? lsn_tbs_table_32A[L - 1] : 0;
      else g.tbs = ra_tbs_from_idx((int)m + 6, L);
    }
  } else {
    g.mod = 0; g.tbs = 0; g.rv = (int)m - 28;  // last_tb is empty without HARQ state
  }
  return true;
}

int ulTrialPlan(uint32_t mcs, int qb, uint32_t L256, int mod256, int mod, UlTry a[3])
{
  const bool ok256 = L256 < 110 && L256 > 0;
  const int q16 = qb > 4 ? 4 : qb;
  int n = 0;
  auto add = [&](bool use256, int qm, int l) { a[n].use256 = use256; a[n].qm = qm; a[n].learn = l; n++; };
  if (mcs > 20 && mcs < 29) {
    if (mod == 2) add(false, 4, 0);
    else if (mod == 3) add(false, qb, 0);
    else if (mod == 4) { if (ok256) add(true, mod256, 0); }
    else if (mod == 1) { add(false, 4, 2); add(false, qb, 3); if (ok256) add(true, mod256, 4); }
  } else if (mcs <= 20) {  // decode_run's second rule (UL_Sniffer_PUSCH.cc:300-303): a passing 256QAM-table attempt with MCS > 0 reports 256QAM_MAX
    const int l256 = mcs > 0 ? 4 : 0;
    if (mod == 2 || mod == 3) add(false, q16, 0);
    else if (mod == 4) { if (ok256) add(true, mod256, l256); }
    else if (mod == 1) { add(false, q16, 0); if (ok256) add(true, mod256, l256); }
  }
  return n;
}

bool ulGrantValid(uint16_t rnti, bool is_rar, int tbs, int tbs_256, uint32_t L_prb)
{
  if (rnti == 0) return false;
  if (is_rar) return true;
  if (tbs == 0 || tbs_256 == 0) return false;
  return ul_valid_prb(L_prb) && L_prb <= 100;
}

bool ul_valid_prb(uint32_t L)
{
  if (L == 0 || L > 110) return false;
  while (L % 2 == 0) L /= 2;
  while (L % 3 == 0) L /= 3;
  while (L % 5 == 0) L /= 5;
  return L == 1;
}

// dl_sniffer_config_mimo, dl_sniffer_pdsch.c:134-276
int dl_sniffer_config_mimo(const Cell& cell, DciFormat f, const DciDl& dci, PdschGrant& g)
{
  switch (f) {
    case FORMAT1: case FORMAT1A: case FORMAT1C: g.tx_scheme = cell.nof_ports == 1 ? TXSCHEME_PORT0 : TXSCHEME_DIVERSITY; break;
    case FORMAT2: g.tx_scheme = (g.nof_tb == 1 && dci.pinfo == 0) ? TXSCHEME_DIVERSITY : TXSCHEME_SPATIALMUX; break;
    case FORMAT2A: g.tx_scheme = (g.nof_tb == 1 && dci.pinfo == 0) ? TXSCHEME_DIVERSITY : TXSCHEME_CDD; break;
    default: return 1;
  }
  if (g.tx_scheme == TXSCHEME_SPATIALMUX) {
    if (g.nof_tb == 1) { if (dci.pinfo < 1 || dci.pinfo > 4) return 2; g.pmi = dci.pinfo - 1; }
    else { if (dci.pinfo >= 2) return 2; g.pmi = dci.pinfo % 2; }
  }
  switch (g.tx_scheme) {
    case TXSCHEME_PORT0: if (g.nof_tb != 1) return 3; g.nof_layers = 1; break;
    case TXSCHEME_DIVERSITY: if (g.nof_tb != 1) return 3; g.nof_layers = cell.nof_ports; break;
    case TXSCHEME_SPATIALMUX: if (g.nof_tb != 1 && g.nof_tb != 2) return 3; g.nof_layers = g.nof_tb; break;
    case TXSCHEME_CDD: if (g.nof_tb != 2) return 3; g.nof_layers = 2; break;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------- search space (36.213 9.1.1)
namespace {
struct Loc { uint32_t L, ncce; };
uint32_t ue_locations(uint32_t nof_cce, Loc* c, uint32_t cap, uint32_t nsubframe, uint16_t rnti)
{
  static const uint32_t ncand[4] = {6, 6, 2, 2};
  uint32_t Yk = rnti, k = 0;
  for (uint32_t m = 0; m <= nsubframe; m++) Yk = (39827u * Yk) % 65537u;
  for (int l = 3; l >= 0; l--) {
    const uint32_t L = 1u << l;
    if (nof_cce < L) continue;
    for (uint32_t i = 0; i < ncand[l]; i++) {
      const uint32_t ncce = L * ((Yk + i) % (nof_cce / L));
      if (k < cap && ncce + L <= nof_cce) c[k++] = {(uint32_t)l, ncce};
    }
  }
  return k;
}
uint32_t common_locations(uint32_t nof_cce, Loc* c, uint32_t cap)
{
  uint32_t k = 0;
  for (int l = 3; l > 1; l--) {
    const uint32_t L = 1u << l;
    for (uint32_t i = 0; i < std::min(nof_cce, 16u) / L; i++) {
      const uint32_t ncce = L * (i % (nof_cce / L));
      if (k < cap && ncce + L <= nof_cce) c[k++] = {(uint32_t)l, ncce};
    }
  }
  return k;
}
}  // namespace

uint32_t pdcch_validate_location(uint32_t nof_cce, uint32_t ncce, uint32_t l, uint32_t nsubframe, uint16_t rnti)
{
  Loc loc[22];
  uint32_t n = 0;
  if (rnti_israr(rnti)) n = common_locations(nof_cce, loc, 22);
  else if (rnti_isuser(rnti)) { n = ue_locations(nof_cce, loc, 22, nsubframe, rnti); n += common_locations(nof_cce, loc + n, 22 - n); }
  else if (rnti >= MRNTI) n = common_locations(nof_cce, loc, 22);
  bool ambiguous = false, valid = false;
  for (uint32_t i = 0; i < n; i++) {
    if (loc[i].ncce != ncce) continue;
    if (l > 0 && loc[i].L == l - 1) ambiguous = true;
    if (loc[i].L == l) valid = true;
  }
  return valid ? (ambiguous ? 1 : 2) : 0;
}

// ---------------------------------------------------------------------------------------------- segmentation
bool cbsegm(int tbs, CbSegm& s)
{
  s = CbSegm();
  if (tbs <= 0) return false;
  const int B = tbs + 24;
  int Bp = B;
  s.C = 1;
  if (B > 6144) { s.C = (B + 6119) / 6120; Bp = B + 24 * s.C; }
  int idx = -1;
  for (int i = 0; i < LSN_QPP_NSIZES && idx < 0; i++) if (s.C * (int)lsn_qpp_table[i][0] >= Bp) idx = i;
  if (idx < 0) return false;
  s.Kp = lsn_qpp_table[idx][0];
  if (s.C == 1) { s.Cp = 1; }
  else {
    if (idx == 0) return false;
    s.Km = lsn_qpp_table[idx - 1][0];
    s.Cm = (s.C * s.Kp - Bp) / (s.Kp - s.Km);
    s.Cp = s.C - s.Cm;
  }
  s.F = s.Cp * s.Kp + s.Cm * s.Km - Bp;
  return true;
}
uint32_t turbo_il_offset(int K)
{
  // called once per code block by the decode threads: a table indexed by K / 8, built on first use (thread-safe static initialisation)
  struct Tab {
    uint32_t off[6144 / 8 + 1], total;
    Tab()
    {
      uint32_t o = 0;
      for (auto& v : off) v = 0xFFFFFFFFu;
      for (int i = 0; i < LSN_QPP_NSIZES; i++) { off[lsn_qpp_table[i][0] / 8] = o; o += (uint32_t)lsn_turbo_il_words(lsn_qpp_table[i][0]); }
      total = o;
    }
  };
  static const Tab tab;
  if (K <= 0 || K > 6144 || (K & 7) || tab.off[K / 8] == 0xFFFFFFFFu) return tab.total;  // K not a block size: the total number of words
  return tab.off[K / 8];
}
uint32_t turbo_nwin(int K)
{
  // lsn_turbo_nwin(K) (lsn_rm.h: a search over the divisors of K) from a table by K / 8, built on first use: the decode threads ask it three times per code block,
  // and the kernels take it from the descriptor (LsnCbDev::nwin) instead of running the search once per THREAD (round 6: a third of k_rm's vector instructions)
  struct Tab {
    uint8_t n[6144 / 8 + 1];
    Tab() { for (int k = 0; k <= 6144 / 8; k++) n[k] = (uint8_t)(k ? lsn_turbo_nwin(8 * k) : 1); }
  };
  static const Tab tab;
  if (K <= 0 || K > 6144 || (K & 7)) return (uint32_t)lsn_turbo_nwin(K);
  return tab.n[K / 8];
}
bool qpp_params(int K, uint32_t& f1, uint32_t& f2)
{
  for (int i = 0; i < LSN_QPP_NSIZES; i++) if (lsn_qpp_table[i][0] == K) { f1 = lsn_qpp_table[i][1]; f2 = lsn_qpp_table[i][2]; return true; }
  return false;
}

// ---------------------------------------------------------------------------------------------- Histogram / RNTIManager
Histogram::Histogram(uint32_t itemCount, uint32_t valueRange)
    : rnti_histogram(valueRange, 0), rnti_history(itemCount, 0), rnti_history_current(0), rnti_history_end(itemCount), rnti_histogram_ready(false),
      nz_pos(itemCount + 1, 0), nz_head(0), nz_tail(0) {}
// The window is a ring that is written in order, so its non-zero entries leave it in the order they came: their positions wait in a queue (nz_pos), and
// moving the window over a stretch of padding costs a look at the head of that queue instead of a look at every slot (round 6: the nine windows of a
// subframe were a fifth of the sequential search's time - and that thread is what bounds one cell).
void Histogram::add(uint16_t item, uint32_t nTimes)
{
  while (nTimes-- > 0) {
    if (rnti_histogram_ready) {
      const uint16_t old = rnti_history[rnti_history_current];
      rnti_histogram[old]--; total[old]--;
      if (old) { if (++nz_head == nz_pos.size()) nz_head = 0; }   // (the oldest non-zero entry: the head of the queue)
    }
    rnti_history[rnti_history_current] = item;
    rnti_histogram[item]++;
    total[item]++;
    if (item) { nz_pos[nz_tail] = rnti_history_current; if (++nz_tail == nz_pos.size()) nz_tail = 0; }
    if (++rnti_history_current == rnti_history_end) { rnti_histogram_ready = true; rnti_history_current = 0; }
  }
}

void Histogram::addZeros(uint32_t nTimes)
{
  uint32_t gained = 0;  // net increase of the count of item 0
  while (nTimes > 0 && !rnti_histogram_ready) {   // the first turn of the window: nothing leaves it
    rnti_history[rnti_history_current] = 0;
    gained++; nTimes--;
    if (++rnti_history_current == rnti_history_end) { rnti_histogram_ready = true; rnti_history_current = 0; }
  }
  while (nTimes > 0) {
    const uint32_t n = std::min(nTimes, rnti_history_end - rnti_history_current), lim = rnti_history_current + n;   // (no wrap inside this stretch)
    while (nz_head != nz_tail && nz_pos[nz_head] >= rnti_history_current && nz_pos[nz_head] < lim) {   // the non-zero entries of the stretch leave the window
      const uint32_t p = nz_pos[nz_head];
      const uint16_t old = rnti_history[p];
      rnti_histogram[old]--; total[old]--; rnti_history[p] = 0; gained++;
      if (++nz_head == nz_pos.size()) nz_head = 0;
    }
    rnti_history_current = lim == rnti_history_end ? 0 : lim;
    nTimes -= n;
  }
  rnti_histogram[0] += gained;
  total[0] += gained;
}

RNTIManager::RNTIManager(uint32_t nf, uint32_t maxCand, uint32_t thr)
    : nformats(nf), histograms(nf, Histogram(200 * (304 / 5), 65536)), evergreen(nf), forbidden(nf), active(65536, 0), reason(65536, 0),
      lastSeen(65536, 0), assocFormatIdx(65536, 0), nactive(0), timestamp(0), lifetime(10000), threshold(thr),
      maxCandidatesPerStepPerFormat(maxCand), remainingCandidates(nf, (int32_t)maxCand)
{
  totals.assign(65536, 0);
  active_bits.assign(2048, 0u);
  rar_bits.assign(2048, 0u);
  for (auto& h : histograms) h.setTotals(totals.data());
}
void RNTIManager::addCandidate(uint16_t rnti, uint32_t f) { histograms[f].add(rnti); remainingCandidates[f]--; }
bool RNTIManager::isEvergreen(uint16_t rnti, uint32_t f) const { for (auto& i : evergreen[f]) if (i.matches(rnti)) return true; return false; }
bool RNTIManager::isForbidden(uint16_t rnti, uint32_t f) const { for (auto& i : forbidden[f]) if (i.matches(rnti)) return true; return false; }
void RNTIManager::activateRNTI(uint16_t rnti, ActivationReason r)
{
  if (!active[rnti]) {
    active[rnti] = 1; active_bits[rnti >> 5] |= 1u << (rnti & 31u); reason[rnti] = (uint8_t)r; nactive++;
    if (r == RM_ACT_RAR) rar_bits[rnti >> 5] |= 1u << (rnti & 31u);
  }
}
void RNTIManager::deactivateRNTI(uint16_t rnti)
{
  if (active[rnti]) { active[rnti] = 0; active_bits[rnti >> 5] &= ~(1u << (rnti & 31u)); rar_bits[rnti >> 5] &= ~(1u << (rnti & 31u)); assocFormatIdx[rnti] = 0; nactive--; }
}
uint32_t RNTIManager::getLikelyDlFormatIdx(uint16_t rnti) const
{
  uint32_t best = 0, mx = 0;
  for (uint32_t f = 1; f < nformats; f++) { uint32_t c = histograms[f].getFrequency(rnti); if (c > mx) { mx = c; best = f; } }
  return best;
}
bool RNTIManager::validate(uint16_t rnti, uint32_t f)
{
  if (isEvergreen(rnti, f)) return true;
  if (isForbidden(rnti, f)) return false;
  if ((active_bits[rnti >> 5] >> (rnti & 31u)) & 1u) {   // (active[rnti], from the 8 KB of bits instead of the 64 KB of bytes: the search asks for random RNTIs)
    if (timestamp - lastSeen[rnti] < lifetime) return true;
    deactivateRNTI(rnti);
  }
  // RNTIManager::validateByHistogram needs freq_UL + freq_bestDL > threshold; both are bounded by the sum over all formats
  if (totals[rnti] <= threshold) return false;
  const uint32_t likely = getLikelyDlFormatIdx(rnti);
  if (f != 0 && f != likely) return false;
  const uint32_t ul = histograms[0].getFrequency(rnti), dl = likely ? histograms[likely].getFrequency(rnti) : 0;
  if (ul + dl <= threshold) return false;
  activateRNTI(rnti, RM_ACT_HISTOGRAM);
  assocFormatIdx[rnti] = dl > threshold ? likely : 0;
  return true;
}
bool RNTIManager::validateAndRefresh(uint16_t rnti, uint32_t f) { bool ok = validate(rnti, f); if (ok) lastSeen[rnti] = timestamp; return ok; }
void RNTIManager::activateAndRefresh(uint16_t rnti, uint32_t f, ActivationReason r) { activateRNTI(rnti, r); lastSeen[rnti] = timestamp; assocFormatIdx[rnti] = f; }
void RNTIManager::stepTime()
{
  for (uint32_t i = 0; i < nformats; i++) {
    if (remainingCandidates[i] > 0) histograms[i].addZeros((uint32_t)remainingCandidates[i]);
    remainingCandidates[i] = (int32_t)maxCandidatesPerStepPerFormat;
  }
  timestamp++;
}

// ---------------------------------------------------------------------------------------------- DCIMetaFormats
DCIMetaFormats::DCIMetaFormats(uint32_t nformats, double ratio) : all(nformats), primary(nformats), secondary(nformats), split_ratio(ratio)
{
  for (uint32_t i = 0; i < nformats; i++) all[i] = {falcon_ue_all_formats[i], i, 0};
  update_formats();
}
void DCIMetaFormats::update_formats()
{
  const int n = (int)all.size();
  std::vector<MetaFormat*> sorted(n);
  double total = 0;
  for (int i = 0; i < n; i++) { sorted[i] = &all[i]; total += all[i].hits; }
  for (int i = 0; i < n - 1; i++) {  // selection sort exactly as MetaFormats.cc:52-62 (ties keep the first maximum)
    int mx = i;
    for (int j = mx; j < n; j++) if (sorted[j]->hits > sorted[mx]->hits) mx = j;
    std::swap(sorted[i], sorted[mx]);
  }
  const double thr = total * split_ratio;
  double cum = 0;
  nprimary = nsecondary = 0;
  for (int i = 0; i < n; i++) {
    if (cum <= thr) primary[nprimary++] = sorted[i]; else secondary[nsecondary++] = sorted[i];
    cum += sorted[i]->hits;
    sorted[i]->hits = 0;
  }
}

// ---------------------------------------------------------------------------------------------- MCSTracking
McsTable MCSTracking::find_tracking_info_RNTI_dl(uint16_t rnti, uint32_t now)
{
  if (!db[rnti].present) return count < max_size ? TABLE_UNKNOWN : TABLE_FULL_BUFFER;
  db[rnti].time = now;
  return (McsTable)db[rnti].table;
}
void MCSTracking::add_RNTI_dl(uint16_t rnti, uint32_t now)
{
  if (db[rnti].present) return;
  db[rnti] = Entry();
  db[rnti].present = 1;
  db[rnti].time = now;
  count++;
  ue_cfg[rnti] = default_cfg;  // MCSTracking.cc:791-792
  ue_cfg[rnti].has_ue_config = false;
}
UeSpecConfig MCSTracking::get_ue_config_rnti(uint16_t rnti) const
{
  if (db[rnti].present) return ue_cfg[rnti];
  UeSpecConfig c = default_cfg;
  c.has_ue_config = false;
  return c;
}
void MCSTracking::update_ue_config_rnti(uint16_t rnti, const UeSpecConfig& c, uint32_t now)
{
  add_RNTI_dl(rnti, now);
  ue_cfg[rnti] = c;
}
int MCSTracking::setups_of_pdu(const uint8_t* pdu, int len, UeSpecConfig* out, int cap, bool any_lcid)
{
  MacSubheader sub[20];
  const int n = mac_dlsch_parse(pdu, len, sub, 20);
  int k = 0;
  for (int i = 0; i < n && k < cap; i++) {
    if (!(sub[i].is_sdu && (any_lcid || sub[i].lcid == 0))) continue;
    UeSpecConfig c;
    if (rrc_conn_setup_decode(pdu + sub[i].off, (int)sub[i].len, c)) { c.from_lcid0 = sub[i].lcid == 0; out[k++] = c; }
  }
  return k;
}
bool MCSTracking::learn_setups(const UeSpecConfig* c, int n, uint16_t rnti, uint32_t now, bool any_lcid)
{
  bool any = false;
  for (int i = 0; i < n; i++) {
    if (!any_lcid && !c[i].from_lcid0) continue;
    UeSpecConfig u = c[i];
    u.from_lcid0 = true;
    if (!has_default) update_default_ue_config(u);  // the first connection setup seen, DL_Sniffer_PDSCH.cc:1061-1065
    update_ue_config_rnti(rnti, u, now);
    any = true;
  }
  return any;
}
bool MCSTracking::learn_from_pdu(const uint8_t* pdu, int len, uint16_t rnti, uint32_t now)
{
  UeSpecConfig c[20];
  return learn_setups(c, setups_of_pdu(pdu, len, c, 20), rnti, now);
}
void MCSTracking::update_RNTI_dl(uint16_t rnti, McsTable t, uint32_t now)
{
  Entry& e = db[rnti];
  if (!e.present) { add_RNTI_dl(rnti, now); return; }
  if (e.has_rar) {
    if (e.nof_msg_after_rar > rar_thresold) { e.table = (uint8_t)t; e.has_rar = 0; }
    else e.table = TABLE_UNKNOWN;
  } else {
    e.table = (uint8_t)t;
  }
}
void MCSTracking::update_rar_time_crnti(uint16_t crnti, uint32_t now) { add_RNTI_dl(crnti, now); db[crnti].has_rar = 1; db[crnti].table = TABLE_UNKNOWN; }
void MCSTracking::update_statistic_dl(uint16_t rnti, DciFormat f, McsTable table, const bool tb_en[2], const bool success[2], int mimo_ret, uint32_t now)
{
  add_RNTI_dl(rnti, now);
  Entry& e = db[rnti];
  if (f > FORMAT1A && e.has_rar) e.nof_msg_after_rar++;
  if (table == TABLE_64QAM || table == TABLE_256QAM || table == TABLE_UNKNOWN)  // :1292-1384, transmission_type = NEW_TX throughout
    for (int i = 0; i < 2; i++) {
      if (tb_en[i]) e.nof_active++;
      if (success[i]) e.nof_success_mgs++;
      if (mimo_ret == -1 && tb_en[i]) e.nof_unsupport_mimo++;
      else if (mimo_ret == -2 && tb_en[i]) e.nof_pinfo++;
      else if (mimo_ret == -3 && tb_en[i]) e.nof_other_mimo++;
    }
}
void MCSTracking::update_database_dl(uint32_t now, std::vector<uint16_t>* changed)
{
  for (uint32_t r = 0; r < 65536; r++) {
    Entry& e = db[r];
    if (!e.present) continue;
    const uint32_t cur_interval = (now - e.time) / 1000u;  // whole seconds, like (cur_time - last_time) / CLOCKS_PER_SEC
    const bool wrong_detect = e.nof_active == 0 || (e.nof_active <= 10 && e.nof_success_mgs == 0 && (e.nof_unsupport_mimo > 0 || e.nof_pinfo > 0 || e.nof_other_mimo > 0));
    if (cur_interval > interval || wrong_detect || e.nof_active == 0) {
      e = Entry();
      count--;
      if (changed) changed->push_back((uint16_t)r);
    } else if ((float)e.nof_success_mgs / (float)e.nof_active < 0.15f && e.table != TABLE_UNKNOWN) {
      e.table = TABLE_UNKNOWN;
      if (changed) changed->push_back((uint16_t)r);
    }
  }
}

// ---------------------------------------------------------------------------------------------- CRC helpers
uint32_t crc_bits(uint32_t poly, int order, const uint8_t* bits, int n)
{
  uint32_t reg = 0, top = 1u << order;
  for (int i = 0; i < n + order; i++) { reg = (reg << 1) | (i < n ? (bits[i] & 1u) : 0u); if (reg & top) reg ^= poly; }
  return reg & (top - 1);
}
uint32_t crc24a_mulmod(uint32_t a, uint32_t b)
{
  uint32_t r = 0;
  for (int i = 23; i >= 0; i--) {
    r <<= 1;
    if (r & 0x1000000u) r ^= 0x1864CFBu;
    if ((b >> i) & 1u) r ^= a;
  }
  return r & 0xFFFFFFu;
}
// x^(8 nbytes) mod g for the payload sizes of code blocks (at most 768 bytes): computed once per size
uint32_t crc24a_xpow_bytes(uint32_t nbytes)
{
  static std::atomic<uint32_t> memo[1024];   // 0 = not computed yet (a power of x is never 0 mod g)
  if (nbytes >= 1024) return crc24a_xpow(8ull * nbytes);
  uint32_t v = memo[nbytes].load(std::memory_order_relaxed);
  if (!v) { v = crc24a_xpow(8ull * nbytes); memo[nbytes].store(v, std::memory_order_relaxed); }
  return v;
}
uint32_t crc24a_xpow(uint64_t n)
{
  uint32_t result = 1, base = 2;  // x^0, x^1
  while (n) { if (n & 1) result = crc24a_mulmod(result, base); base = crc24a_mulmod(base, base); n >>= 1; }
  return result;
}

// ------------------------------------------------------------------------------------------------ HARQ database
HarqRet HarqDatabase::is_retransmission(uint16_t rnti, uint32_t pid, int tid, bool ndi, int tbs, uint32_t sfn, uint32_t sf_idx, int& entity)
{
  // The reference scans its 300 entities for the RNTI and remembers the LAST free one on the way (HARQ.cc:84-90).  An RNTI owns at most one entity (one is
  // only given out when the scan found none), so the scan's answer is an index look-up; the free entity is only needed when there is none, which is rare.
  // (Round 6: the scan ran once per transport block in the commit walk and in every pass of harqScout - most of the HARQ leg's commit time.)
  int found = rnti ? (int)ent_of_rnti[rnti] : -1, avail = -1;
  if (found < 0)
    for (int i = 0; i < NENT; i++) {
      if (ent[i].rnti == rnti) found = i;   // (rnti 0: every free entity "matches", as in the reference's scan)
      else if (ent[i].rnti == 0) avail = i;
    }
  entity = found;
  HarqRet r;
  if (found < 0 && avail >= 0) {
    ent[avail].rnti = rnti;
    if (rnti) ent_of_rnti[rnti] = (int16_t)avail;
    if (nof_aval > 0) nof_aval--;
    entity = avail;
    r = HARQ_NEW_TX;
  } else if (found < 0) {
    r = HARQ_FULL_BUFFER;
  } else {
    const Tb& t = ent[found].tb[pid & 7][tid];
    const uint32_t last_tti = t.sfn * 10 + t.sf_idx, cur_tti = sfn * 10 + sf_idx;
    if (!(cur_tti - last_tti == 8 || cur_tti + 10240 - last_tti == 8)) r = HARQ_NEW_TX;  // comparetti
    else if (ndi != t.ndi || t.is_first || t.tbs != tbs) r = HARQ_NEW_TX;
    else r = t.last_decoded ? HARQ_DECODED : HARQ_RE_TX;
  }
  stats[r]++;
  return r;
}
void HarqDatabase::update(int entity, uint32_t pid, int tid, uint32_t sfn, uint32_t sf_idx, bool last_decoded, bool ndi, int rv, int tbs, uint32_t now)
{
  if (entity < 0) return;
  Entity& e = ent[entity];
  e.time = now;
  Tb& t = e.tb[pid & 7][tid];
  t.sfn = sfn; t.sf_idx = sf_idx; t.last_decoded = last_decoded; t.ndi = ndi; t.rv = rv; t.tbs = tbs; t.is_first = false;
}
int HarqDatabase::getlastTbs(uint16_t rnti, uint32_t pid, int tid) const
{
  int tbs = 0;
  for (const Entity& e : ent)
    if (e.rnti == rnti) tbs = e.tb[pid & 7u][tid & 1].tbs;
  return tbs;
}

void HarqDatabase::update_database(uint32_t now)
{
  if (nof_aval > 10) return;
  for (size_t i = 0; i < ent.size(); i++) {
    Entity& e = ent[i];
    if ((now - e.time) / 1000u > 5u) {  // (free entities included, as in the reference: nof_aval over-counts)
      if (e.rnti) ent_of_rnti[e.rnti] = -1;
      e.rnti = 0; e.time = 0;
      for (auto& p : e.tb) for (Tb& t : p) { t.is_first = true; t.last_decoded = false; t.tbs = 0; t.sf_idx = 0; }
      nof_aval++;
    }
  }
}

}  // namespace lsn
