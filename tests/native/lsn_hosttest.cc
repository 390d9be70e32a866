// Test glue (NOT product): exposes the HIP-free host logic of the product (lsn_lte.cc, lsn_search.cc) through a
// C interface so that the CPU test-suite can compare it with the oracle without a GPU.
#include "../../ltesniffer_amd/csrc/host/lsn_search.h"
#include "../../ltesniffer_amd/csrc/kernels/lsn_rm.h"
#include <chrono>
#include <cmath>
#include <cstring>

using namespace lsn;

struct hsearch {
  Cell cell;
  std::unique_ptr<FalconSearch> s;
  SubframeCtx ctx;
};

// same layout as oracle/lsn_oracle.h o_pdsch_grant_t
struct grant_out {
  uint8_t prb_idx[2][110];
  uint32_t nof_prb, nof_re, nof_tb;
  struct { uint32_t mcs_idx; int rv; uint32_t cw_idx; int enabled; int mod; int tbs; int nof_bits; } tb[2];
  int tx_scheme; uint32_t pmi; uint32_t nof_layers;
};

extern "C" {

uint32_t lsnh_dci_format_sizeof(uint32_t nof_prb, uint32_t nof_ports, int format)
{
  Cell c; c.nof_prb = nof_prb; c.nof_ports = nof_ports;
  return dci_format_sizeof(c, (DciFormat)format);
}

uint32_t lsnh_validate_location(const uint32_t* nof_cce3, uint32_t cfi, uint32_t ncce, uint32_t l, uint32_t nsubframe, uint16_t rnti)
{
  SearchSpace sp;
  sp.init(nof_cce3);
  return sp.validate(cfi, ncce, l, nsubframe, rnti);
}
// brute-force form kept in lsn_lte.cc (enumerates the locations like the reference does)
uint32_t lsnh_validate_location_enum(uint32_t nof_cce, uint32_t ncce, uint32_t l, uint32_t nsubframe, uint16_t rnti)
{
  return pdcch_validate_location(nof_cce, ncce, l, nsubframe, rnti);
}

// unpack + dci->grant (+ config_mimo) for one DL DCI; returns bit0 unpack ok, bit1 grant ok, bits 8.. mimo return code
int lsnh_dl_grant(uint32_t nof_prb, uint32_t nof_ports, uint32_t cell_id, uint32_t sf_idx, uint32_t cfi, int alt, const uint8_t* payload,
                  uint32_t nof_bits, int format, uint16_t rnti, int do_mimo, grant_out* out)
{
  Cell c; c.nof_prb = nof_prb; c.nof_ports = nof_ports; c.id = cell_id;
  DciDl d;
  std::memset(out, 0, sizeof(*out));
  if (!dci_msg_unpack_pdsch(c, payload, nof_bits, (DciFormat)format, rnti, d)) return 0;
  PdschGrant g;
  int r = 1;
  bool ok = dl_sniffer_ra_dl_dci_to_grant(c, sf_idx, cfi, alt != 0, d, g);
  {  // the fused two-table variant with the RE-count tables must agree with the plain single-table call
    Cell c2 = c;
    cell_build_re_tables(c2);
    PdschGrant a, b; bool oka, okb;
    dl_sniffer_ra_dl_dci_to_grant_both(c2, sf_idx, cfi, d, a, oka, b, okb);
    const PdschGrant& x = alt ? b : a;
    const bool okx = alt ? okb : oka;
    if (okx != ok) return -99;
    if (ok && (std::memcmp(x.prb_idx, g.prb_idx, sizeof(g.prb_idx)) || x.nof_prb != g.nof_prb || x.nof_re != g.nof_re || x.nof_tb != g.nof_tb)) return -98;
    for (int i = 0; ok && i < 2; i++)
      if (x.tb[i].mod != g.tb[i].mod || x.tb[i].tbs != g.tb[i].tbs || x.tb[i].nof_bits != g.tb[i].nof_bits || x.tb[i].rv != g.tb[i].rv ||
          x.tb[i].enabled != g.tb[i].enabled || x.tb[i].cw_idx != g.tb[i].cw_idx || x.tb[i].mcs_idx != g.tb[i].mcs_idx) return -97;
  }
  if (ok) r |= 2;
  if (ok && do_mimo) r |= dl_sniffer_config_mimo(c, (DciFormat)format, d, g) << 8;
  for (int s = 0; s < 2; s++)
    for (uint32_t i = 0; i < nof_prb; i++) out->prb_idx[s][i] = g.prb_idx[s][i];
  out->nof_prb = g.nof_prb; out->nof_re = g.nof_re; out->nof_tb = g.nof_tb;
  for (int i = 0; i < 2; i++) {
    out->tb[i].mcs_idx = g.tb[i].mcs_idx; out->tb[i].rv = g.tb[i].rv; out->tb[i].cw_idx = g.tb[i].cw_idx; out->tb[i].enabled = g.tb[i].enabled;
    out->tb[i].mod = g.tb[i].mod; out->tb[i].tbs = g.tb[i].tbs; out->tb[i].nof_bits = g.tb[i].nof_bits;
  }
  out->tx_scheme = (int)g.tx_scheme; out->pmi = g.pmi; out->nof_layers = g.nof_layers;
  return r;
}

int lsnh_ul_grant_hop(uint32_t nof_prb, uint32_t nof_ports, uint32_t hop_offset, const uint8_t* payload, uint32_t nof_bits, uint16_t rnti, uint32_t* out8)
{
  Cell c; c.nof_prb = nof_prb; c.nof_ports = nof_ports; c.pusch_hop_offset = hop_offset;
  DciUl d;
  if (!dci_msg_unpack_pusch(c, payload, nof_bits, rnti, d)) return 0;
  PuschGrant g;
  if (!ra_ul_dci_to_grant(c, d, g)) return 1;
  out8[0] = g.L_prb; out8[1] = g.n_prb; out8[2] = g.mcs_idx; out8[3] = (uint32_t)g.mod; out8[4] = (uint32_t)g.tbs; out8[5] = (uint32_t)g.rv; out8[6] = g.n_prb2; out8[7] = g.hop;
  return 3;
}
int lsnh_ul_grant(uint32_t nof_prb, uint32_t nof_ports, const uint8_t* payload, uint32_t nof_bits, uint16_t rnti, uint32_t* out6)
{
  Cell c; c.nof_prb = nof_prb; c.nof_ports = nof_ports;
  DciUl d;
  if (!dci_msg_unpack_pusch(c, payload, nof_bits, rnti, d)) return 0;
  PuschGrant g;
  if (!ra_ul_dci_to_grant(c, d, g)) return 1;
  out6[0] = g.L_prb; out6[1] = g.n_prb; out6[2] = g.mcs_idx; out6[3] = (uint32_t)g.mod; out6[4] = (uint32_t)g.tbs; out6[5] = (uint32_t)g.rv;
  return 3;
}

uint32_t lsnh_turbo_il_offset(int K) { return turbo_il_offset(K); }
int lsnh_turbo_two_wave_class(int K) { return lsn_turbo_two_wave_class(K) ? 1 : 0; }

int lsnh_cbsegm(int tbs, int* out6)
{
  CbSegm s;
  if (!cbsegm(tbs, s)) return -1;
  out6[0] = s.C; out6[1] = s.Cp; out6[2] = s.Cm; out6[3] = s.Kp; out6[4] = s.Km; out6[5] = s.F;
  return 0;
}

hsearch* lsnh_search_new(uint32_t nof_prb, uint32_t nof_ports, uint32_t cell_id, const uint32_t* nof_cce3, uint32_t threshold, double split, int skip)
{
  hsearch* h = new hsearch();
  h->cell.nof_prb = nof_prb; h->cell.nof_ports = nof_ports; h->cell.id = cell_id;
  h->s.reset(new FalconSearch(threshold, split, skip != 0));
  h->s->setCell(h->cell, nof_cce3);
  h->s->setupDefaultIntervals();
  return h;
}
void lsnh_search_free(hsearch* h) { delete h; }
void lsnh_search_set_shortcut_discovery(hsearch* h, int enable) { h->s->setShortcutDiscovery(enable != 0); }
int lsnh_search_size_index(hsearch* h, int format) { return h->s->sizeIndexOfFormat(format); }
uint32_t lsnh_search_nof_sizes(hsearch* h) { return h->s->nofSizes(); }
uint32_t lsnh_search_size(hsearch* h, uint32_t i) { return h->s->sizes()[i]; }
// returns number of accepted DCIs, 6 words each
uint32_t lsnh_search_run(hsearch* h, uint32_t tti, uint32_t cfi, float snr_db, const LsnCand* cand, const float* ccepow, int update_meta,
                         uint32_t* out, uint32_t max_words)
{
  h->ctx.reset(tti);
  h->ctx.cfi = cfi; h->ctx.snr_db = snr_db;
  h->s->search(h->ctx, cand, ccepow, update_meta != 0);
  h->s->finishSubframe(h->ctx);  // what the engine's decode threads do with the accepted DCIs (grants + collision statistics)
  const uint32_t n = (uint32_t)h->ctx.accepted.size();
  std::memcpy(out, h->ctx.accepted.data(), sizeof(uint32_t) * (n < max_words ? n : max_words));
  return n / 6;
}
void lsnh_search_activate_rar(hsearch* h, uint16_t rnti) { h->s->rntiManager().activateAndRefresh(rnti, 0, RM_ACT_RAR); }
void lsnh_search_stats(hsearch* h, uint32_t* out7)
{
  const BlindStats b = h->s->getStats();
  out7[0] = b.nof_decoded_locations; out7[1] = b.nof_cce; out7[2] = b.nof_missed_cce; out7[3] = b.nof_subframes;
  out7[4] = b.nof_subframe_collisions_dw; out7[5] = b.nof_subframe_collisions_up; out7[6] = b.nof_locations;
}
uint32_t lsnh_search_nof_active(hsearch* h) { return h->s->rntiManager().nofActive(); }
// microseconds per subframe of the search over n recorded subframes, repeated `reps` times (state keeps evolving)
double lsnh_search_bench(hsearch* h, uint32_t n, uint32_t reps, const uint32_t* tti, const uint32_t* cfi, const LsnCand* cand, const float* ccepow)
{
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t r = 0; r < reps; r++)
    for (uint32_t i = 0; i < n; i++) {
      h->ctx.reset(tti[i] + r * n);
      h->ctx.cfi = cfi[i]; h->ctx.snr_db = 20.0f;
      h->s->search(h->ctx, cand + (size_t)i * LSN_MAX_LOC * LSN_MAX_SIZES, ccepow + (size_t)i * LSN_CCE_STRIDE, (r * n + i) % 500 == 0 && (r || i));  // LTESniffer_Core.cc:434
    }
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  return us / ((double)n * reps);
}

int lsnh_tbs(int i_tbs, uint32_t n_prb) { return ra_tbs_from_idx(i_tbs, n_prb); }
// MAC RAR PDU -> up to cap entries of 8 words: t_crnti, rapid, ta, riv, mcs, grant_ok, n_prb, L_prb | tbs in the high half of word 7 is not needed: word 7 = L_prb, word 6 = n_prb; tbs returned via tbs_out
int lsnh_rar_parse(uint32_t nof_prb, const uint8_t* p, int len, uint32_t* out, int* tbs_out, int cap)
{
  Cell c; c.nof_prb = nof_prb; c.nof_ports = 1; c.id = 0;
  RarEntry r[32];
  const int n = rar_parse(c, p, len, r, cap < 32 ? cap : 32);
  for (int i = 0; i < n; i++) {
    uint32_t* o = out + 8 * i;
    o[0] = r[i].t_crnti; o[1] = r[i].rapid; o[2] = r[i].ta; o[3] = r[i].riv; o[4] = r[i].mcs; o[5] = r[i].grant_ok ? 1u : 0u; o[6] = r[i].grant.n_prb; o[7] = r[i].grant.L_prb;
    tbs_out[i] = r[i].grant.tbs;
  }
  return n;
}

// MAC DL-SCH walk: out = n x {lcid, is_sdu, off, len}
int lsnh_mac_dlsch_parse(const uint8_t* pdu, int len, uint32_t* out, int cap)
{
  MacSubheader sub[32];
  const int n = mac_dlsch_parse(pdu, len, sub, cap < 32 ? cap : 32);
  for (int i = 0; i < n; i++) { out[4 * i] = sub[i].lcid; out[4 * i + 1] = sub[i].is_sdu; out[4 * i + 2] = sub[i].off; out[4 * i + 3] = sub[i].len; }
  return n;
}
// RRCConnectionSetup decode: out = {p_a (float bits), i_offset_ack, i_offset_cqi, i_offset_ri, cqi_type}
int lsnh_rrc_conn_setup(const uint8_t* sdu, int len, uint32_t* out)
{
  UeSpecConfig c;
  if (!rrc_conn_setup_decode(sdu, len, c)) return 0;
  std::memcpy(&out[0], &c.p_a, 4);
  out[1] = c.i_offset_ack; out[2] = c.i_offset_cqi; out[3] = c.i_offset_ri; out[4] = c.cqi_type;
  return 1;
}
// SIB2 decode: out[13] = Sib2Config fields in declaration order; returns sib2_decode's verdict (0 / 1 / 2)
int lsnh_sib2_decode(const uint8_t* pdu, int len, uint32_t* out)
{
  Sib2Config c;
  const int r = sib2_decode(pdu, len, c);
  if (r == 2) {
    const uint32_t v[13] = {c.n_sb, c.hopping_mode, c.pusch_hop_offset, c.enable_64qam, c.group_hopping_enabled, c.group_assignment_pusch,
                            c.sequence_hopping_enabled, c.cyclic_shift, c.root_seq_idx, c.prach_config_idx, c.high_speed_flag, c.zero_corr_zone, c.prach_freq_offset};
    std::memcpy(out, v, sizeof(v));
  }
  return r;
}
// paging records: out = n x {is_imsi, nof_digits, mmec, m_tmsi, digits[21] packed one per word} (25 words per record); returns paging_decode's value
int lsnh_paging_decode(const uint8_t* pdu, int len, uint32_t* out, int cap)
{
  PagingId rec[16];
  const int n = paging_decode(pdu, len, rec, cap < 16 ? cap : 16);
  for (int i = 0; i < n; i++) {
    uint32_t* o = out + 25 * i;
    o[0] = rec[i].is_imsi; o[1] = rec[i].nof_digits; o[2] = rec[i].mmec; o[3] = rec[i].m_tmsi;
    for (int k = 0; k < 21; k++) o[4 + k] = rec[i].digits[k];
  }
  return n;
}
// run_api_dl_mode's report for one block: events as {tti, rnti, id_type, msg_type} + value[24] (10 words each); returns nev | to_pcap << 16
int lsnh_api_dl_events(int api_mode, int name, const uint8_t* pdu, int len, uint16_t rnti, uint32_t tti, uint32_t* out, int cap)
{
  ApiEvent ev[20];
  int nev = 0;
  const bool keep = api_dl_events(api_mode, (char)name, pdu, len, rnti, tti, ev, cap < 20 ? cap : 20, &nev);
  for (int i = 0; i < nev; i++) {
    uint32_t* o = out + 10 * i;
    o[0] = ev[i].tti; o[1] = ev[i].rnti; o[2] = ev[i].id_type; o[3] = ev[i].msg_type;
    std::memcpy(o + 4, ev[i].value, 24);
  }
  return nev | (keep ? 1 << 16 : 0);
}
int lsnh_api_ul_dcch_events(int api_mode, const uint8_t* pdu, int len, uint16_t rnti, uint32_t tti, uint32_t* out, int cap)
{
  ApiEvent ev[10];
  int nev = 0;
  const bool keep = api_ul_dcch_events(api_mode, pdu, len, rnti, tti, ev, cap < 10 ? cap : 10, &nev);
  for (int i = 0; i < nev; i++) {
    uint32_t* o = out + 10 * i;
    o[0] = ev[i].tti; o[1] = ev[i].rnti; o[2] = ev[i].id_type; o[3] = ev[i].msg_type;
    std::memcpy(o + 4, ev[i].value, 24);
  }
  return nev | (keep ? 1 << 16 : 0);
}
int lsnh_api_ul_msg3_events(int api_mode, const uint8_t* pdu, int len, uint16_t rnti, uint32_t tti, uint32_t* out, int cap)
{
  ApiEvent ev[10];
  int nev = 0;
  const bool keep = api_ul_msg3_events(api_mode, pdu, len, rnti, tti, ev, cap < 10 ? cap : 10, &nev);
  for (int i = 0; i < nev; i++) {
    uint32_t* o = out + 10 * i;
    o[0] = ev[i].tti; o[1] = ev[i].rnti; o[2] = ev[i].id_type; o[3] = ev[i].msg_type;
    std::memcpy(o + 4, ev[i].value, 24);
  }
  return nev | (keep ? 1 << 16 : 0);
}
// MCSTracking UE-configuration database driven by a sequence of (rnti, pdu) events; returns the configuration get_ue_config_rnti(query) ends with
void* lsnh_mcs_new() { return new MCSTracking(); }
void lsnh_mcs_free(void* m) { delete (MCSTracking*)m; }
int lsnh_mcs_learn(void* m, const uint8_t* pdu, int len, uint16_t rnti) { return ((MCSTracking*)m)->learn_from_pdu(pdu, len, rnti, 0) ? 1 : 0; }
void lsnh_mcs_touch(void* m, uint16_t rnti)
{
  const bool none[2] = {false, false};
  ((MCSTracking*)m)->update_statistic_dl(rnti, FORMAT1, TABLE_FULL_BUFFER, none, none, 0, 0);
}
// database ageing (MCSTracking.cc:758-927) driven event by event; `now` = subframes processed so far
int lsnh_mcs_find(void* m, uint16_t rnti, uint32_t now) { return (int)((MCSTracking*)m)->find_tracking_info_RNTI_dl(rnti, now); }
void lsnh_mcs_update(void* m, uint16_t rnti, int table, uint32_t now) { ((MCSTracking*)m)->update_RNTI_dl(rnti, (McsTable)table, now); }
void lsnh_mcs_rar(void* m, uint16_t rnti, uint32_t now) { ((MCSTracking*)m)->update_rar_time_crnti(rnti, now); }
void lsnh_mcs_stat(void* m, uint16_t rnti, int format, int table, int en0, int en1, int ok0, int ok1, int mimo_ret, uint32_t now)
{
  const bool en[2] = {en0 != 0, en1 != 0}, ok[2] = {ok0 != 0, ok1 != 0};
  ((MCSTracking*)m)->update_statistic_dl(rnti, (DciFormat)format, (McsTable)table, en, ok, mimo_ret, now);
}
void lsnh_mcs_update_database(void* m, uint32_t now) { ((MCSTracking*)m)->update_database_dl(now); }
uint32_t lsnh_mcs_count(void* m) { return ((MCSTracking*)m)->nof_RNTI_member_dl(); }
int lsnh_mcs_peek(void* m, uint16_t rnti) { return ((MCSTracking*)m)->present(rnti) ? (int)((MCSTracking*)m)->peek(rnti) : -1; }
void lsnh_mcs_get(void* m, uint16_t rnti, uint32_t* out)
{
  const UeSpecConfig c = ((MCSTracking*)m)->get_ue_config_rnti(rnti);
  std::memcpy(&out[0], &c.p_a, 4);
  out[1] = c.i_offset_ack; out[2] = c.i_offset_cqi; out[3] = c.i_offset_ri; out[4] = c.cqi_type; out[5] = c.has_ue_config;
}

// tests/lsn_testlib.py: candidate_table() in C (the Python loops are the definition; tests/test_ref_dci_search.py checks that both build the same tables): what
// k_viterbi + k_cce_power produce for one subframe, computed with the ORACLE's candidate decoder and search-space check, handed in as function pointers
typedef uint16_t (*lsnh_decode_fn)(const float* llr, int E, int nof_bits, uint8_t* payload);
typedef uint32_t (*lsnh_validate_fn)(uint32_t nof_cce, uint32_t ncce, uint32_t l, uint32_t nsubframe, uint16_t rnti);
void lsnh_candidate_table(const float* llr, uint32_t nof_cce, const uint32_t* sizes, uint32_t nsizes, uint32_t sf_idx, void* decode_fn, void* validate_fn,
                          LsnCand* cand /* [160 * 8], zeroed here */, float* pw /* [96], zeroed here */)
{
  const lsnh_decode_fn decode = (lsnh_decode_fn)decode_fn;
  const lsnh_validate_fn validate = (lsnh_validate_fn)validate_fn;
  std::memset(cand, 0, sizeof(LsnCand) * 160 * 8);
  std::memset(pw, 0, sizeof(float) * 96);
  for (uint32_t c = 0; c < nof_cce; c++) {
    double m = 0.0;
    for (int i = 0; i < 72; i++) m += std::fabs((double)llr[72 * c + i]);
    pw[c] = (float)(m / 72);
  }
  const uint32_t lim = nof_cce < 84 ? nof_cce : 84;
  uint32_t li = 0;
  uint8_t payload[256];
  for (int l = 3; l >= 0; l--) {
    const uint32_t L = 1u << l;
    for (uint32_t i = 0; i < lim / L; i++, li++) {
      const uint32_t ncce = L * (i % (nof_cce / L)), E = 72 * L;
      bool ok = ncce * 72 + E <= nof_cce * 72;
      for (uint32_t q = 0; ok && q < L; q++) ok = pw[ncce + q] >= 0.7f;
      if (ok) {
        bool any = false;
        for (uint32_t k = 0; k < E && !any; k++) any = llr[ncce * 72 + k] != 0.f;
        ok = any;
      }
      if (!ok) continue;
      for (uint32_t si = 0; si < nsizes; si++) {
        const uint16_t rnti = decode(llr + ncce * 72, (int)E, (int)sizes[si], payload);
        unsigned long long bits = 0;
        for (uint32_t b = 0; b < sizes[si]; b++) bits |= (unsigned long long)payload[b] << (63 - b);
        LsnCand& e = cand[li * 8 + si];
        e.bits = bits; e.rnti = rnti; e.flags = 1u | (validate(nof_cce, ncce, (uint32_t)l, sf_idx, rnti) << 1);
      }
    }
  }
}

// the product's downlink HARQ database on its own (tests/test_ref_harq.py: random grant sequences against the REFERENCE's HARQ.cc, oracle/_ref)
void* lsnh_harq_new() { return new HarqDatabase(); }
void lsnh_harq_free(void* h) { delete (HarqDatabase*)h; }
int lsnh_harq_is_retransmission(void* h, uint16_t rnti, uint32_t pid, int tid, int ndi, int tbs, uint32_t sfn, uint32_t sf_idx, int* entity)
{
  return (int)((HarqDatabase*)h)->is_retransmission(rnti, pid, tid, ndi != 0, tbs, sfn, sf_idx, *entity);
}
void lsnh_harq_update(void* h, int entity, uint32_t pid, int tid, uint32_t sfn, uint32_t sf_idx, int decoded, int ndi, int rv, int tbs, uint32_t now)
{
  ((HarqDatabase*)h)->update(entity, pid, tid, sfn, sf_idx, decoded != 0, ndi != 0, rv, tbs, now);
}
void lsnh_harq_update_database(void* h, uint32_t now) { ((HarqDatabase*)h)->update_database(now); }

// one random-access response with the given 20-bit grant through rar_parse (falcon_dci.c:636-683 + DL_Sniffer_PDSCH.cc:632-671): out = hopping, riv, mcs, tpc, ul_delay,
// csi_req, grant_ok, then the 9 grant words of put_ul_grant
int lsnh_rar_grant(uint32_t nof_prb, uint32_t cp, uint32_t n_rb_ho, uint32_t grant20, uint32_t* out)
{
  Cell c; c.nof_prb = nof_prb; c.nof_ports = 1; c.cp = cp; c.pusch_hop_offset = n_rb_ho;
  const uint8_t pdu[7] = {0x41, 0x00, (uint8_t)(0x10 | ((grant20 >> 16) & 0xF)), (uint8_t)(grant20 >> 8), (uint8_t)grant20, 0x12, 0x34};
  RarEntry r[2];
  if (rar_parse(c, pdu, 7, r, 2) != 1) return -1;
  out[0] = r[0].hopping; out[1] = r[0].riv; out[2] = r[0].mcs; out[3] = r[0].tpc; out[4] = r[0].ul_delay; out[5] = r[0].csi_req; out[6] = r[0].grant_ok ? 1 : 0;
  std::memset(out + 7, 0, 9 * sizeof(uint32_t));
  if (r[0].grant_ok) {
    const PuschGrant& g = r[0].grant;
    out[7] = g.L_prb; out[8] = g.n_prb; out[9] = g.hop == 1 ? g.n_prb2 : g.n_prb; out[10] = g.hop; out[11] = g.L_prb ? (uint32_t)g.mod : 0; out[12] = (uint32_t)g.tbs;
    out[13] = (uint32_t)g.rv; out[14] = g.mcs_idx; out[15] = g.L_prb * 12u * 2u * (c.nslot() - 1);
  }
  return r[0].t_crnti == 0x1234 && r[0].rapid == 1 && r[0].ta == 1 ? 0 : -2;
}

// ---- DCICollection::addCandidate as the product does it (tests/test_ref_collect.py: the REFERENCE's DCICollection.cc / falcon_dci.c, oracle/_ref) ----
// FalconSearch::finishSubframe (unpack, both grant conversions, collision statistics) + the commit-side helpers of lsn_search.h the engine itself uses
// (collection_table, table_view, collection_last_tbs).  Rows in the flat layout of oracle/ref_shim_search/collect_glue.cc.
struct hcollect { Cell cell; std::unique_ptr<FalconSearch> s; MCSTracking mcs; HarqDatabase harq; int mode = 1, harq_mode = 0; uint32_t now = 0; };
static void put_mask(uint32_t* o, const bool* prb, uint32_t n)
{
  o[0] = o[1] = o[2] = o[3] = 0;
  for (uint32_t i = 0; i < n && i < 128; i++)
    if (prb[i]) o[i >> 5] |= 1u << (i & 31);
}
static void put_dl_grant(uint32_t* o, const PdschGrant& g, uint32_t nof_prb)
{
  o[0] = g.nof_prb; o[1] = g.nof_re; o[2] = g.nof_tb;
  put_mask(o + 3, g.prb_idx[0], nof_prb);
  put_mask(o + 7, g.prb_idx[1], nof_prb);
  for (int i = 0; i < 2; i++) {
    uint32_t* t = o + 11 + 7 * i;
    t[0] = g.tb[i].enabled; t[1] = g.tb[i].enabled ? (uint32_t)g.tb[i].mod : 0; t[2] = (uint32_t)g.tb[i].tbs; t[3] = (uint32_t)g.tb[i].nof_bits;
    t[4] = (uint32_t)g.tb[i].rv; t[5] = g.tb[i].mcs_idx; t[6] = g.tb[i].cw_idx;
  }
}
static void put_ul_grant(uint32_t* o, const PuschGrant& g, const Cell& cell)
{
  o[0] = g.L_prb; o[1] = g.n_prb; o[2] = g.hop == 1 ? g.n_prb2 : g.n_prb; o[3] = g.hop; o[4] = g.L_prb ? (uint32_t)g.mod : 0; o[5] = (uint32_t)g.tbs;
  o[6] = (uint32_t)g.rv; o[7] = g.mcs_idx; o[8] = g.L_prb * 12u * 2u * (cell.nslot() - 1);
}
void* lsnh_collect_new(uint32_t nof_prb, uint32_t nof_ports, uint32_t cell_id, uint32_t cp, int mcs_tracking_mode, int harq_mode)
{
  hcollect* h = new hcollect();
  h->cell.nof_prb = nof_prb; h->cell.nof_ports = nof_ports; h->cell.id = cell_id; h->cell.cp = cp;
  h->mode = mcs_tracking_mode; h->harq_mode = harq_mode;
  h->s.reset(new FalconSearch(5, 0.99, false));
  const uint32_t ncce[3] = {20, 54, 87};
  h->s->setCell(h->cell, ncce);
  return h;
}
void lsnh_collect_free(void* p) { delete (hcollect*)p; }
void lsnh_collect_set_hop_offset(void* p, uint32_t n_rb_ho) { ((hcollect*)p)->s->setPuschHopOffset(n_rb_ho); ((hcollect*)p)->cell.pusch_hop_offset = n_rb_ho; }
void lsnh_collect_set_now(void* p, uint32_t now) { ((hcollect*)p)->now = now; }
void lsnh_collect_mcs_update(void* p, uint16_t rnti, int table) { hcollect* h = (hcollect*)p; h->mcs.update_RNTI_dl(rnti, (McsTable)table, h->now); }
void lsnh_collect_harq_update(void* p, uint16_t rnti, uint32_t pid, int tid, uint32_t sfn, uint32_t sf_idx, int decoded, int ndi, int rv, int tbs)
{
  hcollect* h = (hcollect*)p;
  int ent = -1;
  const HarqRet hr = h->harq.is_retransmission(rnti, pid, tid, ndi != 0, tbs, sfn, sf_idx, ent);
  if (hr == HARQ_NEW_TX || hr == HARQ_RE_TX) h->harq.update(ent, pid, tid, sfn, sf_idx, decoded != 0, ndi != 0, rv, tbs, h->now);
}
// n accepted DCI: (rnti, format, L, ncce, histval, nof_bits) x n in meta6, payload bits (one byte per bit, 128 per DCI) in bits.  Returns the collision flags.
uint32_t lsnh_collect_subframe(void* p, uint32_t sfn, uint32_t sf_idx, uint32_t cfi, uint32_t n, const uint32_t* meta6, const uint8_t* bits, uint32_t* dl, uint32_t* ul,
                               uint32_t* counts2)
{
  hcollect* h = (hcollect*)p;
  SubframeCtx c;
  c.reset(sfn * 10 + sf_idx);
  c.cfi = cfi; c.searched = true;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t* m = meta6 + 6 * i;
    unsigned long long w = 0;
    for (uint32_t b = 0; b < m[5] && b < 64; b++) w |= (unsigned long long)(bits[128 * i + b] & 1) << (63 - b);
    c.raw.push_back(AcceptedDci{(uint16_t)m[0], (uint8_t)m[1], (uint8_t)m[2], (uint16_t)m[3], (uint16_t)m[5], m[4], w});
  }
  const BlindStats b0 = h->s->getStats();
  h->s->finishSubframe(c);
  const BlindStats b1 = h->s->getStats();
  // the engine's commit turn (lsn_engine.cc: commitChunk): tables of every DCI first, then the uplink look-ups, then entry by entry
  std::vector<McsTable> tables;
  for (const DlEntry& e : c.dl) tables.push_back(collection_table(h->mode, e.rnti, e.format, h->mcs, h->now));
  if (h->mode == 1)
    for (const UlEntry& u : c.ul)
      if (!(u.rnti == SIRNTI || u.rnti == PRNTI || rnti_israr(u.rnti))) (void)h->mcs.find_tracking_info_RNTI_dl(u.rnti, h->now);
  for (size_t i = 0; i < c.dl.size(); i++) {
    DlEntry& e = c.dl[i];
    uint32_t* o = dl + i * 64;
    std::memset(o, 0, 64 * sizeof(uint32_t));
    const TableView tv = table_view(tables[i], e.rnti, e.unpack_ok, e.ok64, e.ok256);
    const bool check = collection_last_tbs(h->harq_mode != 0, tables[i], e, h->harq);
    o[0] = e.rnti; o[1] = (uint32_t)e.format; o[2] = (uint32_t)tables[i]; o[3] = tv.dci_rnti_ok ? e.rnti : 0; o[4] = e.dci.pid; o[5] = e.dci.pinfo; o[6] = e.dci.tb_cw_swap;
    for (int t = 0; t < 2; t++) { o[7 + 3 * t] = e.dci.tb[t].mcs_idx; o[8 + 3 * t] = (uint32_t)e.dci.tb[t].rv; o[9 + 3 * t] = e.dci.tb[t].ndi; }
    o[13] = check ? 1 : 0;
    if (tv.has64) put_dl_grant(o + 14, e.grant64, h->cell.nof_prb);
    if (tv.has256) put_dl_grant(o + 39, e.grant256, h->cell.nof_prb);
  }
  for (size_t i = 0; i < c.ul.size(); i++) {
    const UlEntry& u = c.ul[i];
    uint32_t* o = ul + i * 32;
    std::memset(o, 0, 32 * sizeof(uint32_t));
    o[0] = u.rnti; o[1] = u.ok ? u.rnti : 0; o[2] = u.dci.n_dmrs; o[3] = u.dci.cqi_req; o[4] = u.dci.ndi; o[5] = u.dci.tpc; o[6] = (uint32_t)u.dci.hop_type; o[7] = u.dci.riv;
    o[8] = u.dci.mcs_idx; o[9] = 0;
    if (u.ok) { put_ul_grant(o + 10, u.grant, h->cell); put_ul_grant(o + 19, u.grant256, h->cell); o[28] = u.grant.L_prb; o[29] = u.grant.n_prb; }
  }
  counts2[0] = (uint32_t)c.dl.size(); counts2[1] = (uint32_t)c.ul.size();
  return (b1.nof_subframe_collisions_dw != b0.nof_subframe_collisions_dw ? 1u : 0u) | (b1.nof_subframe_collisions_up != b0.nof_subframe_collisions_up ? 2u : 0u);
}

// the product's RNTIManager on its own (tests/test_ref_rnti_manager.py: operation sequences against the REFERENCE's RNTIManager.cc, oracle/_ref)
void* lsnh_rm_new(uint32_t nformats, uint32_t maxcand, uint32_t threshold) { return new RNTIManager(nformats, maxcand, threshold); }
void lsnh_rm_free(void* h) { delete (RNTIManager*)h; }
void lsnh_rm_add_evergreen(void* h, uint16_t a, uint16_t b, uint32_t f) { ((RNTIManager*)h)->addEvergreen(a, b, f); }
void lsnh_rm_add_forbidden(void* h, uint16_t a, uint16_t b, uint32_t f) { ((RNTIManager*)h)->addForbidden(a, b, f); }
void lsnh_rm_add_candidate(void* h, uint16_t rnti, uint32_t f) { ((RNTIManager*)h)->addCandidate(rnti, f); }
int lsnh_rm_validate(void* h, uint16_t rnti, uint32_t f) { return ((RNTIManager*)h)->validate(rnti, f) ? 1 : 0; }
int lsnh_rm_validate_and_refresh(void* h, uint16_t rnti, uint32_t f) { return ((RNTIManager*)h)->validateAndRefresh(rnti, f) ? 1 : 0; }
void lsnh_rm_activate_and_refresh(void* h, uint16_t rnti, uint32_t f, int reason) { ((RNTIManager*)h)->activateAndRefresh(rnti, f, (ActivationReason)reason); }
int lsnh_rm_is_evergreen(void* h, uint16_t rnti, uint32_t f) { return ((RNTIManager*)h)->isEvergreen(rnti, f) ? 1 : 0; }
int lsnh_rm_is_forbidden(void* h, uint16_t rnti, uint32_t f) { return ((RNTIManager*)h)->isForbidden(rnti, f) ? 1 : 0; }
void lsnh_rm_step_time(void* h, uint32_t n) { while (n--) ((RNTIManager*)h)->stepTime(); }
uint32_t lsnh_rm_get_frequency(void* h, uint16_t rnti, uint32_t f) { return ((RNTIManager*)h)->getFrequency(rnti, f); }
int lsnh_rm_get_activation_reason(void* h, uint16_t rnti) { return (int)((RNTIManager*)h)->getActivationReason(rnti); }
void lsnh_rm_set_threshold(void* h, uint32_t t) { ((RNTIManager*)h)->setHistogramThreshold(t); }
uint32_t lsnh_rm_nof_active(void* h) { return ((RNTIManager*)h)->nofActive(); }

// PUSCH_Decoder::decode's trial order and grant test (lsn_lte.cc: ulTrialPlan, ulGrantValid) for tests/test_ref_ul_decode.py: out = n x {use256, qm, learn}
int lsn_host_ul_trial(uint32_t mcs_idx, int grant_mod_bits, uint32_t L_prb_256, int mod_bits_256, int tracked, int32_t* out9)
{
  lsn::UlTry t[3];
  const int n = lsn::ulTrialPlan(mcs_idx, grant_mod_bits, L_prb_256, mod_bits_256, tracked, t);
  for (int i = 0; i < n; i++) { out9[3 * i] = t[i].use256; out9[3 * i + 1] = t[i].qm; out9[3 * i + 2] = t[i].learn; }
  return n;
}
int lsn_host_ul_grant_valid(uint32_t rnti, int is_rar, int tbs, int tbs_256, uint32_t L_prb) { return lsn::ulGrantValid((uint16_t)rnti, is_rar != 0, tbs, tbs_256, L_prb) ? 1 : 0; }
}  // extern "C"
