/* o_rrc.c - ORACLE (test infrastructure only): what the decode loop learns from a decoded C-RNTI transport block.
 * Restates (a) the MAC DL-SCH PDU walk the reference does with srsran::sch_pdu::parse_packet / next / get
 * (/root/reference/src/src/DL_Sniffer_PDSCH.cc:1041-1070, 1133-1160, 279-310) [srsRAN lib/src/mac/pdu.cc, not in tree;
 * TS 36.321 6.1.2 / 6.2.1] and (b) PDSCH_Decoder::decode_rrc_connection_setup (DL_Sniffer_PDSCH.cc:129-181), which unpacks a
 * DL-CCCH-Message with srsRAN's ASN.1 code [not in tree] and keeps four things of an RRCConnectionSetup: p-a, the three
 * betaOffset indices and the aperiodic CQI report mode.  The UPER walk below follows TS 36.331 6.2.2 / 6.3.2 (Rel-8 root
 * components; extension additions of any release are skipped through their length determinants, X.691 10.9 / 18).
 * Pinned by the five RRCConnectionSetup messages in the reference's own capture (pcap_file_example/api_collector.pcap,
 * committed as tests/golden/msg4_conn_setup.json): every field decodes to a legal value, the walk ends on the last octet.
 * Not restated: DRB / SPS components (never part of a connection setup: SRB1 only) - such a message is reported as
 * "not a connection setup", where srsRAN would decode it. */
#include "lsn_oracle.h"
#include <string.h>

/* ---------------- MAC DL-SCH PDU ---------------- */
static int dl_ce_size(uint32_t lcid) /* sch_subh::sizeof_ce, downlink */
{
  switch (lcid) {
    case 28: return 6; /* UE contention resolution identity */
    case 29: return 1; /* timing advance command */
    case 27: return 1; /* SCell activation */
    default: return 0; /* DRX command, padding, reserved */
  }
}

/* returns the number of subheaders, 0 when the PDU does not parse (sch_pdu::parse_packet fails -> next() yields nothing) */
int o_mac_dlsch_parse(const uint8_t* pdu, int len, o_mac_subh_t* out, int cap)
{
  int pos = 0, n = 0, more = 1;
  if (len <= 0) return 0;
  while (more && n < cap && pos < len) {
    const uint8_t b = pdu[pos++];
    o_mac_subh_t* s = &out[n++];
    s->lcid = b & 0x1Fu;
    s->is_sdu = s->lcid < 26; /* sch_subh::is_sdu: everything below the control-element LCIDs */
    s->len = 0;
    more = (b >> 5) & 1;
    if (s->is_sdu && more) { /* F / L: only when another subheader follows */
      if (pos >= len) return 0;
      const uint8_t l = pdu[pos++];
      s->len = l & 0x7Fu;
      if (l & 0x80u) {
        if (pos >= len) return 0;
        s->len = (s->len << 8) | pdu[pos++];
      }
    }
    if (more && pos >= len) return 0; /* a further subheader is announced but the PDU is used up */
  }
  if (more && n == cap) return 0;
  for (int i = 0; i < n; i++) {
    o_mac_subh_t* s = &out[i];
    if (!s->is_sdu) s->len = (uint32_t)dl_ce_size(s->lcid);
    s->off = (uint32_t)pos;
    if (i == n - 1 && s->is_sdu) s->len = (uint32_t)(len - pos); /* the last SDU takes what is left */
    pos += (int)s->len;
    if (pos > len) return 0;
  }
  return n;
}

/* ---------------- UPER bit reader ---------------- */
typedef struct { const uint8_t* p; uint32_t nbits, pos; int err; } br_t;
static uint32_t rd(br_t* b, uint32_t n)
{
  uint32_t v = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (b->pos >= b->nbits) { b->err = 1; return 0; }
    v = (v << 1) | ((b->p[b->pos >> 3] >> (7 - (b->pos & 7))) & 1u);
    b->pos++;
  }
  return v;
}
/* constrained whole number lb..ub in the minimum number of bits; a value above ub is a decode error */
static uint32_t rd_int(br_t* b, uint32_t lb, uint32_t ub)
{
  uint32_t range = ub - lb + 1, nb = 0;
  while ((1u << nb) < range) nb++;
  uint32_t v = rd(b, nb);
  if (v >= range) { b->err = 1; return lb; }
  return lb + v;
}
/* general length determinant (X.691 10.9.3.5-7, unaligned) */
static uint32_t rd_len(br_t* b)
{
  if (!rd(b, 1)) return rd(b, 7);
  if (!rd(b, 1)) return rd(b, 14);
  b->err = 1; /* fragmented lengths do not occur in these messages */
  return 0;
}
/* extension additions of a SEQUENCE whose extension bit was set (X.691 18.7-9): a "normally small" count, the presence
 * bitmap, then one open type (length in octets + content) per addition present */
static void skip_ext(br_t* b)
{
  uint32_t n;
  if (!rd(b, 1)) n = rd(b, 6) + 1; else n = rd_len(b);
  uint32_t present = 0;
  for (uint32_t i = 0; i < n && !b->err; i++) present += rd(b, 1);
  for (uint32_t i = 0; i < present && !b->err; i++) {
    const uint32_t octets = rd_len(b);
    if (b->pos + 8u * octets > b->nbits) { b->err = 1; return; }
    b->pos += 8u * octets;
  }
}

static void rlc_config(br_t* b)
{
  if (rd(b, 1)) { b->err = 1; return; } /* a choice outside the root */
  switch (rd(b, 2)) {
    case 0: rd(b, 6); rd(b, 3); rd(b, 4); rd(b, 3); rd(b, 5); rd(b, 6); break; /* am: ul-AM-RLC, dl-AM-RLC */
    case 1: rd(b, 1); rd(b, 1); rd(b, 5); break;                               /* um-Bi-Directional */
    case 2: rd(b, 1); break;                                                   /* um-Uni-Directional-UL */
    default: rd(b, 1); rd(b, 5); break;                                        /* um-Uni-Directional-DL */
  }
}
static void logical_channel_config(br_t* b)
{
  const uint32_t ext = rd(b, 1), ul = rd(b, 1);
  if (ul) {
    const uint32_t grp = rd(b, 1);
    rd(b, 4); rd(b, 4); rd(b, 3);
    if (grp) rd(b, 2);
  }
  if (ext) skip_ext(b);
}
static void mac_main_config(br_t* b)
{
  const uint32_t ext = rd(b, 1), ulsch = rd(b, 1), drx = rd(b, 1), phr = rd(b, 1);
  if (ulsch) {
    const uint32_t harq = rd(b, 1), bsr = rd(b, 1);
    if (harq) rd(b, 4);
    if (bsr) rd(b, 4);
    rd(b, 3); rd(b, 1);
  }
  if (drx && rd(b, 1)) { /* setup */
    static const uint16_t cyc[16] = {10, 20, 32, 40, 64, 80, 128, 160, 256, 320, 512, 640, 1024, 1280, 2048, 2560};
    const uint32_t shortdrx = rd(b, 1);
    rd(b, 4); rd(b, 5); rd(b, 3);
    rd_int(b, 0, cyc[rd(b, 4)] - 1u);
    if (shortdrx) { rd(b, 4); rd(b, 4); }
  }
  rd(b, 3); /* timeAlignmentTimerDedicated */
  if (phr && rd(b, 1)) { rd(b, 3); rd(b, 3); rd(b, 2); }
  if (ext) skip_ext(b);
}
static void tpc_pdcch_config(br_t* b)
{
  if (!rd(b, 1)) return; /* release */
  rd(b, 16);
  if (rd(b, 1)) rd_int(b, 1, 31); else rd_int(b, 1, 15);
}

/* 1: an RRCConnectionSetup, *out filled; 0: anything else (decode_rrc_connection_setup returns SRSRAN_ERROR) */
int o_rrc_conn_setup_decode(const uint8_t* sdu, int len, o_ue_cfg_t* out)
{
  static const float p_a_db[8] = {-6.0f, -4.77f, -3.0f, -1.77f, 0.0f, 1.0f, 2.0f, 3.0f}; /* DL_Sniffer_PDSCH.cc:3 */
  br_t br = {sdu, len > 0 ? 8u * (uint32_t)len : 0u, 0, 0};
  br_t* b = &br;
  memset(out, 0, sizeof *out); /* ltesniffer_ue_spec_config_t ue_config = {}: p_a 0 dB, offsets 0, CQI type 0 = wideband */
  if (rd(b, 1)) return 0;      /* DL-CCCH-MessageType: c1 */
  if (rd(b, 2) != 3) return 0; /* c1: reestablishment, reestablishmentReject, reject, setup */
  rd(b, 2);                    /* rrc-TransactionIdentifier */
  if (rd(b, 1)) return 0;      /* criticalExtensions: c1 */
  if (rd(b, 3) != 0) return 0; /* c1: rrcConnectionSetup-r8 */
  const uint32_t noncrit = rd(b, 1);
  /* RadioResourceConfigDedicated */
  const uint32_t rr_ext = rd(b, 1), srb = rd(b, 1), drb_add = rd(b, 1), drb_rel = rd(b, 1), mac = rd(b, 1), sps = rd(b, 1), phy = rd(b, 1);
  if (drb_add || sps) return 0; /* see the header: not part of a connection setup */
  if (srb) {
    const uint32_t n = rd(b, 1) + 1;
    for (uint32_t i = 0; i < n && !b->err; i++) {
      const uint32_t ext = rd(b, 1), rlc = rd(b, 1), lc = rd(b, 1);
      rd(b, 1); /* srb-Identity */
      if (rlc && !rd(b, 1)) rlc_config(b);
      if (lc && !rd(b, 1)) logical_channel_config(b);
      if (ext) skip_ext(b);
    }
  }
  if (drb_rel) {
    const uint32_t n = rd_int(b, 1, 11);
    for (uint32_t i = 0; i < n; i++) rd(b, 5);
  }
  if (mac && !rd(b, 1)) mac_main_config(b);
  if (phy) {
    const uint32_t ext = rd(b, 1);
    uint32_t f[10];
    for (int i = 0; i < 10; i++) f[i] = rd(b, 1);
    if (f[0]) out->p_a = p_a_db[rd(b, 3)]; /* pdsch-ConfigDedicated */
    if (f[1]) {                            /* pucch-ConfigDedicated */
      const uint32_t tdd = rd(b, 1);
      if (rd(b, 1)) { rd(b, 2); rd(b, 11); }
      if (tdd) rd(b, 1);
    }
    if (f[2]) { /* pusch-ConfigDedicated: betaOffset-ACK-Index, betaOffset-RI-Index, betaOffset-CQI-Index */
      out->i_offset_ack = rd(b, 4);
      out->i_offset_ri = rd(b, 4);
      out->i_offset_cqi = rd(b, 4);
    }
    if (f[3]) { /* uplinkPowerControlDedicated */
      const uint32_t fc = rd(b, 1);
      rd(b, 4); rd(b, 1); rd(b, 1); rd(b, 4); rd(b, 4);
      if (fc) { if (rd(b, 1)) b->err = 1; rd(b, 4); }
    }
    if (f[4]) tpc_pdcch_config(b);
    if (f[5]) tpc_pdcch_config(b);
    if (f[6]) { /* cqi-ReportConfig */
      const uint32_t aper = rd(b, 1), per = rd(b, 1);
      if (aper) {
        const uint32_t m = rd(b, 3); /* rm12, rm20, rm22, rm30, rm31, spare */
        if (m == 0) out->cqi_type = 0;                 /* SRSRAN_CQI_TYPE_WIDEBAND */
        else if (m == 1 || m == 2) out->cqi_type = 1;  /* SUBBAND_UE */
        else if (m == 3 || m == 4) out->cqi_type = 2;  /* SUBBAND_HL */
      }
      rd(b, 3); /* nomPDSCH-RS-EPRE-Offset */
      if (per && rd(b, 1)) {
        const uint32_t ri = rd(b, 1);
        rd_int(b, 0, 1185);
        rd(b, 10);
        if (rd(b, 1)) rd(b, 2);
        if (ri) rd(b, 10);
        rd(b, 1);
      }
    }
    if (f[7] && rd(b, 1)) { rd(b, 2); rd(b, 2); rd_int(b, 0, 23); rd(b, 1); rd(b, 10); rd(b, 1); rd(b, 3); } /* soundingRS-UL-ConfigDedicated */
    if (f[8] && !rd(b, 1)) { /* antennaInfo: explicitValue */
      static const uint8_t cbsr_bits[8] = {2, 4, 6, 64, 4, 16, 4, 16};
      const uint32_t cb = rd(b, 1);
      rd(b, 3);
      if (cb) {
        uint32_t nb = cbsr_bits[rd(b, 3)];
        while (nb) { const uint32_t k = nb > 16 ? 16 : nb; rd(b, k); nb -= k; }
      }
      if (rd(b, 1)) rd(b, 1);
    }
    if (f[9] && rd(b, 1)) { rd(b, 11); rd_int(b, 0, 157); rd(b, 3); } /* schedulingRequestConfig */
    if (ext) skip_ext(b);
  }
  if (rr_ext) skip_ext(b);
  if (noncrit) { /* RRCConnectionSetup-v8a0-IEs: lateNonCriticalExtension OCTET STRING, empty nonCriticalExtension */
    const uint32_t late = rd(b, 1), more = rd(b, 1);
    if (late) { const uint32_t octets = rd_len(b); if (b->pos + 8u * octets > b->nbits) b->err = 1; else b->pos += 8u * octets; }
    (void)more;
  }
  if (b->err) return 0;
  out->has_ue_config = 1;
  out->bits_used = b->pos;
  return 1;
}

/* ---------------- BCCH-DL-SCH-Message: SystemInformationBlockType2 ----------------
 * PDSCH_Decoder::decode_SIB (/root/reference/src/src/DL_Sniffer_PDSCH.cc:531-557) unpacks a BCCH-DL-SCH-Message with srsRAN's
 * ASN.1 code [not in tree] and keeps the sib2 entry of a SystemInformation; ULSchedule::set_config (ULSchedule.cc:140-158) and
 * SubframeWorker.cc:271-273 read the PUSCH hopping / reference-signal and PRACH fields out of it.  UPER walk after TS 36.331
 * 6.2.1 / 6.2.2 / 6.3.1 / 6.3.2.  Pinned by the SIB2 message the reference itself wrote into pcap_file_example/ltesniffer_ul_mode.pcap
 * (tests/golden/pcap_records.json "si_pdus"): the walk ends inside the message and yields srsENB's stock sib.conf values.
 * Returns 0: does not unpack, 1: unpacks, no SIB2 in front (SIB1 or another block), 2: SIB2 (out filled).
 * Not restated: blocks BEHIND SIB2 in the same SystemInformation are not unpacked. */
int o_sib2_decode(const uint8_t* pdu, int len, o_sib2_t* out)
{
  br_t b = {pdu, len > 0 ? 8u * (uint32_t)len : 0u, 0, 0};
  o_sib2_t o;
  memset(&o, 0, sizeof(o));
  if (rd(&b, 1)) return 0;                /* messageClassExtension */
  if (rd(&b, 1)) return b.err ? 0 : 1;    /* systemInformationBlockType1 */
  if (rd(&b, 1)) return 0;                /* criticalExtensionsFuture */
  rd(&b, 1);                              /* nonCriticalExtension present */
  rd(&b, 5);                              /* number of blocks - 1 */
  if (rd(&b, 1)) return 0;                /* extension alternative of the block CHOICE */
  uint32_t alt = rd(&b, 4);
  if (b.err || alt > 9) return 0;
  if (alt != 0) return 1;
  int sib_ext = (int)rd(&b, 1), ac = (int)rd(&b, 1), mbsfn = (int)rd(&b, 1);
  if (ac) {
    int sig = (int)rd(&b, 1), dat = (int)rd(&b, 1);
    rd(&b, 1);
    if (sig) rd(&b, 12);
    if (dat) rd(&b, 12);
  }
  int rr_ext = (int)rd(&b, 1);
  { /* RACH-ConfigCommon */
    int ext = (int)rd(&b, 1), grp = (int)rd(&b, 1);
    rd(&b, 4);
    if (grp) {
      int gext = (int)rd(&b, 1);
      rd_int(&b, 0, 14); rd(&b, 2); rd(&b, 3);
      if (gext) skip_ext(&b);
    }
    rd(&b, 2); rd(&b, 4);
    rd_int(&b, 0, 10); rd(&b, 3); rd(&b, 3);
    rd(&b, 3);
    if (ext) skip_ext(&b);
  }
  rd(&b, 2);          /* BCCH-Config */
  rd(&b, 5);          /* PCCH-Config */
  o.root_seq_idx = rd_int(&b, 0, 837);
  o.prach_config_idx = rd(&b, 6);
  o.high_speed_flag = rd(&b, 1);
  o.zero_corr_zone = rd(&b, 4);
  o.prach_freq_offset = rd_int(&b, 0, 94);
  rd_int(&b, 0, 110); rd(&b, 2); /* PDSCH-ConfigCommon */
  o.n_sb = rd(&b, 2) + 1;
  o.hopping_mode = rd(&b, 1);
  o.pusch_hop_offset = rd_int(&b, 0, 98);
  o.enable_64qam = rd(&b, 1);
  o.group_hopping_enabled = rd(&b, 1);
  o.group_assignment_pusch = rd_int(&b, 0, 29);
  o.sequence_hopping_enabled = rd(&b, 1);
  o.cyclic_shift = rd(&b, 3);
  rd_int(&b, 0, 2); rd_int(&b, 0, 98); rd(&b, 3); rd(&b, 11); /* PUCCH-ConfigCommon */
  if (rd(&b, 1)) { rd(&b, 1); rd(&b, 8); }                     /* SoundingRS-UL-ConfigCommon setup */
  rd_int(&b, 0, 150); rd(&b, 3); rd(&b, 5);                    /* UplinkPowerControlCommon */
  rd_int(&b, 0, 2); rd_int(&b, 0, 2); rd(&b, 2); rd_int(&b, 0, 2); rd_int(&b, 0, 2);
  rd(&b, 3);
  rd(&b, 1);                                                   /* UL-CyclicPrefixLength */
  if (rr_ext) skip_ext(&b);
  { /* UE-TimersAndConstants */
    int ext = (int)rd(&b, 1);
    rd(&b, 3); rd(&b, 3); rd_int(&b, 0, 6); rd(&b, 3); rd_int(&b, 0, 6); rd(&b, 3);
    if (ext) skip_ext(&b);
  }
  { /* freqInfo */
    int carrier = (int)rd(&b, 1), bw = (int)rd(&b, 1);
    if (carrier) rd(&b, 16);
    if (bw) rd_int(&b, 0, 5);
    rd(&b, 5);
  }
  if (mbsfn) {
    uint32_t n = rd(&b, 3) + 1;
    for (uint32_t i = 0; i < n && !b.err; i++) { rd_int(&b, 0, 5); rd(&b, 3); if (rd(&b, 1)) rd(&b, 24); else rd(&b, 6); }
  }
  rd(&b, 3); /* TimeAlignmentTimer */
  if (sib_ext) skip_ext(&b);
  if (b.err) return 0;
  o.bits_used = b.pos;
  *out = o;
  return 2;
}

/* ---------------- security-API view of a decoded downlink block ----------------
 * PDSCH_Decoder::run_api_dl_mode (DL_Sniffer_PDSCH.cc:804-879) and decode_imsi_tmsi_paging (:84-127): paging records (IMSI / S-TMSI) and
 * the contention resolution identity next to an RRCConnectionSetup.  PCCH-Message per TS 36.331 6.2.1 / 6.2.2.  parity unpinned: the
 * reference's captures hold no paging record; the connection-setup side is pinned by api_collector.pcap (one API record per setup).
 * Not restated: RRCConnectionReconfiguration / NAS identities (LCID 1), the uplink API parsers. */
int o_paging_decode(const uint8_t* pdu, int len, o_paging_id_t* out, int cap)
{
  br_t b = {pdu, len > 0 ? 8u * (uint32_t)len : 0u, 0, 0};
  if (rd(&b, 1)) return -1;
  int list = (int)rd(&b, 1);
  rd(&b, 3);
  if (b.err) return -1;
  int n = 0;
  if (list) {
    uint32_t cnt = rd(&b, 4) + 1;
    for (uint32_t i = 0; i < cnt && !b.err; i++) {
      int ext = (int)rd(&b, 1);
      o_paging_id_t id;
      memset(&id, 0, sizeof(id));
      if (rd(&b, 1)) return -1;
      if (!rd(&b, 1)) {
        id.mmec = rd(&b, 8);
        id.m_tmsi = rd(&b, 32);
      } else {
        id.is_imsi = 1;
        id.nof_digits = rd(&b, 4) + 6;
        if (id.nof_digits > 21) return -1;
        for (uint32_t k = 0; k < id.nof_digits; k++) { id.digits[k] = (uint8_t)rd(&b, 4); if (id.digits[k] > 9) b.err = 1; }
      }
      rd(&b, 1);
      if (ext) skip_ext(&b);
      if (b.err) return -1;
      if (n < cap) out[n++] = id;
    }
  }
  return b.err ? -1 : n;
}

#include <stdio.h>
/* ---------------- RRCConnectionReconfiguration -> attach accept -> GUTI (decode_rrc_connection_reconfig, DL_Sniffer_PDSCH.cc:181-220) ----------------
 * The reference unpacks the DL-DCCH message with srsRAN's ASN.1 code and the first NAS container with liblte_mme [both not in tree]: TS 36.331
 * 6.2.2 (RRCConnectionReconfiguration-r8-IEs), 6.3.5 (MeasConfig), TS 24.301 8.2.1 (attach accept), 9.9.3.12 (EPS mobile identity).  This
 * restatement keeps a cursor over the bit string and consumes the components in front of dedicatedInfoNASList one by one. */
typedef struct { const uint8_t* d; uint32_t n, at; int bad; } cur_t;
static uint32_t take(cur_t* c, uint32_t k)
{
  uint32_t v = 0;
  while (k--) {
    if (c->at >= c->n) { c->bad = 1; return 0; }
    v = (v << 1) | ((c->d[c->at >> 3] >> (7u - (c->at & 7u))) & 1u);
    c->at++;
  }
  return v;
}
static void drop_list5(cur_t* c, uint32_t item_bits) /* SEQUENCE (SIZE (1..32)) OF fixed-width items */
{
  uint32_t cnt = take(c, 5) + 1;
  for (uint32_t i = 0; i < cnt; i++) take(c, item_bits);
}
static void drop_threshold(cur_t* c) { take(c, take(c, 1) ? 6 : 7); }
/* an open type (X.691 10.2): octet count in one or two octets, then the octets themselves, none of which is looked at */
static int drop_open(cur_t* c)
{
  uint32_t octets = take(c, 8);
  if (octets >= 0xC0u) return 0;
  if (octets >= 0x80u) octets = ((octets - 0x80u) << 8) + take(c, 8);
  for (uint32_t i = 0; i < octets; i++) take(c, 8);
  return !c->bad;
}
/* what follows the root of an extensible SEQUENCE whose extension bit was set (X.691 19.7-19.9) */
static int drop_additions(cur_t* c)
{
  if (take(c, 1)) return 0; /* a count above 64 */
  uint32_t cnt = take(c, 6) + 1, sent = 0;
  uint8_t mark[64];
  for (uint32_t i = 0; i < cnt; i++) mark[i] = (uint8_t)take(c, 1);
  for (uint32_t i = 0; i < cnt; i++) if (mark[i]) { if (!drop_open(c)) return 0; sent++; }
  (void)sent;
  return !c->bad;
}
/* the other radio access technologies (TS 36.331 6.3.5, release-8 components; later additions are open types behind the roots):
 * 1 UTRA: ARFCN 14 bits | offset 5 | cells to remove | cells to add: FDD (5 + 9 bits each) or TDD (5 + 7) | cell for CGI: FDD 9 / TDD 7
 * 2 GERAN: ARFCN 10, band 1, following ARFCNs as list (count 0..31, 10 bits each) / spacing 3 + count 5 / bit map of 1..16 octets | offset 5 | NCC mask 8 | cell for CGI 3 + 3
 * 3 CDMA2000: type 1, band class 1 + 5, ARFCN 11 | search window 4 | offset 5 | cells to remove | cells to add (5 + 9 each) | cell for CGI 9 */
static int drop_meas_object_other_rat(cur_t* c, uint32_t which)
{
  uint32_t more = take(c, 1);
  if (which == 1) {
    uint32_t has = take(c, 4);
    take(c, 14);
    if (has & 8u) take(c, 5);
    if (has & 4u) drop_list5(c, 5);
    if (has & 2u) { uint32_t tdd = take(c, 1); drop_list5(c, tdd ? 12 : 14); }
    if (has & 1u) { uint32_t tdd = take(c, 1); take(c, tdd ? 7 : 9); }
  } else if (which == 2) {
    uint32_t has = take(c, 3);
    take(c, 11);
    uint32_t form = take(c, 2);
    if (form == 0) { uint32_t cnt = take(c, 5); for (uint32_t i = 0; i < cnt; i++) take(c, 10); }
    else if (form == 1) take(c, 8);
    else if (form == 2) { uint32_t cnt = take(c, 4) + 1; for (uint32_t i = 0; i < cnt; i++) take(c, 8); }
    else return 0;
    if (has & 4u) take(c, 5);
    if (has & 2u) take(c, 8);
    if (has & 1u) take(c, 6);
  } else {
    uint32_t has = take(c, 5);
    take(c, 1);
    if (take(c, 1)) return 0;
    take(c, 16);
    if (has & 16u) take(c, 4);
    if (has & 8u) take(c, 5);
    if (has & 4u) drop_list5(c, 5);
    if (has & 2u) drop_list5(c, 14);
    if (has & 1u) take(c, 9);
  }
  if (more && !drop_additions(c)) return 0;
  return !c->bad;
}
static int drop_meas_object(cur_t* c)
{
  take(c, 5);
  if (take(c, 1)) return 0;          /* measObject choice extended */
  uint32_t which = take(c, 2);
  if (which != 0) return drop_meas_object_other_rat(c, which); /* UTRA / GERAN / CDMA2000 */
  uint32_t more = take(c, 1);        /* MeasObjectEUTRA carries extension additions */
  uint32_t has = take(c, 6);         /* offsetFreq, cellsToRemove, cellsToAddMod, blackCellsToRemove, blackCellsToAddMod, cellForWhichToReportCGI */
  take(c, 16 + 3 + 1 + 2);
  if (has & 32u) take(c, 5);
  if (has & 16u) drop_list5(c, 5);
  if (has & 8u) drop_list5(c, 19);
  if (has & 4u) drop_list5(c, 5);
  if (has & 2u) {
    uint32_t cnt = take(c, 5) + 1;
    for (uint32_t i = 0; i < cnt; i++) { take(c, 5); uint32_t rng = take(c, 1); take(c, 9); if (rng) take(c, 4); }
  }
  if (has & 1u) take(c, 9);
  if (more && !drop_additions(c)) return 0;
  return !c->bad;
}
/* ThresholdUTRA (RSCP 7 bits / EcN0 6 bits), ThresholdGERAN, ThresholdCDMA2000 (6 bits each) */
static int drop_threshold_other_rat(cur_t* c)
{
  uint32_t rat = take(c, 2);
  if (rat == 0) take(c, take(c, 1) ? 6 : 7);
  else if (rat <= 2) take(c, 6);
  else return 0;
  return 1;
}
/* ReportConfigInterRAT: event b1 (threshold of the other RAT) / b2 (E-UTRA threshold + threshold of the other RAT) with hysteresis and time to trigger, or periodical with a
 * purpose of three values; then maxReportCells 3, reportInterval 4, reportAmount 3 */
static int drop_report_config_inter_rat(cur_t* c)
{
  uint32_t more = take(c, 1);
  if (take(c, 1) == 0) {
    if (take(c, 1)) {
      if (take(c, 1)) return 0;
      take(c, 6);
      if (!drop_open(c)) return 0;
    } else if (take(c, 1) == 0) {
      if (!drop_threshold_other_rat(c)) return 0;
    } else {
      drop_threshold(c);
      if (!drop_threshold_other_rat(c)) return 0;
    }
    take(c, 9);
  } else {
    take(c, 2);
  }
  take(c, 10);
  if (more && !drop_additions(c)) return 0;
  return !c->bad;
}
static int drop_report_config(cur_t* c)
{
  take(c, 5);
  if (take(c, 1)) return drop_report_config_inter_rat(c);
  uint32_t more = take(c, 1);        /* extension additions behind reportAmount */
  if (take(c, 1) == 0) {             /* event */
    if (take(c, 1)) {                /* an event the r8 choice does not know (a6 ...): small index, then an open type */
      if (take(c, 1)) return 0;
      take(c, 6);
      if (!drop_open(c)) return 0;
    } else {
      uint32_t ev = take(c, 3);
      if (ev == 0 || ev == 1 || ev == 3) drop_threshold(c);
      else if (ev == 2) take(c, 7);
      else if (ev == 4) { drop_threshold(c); drop_threshold(c); }
      else return 0;
    }
    take(c, 5 + 4);
  } else {
    take(c, 1);
  }
  take(c, 1 + 1 + 3 + 4 + 3);
  if (more && !drop_additions(c)) return 0;
  return !c->bad;
}
static int drop_meas_config(cur_t* c)
{
  uint32_t more = take(c, 1);
  uint32_t has = take(c, 11); /* bit 10 = first optional component */
  if (has & (1u << 10)) drop_list5(c, 5);
  if (has & (1u << 9)) { uint32_t cnt = take(c, 5) + 1; for (uint32_t i = 0; i < cnt; i++) if (!drop_meas_object(c)) return 0; }
  if (has & (1u << 8)) drop_list5(c, 5);
  if (has & (1u << 7)) { uint32_t cnt = take(c, 5) + 1; for (uint32_t i = 0; i < cnt; i++) if (!drop_report_config(c)) return 0; }
  if (has & (1u << 6)) drop_list5(c, 5);
  if (has & (1u << 5)) drop_list5(c, 15);
  if (has & (1u << 4)) {
    uint32_t qmore = take(c, 1);
    uint32_t q = take(c, 4);
    if (q & 8u) {
      uint32_t d = take(c, 2);
      for (int k = 1; k >= 0; k--) if (d & (1u << k)) { if (take(c, 1)) return 0; take(c, 4); }
    }
    if (q & 4u) { uint32_t fc = take(c, 1); take(c, 1); if (fc) { if (take(c, 1)) return 0; take(c, 4); } } /* UTRA: FDD quantity (2 values), TDD quantity (1 value), filter */
    if (q & 2u) { uint32_t fc = take(c, 1); if (fc) { if (take(c, 1)) return 0; take(c, 4); } }             /* GERAN: quantity (1 value), filter */
    if (q & 1u) take(c, 1);                                                                                  /* CDMA2000: quantity (2 values) */
    if (qmore && !drop_additions(c)) return 0;
  }
  if ((has & (1u << 3)) && take(c, 1)) { /* measGapConfig setup: gapOffset is an extensible choice (36.331: gp0, gp1, ...) */
    if (take(c, 1)) return 0;            /* an alternative behind the extension marker */
    take(c, take(c, 1) ? 7 : 6);
  }
  if (has & (1u << 2)) take(c, 7);
  if (has & 2u) { /* HRPD pre-registration: allowed flag, zone id, one or two secondary zone ids */
    uint32_t o = take(c, 2);
    take(c, 1);
    if (o & 2u) take(c, 8);
    if (o & 1u) { uint32_t cnt = take(c, 1) + 1; take(c, 8 * cnt); }
  }
  if ((has & 1u) && take(c, 1)) take(c, 18); /* speed-state parameters set up: two timers (3 + 3), two counts (4 + 4), two scale factors (2 + 2) */
  if (more && !drop_additions(c)) return 0;
  return !c->bad;
}

/* 1 + *m_tmsi when the SDU (behind the assumed three header octets) is an RRCConnectionReconfiguration whose first NAS message is an attach
 * accept that assigns a GUTI, else 0 */
int o_rrc_reconfig_tmsi(const uint8_t* sdu, int len, uint32_t* m_tmsi)
{
  cur_t c = {sdu, len > 0 ? (uint32_t)len * 8u : 0u, 0, 0};
  if (take(&c, 1) != 0 || take(&c, 4) != 4) return 0;
  take(&c, 2);
  if (take(&c, 1) != 0 || take(&c, 3) != 0) return 0;
  uint32_t has = take(&c, 6);
  if (!(has & 8u) || (has & 16u)) return 0; /* no NAS list, or mobilityControlInfo in front of it */
  if ((has & 32u) && !drop_meas_config(&c)) return 0;
  take(&c, 4);
  uint32_t n = take(&c, 8);
  if (n & 0x80u) { if (n & 0x40u) return 0; n = ((n & 0x3Fu) << 8) | take(&c, 8); }
  if (c.bad || n < 8 || n > 256 || c.at + 8u * n > c.n) return 0;
  uint8_t m[256];
  for (uint32_t i = 0; i < n; i++) m[i] = (uint8_t)take(&c, 8);
  uint32_t at = 0, sht = m[0] >> 4;
  if (sht >= 1 && sht <= 4) at = 6; else if (sht != 0) return 0;
  if (at + 2 > n || (m[at] & 15u) != 7u || m[at + 1] != 0x42u) return 0;
  at += 4; /* message header, attach result, T3412 */
  if (at >= n || m[at] < 6 || m[at] > 96) return 0;
  at += 1u + m[at];
  if (at + 2 > n) return 0;
  at += 2u + (((uint32_t)m[at] << 8) | m[at + 1]);
  if (at + 13 > n || m[at] != 0x50u || m[at + 1] != 11u || (m[at + 2] & 7u) != 6u) return 0;
  *m_tmsi = ((uint32_t)m[at + 9] << 24) | ((uint32_t)m[at + 10] << 16) | ((uint32_t)m[at + 11] << 8) | (uint32_t)m[at + 12];
  return 1;
}

int o_api_dl_events(int api_mode, char name, const uint8_t* pdu, int len, uint16_t rnti, uint32_t tti, o_api_event_t* ev, int cap, int* nev)
{
  int n = 0, to_pcap = 0;
  if (name == 'P' && (api_mode == 2 || api_mode == 3)) {
    o_paging_id_t rec[16];
    int nr = o_paging_decode(pdu, len, rec, 16);
    for (int i = 0; i < nr; i++) {
      if (n < cap) {
        o_api_event_t* e = &ev[n++];
        memset(e, 0, sizeof(*e));
        e->tti = tti; e->rnti = 65534; e->msg_type = 5;
        if (rec[i].is_imsi) {
          e->id_type = 3;
          for (uint32_t k = 0; k < 15 && k < rec[i].nof_digits; k++) e->value[k] = (char)('0' + rec[i].digits[k]);
        } else {
          e->id_type = 1;
          snprintf(e->value, sizeof(e->value), "%08x", rec[i].m_tmsi);
        }
      }
      to_pcap = 1;
    }
  }
  if (name == 'C' && (api_mode == 0 || api_mode == 3)) {
    o_mac_subh_t sub[20];
    int ns = o_mac_dlsch_parse(pdu, len, sub, 20), setup = 0, found = 0, seen[10], nseen = 0;
    for (int i = 0; i < ns; i++) {
      if (sub[i].is_sdu && sub[i].lcid == 0) {
        o_ue_cfg_t c;
        if (o_rrc_conn_setup_decode(pdu + sub[i].off, (int)sub[i].len, &c)) setup = 1;
      } else if (sub[i].is_sdu && sub[i].lcid == 1) {
        uint32_t tm = 0; /* "sdu_ptr + 3": RLC + PDCP header octets assumed (DL_Sniffer_PDSCH.cc:836-838) */
        if (sub[i].len > 3 && o_rrc_reconfig_tmsi(pdu + sub[i].off + 3, (int)sub[i].len - 3, &tm) && n < cap) {
          o_api_event_t* e = &ev[n++];
          memset(e, 0, sizeof(*e));
          e->tti = tti; e->rnti = rnti; e->id_type = 1; e->msg_type = 6; /* ID_TMSI, MSG_CON_RECONFIG */
          snprintf(e->value, sizeof(e->value), "%08x", tm);
        }
      } else {
        if (nseen < 10) seen[nseen++] = i; else break;
      }
      if (setup) {
        for (int h = 0; h < nseen && !found; h++)
          if (sub[seen[h]].lcid == 28 && sub[seen[h]].len == 6) {
            unsigned long long id = 0;
            char hex[24];
            for (int k = 0; k < 6; k++) id = (id << 8) | pdu[sub[seen[h]].off + k];
            snprintf(hex, sizeof(hex), "%llx", id);
            if (n < cap) {
              o_api_event_t* e = &ev[n++];
              memset(e, 0, sizeof(*e));
              e->tti = tti; e->rnti = rnti; e->id_type = 2; e->msg_type = 1;
              if (strlen(hex) >= 3) snprintf(e->value, sizeof(e->value), "%.8s", hex + 3);
            }
            found = 1;
          }
        to_pcap = 1;
      }
    }
  }
  if (nev) *nev = n;
  return to_pcap;
}

/* Uplink side of the identity mapping: a decoded Msg3 (PUSCH of a RAR grant), PUSCH_Decoder::decode_run's API part (UL_Sniffer_PUSCH.cc:306-327)
 * with decode_rrc_connection_request (:47-93).  MAC UL-SCH walk per TS 36.321 6.1.2 / 6.2.1 [srsran::sch_pdu, uplink], UL-CCCH-Message per
 * TS 36.331 6.2.2.  Pinned by the five Msg3 blocks of the reference's api_collector.pcap: the value reported equals the characters the matching
 * connection setup reports for its contention resolution identity.  Returns whether the block goes to the API pcap. */
static int ul_ce_size(uint32_t lcid) { return lcid == 26 ? 1 : lcid == 27 ? 2 : (lcid == 28 || lcid == 29) ? 1 : lcid == 30 ? 3 : 0; }
int o_api_ul_msg3_events(int api_mode, const uint8_t* pdu, int len, uint16_t rnti, uint32_t tti, o_api_event_t* ev, int cap, int* nev)
{
  int n = 0, ok = 1;
  if (nev) *nev = 0;
  if (!(api_mode == 0 || api_mode == 3)) return 0;
  /* sub-headers */
  struct { uint32_t lcid, is_sdu, off, len; } sub[10];
  int ns = 0, pos = 0, more = 1, bad = len <= 0;
  while (!bad && more && ns < 10 && pos < len) {
    uint8_t b = pdu[pos++];
    sub[ns].lcid = b & 0x1Fu; sub[ns].is_sdu = sub[ns].lcid < 26; sub[ns].len = 0;
    more = (b >> 5) & 1;
    if (sub[ns].is_sdu && more) {
      if (pos >= len) { bad = 1; break; }
      uint8_t l = pdu[pos++];
      sub[ns].len = l & 0x7Fu;
      if (l & 0x80u) { if (pos >= len) { bad = 1; break; } sub[ns].len = (sub[ns].len << 8) | pdu[pos++]; }
    }
    ns++;
    if (more && pos >= len) bad = 1;
  }
  if (more && ns == 10) bad = 1;
  for (int i = 0; i < ns && !bad; i++) {
    if (!sub[i].is_sdu) sub[i].len = (uint32_t)ul_ce_size(sub[i].lcid);
    sub[i].off = (uint32_t)pos;
    if (i == ns - 1 && sub[i].is_sdu) sub[i].len = (uint32_t)(len - pos);
    pos += (int)sub[i].len;
    if (pos > len) bad = 1;
  }
  if (bad) ns = 0;
  for (int i = 0; i < ns; i++) {
    if (!sub[i].is_sdu) continue;
    ok = 0;
    br_t b = {pdu + sub[i].off, 8u * sub[i].len, 0, 0};
    if (rd(&b, 1)) continue;
    int req = (int)rd(&b, 1);
    if (rd(&b, 1)) continue;
    if (!req) continue; /* rrcConnectionReestablishmentRequest: "do nothing" */
    int random = (int)rd(&b, 1);
    rd(&b, 8);
    uint32_t lo = rd(&b, 32);
    rd(&b, 4);
    if (b.err) continue;
    if (n < cap) {
      o_api_event_t* e = &ev[n++];
      memset(e, 0, sizeof(*e));
      e->tti = tti; e->rnti = rnti; e->msg_type = 0; e->id_type = random ? 0u : 1u;
      if (random) snprintf(e->value, sizeof(e->value), "%08x", lo); else snprintf(e->value, sizeof(e->value), "%x", lo);
    }
    ok = 1;
  }
  if (nev) *nev = n;
  return ok;
}

/* Uplink SRB messages (api_mode 1 / 2 / 3): PUSCH_Decoder::decode_run :328-372 (MAC UL-SCH walk, RLC AM data PDU header [srsRAN rlc_am_lte.cc],
 * one PDCP octet), decode_ul_dcch :95-143 (UL-DCCH-Message, TS 36.331 6.2.1 / 6.2.2) and decode_nas_ul :146-247 (TS 24.301 8.2.4 attach request,
 * 8.2.19 identity response; mobile identities TS 24.008 10.5.1.4 / TS 24.301 9.9.3.12).  Pinned by the reference's api_collector.pcap: its five
 * 333-byte blocks are RRCConnectionSetupComplete + attach request with a GUTI, its five 533-byte blocks UECapabilityInformation - the blocks the
 * reference's own parsers accepted.  The body of a UECapabilityInformation is not unpacked (the reference unpacks it and would reject a
 * malformed one).  The "NAS security header" test of :349-354 reads the PDCP octet (buffer-layout cast): high nibble 0, 1 or 3. */
static int o_bcd(const uint8_t* v, int len, int ndig, char* out)
{
  int n = 0;
  if (len < 1) return 0;
  out[n++] = (char)('0' + (v[0] >> 4));
  for (int i = 1; i < len && n < ndig; i++) {
    out[n++] = (char)('0' + (v[i] & 0xF));
    if (n < ndig) out[n++] = (char)('0' + (v[i] >> 4));
  }
  out[n] = 0;
  return n;
}
static void o_api_add(o_api_event_t* ev, int cap, int* n, uint32_t tti, uint16_t rnti, uint32_t id, uint32_t msg, const char* v)
{
  if (*n >= cap) return;
  o_api_event_t* e = &ev[(*n)++];
  memset(e, 0, sizeof(*e));
  e->tti = tti; e->rnti = rnti; e->id_type = id; e->msg_type = msg;
  snprintf(e->value, sizeof(e->value), "%s", v);
}
static int o_nas_ul_identity(const uint8_t* nas, int len, uint16_t rnti, uint32_t tti, o_api_event_t* ev, int cap, int* n)
{
  char v[24];
  if (len < 2) return 0;
  uint32_t sec = nas[0] >> 4;
  if (sec == 2 || sec == 4) return 0;
  int body = sec == 0 ? 0 : 6;
  if (len < body + 2) return 0;
  uint32_t msg_type = nas[body + 1];
  const uint8_t* p = nas + body + 2;
  int left = len - body - 2;
  if (msg_type == 0x56) {
    if (left < 2 || p[0] < 1 || p[0] > left - 1) return 0;
    uint32_t type = p[1] & 7u;
    if (type == 1) { o_bcd(p + 1, p[0], 15, v); o_api_add(ev, cap, n, tti, rnti, 3, 3, v); return 1; }
    if (type == 2) { o_bcd(p + 1, p[0], 15, v); o_api_add(ev, cap, n, tti, rnti, 4, 3, v); return 1; }
    if (type == 3) { o_bcd(p + 1, p[0], 16, v); o_api_add(ev, cap, n, tti, rnti, 5, 3, v); return 1; }
    return 0;
  }
  if (msg_type == 0x41) {
    if (left < 3) return 0;
    p += 1; left -= 1;
    if (p[0] < 1 || p[0] > left - 1) return 0;
    uint32_t type = p[1] & 7u;
    if (type == 1) { o_bcd(p + 1, p[0], 15, v); o_api_add(ev, cap, n, tti, rnti, 3, 2, v); return 1; }
    if (type == 6) {
      if (p[0] < 11) return 0;
      uint32_t m_tmsi = ((uint32_t)p[8] << 24) | ((uint32_t)p[9] << 16) | ((uint32_t)p[10] << 8) | p[11];
      snprintf(v, sizeof(v), "%x", m_tmsi);
      o_api_add(ev, cap, n, tti, rnti, 1, 2, v);
      return 1;
    }
    if (type == 3) { o_bcd(p + 1, p[0], 15, v); o_api_add(ev, cap, n, tti, rnti, 4, 2, v); return 1; }
    return 0;
  }
  return 0;
}
static int o_octets(br_t* b, uint8_t* out, int cap)
{
  uint32_t n = rd_len(b);
  if (b->err || b->pos + 8u * n > b->nbits || (int)n > cap) return -1;
  for (uint32_t i = 0; i < n; i++) out[i] = (uint8_t)rd(b, 8);
  return b->err ? -1 : (int)n;
}
static int o_ul_dcch(int api_mode, const uint8_t* rrc, int len, uint16_t rnti, uint32_t tti, o_api_event_t* ev, int cap, int* n)
{
  br_t b = {rrc, len > 0 ? 8u * (uint32_t)len : 0u, 0, 0};
  uint8_t nas[2048];
  if (rd(&b, 1)) return 0;
  uint32_t type = rd(&b, 4);
  if (b.err) return 0;
  if (type == 7 && (api_mode == 1 || api_mode == 3)) {
    rd(&b, 2);
    if (rd(&b, 1) || b.err) return 0;
    o_api_add(ev, cap, n, tti, rnti, 0xFFFFFFFFu, 4, "-");
    return 1;
  }
  if (type == 4 && (api_mode == 2 || api_mode == 3)) {
    rd(&b, 2);
    if (rd(&b, 1)) return 0;
    if (rd(&b, 2) != 0) return 0;
    int mme = (int)rd(&b, 1);
    rd(&b, 1);
    rd_int(&b, 1, 6);
    if (mme) {
      if (rd(&b, 1)) {
        if (rd(&b, 1)) { rd_int(&b, 0, 9); rd_int(&b, 0, 9); rd_int(&b, 0, 9); }
        uint32_t nd = rd(&b, 1) + 2;
        for (uint32_t i = 0; i < nd; i++) rd_int(&b, 0, 9);
      }
      rd(&b, 16); rd(&b, 8);
    }
    int nn = o_octets(&b, nas, (int)sizeof(nas));
    if (nn < 0) return 0;
    return o_nas_ul_identity(nas, nn, rnti, tti, ev, cap, n);
  }
  if (type == 9 && (api_mode == 2 || api_mode == 3)) {
    if (rd(&b, 1)) return 0;
    if (rd(&b, 2) != 0) return 0;
    rd(&b, 1);
    if (rd(&b, 2) != 0) return 0;
    int nn = o_octets(&b, nas, (int)sizeof(nas));
    if (nn < 0) return 0;
    return o_nas_ul_identity(nas, nn, rnti, tti, ev, cap, n);
  }
  return 0;
}
int o_api_ul_dcch_events(int api_mode, const uint8_t* pdu, int len, uint16_t rnti, uint32_t tti, o_api_event_t* ev, int cap, int* nev)
{
  int n = 0, ok = 0;
  if (nev) *nev = 0;
  if (api_mode <= 0) return 0;
  struct { uint32_t lcid, is_sdu, off, len; } sub[10];
  int ns = 0, pos = 0, more = 1, bad = len <= 0;
  while (!bad && more && ns < 10 && pos < len) {
    uint8_t b = pdu[pos++];
    sub[ns].lcid = b & 0x1Fu; sub[ns].is_sdu = sub[ns].lcid < 26; sub[ns].len = 0;
    more = (b >> 5) & 1;
    if (sub[ns].is_sdu && more) {
      if (pos >= len) { bad = 1; break; }
      uint8_t l = pdu[pos++];
      sub[ns].len = l & 0x7Fu;
      if (l & 0x80u) { if (pos >= len) { bad = 1; break; } sub[ns].len = (sub[ns].len << 8) | pdu[pos++]; }
    }
    ns++;
    if (more && pos >= len) bad = 1;
  }
  if (more && ns == 10) bad = 1;
  for (int i = 0; i < ns && !bad; i++) {
    if (!sub[i].is_sdu) sub[i].len = (uint32_t)ul_ce_size(sub[i].lcid);
    sub[i].off = (uint32_t)pos;
    if (i == ns - 1 && sub[i].is_sdu) sub[i].len = (uint32_t)(len - pos);
    pos += (int)sub[i].len;
    if (pos > len) bad = 1;
  }
  if (bad) ns = 0;
  for (int i = 0; i < ns; i++) {
    if (!(sub[i].is_sdu && (sub[i].lcid == 1 || sub[i].lcid == 2))) continue;
    const uint8_t* p = pdu + sub[i].off;
    int left = (int)sub[i].len, hdr = 2;
    if (left < 3 || !(p[0] & 0x80u)) continue;
    if (p[0] & 0x40u) continue; /* re-segmentation */
    uint32_t fi = (p[0] >> 3) & 3u;
    if (p[0] & 0x04u) {
      int bitpos = 16, m = 1;
      while (m) {
        if ((bitpos + 12 + 7) / 8 > left) { hdr = -1; break; }
        m = (p[bitpos >> 3] >> (7 - (bitpos & 7))) & 1;
        bitpos += 12;
      }
      if (hdr < 0) continue;
      hdr = (bitpos + 7) / 8;
    }
    if (fi != 0 || left < hdr + 2) continue;
    p += hdr; left -= hdr;
    uint32_t nib = p[0] >> 4;
    if (!(nib == 0 || nib == 1 || nib == 3)) continue;
    ok = o_ul_dcch(api_mode, p + 1, left - 1, rnti, tti, ev, cap, &n);
  }
  if (nev) *nev = n;
  return ok;
}
