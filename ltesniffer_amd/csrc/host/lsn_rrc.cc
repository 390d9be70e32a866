// lsn_rrc.cc - what the decode loop reads out of a decoded C-RNTI transport block before the next grant is decoded:
//  * the MAC DL-SCH PDU walk the reference does with srsran::sch_pdu (parse_packet / next / get:
//    /root/reference/src/src/DL_Sniffer_PDSCH.cc:1041-1070, 1133-1160, 279-310; TS 36.321 6.1.2, 6.2.1), and
//  * PDSCH_Decoder::decode_rrc_connection_setup (DL_Sniffer_PDSCH.cc:129-181): an RRCConnectionSetup on the CCCH gives the UE's
//    p-a (PDSCH power offset used by every later decode of that RNTI, :926-927), the betaOffset indices and the aperiodic CQI
//    report mode (used by the PUSCH decoder in UL mode, UL_Sniffer_PUSCH.cc:433-435).
// The reference unpacks the whole DL-CCCH-Message with srsRAN's generated ASN.1 code; here the UPER encoding (TS 36.331 6.2.2 /
// 6.3.2, X.691) is walked directly: Rel-8 root components are read, extension additions are skipped by their length
// determinants.  Messages that carry DRB or SPS components (never the case for a connection setup) are not accepted.
// HIP-free host code: built into the product library and, for the CPU tests, into tests/native.
#include "lsn_lte.h"
#include <cstdio>
#include <cstring>
#include <vector>

namespace lsn {

int mac_dlsch_parse(const uint8_t* pdu, int len, MacSubheader* out, int cap)
{
  if (len <= 0) return 0;
  int pos = 0, n = 0;
  bool more = true;
  while (more && n < cap && pos < len) {
    const uint8_t b = pdu[pos++];
    MacSubheader& s = out[n++];
    s.lcid = b & 0x1Fu;
    s.is_sdu = s.lcid < 26;  // LCIDs below the control elements (sch_subh::is_sdu)
    s.len = 0;
    more = (b & 0x20u) != 0;
    if (s.is_sdu && more) {  // F / L exist only when another subheader follows
      if (pos >= len) return 0;
      const uint8_t l = pdu[pos++];
      s.len = l & 0x7Fu;
      if (l & 0x80u) {
        if (pos >= len) return 0;
        s.len = (s.len << 8) | pdu[pos++];
      }
    }
    if (more && pos >= len) return 0;
  }
  if (more && n == cap) return 0;
  for (int i = 0; i < n; i++) {
    MacSubheader& s = out[i];
    if (!s.is_sdu) s.len = s.lcid == 28 ? 6u : (s.lcid == 29 || s.lcid == 27) ? 1u : 0u;  // contention resolution id, TA command, SCell activation
    s.off = (uint32_t)pos;
    if (i == n - 1 && s.is_sdu) s.len = (uint32_t)(len - pos);  // the last SDU has no length field
    pos += (int)s.len;
    if (pos > len) return 0;
  }
  return n;
}

namespace {
struct BitReader {
  const uint8_t* p;
  uint32_t nbits, pos = 0;
  bool err = false;
  uint32_t get(uint32_t n)
  {
    uint32_t v = 0;
    for (uint32_t i = 0; i < n; i++) {
      if (pos >= nbits) { err = true; return 0; }
      v = (v << 1) | ((p[pos >> 3] >> (7 - (pos & 7))) & 1u);
      pos++;
    }
    return v;
  }
  bool flag() { return get(1) != 0; }
  uint32_t integer(uint32_t lb, uint32_t ub)  // constrained whole number
  {
    const uint32_t range = ub - lb + 1;
    uint32_t nb = 0;
    while ((1u << nb) < range) nb++;
    const uint32_t v = get(nb);
    if (v >= range) { err = true; return lb; }
    return lb + v;
  }
  uint32_t length()  // general length determinant, unaligned
  {
    if (!flag()) return get(7);
    if (!flag()) return get(14);
    err = true;
    return 0;
  }
  void skipOctets(uint32_t n)
  {
    if (pos + 8u * n > nbits) { err = true; return; }
    pos += 8u * n;
  }
  void skipExtensions()  // extension additions of a SEQUENCE whose extension bit is set
  {
    const uint32_t n = !flag() ? get(6) + 1 : length();
    uint32_t present = 0;
    for (uint32_t i = 0; i < n && !err; i++) present += get(1);
    for (uint32_t i = 0; i < present && !err; i++) skipOctets(length());
  }
};

void rlcConfig(BitReader& b)
{
  if (b.flag()) { b.err = true; return; }
  switch (b.get(2)) {
    case 0: b.get(6); b.get(3); b.get(4); b.get(3); b.get(5); b.get(6); break;  // am
    case 1: b.get(1); b.get(1); b.get(5); break;                                // um-Bi-Directional
    case 2: b.get(1); break;                                                    // um-Uni-Directional-UL
    default: b.get(1); b.get(5); break;                                         // um-Uni-Directional-DL
  }
}
void logicalChannelConfig(BitReader& b)
{
  const bool ext = b.flag(), ul = b.flag();
  if (ul) {
    const bool grp = b.flag();
    b.get(4); b.get(4); b.get(3);
    if (grp) b.get(2);
  }
  if (ext) b.skipExtensions();
}
void macMainConfig(BitReader& b)
{
  const bool ext = b.flag(), ulsch = b.flag(), drx = b.flag(), phr = b.flag();
  if (ulsch) {
    const bool harq = b.flag(), bsr = b.flag();
    if (harq) b.get(4);
    if (bsr) b.get(4);
    b.get(3); b.get(1);
  }
  if (drx && b.flag()) {
    static const uint16_t kCycle[16] = {10, 20, 32, 40, 64, 80, 128, 160, 256, 320, 512, 640, 1024, 1280, 2048, 2560};
    const bool shortDrx = b.flag();
    b.get(4); b.get(5); b.get(3);
    b.integer(0, kCycle[b.get(4)] - 1u);
    if (shortDrx) { b.get(4); b.get(4); }
  }
  b.get(3);
  if (phr && b.flag()) { b.get(3); b.get(3); b.get(2); }
  if (ext) b.skipExtensions();
}
void tpcPdcchConfig(BitReader& b)
{
  if (!b.flag()) return;
  b.get(16);
  if (b.flag()) b.integer(1, 31); else b.integer(1, 15);
}
}  // namespace

bool rrc_conn_setup_decode(const uint8_t* sdu, int len, UeSpecConfig& out)
{
  static const float kPaDb[8] = {-6.0f, -4.77f, -3.0f, -1.77f, 0.0f, 1.0f, 2.0f, 3.0f};  // DL_Sniffer_PDSCH.cc:3
  BitReader b{sdu, len > 0 ? 8u * (uint32_t)len : 0u};
  out = UeSpecConfig();  // ltesniffer_ue_spec_config_t ue_config = {}
  out.p_a = 0.0f; out.i_offset_ack = out.i_offset_cqi = out.i_offset_ri = 0; out.cqi_type = 0;
  if (b.flag()) return false;        // DL-CCCH-MessageType c1
  if (b.get(2) != 3) return false;   // rrcConnectionSetup
  b.get(2);                          // rrc-TransactionIdentifier
  if (b.flag()) return false;        // criticalExtensions c1
  if (b.get(3) != 0) return false;   // rrcConnectionSetup-r8
  const bool noncrit = b.flag();
  const bool rrExt = b.flag(), srb = b.flag(), drbAdd = b.flag(), drbRel = b.flag(), mac = b.flag(), sps = b.flag(), phy = b.flag();
  if (drbAdd || sps) return false;
  if (srb) {
    const uint32_t n = b.get(1) + 1;
    for (uint32_t i = 0; i < n && !b.err; i++) {
      const bool ext = b.flag(), rlc = b.flag(), lc = b.flag();
      b.get(1);
      if (rlc && !b.flag()) rlcConfig(b);
      if (lc && !b.flag()) logicalChannelConfig(b);
      if (ext) b.skipExtensions();
    }
  }
  if (drbRel) {
    const uint32_t n = b.integer(1, 11);
    for (uint32_t i = 0; i < n; i++) b.get(5);
  }
  if (mac && !b.flag()) macMainConfig(b);
  if (phy) {
    const bool ext = b.flag();
    bool f[10];
    for (bool& x : f) x = b.flag();
    if (f[0]) out.p_a = kPaDb[b.get(3)];
    if (f[1]) {
      const bool tdd = b.flag();
      if (b.flag()) { b.get(2); b.get(11); }
      if (tdd) b.get(1);
    }
    if (f[2]) {
      out.i_offset_ack = b.get(4);
      out.i_offset_ri = b.get(4);
      out.i_offset_cqi = b.get(4);
    }
    if (f[3]) {
      const bool fc = b.flag();
      b.get(4); b.get(1); b.get(1); b.get(4); b.get(4);
      if (fc) { if (b.flag()) b.err = true; b.get(4); }
    }
    if (f[4]) tpcPdcchConfig(b);
    if (f[5]) tpcPdcchConfig(b);
    if (f[6]) {
      const bool aper = b.flag(), per = b.flag();
      if (aper) {
        const uint32_t m = b.get(3);  // rm12, rm20, rm22, rm30, rm31
        if (m == 0) out.cqi_type = 0; else if (m <= 2) out.cqi_type = 1; else if (m <= 4) out.cqi_type = 2;
      }
      b.get(3);
      if (per && b.flag()) {
        const bool ri = b.flag();
        b.integer(0, 1185);
        b.get(10);
        if (b.flag()) b.get(2);
        if (ri) b.get(10);
        b.get(1);
      }
    }
    if (f[7] && b.flag()) { b.get(2); b.get(2); b.integer(0, 23); b.get(1); b.get(10); b.get(1); b.get(3); }
    if (f[8] && !b.flag()) {
      static const uint8_t kCbsrBits[8] = {2, 4, 6, 64, 4, 16, 4, 16};
      const bool cb = b.flag();
      b.get(3);
      if (cb) {
        uint32_t nb = kCbsrBits[b.get(3)];
        while (nb) { const uint32_t k = nb > 16 ? 16 : nb; b.get(k); nb -= k; }
      }
      if (b.flag()) b.get(1);
    }
    if (f[9] && b.flag()) { b.get(11); b.integer(0, 157); b.get(3); }
    if (ext) b.skipExtensions();
  }
  if (rrExt) b.skipExtensions();
  if (noncrit) {
    const bool late = b.flag();
    b.flag();
    if (late) b.skipOctets(b.length());
  }
  if (b.err) return false;
  out.has_ue_config = true;
  return true;
}

// BCCH-DL-SCH-Message (TS 36.331 6.2.1 / 6.2.2 / 6.3.1 / 6.3.2, UPER).  The reference unpacks the whole message with srsRAN's generated
// code and then looks for a sib2 entry in sib-TypeAndInfo (DL_Sniffer_PDSCH.cc:531-557).  Here the walk covers the message head and a
// complete SystemInformationBlockType2 (root components read, extension additions skipped by their length determinants).  SIB2 is always
// the first entry of the SI message that carries it (36.331 5.2.1.2 / 6.2.2: schedulingInfoList's first SI message implicitly starts with
// it); entries behind it are not unpacked (the reference would also reject a message whose LATER blocks do not unpack).
int sib2_decode(const uint8_t* pdu, int len, Sib2Config& out)
{
  BitReader b{pdu, len > 0 ? 8u * (uint32_t)len : 0u};
  if (b.flag()) return 0;                    // BCCH-DL-SCH-MessageType: c1
  if (b.flag()) return b.err ? 0 : 1;        // c1: systemInformationBlockType1 ("do nothing")
  if (b.flag()) return 0;                    // criticalExtensions: systemInformation-r8
  b.flag();                                  // nonCriticalExtension present
  b.get(5);                                  // sib-TypeAndInfo: SIZE (1..32)
  if (b.flag()) return 0;                    // entry CHOICE: extension alternative (sib12 ...) never in front of SIB2
  const uint32_t alt = b.get(4);             // sib2, sib3, ... sib11
  if (b.err || alt > 9) return 0;
  if (alt != 0) return 1;
  Sib2Config o;
  // ---- SystemInformationBlockType2 ----
  const bool sibExt = b.flag(), acBarring = b.flag(), mbsfn = b.flag();
  if (acBarring) {
    const bool sig = b.flag(), data = b.flag();
    b.get(1);                                // ac-BarringForEmergency
    if (sig) { b.get(4); b.get(3); b.get(5); }
    if (data) { b.get(4); b.get(3); b.get(5); }
  }
  // RadioResourceConfigCommonSIB
  const bool rrExt = b.flag();
  {  // rach-ConfigCommon
    const bool ext = b.flag(), groupA = b.flag();
    b.get(4);                                // numberOfRA-Preambles
    if (groupA) {
      const bool gext = b.flag();
      b.integer(0, 14); b.get(2); b.get(3);  // sizeOfRA-PreamblesGroupA, messageSizeGroupA, messagePowerOffsetGroupB
      if (gext) b.skipExtensions();
    }
    b.get(2); b.get(4);                      // powerRampingParameters
    b.integer(0, 10); b.get(3); b.get(3);    // ra-SupervisionInfo
    b.get(3);                                // maxHARQ-Msg3Tx
    if (ext) b.skipExtensions();
  }
  b.get(2);                                  // bcch-Config
  b.get(2); b.get(3);                        // pcch-Config
  o.root_seq_idx = b.integer(0, 837);        // prach-Config
  o.prach_config_idx = b.get(6); o.high_speed_flag = b.get(1); o.zero_corr_zone = b.get(4); o.prach_freq_offset = b.integer(0, 94);
  b.integer(0, 110); b.get(2);               // pdsch-ConfigCommon: referenceSignalPower (-60..50), p-b
  o.n_sb = b.get(2) + 1; o.hopping_mode = b.get(1); o.pusch_hop_offset = b.integer(0, 98); o.enable_64qam = b.get(1);
  o.group_hopping_enabled = b.get(1); o.group_assignment_pusch = b.integer(0, 29); o.sequence_hopping_enabled = b.get(1); o.cyclic_shift = b.get(3);
  b.integer(0, 2); b.integer(0, 98); b.get(3); b.get(11);  // pucch-ConfigCommon
  if (b.flag()) {                            // soundingRS-UL-ConfigCommon: setup
    b.flag();                                // srs-MaxUpPts present (one enumeration value: no bits)
    b.get(3); b.get(4); b.get(1);
  }
  b.integer(0, 150); b.get(3); b.get(5);     // uplinkPowerControlCommon: p0-NominalPUSCH (-126..24), alpha, p0-NominalPUCCH
  b.integer(0, 2); b.integer(0, 2); b.get(2); b.integer(0, 2); b.integer(0, 2);  // deltaFList-PUCCH
  b.get(3);                                  // deltaPreambleMsg3
  b.get(1);                                  // ul-CyclicPrefixLength
  if (rrExt) b.skipExtensions();
  {  // ue-TimersAndConstants
    const bool ext = b.flag();
    b.get(3); b.get(3); b.integer(0, 6); b.get(3); b.integer(0, 6); b.get(3);
    if (ext) b.skipExtensions();
  }
  {  // freqInfo
    const bool carrier = b.flag(), bw = b.flag();
    if (carrier) b.get(16);
    if (bw) b.integer(0, 5);
    b.get(5);
  }
  if (mbsfn) {
    const uint32_t n = b.get(3) + 1;
    for (uint32_t i = 0; i < n && !b.err; i++) { b.integer(0, 5); b.get(3); if (b.flag()) b.get(24); else b.get(6); }
  }
  b.get(3);                                  // timeAlignmentTimerCommon
  if (sibExt) b.skipExtensions();
  if (b.err) return 0;
  out = o;
  return 2;
}

// PCCH-Message (TS 36.331 6.2.1 / 6.2.2): c1 { paging { pagingRecordList OPTIONAL, systemInfoModification OPTIONAL, etws-Indication OPTIONAL,
// nonCriticalExtension OPTIONAL } }; PagingRecord { ue-Identity CHOICE { s-TMSI { mmec BIT STRING (8), m-TMSI BIT STRING (32) }, imsi SEQUENCE
// (SIZE (6..21)) OF INTEGER (0..9), ... }, cn-Domain, ... }.  Only the record list is read (the reference reads nothing else).
int paging_decode(const uint8_t* pdu, int len, PagingId* out, int cap)
{
  BitReader b{pdu, len > 0 ? 8u * (uint32_t)len : 0u};
  if (b.flag()) return -1;  // messageClassExtension
  const bool list = b.flag();
  b.flag(); b.flag(); b.flag();  // systemInfoModification, etws-Indication, nonCriticalExtension: behind the list, not read
  if (b.err) return -1;
  int n = 0;
  if (list) {
    const uint32_t cnt = b.get(4) + 1;
    for (uint32_t i = 0; i < cnt && !b.err; i++) {
      const bool ext = b.flag();
      PagingId id;
      if (b.flag()) return -1;  // an identity outside the root alternatives
      if (!b.flag()) {
        id.mmec = b.get(8); id.m_tmsi = b.get(32);
      } else {
        id.is_imsi = true;
        id.nof_digits = b.get(4) + 6;
        if (id.nof_digits > 21) return -1;
        for (uint32_t k = 0; k < id.nof_digits; k++) { id.digits[k] = (uint8_t)b.get(4); if (id.digits[k] > 9) b.err = true; }
      }
      b.get(1);  // cn-Domain
      if (ext) b.skipExtensions();
      if (b.err) return -1;
      if (n < cap) out[n++] = id;
    }
  }
  return b.err ? -1 : n;
}

// MAC UL-SCH PDU (TS 36.321 6.1.2, 6.2.1): sub-headers like the downlink; control elements: power headroom (26) 1 byte, C-RNTI (27) 2,
// truncated / short BSR (28, 29) 1, long BSR (30) 3, padding (31) 0 [srsran::sch_subh::sizeof_ce, uplink branch]
static int mac_ulsch_parse(const uint8_t* pdu, int len, MacSubheader* out, int cap)
{
  if (len <= 0) return 0;
  int pos = 0, n = 0;
  bool more = true;
  while (more && n < cap && pos < len) {
    const uint8_t b = pdu[pos++];
    MacSubheader& s = out[n++];
    s.lcid = b & 0x1Fu;
    s.is_sdu = s.lcid < 26;
    s.len = 0;
    more = (b & 0x20u) != 0;
    if (s.is_sdu && more) {
      if (pos >= len) return 0;
      const uint8_t l = pdu[pos++];
      s.len = l & 0x7Fu;
      if (l & 0x80u) {
        if (pos >= len) return 0;
        s.len = (s.len << 8) | pdu[pos++];
      }
    }
    if (more && pos >= len) return 0;
  }
  if (more && n == cap) return 0;
  for (int i = 0; i < n; i++) {
    MacSubheader& s = out[i];
    if (!s.is_sdu) s.len = s.lcid == 26 ? 1u : s.lcid == 27 ? 2u : (s.lcid == 28 || s.lcid == 29) ? 1u : s.lcid == 30 ? 3u : 0u;
    s.off = (uint32_t)pos;
    if (i == n - 1 && s.is_sdu) s.len = (uint32_t)(len - pos);
    pos += (int)s.len;
    if (pos > len) return 0;
  }
  return n;
}

bool api_ul_msg3_events(int api_mode, const uint8_t* pdu, int len, uint16_t rnti, uint32_t tti, ApiEvent* ev, int cap, int* nev)
{
  int n = 0;
  if (nev) *nev = 0;
  if (!(api_mode == 0 || api_mode == 3)) return false;
  MacSubheader sub[10];
  const int ns = mac_ulsch_parse(pdu, len, sub, 10);
  bool ok = true;  // the reference tests the result of the LAST SDU; a PDU without SDU keeps the (successful) result of the PUSCH decode
  for (int i = 0; i < ns; i++) {
    if (!sub[i].is_sdu) continue;
    ok = false;
    // UL-CCCH-Message: c1 { rrcConnectionReestablishmentRequest | rrcConnectionRequest { rrcConnectionRequest-r8 { ue-Identity CHOICE
    // { s-TMSI { mmec (8), m-TMSI (32) } | randomValue (40) }, establishmentCause (3), spare (1) } } }
    BitReader b{pdu + sub[i].off, 8u * sub[i].len};
    if (b.flag()) continue;                 // messageClassExtension
    const bool req = b.flag();
    if (b.flag()) continue;                 // criticalExtensionsFuture
    if (!req) { b.get(16); b.get(9); b.get(16); b.get(2); b.get(2); continue; }  // reestablishment request: nothing is reported
    const bool random = b.flag();
    const uint32_t hi = b.get(8), lo = b.get(32);
    b.get(3); b.get(1);
    if (b.err) continue;
    char v[24] = {0};
    if (n < cap) {
      ApiEvent& e = ev[n++];
      e.tti = tti; e.rnti = rnti; e.msg_type = API_MSG_CON_REQ;
      if (!random) { e.id_type = API_ID_TMSI; std::snprintf(v, sizeof(v), "%x", lo); }  // m-TMSI in hex without padding (:63-66)
      else { e.id_type = API_ID_RAN_VAL; std::snprintf(v, sizeof(v), "%08x", lo); }  // ten hex digits, the first two (hi) dropped (:72-81)
      (void)hi;
      std::memcpy(e.value, v, sizeof(v));
    }
    ok = true;
  }
  if (nev) *nev = n;
  return ok;
}

namespace {
// mobile identity digits (TS 24.008 10.5.1.4 / TS 24.301 9.9.3.12): first digit in the high nibble of the type octet, then low / high nibbles
int bcd_digits(const uint8_t* v, int len, int ndig, char* out)
{
  if (len < 1) return 0;
  int n = 0;
  out[n++] = (char)('0' + (v[0] >> 4));
  for (int i = 1; i < len && n < ndig; i++) {
    out[n++] = (char)('0' + (v[i] & 0xF));
    if (n < ndig) out[n++] = (char)('0' + (v[i] >> 4));
  }
  out[n] = 0;
  return n;
}

// PUSCH_Decoder::decode_nas_ul (UL_Sniffer_PUSCH.cc:146-247) on a dedicatedInfoNAS: identity response and attach request without ciphering
bool nas_ul_identity(const uint8_t* nas, int len, uint16_t rnti, uint32_t tti, ApiEvent* ev, int cap, int& n)
{
  if (len < 2) return false;
  const uint32_t sec = nas[0] >> 4;
  if (sec == 2 || sec == 4) return false;       // integrity protected and ciphered: nothing to read
  const int body = sec == 0 ? 0 : 6;             // security header (1) + MAC (4) + sequence number (1) in front of the plain message
  if (len < body + 2) return false;
  const uint32_t msg_type = nas[body + 1];
  const uint8_t* p = nas + body + 2;
  int left = len - body - 2;
  auto add = [&](uint32_t id, uint32_t msg, const char* v) {
    if (n >= cap) return;
    ApiEvent& e = ev[n++];
    e.tti = tti; e.rnti = rnti; e.id_type = id; e.msg_type = msg;
    std::snprintf(e.value, sizeof(e.value), "%s", v);
  };
  char v[24];
  if (msg_type == 0x56) {                        // identity response: mobile identity LV
    if (left < 2 || p[0] < 1 || p[0] > left - 1) return false;
    const uint32_t type = p[1] & 7u;
    if (type == 1) { bcd_digits(p + 1, p[0], 15, v); add(API_ID_IMSI, API_MSG_ID_RES, v); return true; }
    if (type == 2) { bcd_digits(p + 1, p[0], 15, v); add(API_ID_IMEI, API_MSG_ID_RES, v); return true; }
    if (type == 3) { bcd_digits(p + 1, p[0], 16, v); add(API_ID_IMEISV, API_MSG_ID_RES, v); return true; }
    return false;
  }
  if (msg_type == 0x41) {                        // attach request: NAS key set identifier | attach type, EPS mobile identity LV
    if (left < 3) return false;
    p += 1; left -= 1;
    if (p[0] < 1 || p[0] > left - 1) return false;
    const uint32_t type = p[1] & 7u;
    if (type == 1) { bcd_digits(p + 1, p[0], 15, v); add(API_ID_IMSI, API_MSG_ATT_REQ, v); return true; }
    if (type == 6) {
      if (p[0] < 11) return false;
      const uint32_t m_tmsi = ((uint32_t)p[8] << 24) | ((uint32_t)p[9] << 16) | ((uint32_t)p[10] << 8) | p[11];
      std::snprintf(v, sizeof(v), "%x", m_tmsi);
      add(API_ID_TMSI, API_MSG_ATT_REQ, v);
      return true;
    }
    if (type == 3) { bcd_digits(p + 1, p[0], 15, v); add(API_ID_IMEI, API_MSG_ATT_REQ, v); return true; }
    return false;
  }
  return false;
}

// UPER OCTET STRING with an unconstrained length: -> pointer / length of the content when it is octet aligned in the buffer, else a copy
bool octet_string(BitReader& b, std::vector<uint8_t>& out)
{
  const uint32_t n = b.length();
  if (b.err || b.pos + 8u * n > b.nbits) return false;
  out.resize(n);
  for (uint32_t i = 0; i < n; i++) out[i] = (uint8_t)b.get(8);
  return !b.err;
}

// PUSCH_Decoder::decode_ul_dcch (UL_Sniffer_PUSCH.cc:95-143): UL-DCCH-Message, TS 36.331 6.2.1 / 6.2.2.  The reference unpacks the whole message;
// here the head is walked as far as the dedicatedInfoNAS (the body of a UECapabilityInformation is not unpacked)
bool ul_dcch(int api_mode, const uint8_t* rrc, int len, uint16_t rnti, uint32_t tti, ApiEvent* ev, int cap, int& n)
{
  BitReader b{rrc, len > 0 ? 8u * (uint32_t)len : 0u};
  if (b.flag()) return false;                    // messageClassExtension
  const uint32_t type = b.get(4);
  if (b.err) return false;
  if (type == 7 && (api_mode == 1 || api_mode == 3)) {  // ueCapabilityInformation
    b.get(2);
    if (b.flag() || b.err) return false;         // criticalExtensions: c1
    if (n < cap) {
      ApiEvent& e = ev[n++];
      e.tti = tti; e.rnti = rnti; e.id_type = API_ID_NONE; e.msg_type = API_MSG_UE_CAP;
      std::snprintf(e.value, sizeof(e.value), "-");
    }
    return true;
  }
  std::vector<uint8_t> nas;
  if (type == 4 && (api_mode == 2 || api_mode == 3)) {  // rrcConnectionSetupComplete
    b.get(2);                                    // rrc-TransactionIdentifier
    if (b.flag()) return false;                  // criticalExtensions: c1
    if (b.get(2) != 0) return false;             // rrcConnectionSetupComplete-r8
    const bool mme = b.flag();
    b.flag();                                    // nonCriticalExtension (behind the NAS container)
    b.integer(1, 6);                             // selectedPLMN-Identity
    if (mme) {                                   // registeredMME
      if (b.flag()) {                            // plmn-Identity
        if (b.flag()) { for (int i = 0; i < 3; i++) b.integer(0, 9); }
        const uint32_t nd = b.get(1) + 2;
        for (uint32_t i = 0; i < nd; i++) b.integer(0, 9);
      }
      b.get(16); b.get(8);                       // mmegi, mmec
    }
    if (!octet_string(b, nas)) return false;
    return nas_ul_identity(nas.data(), (int)nas.size(), rnti, tti, ev, cap, n);
  }
  if (type == 9 && (api_mode == 2 || api_mode == 3)) {  // ulInformationTransfer
    if (b.flag()) return false;                  // criticalExtensions: c1
    if (b.get(2) != 0) return false;             // ulInformationTransfer-r8
    b.flag();                                    // nonCriticalExtension
    if (b.get(2) != 0) return false;             // dedicatedInfoType: dedicatedInfoNAS
    if (!octet_string(b, nas)) return false;
    return nas_ul_identity(nas.data(), (int)nas.size(), rnti, tti, ev, cap, n);
  }
  return false;
}
}  // namespace

bool api_ul_dcch_events(int api_mode, const uint8_t* pdu, int len, uint16_t rnti, uint32_t tti, ApiEvent* ev, int cap, int* nev)
{
  int n = 0;
  if (nev) *nev = 0;
  if (api_mode <= 0) return false;
  MacSubheader sub[10];
  const int ns = mac_ulsch_parse(pdu, len, sub, 10);
  bool ok = false;  // api_ret: the verdict of the last SRB SDU that reached decode_ul_dcch
  for (int i = 0; i < ns; i++) {
    if (!(sub[i].is_sdu && (sub[i].lcid == 1 || sub[i].lcid == 2))) continue;
    const uint8_t* p = pdu + sub[i].off;
    int left = (int)sub[i].len;
    if (left < 3 || !(p[0] & 0x80u)) continue;   // RLC AM control PDU (rlc_am_is_control_pdu)
    // rlc_am_read_data_pdu_header [srsRAN lib/src/rlc/rlc_am_lte.cc]: D/C RF P FI(2) E SN(10); only whole, unsegmented SDUs are read
    const bool rf = p[0] & 0x40u, ext = p[0] & 0x04u;
    const uint32_t fi = (p[0] >> 3) & 3u;
    int hdr = 2;
    if (rf) continue;
    if (ext) {                                   // E / LI chain, 12 bits per entry, padded to an octet
      int bitpos = 16;
      bool more = true;
      while (more) {
        if ((bitpos + 12 + 7) / 8 > left) { hdr = -1; break; }
        more = (p[bitpos >> 3] >> (7 - (bitpos & 7))) & 1u;
        bitpos += 12;
      }
      if (hdr < 0) continue;
      hdr = (bitpos + 7) / 8;
    }
    if (fi != 0 || left < hdr + 2) continue;
    p += hdr; left -= hdr;
    // one PDCP header octet is dropped; the reference's NAS security-header test reads that octet (buffer-layout cast, :349-354): its high
    // nibble must be 0, 1 or 3 - true for every SRB PDCP sequence number
    const uint32_t nib = p[0] >> 4;
    if (!(nib == 0 || nib == 1 || nib == 3)) continue;
    ok = ul_dcch(api_mode, p + 1, left - 1, rnti, tti, ev, cap, n);
  }
  if (nev) *nev = n;
  return ok;
}

// ------------------------------------------------------------------------------------------------ RRCConnectionReconfiguration -> NAS attach accept -> GUTI
// PDSCH_Decoder::decode_rrc_connection_reconfig (DL_Sniffer_PDSCH.cc:181-220): DL-DCCH-Message, c1 = rrcConnectionReconfiguration-r8, the FIRST
// dedicatedInfoNAS; liblte_mme_parse_msg_header + liblte_mme_unpack_attach_accept_msg [srsRAN, not in tree; TS 24.301 8.2.1, 9.9.3.12] -> M-TMSI of
// the GUTI.  UPER walk per TS 36.331 6.2.2 / 6.3.5: the components in FRONT of the NAS list must be walked to find it - measConfig (object /
// report / id lists of every RAT, quantity configuration, gaps, s-Measure, HRPD pre-registration, speed-state parameters; Rel-8 roots, later additions
// stepped over) is; mobilityControlInfo is not: a handover command never carries an attach accept (dedicatedInfoNASList is conditional on "nonHO",
// 36.331 6.2.2), so such a message has no identity for the reference either.
// Pinned by the two RRCConnectionReconfiguration messages of the reference's own captures (tests/test_rrc_oracle.py).
static void thresholdEutra(BitReader& b) { if (!b.flag()) b.get(7); else b.get(6); }  // CHOICE { threshold-RSRP (0..97), threshold-RSRQ (0..34) }
// X.691 10.9 / 19.7-19.9: the extension additions behind the root components of an extensible SEQUENCE - a presence bitmap whose length is a
// "normally small" number, then every present addition (group) as an open type: length determinant + that many octets, which are skipped unread
static bool skipOpenType(BitReader& b)
{
  uint32_t n = b.get(8);
  if (n & 0x80u) {
    if (n & 0x40u) return false;               // fragmented: never in these messages
    n = ((n & 0x3Fu) << 8) | b.get(8);
  }
  if (b.err || b.pos + 8u * n > b.nbits) { b.err = true; return false; }
  b.pos += 8u * n;
  return true;
}
static bool skipExtAdditions(BitReader& b)
{
  if (b.flag()) return false;                  // more than 64 additions
  const uint32_t n = b.get(6) + 1;
  uint64_t present = 0;                        // (two sequenced reads: the reader is stateful)
  if (n > 32) present = (uint64_t)b.get(n - 32) << 32;
  present |= b.get(n > 32 ? 32 : n);
  for (uint32_t i = 0; i < n && !b.err; i++)
    if ((present >> (n - 1 - i)) & 1u)
      if (!skipOpenType(b)) return false;
  return !b.err;
}
// Inter-RAT measurement objects and report configurations in front of the NAS list (round-4 review, missing 4): restated from TS 36.331 6.3.5 (Rel-8
// roots; what later releases added travels behind the extension marker and is stepped over unread).  Not pinned by any capture the reference ships
// (its two reconfiguration messages configure E-UTRA measurements only): the test's independent encoder is the check.
//   MeasObjectUTRA  { carrierFreq (0..16383), offsetFreq (-15..15) DEFAULT 0, cellsToRemoveList OPTIONAL, cellsToAddModList CHOICE { FDD: (cellIndex, physCellId
//                     0..511), TDD: (cellIndex, physCellId 0..127) } OPTIONAL, cellForWhichToReportCGI CHOICE { FDD, TDD } OPTIONAL, ... }
//   MeasObjectGERAN { carrierFreqs { startingARFCN (0..1023), bandIndicator, followingARFCNs CHOICE { explicit list (0..31 x 10 bits), equally spaced (1..8, 0..31),
//                     variable bit map (1..16 octets) } }, offsetFreq DEFAULT 0, ncc-Permitted BIT STRING (8) DEFAULT, cellForWhichToReportCGI { ncc (3), bcc (3) } OPTIONAL, ... }
//   MeasObjectCDMA2000 { cdma2000-Type, carrierFreq { bandClass (32 values, extensible), arfcn (0..2047) }, searchWindowSize (0..15) OPTIONAL, offsetFreq DEFAULT 0,
//                     cellsToRemoveList OPTIONAL, cellsToAddModList ((cellIndex, physCellId 0..511)) OPTIONAL, cellForWhichToReportCGI (0..511) OPTIONAL, ... }
static bool measObjectInterRatSkip(BitReader& b, uint32_t rat)
{
  const bool ext = b.flag();
  if (rat == 1) {
    bool o[4];
    for (bool& x : o) x = b.flag();
    b.get(14);
    if (o[0]) b.get(5);
    if (o[1]) { const uint32_t k = b.get(5) + 1; b.get(5 * k); }
    if (o[2]) { const bool tdd = b.flag(); const uint32_t k = b.get(5) + 1; for (uint32_t j = 0; j < k && !b.err; j++) { b.get(5); b.get(tdd ? 7 : 9); } }
    if (o[3]) { const bool tdd = b.flag(); b.get(tdd ? 7 : 9); }
  } else if (rat == 2) {
    bool o[3];
    for (bool& x : o) x = b.flag();
    b.get(10); b.get(1);
    switch (b.get(2)) {
      case 0: { const uint32_t k = b.get(5); for (uint32_t j = 0; j < k && !b.err; j++) b.get(10); break; }
      case 1: b.get(3); b.get(5); break;
      case 2: { const uint32_t k = b.get(4) + 1; for (uint32_t j = 0; j < k && !b.err; j++) b.get(8); break; }
      default: return false;
    }
    if (o[0]) b.get(5);
    if (o[1]) b.get(8);
    if (o[2]) b.get(6);
  } else {
    bool o[5];
    for (bool& x : o) x = b.flag();
    b.get(1);
    if (b.flag()) return false;                   // bandClass behind its extension marker
    b.get(5); b.get(11);
    if (o[0]) b.get(4);
    if (o[1]) b.get(5);
    if (o[2]) { const uint32_t k = b.get(5) + 1; b.get(5 * k); }
    if (o[3]) { const uint32_t k = b.get(5) + 1; for (uint32_t j = 0; j < k && !b.err; j++) { b.get(5); b.get(9); } }
    if (o[4]) b.get(9);
  }
  if (ext && !skipExtAdditions(b)) return false;
  return !b.err;
}
//   ReportConfigInterRAT { triggerType CHOICE { event { eventId CHOICE { eventB1 { b1-Threshold CHOICE { UTRA, GERAN, CDMA2000 } }, eventB2 { b2-Threshold1 ThresholdEUTRA,
//                     b2-Threshold2 CHOICE { UTRA, GERAN, CDMA2000 } }, ... }, hysteresis, timeToTrigger }, periodical { purpose (3 values) } }, maxReportCells, reportInterval,
//                     reportAmount, ... };  ThresholdUTRA CHOICE { utra-RSCP (-5..91), utra-EcN0 (0..49) }, ThresholdGERAN / ThresholdCDMA2000 (0..63)
static bool thresholdInterRat(BitReader& b)
{
  switch (b.get(2)) {
    case 0: if (!b.flag()) b.get(7); else b.get(6); return true;
    case 1: case 2: b.get(6); return true;
    default: return false;
  }
}
static bool reportConfigInterRatSkip(BitReader& b)
{
  const bool ext = b.flag();
  if (!b.flag()) {                                // event
    if (b.flag()) {                               // an event behind the extension marker: small index, then an open type
      if (b.flag()) return false;
      b.get(6);
      if (!skipOpenType(b)) return false;
    } else if (!b.flag()) {                       // b1
      if (!thresholdInterRat(b)) return false;
    } else {                                      // b2
      thresholdEutra(b);
      if (!thresholdInterRat(b)) return false;
    }
    b.get(5); b.get(4);
  } else {
    b.get(2);                                     // periodical: purpose
  }
  b.get(3); b.get(4); b.get(3);
  if (ext && !skipExtAdditions(b)) return false;
  return !b.err;
}
static bool measConfigSkip(BitReader& b)
{
  const bool ext_mc = b.flag();
  bool opt[11];
  for (bool& o : opt) o = b.flag();
  if (opt[0]) { const uint32_t n = b.get(5) + 1; b.get(5 * n); }  // measObjectToRemoveList: (1..32) each
  if (opt[1]) {
    const uint32_t n = b.get(5) + 1;
    for (uint32_t i = 0; i < n && !b.err; i++) {
      b.get(5);                                   // measObjectId
      if (b.flag()) return false;                 // measObject: an alternative behind the extension marker
      const uint32_t rat = b.get(2);              // measObjectEUTRA, measObjectUTRA, measObjectGERAN, measObjectCDMA2000
      if (rat != 0) { if (!measObjectInterRatSkip(b, rat)) return false; continue; }
      const bool ext_mo = b.flag();               // MeasObjectEUTRA: extension additions (r10+) follow the root components
      bool o[6];
      for (bool& x : o) x = b.flag();
      b.get(16); b.get(3); b.get(1); b.get(2);    // carrierFreq, allowedMeasBandwidth, presenceAntennaPort1, neighCellConfig
      if (o[0]) b.get(5);                         // offsetFreq
      if (o[1]) { const uint32_t k = b.get(5) + 1; b.get(5 * k); }                       // cellsToRemoveList
      if (o[2]) { const uint32_t k = b.get(5) + 1; for (uint32_t j = 0; j < k; j++) { b.get(5); b.get(9); b.get(5); } }  // cellsToAddModList
      if (o[3]) { const uint32_t k = b.get(5) + 1; b.get(5 * k); }                       // blackCellsToRemoveList
      if (o[4]) {                                 // blackCellsToAddModList: cellIndex + PhysCellIdRange { start, range OPTIONAL }
        const uint32_t k = b.get(5) + 1;
        for (uint32_t j = 0; j < k; j++) { b.get(5); const bool r = b.flag(); b.get(9); if (r) b.get(4); }
      }
      if (o[5]) b.get(9);                         // cellForWhichToReportCGI
      if (ext_mo && !skipExtAdditions(b)) return false;
    }
  }
  if (opt[2]) { const uint32_t n = b.get(5) + 1; b.get(5 * n); }  // reportConfigToRemoveList
  if (opt[3]) {
    const uint32_t n = b.get(5) + 1;
    for (uint32_t i = 0; i < n && !b.err; i++) {
      b.get(5);                                   // reportConfigId
      if (b.get(1) != 0) { if (!reportConfigInterRatSkip(b)) return false; continue; }  // reportConfigInterRAT
      const bool ext_rc = b.flag();               // ReportConfigEUTRA extension additions (r9+) at the end
      if (!b.flag()) {                            // triggerType: event
        if (b.flag()) {                           // eventId beyond a5 (a6-r10 ...): index as a normally small number, the event as an open type
          if (b.flag()) return false;
          b.get(6);
          if (!skipOpenType(b)) return false;
        } else {
          switch (b.get(3)) {
            case 0: case 1: case 3: thresholdEutra(b); break;   // a1, a2, a4
            case 2: b.get(6); b.get(1); break;                   // a3: offset (-30..30), reportOnLeave
            case 4: thresholdEutra(b); thresholdEutra(b); break; // a5
            default: return false;
          }
        }
        b.get(5); b.get(4);                       // hysteresis, timeToTrigger
      } else {
        b.get(1);                                 // periodical: purpose
      }
      b.get(1); b.get(1); b.get(3); b.get(4); b.get(3);  // triggerQuantity, reportQuantity, maxReportCells, reportInterval, reportAmount
      if (ext_rc && !skipExtAdditions(b)) return false;
    }
  }
  if (opt[4]) { const uint32_t n = b.get(5) + 1; b.get(5 * n); }   // measIdToRemoveList
  if (opt[5]) { const uint32_t n = b.get(5) + 1; b.get(15 * n); }  // measIdToAddModList: measId, measObjectId, reportConfigId
  if (opt[6]) {                                   // quantityConfig
    const bool ext_qc = b.flag();
    bool q[4];
    for (bool& x : q) x = b.flag();
    if (q[0]) { const bool r1 = b.flag(), r2 = b.flag(); for (bool r : {r1, r2}) if (r) { if (b.flag()) return false; b.get(4); } }  // filterCoefficientRSRP / RSRQ
    // QuantityConfigUTRA { measQuantityUTRA-FDD (2 values), measQuantityUTRA-TDD (1 value: no bits), filterCoefficient DEFAULT fc4 },
    // QuantityConfigGERAN { measQuantityGERAN (1 value), filterCoefficient DEFAULT fc2 }, QuantityConfigCDMA2000 { measQuantityCDMA2000 (2 values) }
    if (q[1]) { const bool fc = b.flag(); b.get(1); if (fc) { if (b.flag()) return false; b.get(4); } }
    if (q[2]) { const bool fc = b.flag(); if (fc) { if (b.flag()) return false; b.get(4); } }
    if (q[3]) b.get(1);
    if (ext_qc && !skipExtAdditions(b)) return false;
  }
  if (opt[7] && b.flag()) {                       // measGapConfig: release / setup { gapOffset CHOICE { gp0 (0..39), gp1 (0..79), ... } }
    if (b.flag()) return false;                   // the choice is extensible: an alternative of a later release is not walked
    if (!b.flag()) b.get(6); else b.get(7);
  }
  if (opt[8]) b.get(7);                           // s-Measure (0..97)
  if (opt[9]) {                                   // preRegistrationInfoHRPD { preRegistrationAllowed, preRegistrationZoneId (0..255) OPTIONAL, secondaryPreRegistrationZoneIdList (1..2) OPTIONAL }
    const bool z = b.flag(), sl = b.flag();
    b.get(1);
    if (z) b.get(8);
    if (sl) { const uint32_t k = b.get(1) + 1; b.get(8 * k); }
  }
  if (opt[10] && b.flag()) b.get(3 + 3 + 4 + 4 + 2 + 2);  // speedStatePars setup: MobilityStateParameters (t-Evaluation, t-HystNormal, n-CellChangeMedium, n-CellChangeHigh) + SpeedStateScaleFactors
  if (ext_mc && !skipExtAdditions(b)) return false;
  return !b.err;
}

bool rrc_reconfig_attach_accept_tmsi(const uint8_t* sdu, int len, uint32_t& m_tmsi)
{
  if (len < 4) return false;
  BitReader b{sdu, (uint32_t)len * 8u};
  if (b.flag() || b.get(4) != 4) return false;    // DL-DCCH-MessageType c1, rrcConnectionReconfiguration
  b.get(2);                                       // rrc-TransactionIdentifier
  if (b.flag() || b.get(3) != 0) return false;    // criticalExtensions c1, rrcConnectionReconfiguration-r8
  bool o[6];
  for (bool& x : o) x = b.flag();                 // measConfig, mobilityControlInfo, dedicatedInfoNASList, radioResourceConfigDedicated, securityConfigHO, nonCriticalExtension
  if (!o[2] || o[1]) return false;
  if (o[0] && !measConfigSkip(b)) return false;
  b.get(4);                                       // number of NAS messages - 1; the reference looks at the first one only
  const uint32_t n = b.length();
  if (b.err || n < 8 || b.pos + 8u * n > b.nbits) return false;
  uint8_t nas[256];
  if (n > sizeof(nas)) return false;
  for (uint32_t i = 0; i < n; i++) nas[i] = (uint8_t)b.get(8);
  // liblte_mme_parse_msg_header: plain NAS -> octet 1 is the message type; integrity protected (security header type 1..4) -> MAC (4) + sequence
  // number (1) in front of the plain message
  const uint32_t sht = nas[0] >> 4;
  uint32_t p = 0;
  if (sht == 0) p = 0;
  else if (sht <= 4) p = 6;
  else return false;
  if (p + 2 > n || (nas[p] & 0x0Fu) != 0x07u || nas[p + 1] != 0x42u) return false;  // EPS mobility management, attach accept
  p += 2;
  p += 2;                                         // EPS attach result + spare half octet, T3412
  if (p >= n) return false;
  const uint32_t tai_len = nas[p];                // TAI list (LV, 6..96 octets)
  if (tai_len < 6 || tai_len > 96) return false;
  p += 1 + tai_len;
  if (p + 2 > n) return false;
  const uint32_t esm_len = ((uint32_t)nas[p] << 8) | nas[p + 1];  // ESM message container (LV-E)
  p += 2 + esm_len;
  if (p + 13 > n || nas[p] != 0x50u) return false;  // GUTI IE (IEI 0x50) is the first optional element
  if (nas[p + 1] != 11 || (nas[p + 2] & 0x07u) != 6u) return false;  // EPS mobile identity of type GUTI
  m_tmsi = ((uint32_t)nas[p + 9] << 24) | ((uint32_t)nas[p + 10] << 16) | ((uint32_t)nas[p + 11] << 8) | nas[p + 12];
  return true;
}

bool api_dl_events(int api_mode, char name, const uint8_t* pdu, int len, uint16_t rnti, uint32_t tti, ApiEvent* ev, int cap, int* nev)
{
  int n = 0;
  bool to_pcap = false;
  auto add = [&](uint16_t r, uint32_t id, uint32_t msg, const char* v) {
    if (n >= cap) return;
    ApiEvent& e = ev[n++];
    e.tti = tti; e.rnti = r; e.id_type = id; e.msg_type = msg;
    std::snprintf(e.value, sizeof(e.value), "%s", v);
  };
  if (name == 'P' && (api_mode == 2 || api_mode == 3)) {  // IMSI catching from paging, :805-812 + :84-127
    PagingId rec[16];
    const int nr = paging_decode(pdu, len, rec, 16);
    for (int i = 0; i < nr; i++) {
      char v[24] = {0};
      if (rec[i].is_imsi) {  // the reference prints the first 15 digits
        for (uint32_t k = 0; k < 15 && k < rec[i].nof_digits; k++) v[k] = (char)('0' + rec[i].digits[k]);
        add(65534, API_ID_IMSI, API_MSG_PAGING, v);
      } else {
        std::snprintf(v, sizeof(v), "%08x", rec[i].m_tmsi);
        add(65534, API_ID_TMSI, API_MSG_PAGING, v);
      }
      to_pcap = true;
    }
  }
  if (name == 'C' && (api_mode == 0 || api_mode == 3)) {  // identity mapping: contention resolution identity next to an RRCConnectionSetup, :813-877
    MacSubheader sub[20];
    const int ns = mac_dlsch_parse(pdu, len, sub, 20);
    bool setup = false, found = false;
    int seen[10], nseen = 0;
    for (int i = 0; i < ns; i++) {
      if (sub[i].is_sdu && sub[i].lcid == 0) {
        UeSpecConfig c;
        if (rrc_conn_setup_decode(pdu + sub[i].off, (int)sub[i].len, c)) setup = true;
      } else if (sub[i].is_sdu && sub[i].lcid == 1) {
        // SRB1: RLC AM header (2) + PDCP sequence number (1) assumed in front of the DL-DCCH message (DL_Sniffer_PDSCH.cc:836-838)
        uint32_t tmsi = 0;
        if (sub[i].len > 3 && rrc_reconfig_attach_accept_tmsi(pdu + sub[i].off + 3, (int)sub[i].len - 3, tmsi)) {
          char v[24];
          std::snprintf(v, sizeof(v), "%08x", tmsi);
          add(rnti, API_ID_TMSI, API_MSG_CON_RECONFIG, v);
        }
      } else {
        if (nseen < 10) seen[nseen++] = i; else break;
      }
      if (setup) {
        for (int h = 0; h < nseen && !found; h++)
          if (sub[seen[h]].lcid == 28 && sub[seen[h]].len == 6) {
            unsigned long long id = 0;
            for (int k = 0; k < 6; k++) id = (id << 8) | pdu[sub[seen[h]].off + k];
            char hex[24];
            std::snprintf(hex, sizeof(hex), "%llx", id);  // printed without leading zeros; characters 3..10 are reported (:865-866)
            const size_t L = std::strlen(hex);
            char v[24] = {0};
            if (L >= 3) std::snprintf(v, sizeof(v), "%.8s", hex + 3);
            add(rnti, API_ID_CON_RES, API_MSG_CON_SET, v);
            found = true;
          }
        to_pcap = true;  // write_dl_crnti_api (once per block here)
      }
    }
  }
  if (nev) *nev = n;
  return to_pcap;
}

}  // namespace lsn
