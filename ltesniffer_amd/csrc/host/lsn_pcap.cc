// lsn_pcap.cc - MAC-LTE (DLT 147) pcap writer, the native consumer of the PDU sink.
// Replaces LTESniffer_pcap_writer (/root/reference/src/src/PcapWriter.cc:75-118: open, pack_and_write) and the srsRAN
// primitives it calls (DLT_PCAP_Open, LTE_PCAP_MAC_WritePDU [srsRAN pcap.h, not in tree]); record layout pinned by the
// reference's example captures (pcap_file_example/*.pcap).
#include "../../../include/ltesniffer_amd.h"
#include <cstdio>
#include <cstring>
#include <mutex>
#include <sys/time.h>
#include <vector>

struct lsn_pcap {
  FILE* f = nullptr;
  std::vector<uint8_t> mem;
  bool to_mem = false;
  bool wall_clock = true;
  uint32_t nrec = 0;
  bool store = true;            // false: records are only counted and digested (long benchmark streams)
  uint64_t digest = 0x9E3779B97F4A7C15ull, nbytes = 0;
  std::mutex mtx;  // PcapWriter.h:53
  void put(const void* d, size_t n)
  {
    if (!store) return;
    if (to_mem) mem.insert(mem.end(), (const uint8_t*)d, (const uint8_t*)d + n);
    else if (f) fwrite(d, 1, n, f);
  }
  // order-sensitive 64-bit digest of the record stream without the timestamps (record length, 19-byte MAC-LTE context, PDU):
  // 8 bytes per multiply, the stream position is mixed in through the chaining
  // per-block digests (lsn_pcap_set_digest_blocks): the stream cut every `blk_period` subframes from the TTI `blk_origin`, each block
  // hashed on its own chain - a long stream is then comparable piecewise with a reference that was produced elsewhere
  struct Blk { uint64_t digest = 0x9E3779B97F4A7C15ull, nbytes = 0; uint32_t nrec = 0; };
  std::vector<Blk> blks;
  uint32_t blk_period = 0, blk_origin = 0;
  int64_t blk_last = -1;   // unwrapped subframe number (since the origin) of the latest record
  void mix(const uint8_t* d, size_t n) { mixInto(digest, nbytes, d, n); }
  static void mixInto(uint64_t& digest, uint64_t& nbytes, const uint8_t* d, size_t n)
  {
    uint64_t h = digest;
    size_t i = 0;
    if (n >= 64) {  // four independent multiply chains over 32-byte strides (a PDU is a few hundred bytes), folded into the running value
      uint64_t a = h, b = h ^ 0x9E3779B97F4A7C15ull, c = h + 0x632BE59BD9B4E019ull, e = ~h;
      for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        std::memcpy(w, d + i, 32);
        a = (a ^ w[0]) * 0x100000001B3ull; b = (b ^ w[1]) * 0xC2B2AE3D27D4EB4Full; c = (c ^ w[2]) * 0x165667B19E3779F9ull; e = (e ^ w[3]) * 0x27D4EB2F165667C5ull;
      }
      h = ((a ^ (b >> 31)) * 0x100000001B3ull) ^ ((c ^ (e >> 29)) * 0xC2B2AE3D27D4EB4Full);
      h ^= h >> 32;
    }
    for (; i + 8 <= n; i += 8) { uint64_t w; std::memcpy(&w, d + i, 8); h = (h ^ w) * 0x100000001B3ull; h ^= h >> 29; }
    uint64_t w = 0;
    std::memcpy(&w, d + i, n - i);
    h = (h ^ w ^ ((uint64_t)n << 56)) * 0x100000001B3ull; h ^= h >> 32;
    digest = h;
    nbytes += n;
  }
  void header()
  {
    const uint32_t h[6] = {0xa1b2c3d4u, 0x00040002u, 0, 0, 65535, 147};  // MAC_LTE_DLT 147, PcapWriter.cc:77
    put(h, sizeof(h));
  }
};

extern "C" {

lsn_pcap_t* lsn_pcap_open(const char* path)
{
  if (!path) return nullptr;
  lsn_pcap* p = new lsn_pcap();
  p->f = fopen(path, "wb");
  if (!p->f) { delete p; return nullptr; }
  setvbuf(p->f, nullptr, _IOFBF, (size_t)4 << 20);  // records are a few hundred bytes each: one write() per 4 MB instead of one per 4 KB (0.7-1.6 GB/s of records at replay speed)
  p->header();
  return p;
}

lsn_pcap_t* lsn_pcap_open_mem(void)
{
  lsn_pcap* p = new lsn_pcap();
  p->to_mem = true;
  p->mem.reserve((size_t)256 << 20);  // address space only: no re-allocation copies in the commit turn while a capture grows
  p->wall_clock = false;  // deterministic captures: timestamps are zero
  p->header();
  return p;
}

void lsn_pcap_set_wall_clock(lsn_pcap_t* p, int on) { if (p) p->wall_clock = on != 0; }

int lsn_pcap_write(lsn_pcap_t* p, const lsn_pdu_ctx_t* c, const uint8_t* pdu, uint32_t len)
{
  if (!p || !c || (!pdu && len)) return LSN_ERROR_INVALID_INPUTS;
  uint8_t h[19];
  const uint16_t fs = (uint16_t)((((c->tti / 10) & 0xFFF) << 4) | (c->tti % 10));  // PcapWriter.cc:102-103
  h[0] = 1;  // FDD_RADIO, PcapWriter.cc:108
  h[1] = c->direction;
  h[2] = c->rnti_type;
  h[3] = 0x02; h[4] = (uint8_t)(c->rnti >> 8); h[5] = (uint8_t)c->rnti;
  h[6] = 0x03; h[7] = 0; h[8] = 0;
  h[9] = 0x04; h[10] = (uint8_t)(fs >> 8); h[11] = (uint8_t)fs;
  h[12] = 0x07; h[13] = c->crc_ok;
  h[14] = 0x0a; h[15] = 0;
  h[16] = 0x0f; h[17] = 0;
  h[18] = 0x01;
  uint32_t rec[4] = {0, 0, len + 19, len + 19};
  if (p->wall_clock) {
    struct timeval t;
    gettimeofday(&t, nullptr);
    rec[0] = (uint32_t)t.tv_sec; rec[1] = (uint32_t)t.tv_usec;
  }
  std::lock_guard<std::mutex> lk(p->mtx);
  p->put(rec, sizeof(rec));
  p->put(h, sizeof(h));
  p->put(pdu, len);
  p->mix((const uint8_t*)&rec[2], 4);
  p->mix(h, sizeof(h));
  p->mix(pdu, len);
  p->nrec++;
  if (p->blk_period) {
    // the TTI counts modulo 10240: unwrap against the latest record (uplink records may step back a few subframes)
    const int64_t ref = p->blk_last < 0 ? 0 : p->blk_last;
    const int64_t want = ((int64_t)c->tti - (int64_t)p->blk_origin + 10240) % 10240;
    int64_t d = (want - ref % 10240 + 10240 + 5120) % 10240 - 5120;
    int64_t a = ref + d;
    if (a < 0) a = 0;
    if (a > p->blk_last) p->blk_last = a;
    const size_t bi = (size_t)(a / p->blk_period);
    if (p->blks.size() <= bi) p->blks.resize(bi + 1);
    lsn_pcap::Blk& b = p->blks[bi];
    lsn_pcap::mixInto(b.digest, b.nbytes, (const uint8_t*)&rec[2], 4);
    lsn_pcap::mixInto(b.digest, b.nbytes, h, sizeof(h));
    lsn_pcap::mixInto(b.digest, b.nbytes, pdu, len);
    b.nrec++;
  }
  return LSN_SUCCESS;
}

void lsn_pcap_sink(void* user, const lsn_pdu_ctx_t* ctx, const uint8_t* pdu, uint32_t len) { (void)lsn_pcap_write((lsn_pcap_t*)user, ctx, pdu, len); }

const uint8_t* lsn_pcap_mem(lsn_pcap_t* p, size_t* len)
{
  if (!p || !len) return nullptr;
  *len = p->mem.size();
  return p->mem.data();
}
uint32_t lsn_pcap_nof_records(lsn_pcap_t* p) { return p ? p->nrec : 0; }
void lsn_pcap_reset(lsn_pcap_t* p)
{
  if (!p) return;
  std::lock_guard<std::mutex> lk(p->mtx);
  if (p->to_mem) { p->mem.clear(); p->header(); }
  p->nrec = 0;
  p->digest = 0x9E3779B97F4A7C15ull; p->nbytes = 0;
  p->blks.clear(); p->blk_last = -1;
}
void lsn_pcap_set_digest_blocks(lsn_pcap_t* p, uint32_t subframes_per_block, uint32_t origin_tti)
{
  if (!p) return;
  std::lock_guard<std::mutex> lk(p->mtx);
  p->blk_period = subframes_per_block; p->blk_origin = origin_tti % 10240;
  p->blks.clear(); p->blk_last = -1;
}
uint32_t lsn_pcap_block_digests(lsn_pcap_t* p, uint64_t* digests, uint32_t* nof_records, uint32_t cap)
{
  if (!p) return 0;
  std::lock_guard<std::mutex> lk(p->mtx);
  const uint32_t n = (uint32_t)p->blks.size();
  for (uint32_t i = 0; i < n && i < cap; i++) {
    if (digests) digests[i] = p->blks[i].digest;
    if (nof_records) nof_records[i] = p->blks[i].nrec;
  }
  return n;
}
void lsn_pcap_set_store(lsn_pcap_t* p, int on) { if (p) p->store = on != 0; }
int lsn_pcap_digest(lsn_pcap_t* p, uint64_t* digest, uint64_t* nbytes)
{
  if (!p) return LSN_ERROR_INVALID_INPUTS;
  std::lock_guard<std::mutex> lk(p->mtx);
  if (digest) *digest = p->digest;
  if (nbytes) *nbytes = p->nbytes;
  return LSN_SUCCESS;
}
void lsn_pcap_close(lsn_pcap_t* p)
{
  if (!p) return;
  if (p->f) fclose(p->f);
  delete p;
}

}  // extern "C"
