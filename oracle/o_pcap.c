/* o_pcap.c - ORACLE (test infrastructure only): MAC-LTE (DLT 147) pcap records.
 * Follows LTESniffer_pcap_writer::pack_and_write (/root/reference/src/src/PcapWriter.cc:93-118) and the
 * framing observed in /root/reference/pcap_file_example/*.pcap; srsRAN's
 * LTE_PCAP_MAC_WritePDU itself is not in the tree.  Pinned by tests/test_oracle_pcap.py against those files. */
#include "lsn_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct o_pcap {
  FILE* f;
  uint8_t* mem;
  size_t len, cap;
  uint32_t nrec;
};

static void put(o_pcap_t* p, const void* d, size_t n)
{
  if (p->f) {
    fwrite(d, 1, n, p->f);
    return;
  }
  if (p->len + n > p->cap) {
    p->cap = (p->len + n) * 2 + 4096;
    p->mem = (uint8_t*)realloc(p->mem, p->cap);
  }
  memcpy(p->mem + p->len, d, n);
  p->len += n;
}

static void header(o_pcap_t* p)
{
  const uint32_t h[6] = {0xa1b2c3d4u, 0x00040002u, 0, 0, 65535, 147};
  put(p, h, sizeof(h));
}

o_pcap_t* o_pcap_open_mem(void)
{
  o_pcap_t* p = (o_pcap_t*)calloc(1, sizeof(*p));
  header(p);
  return p;
}
o_pcap_t* o_pcap_open_file(const char* path)
{
  o_pcap_t* p = (o_pcap_t*)calloc(1, sizeof(*p));
  p->f = fopen(path, "wb");
  if (!p->f) {
    free(p);
    return NULL;
  }
  header(p);
  return p;
}

void o_pcap_write(o_pcap_t* p, const uint8_t* pdu, uint32_t len, uint32_t tti, uint16_t rnti, uint8_t direction,
                  uint8_t rnti_type, uint8_t crc_ok, uint32_t ts_sec, uint32_t ts_usec)
{
  uint8_t c[23];
  uint16_t sfn = (uint16_t)(tti / 10), sf = (uint16_t)(tti % 10), fs = (uint16_t)((sfn << 4) | sf);
  c[0] = 1; /* FDD_RADIO */
  c[1] = direction;
  c[2] = rnti_type;
  c[3] = 0x02; c[4] = (uint8_t)(rnti >> 8); c[5] = (uint8_t)rnti;
  c[6] = 0x03; c[7] = 0; c[8] = 0; /* ue_id 0 */
  c[9] = 0x04; c[10] = (uint8_t)(fs >> 8); c[11] = (uint8_t)fs;
  c[12] = 0x07; c[13] = crc_ok;
  c[14] = 0x0a; c[15] = 0;
  c[16] = 0x0f; c[17] = 0;
  c[18] = 0x01;
  uint32_t rec[4] = {ts_sec, ts_usec, len + 19, len + 19};
  put(p, rec, sizeof(rec));
  put(p, c, 19);
  put(p, pdu, len);
  p->nrec++;
}
const uint8_t* o_pcap_mem(o_pcap_t* p, size_t* len)
{
  *len = p->len;
  return p->mem;
}
uint32_t o_pcap_nof_records(o_pcap_t* p) { return p->nrec; }
void o_pcap_close(o_pcap_t* p)
{
  if (!p) return;
  if (p->f) fclose(p->f);
  free(p->mem);
  free(p);
}
