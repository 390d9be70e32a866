/* o_trace.c - ORACLE (test infrastructure only): a recorder behind the data-plane stages of the restated receive chain, so that the GPU
 * parity tests can compare the product's stage-C taps (LSN_TAP_PDSCH_LLR16 / LSN_TAP_RM_WORDS / LSN_TAP_CB_RESULT) bit by bit instead of
 * observing the chain only through CRC verdicts and payload bytes.  What is recorded per srsran_ue_dl_decode_pdsch / srsran_pusch_decode
 * call of the worker (call sites /root/reference/src/src/DL_Sniffer_PDSCH.cc:997,1110,1207, UL_Sniffer_PUSCH.cc:262): the descrambled
 * int16 soft bits of both codewords, and per code block the de-rate-matched streams d0|d1|d2 (36.212 5.1.4.1.2 inverted), the number of
 * turbo iterations run and the block-CRC verdict.  Off by default; nothing here changes a result. */
#include "lsn_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef struct {
  o_trace_job_hdr_t h;
  int16_t* llr[2];
} tr_job_t;
typedef struct {
  o_trace_cb_hdr_t h;
  int16_t* d3;
} tr_cb_t;

static int g_on = 0, g_tb = 0;
static tr_job_t* g_jobs = NULL;
static tr_cb_t* g_cbs = NULL;
static uint32_t g_njobs = 0, g_capjobs = 0, g_ncbs = 0, g_capcbs = 0;

void o_trace_enable(int on)
{
  for (uint32_t i = 0; i < g_njobs; i++) { free(g_jobs[i].llr[0]); free(g_jobs[i].llr[1]); }
  for (uint32_t i = 0; i < g_ncbs; i++) free(g_cbs[i].d3);
  g_njobs = g_ncbs = 0;
  g_on = on;
}
int o_trace_enabled(void) { return g_on; }

void o_trace_begin_job(uint32_t tti, uint16_t rnti, uint32_t nof_re, const int* qm, const int16_t* llr0, const int16_t* llr1, int is_ul)
{
  if (!g_on) return;
  if (g_njobs == g_capjobs) { g_capjobs = g_capjobs ? 2 * g_capjobs : 256; g_jobs = (tr_job_t*)realloc(g_jobs, sizeof(tr_job_t) * g_capjobs); }
  tr_job_t* j = &g_jobs[g_njobs++];
  memset(j, 0, sizeof(*j));
  j->h.tti = tti; j->h.rnti = rnti; j->h.nof_re = nof_re; j->h.is_ul = (uint32_t)is_ul; j->h.cb_first = g_ncbs;
  const int16_t* src[2] = {llr0, llr1};
  for (int q = 0; q < 2; q++) {
    j->h.qm[q] = qm ? (uint32_t)qm[q] : 0u;
    j->h.llr_len[q] = (src[q] && qm && qm[q]) ? nof_re * (uint32_t)qm[q] : 0u;
    if (j->h.llr_len[q]) {
      j->llr[q] = (int16_t*)malloc(sizeof(int16_t) * j->h.llr_len[q]);
      memcpy(j->llr[q], src[q], sizeof(int16_t) * j->h.llr_len[q]);
    }
  }
  g_tb = 0;
}
void o_trace_set_tb(int tb) { g_tb = tb; }

void o_trace_cb(int K, int F, int E, int rv, const int16_t* d3, int iters, int ok)
{
  if (!g_on || g_njobs == 0) return;
  if (g_ncbs == g_capcbs) { g_capcbs = g_capcbs ? 2 * g_capcbs : 1024; g_cbs = (tr_cb_t*)realloc(g_cbs, sizeof(tr_cb_t) * g_capcbs); }
  tr_cb_t* c = &g_cbs[g_ncbs++];
  c->h.job = g_njobs - 1; c->h.tb = (uint32_t)g_tb; c->h.K = (uint32_t)K; c->h.F = (uint32_t)F; c->h.E = (uint32_t)E; c->h.rv = (uint32_t)rv;
  c->h.iters = (uint32_t)iters; c->h.ok = (uint32_t)ok;
  c->d3 = (int16_t*)malloc(sizeof(int16_t) * 3 * (size_t)(K + 4));
  memcpy(c->d3, d3, sizeof(int16_t) * 3 * (size_t)(K + 4));
  g_jobs[g_njobs - 1].h.ncb++;
}

uint32_t o_trace_njobs(void) { return g_njobs; }
uint32_t o_trace_ncbs(void) { return g_ncbs; }
int o_trace_job(uint32_t i, o_trace_job_hdr_t* out) { if (i >= g_njobs) return -1; *out = g_jobs[i].h; return 0; }
int o_trace_job_llr(uint32_t i, int cw, int16_t* out, uint32_t cap)
{
  if (i >= g_njobs || cw < 0 || cw > 1 || g_jobs[i].h.llr_len[cw] > cap) return -1;
  if (g_jobs[i].h.llr_len[cw]) memcpy(out, g_jobs[i].llr[cw], sizeof(int16_t) * g_jobs[i].h.llr_len[cw]);
  return (int)g_jobs[i].h.llr_len[cw];
}
int o_trace_cb_get(uint32_t i, o_trace_cb_hdr_t* out, int16_t* d3, uint32_t cap)
{
  if (i >= g_ncbs) return -1;
  *out = g_cbs[i].h;
  const uint32_t n = 3 * (g_cbs[i].h.K + 4);
  if (d3) { if (n > cap) return -1; memcpy(d3, g_cbs[i].d3, sizeof(int16_t) * n); }
  return (int)n;
}
