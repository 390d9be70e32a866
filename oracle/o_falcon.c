/* o_falcon.c - ORACLE (test infrastructure only): PDCCH search-space validation and the RNTI manager.
 * Follows /root/reference/lib/src/phy/falcon_phch/falcon_pdcch.c:183-250 (generic locations, validate),
 * /root/reference/lib/src/util/RNTIManager.cc:131-444, Histogram.cc:27-63, Interval.cc:37-39 and
 * srsran_pdcch_ue_locations_ncce / srsran_pdcch_common_locations_ncce [srsRAN; same arithmetic as the in-tree
 * srsran_pdcch_ue_locations_check, falcon_pdcch.c:49-99]. */
#include "lsn_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t L, ncce; } loc_t;

static uint32_t ue_locations(uint32_t nof_cce, loc_t* c, uint32_t max, uint32_t nsubframe, uint16_t rnti)
{
  static const uint32_t ncand[4] = {6, 6, 2, 2};
  uint32_t Yk = rnti, k = 0;
  for (uint32_t m = 0; m < nsubframe + 1; m++) Yk = (39827u * Yk) % 65537u;
  for (int l = 3; l >= 0; l--) {
    uint32_t L = 1u << l;
    for (uint32_t i = 0; i < ncand[l]; i++)
      if (nof_cce >= L) {
        uint32_t ncce = L * ((Yk + i) % (nof_cce / L));
        if (k < max && ncce + L <= nof_cce) {
          c[k].L = (uint32_t)l;
          c[k].ncce = ncce;
          k++;
        }
      }
  }
  return k;
}

static uint32_t common_locations(uint32_t nof_cce, loc_t* c, uint32_t max)
{
  uint32_t k = 0;
  for (int l = 3; l > 1; l--) {
    uint32_t L = 1u << l;
    uint32_t lim = (nof_cce < 16 ? nof_cce : 16) / L;
    for (uint32_t i = 0; i < lim; i++) {
      uint32_t ncce = L * (i % (nof_cce / L));
      if (k < max && ncce + L <= nof_cce) {
        c[k].L = (uint32_t)l;
        c[k].ncce = ncce;
        k++;
      }
    }
  }
  return k;
}

/* falcon_pdcch.c:183-207 */
static uint32_t generic_locations(uint32_t nof_cce, loc_t* c, uint32_t max, uint32_t nsubframe, uint16_t rnti)
{
  if (rnti >= O_RARNTI_START && rnti <= O_RARNTI_END) return common_locations(nof_cce, c, max);
  if (rnti >= O_CRNTI_START && rnti <= O_CRNTI_END) {
    uint32_t n = ue_locations(nof_cce, c, max, nsubframe, rnti);
    n += common_locations(nof_cce, &c[n], max - n);
    return n;
  }
  if (rnti >= O_MRNTI) return common_locations(nof_cce, c, max); /* M/P/SI-RNTI */
  return 0; /* reserved interval, or rnti 0 */
}

/* falcon_pdcch.c:223-250: 0 invalid, 1 valid but ambiguous with L-1, 2 valid */
uint32_t o_validate_location(uint32_t nof_cce, uint32_t ncce, uint32_t l, uint32_t nsubframe, uint16_t rnti)
{
  loc_t loc[22];
  uint32_t amb = 0, valid = 0;
  uint32_t n = generic_locations(nof_cce, loc, 22, nsubframe, rnti);
  for (uint32_t i = 0; i < n; i++)
    if (loc[i].ncce == ncce) {
      if (l > 0 && (l - 1 == loc[i].L)) amb = 1;
      if (loc[i].L == l) valid = 1;
    }
  if (valid && !amb) valid = 2;
  return valid;
}

/* ---------------- Histogram (Histogram.cc) ---------------- */
typedef struct {
  uint32_t* hist;    /* [65536] */
  uint16_t* history; /* [depth] */
  uint32_t cur, end;
  int ready;
} hist_t;

static void hist_init(hist_t* h, uint32_t depth)
{
  h->hist = (uint32_t*)calloc(65536, sizeof(uint32_t));
  h->history = (uint16_t*)calloc(depth, sizeof(uint16_t));
  h->cur = 0;
  h->end = depth;
  h->ready = 0;
}
static void hist_add(hist_t* h, uint16_t item, uint32_t ntimes)
{
  while (ntimes-- > 0) {
    if (h->ready) h->hist[h->history[h->cur]]--;
    h->history[h->cur] = item;
    h->hist[item]++;
    h->cur++;
    if (h->cur == h->end) {
      h->ready = 1;
      h->cur = 0;
    }
  }
}

/* ---------------- RNTIManager (RNTIManager.cc) ---------------- */
#define RM_MAX_INTERVALS 8
struct o_rntiman {
  uint32_t nformats;
  hist_t* h;
  uint16_t ev[O_NOF_FORMATS][RM_MAX_INTERVALS][2];
  uint32_t nev[O_NOF_FORMATS];
  uint16_t fb[O_NOF_FORMATS][RM_MAX_INTERVALS][2];
  uint32_t nfb[O_NOF_FORMATS];
  uint8_t* active;   /* [65536] */
  uint8_t* reason;   /* [65536], valid while active */
  uint32_t* last_seen;
  uint32_t* assoc;
  uint32_t nactive;
  uint32_t timestamp, lifetime, threshold, maxcand;
  int32_t remaining[O_NOF_FORMATS];
};

o_rntiman_t* o_rntiman_new(uint32_t nformats, uint32_t maxcand, uint32_t threshold)
{
  o_rntiman_t* r = (o_rntiman_t*)calloc(1, sizeof(*r));
  r->nformats = nformats;
  r->h = (hist_t*)calloc(nformats, sizeof(hist_t));
  for (uint32_t i = 0; i < nformats; i++) hist_init(&r->h[i], 200u * (304u / 5u)); /* RNTIManager.h:47-49 */
  r->active = (uint8_t*)calloc(65536, 1);
  r->reason = (uint8_t*)calloc(65536, 1);
  r->last_seen = (uint32_t*)calloc(65536, sizeof(uint32_t));
  r->assoc = (uint32_t*)calloc(65536, sizeof(uint32_t));
  r->lifetime = 10000; /* RRC_INACTIVITY_TIMER_MS, RNTIManager.h:42 */
  r->threshold = threshold;
  r->maxcand = maxcand;
  for (uint32_t i = 0; i < nformats; i++) r->remaining[i] = (int32_t)maxcand;
  return r;
}
void o_rntiman_free(o_rntiman_t* r)
{
  if (!r) return;
  for (uint32_t i = 0; i < r->nformats; i++) {
    free(r->h[i].hist);
    free(r->h[i].history);
  }
  free(r->h);
  free(r->active);
  free(r->reason);
  free(r->last_seen);
  free(r->assoc);
  free(r);
}
void o_rntiman_add_evergreen(o_rntiman_t* r, uint16_t a, uint16_t b, uint32_t f)
{
  if (r->nev[f] < RM_MAX_INTERVALS) {
    r->ev[f][r->nev[f]][0] = a;
    r->ev[f][r->nev[f]][1] = b;
    r->nev[f]++;
  }
}
void o_rntiman_add_forbidden(o_rntiman_t* r, uint16_t a, uint16_t b, uint32_t f)
{
  if (r->nfb[f] < RM_MAX_INTERVALS) {
    r->fb[f][r->nfb[f]][0] = a;
    r->fb[f][r->nfb[f]][1] = b;
    r->nfb[f]++;
  }
}
void o_rntiman_add_candidate(o_rntiman_t* r, uint16_t rnti, uint32_t f)
{
  hist_add(&r->h[f], rnti, 1);
  r->remaining[f]--;
}
static int is_evergreen(o_rntiman_t* r, uint16_t rnti, uint32_t f)
{
  for (uint32_t i = 0; i < r->nev[f]; i++)
    if (rnti >= r->ev[f][i][0] && rnti <= r->ev[f][i][1]) return 1;
  return 0;
}
int o_rntiman_is_forbidden(o_rntiman_t* r, uint16_t rnti, uint32_t f)
{
  for (uint32_t i = 0; i < r->nfb[f]; i++)
    if (rnti >= r->fb[f][i][0] && rnti <= r->fb[f][i][1]) return 1;
  return 0;
}
static void activate(o_rntiman_t* r, uint16_t rnti, int reason)
{
  if (!r->active[rnti]) {
    r->active[rnti] = 1;
    r->reason[rnti] = (uint8_t)reason;
    r->nactive++;
  }
}
static void deactivate(o_rntiman_t* r, uint16_t rnti)
{
  if (r->active[rnti]) {
    r->active[rnti] = 0;
    r->assoc[rnti] = 0;
    r->reason[rnti] = O_ACT_UNSET;
    r->nactive--;
  }
}
static uint32_t likely_dl_format(o_rntiman_t* r, uint16_t rnti)
{
  uint32_t res = 0, mx = 0;
  for (uint32_t f = 1; f < r->nformats; f++) {
    uint32_t c = r->h[f].hist[rnti];
    if (c > mx) {
      mx = c;
      res = f;
    }
  }
  return res;
}
static int validate(o_rntiman_t* r, uint16_t rnti, uint32_t f)
{
  if (is_evergreen(r, rnti, f)) return 1;
  if (o_rntiman_is_forbidden(r, rnti, f)) return 0;
  if (r->active[rnti]) { /* validateByActiveList, RNTIManager.cc:315-341 */
    if (r->timestamp - r->last_seen[rnti] < r->lifetime) return 1;
    deactivate(r, rnti);
  }
  /* validateByHistogram, RNTIManager.cc:343-369 */
  uint32_t likely = likely_dl_format(r, rnti);
  if (f != 0 && f != likely) return 0;
  uint32_t ul = r->h[0].hist[rnti];
  uint32_t dl = likely != 0 ? r->h[likely].hist[rnti] : 0;
  if (ul + dl > r->threshold) {
    activate(r, rnti, O_ACT_HISTOGRAM);
    r->assoc[rnti] = dl > r->threshold ? likely : 0;
    return 1;
  }
  return 0;
}
int o_rntiman_validate_and_refresh(o_rntiman_t* r, uint16_t rnti, uint32_t f)
{
  int ok = validate(r, rnti, f);
  if (ok) r->last_seen[rnti] = r->timestamp;
  return ok;
}
void o_rntiman_activate_and_refresh(o_rntiman_t* r, uint16_t rnti, uint32_t f, int reason)
{
  activate(r, rnti, reason);
  r->last_seen[rnti] = r->timestamp;
  r->assoc[rnti] = f;
}
uint32_t o_rntiman_get_frequency(o_rntiman_t* r, uint16_t rnti, uint32_t f) { return r->h[f].hist[rnti]; }
int o_rntiman_get_activation_reason(o_rntiman_t* r, uint16_t rnti) { return r->active[rnti] ? r->reason[rnti] : O_ACT_UNSET; }
void o_rntiman_step_time(o_rntiman_t* r)
{
  for (uint32_t i = 0; i < r->nformats; i++) {
    if (r->remaining[i] > 0) hist_add(&r->h[i], 0, (uint32_t)r->remaining[i]);
    r->remaining[i] = (int32_t)r->maxcand;
  }
  r->timestamp++;
}
uint32_t o_rntiman_nof_active(o_rntiman_t* r) { return r->nactive; }
