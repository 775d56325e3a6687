/* oracle/or_segment.c -- TEST INFRASTRUCTURE: restatement of SMALT's candidate-region
 * binning: S4 (hits -> hit regions -> seeds -> constant-shift segments, segment.c:396-584,
 * 763-810), S5 (segments -> candidates with shift band, :929-1059, 1140-1223), S6 (coverage
 * ranking and depth cut, :1616-1785) and S7 (candidate -> windows + band, :1861-1985). */
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include "or_internal.h"

enum { HALFBIT = 31, SEGMENTING_DIFFSHIFT = 3, MAXIMUM_DEPTH = 8000, DEFAULT_TARGET_DEPTH = 200,
       EDGE_BAND_FACTOR = 4, MAX_BANDEDGE_2POW = 4 };
enum { CANDFLG_REVERSE = 1, CANDFLG_MMALI = 4 };
#define HALFMASK ((uint64_t) 0x7FFFFFFF)
#define SOFFSMASK ((uint64_t) 0xFFFFFFFF)

#define GROW(ptr, n, cap, T) \
  if ((n) >= (cap)) { (cap) = (cap)? 2*(cap): 1024; (ptr) = (T *) realloc((ptr), (size_t) (cap)*sizeof(T)); }

/* S4 -- segLstFillHits (segment.c:763-810) */
void or_seglst_fill(OrSegLst *sl, uint32_t min_ktup, const OrHitList *hl, const OrIndex *ix)
{
  const int k = ix->k, s = ix->s;
  const uint64_t *dat = hl->sqdat;
  const int nhits = hl->nhits;
  const uint8_t *qm;
  uint32_t ds, max_dshift, r;
  uint64_t dsthresh;
  int i, j;

  sl->nhreg = sl->nseed = sl->nsegm = 0;
  sl->is_reverse = hl->is_reverse;
  sl->qlen = hl->qlen;

  /* :781-788 -- the mask consulted is the HIT LIST's (all NOHIT in sequence-by-sequence mode) */
  for (qm = hl->qmask; *qm; qm++) {
    if (*qm == OR_HQ_NORMHIT) continue;
    if (min_ktup < 2) break;
    min_ktup--;
  }

  /* defineHitRegions, :396-453 */
  if (nhits >= 1) {
    max_dshift = (uint32_t) (k*SEGMENTING_DIFFSHIFT/s);
    ds = (hl->qlen - k)/s + 1;
    if (ds < max_dshift) max_dshift = (uint16_t) ds;
    max_dshift &= 0xffff;
    dsthresh = ((uint64_t) max_dshift) << HALFBIT;
    for (i = 0; i < nhits;) {
      for (j = i + 1; j < nhits; j++)
        if (dat[j] - dat[j-1] >= dsthresh) break;
      if ((uint32_t) (j - i) >= min_ktup) {
        GROW(sl->hreg, sl->nhreg, sl->cap_hreg, OrHitRegion)
        sl->hreg[sl->nhreg].idx = (uint32_t) i;
        sl->hreg[sl->nhreg].num = j - i;
        sl->nhreg++;
      }
      i = j;
    }
  }

  /* makeSeedsFromHits, :455-533 */
  for (r = 0; r < sl->nhreg; r++) {
    uint32_t a = sl->hreg[r].idx, b, end = a + (uint32_t) sl->hreg[r].num;
    sl->hreg[r].idx = sl->nseed;
    while (a < end) {
      uint64_t shift = dat[a] & ~HALFMASK;
      uint32_t qoffs = (uint32_t) (dat[a] & HALFMASK), lastq = qoffs + k, qo;
      for (b = a + 1; b < end; b++) {
        if ((dat[b] & ~HALFMASK) != shift) break;
        qo = (uint32_t) (dat[b] & HALFMASK);
        if (qo > lastq || ((qo - qoffs) % s)) break;
        lastq = qo + k;
      }
      GROW(sl->seed, sl->nseed, sl->cap_seed, OrSegSeed)
      sl->seed[sl->nseed].sqo = dat[a];
      sl->seed[sl->nseed].len = (int32_t) (lastq - qoffs);
      sl->nseed++;
      a = b;
    }
    sl->hreg[r].num = (int32_t) (sl->nseed - sl->hreg[r].idx);
  }

  /* makeSegmentsFromSeeds, :535-584 */
  for (r = 0; r < sl->nhreg; r++) {
    uint32_t a = sl->hreg[r].idx, b, end = a + (uint32_t) sl->hreg[r].num;
    sl->hreg[r].idx = sl->nsegm;
    sl->hreg[r].num = 0;
    while (a < end) {
      uint64_t shift = sl->seed[a].sqo & ~HALFMASK;
      uint32_t qoffs = (uint32_t) (sl->seed[a].sqo & HALFMASK);
      uint32_t cover = (uint32_t) sl->seed[a].len;
      for (b = a + 1; b < end; b++) {
        if ((sl->seed[b].sqo & ~HALFMASK) != shift ||
            (((uint32_t) (sl->seed[b].sqo & HALFMASK)) - qoffs) % s) break;
        cover += (uint32_t) sl->seed[b].len;
      }
      GROW(sl->segm, sl->nsegm, sl->cap_segm, OrSegment)
      sl->segm[sl->nsegm].ix = a;
      sl->segm[sl->nsegm].nseed = (int32_t) (b - a);
      sl->segm[sl->nsegm].cover = cover;
      sl->nsegm++;
      sl->hreg[r].num++;
      a = b;
    }
  }
}

/* calcSegmentBoundaries, segment.c:635-668 */
static void segment_bounds(uint32_t *qs, uint32_t *qe, uint32_t *rs, uint32_t *re, const OrSegment *sg,
                           const OrSegSeed *seedr, int k, int s, int is_reverse)
{
  const OrSegSeed *a = seedr + sg->ix, *b = a + sg->nseed - 1;
  *qs = (uint32_t) (a->sqo & HALFMASK);
  *qe = (uint32_t) (b->sqo & HALFMASK) + (uint32_t) b->len - 1;
  if (is_reverse) {
    *rs = (uint32_t) (((b->sqo >> HALFBIT) - (b->sqo & HALFMASK)/s) & SOFFSMASK);
    *rs -= (uint32_t) ((b->len - k)/s);
    *re = (uint32_t) (((a->sqo >> HALFBIT) - (*qs)/s) & SOFFSMASK);
  } else {
    *rs = (uint32_t) (((a->sqo >> HALFBIT) + (*qs)/s) & SOFFSMASK);
    *re = (uint32_t) (((b->sqo >> HALFBIT) + (b->sqo & HALFMASK)/s) & SOFFSMASK);
    *re += (uint32_t) ((b->len - k)/s);
  }
}

/* derriveSEGCAND, segment.c:929-1059 */
static int derive_cand(OrSegCand *c, int first, int nseg, OrSegment *segbase, const OrSegSeed *seedr,
                       int k, int s, uint32_t cover, uint32_t mincover_noindel, uint32_t hregix, int is_reverse)
{
  const uint64_t offbit = ((uint64_t) 1) << (HALFBIT + 1);
  OrSegment *sg0 = segbase + first, *sg = sg0 + 1;
  int64_t shift_min, shift_2mm, shift_start, diff_shift;
  uint64_t shift_range;
  uint32_t qs, qe, rs, re, maxcover;
  uint8_t flag = 0;
  int n;

  if (sg0->nseed < 0) return OR_ERR;
  segment_bounds(&c->qs, &c->qe, &c->rs, &c->re, sg0, seedr, k, s, is_reverse);
  sg0->nseed *= -1;
  shift_2mm = shift_min = (int64_t) (seedr[sg0->ix].sqo >> HALFBIT);
  maxcover = sg0->cover;
  for (n = 1; n < nseg; n++, sg++) {
    if (sg->nseed < 0) return OR_ERR;
    segment_bounds(&qs, &qe, &rs, &re, sg, seedr, k, s, is_reverse);
    if (sg->cover > maxcover) { shift_2mm = (int64_t) (seedr[sg->ix].sqo >> HALFBIT); maxcover = sg->cover; }
    sg->nseed *= -1;
    if (qs < c->qs) c->qs = qs;
    if (qe > c->qe) c->qe = qe;
    if (rs < c->rs) c->rs = rs;
    if (re > c->re) c->re = re;
  }
  sg--;
  if (is_reverse) {
    flag |= CANDFLG_REVERSE;
    shift_start = ((int64_t) c->rs) + (c->qe - k + 1)/s;
  } else {
    shift_start = (int64_t) ((((uint64_t) c->rs) | offbit) - c->qs/s);
  }
  shift_range = (uint64_t) (((int64_t) (seedr[sg->ix].sqo >> HALFBIT)) - shift_min);
  diff_shift = shift_min - shift_start;
  if (shift_range > SHRT_MAX) return OR_ERR;
  if (diff_shift < SHRT_MIN || diff_shift > SHRT_MAX) return OR_ERR;
  c->shiftoffs = (short) diff_shift;
  if (maxcover >= mincover_noindel) {
    int64_t ds = shift_2mm - shift_start;
    flag |= CANDFLG_MMALI;
    if (ds < SHRT_MIN || ds > SHRT_MAX) return OR_ERR;
    c->shift2mm = (short) ds;
  } else {
    c->shift2mm = 0;
  }
  c->flag = flag;
  c->srange = (short) shift_range;
  c->cover = cover;
  c->nseg = nseg;
  c->hregix = hregix;
  c->seqidx = -1;
  return OR_OK;
}

void or_segcands_blank(OrSegCands *sc)
{
  sc->ncand = 0; sc->n_sort = 0; sc->n_mincover = 0; sc->max_cover = 0; sc->max2nd_cover = 0;
  sc->cover_deficit[0] = sc->cover_deficit[1] = 0;
}

/* S5 -- segAliCandsAddFast -> addCandsFast (segment.c:1530-1557, 1140-1223).  `mask` is a
 * byte-per-read-base scratch (>= qlen). */
int or_segcands_add_fast(OrSegCands *sc, uint8_t *mask, OrSegLst *sl, const OrIndex *ix, uint32_t mincover, int32_t seqidx)
{
  const int k = ix->k, s = ix->s;
  uint32_t r;
  for (r = 0; r < sl->nhreg; r++) {
    const OrHitRegion *hr = sl->hreg + r;
    OrSegment *base = sl->segm + hr->idx;
    int i, j;
    for (i = 0; i < hr->num;) {
      OrSegment *sg = base + i;
      const OrSegSeed *sd;
      uint32_t cover, cover_new;
      int l, q;
      /* INIT_COVERAGE_CALC, :293-304 */
      memset(mask, 0, sl->qlen);
      for (l = sg->nseed, sd = sl->seed + sg->ix; l > 0; l--, sd++) {
        uint8_t *u = mask + (sd->sqo & HALFMASK);
        for (q = 0; q < sd->len; q++) u[q] = 1;
      }
      cover = sg->cover;
      sg++;
      for (j = i + 1; j < hr->num; j++, sg++) {
        if (sg->nseed < 0) break;
        /* CALC_COVERAGE, :327-338 */
        cover_new = 0;
        for (l = sg->nseed, sd = sl->seed + sg->ix; l > 0; l--, sd++) {
          uint8_t *u = mask + (sd->sqo & HALFMASK);
          for (q = 0; q < sd->len; q++) if (!u[q]) { cover_new++; u[q] = 1; }
        }
        if ((cover_new << 1) < sg->cover && cover >= mincover) break;
        cover += cover_new;
      }
      if (cover >= mincover) {
        OrSegCand *c;
        GROW(sc->cand, sc->ncand, sc->cap_cand, OrSegCand)
        c = sc->cand + sc->ncand;
        memset(c, 0, sizeof(*c));
        if (derive_cand(c, i, j - i, base, sl->seed, k, s, cover, mincover, r, sl->is_reverse)) return OR_ERR;
        sc->ncand++;
        c->seqidx = seqidx;
        if (cover > sc->max2nd_cover) {
          if (cover > sc->max_cover) { sc->max2nd_cover = sc->max_cover; sc->max_cover = cover; }
          else if (cover != sc->max_cover) sc->max2nd_cover = cover;
        }
      }
      i = j;
    }
  }
  return OR_OK;
}

/* S6 -- segAliCandsStats (segment.c:1616-1785) */
int or_segcands_stats(OrSegCands *sc, const OrIndex *ix, uint32_t min_cover_below_max, const OrHitInfo *hf,
                      const OrHitInfo *hr, uint32_t target_depth, uint32_t max_depth, int is_sensitive)
{
  const uint32_t s = (uint32_t) ix->s, n_cands = sc->ncand;
  uint32_t i, j, min_cover, cdf = 0, adj[2];
  const OrSegCand *scp = sc->cand;

  if (max_depth < 1 || max_depth > MAXIMUM_DEPTH) max_depth = MAXIMUM_DEPTH;
  if (target_depth < 1) target_depth = DEFAULT_TARGET_DEPTH;
  if (target_depth > max_depth) target_depth = max_depth;

  min_cover = (min_cover_below_max > sc->max_cover)? 0: sc->max_cover - min_cover_below_max;
  if (min_cover > sc->max2nd_cover) { cdf = min_cover - sc->max2nd_cover; min_cover = sc->max2nd_cover; }
  sc->cover_deficit[0] = or_hitinfo_cover_deficit(hf, ix);
  sc->cover_deficit[1] = or_hitinfo_cover_deficit(hr, ix);
  for (i = 0; i < 2; i++) {
    adj[i] = sc->cover_deficit[0];          /* [0] for both strands, :1676 */
    adj[i] = (adj[i] > cdf)? adj[i] - cdf: 0;
  }
  if (n_cands + 1 > sc->cap_sort) {
    sc->cap_sort = n_cands + 1024;
    sc->sort_keys = realloc(sc->sort_keys, sc->cap_sort*sizeof(uint32_t));
    sc->sort_idx = realloc(sc->sort_idx, sc->cap_sort*sizeof(uint32_t));
  }
  for (i = j = 0; i < n_cands; i++) {
    int rev = (scp[i].flag & CANDFLG_REVERSE)? 1: 0;
    if (scp[i].cover + adj[rev] < min_cover) continue;
    if (scp[i].cover > sc->max_cover) return OR_ERR;
    sc->sort_keys[j] = sc->max_cover - scp[i].cover;
    sc->sort_idx[j] = i;
    j++;
  }
  or_sort2_u32((int) j, sc->sort_keys, sc->sort_idx);
  sc->n_mincover = j;
  if (j > target_depth) {
    uint32_t maxj = (j < max_depth)? j: max_depth;
    if (is_sensitive) {
      for (j = target_depth; j < maxj; j++)   /* scp[j], not scp[sort_idx[j]]: :1761-1762 */
        if (sc->sort_keys[j] >= adj[(scp[j].flag & CANDFLG_REVERSE)? 1: 0]) break;
      for (; j < sc->n_mincover && sc->sort_keys[j] < s; j++);
    } else {
      uint32_t cov = sc->sort_keys[j/2];
      if (cov < s) cov = s;
      for (j = target_depth; j < maxj && sc->sort_keys[j] < cov; j++);
    }
  }
  sc->n_sort = j;
  return OR_OK;
}

/* S7 -- segAliCandsCalcSegmentOffsets (segment.c:1861-1985) */
int or_segcands_offsets(OrCand *c, const OrSegCands *sc, const OrIndex *ix, uint32_t scidx, int edgelen, uint32_t qlen)
{
  const int s = ix->s, k = ix->k;
  const OrSegCand *p;
  uint64_t roffs, rlen, rs, re;
  uint32_t qs, qe;
  int bl, br, band_offs, ds, q_edge_l, q_edge_r, r_edge_l, r_edge_r, edge_band;

  if (scidx >= sc->n_sort) return OR_ERR;
  p = sc->cand + sc->sort_idx[scidx];
  c->sqidx = p->seqidx;
  c->flags = (p->flag & CANDFLG_REVERSE)? 1: 0;
  c->cover = p->cover;
  if (p->seqidx < 0 || p->seqidx >= ix->nseq) { roffs = 0; rlen = ix->sop[ix->nseq]; }
  else { roffs = ix->sop[p->seqidx]; rlen = ix->sop[p->seqidx + 1] - roffs; }
  rs = ((uint64_t) p->rs)*s;
  re = ((uint64_t) p->re)*s + k - 1;
  if (rs < roffs || re < rs) return OR_ERR;
  rs -= roffs; re -= roffs;
  if (re >= rlen) return OR_ERR;
  if (p->qe < p->qs || p->qs >= qlen) return OR_ERR;
  if (p->flag & CANDFLG_REVERSE) { qs = qlen - p->qe - 1; qe = qlen - p->qs - 1; }
  else { qs = p->qs; qe = p->qe; }

  edge_band = (int) (qlen - p->cover)/EDGE_BAND_FACTOR;
  if (edge_band > s) {
    if (edge_band > (int) (qlen >> MAX_BANDEDGE_2POW)) edge_band = (int) (qlen >> MAX_BANDEDGE_2POW);
    edge_band -= s - 1;
  } else edge_band = 0;
  br = (-p->shiftoffs + 1)*s + edge_band + 1;
  bl = br - (p->srange + 2)*s - 2*edge_band - 2;

  q_edge_l = (qs >= (uint32_t) edgelen && edgelen > 0)? edgelen: (int) qs;
  q_edge_r = (qe + edgelen + 1 <= qlen && edgelen > 0)? edgelen: (int) (qlen - qe - 1);
  qs -= q_edge_l;
  qe += q_edge_r;
  r_edge_l = q_edge_l + br;
  r_edge_r = q_edge_r - bl;
  if (r_edge_l > 0 && rs < (uint64_t) r_edge_l) { r_edge_l = (int) rs; rs = 0; }
  else rs -= r_edge_l;
  if (re + r_edge_r >= rlen) { r_edge_r = (int) (rlen - re - 1); re = rlen - 1; }
  else re += r_edge_r;
  if (re < rs) return OR_ERR;
  band_offs = q_edge_l - r_edge_l;
  ds = p->shift2mm*s + band_offs;
  c->band_l = bl + band_offs + (int) qs;
  c->band_r = br + band_offs + (int) qs;
  if (ds < 0) { c->dqo = qs - ds; c->dro = 0; } else { c->dqo = qs; c->dro = ds; }
  c->qs = qs; c->qe = qe; c->rs = rs; c->re = re;
  (void) r_edge_r;
  return OR_OK;
}
