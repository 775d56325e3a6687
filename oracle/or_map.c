/* oracle/or_map.c -- TEST INFRASTRUCTURE: restatement of SMALT's per-read control flow
 * (rmapSingle rmap.c:1648 -> mapSingleRead :1228 -> scoreRMAPCAND :588 ->
 * alignRMAPCANDFull :790 -> resultSetAddFromAli results.c:1852) on top of the stage
 * functions in or_seed.c / or_segment.c / or_align.c, and the stage dump in the line format
 * of oracle/refdump (DUMPFORMAT.md). */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include "or_internal.h"

enum { HASH_MAXNHITS = 16*1024, MINLEN_QUERY_STRIPED = 32, BWSCAL_QLEN = 48 };   /* rmap.c:50,83-86 */

OrMap *or_map_create(const OrIndex *ix)
{
  OrMap *m = calloc(1, sizeof(OrMap));
  m->ix = ix;
  or_hitinfo_init(&m->hi[0], ix->s);
  or_hitinfo_init(&m->hi[1], ix->s);
  m->keep_hitlists = 1;
  m->niv = m->niv_next = -1;
  return m;
}

static void keep_clear(OrMap *m)
{
  int i;
  for (i = 0; i < m->hl_keep_cnt; i++) free(m->hl_keep[i]);
  m->hl_keep_cnt = 0;
}

void or_map_free(OrMap *m)
{
  if (!m) return;
  keep_clear(m);
  free(m->hl_keep); free(m->hl_keep_n);
  or_hitinfo_free(&m->hi[0]); or_hitinfo_free(&m->hi[1]);
  free(m->hl.sqdat); free(m->hl.qmask);
  free(m->sl.hreg); free(m->sl.seed); free(m->sl.segm);
  free(m->sc.cand); free(m->sc.sort_keys); free(m->sc.sort_idx);
  free(m->qmaskbuf); free(m->read[0]); free(m->read[1]); free(m->qual); free(m->win);
  free(m->cand); free(m->res); free(m->diffstr);
  free(m);
}

static void keep_hitlist(OrMap *m)
{
  int n = m->hl_keep_cnt;
  if (!m->keep_hitlists) return;
  m->hl_keep = realloc(m->hl_keep, (n + 1)*sizeof(uint64_t *));
  m->hl_keep_n = realloc(m->hl_keep_n, (n + 1)*sizeof(int));
  m->hl_keep[n] = malloc(((size_t) m->hl.nhits + 1)*sizeof(uint64_t));
  memcpy(m->hl_keep[n], m->hl.sqdat, (size_t) m->hl.nhits*sizeof(uint64_t));
  m->hl_keep_n[n] = m->hl.nhits;
  m->hl_keep_cnt++;
}

/* collectHits, rmap.c:273-351 */
static int collect_hits(OrMap *m, OrHitInfo *hi, uint32_t n_hit_max, uint32_t n_ktup_min, uint32_t cover_min, int with_seqidx)
{
  const OrIndex *ix = m->ix;
  int rv = OR_OK;
  if (with_seqidx) {
    int64_t s;
    for (s = 0; s < ix->nseq; s++) {
      or_collect_hits_segment(&m->hl, hi, ix, ix->sop[s], ix->sop[s+1], n_hit_max, 1);
      keep_hitlist(m);
      or_seglst_fill(&m->sl, n_ktup_min, &m->hl, ix);
      if ((rv = or_segcands_add_fast(&m->sc, m->qmaskbuf, &m->sl, ix, cover_min, (int32_t) s))) break;
    }
  } else {
    or_collect_hits_cutoff(&m->hl, hi, ix, n_hit_max);
    keep_hitlist(m);
    or_seglst_fill(&m->sl, n_ktup_min, &m->hl, ix);
    rv = or_segcands_add_fast(&m->sc, m->qmaskbuf, &m->sl, ix, cover_min, -1);
  }
  return rv;
}

/* collectHitsFromInterVal, rmap.c:438-492: one hit list per interval, all seeds (use_short_hitinfo = 0) */
static int collect_hits_intervals(OrMap *m, OrHitInfo *hi, uint32_t n_hit_max, uint32_t n_ktup_min, uint32_t cover_min)
{
  const OrIndex *ix = m->ix;
  int i, rv = OR_OK;
  for (i = 0; i < m->niv; i++) {
    const uint64_t offs = ix->sop[m->iv_sx[i]];
    or_collect_hits_segment(&m->hl, hi, ix, offs + m->iv_lo[i], offs + m->iv_hi[i] + 1, n_hit_max, 0);
    keep_hitlist(m);
    or_seglst_fill(&m->sl, n_ktup_min, &m->hl, ix);
    if ((rv = or_segcands_add_fast(&m->sc, m->qmaskbuf, &m->sl, ix, cover_min, (int32_t) m->iv_sx[i]))) break;
  }
  return rv;
}

static const uint8_t *fetch_window(OrMap *m, const OrCand *c)
{
  const uint32_t len = (uint32_t) (c->re - c->rs + 1);
  const uint64_t base = (c->sqidx < 0)? 0: m->ix->sop[c->sqidx];
  if (len + 1 > m->win_cap) { m->win_cap = len + 1024; m->win = realloc(m->win, m->win_cap); }
  or_index_fetch(m->ix, base + c->rs, len, m->win);
  return m->win;
}

/* scoreRMAPCAND, rmap.c:588-788 */
static int score_cands(OrMap *m, const int8_t M[8][8], int *max1, int *max2)
{
  const OrParams *p = &m->par;
  const int mmscordiff = p->match - p->mismatch;
  const uint32_t n_candseg = m->sc.n_sort, qlen = m->qlen;
  uint32_t i, cover, max_cover = 0, min_cover = 0, dcov, cdf;
  int rv;
  if (mmscordiff < 1) return OR_ERR;
  if (n_candseg + 1 > m->cap_cand) { m->cap_cand = n_candseg + 512; m->cand = realloc(m->cand, m->cap_cand*sizeof(OrCand)); }
  *max1 = *max2 = 0;
  for (i = 0; i < n_candseg; i++) {
    OrCand *c = m->cand + i;
    const uint8_t *win, *q;
    uint32_t wlen;
    int simd;
    memset(c, 0, sizeof(*c));
    /* makeRMAPCANDfromSegment, rmap.c:535-586 (edgelen 0 in SIMD builds) */
    if ((rv = or_segcands_offsets(c, &m->sc, m->ix, i, 0, qlen))) return rv;
    cover = c->cover;
    if (c->qe > INT_MAX || c->re < c->rs || c->re - c->rs > INT_MAX) return OR_ERR;
    win = fetch_window(m, c);
    wlen = (uint32_t) (c->re - c->rs + 1);
    q = m->read[c->flags & 1];
    simd = (qlen >= MINLEN_QUERY_STRIPED && ((uint32_t) (c->band_r - c->band_l)*BWSCAL_QLEN) > qlen &&
            c->qs == 0 && c->qe >= qlen - 1);     /* rmap.c:715-718 */
    c->used_simd = simd;
    if (simd) {
      c->swscor = or_sw_full(q, qlen, win, wlen, M, p->gap_init, p->gap_ext);
      if (c->swscor >= 65535) simd = 0;           /* ERRCODE_SWATEXCEED -> banded pass, :730 */
    }
    if (!simd) {
      if ((rv = or_sw_band_fast(&c->swscor, q, qlen, win, (int) wlen, M, p->gap_init, p->gap_ext,
                                c->band_l, c->band_r, (int) c->qs, (int) c->qe, 0, (int) wlen - 1))) return rv;
    }
    c->flags |= 2;
    cdf = m->sc.cover_deficit[c->flags & 1];
    if ((p->flags & OR_FLG_BEST) && cover + cdf < min_cover) break;
    if (c->swscor > *max2) {
      if (c->swscor > *max1) {
        *max2 = *max1; *max1 = c->swscor;
        if (cover + cdf > max_cover) max_cover = (cover > cdf)? cover - cdf: 0;
      } else *max2 = c->swscor;
      dcov = (uint32_t) (((int) ((*max1 - *max2)/mmscordiff) + 1)*m->ix->s);
      if (dcov + cdf + min_cover < max_cover) min_cover = max_cover - dcov;
    }
  }
  m->ncand = i;
  return OR_OK;
}

static void add_diffstr(OrMap *m, const uint8_t *d, int len)
{
  if (m->ndiff + (uint32_t) len + 1 > m->cap_diff) { m->cap_diff = 2*(m->ndiff + len) + 256; m->diffstr = realloc(m->diffstr, m->cap_diff); }
  memcpy(m->diffstr + m->ndiff, d, (size_t) len);
  m->ndiff += (uint32_t) len;
}

/* resultSetAddFromAli, results.c:1852-1942, including its handling of duplicates: a result
 * equal to its predecessor is popped, and anything written after a popped slot within the
 * same call is lost from the array (but still raises the score maxima). */
static void add_results(OrMap *m, const OrAli *a, int nres, uint64_t soffs, uint32_t qlen, int64_t seqidx, int is_reverse)
{
  OrResult *rp;
  int i, is_new = 0;
  uint32_t arrlen;
  if (nres < 1) return;
  if (m->nres + (uint32_t) nres + 2 > m->cap_res) { m->cap_res = 2*(m->nres + nres) + 64; m->res = realloc(m->res, m->cap_res*sizeof(OrResult)); }
  arrlen = m->nres;
  rp = m->res + arrlen++;
  memset(rp, 0, sizeof(*rp));
  for (i = 0; i < nres; i++) {
    if (is_new) { rp = m->res + arrlen++; is_new = 0; }
    rp->swatscor = a[i].score;
    if (is_reverse) { rp->q_start = qlen - a[i].qe; rp->q_end = qlen - a[i].qs; }
    else { rp->q_start = a[i].qs + 1; rp->q_end = a[i].qe + 1; }
    rp->s_start = (uint32_t) soffs + a[i].rs + 1;      /* soffs is passed as SEQLEN_t */
    rp->s_end = (uint32_t) soffs + a[i].re + 1;
    rp->sidx = seqidx;
    {
      const OrResult *pp = rp - 1;
      is_new = (arrlen < 2 || !(rp->s_start == pp->s_start && rp->s_end == pp->s_end && rp->q_start == pp->q_start &&
                                rp->q_end == pp->q_end && rp->swatscor == pp->swatscor && rp->sidx == pp->sidx));
      if (m->par.flags & OR_FLG_RAWRESULTS) is_new = 1;   /* the caller holds the set this call appends to and compares there */
    }
    if (is_new) {
      rp->stroffs = (int) m->ndiff;
      rp->strlen = a[i].dlen;                            /* DIFFSTR_LENGTH counts the terminating M:0 */
      add_diffstr(m, a[i].diffstr, a[i].dlen);
      if (rp->swatscor > m->sw2nd) {                     /* UPDATE_SWATSCORMAX, results.c:1013 */
        if (rp->swatscor > m->swmax) { m->sw2nd = m->swmax; m->swmax = rp->swatscor; }
        else if (rp->swatscor < m->swmax) m->sw2nd = rp->swatscor;
      }
      rp->reverse = is_reverse;
      rp->cand_first = (i == 0);
    } else {
      arrlen--;
    }
  }
  m->nres = arrlen;
}

/* alignRMAPCANDFull, rmap.c:790-928 */
static int align_cands(OrMap *m, const int8_t M[8][8], int min_swatscor, int scorlen_min, int bandwidth_min)
{
  const OrParams *p = &m->par;
  uint32_t i;
  for (i = 0; i < m->ncand; i++) {
    OrCand *c = m->cand + i;
    const uint8_t *win;
    int wlen, bw, band_l, band_r, rv, nali;
    OrAli *ali;
    if ((c->flags & 2) && c->swscor < min_swatscor) continue;
    win = fetch_window(m, c);
    wlen = (int) (c->re - c->rs + 1);
    if (c->qs > c->qe || c->qe >= m->qlen) return OR_ERR;
    if ((p->flags & OR_FLG_BEST) && m->sw2nd > min_swatscor) min_swatscor = m->sw2nd;
    bw = c->band_r - c->band_l;
    if (bw < bandwidth_min) { bw = (bandwidth_min - bw + 1)/2; band_l = c->band_l - bw; band_r = c->band_r + bw; }
    else { band_l = c->band_l; band_r = c->band_r; }
    rv = or_sw_band_full(&ali, &nali, m->read[c->flags & 1], m->qlen, win, wlen, M, p->gap_init, p->gap_ext, p->match,
                         band_l, band_r, (int) c->qs, (int) c->qe, 0, wlen - 1, min_swatscor, scorlen_min);
    if (!rv) add_results(m, ali, nali, c->rs, m->qlen, c->sqidx, (int) (c->flags & 1));
    or_ali_free(ali, nali);
    if (rv) return rv;
  }
  return OR_OK;
}

/* mapSingleRead, rmap.c:1228-1433 (without the results.c post-processing, row N1) */
static int map_single_read(OrMap *m)
{
  const OrParams *p = &m->par;
  const OrIndex *ix = m->ix;
  const int k = ix->k, s = ix->s;
  const int mismatchdiff = p->match - p->mismatch;
  int8_t M[8][8];
  int scorlen_min = k + s, bandwidth_min, min_swatscor = p->min_swatscor, below_max = p->min_swatscor_below_max;
  int max1 = 0, max2 = 0, maxscor_perfect, rv;
  uint32_t min_cover = p->min_cover, min_ktup, mincov_below_max, nr, ntot;

  /* calcMinKtup, rmap.c:240-247 */
  min_ktup = (min_cover >= (uint32_t) (k + s))? (min_cover - k)/s: 1;
  min_cover = (min_ktup - 1)*s + k;
  if (mismatchdiff < 0 || p->gap_ext >= 0 || p->mismatch >= 0) return OR_ERR;
  maxscor_perfect = (int) m->qlen*p->match;
  if (below_max < 0) mincov_below_max = m->qlen - 1;
  else {
    mincov_below_max = ((uint32_t) (below_max/mismatchdiff))*s;
    if (mincov_below_max < (uint32_t) k || (p->flags & OR_FLG_BEST)) mincov_below_max = k + 2*(s - 1);
  }
  or_score_matrix(M, p->match, p->mismatch);

  /* fillRMAPBUFF, rmap.c:1153-1226 */
  or_segcands_blank(&m->sc);
  if (m->niv >= 0) {                                   /* filtered (rmap.c:1177-1198) */
    if ((rv = collect_hits_intervals(m, &m->hi[0], (uint32_t) p->ncut, min_ktup, min_cover))) return rv;
    if ((rv = collect_hits_intervals(m, &m->hi[1], (uint32_t) p->ncut, min_ktup, min_cover))) return rv;
  } else {
  if ((rv = collect_hits(m, &m->hi[0], (uint32_t) p->ncut, min_ktup, min_cover, (p->flags & OR_FLG_SEQBYSEQ) != 0))) return rv;
  if ((rv = collect_hits(m, &m->hi[1], (uint32_t) p->ncut, min_ktup, min_cover, (p->flags & OR_FLG_SEQBYSEQ) != 0))) return rv;
  }

  if ((rv = or_segcands_stats(&m->sc, ix, mincov_below_max, &m->hi[0], &m->hi[1], (uint32_t) p->target_depth,
                              (uint32_t) p->max_depth, (p->flags & OR_FLG_SENSITIVE) != 0))) return rv;
  m->nseg = (int) m->sc.n_sort;
  m->nseg_tot = (int) m->sc.n_mincover;
  m->nhit = (int) or_hitinfo_hit_numbers(&m->hi[0], &nr); ntot = nr;      /* calcTotalHitNumStats, rmap.c:1086 */
  m->nhit += (int) or_hitinfo_hit_numbers(&m->hi[1], &nr); ntot += nr;
  { int t = m->nhit; m->nhit = (int) ntot; m->nhit_tot = t; }

  rv = score_cands(m, M, &max1, &max2);
  m->max1 = max1; m->max2 = max2;
  if (rv) return rv;
  if (max1 > maxscor_perfect) return OR_ERR;
  if (max1 < 1) return OR_OK;

  /* rmap.c:1379-1400 */
  bandwidth_min = (maxscor_perfect - max1)/(-1*p->gap_ext);
  if (below_max >= max1) below_max = max1;
  if (min_swatscor > max2 && max2 > 0) min_swatscor = max2;
  if (below_max >= 0) {
    int minswc = (max2 > 0)? max2: max1;
    if (p->flags & OR_FLG_BEST) { if (minswc > min_swatscor) min_swatscor = minswc; }
    else if (min_swatscor + below_max < max1) {
      min_swatscor = max1 - below_max;
      if (min_swatscor > minswc) min_swatscor = minswc;
    }
  }
  if (min_swatscor > scorlen_min*p->match && p->match > 0) scorlen_min = min_swatscor/p->match;
  m->th_bandwidth_min = bandwidth_min; m->th_min_swatscor = min_swatscor; m->th_scorlen_min = scorlen_min;
  return align_cands(m, M, min_swatscor, scorlen_min, bandwidth_min);
}

int or_map_single_restricted(OrMap *m, const char *bases, const char *quals, uint32_t len, const OrParams *p,
                             int niv, const int64_t *sx, const uint32_t *lo, const uint32_t *hi)
{
  int rv;
  m->niv_next = niv; m->iv_sx = sx; m->iv_lo = lo; m->iv_hi = hi;
  rv = or_map_single(m, bases, quals, len, p);
  m->niv_next = -1;
  return rv;
}

void or_map_set_prevmax(OrMap *m, int swmax, int sw2nd) { m->prevmax[0] = swmax; m->prevmax[1] = sw2nd; }
void or_map_set_seed_range(OrMap *m, uint32_t first, uint32_t last) { m->seed_range[0] = first; m->seed_range[1] = last; }

/* calcTotalNumberOfHits, rmap.c:1076-1081 -> hashCalcHitInfoNumberOfHits, hashhit.c:1171-1197 */
uint32_t or_map_hit_total(const OrMap *m, int ktuple_maxhit)
{
  uint32_t st, i, tot = 0;
  const uint32_t cut = (ktuple_maxhit < 1)? 0: (uint32_t) ktuple_maxhit;
  if (m->qlen < (uint32_t) m->ix->k) return 0;
  for (st = 0; st < 2; st++)
    for (i = 0; i < m->hi[st].n_seeds; i++)
      if (!cut || m->hi[st].sortkey[i] <= cut) tot += m->hi[st].sortkey[i];
  return tot;
}

/* rmapSingle, rmap.c:1648-1742; the second call of a split read (mapSecondary, rmap.c:1435-1505) is this function again with
 * or_map_set_seed_range + or_map_set_prevmax and OR_FLG_RAWRESULTS (the caller holds the set it appends to) */
int or_map_single(OrMap *m, const char *bases, const char *quals, uint32_t len, const OrParams *p)
{
  const OrIndex *ix = m->ix;
  uint32_t i;
  int rv;
  m->par = *p;
  m->niv = m->niv_next;
  /* a call that appends to a ResultSet (rmapPair) continues that set's running score maxima: alignRMAPCANDFull raises its
   * threshold to the set's second-best score (rmap.c:881-885), whatever call it came from */
  m->nres = 0; m->ndiff = 0; m->swmax = m->prevmax[0]; m->sw2nd = m->prevmax[1]; m->ncand = 0;
  m->prevmax[0] = m->prevmax[1] = 0;
  const uint32_t q0 = m->seed_range[0], q1 = m->seed_range[1];
  m->seed_range[0] = m->seed_range[1] = 0;
  m->nseg = m->nseg_tot = m->nhit = m->nhit_tot = m->max1 = m->max2 = 0;
  m->th_bandwidth_min = m->th_min_swatscor = m->th_scorlen_min = 0;
  m->sc.ncand = 0; m->sc.n_sort = 0;
  keep_clear(m);
  if (len + 2 > m->read_cap) {
    m->read_cap = len + 1024;
    m->read[0] = realloc(m->read[0], m->read_cap); m->read[1] = realloc(m->read[1], m->read_cap);
    m->qual = realloc(m->qual, m->read_cap);
  }
  if (len + 2 > m->qmaskbuf_cap) { m->qmaskbuf_cap = len + 1024; m->qmaskbuf = realloc(m->qmaskbuf, m->qmaskbuf_cap); }
  m->qlen = len;
  /* forward codes; reverse complement keeps non-standard codes (sequence.c:884-896) */
  for (i = 0; i < len; i++) m->read[0][i] = or_code_of((unsigned char) bases[i]);
  for (i = 0; i < len; i++) { uint8_t c = m->read[0][len - 1 - i]; m->read[1][i] = (uint8_t) ((c & 4)? c: 3 - c); }
  if (quals) memcpy(m->qual, quals, len);
  m->err = OR_OK;
  if (len < (uint32_t) ix->k) { m->hi[0].n_seeds = m->hi[1].n_seeds = 0; return OR_OK; }   /* ERRCODE_SHORTSEQ swallowed */

  if (p->flags & OR_FLG_NOSHRTINFO) {          /* initRMAPINFO, rmap.c:1027-1044 (with a stretch of the read: mapSecondary, rmap.c:1483) */
    rv = or_collect_hitinfo(&m->hi[0], ix, 0, 0, p->min_basq, q0, q1, m->read[0], quals? m->qual: NULL, len);
    if (!rv) rv = or_collect_hitinfo(&m->hi[1], ix, 1, 0, p->min_basq, q0, q1, m->read[0], quals? m->qual: NULL, len);
  } else {
    rv = or_collect_hitinfo_short(&m->hi[0], ix, 0, (uint32_t) ((p->ncut > 0)? p->ncut: 0), HASH_MAXNHITS, p->min_basq, m->read[0], quals? m->qual: NULL, len);
    if (!rv) rv = or_collect_hitinfo_short(&m->hi[1], ix, 1, (uint32_t) ((p->ncut > 0)? p->ncut: 0), HASH_MAXNHITS, p->min_basq, m->read[0], quals? m->qual: NULL, len);
  }
  if (rv) { m->err = rv; return rv; }
  rv = map_single_read(m);
  m->err = rv;
  return rv;
}

long or_map_dump_str(const OrMap *m, unsigned long long readno, const char *name, int with_hitlists, char *buf, size_t cap)
{
  char *mem = NULL;
  size_t len = 0;
  FILE *fp = open_memstream(&mem, &len);
  if (!fp) return -1;
  or_map_dump(m, fp, readno, name, with_hitlists);
  fclose(fp);
  if (buf && cap) { size_t c = (len < cap - 1)? len: cap - 1; memcpy(buf, mem, c); buf[c] = 0; }
  free(mem);
  return (long) len;
}

const OrResult *or_map_results(const OrMap *m, int *n, const uint8_t **diffstr)
{
  *n = (int) m->nres;
  if (diffstr) *diffstr = m->diffstr;
  return m->res;
}

const OrCand *or_map_cands(const OrMap *m, int *n_scored)
{
  *n_scored = (int) m->ncand;
  return m->cand;
}

void or_map_stats(const OrMap *m, int out[8])
{
  out[0] = m->swmax; out[1] = m->sw2nd; out[2] = m->nseg; out[3] = m->nseg_tot;
  out[4] = m->nhit; out[5] = m->nhit_tot; out[6] = m->max1; out[7] = m->max2;
}

static void dump_hitinfo(FILE *fp, char strand, const OrHitInfo *hi)
{
  uint32_t i;
  fprintf(fp, "HI %c nseeds=%u rank=%u status=%u\n", strand, hi->n_seeds, hi->seed_rank, (unsigned) hi->status);
  fprintf(fp, "QM %c ", strand);
  for (i = 0; i < hi->qlen; i++) fputc('0' + hi->qmask[i], fp);
  fputc('\n', fp);
  for (i = 0; i < hi->n_seeds; i++) {
    const OrSeed *sp = hi->seed + hi->sidx[i];
    fprintf(fp, "SD %c %u %u %u %u\n", strand, i, sp->qoffs, sp->nhits, sp->posidx);
  }
}

void or_map_dump(const OrMap *m, FILE *fp, unsigned long long readno, const char *name, int with_hitlists)
{
  uint32_t i;
  int j;
  fprintf(fp, "READ %llu %s len=%u err=%d\n", readno, name? name: "-", m->qlen, 0);
  if (m->qlen >= (uint32_t) m->ix->k) {
    dump_hitinfo(fp, 'F', &m->hi[0]);
    dump_hitinfo(fp, 'R', &m->hi[1]);
    for (i = 0; i < m->sc.ncand; i++) {
      const OrSegCand *c = m->sc.cand + i;
      fprintf(fp, "CA %u %u %u %u %u %d %d %d %u %u %d %d\n", i, c->qs, c->qe, c->rs, c->re, (int) c->shiftoffs,
              (int) c->srange, (int) c->shift2mm, c->cover, (unsigned) c->flag, c->nseg, c->seqidx);
    }
    fprintf(fp, "ST %u %u %u %u %u %u %u\n", m->sc.max_cover, m->sc.max2nd_cover, m->sc.ncand, m->sc.n_mincover,
            m->sc.n_sort, m->sc.cover_deficit[0], m->sc.cover_deficit[1]);
    for (i = 0; i < m->sc.n_sort; i++) fprintf(fp, "SI %u %u %u\n", i, m->sc.sort_idx[i], m->sc.sort_keys[i]);
    for (i = 0; i < m->ncand; i++) {
      const OrCand *c = m->cand + i;
      fprintf(fp, "RC %u %u %u %u %llu %llu %d %d %lld %d\n", i, c->flags, c->qs, c->qe, (unsigned long long) c->rs,
              (unsigned long long) c->re, c->band_l, c->band_r, (long long) c->sqidx, c->swscor);
    }
  }
  for (i = 0; i < m->nres; i++) {
    const OrResult *r = m->res + i;
    fprintf(fp, "RS %u %c %d %u %u %llu %llu %lld ", i, r->reverse? 'R': 'F', r->swatscor, r->q_start, r->q_end,
            (unsigned long long) r->s_start, (unsigned long long) r->s_end, (long long) r->sidx);
    for (j = 0; j < r->strlen; j++) fprintf(fp, "%02x", (unsigned) m->diffstr[r->stroffs + j]);
    fputc('\n', fp);
  }
  fprintf(fp, "RX %u %d %d %d %d %u %u\n", m->nres, m->swmax, m->sw2nd, m->nseg, m->nseg_tot, (unsigned) m->nhit, (unsigned) m->nhit_tot);
  if (with_hitlists && m->qlen >= (uint32_t) m->ix->k) {
    const int per = (m->par.flags & OR_FLG_SEQBYSEQ)? (int) m->ix->nseq: 1;
    for (j = 0; j < m->hl_keep_cnt; j++) {
      int n;
      fprintf(fp, "HL %c %d %d", (j >= per)? 'R': 'F', j % per, m->hl_keep_n[j]);
      for (n = 0; n < m->hl_keep_n[j]; n++) fprintf(fp, " %llx", (unsigned long long) m->hl_keep[j][n]);
      fputc('\n', fp);
    }
  }
}
