/* oracle/or_seed.c -- TEST INFRASTRUCTURE: restatement of SMALT's seeding stages
 * S1 (k-mer lookup, hashhit.c:480), S2 (rarity ranking + budget, hashhit.c:769,1007),
 * S3 (hit gather + pack + sort, hashhit.c:1416,1593,1691) and of the quicksorts whose tie
 * order is observable (sort.c:233,415).  See smalt_oracle.h. */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include "or_internal.h"

enum { NREPEATS = 4, MINHIT_PER_TUPLE = 16, HITLST_MINSIZ = 8192, HITLST_BLKSZ = 16384,
       HITLST_LOGQLEN_FACT = 32, HITINFO_MINSEEDNUM = 3, HITINFO_MAXCOVER_PERCENT = 80,
       HITINFO_MINCOVER_KMER = 2, HALFBIT = 31 };
enum { HI_REVERSE = 1, HI_SORTED = 2, HI_RANK = 4 };   /* hashhit.c:84-92 */

/* ------------------------------------------------------------------------------------
 * sort.c:233-330 / :415-495 -- median-of-three quicksort with an explicit stack and
 * straight insertion below 7 elements.  Not stable: the order of equal keys is a
 * property of this exact exchange sequence, so it is restated step for step.
 * ------------------------------------------------------------------------------------ */
#define SORT_BODY(KT, HAVE_VAL)                                                          \
  int lo = 0, hi = n - 1, i, j, mid, sp = 0;                                             \
  int stk[64];                                                                           \
  KT pk, tk; uint32_t pv = 0, tv;                                                        \
  (void) pv; (void) tv;                                                                  \
  for (;;) {                                                                             \
    if (hi - lo < 7) {                                                                   \
      for (j = lo + 1; j <= hi; j++) {                                                   \
        pk = key[j]; if (HAVE_VAL) pv = val[j];                                          \
        for (i = j - 1; i >= lo && key[i] > pk; i--) { key[i+1] = key[i]; if (HAVE_VAL) val[i+1] = val[i]; } \
        key[i+1] = pk; if (HAVE_VAL) val[i+1] = pv;                                      \
      }                                                                                  \
      if (!sp) return;                                                                   \
      hi = stk[sp--]; lo = stk[sp--];                                                    \
    } else {                                                                             \
      mid = (lo + hi) >> 1;                                                              \
      SWP(mid, lo + 1);                                                                  \
      if (key[lo] > key[hi]) { SWP(lo, hi); }                                            \
      if (key[lo+1] > key[hi]) { SWP(lo + 1, hi); }                                      \
      if (key[lo] > key[lo+1]) { SWP(lo, lo + 1); }                                      \
      i = lo + 1; j = hi;                                                                \
      pk = key[lo+1]; if (HAVE_VAL) pv = val[lo+1];                                      \
      for (;;) {                                                                         \
        do i++; while (key[i] < pk);                                                     \
        do j--; while (key[j] > pk);                                                     \
        if (j < i) break;                                                                \
        SWP(i, j);                                                                       \
      }                                                                                  \
      key[lo+1] = key[j]; key[j] = pk;                                                   \
      if (HAVE_VAL) { val[lo+1] = val[j]; val[j] = pv; }                                 \
      sp += 2;                                                                           \
      if (sp > 60) return; /* ERRCODE_SORTSTACK in the reference */                      \
      if (hi - i + 1 >= j - lo) { stk[sp] = hi; stk[sp-1] = i; hi = j - 1; }             \
      else { stk[sp] = j - 1; stk[sp-1] = lo; lo = i; }                                  \
    }                                                                                    \
  }

void or_sort2_u32(int n, uint32_t *key, uint32_t *val)
{
#define SWP(a, b) tk = key[a]; key[a] = key[b]; key[b] = tk; tv = val[a]; val[a] = val[b]; val[b] = tv
  SORT_BODY(uint32_t, 1)
#undef SWP
}

void or_sort_u64(int n, uint64_t *key)
{
  uint32_t *val = NULL;
#define SWP(a, b) tk = key[a]; key[a] = key[b]; key[b] = tk
  SORT_BODY(uint64_t, 0)
#undef SWP
}

/* ------------------------------------------------------------------------------------ */
void or_hitinfo_init(OrHitInfo *hi, int s)
{
  memset(hi, 0, sizeof(*hi));
  hi->count = calloc(s, sizeof(uint32_t));
  hi->frame = calloc(s, sizeof(uint32_t *));
}

void or_hitinfo_free(OrHitInfo *hi)
{
  free(hi->qmask); free(hi->qbuf); free(hi->seed); free(hi->sidx); free(hi->sortkey);
  free(hi->count); free(hi->framebuf); free(hi->frame);
}

static void hitinfo_reserve(OrHitInfo *hi, uint32_t qlen, int s)
{
  int o;
  if (qlen + 2 <= hi->cap) return;
  hi->cap = qlen + 2 + 256;
  hi->qmask = realloc(hi->qmask, hi->cap);
  hi->qbuf = realloc(hi->qbuf, hi->cap);
  hi->seed = realloc(hi->seed, hi->cap*sizeof(OrSeed));
  hi->sidx = realloc(hi->sidx, hi->cap*sizeof(uint32_t));
  hi->sortkey = realloc(hi->sortkey, hi->cap*sizeof(uint32_t));
  hi->framebuf = realloc(hi->framebuf, ((size_t) hi->cap + s)*s*sizeof(uint32_t));
  for (o = 0; o < s; o++) hi->frame[o] = hi->framebuf + (size_t) o*(hi->cap + 1);
  memset(hi->sortkey, 0, hi->cap*sizeof(uint32_t));
}

/* S1 -- hashhit.c:480-657 (collectHitInfo).  `codes` are 3-bit codes of the read in its
 * original orientation for BOTH strands: the reverse-complement word is built incrementally
 * (MAKE_NEXT_WORD, :254-259) so that hit positions refer to the forward reference. */
int or_collect_hitinfo(OrHitInfo *hi, const OrIndex *ix, int is_reverse, uint32_t ncut, int min_basq,
                       uint32_t seq_start, uint32_t seq_end, const uint8_t *codes, const uint8_t *qual, uint32_t qlen)
{
  const int k = ix->k;
  const uint64_t wordmask = (((uint64_t) 1) << (2*k)) - 1;
  const int rc_addpos = (k - 1) << 1;
  const int minqval = min_basq + 33;
  int64_t tdrf[NREPEATS];
  uint64_t word = 0;
  uint32_t s, t, nseeds = 0, posidx = 0, nhits;
  int bad = 0, i;

  hi->status = 0;
  if ((uint32_t) k > qlen) return OR_ERR_SHORTSEQ;
  hitinfo_reserve(hi, qlen, ix->s);
  if (is_reverse) hi->status |= HI_REVERSE;
  hi->is_reverse = is_reverse;
  hi->qlen = qlen;
  if (seq_end >= qlen) seq_end = qlen - 1;
  if (seq_end < seq_start + k - 1) { seq_start = 0; seq_end = qlen - 1; }
  for (s = 0; s < seq_start; s++) hi->qmask[s] = OR_HQ_NOHIT;
  for (i = 0; i < NREPEATS; i++) tdrf[i] = -i - 1;       /* :342-346 */

#define NEXT_WORD(c) \
  if (((c) & 4) || (qual && qual[s] < minqval)) bad = k; else if (bad) bad--; \
  if (is_reverse) word = (word >> 2) + (((uint64_t) (((c) ^ 3) & 3)) << rc_addpos); \
  else word = (word << 2) + ((c) & 3);

  t = s = seq_start;
  for (; s < seq_start + k - 1; s++) { NEXT_WORD(codes[s]) }
  for (; s <= seq_end; t++, s++) {
    int64_t w;
    int rep = 0;
    NEXT_WORD(codes[s])
    if (bad) { hi->qmask[t] = OR_HQ_NONSTDNT; continue; }
    w = (int64_t) (word & wordmask);
    for (i = 0; i < NREPEATS; i++) if (w == tdrf[i]) { rep = 1; break; }   /* :325-340 */
    memmove(tdrf + 1, tdrf, (NREPEATS - 1)*sizeof(int64_t));
    tdrf[0] = w;
    if (rep) { hi->qmask[t] = OR_HQ_REPEAT; continue; }
    nhits = or_index_lookup(ix, word, &posidx);
    if (nhits < 1) { hi->qmask[t] = OR_HQ_NOHIT; continue; }
    if (ncut > 0 && nhits > ncut) { hi->qmask[t] = OR_HQ_MULTIHIT; continue; }
    hi->sortkey[nseeds] = nhits;
    hi->qmask[t] = OR_HQ_NORMHIT;
    hi->seed[nseeds].posidx = posidx;
    hi->seed[nseeds].nhits = nhits;
    hi->seed[nseeds].cix = 0;
    hi->seed[nseeds].qoffs = t;
    hi->sidx[nseeds] = nseeds;
    nseeds++;
  }
#undef NEXT_WORD
  for (; t < qlen; t++) hi->qmask[t] = OR_HQ_TERM;
  hi->qmask[qlen] = 0;
  hi->n_seeds = nseeds;
  hi->seed_rank = 0;
  return OR_OK;
}

/* S2 -- hashhit.c:769-891 (getHitInfoMaxRank, build without hashhit_minimise_coverdeficit) */
static void hitinfo_max_rank(OrHitInfo *hi, const OrIndex *ix, uint32_t mincover, uint32_t maxcover, uint32_t maxhit)
{
  const int s = ix->s, k = ix->k;
  uint32_t i, imax, nmax, ntot, cover, n;
  int f;

  for (f = 0; f < s; f++) hi->count[f] = 0;
  for (i = 0; i < hi->n_seeds; i++) {
    f = (int) (hi->seed[hi->sidx[i]].qoffs % s);
    hi->frame[f][hi->count[f]++] = i;            /* the rank, :809 */
  }
  /* largest prefix of the rarity-sorted seeds with summed hits <= maxhit (:822-827; the
   * reference's loop reads one key past the end, which cannot change n) */
  ntot = hi->sortkey[0];
  for (i = 1; i <= hi->n_seeds && ntot <= maxhit; i++)
    if (i < hi->n_seeds) ntot += hi->sortkey[i];
  n = nmax = i - 1;

  for (f = 0; f < s; f++) {
    const uint32_t *ixp = hi->frame[f];
    imax = hi->count[f];
    if (!imax) continue;
    memset(hi->qbuf, 0, hi->qlen);
    cover = 0;
    for (i = 0; i < imax && cover <= maxcover && (cover < mincover || ixp[i] <= n); i++) {
      const OrSeed *sp = hi->seed + hi->sidx[ixp[i]];
      uint32_t q;
      for (q = sp->qoffs; q < sp->qoffs + k - 1; q++)   /* k-1 bases, :873 */
        if (!hi->qbuf[q]) { hi->qbuf[q] = 1; cover++; }
    }
    if (i > 0 && ixp[i-1] > nmax) nmax = ixp[i-1];
  }
  if (nmax < HITINFO_MINSEEDNUM)
    hi->seed_rank = (HITINFO_MINSEEDNUM < hi->n_seeds)? HITINFO_MINSEEDNUM: hi->n_seeds;
  else
    hi->seed_rank = nmax;
}

/* hashhit.c:1007-1080 (hashCollectHitInfoShort) */
int or_collect_hitinfo_short(OrHitInfo *hi, const OrIndex *ix, int is_reverse, uint32_t ncut, uint32_t maxhit_total,
                             int min_basq, const uint8_t *codes, const uint8_t *qual, uint32_t qlen)
{
  uint32_t mincover, maxcover;
  int rv = or_collect_hitinfo(hi, ix, is_reverse, ncut, min_basq, 0, 0, codes, qual, qlen);
  if (rv) return rv;
  if (hi->n_seeds <= 1) {
    hi->status |= HI_SORTED;
    hi->seed_rank = hi->n_seeds;
    return OR_OK;
  }
  or_sort2_u32((int) hi->n_seeds, hi->sortkey, hi->sidx);
  hi->status |= HI_SORTED;
  hi->seed_rank = 0;
  mincover = HITINFO_MINCOVER_KMER*ix->k + ix->s;
  maxcover = qlen*HITINFO_MAXCOVER_PERCENT/100;
  if (maxcover < (uint32_t) (ix->k + ix->s)) maxcover = ix->k + ix->s;
  else if (maxcover > qlen - ix->s) maxcover = qlen - ix->s;
  if (mincover > maxcover) { mincover = 0; maxcover = qlen; }
  hitinfo_max_rank(hi, ix, mincover, maxcover, maxhit_total);
  hi->status |= HI_RANK;
  return OR_OK;
}

/* hashhit.c:1096-1169 (hashCalcHitInfoCoverDeficit) */
uint32_t or_hitinfo_cover_deficit(const OrHitInfo *hi, const OrIndex *ix)
{
  const int s = ix->s, k = ix->k;
  uint32_t deficit, d, i;
  int f;
  if (hi->status & HI_RANK) {
    uint32_t cover, maxcover = 0;
    d = hi->qlen;
    for (f = 0; f < s; f++) {
      const uint32_t *ixp = hi->frame[f];
      uint32_t imax = hi->count[f];
      if (!imax) continue;
      memset(hi->qbuf, 0, hi->qlen);
      cover = 0;
      for (i = 0; i < imax && ixp[i] < hi->seed_rank; i++) {
        const OrSeed *sp = hi->seed + hi->sidx[ixp[i]];
        uint32_t q;
        for (q = sp->qoffs; q < sp->qoffs + k; q++)
          if (!hi->qbuf[q]) { hi->qbuf[q] = 1; cover++; }
      }
      if (cover < d) d = cover;
      if (cover > maxcover) maxcover = cover;
    }
    deficit = maxcover - d + 1;
  } else {
    uint8_t ctr, kk = (uint8_t) (k/s);
    if (kk > 0) kk--;
    deficit = 0;
    for (f = 0; f < s; f++) {
      d = 0;
      for (ctr = 0, i = f; i < hi->qlen; i += s) {
        if (hi->qmask[i] == OR_HQ_NORMHIT) ctr = kk;
        else if (ctr) ctr--;
        else d += s;
      }
      if (d > deficit) deficit = d;
    }
  }
  return deficit;
}

/* hashhit.c:1200-1219 (hashHitInfoCalcHitNumbers) */
uint32_t or_hitinfo_hit_numbers(const OrHitInfo *hi, uint32_t *nhit_rank)
{
  uint32_t i, nr = 0, ns = (hi->seed_rank > 0)? hi->seed_rank: hi->n_seeds;
  for (i = 0; i < ns; i++) nr += hi->sortkey[i];
  *nhit_rank = nr;
  for (; i < hi->n_seeds; i++) nr += hi->sortkey[i];
  return nr;
}

/* hashhit.c:1262-1296 (initHitList).  The reference keeps its allocation across reads (it
 * only grows, in blocks of 16384); with one read length per run this equals the stateless
 * value used here: max(16384, target rounded up to a multiple of 16384). */
static void hitlist_init(OrHitList *hl, const OrHitInfo *hi)
{
  size_t target = (size_t) (hi->qlen*log((double) hi->qlen)*HITLST_LOGQLEN_FACT);
  int alloc;
  if (target > INT32_MAX) target = INT32_MAX; else if (target < HITLST_MINSIZ) target = HITLST_MINSIZ;
  alloc = HITLST_BLKSZ;
  if ((int) target > alloc) alloc = (int) (((target + HITLST_BLKSZ - 1)/HITLST_BLKSZ)*HITLST_BLKSZ);
  if (alloc > hl->nhits_alloc) {
    hl->sqdat = realloc(hl->sqdat, (size_t) alloc*sizeof(uint64_t));
  }
  hl->nhits_alloc = alloc;
  hl->nhits_max = (int) target;
  if (hi->qlen + 1 > hl->qmask_cap) { hl->qmask_cap = hi->qlen + 512; hl->qmask = realloc(hl->qmask, hl->qmask_cap); }
  hl->qlen = hi->qlen;
  hl->nhits = 0;
  memset(hl->qmask, OR_HQ_NOHIT, hl->qlen);        /* blankHitList, :1224-1231 */
  hl->qmask[hl->qlen] = 0;
  hl->is_reverse = hi->is_reverse;
}

/* SET_NEXT_SHIFT, hashhit.c:283-288 */
static inline uint64_t pack_hit(int is_reverse, uint32_t pos, uint32_t q, int s)
{
  const uint64_t offbit = ((uint64_t) 1) << (HALFBIT + 1);
  if (is_reverse) return ((((uint64_t) pos) + q/s) << HALFBIT) + q;
  return (((((uint64_t) pos) | offbit) - q/s) << HALFBIT) + q;
}

/* hashhit.c:1416-1546 (fillHitListFromHitInfoSegment, hhfp == NULL) */
static int fill_segment(OrHitList *hl, OrHitInfo *hi, const OrIndex *ix, uint32_t lo, uint32_t hi_pos,
                        uint32_t maxhit_per_tuple, int use_short)
{
  const uint32_t n_seeds = (use_short && hi->seed_rank > 0)? hi->seed_rank: hi->n_seeds;
  uint32_t n, i, nh, nhits;
  hitlist_init(hl, hi);
  for (n = 0; n < n_seeds; n++) {
    OrSeed *sp = hi->seed + (use_short? hi->sidx[n]: n);
    const uint32_t *posp;
    uint64_t *dst;
    if (maxhit_per_tuple > 0 && hi->sortkey[n] > maxhit_per_tuple) { hi->qmask[sp->qoffs] = OR_HQ_MULTIHIT; continue; }
    nhits = or_index_positions(ix, sp->posidx, &posp);
    if (sp->cix >= nhits) {
      if (posp[nhits-1] < lo) continue;
      sp->cix = 0;
    }
    if (posp[sp->cix] > lo) sp->cix = 0;
    posp += sp->cix;
    nh = nhits - sp->cix;
    for (i = 0; i < nh && posp[i] < lo; i++);
    nh -= i; sp->cix += i; posp += i;
    if (hl->nhits + nh > (uint32_t) hl->nhits_alloc) {
      if (maxhit_per_tuple > 0) return OR_ERR;   /* ERRCODE_ALLOCBOUNDARY */
      hi->qmask[sp->qoffs] = OR_HQ_MULTIHIT;
      continue;
    }
    dst = hl->sqdat + hl->nhits;
    for (i = 0; i < nh && posp[i] < hi_pos; i++) dst[i] = pack_hit(hi->is_reverse, posp[i], sp->qoffs, ix->s);
    sp->cix += i;
    hl->nhits += (int) i;
  }
  return OR_OK;
}

/* S3 (sequence by sequence) -- hashhit.c:1691-1769 (hashCollectHitsForSegment) */
int or_collect_hits_segment(OrHitList *hl, OrHitInfo *hi, const OrIndex *ix, uint64_t seg_lo, uint64_t seg_hi,
                            uint32_t nhit_max, int use_short)
{
  int rv;
  seg_lo /= ix->s;
  seg_hi /= ix->s;
  if (seg_hi > UINT32_MAX) seg_hi = UINT32_MAX;
  do {
    rv = fill_segment(hl, hi, ix, (uint32_t) seg_lo, (uint32_t) seg_hi, nhit_max, use_short);
    nhit_max /= 2;
  } while (rv == OR_ERR && nhit_max > MINHIT_PER_TUPLE);
  or_sort_u64(hl->nhits, hl->sqdat);
  return OR_OK;
}

/* S3 (concatenated reference) -- hashhit.c:1593-1689 (hashCollectHitsUsingCutoff) */
int or_collect_hits_cutoff(OrHitList *hl, OrHitInfo *hi, const OrIndex *ix, uint32_t max_nhit_per_tup)
{
  const uint32_t n_seeds = (hi->seed_rank)? hi->seed_rank: hi->n_seeds;
  int ceiling;
  hitlist_init(hl, hi);
  do {
    uint32_t i, j;
    ceiling = 0;
    hl->nhits = 0;
    memset(hl->qmask, OR_HQ_NOHIT, hl->qlen);
    for (i = 0; i < n_seeds; i++) {
      const uint32_t nh = hi->sortkey[i];
      const OrSeed *sp;
      const uint32_t *posp;
      uint32_t q;
      if (nh < 1) continue;
      sp = hi->seed + hi->sidx[i];
      q = sp->qoffs;
      if (max_nhit_per_tup > 0 && nh > max_nhit_per_tup) { hl->qmask[q] = OR_HQ_MULTIHIT; continue; }
      if ((int) (hl->nhits + nh) > hl->nhits_max) { ceiling = 1; break; }
      if (hl->nhits + (int) nh > hl->nhits_alloc) {
        hl->nhits_alloc = ((hl->nhits + (int) nh + HITLST_BLKSZ - 1)/HITLST_BLKSZ)*HITLST_BLKSZ;
        hl->sqdat = realloc(hl->sqdat, (size_t) hl->nhits_alloc*sizeof(uint64_t));
      }
      or_index_positions(ix, sp->posidx, &posp);
      hl->qmask[q] = OR_HQ_NORMHIT;
      for (j = 0; j < nh; j++) hl->sqdat[hl->nhits + j] = pack_hit(hl->is_reverse, posp[j], q, ix->s);
      hl->nhits += (int) nh;
    }
    max_nhit_per_tup /= 2;
  } while (ceiling && max_nhit_per_tup > MINHIT_PER_TUPLE);
  or_sort_u64(hl->nhits, hl->sqdat);
  return OR_OK;
}
