/* oracle/or_internal.h -- TEST INFRASTRUCTURE: private state of the CPU restatement. */
#ifndef OR_INTERNAL_H
#define OR_INTERNAL_H
#include "smalt_oracle.h"

typedef struct OrHitList {          /* hashhit.c:215 */
  int is_reverse;
  int nhits, nhits_max, nhits_alloc;
  uint64_t *sqdat;
  uint32_t qlen;
  uint8_t *qmask; uint32_t qmask_cap;
} OrHitList;

typedef struct OrSegSeed { uint64_t sqo; int32_t len; } OrSegSeed;          /* segment.c:162 */
typedef struct OrHitRegion { uint32_t idx; int32_t num; } OrHitRegion;      /* segment.c:195 */
typedef struct OrSegment { uint32_t ix; int32_t nseed; uint32_t cover; } OrSegment; /* segment.c:206 */

typedef struct OrSegLst {
  OrHitRegion *hreg; uint32_t nhreg, cap_hreg;
  OrSegSeed *seed;   uint32_t nseed, cap_seed;
  OrSegment *segm;   uint32_t nsegm, cap_segm;
  int is_reverse;
  uint32_t qlen;
} OrSegLst;

typedef struct OrSegCands {         /* segment.c:267 */
  OrSegCand *cand; uint32_t ncand, cap_cand;
  uint32_t *sort_keys, *sort_idx; uint32_t cap_sort;
  uint32_t n_sort, n_mincover, max_cover, max2nd_cover;
  uint32_t cover_deficit[2];
} OrSegCands;

struct OrMap {
  const OrIndex *ix;
  OrHitInfo hi[2];
  OrHitList hl;
  OrSegLst sl;
  OrSegCands sc;
  uint8_t *qmaskbuf; uint32_t qmaskbuf_cap;
  uint8_t *read[2]; uint8_t *qual; uint32_t qlen, read_cap;
  uint8_t *win; uint32_t win_cap;
  OrCand *cand; uint32_t ncand, cap_cand;
  OrResult *res; uint32_t nres, cap_res;
  uint8_t *diffstr; uint32_t ndiff, cap_diff;
  int swmax, sw2nd;
  int nseg, nseg_tot, nhit, nhit_tot, max1, max2;
  int th_bandwidth_min, th_min_swatscor, th_scorlen_min;
  int err;
  /* retained per-(strand,seq) hit lists for dumps */
  uint64_t **hl_keep; int *hl_keep_n; int hl_keep_cnt;
  int keep_hitlists;
  OrParams par;
  /* interval restriction of the current call (rmapPair); niv < 0: none */
  int prevmax[2];            /* running score maxima of the ResultSet the next call appends to (0, 0: a blank set) */
  uint32_t seed_range[2];    /* the next call takes its k-mer words from bases [first, last] only (0, 0: the whole read) */
  int niv, niv_next; const int64_t *iv_sx; const uint32_t *iv_lo, *iv_hi;
};

/* or_seed.c */
void or_hitinfo_init(OrHitInfo *hi, int s);
void or_hitinfo_free(OrHitInfo *hi);
int or_collect_hitinfo(OrHitInfo *hi, const OrIndex *ix, int is_reverse, uint32_t ncut, int min_basq,
                       uint32_t seq_start, uint32_t seq_end, const uint8_t *codes, const uint8_t *qual, uint32_t qlen);
int or_collect_hitinfo_short(OrHitInfo *hi, const OrIndex *ix, int is_reverse, uint32_t ncut, uint32_t maxhit_total,
                             int min_basq, const uint8_t *codes, const uint8_t *qual, uint32_t qlen);
uint32_t or_hitinfo_cover_deficit(const OrHitInfo *hi, const OrIndex *ix);
uint32_t or_hitinfo_hit_numbers(const OrHitInfo *hi, uint32_t *nhit_rank);
int or_collect_hits_segment(OrHitList *hl, OrHitInfo *hi, const OrIndex *ix, uint64_t seg_lo, uint64_t seg_hi,
                            uint32_t nhit_max, int use_short);
int or_collect_hits_cutoff(OrHitList *hl, OrHitInfo *hi, const OrIndex *ix, uint32_t max_nhit_per_tup);

/* or_segment.c */
void or_seglst_fill(OrSegLst *sl, uint32_t min_ktup, const OrHitList *hl, const OrIndex *ix);
void or_segcands_blank(OrSegCands *sc);
int or_segcands_add_fast(OrSegCands *sc, uint8_t *mask, OrSegLst *sl, const OrIndex *ix, uint32_t mincover, int32_t seqidx);
int or_segcands_stats(OrSegCands *sc, const OrIndex *ix, uint32_t min_cover_below_max, const OrHitInfo *hf,
                      const OrHitInfo *hr, uint32_t target_depth, uint32_t max_depth, int is_sensitive);
int or_segcands_offsets(OrCand *c, const OrSegCands *sc, const OrIndex *ix, uint32_t scidx, int edgelen, uint32_t qlen);

#endif
