/* oracle/smalt_oracle.h -- CPU restatement of SMALT 0.7.6's seed-and-extend hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the checker for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (smalt_amd/csrc, libsmaltgpu.so) never includes, links or calls anything in oracle/.
 *
 * Parity pin: every stage below is compared, on seeded inputs, with the per-stage dumps of
 * the unmodified reference (oracle/_ref/refdump, built from /root/reference by
 * oracle/Makefile) -- fixtures under tests/golden/ -- and end-to-end with the known-answer
 * vectors embedded in the reference's own test/bam_cigar_test.py and test/xali_test.py.
 *
 * All citations `file:line` refer to /root/reference/src/.
 */
#ifndef SMALT_ORACLE_H
#define SMALT_ORACLE_H
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- flags (values follow rmap.h:53-65 so that dumps are comparable) ---- */
enum { OR_FLG_BEST = 0x02, OR_FLG_SEQBYSEQ = 0x10, OR_FLG_NOSHRTINFO = 0x20, OR_FLG_SENSITIVE = 0x80,
       OR_FLG_RAWRESULTS = 0x10000 };   /* not a flag of the reference: every alignment of the call is returned, without the duplicate
                                          * handling of resultSetAddFromAli -- for calls that append to a ResultSet, whose caller applies it */
/* hit qualifiers, hashhit.h:57-65 */
enum { OR_HQ_TERM = 0, OR_HQ_NORMHIT = 1, OR_HQ_MULTIHIT = 2, OR_HQ_REPEAT = 3, OR_HQ_NOHIT = 4, OR_HQ_NONSTDNT = 5 };
enum { OR_IDX_PERFECT = 0, OR_IDX_HASH32MIX = 1 };
enum { OR_OK = 0, OR_ERR = 1, OR_ERR_SHORTSEQ = 2, OR_ERR_BAND = 3, OR_ERR_SWATSCOR = 4, OR_ERR_FILE = 5 };

/* ---- I1 + I2: hash index and packed reference ---- */
typedef struct OrIndex {
  int k, s, typ, nbits_key, nbits_lo;
  uint32_t nkeys;      /* 2^nbits_key */
  uint32_t *idx;       /* nkeys+1 */
  uint32_t *pos;       /* npos k-mer serial numbers (= base offset / s) */
  uint32_t npos, maxpos;
  uint32_t nwords;     /* HASH32MIX only */
  uint32_t *wordidx;   /* nwords+1 */
  uint32_t *posidx;    /* nwords+1 */
  int64_t nseq;
  uint64_t *sop;       /* nseq+1 cumulative base offsets (sequence.c:163) */
  uint64_t totlen;
  uint32_t *packed;    /* totlen/10+1 words, 3 bits/base, first base in bits 29..27 */
  char *names;         /* concatenated 0-terminated names */
  uint64_t namsiz;
} OrIndex;

OrIndex *or_index_build(int nseq, const char *const *seqs, const uint32_t *lens,
                        const char *const *names, int k, int s);
int or_index_write(const OrIndex *ix, const char *prefix);      /* <prefix>.sma + .smi */
OrIndex *or_index_read(const char *prefix);
void or_index_free(OrIndex *ix);
/* the on-the-fly index of rmapPair's rescue round (rmap.c:495-517): perfect type over the interval windows only; shares the
 * reference arrays of `main` */
OrIndex *or_index_build_fine(const OrIndex *main, int niv, const int64_t *sx, const uint32_t *lo, const uint32_t *hi, int k, int s);
void or_index_free_fine(OrIndex *ix);
uint32_t or_index_lookup(const OrIndex *ix, uint64_t word, uint32_t *posidx); /* hashidx.c:1146 */
uint32_t or_index_positions(const OrIndex *ix, uint32_t posidx, const uint32_t **posp); /* :1193 */
void or_index_fetch(const OrIndex *ix, uint64_t start, uint32_t len, uint8_t *codes); /* 3-bit codes */

/* ---- mapping parameters (menu.c:593-626 defaults; smalt.c:1185 call) ---- */
typedef struct OrParams {
  int ncut;                  /* ktuple_maxhit, default 10000 */
  uint32_t min_cover;        /* covermin_tuple, default 0 */
  int min_swatscor;          /* default k+s-1 */
  int min_swatscor_below_max;/* -d, default 0 */
  int min_basq;              /* default 0 */
  int target_depth, max_depth; /* 512, 2048 */
  uint32_t flags;            /* OR_FLG_* */
  int match, mismatch, gap_init, gap_ext;  /* +1 -2 -4 -3 */
} OrParams;
void or_params_default(OrParams *p, const OrIndex *ix);

/* ---- per-read working state; all stage outputs stay inspectable ---- */
typedef struct OrSeed { uint32_t posidx, nhits, cix, qoffs; } OrSeed;      /* hashhit.c:148 */
typedef struct OrHitInfo {                                                  /* hashhit.c:164 */
  int is_reverse, status;
  uint32_t qlen, n_seeds, seed_rank;
  uint8_t *qmask, *qbuf;
  OrSeed *seed;
  uint32_t *sidx, *sortkey;
  uint32_t *count;      /* [s] */
  uint32_t *framebuf;   /* ranks per frame */
  uint32_t **frame;     /* [s] */
  uint32_t cap;
} OrHitInfo;

typedef struct OrSegCand {                                                  /* segment.c:239 */
  uint32_t qs, qe, rs, re;
  short shiftoffs, shift2mm, srange;
  uint32_t cover;
  uint8_t flag;
  int32_t nseg;
  uint32_t hregix;
  int32_t seqidx;
} OrSegCand;

typedef struct OrCand {                                                     /* rmap.c:111 */
  uint32_t flags;       /* 1 reverse, 2 scored */
  uint32_t qs, qe;
  uint64_t rs, re;
  int band_l, band_r;
  int64_t sqidx;
  uint32_t dqo; int dro;
  int swscor;
  uint32_t cover;
  int used_simd;        /* which score kernel the predicate picked (K2a=1, K2b=0) */
} OrCand;

typedef struct OrResult {                                                   /* results.c:121 */
  int reverse;
  int swatscor;
  uint32_t q_start, q_end;     /* 1-based on the original read */
  uint64_t s_start, s_end;     /* 1-based in sequence sidx */
  int64_t sidx;
  int stroffs, strlen;         /* into OrMap.diffstr (strlen includes the terminating 0) */
  int cand_first;              /* 1: first alignment of its candidate, i.e. of one resultSetAddFromAli call (results.c:1852) */
} OrResult;

typedef struct OrMap OrMap;
OrMap *or_map_create(const OrIndex *ix);
void or_map_free(OrMap *m);
/* bases: ASCII; quals: ASCII phred+33 or NULL.  Returns OR_OK (also for too-short reads). */
int or_map_single(OrMap *m, const char *bases, const char *quals, uint32_t len, const OrParams *p);
/* One mapSingleRead call of rmapPair with an interval set (rmap.c:1940-1954, :2010-2039): seeding restricted to the windows
 * [lo, hi] (0-based, inclusive) of sequences sx (collectHitsFromInterVal, rmap.c:438-492), in the given order.  niv = 0 is
 * a valid (empty) restriction.  With an index from or_index_build_fine and OR_FLG_NOSHRTINFO this is the rescue round. */
int or_map_single_restricted(OrMap *m, const char *bases, const char *quals, uint32_t len, const OrParams *p,
                             int niv, const int64_t *sx, const uint32_t *lo, const uint32_t *hi);
/* The next call appends to a ResultSet whose running score maxima (UPDATE_SWATSCORMAX, results.c:1013) are these: the
 * traceback pass raises its threshold to the set's second-best score (rmap.c:881-885).  or_map_stats then returns the
 * maxima after the call. */
void or_map_set_prevmax(OrMap *m, int swmax, int sw2nd);
/* The next call (with OR_FLG_NOSHRTINFO) takes its k-mer words from bases [first, last] of the read only (0-based, inclusive):
 * hashCollectHitInfo with a range (hashhit.c:987, :536-551), as mapSecondary (rmap.c:1435-1505) calls it for the part of a
 * split read that its best alignment leaves uncovered. */
void or_map_set_seed_range(OrMap *m, uint32_t first, uint32_t last);
/* calcTotalNumberOfHits (rmap.c:1076) of the last read: what rmapPair compares to pick the mate it maps first */
uint32_t or_map_hit_total(const OrMap *m, int ktuple_maxhit);
/* print the stage state of the last read in the refdump line format */
void or_map_dump(const OrMap *m, FILE *fp, unsigned long long readno, const char *name, int with_hitlists);
/* the same into a buffer; returns the number of bytes needed (excluding the terminating 0) */
long or_map_dump_str(const OrMap *m, unsigned long long readno, const char *name, int with_hitlists, char *buf, size_t cap);

/* accessors for ctypes */
const OrResult *or_map_results(const OrMap *m, int *n, const uint8_t **diffstr);
const OrCand *or_map_cands(const OrMap *m, int *n_scored);
void or_map_stats(const OrMap *m, int out[8]); /* swmax, sw2nd, nseg, nseg_tot, nhit, nhit_tot, max1, max2 */

/* ---- stand-alone kernels (the spec the HIP kernels are tested against) ---- */
/* K2a: textbook Gotoh local max over the whole matrix; swsimd.c:868 */
int or_sw_full(const uint8_t *q, uint32_t qlen, const uint8_t *r, uint32_t rlen,
               const int8_t M[8][8], int gap_init, int gap_ext);
/* K2b: alignment.c:1603 -> 310 + 1029; returns OR_OK or OR_ERR_BAND */
int or_sw_band_fast(int *score, const uint8_t *q, uint32_t qlen, const uint8_t *r, int rlen,
                    const int8_t M[8][8], int gap_init, int gap_ext,
                    int l_edge, int r_edge, int q_left, int q_right, int s_left, int s_right);
typedef struct OrAli { int score, qs, qe, rs, re; int dlen; uint8_t *diffstr; } OrAli;
/* K3: alignment.c:1548; results appended in the reference's order (node, left, right) */
int or_sw_band_full(OrAli **out, int *nout, const uint8_t *q, uint32_t qlen, const uint8_t *r, int rlen,
                    const int8_t M[8][8], int gap_init, int gap_ext, int match_avg,
                    int l_edge, int r_edge, int q_left, int q_right, int s_left, int s_right,
                    int minscore, int minscorlen);
void or_ali_free(OrAli *a, int n);
void or_score_matrix(int8_t M[8][8], int match, int mismatch);   /* score.c:138 */
/* sorts with the reference's tie behaviour (sort.c:233, :415) */
void or_sort2_u32(int n, uint32_t *key, uint32_t *val);
void or_sort_u64(int n, uint64_t *a);
uint8_t or_code_of(unsigned char c);  /* sequence.c:287 */

#ifdef __cplusplus
}
#endif
#endif
