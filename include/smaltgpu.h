/* include/smaltgpu.h -- C ABI of libsmaltgpu.so: SMALT's seed-and-extend hot path on MI355X.
 *
 * Every entry point below is what the reference's per-read mapping API would bind for this
 * path (plain pointers and sizes only; no torch / HIP types).  `file:line` citations refer to
 * the reference tree (SMALT 0.7.6, src/).  INTEGRATION.md shows the rmap.c-side shim.
 *
 * All functions return 0 on success or a negative SMALTGPU_E* code; smaltgpu_last_error()
 * gives a message.  There is NO CPU fallback: without a HIP device every call fails.
 */
#ifndef SMALTGPU_H
#define SMALTGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  SMALTGPU_OK = 0,
  SMALTGPU_ENODEV = -1,     /* no HIP device / HIP runtime error */
  SMALTGPU_EFILE = -2,      /* cannot read .smi/.sma (ERRCODE_NOFILE / FILEFORM analogue) */
  SMALTGPU_EARG = -3,       /* bad argument (ERRCODE_ARGRANGE analogue) */
  SMALTGPU_ENOMEM = -4,     /* host or device allocation failed (ERRCODE_NOMEM) */
  SMALTGPU_ECAP = -5,       /* a work pool overflowed: smaltgpu_map_batch recovers by itself (reads re-mapped in smaller batches);
                             * from smaltgpu_fetch_* / in stat[].errcode it marks the reads to map again */
  SMALTGPU_EINTERNAL = -6,  /* device-side assertion (ERRCODE_ASSERT analogue) */
  SMALTGPU_ESCORE = -8      /* in stat[].errcode: the traceback of one of the read's alignments does not add up to the score of its pass.
                             * The reference stops at such a read with ERRCODE_SWATSCOR (alignment.c:767, "Inconsistency when calculating
                             * Smith-Waterman scores"): it happens with alignment scores (-S) whose gap extension is much cheaper than
                             * the opening, because the banded passes do not re-open a gap from a gap cell (alignment.c:1126-1193) */
};
/* return values of the mapping calls that mean "the batch is complete, some reads carry a code of their own in stat[].errcode" */
#define SMALTGPU_IS_READ_ERROR(rv) ((rv) == SMALTGPU_ECAP || (rv) == SMALTGPU_EINTERNAL || (rv) == SMALTGPU_ESCORE)

/* RMAP_FLAGS subset honoured on this path (same values as rmap.h:53-65). */
enum {
  SMALTGPU_FLG_BEST = 0x02,
  SMALTGPU_FLG_SPLIT = 0x08,     /* split reads: honoured by smaltgpu_map_pairs (second calls of both mates ahead of the pairing, rmap.c:2073-2097);
                                  * the mapping calls themselves ignore it -- for single reads the two calls are smaltgpu_map_split */
  SMALTGPU_FLG_SEQBYSEQ = 0x10,
  SMALTGPU_FLG_NOSHRTINFO = 0x20,
  SMALTGPU_FLG_SENSITIVE = 0x80
};

typedef struct smaltgpu_index smaltgpu_index;    /* immutable index image resident in HBM */
typedef struct smaltgpu_mapper smaltgpu_mapper;  /* work buffers + stream; one per host thread */

/* Host-side description of an index: replaces struct _HashTable (hashidx.c:105-146) and the
 * SeqSet fields the path reads (sequence.c:148-171).  Pointers may be host or device memory
 * (see `on_device`). */
typedef struct smaltgpu_index_desc {
  int32_t k, s;              /* word length, sampling stride (hashidx.c:108-109) */
  int32_t typ;               /* 0 perfect, 1 hash32mix (hashidx.h:47-50) */
  int32_t nbits_key, nbits_lo;
  uint32_t npos, nwords;
  const uint32_t *idx;       /* 2^nbits_key + 1 */
  const uint32_t *pos;       /* npos */
  const uint32_t *wordidx;   /* nwords + 1 (hash32mix only) */
  const uint32_t *posidx;    /* nwords + 1 (hash32mix only) */
  int64_t nseq;
  const uint64_t *sop;       /* nseq + 1 cumulative base offsets; ALWAYS host memory */
  const uint32_t *packed;    /* totlen/10 + 1 words, 3 bits/base (sequence.c:1360) */
  int32_t on_device;         /* non-zero: idx/pos/wordidx/posidx/packed are device pointers (adopted, not copied) */
} smaltgpu_index_desc;

/* Scalar arguments of rmapSingle (rmap.h:127-145) + the score penalties (score.c:41-47). */
typedef struct smaltgpu_params {
  int32_t ktuple_maxhit;             /* -H, default 10000 */
  uint32_t min_cover;                /* covermin_tuple (smalt.c:1113-1126) */
  int32_t min_swatscor;              /* -m, default k+s-1 (smalt.c:608-615) */
  int32_t min_swatscor_below_max;    /* -d, default 0 */
  int32_t min_basqval;               /* -q, default 0 */
  int32_t target_depth, max_depth;   /* 512, 2048 (smalt.c:60-61) */
  uint32_t rmapflg;                  /* SMALTGPU_FLG_* */
  int32_t match, mismatch, gap_init, gap_ext;  /* +1 -2 -4 -3 */
  double min_cover_frac;             /* -c below 1.01: > 0 overrides min_cover with (uint32_t)(frac * read length) per read (smalt.c:1113-1122) */
} smaltgpu_params;

/* One alignment as left by resultSetAddFromAli (results.c:1852) before sorting/MAPQ. */
typedef struct smaltgpu_result {
  int32_t swatscor;
  uint32_t q_start, q_end;     /* 1-based, on the original read */
  uint64_t s_start, s_end;     /* 1-based, in sequence sidx (concatenated offset if sidx < 0) */
  int32_t sidx;
  uint32_t reverse;            /* bit 0: reverse-complement strand; bit 1 (SMALTGPU_RES_CANDFIRST): first alignment of its
                                * candidate, i.e. of one resultSetAddFromAli call (what a binding needs to append a call's
                                * alignments to a ResultSet that is not empty, results.c:1906-1935) */
  uint32_t stroffs, strlen;    /* DiffStr bytes in `diffstr`, strlen counts the terminating 0 */
} smaltgpu_result;
enum { SMALTGPU_RES_REVERSE = 1, SMALTGPU_RES_CANDFIRST = 2 };

/* Per-read scalars the host post-processing needs (resultSetAlignmentStats rmap.c:1338,
 * resultSetGetMaxSwat results.c:2163). */
typedef struct smaltgpu_readstat {
  int32_t swatscor_max, swatscor_2ndmax;
  int32_t n_ali_done, n_ali_tot;      /* nseg, nseg_tot */
  uint32_t n_hits_used, n_hits_tot;
  int32_t errcode;                    /* 0 or SMALTGPU_E* for this read */
  uint32_t nres;
  int32_t max1scor;                   /* best score of the score pass (rmap.c:1355); < 1: mapSingleRead returned before the
                                       * traceback pass and did NOT re-sort the ResultSet (rmap.c:1376) */
  int32_t errsite;                    /* diagnostic: source line (library build) of the device-side limit or assertion behind errcode; 0 if none */
} smaltgpu_readstat;

typedef struct smaltgpu_batch_out {
  uint32_t nreads;
  const uint64_t *res_off;            /* nreads + 1 offsets into `res` */
  const smaltgpu_result *res;
  const uint8_t *diffstr;
  const smaltgpu_readstat *stat;      /* nreads */
} smaltgpu_batch_out;

/* ---- index (replaces hashTableRead hashidx.c:1257 + seqSetReadBinFil sequence.c:2521) ---- */
int smaltgpu_index_load(smaltgpu_index **out, const char *prefix, int device);
int smaltgpu_index_create(smaltgpu_index **out, const smaltgpu_index_desc *desc, int device);
/* A copy of the image on another device of the node, device to device (hipMemcpyPeer: xGMI), instead of a second load from
 * disk: the reference's worker threads share one read-only HashTable/SeqSet (threads.c:793-985, rmap.h:83); with one image
 * per GPU the others are filled from the first.  (One process per GPU: broadcast the arrays of smaltgpu_index_info with
 * RCCL and adopt them with smaltgpu_index_create, as bench.py does.) */
int smaltgpu_index_clone(smaltgpu_index **out, const smaltgpu_index *src, int device);
void smaltgpu_index_free(smaltgpu_index *ix);
int smaltgpu_index_info(const smaltgpu_index *ix, smaltgpu_index_desc *desc_out); /* device pointers */
/* host copy of the packed reference (desc.packed), made on first request; for smaltgpu_postprocess.  NULL on failure */
const uint32_t *smaltgpu_index_packed_host(const smaltgpu_index *ix);
void smaltgpu_params_default(smaltgpu_params *p, const smaltgpu_index *ix);

/* ---- index construction (replaces `smalt index`: selectHashTyp smalt.c:268-332, hashTableSetUp
 * hashidx.c:829-998, seqSetCompress sequence.c:1360-1424, and the writers hashTableWrite hashidx.c:1214-1255 +
 * seqSetWriteBinFil sequence.c:2448-2519).  `bases`: the reference sequences concatenated, letters as in FASTA (anything
 * but ACGTU in either case counts as N); `seq_off`: nseq + 1 offsets into it (host memory); `names`: nseq strings.
 * _build takes host memory, _build_device bases already in HBM.  The image is built entirely on the device and is
 * ready for smaltgpu_mapper_create; smaltgpu_index_save writes <prefix>.sma / <prefix>.smi byte-compatible with the
 * reference's files.  build_ms (may be NULL) receives the device time of the construction. ---- */
int smaltgpu_index_build(smaltgpu_index **out, int device, const uint8_t *bases, const uint64_t *seq_off, const char *const *names,
                         int64_t nseq, int32_t k, int32_t s, float *build_ms);
int smaltgpu_index_build_device(smaltgpu_index **out, int device, const uint8_t *d_bases, const uint64_t *seq_off, const char *const *names,
                                int64_t nseq, int32_t k, int32_t s, float *build_ms);
int smaltgpu_index_save(const smaltgpu_index *ix, const char *prefix);

/* ---- mapper (replaces rmapCreate rmap.c:1511 / rmapDelete :1597) ---- */
int smaltgpu_mapper_create(smaltgpu_mapper **out, const smaltgpu_index *ix, uint32_t max_batch_reads,
                           uint32_t max_read_len);
/* Same with sizing options (0 = default): the RMap's buffers grow on demand (array.c), the mapper's pools are sized once.
 * cands_per_read: ranked candidates per read of the batch-wide pool (default 640 for large batches; the depth cut of
 * segment.c:1745-1775 leaves ~270 on a human-size reference, at most 2048).  slot_budget_gb: HBM for the scratch slots of
 * the persistent kernels (default 64) -- lower it when several mappers share a device.  A batch that overflows a pool is
 * not an error: smaltgpu_map_batch re-maps the reads that did not fit in smaller batches. */
typedef struct smaltgpu_mapper_opts {
  uint32_t cands_per_read;
  uint32_t slot_budget_gb;
} smaltgpu_mapper_opts;
int smaltgpu_mapper_create_ex(smaltgpu_mapper **out, const smaltgpu_index *ix, uint32_t max_batch_reads,
                              uint32_t max_read_len, const smaltgpu_mapper_opts *opts);
void smaltgpu_mapper_free(smaltgpu_mapper *m);
/* host threads a mapper may use for its own host work (the copy of a batch's results into read order in smaltgpu_fetch_end);
 * default 1 -- the reference's worker owns one thread (threads.c).  smaltgpu_map_pairs sets it from pair_opts.nthreads. */
int smaltgpu_mapper_set_host_threads(smaltgpu_mapper *m, int nthreads);

/* Map a block of reads (the unit processArgBlock smalt.c:1221 hands to a worker).  `bases`:
 * concatenated ASCII reads, `quals`: concatenated phred+33 or NULL, `read_off`: nreads+1
 * offsets.  Replaces nreads calls of rmapSingle (rmap.c:1648) + rmapGetData (:1634).  The
 * returned arrays are owned by the mapper and valid until its next call. */
int smaltgpu_map_batch(smaltgpu_mapper *m, const uint8_t *bases, const uint8_t *quals,
                       const uint64_t *read_off, uint32_t nreads, const smaltgpu_params *par,
                       smaltgpu_batch_out *out);

/* ---- paired reads: the mapSingleRead calls inside rmapPair (rmap.c:1744-2112) ----
 * rmapPair maps the mate with fewer k-mer hits first, then the other mate with its seeding restricted to the intervals the
 * first one implies (setupInterValFromResultSet rmap.c:354, collectHitsFromInterVal :438), and -- depending on the mapping
 * qualities and proper pairs found (results.c / resultpairs.c, host side) -- the second mate again without restriction and
 * the first mate again restricted, seeded against an on-the-fly k=5 s=1 index over the interval windows (setupFineHashTable
 * :495).  A binding runs a block of pairs through these rounds as batches: every round is one smaltgpu_map_batch_ctx call
 * over the reads that take part in it, with the per-read context of the round. */
typedef struct smaltgpu_interval { int32_t sidx; uint32_t lo, hi; } smaltgpu_interval;   /* interval.c:44-49: 0-based, inclusive, in sequence sidx */
typedef struct smaltgpu_callctx {
  const uint64_t *iv_off;        /* nreads + 1 offsets into iv, or NULL: no read is restricted.  An empty list is a valid
                                  * restriction (no seeds).  Intervals as interValPrune leaves them: sorted, disjoint */
  const smaltgpu_interval *iv;
  const int32_t *min_swatscor;   /* per read, overrides par->min_swatscor (rmap.c:2031), or NULL */
  const int32_t *prev_max;       /* per read (swatscor_max, swatscor_2ndmax) of the ResultSet the call appends to, or NULL (blank
                                  * sets): the traceback pass raises its threshold to the set's second-best score
                                  * (rmap.c:881-885); stat[].swatscor_max / _2ndmax return the pair after the call */
  int32_t fine_index;            /* != 0: seed against the on-the-fly index of each read's intervals (needs iv_off);
                                  * hit info is collected in the long form (initRMAPINFO, rmap.c:2024) */
  int32_t raw_alignments;        /* != 0: return every alignment of the call, candidate by candidate (SMALTGPU_RES_CANDFIRST marks the first
                                  * of each), without resultSetAddFromAli's duplicate handling (results.c:1906-1935: an alignment that
                                  * repeats the one before it is taken off again, and the alignment after it is lost).  For calls that
                                  * append to a non-empty ResultSet: the first comparison is with the set's last alignment, which only
                                  * the caller has (smgpost::Table::take_call; resultSetAppendRaw in integration/results_inject.c) */
  const uint32_t *hitlist_len;   /* per read, or NULL: length of the longest read that the reference's hit list has held up to and
                                  * including this call.  The list's capacity only grows (initHitList, hashhit.c:1280-1282) and decides
                                  * where the allocation-boundary protocol sets in (hashhit.c:1497), so a serial `smalt map` over reads
                                  * of different lengths depends on their order; NULL = every read as if it were the first
                                  * (smaltgpu_mapper_set_history keeps this array for the caller) */
  const uint32_t *seed_range;    /* per read (first, last) base, 0-based and inclusive, or NULL: k-mer words are taken from that stretch of
                                  * the read only -- the second call of a split read (mapSecondary, rmap.c:1435-1505 -> collectHitInfo with
                                  * a range, hashhit.c:536-551); a stretch shorter than a word means the whole read, as there.  The
                                  * alignment passes still see the whole read */
} smaltgpu_callctx;
int smaltgpu_map_batch_ctx(smaltgpu_mapper *m, const uint8_t *bases, const uint8_t *quals, const uint64_t *read_off, uint32_t nreads,
                           const smaltgpu_params *par, const smaltgpu_callctx *ctx, smaltgpu_batch_out *out);
/* Serial-order mode of a mapper (off by default): smaltgpu_map_batch / smaltgpu_map_batch_ctx calls without ctx->hitlist_len take the
 * reads of consecutive calls as ONE serial run of `smalt map -n 0` -- the hit-list capacity of a read is that of the longest read
 * of length >= k seen so far, carried from call to call (rmap.c:1123 creates the list once per thread; reads shorter than k return
 * before they reach it, rmap.c:1274).  on = 0 switches the mode off, any call with on != 0 starts a new run. */
int smaltgpu_mapper_set_history(smaltgpu_mapper *m, int on);
/* calcTotalNumberOfHits (rmap.c:1076) of every read: k-mer hits over both strands counting only words with at most
 * par->ktuple_maxhit hits -- what rmapPair compares to decide which mate is mapped first (rmap.c:1866-1870). */
int smaltgpu_hit_totals(smaltgpu_mapper *m, const uint8_t *bases, const uint8_t *quals, const uint64_t *read_off, uint32_t nreads,
                        const smaltgpu_params *par, uint32_t *nhits);

/* The same two calls for rounds whose reads come from two batches that are already resident in HBM (reads and mates of a block of
 * pairs): read i of the round is read ids[i] >> 1 of batch ids[i] & 1; the mapper gathers the round on the device.  Only the
 * offsets are needed on the host (read lengths). */
typedef struct smaltgpu_resident_reads {
  const uint8_t *d_bases[2], *d_quals[2];     /* device; d_quals: both or neither */
  const uint64_t *d_read_off[2];              /* device: nreads[w] + 1 offsets */
  const uint64_t *read_off[2];                /* the same offsets in host memory */
  uint32_t nreads[2];
} smaltgpu_resident_reads;
int smaltgpu_map_batch_ctx_resident(smaltgpu_mapper *m, const smaltgpu_resident_reads *src, const uint32_t *ids, uint32_t nreads,
                                    const smaltgpu_params *par, const smaltgpu_callctx *ctx, smaltgpu_batch_out *out);
int smaltgpu_hit_totals_resident(smaltgpu_mapper *m, const smaltgpu_resident_reads *src, const uint32_t *ids, uint32_t nreads,
                                 const smaltgpu_params *par, uint32_t *nhits);

/* ---- paired reads, whole: rmapPair (rmap.h:175-194, rmap.c:1744-2112) for a BLOCK of pairs -------------------
 * smaltgpu_map_pairs runs the block through the rounds above -- hit totals, first mate, second mate restricted, second mate
 * again, first mate again over the on-the-fly index; each round one batch of smaltgpu_map_batch_ctx -- and takes the decisions
 * between them itself: the post-call pass of every call (smaltgpu_postprocess's rules), the proper-pair probe
 * (resultSetFindProperPairs, resultpairs.c:1162), the search intervals (setupInterValFromResultSet, rmap.c:354).  What comes
 * back is a `smaltgpu_pairs`: per pair the two alignment sets as rmapPair leaves them in rmp->rsrp / rmp->rsmp and the pair
 * flags (*pairflgp).  smaltgpu_report_emit_pairs does the rest of the reference's flow for a pair: resultSetFindPairs
 * (resultpairs.c:1116), the output filters, resultSetAddPairToReport (:1222) and the CIGAR / SAM lines of both mates
 * (report.c:1758).  `par` as for single reads (rmapPair maps best-only: min_swatscor_below_max is taken as 0); reads and mates
 * in the batch layout of smaltgpu_map_batch, pair i = (read i of the first, read i of the second batch). */
enum { SMALTGPU_LIB_PE = 1, SMALTGPU_LIB_MP = 2, SMALTGPU_LIB_PP = 3, SMALTGPU_LIB_ANY = 4 };   /* -l pe | mp | pp (RSLTPAIRLIB_*, resultpairs.h:68-82) */
typedef struct smaltgpu_pair_opts {
  int32_t insert_min, insert_max;      /* -j, -i: d_min, d_max of rmapPair */
  int32_t library;                     /* SMALTGPU_LIB_* */
  int32_t every_pair;                  /* != 0: the unrestricted round for every pair (-x: RMAPFLG_ALLPAIR, smalt.c:533) */
  int32_t nthreads;                    /* host threads for the work between the rounds */
} smaltgpu_pair_opts;
typedef struct smaltgpu_pair_info {
  uint8_t pairflg;                     /* RSLTPAIRFLG_* (resultpairs.h:52-66): PAIRED 0x01, RAREMATE 0x02, RESTRICT_1st 0x04, RESTRICT_2nd 0x08 */
  uint8_t rounds;                      /* bit r: the pair took part in round r (0 first mate, 1 second mate restricted, 2 second mate again, 3 first mate again) */
  uint16_t nali[2];                    /* alignments that survived the post-call pass, read and mate */
} smaltgpu_pair_info;
typedef struct smaltgpu_pairs smaltgpu_pairs;
smaltgpu_pairs *smaltgpu_pairs_create(void);
void smaltgpu_pairs_free(smaltgpu_pairs *p);
int smaltgpu_map_pairs(smaltgpu_mapper *m, const uint8_t *bases1, const uint8_t *quals1, const uint64_t *read_off1, const uint8_t *bases2,
                       const uint8_t *quals2, const uint64_t *read_off2, uint32_t npairs, const smaltgpu_params *par, const smaltgpu_pair_opts *po,
                       smaltgpu_pairs *out);
/* Same with the reads and mates of the block resident in HBM (bench.py: the timed region starts with the inputs on the device).
 * `src` as above; host copies of the bases / qualities are optional: `bases1/2` are needed when alignments can cross reference
 * sequences (concatenated mode: the pieces are scored again on the host), `quals1/2` for the base-quality rule among equally
 * good alignments (results.c:1247-1285; without them reads count as FASTA input there). */
int smaltgpu_map_pairs_resident(smaltgpu_mapper *m, const smaltgpu_resident_reads *src, const uint8_t *bases1, const uint8_t *quals1, const uint8_t *bases2,
                                const uint8_t *quals2, uint32_t npairs, const smaltgpu_params *par, const smaltgpu_pair_opts *po, smaltgpu_pairs *out);
/* what a block looked like: per pair the flags and counts above; calls[4] (may be NULL) = mapping calls per round; round_ms[4]
 * (may be NULL) = host wall time of each round's batch incl. its copies */
int smaltgpu_pairs_info(const smaltgpu_pairs *p, uint32_t *npairs, const smaltgpu_pair_info **info, uint64_t *calls, double *round_ms);
/* device time per kernel (ms[5][16]: rounds 0-3 and the hit totals, kernels in the order of smaltgpu_timer_name) and the mapper's
 * work counters (work[5][32]) summed over the block's batches; either may be NULL.  For bench.py's roofline object. */
int smaltgpu_pairs_timers(const smaltgpu_pairs *p, double *kernel_ms, uint64_t *work);
/* host wall time [ms] of the work between the rounds of the block: passes behind round A (with the search intervals), behind B, the
 * proper-pair probe, behind C, the plan of round D, behind D, the hit-totals batches, and the whole call; returns the number of entries (8) */
int smaltgpu_pairs_host_times(const smaltgpu_pairs *p, double *ms, int n);
/* the index a mapper was created on, and the batch it was sized for (smaltgpu_map_pairs takes blocks of up to max_batch_reads pairs) */
const smaltgpu_index *smaltgpu_mapper_index(const smaltgpu_mapper *m);
int smaltgpu_mapper_capacity(const smaltgpu_mapper *m, uint32_t *max_batch_reads, uint32_t *max_read_len, uint64_t *max_bases);

/* Same with the inputs already resident in HBM (device pointers); results stay on the device
 * until smaltgpu_fetch_results().  Used by bench.py so that the timed region starts with the
 * reads in HBM. */
int smaltgpu_map_batch_device(smaltgpu_mapper *m, const uint8_t *d_bases, const uint8_t *d_quals,
                              const uint64_t *d_read_off, uint32_t nreads, uint64_t total_bases,
                              const smaltgpu_params *par);
int smaltgpu_fetch_results(smaltgpu_mapper *m, smaltgpu_batch_out *out);
/* The same in two steps, so that the device does not wait for the host between batches: _begin waits for the batch and
 * enqueues the copies of its results; the next smaltgpu_map_batch_device may follow at once; _end hands the results out
 * (valid until the next _begin on this mapper). */
int smaltgpu_fetch_begin(smaltgpu_mapper *m);
int smaltgpu_fetch_end(smaltgpu_mapper *m, smaltgpu_batch_out *out);
int smaltgpu_synchronize(smaltgpu_mapper *m);

/* ---- result post-processing: what resultSetSortAndAssignSequence (results.c:2022) does to the raw
 * alignments of every read of a batch -- assignSequenceIndex (:1695: concatenated mode, sequence and offsets),
 * sortAndPrune (:759: contained alignments out, output order, score ranks), labelComplementarySegments (:707) and
 * calcPhredScaledMappingQuality (:1143, with propagateMapQualAsProb :1343).  Host code on worker threads: the mapping
 * quality is double arithmetic through libm and the orders are libc qsort's on the reference's comparators, so
 * bit-identical results need the same libm / libc the reference runs on.  `raw` is a batch as smaltgpu_map_batch returns it
 * (alignments of ONE mapSingleRead call per read), `quals` / `read_off` the reads' phred+33 qualities (NULL: FASTA input),
 * `sop` the nseq + 1 cumulative sequence offsets (smaltgpu_index_info).  Concatenated mode: an alignment that spans several
 * reference sequences is cut at the junctions and its fragments re-scored (splitMultiSpan, results.c:1472: the fragments are
 * appended to the read's results, out->diffstr then holds their strings too) when `bases` (the reads, ASCII, by read_off),
 * `packed_host` (a host copy of the packed reference, sequence.c:1360) and `par` (penalties) are given; with any of them
 * NULL such a read is flagged needs_reference and left alone. ---- */
typedef struct smaltgpu_post_result {  /* struct _RESULT (results.c:121-160) after the post-processing */
  int32_t swatscor;
  uint32_t q_start, q_end;
  uint64_t s_start, s_end;             /* relative to sequence sidx once it is assigned */
  int32_t sidx;
  uint32_t status;                     /* RSLTFLAG_* (results.h:67-78): SELECT 0x01, REVERSE 0x04, NOSEQID 0x08, SINGLE 0x100 */
  int32_t mapscor;                     /* PHRED-scaled mapping quality */
  double prob;
  int16_t rsltx, qsegx, swrank, pad;
  uint32_t stroffs, strlen;
} smaltgpu_post_result;
typedef struct smaltgpu_post_out {
  uint32_t nreads;
  const uint64_t *res_off;             /* nreads + 1 offsets into res: the result array of read i in its original order */
  const smaltgpu_post_result *res;
  const uint8_t *diffstr;              /* raw->diffstr, or a copy of it with the strings of split fragments behind */
  const uint64_t *sort_off;            /* nreads + 1 offsets into sortr / segsrtr */
  const int32_t *sortr;                /* ResultSet.sortr: indices into the read's results, by decreasing score (results.c:478) */
  const int32_t *segsrtr;              /* ResultSet.segsrtr: by read segment, then score */
  const uint64_t *seg_off;             /* nreads + 1 offsets into segnor */
  const int32_t *segnor;               /* ResultSet.segnor: qsegno + 1 bounds of the segments in segsrtr */
  const int32_t *qsegno;               /* per read */
  const uint32_t *setstatus;           /* per read: RSLTSETFLG_* (results.c:93-100) */
  const int32_t *needs_reference;      /* per read: 1 = left to the caller (alignment across a sequence junction) */
} smaltgpu_post_out;
typedef struct smaltgpu_post smaltgpu_post;     /* owns the arrays of a smaltgpu_post_out */
smaltgpu_post *smaltgpu_post_create(void);
void smaltgpu_post_free(smaltgpu_post *p);
int smaltgpu_postprocess(smaltgpu_post *p, const uint64_t *sop, int64_t nseq, const smaltgpu_batch_out *raw, const uint8_t *bases,
                         const uint8_t *quals, const uint64_t *read_off, const uint32_t *packed_host, const smaltgpu_params *par, int nthreads,
                         smaltgpu_post_out *out);

/* ---- read ingest and report emit (host code, no device; smg_report.cpp) ------------------------------------
 * The reference reads FASTQ / FASTA through one reader thread (seqFastqRead, sequence.c:1960; readHeader :1056, readSeqFast
 * :1229) and prints through the main thread (reportWrite, report.c:1758).  Here a text buffer (a whole file or a chunk of it)
 * becomes the batch layout of smaltgpu_map_batch in one call, and the post-processed results of a batch become the text
 * `smalt map` prints for these reads (single reads; CIGAR and SAM formats), both with worker threads.
 *
 * smaltgpu_reads_parse: `text[0..len)`; is_last = 0: a record that may continue behind the buffer is left (view->consumed
 * says how far the text was used: call again from there with more text); max_reads = 0: no limit.  Plain four-line FASTQ is
 * split over `nthreads`; anything else (FASTA, wrapped lines, blank lines) takes the reference's rules on one thread.
 * The view's arrays belong to `rs` and hold until the next call: bases as the reference's codec prints them (upper case,
 * U -> T, non-letters N; smaltgpu_map_batch takes them as they are), quality characters as in the file (NULL without any:
 * a block with a FASTA record has none), names = the first word of each header, NUL-terminated, by name_off. */
typedef struct smaltgpu_reads smaltgpu_reads;
typedef struct smaltgpu_reads_view {
  uint32_t nreads, has_qual;
  const uint8_t *bases, *quals;
  const uint64_t *read_off;            /* nreads + 1 */
  const char *names;
  const uint64_t *name_off;            /* nreads + 1 */
  uint64_t consumed;                   /* bytes of text the reads came from */
} smaltgpu_reads_view;
smaltgpu_reads *smaltgpu_reads_create(void);
void smaltgpu_reads_free(smaltgpu_reads *rs);
int smaltgpu_reads_parse(smaltgpu_reads *rs, const char *text, uint64_t len, int is_last, uint32_t max_reads, int nthreads,
                         smaltgpu_reads_view *view);

/* ---- split reads (smalt map -p): rmapSingle with RMAPFLG_SPLIT (rmap.c:1716-1728 -> mapSecondary, rmap.c:1435-1505) for a batch ----
 * Every read is mapped; where the best alignment of the read's first segment leaves a stretch of at least a word and a step
 * uncovered, the read is mapped once more with k-mer words from that stretch only (smaltgpu_callctx.seed_range), into the same
 * alignment set, and the set is put in order again.  The result has the layout of smaltgpu_postprocess (which this call replaces
 * for split reads) and goes to smaltgpu_report_emit with SMALTGPU_OUT_SPLIT.  par->rmapflg as smalt.c:508 sets it for -p:
 * NOSHRTINFO | SENSITIVE on top of the usual flags.  n_second_calls (may be NULL): how many reads got the second call. */
int smaltgpu_map_split(smaltgpu_mapper *m, smaltgpu_post *post, const uint8_t *bases, const uint8_t *quals, const uint64_t *read_off, uint32_t nreads,
                       const smaltgpu_params *par, const smaltgpu_index *ix, int nthreads, smaltgpu_post_out *out, uint32_t *n_second_calls);

/* Report: which alignments of a read are printed (resultSetFilterResults, results.c:2592; resultSetAddToReport, results.c:2282;
 * reportAddMap's duplicate test, report.c:545) and the lines themselves (fprintREPALIcigar report.c:711, fprintREPALIsam
 * report.c:762, writeDiffStrCIGAR diffstr.c:298, SAM header report.c:1266). */
enum { SMALTGPU_FMT_CIGAR = 0, SMALTGPU_FMT_SAM = 1, SMALTGPU_FMT_SSAHA = 2 };           /* -f cigar | sam | ssaha (REPORTFMT_*, report.h:46-52; fprintREPALIssaha
                                                                                          * report.c:579).  -f gff ends the reference program with a memory
                                                                                          * fault on its first read (report.c:1389-1401 drops the block list
                                                                                          * it has just made), -f bam needs a library this build has not */
enum { SMALTGPU_REP_SOFTCLIP = 0x02, SMALTGPU_REP_HEADER = 0x04, SMALTGPU_REP_XMISMATCH = 0x08 };   /* REPORTMODIF_* (report.h:54-59) */
enum { SMALTGPU_OUT_BEST = 0x01, SMALTGPU_OUT_SINGLE = 0x02, SMALTGPU_OUT_SPLIT = 0x04, SMALTGPU_OUT_RANDSEL = 0x08 };   /* RESULTFLG_* (results.h:55-63); SPLIT: the best
                                                                                          * alignments of the other read segments follow as partial ones (results.c:2250-2278, :2337) */
typedef struct smaltgpu_report_opts {
  int32_t format;
  uint32_t modflags, outflags;
  int32_t min_swscor;                  /* resultSetFilterData (smalt.c:490): the -m value, 18 without one -- NOT the mapping threshold k+s-1 */
  int32_t min_swscor_below_max;        /* -d */
  double min_identity;                 /* -y: fraction of the read length if <= 1, else bases */
} smaltgpu_report_opts;
typedef struct smaltgpu_report smaltgpu_report;      /* owns the text it hands out */
smaltgpu_report *smaltgpu_report_create(void);
void smaltgpu_report_free(smaltgpu_report *rp);
/* text in front of the first read (the SAM header; empty for the other formats); the report keeps the sequence lengths, which
 * SSAHA lines print: call it once before the emit functions.  seqnames / sop / nseq: smaltgpu_index_seqnames */
int smaltgpu_report_header(smaltgpu_report *rp, const char *const *seqnames, const uint64_t *sop, int64_t nseq, const smaltgpu_report_opts *op,
                           const char *prognam, const char *version, int argc, const char *const *argv, const char **text, uint64_t *len);
/* the lines of a batch: post = smaltgpu_postprocess of `raw` (raw may be NULL: it only supplies the per-read error codes),
 * reads = the parsed block the batch was mapped from.  A random choice among equal best alignments (SMALTGPU_OUT_RANDSEL,
 * -r <seed>) draws from drand48() in read order: seed it with srand48 as RANSEED does (randef.h:19). */
int smaltgpu_report_emit(smaltgpu_report *rp, const smaltgpu_post_out *post, const smaltgpu_batch_out *raw, const smaltgpu_reads_view *reads,
                         const char *const *seqnames, int64_t nseq, const smaltgpu_report_opts *op, int nthreads, const char **text, uint64_t *len);
/* the lines of a block of pairs: both mates of pair 0, both mates of pair 1, ... (reportWrite, report.c:1758).  `reads` / `mates` =
 * the parsed blocks the pairs were mapped from.  Random choices (SMALTGPU_OUT_RANDSEL) draw from drand48() in pair order. */
int smaltgpu_report_emit_pairs(smaltgpu_report *rp, const smaltgpu_pairs *pairs, const smaltgpu_reads_view *reads, const smaltgpu_reads_view *mates,
                               const char *const *seqnames, int64_t nseq, const smaltgpu_report_opts *op, const smaltgpu_pair_opts *po, int nthreads,
                               const char **text, uint64_t *len);
/* names and offsets of the reference sequences of an index (for the two calls above); the arrays belong to the index */
int smaltgpu_index_seqnames(const smaltgpu_index *ix, const char *const **names, const uint64_t **sop, int64_t *nseq);

/* Per-kernel device time (ms, HIP events on the mapper's stream) and work counters of the last
 * batch: names in smaltgpu_timer_name().  For bench.py's roofline object. */
int smaltgpu_timers(const smaltgpu_mapper *m, double *ms, uint64_t *work, int n);
const char *smaltgpu_timer_name(int i);

/* Diagnostic: print the per-stage state of read `i` of the last batch in the line format of
 * oracle/DUMPFORMAT.md into buf (returns bytes needed).  Requires smaltgpu_set_debug(m, 1)
 * before the batch so that intermediate buffers are retained. */
int smaltgpu_set_debug(smaltgpu_mapper *m, int level);
long smaltgpu_dump_read(smaltgpu_mapper *m, uint32_t i, const char *name, char *buf, size_t bufsiz);

/* Stand-alone kernel entry points used by the parity tests (host pointers in, host out). */
/* K2a (swsimd.c:868, un-banded score pass) over explicit 3-bit code arrays.  packed16 = 1: the kernel that scores two
 * tasks per lane group in 16-bit halves (a task whose query holds non-ACGT codes reports -2: the mapper routes those
 * to the 32-bit kernel); packed16 = 0: the 32-bit kernel.  -1: task too long for the register tiling. */
int smaltgpu_sw_full_batch(smaltgpu_mapper *m, const uint8_t *qcodes, const uint32_t *q_off, const uint8_t *rcodes,
                           const uint32_t *r_off, uint32_t ntask, const smaltgpu_params *par, int32_t *scores, int packed16);

/* The candidate ranking sort of segAliCandsStats (segment.c:1733 -> sort.c:233): `narr` arrays of keys (< 1024), array t
 * in keys[off[t]..off[t+1]); returns the keys and the permutation (index into the array) in the reference's tie order
 * for ranks < nneed (all ranks if nneed < 0; ranks at or beyond nneed may be left unsorted).  in_lds: sort in LDS. */
int smaltgpu_rank_sort_batch(smaltgpu_mapper *m, const uint32_t *keys, const uint32_t *off, uint32_t narr, int nneed, int in_lds,
                             uint32_t *out_keys, uint32_t *out_idx);

const char *smaltgpu_last_error(void);
int smaltgpu_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
