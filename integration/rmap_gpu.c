/* rmap_gpu.c -- the reference-side binding of libsmaltgpu: replaces the reference's
 * rmapSingle() (src/rmap.c:1648) by a version that runs the seed-and-extend path on the GPU through the C ABI of
 * include/smaltgpu.h and hands the raw alignments back to the reference's own, unmodified post-processing
 * (results.c: assignSequenceIndex, sortAndPrune, mapping quality, filters) and reporting code.
 *
 * This is OUR source.  It is compiled only by oracle/Makefile (target ref_gpu), which takes the text of the
 * reference's rmap.c from where it lies at build time (nothing is copied into this repository) so that every
 * other symbol of that file -- rmapCreate, rmapDelete, rmapGetData, rmapPair, the profile helpers -- stays the
 * reference's own code.  Used by tests/test_gpu_dropin.py: `smalt map` built this way must print what the
 * reference prints.
 *
 * Environment: SMALTGPU_INDEX_PREFIX = the index prefix given to `smalt map` (the library reads the same
 * .smi/.sma files; a production binding would hand over the HashTable/SeqSet arrays instead).
 * One read per call: smalt.c consumes the results of a read before it maps the next one (smalt.c:1172-1185),
 * so without the additive block hook of INTEGRATION.md section 2 the batch size is 1 -- correct, not fast.
 */
#define rmapSingle rmapSingle_reference_cpu          /* the CPU implementation stays linkable under this name */
#include "rmap.c"
#undef rmapSingle

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include "smaltgpu.h"
#include "gpu_combine.h"

/* results_inject.c (our TU around the reference's results.c) */
extern int resultSetInjectRaw(ResultSet *rsp, unsigned n, const smaltgpu_result *res, const unsigned char *dstr,
                              int swatscor_max, int swatscor_2ndmax);
extern int resultSetInjectPost(ResultSet *rsp, unsigned n, const smaltgpu_post_result *pr, const unsigned char *pdstr, unsigned nsort, const int32_t *sortr,
                               const int32_t *segsrtr, unsigned nsegnor, const int32_t *segnor, int qsegno, unsigned setstatus);
extern int resultSetAppendRaw(ResultSet *rsp, unsigned n, const smaltgpu_result *res, const unsigned char *dstr,
                              int swatscor_max, int swatscor_2ndmax);

enum { GPU_MAXMAPPERS = 256 };
static int g_max_intervals = 2048;              /* search intervals a restricted device call takes per read (SMALTGPU_MAX_INTERVALS lowers it: test hook) */
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
enum { GPU_MAXDEV = 16 };
static smaltgpu_index *g_ixdev[GPU_MAXDEV];      /* one index image per device, shared by the mappers on it */
static int g_ndev = 0, g_nphys = 0;              /* index images (logical devices) and GPUs present */
static int g_libpost = 1;                        /* SMALTGPU_REF_POST unset: result post-processing by the library */
static int g_combine = 1;                        /* SMALTGPU_NO_COMBINE unset */
static const char *g_prefix = NULL;              /* SMALTGPU_INDEX_PREFIX */
#define g_ix (g_ixdev[0])
static struct {
  const RMap *rmp; smaltgpu_mapper *mp; uint32_t maxlen, maxreads;
  char *bases, *quals; uint64_t *off; size_t basecap;
  smaltgpu_batch_out out; int nbatch;          /* results of the last rmapGpuBatch */
  GpuCombOut comb; int use_comb;               /* ... when they came from a combined batch (gpu_combine.c) */
  smaltgpu_post *post; smaltgpu_post_out pout; int have_post;   /* result post-processing of the last batch by the library (N1) */
  uint64_t *sop; int64_t nseq;
  struct GpuPair_ *pairs; int npairs, cap_pairs; /* paired blocks (rmapGpuPairBatch): per-pair state incl. its own two ResultSets */
  struct { int ktuple_maxhit, min_swatscor; double tupcovmin; UCHAR min_basqval; short target_depth, max_depth; RMAPFLG_t rmapflg;
           const ScoreMatrix *scormtxp; const HashTable *htp; const SeqSet *ssp; const SeqCodec *codecp; } pair_args;   /* for pairs left to the CPU */
  ResultSet *save_rsr, *save_rsm;
} g_map[GPU_MAXMAPPERS];
static int g_nmap = 0;

/* SMALTGPU_TIMING=1: seconds per phase of the binding, summed over the worker threads, on stderr at exit */
#include <time.h>
static int g_timing = 0;
enum { TM_STAGE, TM_GPU, TM_POST, TM_INJECT, TM_REFPOST, TM_FILTER, TM_N };
static double g_tm[GPU_MAXMAPPERS][TM_N], g_t_ready = 0.0;      /* g_t_ready: the index images are on the devices */
static const char *const TM_NAME[TM_N] = {"stage", "gpu_wait", "lib_post", "inject", "ref_post", "filter"};
static double tmNow(void) { struct timespec ts; if (!g_timing) return 0.0; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static int g_most_intervals = 0;                /* largest number of search intervals any read had (diagnostic, under g_lock) */
static long g_pairs_on_cpu = 0;                 /* pairs beyond the interval limit, mapped by the reference's rmapPair (counted under g_lock) */
static void tmReport(void)
{
  int i, j;
  fprintf(stderr, "smaltgpu timing: pairs left to the reference's rmapPair (more search intervals than a device call takes): %ld\n", g_pairs_on_cpu);
  fprintf(stderr, "smaltgpu timing: most search intervals of a read: %d\n", g_most_intervals);
  fprintf(stderr, "smaltgpu timing: window    %8.3f s from the index being resident to exit\n", tmNow() - g_t_ready);
  for (j = 0; j < TM_N; j++) { double t = 0.0; for (i = 0; i < g_nmap; i++) t += g_tm[i][j]; fprintf(stderr, "smaltgpu timing: %-9s %8.3f thread-s over %d workers\n", TM_NAME[j], t, g_nmap); }
}

/* the environment is read once, before any mapper exists; worker threads never call setenv */
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void gpuReadConfig(void)
{
  const char *e = getenv("SMALTGPU_NDEV");       /* worker threads are dealt round-robin to the devices (SMALTGPU_NDEV limits them) */
  g_combine = !getenv("SMALTGPU_NO_COMBINE");
  g_libpost = !getenv("SMALTGPU_REF_POST");
  g_prefix = getenv("SMALTGPU_INDEX_PREFIX");
  if (getenv("SMALTGPU_TIMING")) { g_timing = 1; atexit(tmReport); }
  if (getenv("SMALTGPU_MAX_INTERVALS") && atoi(getenv("SMALTGPU_MAX_INTERVALS")) > 0 && atoi(getenv("SMALTGPU_MAX_INTERVALS")) < g_max_intervals) g_max_intervals = atoi(getenv("SMALTGPU_MAX_INTERVALS"));
  g_nphys = g_ndev = smaltgpu_device_count();
  if (e && atoi(e) > 0) g_ndev = atoi(e);        /* more than there are GPUs: images share devices (rehearsal of the N-device path on one GPU) */
  if (g_ndev > GPU_MAXDEV) g_ndev = GPU_MAXDEV;
  if (g_nphys < 1) g_ndev = 0;
}

/* A GPU-side failure ends the program the way the reference ends on any mapping error (ERRMSGNO -> exit, elib.c:335-353).
 * Before that: no further batch is launched and the batches in flight finish (gpuCombineClose), so that the process is not
 * torn down under running launches.  A worker that is turned away meanwhile (GPUCOMB_CLOSING) waits for the end instead of
 * exiting a second time. */
#include <unistd.h>
static void gpuFailOrderly(const char *what)
{
  if (what && *what) fprintf(stderr, "smaltgpu: %s\n", what);
  gpuCombineClose();
}
static void gpuParkIfClosing(int rv) { if (rv == GPUCOMB_CLOSING) for (;;) pause(); }

static int gpuMapperFor(const RMap *rmp, uint32_t rlen);
static int gpuMapperForBatch(const RMap *rmp, uint32_t rlen, uint32_t nreads, size_t nbases, int need_mapper);

static int gpuMapperFor(const RMap *rmp, uint32_t rlen) { return gpuMapperForBatch(rmp, rlen, 1, rlen, 1); }

static int gpuMapperForBatch(const RMap *rmp, uint32_t rlen, uint32_t nreads, size_t nbases, int need_mapper)
{
  int i, slot = -1;
  pthread_once(&g_once, gpuReadConfig);
  pthread_mutex_lock(&g_lock);
  if (g_ndev < 1) { pthread_mutex_unlock(&g_lock); return -1; }
  for (i = 0; i < g_nmap; i++) if (g_map[i].rmp == rmp) { slot = i; break; }
  if (slot < 0 && g_nmap < GPU_MAXMAPPERS) { slot = g_nmap++; memset(&g_map[slot], 0, sizeof(g_map[slot])); g_map[slot].rmp = rmp; }
  if (slot >= 0 && !g_ixdev[0]) {
    /* the index files are read ONCE, into the first GPU; the other images are copied from it device to device
     * (smaltgpu_index_clone: hipMemcpyPeer over xGMI) -- the workers of the reference share one read-only index */
    int dv;
    if (!g_prefix || smaltgpu_index_load(&g_ixdev[0], g_prefix, 0)) { pthread_mutex_unlock(&g_lock); return -1; }
    for (dv = 1; dv < g_ndev; dv++)
      if (smaltgpu_index_clone(&g_ixdev[dv], g_ixdev[0], dv % g_nphys)) { pthread_mutex_unlock(&g_lock); return -1; }
    g_t_ready = tmNow();
  }
  pthread_mutex_unlock(&g_lock);
  if (slot < 0) return -1;
  if (!need_mapper) {                          /* combined batches: only the host staging buffers of this worker */
    if (g_map[slot].maxreads < nreads) {
      uint32_t rcap = g_map[slot].maxreads > 1 ? g_map[slot].maxreads : 1;
      while (rcap < nreads) rcap *= 2;
      if (g_map[slot].mp) { smaltgpu_mapper_free(g_map[slot].mp); g_map[slot].mp = NULL; g_map[slot].maxlen = 0; }
      free(g_map[slot].off);
      if (!(g_map[slot].off = malloc(((size_t)rcap + 1) * sizeof(uint64_t)))) return -1;
      g_map[slot].maxreads = rcap;
    }
  } else if (!g_map[slot].mp || g_map[slot].maxlen < rlen || g_map[slot].maxreads < nreads) {
    uint32_t cap = g_map[slot].maxlen > 64 ? g_map[slot].maxlen : 64, rcap = g_map[slot].maxreads > 1 ? g_map[slot].maxreads : 1;
    if (cap < rlen) cap = (rlen + 31u) & ~31u;                /* scratch is sized by the longest read */
    smaltgpu_mapper_opts opts = { 0, 6 };                 /* one mapper per worker thread shares the device: 6 GB of scratch slots each */
    while (rcap < nreads) rcap *= 2;
    if (g_map[slot].mp) smaltgpu_mapper_free(g_map[slot].mp);
    g_map[slot].mp = NULL;
    if (smaltgpu_mapper_create_ex(&g_map[slot].mp, g_ixdev[slot % g_ndev], rcap, cap, &opts)) return -1;
    g_map[slot].maxlen = cap; g_map[slot].maxreads = rcap;
    free(g_map[slot].off);
    if (!(g_map[slot].off = malloc(((size_t)rcap + 1) * sizeof(uint64_t)))) return -1;
  }
  if (g_map[slot].basecap < nbases + 1) {
    free(g_map[slot].bases); free(g_map[slot].quals);
    g_map[slot].basecap = 2 * nbases + 1024;
    g_map[slot].bases = malloc(g_map[slot].basecap); g_map[slot].quals = malloc(g_map[slot].basecap);
    if (!g_map[slot].bases || !g_map[slot].quals) return -1;
  }
  return slot;
}

int rmapSingle(ErrMsg *errmsgp, RMap *rmp, SeqFastq *readp, int ktuple_maxhit, uint32_t min_cover, int min_swatscor,
               int min_swatscor_below_max, UCHAR min_basqval, short target_depth, short max_depth, RMAPFLG_t rmapflg,
               const ScoreMatrix *scormtxp, const ResultFilter *rsfp, const HashTable *htp, const SeqSet *ssp,
               const SeqCodec *codecp)
{
  int errcode, slot;
  uint32_t rlen, qlen = 0, i;
  char cod, qcod;
  const char *seqp, *qualp;
  short mismatchscor, gapinitscor, gapextscor, matchscor;
  smaltgpu_params par;
  smaltgpu_batch_out out;
  uint64_t off[2];
  static const char ALPHA[8] = {'A', 'C', 'G', 'T', 'N', 'N', 'N', 'N'};     /* codes 4..7 are non-standard: all encode as N on the way in */

  if (rmapflg & (RMAPFLG_SPLIT | RMAPFLG_CMPLXW))     /* not on the GPU path: the reference's own implementation */
    return rmapSingle_reference_cpu(errmsgp, rmp, readp, ktuple_maxhit, min_cover, min_swatscor, min_swatscor_below_max,
                                    min_basqval, target_depth, max_depth, rmapflg, scormtxp, rsfp, htp, ssp, codecp);
  rmapBlank(rmp);
  if ((errcode = makeRMAPPROFfromRead(rmp->prp, readp, scormtxp, codecp)))
    ERRMSGNO(errmsgp, errcode);
  seqp = seqFastqGetConstSequence(readp, &rlen, &cod);
  if (rlen >= hashTableGetKtupLen(htp, NULL)) {
    if ((slot = gpuMapperFor(rmp, rlen)) < 0) ERRMSGNO(errmsgp, ERRCODE_FAILURE);
    for (i = 0; i < rlen; i++)
      g_map[slot].bases[i] = (cod == SEQCOD_ASCII) ? seqp[i] : ALPHA[seqp[i] & SEQCOD_ALPHA_MASK];
    qualp = seqFastqGetConstQualityFactors(readp, &qlen, &qcod);
    if (qualp && qlen == rlen) memcpy(g_map[slot].quals, qualp, rlen);
    matchscor = scoreProfileGetAvgPenalties(&mismatchscor, &gapinitscor, &gapextscor, rmp->prp->scorprofp);
    smaltgpu_params_default(&par, g_ix);
    par.ktuple_maxhit = ktuple_maxhit; par.min_cover = min_cover; par.min_swatscor = min_swatscor;
    par.min_swatscor_below_max = min_swatscor_below_max; par.min_basqval = min_basqval;
    par.target_depth = target_depth; par.max_depth = max_depth; par.rmapflg = rmapflg;
    par.match = matchscor; par.mismatch = mismatchscor; par.gap_init = gapinitscor; par.gap_ext = gapextscor;
    par.min_cover_frac = 0.0;
    off[0] = 0; off[1] = rlen;
    if (smaltgpu_map_batch(g_map[slot].mp, (const uint8_t *)g_map[slot].bases,
                           (qualp && qlen == rlen) ? (const uint8_t *)g_map[slot].quals : NULL, off, 1, &par, &out))
      ERRMSGNO(errmsgp, ERRCODE_FAILURE);
    if (out.stat[0].errcode) ERRMSGNO(errmsgp, out.stat[0].errcode == SMALTGPU_ESCORE ? ERRCODE_SWATSCOR : ERRCODE_FAILURE);
    if ((errcode = resultSetInjectRaw(rmp->rsrp, (unsigned)(out.res_off[1] - out.res_off[0]), out.res + out.res_off[0], out.diffstr,
                                      out.stat[0].swatscor_max, out.stat[0].swatscor_2ndmax)))
      ERRMSGNO(errmsgp, errcode);
    resultSetAlignmentStats(rmp->rsrp, out.stat[0].n_ali_done, out.stat[0].n_ali_tot, max_depth,
                            out.stat[0].n_hits_used, out.stat[0].n_hits_tot);                 /* rmap.c:1337 */
    if ((errcode = resultSetSortAndAssignSequence(rmp->rsrp, rmp->bfp->sqbfp, 0, readp, rmp->prp->scorprofp,
                                                  rmp->prp->scorprofRCp, ssp, codecp)))        /* rmap.c:1418 */
      ERRMSGNO(errmsgp, errcode);
    if ((errcode = resultSetFilterResults(rmp->rsrp, rsfp, readp)))                            /* rmap.c:1734 */
      ERRMSGNO(errmsgp, errcode);
  }
  return ERRCODE_SUCCESS;
}


/* ---- batched form: used by integration/smalt_batch.c, which replaces the per-read loop of
 *      processArgBlock (smalt.c:1221) by   rmapGpuBatch(block)  +  rmapGpuFinish(read i) per read. ---- */
static const char GPU_ALPHA[8] = {'A', 'C', 'G', 'T', 'N', 'N', 'N', 'N'};

/* phase 1 + 2: all reads of a block (encoded, SEQCOD_MANGLED) through the GPU path; results stay with the mapper */
int rmapGpuBatch(ErrMsg *errmsgp, RMap *rmp, SeqFastq *const *reads, int n, int ktuple_maxhit, double tupcovmin,
                 int min_swatscor, int min_swatscor_below_max, UCHAR min_basqval, short target_depth, short max_depth,
                 RMAPFLG_t rmapflg, const ScoreMatrix *scormtxp, const SeqCodec *codecp)
{
  int i, slot, errcode, has_qual = 1;
  uint32_t maxlen = 1, rlen, qlen, j;
  size_t tot = 0;
  char cod, qcod;
  short mismatchscor, gapinitscor, gapextscor, matchscor;
  smaltgpu_params par;
  for (i = 0; i < n; i++) { (void)seqFastqGetConstSequence(reads[i], &rlen, &cod); tot += rlen; if (rlen > maxlen) maxlen = rlen; }
  pthread_once(&g_once, gpuReadConfig);
  const int split = (rmapflg & RMAPFLG_SPLIT) != 0;          /* smalt map -p: smaltgpu_map_split runs both calls of every read on this worker's mapper */
  const int combine = g_combine && !split;                   /* default: blocks of all worker threads form one GPU batch */
  if ((slot = gpuMapperForBatch(rmp, maxlen, (uint32_t)n, tot, !combine)) < 0) { fprintf(stderr, "smaltgpu: %s\n", smaltgpu_last_error()); ERRMSGNO(errmsgp, ERRCODE_FAILURE); }
  double t0 = tmNow(), t1;
  for (i = 0, tot = 0; i < n; i++) {
    const char *seqp = seqFastqGetConstSequence(reads[i], &rlen, &cod);
    const char *qualp = seqFastqGetConstQualityFactors(reads[i], &qlen, &qcod);
    g_map[slot].off[i] = tot;
    for (j = 0; j < rlen; j++) g_map[slot].bases[tot + j] = (cod == SEQCOD_ASCII) ? seqp[j] : GPU_ALPHA[seqp[j] & SEQCOD_ALPHA_MASK];
    if (qualp && qlen == rlen) memcpy(g_map[slot].quals + tot, qualp, rlen); else has_qual = 0;
    tot += rlen;
  }
  g_map[slot].off[n] = tot;
  /* penalties as the reference derives them for a read (rmap.c:1259): the profile of the first read */
  if ((errcode = makeRMAPPROFfromRead(rmp->prp, reads[0], scormtxp, codecp))) ERRMSGNO(errmsgp, errcode);
  matchscor = scoreProfileGetAvgPenalties(&mismatchscor, &gapinitscor, &gapextscor, rmp->prp->scorprofp);
  smaltgpu_params_default(&par, g_ix);
  par.ktuple_maxhit = ktuple_maxhit; par.min_swatscor = min_swatscor; par.min_swatscor_below_max = min_swatscor_below_max;
  par.min_basqval = min_basqval; par.target_depth = target_depth; par.max_depth = max_depth; par.rmapflg = rmapflg & ~(RMAPFLG_t)RMAPFLG_SPLIT;
  par.match = matchscor; par.mismatch = mismatchscor; par.gap_init = gapinitscor; par.gap_ext = gapextscor;
  if (tupcovmin < 1.01) { par.min_cover = 0; par.min_cover_frac = tupcovmin; }      /* smalt.c:1113-1126 */
  else { par.min_cover = (uint32_t)tupcovmin; par.min_cover_frac = 0.0; }
  g_map[slot].nbatch = 0;
  g_map[slot].use_comb = combine;
  t1 = tmNow(); g_tm[slot][TM_STAGE] += t1 - t0; t0 = t1;
  if (split) {
    /* rmapSingle with RMAPFLG_SPLIT (rmap.c:1716-1728): the read's own call, mapSecondary's call (rmap.c:1435) and the passes of
     * resultSetSortAndAssignSequence behind each, for the whole block; rmapGpuFinish puts the finished sets into the RMap */
    if (!g_map[slot].post && !(g_map[slot].post = smaltgpu_post_create())) ERRMSGNO(errmsgp, ERRCODE_NOMEM);
    g_map[slot].have_post = 0;
    const int rv = smaltgpu_map_split(g_map[slot].mp, g_map[slot].post, (const uint8_t *)g_map[slot].bases, has_qual ? (const uint8_t *)g_map[slot].quals : NULL,
                                      g_map[slot].off, (uint32_t)n, &par, g_ixdev[slot % g_ndev], 1, &g_map[slot].pout, NULL);
    if (rv) {
      fprintf(stderr, "smaltgpu: %s\n", smaltgpu_last_error());
      ERRMSGNO(errmsgp, rv == SMALTGPU_ESCORE ? ERRCODE_SWATSCOR : ERRCODE_FAILURE);
    }
    g_map[slot].have_post = 2;
    g_map[slot].nbatch = n;
    g_tm[slot][TM_GPU] += tmNow() - t0;
    return ERRCODE_SUCCESS;
  }
  if (combine) {
    char emsg[256] = "";
    const int crv = gpuCombineSubmit(g_ndev, (const smaltgpu_index *const *)g_ixdev, g_map[slot].bases, has_qual ? g_map[slot].quals : NULL, g_map[slot].off,
                                     (uint32_t)n, &par, &g_map[slot].comb, emsg, sizeof(emsg));
    if (crv) {
      gpuParkIfClosing(crv);
      gpuFailOrderly(emsg);
      ERRMSGNO(errmsgp, ERRCODE_FAILURE);
    }
    g_map[slot].out.nreads = (uint32_t)n; g_map[slot].out.res_off = g_map[slot].comb.res_off; g_map[slot].out.res = g_map[slot].comb.res;
    g_map[slot].out.diffstr = g_map[slot].comb.dstr; g_map[slot].out.stat = g_map[slot].comb.stat;
  } else {
    const int rv = smaltgpu_map_batch(g_map[slot].mp, (const uint8_t *)g_map[slot].bases, has_qual ? (const uint8_t *)g_map[slot].quals : NULL,
                                      g_map[slot].off, (uint32_t)n, &par, &g_map[slot].out);
    /* a read that fails on its own leaves its code in stat[].errcode; rmapGpuFinish reports it for that read */
    if (rv && !(SMALTGPU_IS_READ_ERROR(rv) && g_map[slot].out.nreads == (uint32_t)n)) {
      fprintf(stderr, "smaltgpu: %s\n", smaltgpu_last_error());
      ERRMSGNO(errmsgp, ERRCODE_FAILURE);
    }
  }
  g_map[slot].nbatch = n;
  t1 = tmNow(); g_tm[slot][TM_GPU] += t1 - t0; t0 = t1;
  /* result post-processing of the whole block by the library (smaltgpu_postprocess: what resultSetSortAndAssignSequence
   * would compute read by read, results.c:2022); SMALTGPU_REF_POST keeps the reference's own routine instead */
  g_map[slot].have_post = 0;
  if (g_libpost) {
    if (!g_map[slot].post) {
      smaltgpu_index_desc ds;
      int64_t s_;
      if (smaltgpu_index_info(g_ix, &ds)) ERRMSGNO(errmsgp, ERRCODE_FAILURE);
      if (!(g_map[slot].post = smaltgpu_post_create()) || !(g_map[slot].sop = malloc(((size_t) ds.nseq + 1) * sizeof(uint64_t)))) ERRMSGNO(errmsgp, ERRCODE_NOMEM);
      for (s_ = 0; s_ <= ds.nseq; s_++) g_map[slot].sop[s_] = ds.sop[s_];       /* = seqSetGetOffsets of the program's SeqSet: same .sma file */
      g_map[slot].nseq = ds.nseq;
    }
    /* concatenated mode (rmapflg without SEQBYSEQ): alignments across sequence junctions are cut by the library too, which
     * needs the packed reference on the host */
    const uint32_t *packed = (rmapflg & RMAPFLG_SEQBYSEQ) ? NULL : smaltgpu_index_packed_host(g_ix);
    if (smaltgpu_postprocess(g_map[slot].post, g_map[slot].sop, g_map[slot].nseq, &g_map[slot].out, (const uint8_t *)g_map[slot].bases,
                             has_qual ? (const uint8_t *)g_map[slot].quals : NULL, g_map[slot].off, packed, &par, 1, &g_map[slot].pout) == SMALTGPU_OK)
      g_map[slot].have_post = 1;
  }
  g_tm[slot][TM_POST] += tmNow() - t0;
  return ERRCODE_SUCCESS;
}

/* phase 3: the ResultSet of read i of the last batch, then the reference's own post-processing */
int rmapGpuFinish(ErrMsg *errmsgp, RMap *rmp, int i, SeqFastq *readp, short max_depth, const ScoreMatrix *scormtxp,
                  const ResultFilter *rsfp, const HashTable *htp, const SeqSet *ssp, const SeqCodec *codecp)
{
  int errcode, slot;
  uint32_t rlen;
  const smaltgpu_batch_out *o;
  if ((slot = gpuMapperForBatch(rmp, 1, 1, 1, 0)) < 0 || i < 0 || i >= g_map[slot].nbatch) ERRMSGNO(errmsgp, ERRCODE_ASSERT);
  o = &g_map[slot].out;
  rmapBlank(rmp);
  if ((errcode = makeRMAPPROFfromRead(rmp->prp, readp, scormtxp, codecp))) ERRMSGNO(errmsgp, errcode);
  (void)seqFastqGetConstSequence(readp, &rlen, NULL);
  if (rlen < hashTableGetKtupLen(htp, NULL)) return ERRCODE_SUCCESS;                      /* ERRCODE_SHORTSEQ is swallowed (rmap.c:1736) */
  if (g_map[slot].have_post == 2) {          /* split reads: the set as rmapSingle leaves it, whole from the library (blank set: every row is new) */
    const smaltgpu_post_out *po = &g_map[slot].pout;
    const unsigned nrow = (unsigned)(po->res_off[i + 1] - po->res_off[i]);
    if (nrow > 0 && (errcode = resultSetInjectPost(rmp->rsrp, nrow, po->res + po->res_off[i], po->diffstr,
                                                   (unsigned)(po->sort_off[i + 1] - po->sort_off[i]), po->sortr + po->sort_off[i], po->segsrtr + po->sort_off[i],
                                                   (unsigned)(po->seg_off[i + 1] - po->seg_off[i]), po->segnor + po->seg_off[i], po->qsegno[i], po->setstatus[i])))
      ERRMSGNO(errmsgp, errcode);
    if ((errcode = resultSetFilterResults(rmp->rsrp, rsfp, readp))) ERRMSGNO(errmsgp, errcode);
    return ERRCODE_SUCCESS;
  }
  if (o->stat[i].errcode) {                /* the reference's own per-read failure keeps its code (alignment.c:767 -> rmap.c:1417) */
    gpuFailOrderly("a read failed on the device");
    ERRMSGNO(errmsgp, o->stat[i].errcode == SMALTGPU_ESCORE ? ERRCODE_SWATSCOR : ERRCODE_FAILURE);
  }
  double t0 = tmNow(), t1;
  if ((errcode = resultSetInjectRaw(rmp->rsrp, (unsigned)(o->res_off[i + 1] - o->res_off[i]), o->res + o->res_off[i], o->diffstr,
                                    o->stat[i].swatscor_max, o->stat[i].swatscor_2ndmax)))
    ERRMSGNO(errmsgp, errcode);
  resultSetAlignmentStats(rmp->rsrp, o->stat[i].n_ali_done, o->stat[i].n_ali_tot, max_depth, o->stat[i].n_hits_used, o->stat[i].n_hits_tot);
  int tm_kind = TM_INJECT;
  if (o->stat[i].max1scor >= 1) {                       /* mapSingleRead sorts only when the score pass found something (rmap.c:1376) */
    const smaltgpu_post_out *po = &g_map[slot].pout;
    if (g_map[slot].have_post && !po->needs_reference[i]) {      /* N1 from the library */
      if ((errcode = resultSetInjectPost(rmp->rsrp, (unsigned)(po->res_off[i + 1] - po->res_off[i]), po->res + po->res_off[i], po->diffstr,
                                         (unsigned)(po->sort_off[i + 1] - po->sort_off[i]), po->sortr + po->sort_off[i], po->segsrtr + po->sort_off[i],
                                         (unsigned)(po->seg_off[i + 1] - po->seg_off[i]), po->segnor + po->seg_off[i], po->qsegno[i], po->setstatus[i])))
        ERRMSGNO(errmsgp, errcode);
    } else {
      tm_kind = TM_REFPOST;
      if ((errcode = resultSetSortAndAssignSequence(rmp->rsrp, rmp->bfp->sqbfp, 0, readp, rmp->prp->scorprofp, rmp->prp->scorprofRCp, ssp, codecp)))
        ERRMSGNO(errmsgp, errcode);                      /* an alignment across a sequence junction (splitMultiSpan), or SMALTGPU_REF_POST */
    }
  }
  t1 = tmNow(); g_tm[slot][tm_kind] += t1 - t0; t0 = t1;
  if ((errcode = resultSetFilterResults(rmp->rsrp, rsfp, readp))) ERRMSGNO(errmsgp, errcode);
  g_tm[slot][TM_FILTER] += tmNow() - t0;
  return ERRCODE_SUCCESS;
}


/* ================================================================================================================
 * Paired reads: rmapPair (rmap.c:1744-2112) for a whole block of pairs.
 *
 * rmapPair makes two to four mapSingleRead calls per pair and decides between them on mapping qualities, proper pairs and
 * scores -- all of it results.c / resultpairs.c, which stay the reference's own code here.  The calls themselves go to the
 * GPU as ROUNDS over the block (smaltgpu_map_batch_ctx):
 *   round A  the mate with fewer k-mer hits (smaltgpu_hit_totals; rmap.c:1866-1905), unrestricted
 *   round B  the other mate, seeding restricted to the intervals round A implies (rmap.c:1933-1954)
 *   round C  pairs without a proper pair / low mapping quality / weak restricted score: the other mate unrestricted (:1965-1989)
 *   round D  of those, pairs whose second mate came out better: the first mate restricted, over the on-the-fly k=5 index (:1994-2039)
 * Every pair owns two ResultSets for the duration of the block; rmapGpuPairFinish makes them the RMap's for the final
 * pairing, filters and the report.  The statements below follow rmapPair line by line; only the mapSingleRead calls are
 * deferred to the next round.
 * ================================================================================================================ */
typedef struct GpuPair_ {
  ResultSet *rs[2];            /* [0] read, [1] mate */
  SeqFastq *sq[2];
  unsigned char first;         /* which of the two is mapped first (rare_mate) */
  unsigned char skip;          /* both shorter than the word length: nothing to do (rmap.c:1834) */
  unsigned char lone, lone_w;  /* exactly one mate (lone_w) is long enough: it is mapped alone */
  int fpp_err;                 /* return code of resultSetFindProperPairs (rmap.c:1956-1961, :2050) */
  unsigned char need_c, need_d;
  unsigned char on_cpu;        /* more search intervals than a restricted device call takes: the reference's own rmapPair maps this pair */
  RSLTPAIRFLG_t pairflg;
  int mapq1, swscor1, swscor2_restricted, n_proper, minsw_d;
} GpuPair;

/* reads into the worker's staging buffers; returns has_qual */
static int gpuStage(int slot, SeqFastq *const *sq, int ns)
{
  int i, has_qual = 1;
  size_t tot = 0;
  uint32_t rlen, qlen, j;
  char cod, qcod;
  for (i = 0; i < ns; i++) {
    const char *seqp = seqFastqGetConstSequence(sq[i], &rlen, &cod);
    const char *qualp = seqFastqGetConstQualityFactors(sq[i], &qlen, &qcod);
    g_map[slot].off[i] = tot;
    for (j = 0; j < rlen; j++) g_map[slot].bases[tot + j] = (cod == SEQCOD_ASCII) ? seqp[j] : GPU_ALPHA[seqp[j] & SEQCOD_ALPHA_MASK];
    if (qualp && qlen == rlen) memcpy(g_map[slot].quals + tot, qualp, rlen); else has_qual = 0;
    tot += rlen;
  }
  g_map[slot].off[ns] = tot;
  return has_qual;
}

/* One round: reads sq[0..ns) through the library with the round's per-read context.  By default the rounds of all worker
 * threads meet in the shared queue (gpu_combine.c: requests of the same kind form one batch on a shared mapper);
 * SMALTGPU_NO_COMBINE gives the worker its own mapper.  kind GPUCOMB_TOTALS fills tot_out instead of out. */
static int gpuPairRoundRange(ErrMsg *errmsgp, int slot, SeqFastq *const *sq, int ns, const smaltgpu_params *par, int kind,
                             const uint64_t *iv_off, const smaltgpu_interval *iv, const int32_t *minsw, const int32_t *prevmax,
                             const uint32_t *seedrange, uint32_t *tot_out, smaltgpu_batch_out *out);
static int gpuPairRound(ErrMsg *errmsgp, int slot, SeqFastq *const *sq, int ns, const smaltgpu_params *par, int kind,
                        const uint64_t *iv_off, const smaltgpu_interval *iv, const int32_t *minsw, const int32_t *prevmax,
                        uint32_t *tot_out, smaltgpu_batch_out *out)
{
  return gpuPairRoundRange(errmsgp, slot, sq, ns, par, kind, iv_off, iv, minsw, prevmax, NULL, tot_out, out);
}
/* seedrange (kind GPUCOMB_SPLIT): per read the stretch its k-mer words come from */
static int gpuPairRoundRange(ErrMsg *errmsgp, int slot, SeqFastq *const *sq, int ns, const smaltgpu_params *par, int kind,
                             const uint64_t *iv_off, const smaltgpu_interval *iv, const int32_t *minsw, const int32_t *prevmax,
                             const uint32_t *seedrange, uint32_t *tot_out, smaltgpu_batch_out *out)
{
  int i, has_qual;
  size_t tot = 0;
  uint32_t rlen, maxlen = 1;
  for (i = 0; i < ns; i++) { (void)seqFastqGetConstSequence(sq[i], &rlen, NULL); tot += rlen; if (rlen > maxlen) maxlen = rlen; }
  if (gpuMapperForBatch(g_map[slot].rmp, maxlen, (uint32_t)ns, tot, !g_combine) != slot) ERRMSGNO(errmsgp, ERRCODE_FAILURE);
  has_qual = gpuStage(slot, sq, ns);
  if (g_combine) {
    GpuCombCtx cc;
    char emsg[256] = "";
    memset(&cc, 0, sizeof(cc));
    cc.kind = kind; cc.iv_off = iv_off; cc.iv = iv; cc.minsw = minsw; cc.prevmax = prevmax; cc.seedrange = seedrange; cc.tot_out = tot_out;
    const int crv = gpuCombineSubmitCtx(g_ndev, (const smaltgpu_index *const *)g_ixdev, g_map[slot].bases, has_qual ? g_map[slot].quals : NULL, g_map[slot].off,
                                        (uint32_t)ns, par, kind == GPUCOMB_PLAIN ? NULL : &cc, &g_map[slot].comb, emsg, sizeof(emsg));
    if (crv) {
      gpuParkIfClosing(crv);
      gpuFailOrderly(emsg);
      ERRMSGNO(errmsgp, ERRCODE_FAILURE);
    }
    if (out) { out->nreads = (uint32_t)ns; out->res_off = g_map[slot].comb.res_off; out->res = g_map[slot].comb.res; out->diffstr = g_map[slot].comb.dstr; out->stat = g_map[slot].comb.stat; }
  } else {
    int rv;
    if (kind == GPUCOMB_TOTALS)
      rv = smaltgpu_hit_totals(g_map[slot].mp, (const uint8_t *)g_map[slot].bases, has_qual ? (const uint8_t *)g_map[slot].quals : NULL, g_map[slot].off, (uint32_t)ns, par, tot_out);
    else {
      smaltgpu_callctx ctx;
      memset(&ctx, 0, sizeof(ctx));
      ctx.iv_off = iv_off; ctx.iv = iv; ctx.min_swatscor = minsw; ctx.prev_max = prevmax; ctx.fine_index = kind == GPUCOMB_FINE;
      ctx.raw_alignments = kind == GPUCOMB_APPEND || kind == GPUCOMB_FINE || kind == GPUCOMB_SPLIT;
      ctx.seed_range = seedrange;
      rv = smaltgpu_map_batch_ctx(g_map[slot].mp, (const uint8_t *)g_map[slot].bases, has_qual ? (const uint8_t *)g_map[slot].quals : NULL,
                                  g_map[slot].off, (uint32_t)ns, par, &ctx, out);
      if (SMALTGPU_IS_READ_ERROR(rv) && out->nreads == (uint32_t)ns) rv = 0;      /* gpuPairTake reports the read */
    }
    if (rv) { fprintf(stderr, "smaltgpu: %s\n", smaltgpu_last_error()); ERRMSGNO(errmsgp, ERRCODE_FAILURE); }
  }
  return ERRCODE_SUCCESS;
}

/* what mapSingleRead does with the alignments (rmap.c:1337, :903-911, :1418): stats, results, sort + mapping qualities */
static int gpuPairTake(ErrMsg *errmsgp, RMap *rmp, GpuPair *pp, int w, const smaltgpu_batch_out *o, int i, short max_depth,
                       const ScoreMatrix *scormtxp, const SeqSet *ssp, const SeqCodec *codecp)
{
  int errcode;
  ResultSet *rs = pp->rs[w];
  if (o->stat[i].errcode) { char m[96]; snprintf(m, sizeof(m), "read failed on the device (code %d)", o->stat[i].errcode); gpuFailOrderly(m); ERRMSGNO(errmsgp, o->stat[i].errcode == SMALTGPU_ESCORE ? ERRCODE_SWATSCOR : ERRCODE_FAILURE); }
  resultSetAlignmentStats(rs, o->stat[i].n_ali_done, o->stat[i].n_ali_tot, max_depth, o->stat[i].n_hits_used, o->stat[i].n_hits_tot);
  if ((errcode = resultSetAppendRaw(rs, (unsigned)(o->res_off[i + 1] - o->res_off[i]), o->res + o->res_off[i], o->diffstr,
                                    o->stat[i].swatscor_max, o->stat[i].swatscor_2ndmax)))
    ERRMSGNO(errmsgp, errcode);
  if (o->stat[i].max1scor < 1) return ERRCODE_SUCCESS;            /* mapSingleRead returned at rmap.c:1376: no re-sort */
  if ((errcode = makeRMAPPROFfromRead(rmp->prp, pp->sq[w], scormtxp, codecp))) ERRMSGNO(errmsgp, errcode);
  if ((errcode = resultSetSortAndAssignSequence(rs, rmp->bfp->sqbfp, 0, pp->sq[w], rmp->prp->scorprofp, rmp->prp->scorprofRCp, ssp, codecp)))
    ERRMSGNO(errmsgp, errcode);
  return ERRCODE_SUCCESS;
}

int rmapGpuPairBatch(ErrMsg *errmsgp, RMap *rmp, SeqFastq *const *reads, SeqFastq *const *mates, int n, int d_min, int d_max,
                     RSLTPAIRLIB_t pairlibcode, int ktuple_maxhit, double tupcovmin, int min_swatscor, UCHAR min_basqval,
                     short target_depth, short max_depth, RMAPFLG_t rmapflg, const ScoreMatrix *scormtxp, const HashTable *htp,
                     const SeqSet *ssp, const SeqCodec *codecp)
{
  int i, slot, ns, errcode = ERRCODE_SUCCESS;
  uint32_t rlen, *tot = NULL;
  UCHAR nskip = 0;
  const UCHAR ktup = hashTableGetKtupLen(htp, &nskip);
  short mismatchscor, gapinitscor, gapextscor, matchscor;
  smaltgpu_params par;
  smaltgpu_batch_out o;
  GpuPair *pairs;
  int *sel = NULL;
  unsigned char *which = NULL;
  uint64_t *iv_off = NULL;
  smaltgpu_interval *iv = NULL;
  size_t niv = 0, cap_iv = 0;
  int32_t *minsw = NULL, *prevmax = NULL;
  SeqFastq **all = NULL, **rsq = NULL;

  pthread_once(&g_once, gpuReadConfig);
  if ((slot = gpuMapperForBatch(rmp, 64, 1, 64, !g_combine)) < 0) { fprintf(stderr, "smaltgpu: %s\n", smaltgpu_last_error()); ERRMSGNO(errmsgp, ERRCODE_FAILURE); }
  if (g_map[slot].cap_pairs < n) {
    GpuPair *np = realloc(g_map[slot].pairs, (size_t)n * sizeof(GpuPair));
    if (!np) ERRMSGNO(errmsgp, ERRCODE_NOMEM);
    memset(np + g_map[slot].cap_pairs, 0, (size_t)(n - g_map[slot].cap_pairs) * sizeof(GpuPair));
    g_map[slot].pairs = np; g_map[slot].cap_pairs = n;
  }
  pairs = g_map[slot].pairs;
  g_map[slot].npairs = 0;
  sel = malloc((size_t)2 * n * sizeof(int)); which = malloc((size_t)2 * n); tot = malloc((size_t)2 * n * sizeof(uint32_t));
  iv_off = malloc(((size_t)n + 1) * sizeof(uint64_t)); minsw = malloc((size_t)n * sizeof(int32_t)); prevmax = malloc((size_t)2 * n * sizeof(int32_t));
  all = malloc((size_t)2 * n * sizeof(SeqFastq *)); rsq = malloc((size_t)n * sizeof(SeqFastq *));
  if (!sel || !which || !tot || !iv_off || !minsw || !prevmax || !all || !rsq) ERRMSGNO(errmsgp, ERRCODE_NOMEM);

  /* penalties as the reference derives them for a read (rmap.c:1259) */
  if ((errcode = makeRMAPPROFfromRead(rmp->prp, reads[0], scormtxp, codecp))) ERRMSGNO(errmsgp, errcode);
  matchscor = scoreProfileGetAvgPenalties(&mismatchscor, &gapinitscor, &gapextscor, rmp->prp->scorprofp);
  smaltgpu_params_default(&par, g_ix);
  par.ktuple_maxhit = ktuple_maxhit; par.min_swatscor = min_swatscor; par.min_swatscor_below_max = MINSCOR_BELOW_MAX_BEST;
  par.min_basqval = min_basqval; par.target_depth = target_depth; par.max_depth = max_depth;
  par.rmapflg = rmapflg & (SMALTGPU_FLG_BEST | SMALTGPU_FLG_SEQBYSEQ | SMALTGPU_FLG_NOSHRTINFO | SMALTGPU_FLG_SENSITIVE);
  par.match = matchscor; par.mismatch = mismatchscor; par.gap_init = gapinitscor; par.gap_ext = gapextscor;
  if (tupcovmin < 1.01) { par.min_cover = 0; par.min_cover_frac = tupcovmin; }      /* smalt.c:1113-1147: per read and per mate */
  else { par.min_cover = (uint32_t)tupcovmin; par.min_cover_frac = 0.0; }

  g_map[slot].pair_args.ktuple_maxhit = ktuple_maxhit; g_map[slot].pair_args.min_swatscor = min_swatscor; g_map[slot].pair_args.tupcovmin = tupcovmin;
  g_map[slot].pair_args.min_basqval = min_basqval; g_map[slot].pair_args.target_depth = target_depth; g_map[slot].pair_args.max_depth = max_depth;
  g_map[slot].pair_args.rmapflg = rmapflg; g_map[slot].pair_args.scormtxp = scormtxp; g_map[slot].pair_args.htp = htp; g_map[slot].pair_args.ssp = ssp;
  g_map[slot].pair_args.codecp = codecp;

  /* ---- which mate first: k-mer hit totals of all 2n reads (rmap.c:1866-1905) ---- */
  for (i = 0; i < n; i++) {
    GpuPair *pp = pairs + i;
    int w;
    for (w = 0; w < 2; w++) {
      if (!pp->rs[w] && !(pp->rs[w] = resultSetCreate(0, 0))) ERRMSGNO(errmsgp, ERRCODE_NOMEM);
      resultSetBlank(pp->rs[w]);
    }
    pp->sq[0] = reads[i]; pp->sq[1] = mates[i];
    pp->pairflg = RSLTPAIRFLG_PAIRED; pp->first = 0; pp->skip = pp->lone = pp->need_c = pp->need_d = pp->on_cpu = 0;
    pp->mapq1 = pp->swscor1 = pp->swscor2_restricted = pp->n_proper = pp->minsw_d = pp->fpp_err = 0;
    all[2 * i] = reads[i]; all[2 * i + 1] = mates[i];
  }
  /* hit totals: one seeding-only batch over both mates of every pair */
  if ((errcode = gpuPairRound(errmsgp, slot, all, 2 * n, &par, GPUCOMB_TOTALS, NULL, NULL, NULL, NULL, tot, NULL))) return errcode;
  for (i = 0; i < n; i++) {
    GpuPair *pp = pairs + i;
    uint32_t l0, l1;
    (void)seqFastqGetConstSequence(pp->sq[0], &l0, NULL);
    (void)seqFastqGetConstSequence(pp->sq[1], &l1, NULL);
    if (l0 < ktup && l1 < ktup) { pp->skip = 1; continue; }          /* rmap.c:1833-1834 */
    /* One mate shorter than the word length (rmap.c:1836-1864): the other one is mapped unrestricted, and the rounds that
     * follow (nothing for the short mate, an empty restriction, no proper pair -> blank + the same unrestricted call)
     * leave exactly that mapping.  nhit of the short mate is 0, so it counts as the mate mapped first. */
    if (l0 < ktup || l1 < ktup) { pp->lone = 1; pp->lone_w = (l0 < ktup) ? 1 : 0; }
    if (tot[2 * i] > tot[2 * i + 1]) { pp->pairflg |= RSLTPAIRFLG_RAREMATE; pp->first = 1; }
  }

  /* ---- round A: the first mate, unrestricted (rmap.c:1907-1918) ---- */
  for (i = 0, ns = 0; i < n; i++) if (!pairs[i].skip && !pairs[i].lone) { sel[ns] = i; which[ns] = pairs[i].first; ns++; }
  if (ns) {
    for (i = 0; i < ns; i++) rsq[i] = pairs[sel[i]].sq[which[i]];
    if ((errcode = gpuPairRound(errmsgp, slot, rsq, ns, &par, GPUCOMB_PLAIN, NULL, NULL, NULL, NULL, NULL, &o))) return errcode;
    for (i = 0; i < ns; i++) if ((errcode = gpuPairTake(errmsgp, rmp, pairs + sel[i], which[i], &o, i, max_depth, scormtxp, ssp, codecp))) return errcode;
  }
  /* ---- intervals from the first mate's results (rmap.c:1920-1938), round B: the second mate restricted (:1940-1954) ---- */
  for (i = 0, niv = 0; i < ns; i++) {
    GpuPair *pp = pairs + sel[i];
    const int w = pp->first;
    int v, nv;
    pp->mapq1 = resultSetGetMappingScore(pp->rs[w], &pp->swscor1);
    if ((errcode = setupInterValFromResultSet(rmp->ivr, d_min, d_max, pp->sq[w], pp->sq[!w], htp, ssp, pp->rs[w]))) ERRMSGNO(errmsgp, errcode);
    interValPrune(rmp->ivr);
    nv = interValNum(rmp->ivr);
    /* The interval number has 11 bits in the hit sort key of a restricted device call (2048 intervals -- one per alignment of
     * the first mate up to the maximum depth; the reference has no limit, interval.c:98-121, and split alignments can add some).  A pair beyond that is not an error of the block: it takes part in the round with an empty
     * restriction and rmapGpuPairFinish maps it with the reference's own rmapPair on this worker's thread. */
    if (g_timing && nv > g_most_intervals) { pthread_mutex_lock(&g_lock); if (nv > g_most_intervals) g_most_intervals = nv; pthread_mutex_unlock(&g_lock); }
    if (nv > g_max_intervals) { pp->on_cpu = 1; nv = 0; }
    iv_off[i] = niv;
    if (niv + (size_t)nv > cap_iv) { cap_iv = 2 * (niv + (size_t)nv) + 64; if (!(iv = realloc(iv, cap_iv * sizeof(*iv)))) ERRMSGNO(errmsgp, ERRCODE_NOMEM); }
    for (v = 0; v < nv; v++) {
      SEQLEN_t lo, hi; SEQNUM_t sx;
      interValGet(&lo, &hi, &sx, NULL, v, rmp->ivr);
      iv[niv].sidx = (int32_t)sx; iv[niv].lo = lo; iv[niv].hi = hi; niv++;
    }
    which[i] = (unsigned char)!w;
  }
  iv_off[ns] = niv;
  if (ns) {
    if (!iv && !(iv = malloc(sizeof(*iv)))) ERRMSGNO(errmsgp, ERRCODE_NOMEM);
    for (i = 0; i < ns; i++) rsq[i] = pairs[sel[i]].sq[which[i]];
    if ((errcode = gpuPairRound(errmsgp, slot, rsq, ns, &par, GPUCOMB_RESTRICTED, iv_off, iv, NULL, NULL, NULL, &o))) return errcode;
    for (i = 0; i < ns; i++) if ((errcode = gpuPairTake(errmsgp, rmp, pairs + sel[i], which[i], &o, i, max_depth, scormtxp, ssp, codecp))) return errcode;
  }
  /* ---- proper pairs so far; who needs the unrestricted round (rmap.c:1956-1969) ---- */
  for (i = 0; i < ns; i++) {
    GpuPair *pp = pairs + sel[i];
    const int w2 = !pp->first;
    if (pp->on_cpu) continue;
    errcode = resultSetFindProperPairs(rmp->pairp, d_min, d_max, MAXNUM_PAIRS_TOTAL, 0, pairlibcode, pp->rs[0], pp->rs[1]);
    if ((errcode) && errcode != ERRCODE_PAIRNUM) ERRMSGNO(errmsgp, errcode);
    pp->fpp_err = errcode;
    resultSetGetMappingScore(pp->rs[w2], &pp->swscor2_restricted);
    resultSetGetNumberOfPairs(&pp->n_proper, rmp->pairp);
    if ((rmapflg & RMAPFLG_ALLPAIR) || pp->n_proper < 1 || pp->mapq1 < MAPSCORE_UNIQUE_MAPPED_1ST ||
        !scorIsAboveFractMax(pp->swscor2_restricted, pp->swscor1, MINFRACT_MAXSCOR_2ND, pp->sq[w2], pp->sq[!w2])) {
      pp->need_c = 1;
      if (pp->n_proper < 1) resultSetBlank(pp->rs[w2]);
    } else {
      pp->pairflg = (RSLTPAIRFLG_t)(pp->pairflg | ((pp->first == 0) ? RSLTPAIRFLG_RESTRICT_2nd : RSLTPAIRFLG_RESTRICT_1st));
      if (errcode) ERRMSGNO(errmsgp, errcode);                   /* rmap.c:2050 with the code of resultSetFindProperPairs */
    }
  }
  /* ---- round C: the second mate unrestricted (rmap.c:1976-1989); lone mates join here ---- */
  for (i = 0, ns = 0; i < n; i++) {
    GpuPair *pp = pairs + i;
    if (pp->skip || !(pp->need_c || pp->lone)) continue;
    sel[ns] = i; which[ns] = pp->lone ? pp->lone_w : (unsigned char)!pp->first;
    prevmax[2 * ns] = resultSetGetMaxSwat(pp->rs[which[ns]], &prevmax[2 * ns + 1]);
    ns++;
  }
  if (ns) {
    for (i = 0; i < ns; i++) rsq[i] = pairs[sel[i]].sq[which[i]];
    if ((errcode = gpuPairRound(errmsgp, slot, rsq, ns, &par, GPUCOMB_APPEND, NULL, NULL, NULL, prevmax, NULL, &o))) return errcode;
    for (i = 0; i < ns; i++) if ((errcode = gpuPairTake(errmsgp, rmp, pairs + sel[i], which[i], &o, i, max_depth, scormtxp, ssp, codecp))) return errcode;
  }
  /* ---- who needs the first mate again, restricted by the second one's results (rmap.c:1991-2008) ---- */
  {
    int nd = 0;
    for (i = 0, niv = 0; i < ns; i++) {
      GpuPair *pp = pairs + sel[i];
      const int w2 = which[i], w1 = !w2;
      int mapq2, swscor2, v, nv;
      if (pp->lone) continue;                                     /* the first mate is shorter than the word length (rmap.c:2012) */
      mapq2 = resultSetGetMappingScore(pp->rs[w2], &swscor2);
      if (!(mapq2 > MAPSCORE_UNIQUE_MAPPED_1ST || swscor2 > pp->swscor2_restricted || swscor2 > pp->swscor1)) {
        if (pp->fpp_err) ERRMSGNO(errmsgp, pp->fpp_err);          /* rmap.c:2050: the code of resultSetFindProperPairs is still standing */
        continue;
      }
      resultSetGetScorStats(pp->rs[w1], NULL, NULL, &pp->minsw_d, NULL);
      if ((errcode = setupInterValFromResultSet(rmp->ivr, d_min, d_max, pp->sq[w2], pp->sq[w1], htp, ssp, pp->rs[w2]))) ERRMSGNO(errmsgp, errcode);
      interValPrune(rmp->ivr);
      (void)seqFastqGetConstSequence(pp->sq[w1], &rlen, NULL);
      if (ktup > rlen) continue;
      nv = interValNum(rmp->ivr);
      if (nv > g_max_intervals) { pp->on_cpu = 1; continue; }      /* as in round B: the reference's rmapPair takes the pair */
      iv_off[nd] = niv;
      if (niv + (size_t)nv > cap_iv) { cap_iv = 2 * (niv + (size_t)nv) + 64; if (!(iv = realloc(iv, cap_iv * sizeof(*iv)))) ERRMSGNO(errmsgp, ERRCODE_NOMEM); }
      for (v = 0; v < nv; v++) {
        SEQLEN_t lo, hi; SEQNUM_t sx;
        interValGet(&lo, &hi, &sx, NULL, v, rmp->ivr);
        iv[niv].sidx = (int32_t)sx; iv[niv].lo = lo; iv[niv].hi = hi; niv++;
      }
      pp->need_d = 1;
      sel[nd] = sel[i]; which[nd] = (unsigned char)w1;
      minsw[nd] = pp->minsw_d;
      prevmax[2 * nd] = resultSetGetMaxSwat(pp->rs[w1], &prevmax[2 * nd + 1]);
      nd++;
    }
    iv_off[nd] = niv;
    /* ---- round D: over the on-the-fly index of the intervals (rmap.c:2010-2039; setupFineHashTable cannot run out of
     *      positions here: the windows of one pair hold far fewer than FINEHASH_MAXKTUPPOS words) ---- */
    if (nd) {
      for (i = 0; i < nd; i++) rsq[i] = pairs[sel[i]].sq[which[i]];
      if ((errcode = gpuPairRound(errmsgp, slot, rsq, nd, &par, GPUCOMB_FINE, iv_off, iv, minsw, prevmax, NULL, &o))) return errcode;
      for (i = 0; i < nd; i++) if ((errcode = gpuPairTake(errmsgp, rmp, pairs + sel[i], which[i], &o, i, max_depth, scormtxp, ssp, codecp))) return errcode;
    }
  }
  /* ---- split reads (rmap.c:2073-2097): mapSecondary for the read, then for the mate, of every pair -- k-mer words from the stretch
   *      the best alignment of the first read segment leaves uncovered (rmap.c:1459-1481), appended to the mate's set ---- */
  if (rmapflg & RMAPFLG_SPLIT) {
    uint32_t *stretch = malloc((size_t)2 * n * sizeof(uint32_t) + 8);
    int w;
    if (!stretch) ERRMSGNO(errmsgp, ERRCODE_NOMEM);
    for (w = 0; w < 2; w++) {
      for (i = 0, ns = 0; i < n; i++) {
        GpuPair *pp = pairs + i;
        const Result *rp;
        SEQLEN_t qs, qe, qlen;
        if (pp->skip || pp->on_cpu) continue;
        if (resultSetGetResultInSegment(&rp, 0, 0, pp->rs[w]) != ERRCODE_SUCCESS) continue;      /* nothing aligned: mapSecondary keeps silent */
        if (resultGetData(&qs, &qe, NULL, NULL, NULL, NULL, NULL, rp)) continue;
        (void)seqFastqGetConstSequence(pp->sq[w], &qlen, NULL);
        if (qe > qlen || qs > qe) { free(stretch); ERRMSGNO(errmsgp, ERRCODE_ASSERT); }
        if (qs + qe > qlen) { qe = (qs > 1) ? qs - 2 : 0; qs = 0; } else { qs = qe; qe = qlen - 1; }
        if (qs + ktup + nskip > qe + 1 || qlen < ktup) continue;
        sel[ns] = i; which[ns] = (unsigned char)w;
        stretch[2 * ns] = qs; stretch[2 * ns + 1] = qe;
        prevmax[2 * ns] = resultSetGetMaxSwat(pp->rs[w], &prevmax[2 * ns + 1]);
        ns++;
      }
      if (!ns) continue;
      for (i = 0; i < ns; i++) rsq[i] = pairs[sel[i]].sq[which[i]];
      if ((errcode = gpuPairRoundRange(errmsgp, slot, rsq, ns, &par, GPUCOMB_SPLIT, NULL, NULL, NULL, prevmax, stretch, NULL, &o))) { free(stretch); return errcode; }
      for (i = 0; i < ns; i++) if ((errcode = gpuPairTake(errmsgp, rmp, pairs + sel[i], which[i], &o, i, max_depth, scormtxp, ssp, codecp))) { free(stretch); return errcode; }
    }
    free(stretch);
  }
  g_map[slot].npairs = n;
  free(sel); free(which); free(tot); free(iv_off); free(iv); free(minsw); free(prevmax); free(all); free(rsq);
  return ERRCODE_SUCCESS;
}

/* the tail of rmapPair for pair i of the block (rmap.c:2080-2110): the pair's ResultSets become the RMap's, so that
 * rmapGetData and the report see them; rmapGpuPairRelease puts the RMap's own sets back */
int rmapGpuPairFinish(ErrMsg *errmsgp, RMap *rmp, int i, RSLTPAIRFLG_t *pairflgp, int d_min, int d_max, RSLTPAIRLIB_t pairlibcode,
                      const ResultFilter *rsfp)
{
  int errcode, slot;
  GpuPair *pp;
  if ((slot = gpuMapperForBatch(rmp, 1, 1, 1, 0)) < 0 || i < 0 || i >= g_map[slot].npairs) ERRMSGNO(errmsgp, ERRCODE_ASSERT);
  pp = g_map[slot].pairs + i;
  rmapBlank(rmp);
  if (pp->on_cpu) {                 /* the whole pair by the reference's own code, into the RMap's own sets (nothing to release) */
    uint32_t len[2], cov[2];
    int w;
    pthread_mutex_lock(&g_lock); g_pairs_on_cpu++; pthread_mutex_unlock(&g_lock);
    for (w = 0; w < 2; w++) {       /* cover thresholds per mate as processMapArgs derives them (smalt.c:1115-1147) */
      (void)seqFastqGetConstSequence(pp->sq[w], &len[w], NULL);
      if (g_map[slot].pair_args.tupcovmin < 1.01) { cov[w] = (uint32_t)(g_map[slot].pair_args.tupcovmin * len[w]); if (cov[w] > len[w]) cov[w] = len[w]; }
      else cov[w] = (uint32_t)g_map[slot].pair_args.tupcovmin;
    }
    return rmapPair(errmsgp, rmp, pp->sq[0], pp->sq[1], pairflgp, d_min, d_max, pairlibcode, g_map[slot].pair_args.ktuple_maxhit, cov[0], cov[1],
                    g_map[slot].pair_args.min_swatscor, g_map[slot].pair_args.min_basqval, g_map[slot].pair_args.target_depth, g_map[slot].pair_args.max_depth,
                    g_map[slot].pair_args.rmapflg, g_map[slot].pair_args.scormtxp, rsfp, g_map[slot].pair_args.htp, g_map[slot].pair_args.ssp,
                    g_map[slot].pair_args.codecp);
  }
  g_map[slot].save_rsr = rmp->rsrp; g_map[slot].save_rsm = rmp->rsmp;
  rmp->rsrp = pp->rs[0]; rmp->rsmp = pp->rs[1];
  *pairflgp = pp->pairflg;
  if (pp->skip) return ERRCODE_SUCCESS;
  if ((errcode = resultSetFindPairs(rmp->pairp, *pairflgp, pairlibcode, d_min, d_max, rmp->rsrp, rmp->rsmp))) ERRMSGNO(errmsgp, errcode);
  if ((errcode = resultSetFilterResults(rmp->rsrp, rsfp, pp->sq[0]))) ERRMSGNO(errmsgp, errcode);
  if ((errcode = resultSetFilterResults(rmp->rsmp, rsfp, pp->sq[1]))) ERRMSGNO(errmsgp, errcode);
  return ERRCODE_SUCCESS;
}

void rmapGpuPairRelease(RMap *rmp)
{
  const int slot = gpuMapperForBatch(rmp, 1, 1, 1, 0);
  if (slot >= 0 && g_map[slot].save_rsr) { rmp->rsrp = g_map[slot].save_rsr; rmp->rsmp = g_map[slot].save_rsm; g_map[slot].save_rsr = g_map[slot].save_rsm = NULL; }
}
