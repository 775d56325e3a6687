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

enum { GPU_MAXMAPPERS = 256 };
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
enum { GPU_MAXDEV = 16 };
static smaltgpu_index *g_ixdev[GPU_MAXDEV];      /* one index image per device, shared by the mappers on it */
static int g_ndev = 0, g_nphys = 0;              /* index images (logical devices) and GPUs present */
static int g_combine = 1;                        /* SMALTGPU_NO_COMBINE unset */
static const char *g_prefix = NULL;              /* SMALTGPU_INDEX_PREFIX */
#define g_ix (g_ixdev[0])
static struct {
  const RMap *rmp; smaltgpu_mapper *mp; uint32_t maxlen, maxreads;
  char *bases, *quals; uint64_t *off; size_t basecap;
  smaltgpu_batch_out out; int nbatch;          /* results of the last rmapGpuBatch */
  GpuCombOut comb; int use_comb;               /* ... when they came from a combined batch (gpu_combine.c) */
} g_map[GPU_MAXMAPPERS];
static int g_nmap = 0;

/* the environment is read once, before any mapper exists; worker threads never call setenv */
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void gpuReadConfig(void)
{
  const char *e = getenv("SMALTGPU_NDEV");       /* worker threads are dealt round-robin to the devices (SMALTGPU_NDEV limits them) */
  g_combine = !getenv("SMALTGPU_NO_COMBINE");
  g_prefix = getenv("SMALTGPU_INDEX_PREFIX");
  g_nphys = g_ndev = smaltgpu_device_count();
  if (e && atoi(e) > 0) g_ndev = atoi(e);        /* more than there are GPUs: images share devices (rehearsal of the N-device path on one GPU) */
  if (g_ndev > GPU_MAXDEV) g_ndev = GPU_MAXDEV;
  if (g_nphys < 1) g_ndev = 0;
}

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
    if (out.stat[0].errcode) ERRMSGNO(errmsgp, ERRCODE_FAILURE);
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
  const int combine = g_combine;                             /* default: blocks of all worker threads form one GPU batch */
  if ((slot = gpuMapperForBatch(rmp, maxlen, (uint32_t)n, tot, !combine)) < 0) { fprintf(stderr, "smaltgpu: %s\n", smaltgpu_last_error()); ERRMSGNO(errmsgp, ERRCODE_FAILURE); }
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
  par.min_basqval = min_basqval; par.target_depth = target_depth; par.max_depth = max_depth; par.rmapflg = rmapflg;
  par.match = matchscor; par.mismatch = mismatchscor; par.gap_init = gapinitscor; par.gap_ext = gapextscor;
  if (tupcovmin < 1.01) { par.min_cover = 0; par.min_cover_frac = tupcovmin; }      /* smalt.c:1113-1126 */
  else { par.min_cover = (uint32_t)tupcovmin; par.min_cover_frac = 0.0; }
  g_map[slot].nbatch = 0;
  g_map[slot].use_comb = combine;
  if (combine) {
    char emsg[256] = "";
    if (gpuCombineSubmit(g_ndev, (const smaltgpu_index *const *)g_ixdev, g_map[slot].bases, has_qual ? g_map[slot].quals : NULL, g_map[slot].off,
                         (uint32_t)n, &par, &g_map[slot].comb, emsg, sizeof(emsg))) {
      fprintf(stderr, "smaltgpu: %s\n", emsg);
      ERRMSGNO(errmsgp, ERRCODE_FAILURE);
    }
    g_map[slot].out.nreads = (uint32_t)n; g_map[slot].out.res_off = g_map[slot].comb.res_off; g_map[slot].out.res = g_map[slot].comb.res;
    g_map[slot].out.diffstr = g_map[slot].comb.dstr; g_map[slot].out.stat = g_map[slot].comb.stat;
  } else {
    const int rv = smaltgpu_map_batch(g_map[slot].mp, (const uint8_t *)g_map[slot].bases, has_qual ? (const uint8_t *)g_map[slot].quals : NULL,
                                      g_map[slot].off, (uint32_t)n, &par, &g_map[slot].out);
    /* a read that fails on its own leaves its code in stat[].errcode; rmapGpuFinish reports it for that read */
    if (rv && !((rv == SMALTGPU_ECAP || rv == SMALTGPU_EINTERNAL) && g_map[slot].out.nreads == (uint32_t)n)) {
      fprintf(stderr, "smaltgpu: %s\n", smaltgpu_last_error());
      ERRMSGNO(errmsgp, ERRCODE_FAILURE);
    }
  }
  g_map[slot].nbatch = n;
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
  if (o->stat[i].errcode) ERRMSGNO(errmsgp, ERRCODE_FAILURE);
  if ((errcode = resultSetInjectRaw(rmp->rsrp, (unsigned)(o->res_off[i + 1] - o->res_off[i]), o->res + o->res_off[i], o->diffstr,
                                    o->stat[i].swatscor_max, o->stat[i].swatscor_2ndmax)))
    ERRMSGNO(errmsgp, errcode);
  resultSetAlignmentStats(rmp->rsrp, o->stat[i].n_ali_done, o->stat[i].n_ali_tot, max_depth, o->stat[i].n_hits_used, o->stat[i].n_hits_tot);
  if ((errcode = resultSetSortAndAssignSequence(rmp->rsrp, rmp->bfp->sqbfp, 0, readp, rmp->prp->scorprofp, rmp->prp->scorprofRCp, ssp, codecp)))
    ERRMSGNO(errmsgp, errcode);
  if ((errcode = resultSetFilterResults(rmp->rsrp, rsfp, readp))) ERRMSGNO(errmsgp, errcode);
  return ERRCODE_SUCCESS;
}
