/* smalt_batch.c -- OUR translation unit around the reference's smalt.c (its text is taken from the reference tree at
 * build time by oracle/Makefile, target ref_gpu; nothing is copied).  The one change to the program: the worker task
 * that maps a block of reads (processArgBlock, smalt.c:1221: one rmapSingle per read) is replaced, at the point where
 * smalt.c registers it with the thread pool, by a version that sends the whole block through the GPU path in one batch
 * (integration/rmap_gpu.c: rmapGpuBatch) and then runs the reference's own per-read tail -- post-processing and report
 * (smalt.c:1172-1185) -- read by read.  A block of read PAIRS goes through rmapPair's rounds on the GPU (rmapGpuPairBatch),
 * pairing and report pair by pair.  Complexity weighting (-w) keeps the reference's own worker. */
#include "threads.h"
static int smaltgpu_threadsSetTask(uint8_t task_typ, short n_threads, THREAD_INITF *initf, const void *initargp, THREAD_PROCF *procf,
                                   THREAD_CLEANF *cleanf, THREAD_CHECKF *checkf, THREAD_CMPF *cmpf, size_t argsz);
#define threadsSetTask smaltgpu_threadsSetTask
#include "smalt.c"
#undef threadsSetTask

extern int rmapGpuBatch(ErrMsg *errmsgp, RMap *rmp, SeqFastq *const *reads, int n, int ktuple_maxhit, double tupcovmin,
                        int min_swatscor, int min_swatscor_below_max, unsigned char min_basqval, short target_depth, short max_depth,
                        RMAPFLG_t rmapflg, const ScoreMatrix *scormtxp, const SeqCodec *codecp);
extern int rmapGpuPairBatch(ErrMsg *errmsgp, RMap *rmp, SeqFastq *const *reads, SeqFastq *const *mates, int n, int d_min, int d_max,
                            RSLTPAIRLIB_t pairlibcode, int ktuple_maxhit, double tupcovmin, int min_swatscor, unsigned char min_basqval,
                            short target_depth, short max_depth, RMAPFLG_t rmapflg, const ScoreMatrix *scormtxp, const HashTable *htp,
                            const SeqSet *ssp, const SeqCodec *codecp);
extern int rmapGpuPairFinish(ErrMsg *errmsgp, RMap *rmp, int i, RSLTPAIRFLG_t *pairflgp, int d_min, int d_max, RSLTPAIRLIB_t pairlibcode,
                             const ResultFilter *rsfp);
extern void rmapGpuPairRelease(RMap *rmp);
extern int rmapGpuFinish(ErrMsg *errmsgp, RMap *rmp, int i, SeqFastq *readp, short max_depth, const ScoreMatrix *scormtxp,
                         const ResultFilter *rsfp, const HashTable *htp, const SeqSet *ssp, const SeqCodec *codecp);

static int processArgBlockGpu(ErrMsg *errmsgp,
#ifdef THREADS_DEBUG
                              uint64_t *readno,
#endif
                              void *targp, void *bufargp)
{
  int errcode = ERRCODE_SUCCESS;
  short i, n;
  SmaltMapArgs *map = (SmaltMapArgs *)targp;
  SmaltArgBlock *blockp = (SmaltArgBlock *)bufargp;
  const SmaltMapConst *macop = map->smconstp;
  const RMAPFLG_t rmapflg = (RMAPFLG_t)(macop->rmapflg & ~RMAPFLG_ALLPAIR);
  SeqFastq **reads;

  n = blockp->n_iobf;
  if (n < 1) return ERRCODE_SUCCESS;
  /* not on the GPU path: complexity weighting (split reads: rmapGpuBatch -> smaltgpu_map_split, pairs: a round of second calls in rmapGpuPairBatch) */
  if ((macop->rmapflg & RMAPFLG_CMPLXW) || macop->tupcovmin < 0)
    return processArgBlock(errmsgp,
#ifdef THREADS_DEBUG
                           readno,
#endif
                           targp, bufargp);
  if (blockp->iobfp[0].isPaired) {             /* rmapPair for the whole block (integration/rmap_gpu.c: rounds A-D on the GPU) */
    SeqFastq **mates;
    if (!(reads = malloc((size_t)2 * n * sizeof(*reads)))) return ERRCODE_NOMEM;
    mates = reads + n;
    for (i = 0; i < n && !errcode; i++) {               /* as the head of processMapArgs (smalt.c:1102-1137) */
      SmaltIOBuffArg *brgp = blockp->iobfp + i;
      ERRMSG_READNO(errmsgp, brgp->readno + 1);
      ERRMSG_READNAM(errmsgp, seqFastqGetSeqName(brgp->readp));
      if (!brgp->isPaired) errcode = ERRCODE_ASSERT;
      if (!errcode) errcode = seqFastqEncode(brgp->readp, macop->codecp);
      if (!errcode) errcode = seqFastqEncode(brgp->matep, macop->codecp);
      if ((errcode)) { free(reads); ERRMSGNO(errmsgp, errcode); }
      reads[i] = brgp->readp; mates[i] = brgp->matep;
    }
    if (!errcode)
      errcode = rmapGpuPairBatch(errmsgp, map->rmp, reads, mates, n, macop->insert_min, macop->insert_max, macop->pairtyp,
                                 macop->nhitmax_tuple, macop->tupcovmin, macop->min_swatscor, macop->minbasq, SMALT_TARGET_DEPTH,
                                 SMALT_MAX_DEPTH, (RMAPFLG_t)(macop->rmapflg | RMAPFLG_PAIRED), macop->scormtxp, macop->htp, macop->ssp,
                                 macop->codecp);
    for (i = 0; i < n && !errcode; i++) {               /* as the tail of rmapPair + processMapArgs (rmap.c:2080-2110, smalt.c:1166-1183) */
      SmaltIOBuffArg *brgp = blockp->iobfp + i;
      const ResultSet *rsltp, *rslt_matep;
      const ResultPairs *pairp;
      ERRMSG_READNO(errmsgp, brgp->readno + 1);
      ERRMSG_READNAM(errmsgp, seqFastqGetSeqName(brgp->readp));
      /* the pair's sets are the RMap's from a Finish attempt until the Release, whatever happens in between */
      errcode = rmapGpuPairFinish(errmsgp, map->rmp, i, &brgp->pairflg, macop->insert_min, macop->insert_max, macop->pairtyp, macop->rfp);
      if (!errcode) {
        rmapGetData(&rsltp, &rslt_matep, &pairp, NULL, NULL, map->rmp);
        errcode = resultSetAddPairToReport(brgp->rep, macop->ihp, pairp, brgp->pairflg, macop->rsltouflg, rsltp, rslt_matep);
        if (!errcode && MENU_SAMPLE == macop->subprogtyp &&
            ERRCODE_SUCCESS == resultSetInferInsertSize(&brgp->isiz, RSLTSAMSPEC_V1P4, rsltp, rslt_matep))
          brgp->pairflg |= RSLTPAIRFLG_INSERTSIZ;
      }
      rmapGpuPairRelease(map->rmp);
      if ((errcode)) { free(reads); ERRMSGNO(errmsgp, errcode); }
    }
    free(reads);
#ifdef THREADS_DEBUG
    *readno = (i > 1) ? blockp->iobfp->readno : 0;
#endif
    return errcode;
  }
  if (!(reads = malloc((size_t)n * sizeof(*reads)))) return ERRCODE_NOMEM;
  for (i = 0; i < n && !errcode; i++) {                 /* as the head of processMapArgs (smalt.c:1102-1112) */
    SmaltIOBuffArg *brgp = blockp->iobfp + i;
    ERRMSG_READNO(errmsgp, brgp->readno + 1);
    ERRMSG_READNAM(errmsgp, seqFastqGetSeqName(brgp->readp));
    if ((errcode = seqFastqEncode(brgp->readp, macop->codecp))) ERRMSGNO(errmsgp, errcode);
    reads[i] = brgp->readp;
  }
  if (!errcode)
    errcode = rmapGpuBatch(errmsgp, map->rmp, reads, n, macop->nhitmax_tuple, macop->tupcovmin, (int)macop->min_swatscor,
                           macop->swatscordiff, macop->minbasq, SMALT_TARGET_DEPTH, SMALT_MAX_DEPTH, rmapflg, macop->scormtxp,
                           macop->codecp);
  for (i = 0; i < n && !errcode; i++) {                 /* as the tail of processMapArgs (smalt.c:1172-1185) */
    SmaltIOBuffArg *brgp = blockp->iobfp + i;
    const ResultSet *rsltp;
    ERRMSG_READNO(errmsgp, brgp->readno + 1);
    ERRMSG_READNAM(errmsgp, seqFastqGetSeqName(brgp->readp));
    if ((errcode = rmapGpuFinish(errmsgp, map->rmp, i, brgp->readp, SMALT_MAX_DEPTH, macop->scormtxp, macop->rfp, macop->htp,
                                 macop->ssp, macop->codecp)))
      break;
    rmapGetData(&rsltp, NULL, NULL, NULL, NULL, map->rmp);
    if ((errcode = resultSetAddToReport(brgp->rep, macop->rsltouflg, rsltp))) ERRMSGNO(errmsgp, errcode);
  }
  free(reads);
#ifdef THREADS_DEBUG
  *readno = (i > 1) ? blockp->iobfp->readno : 0;
#endif
  return errcode;
}

static int smaltgpu_threadsSetTask(uint8_t task_typ, short n_threads, THREAD_INITF *initf, const void *initargp, THREAD_PROCF *procf,
                                   THREAD_CLEANF *cleanf, THREAD_CHECKF *checkf, THREAD_CMPF *cmpf, size_t argsz)
{
  if (task_typ == THRTASK_PROC && !getenv("SMALTGPU_PER_READ")) procf = processArgBlockGpu;     /* SMALTGPU_PER_READ: keep one rmapSingle per read */
  if (task_typ == THRTASK_ARGBUF && !getenv("SMALTGPU_PER_READ") && initargp) {
    /* The reference sizes a block at 32 reads per worker thread (smalt.c:466).  The GPU path wants enough reads in flight
     * for two combined batches whatever the thread count: blocks of at least 1024 reads (SMALTGPU_BLOCK_READS overrides).
     * The block size changes nothing in what is computed for a read. */
    SmaltMapConst *mc = (SmaltMapConst *)initargp;
    int want = getenv("SMALTGPU_BLOCK_READS") ? atoi(getenv("SMALTGPU_BLOCK_READS")) : 1024;
    if (want > 16384) want = 16384;
    if (mc->threadblksz < want) mc->threadblksz = (short)want;
  }
  return threadsSetTask(task_typ, n_threads, initf, initargp, procf, cleanf, checkf, cmpf, argsz);
}
