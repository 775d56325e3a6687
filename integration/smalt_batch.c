/* smalt_batch.c -- OUR translation unit around the reference's smalt.c (its text is taken from the reference tree at
 * build time by oracle/Makefile, target ref_gpu; nothing is copied).  The one change to the program: the worker task
 * that maps a block of reads (processArgBlock, smalt.c:1221: one rmapSingle per read) is replaced, at the point where
 * smalt.c registers it with the thread pool, by a version that sends the whole block through the GPU path in one batch
 * (integration/rmap_gpu.c: rmapGpuBatch) and then runs the reference's own per-read tail -- post-processing and report
 * (smalt.c:1172-1185) -- read by read.  Paired reads and the split/complexity modes keep the reference's own worker. */
#include "threads.h"
static int smaltgpu_threadsSetTask(uint8_t task_typ, short n_threads, THREAD_INITF *initf, const void *initargp, THREAD_PROCF *procf,
                                   THREAD_CLEANF *cleanf, THREAD_CHECKF *checkf, THREAD_CMPF *cmpf, size_t argsz);
#define threadsSetTask smaltgpu_threadsSetTask
#include "smalt.c"
#undef threadsSetTask

extern int rmapGpuBatch(ErrMsg *errmsgp, RMap *rmp, SeqFastq *const *reads, int n, int ktuple_maxhit, double tupcovmin,
                        int min_swatscor, int min_swatscor_below_max, unsigned char min_basqval, short target_depth, short max_depth,
                        RMAPFLG_t rmapflg, const ScoreMatrix *scormtxp, const SeqCodec *codecp);
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
  if (blockp->iobfp[0].isPaired || (rmapflg & (RMAPFLG_SPLIT | RMAPFLG_CMPLXW)) || macop->tupcovmin < 0)
    return processArgBlock(errmsgp,
#ifdef THREADS_DEBUG
                           readno,
#endif
                           targp, bufargp);
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
