/* gpu_combine.h -- see gpu_combine.c */
#ifndef GPU_COMBINE_H
#define GPU_COMBINE_H
#include <stddef.h>
#include <stdint.h>
#include "smaltgpu.h"

typedef struct {              /* results of one request, owned by the caller (buffers grow, never shrink) */
  uint32_t n;
  uint64_t *res_off; smaltgpu_result *res; unsigned char *dstr; smaltgpu_readstat *stat;
  size_t cap_off, cap_res, cap_dstr, cap_stat;
} GpuCombOut;

/* A request that belongs to one of rmapPair's rounds (integration/rmap_gpu.c: rmapGpuPairBatch) carries the per-read context
 * of include/smaltgpu.h's smaltgpu_callctx, indexed like its reads; only requests of the same kind form a batch. */
enum { GPUCOMB_PLAIN = 0, GPUCOMB_TOTALS = 1 /* smaltgpu_hit_totals into tot_out */, GPUCOMB_APPEND = 2 /* unrestricted, running maxima */,
       GPUCOMB_RESTRICTED = 3, GPUCOMB_FINE = 4, GPUCOMB_SPLIT = 5 /* as APPEND, k-mer words from a stretch of the read (mapSecondary) */ };
typedef struct {
  int kind;
  const uint64_t *iv_off; const smaltgpu_interval *iv;     /* RESTRICTED, FINE */
  const int32_t *minsw;                                    /* FINE */
  const int32_t *prevmax;                                  /* APPEND, FINE, SPLIT */
  const uint32_t *seedrange;                               /* SPLIT: (first, last) base per read */
  uint32_t *tot_out;                                       /* TOTALS: n hit totals */
} GpuCombCtx;

/* map the n reads of the caller (bases/quals concatenated, off[n+1]) as part of a combined batch on whichever of the ndev
 * devices (index images ixs[0..ndev)) has a free mapper; blocks until `out` holds the caller's slice.  Requests that differ in their parameters or in
 * whether they carry base qualities are never put into the same batch.  On failure errbuf (may be
 * NULL) receives the library's message (smaltgpu_last_error() is per thread and the batch may have run on another one). */
int gpuCombineSubmit(int ndev, const smaltgpu_index *const *ixs, const char *bases, const char *quals, const uint64_t *off, uint32_t n,
                     const smaltgpu_params *par, GpuCombOut *out, char *errbuf, size_t errcap);
int gpuCombineSubmitCtx(int ndev, const smaltgpu_index *const *ixs, const char *bases, const char *quals, const uint64_t *off, uint32_t n,
                        const smaltgpu_params *par, const GpuCombCtx *ctx, GpuCombOut *out, char *errbuf, size_t errcap);
/* after an error a worker is going to end the program with: no further batch is launched, the call returns when the batches in
 * flight are done.  Requests that arrive or still wait afterwards return GPUCOMB_CLOSING. */
enum { GPUCOMB_CLOSING = -100 };
void gpuCombineClose(void);
#endif
