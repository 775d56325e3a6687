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

/* map the n reads of the caller (bases/quals concatenated, off[n+1]) as part of a combined batch on whichever of the ndev
 * devices (index images ixs[0..ndev)) has a free mapper; blocks until `out` holds the caller's slice.  All concurrent callers must pass the same parameters.  On failure errbuf (may be
 * NULL) receives the library's message (smaltgpu_last_error() is per thread and the batch may have run on another one). */
int gpuCombineSubmit(int ndev, const smaltgpu_index *const *ixs, const char *bases, const char *quals, const uint64_t *off, uint32_t n,
                     const smaltgpu_params *par, GpuCombOut *out, char *errbuf, size_t errcap);
#endif
