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

/* map the n reads of the caller (bases/quals concatenated, off[n+1]) as part of a combined batch on device dev; blocks
 * until `out` holds the caller's slice.  All concurrent callers must pass the same parameters. */
int gpuCombineSubmit(int dev, const smaltgpu_index *ix, const char *bases, const char *quals, const uint64_t *off, uint32_t n,
                     const smaltgpu_params *par, GpuCombOut *out);
#endif
