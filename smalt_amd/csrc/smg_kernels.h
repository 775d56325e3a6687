// smg_kernels.h -- host-callable launchers of the gfx950 kernels (smg_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include "smg_stages.hpp"

namespace smg {

enum : int { SW_FULL_WMAX = 1016 };   // longest reference window of the register-tiled K2a kernel

int launch_encode(hipStream_t s, const uint8_t *bases, const uint64_t *off, uint32_t n, uint8_t *codes, uint8_t *codes_rc);
int launch_seed(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, uint8_t *scratch, size_t sbytes, uint32_t nslots);
int launch_cands(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, uint8_t *scratch, size_t sbytes, uint32_t nslots,
                 uint32_t hcap, uint32_t ngrp, uint32_t segcap, uint32_t candcap, int slot_per_read);
int launch_replay(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p);
int launch_align(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, uint8_t *scratch, size_t sbytes, uint32_t nslots,
                 uint32_t wincap, uint64_t dircap, uint32_t rescap, uint32_t dstrcap);
int sw_full_geometry(uint32_t qmax_len, int *G, int *C);
int launch_sw_full(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, uint32_t qmax_len, uint32_t ntask_cap, uint32_t grid);
int launch_sw_scalar(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, int *rows, uint32_t rowlen, uint32_t nthreads,
                     uint32_t qmax_len);
int launch_sw_full_raw(hipStream_t s, const uint8_t *q, const uint32_t *qo, const uint8_t *r, const uint32_t *ro, uint32_t n,
                       const MapPar &p, int32_t *sc, uint32_t qmax_len);

}  // namespace smg
