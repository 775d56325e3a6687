// smg_kernels.h -- host-callable launchers of the gfx950 kernels (smg_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include "smg_cands.hpp"

namespace smg {

enum : int { SW_FULL_WMAX = 1016,   // longest reference window of the register-tiled K2a kernels
             SW_SHORT_WMAX = 248 };  // windows up to here take the small-LDS instance of the packed kernel (more resident waves)

int launch_encode(hipStream_t s, const uint8_t *bases, const uint64_t *off, uint32_t n, uint8_t *codes, uint8_t *codes_rc);
int launch_gather_reads(hipStream_t s, uint8_t *dst_bases, uint8_t *dst_quals, const uint64_t *dst_off, uint32_t n, const uint32_t *ids,
                        const uint8_t *const src_bases[2], const uint8_t *const src_quals[2], const uint64_t *const src_off[2]);
int launch_fine_index(hipStream_t s, const Batch &b, const DevIndex &ix);     // needs b.iv_off/iv, b.fine_idx/fine_pos/fine_off
int launch_seed(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, uint8_t *scratch, size_t sbytes, uint32_t nslots);
struct CandGeom {              // scratch geometry of the candidate stage (both code paths share one HBM slot)
  uint32_t hcap;               // hits of both strands, power of two (sequential path)
  uint32_t hcap_strand;        // hits of one strand (wave-parallel path)
  uint32_t ngrp, segcap, candcap;
  size_t slot_bytes;
  int debug;                   // keep per-read slots + the grouped hit words for smaltgpu_dump_read
  uint32_t window;             // test hook: hits per LDS window of the candidate stage (0 = default)
  uint32_t lds_hits, tab;      // LDS block of the candidate stage: hits of the working set, per-list table entries
  int pass;                    // 0: only pass; 1: first of two (a read that overflows its slot is deferred); 2: second pass over full-size slots
};
inline size_t cand_slot_bytes(const CandGeom &g, uint32_t qmax, int s) {
  size_t a = cand_scratch_bytes(qmax, s, g.hcap, g.ngrp, g.segcap, g.candcap);
  size_t b = cands_v2_hbm_bytes(qmax, s, g.hcap_strand, g.ngrp, g.candcap, true);
  return a > b ? a : b;
}
int launch_hits(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, uint32_t W, uint32_t tab, uint32_t nwg);
int launch_cands(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, uint8_t *scratch, uint32_t nslots, const CandGeom &g);
int launch_replay(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p);
int launch_align(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, uint8_t *scratch, size_t sbytes, uint32_t nslots,
                 uint32_t wincap, uint64_t dircap, uint32_t rescap, uint32_t dstrcap, int pass);
int sw_full_geometry(uint32_t qmax_len, int *G, int *C);
int launch_sw_full(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, uint32_t qmax_len, uint32_t ntask_cap, uint32_t grid);
int launch_sw_strip(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, void *bnd, uint8_t *win, uint32_t wcap, uint32_t grid);
int launch_sw_strip_raw(hipStream_t s, const uint8_t *q, const uint32_t *qo, const uint8_t *r, const uint32_t *ro, uint32_t n, const MapPar &p,
                        int32_t *sc, void *bnd, uint8_t *win, uint32_t wcap, uint32_t grid);
int launch_sw_scalar(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, int *rows, uint32_t rowlen, uint32_t nthreads,
                     uint32_t qmax_len);
int launch_sw_full_raw(hipStream_t s, const uint8_t *q, const uint32_t *qo, const uint8_t *r, const uint32_t *ro, uint32_t n,
                       const MapPar &p, int32_t *sc, uint32_t qmax_len, int packed16);
int launch_rank_sort_raw(hipStream_t s, const uint32_t *keys, const uint32_t *off, uint32_t narr, int nneed, int in_lds, uint32_t *kv,
                         uint32_t *out_key, uint32_t *out_idx);

}  // namespace smg
