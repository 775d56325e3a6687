// smg_kernels.hip -- gfx950 kernels of libsmaltgpu: thin wave-per-read wrappers around the
// stage functions of smg_stages.hpp, and the wide Smith-Waterman score pass (K2a).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "smg_kernels.h"
#include "smg_stages.hpp"

namespace smg {

// ---------------------------------------------------------------------------------------
// read encoding: ASCII -> 3-bit codes (sequence.c:287-322) + reverse complement
// (sequence.c:884-896: non-standard codes are kept).  One wave per read.
// ---------------------------------------------------------------------------------------
__device__ inline uint8_t code_of(uint8_t c) {
  c &= 0xDF;   // upper case for letters
  return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : (c == 'T' || c == 'U') ? 3 : 5;
}

__global__ void __launch_bounds__(64) k_encode(const uint8_t *bases, const uint64_t *off, uint32_t n, uint8_t *codes, uint8_t *codes_rc) {
  for (uint32_t r = blockIdx.x; r < n; r += gridDim.x) {
    const uint64_t o = off[r];
    const uint32_t len = (uint32_t)(off[r + 1] - o);
    for (uint32_t i = threadIdx.x; i < len; i += 64) {
      uint8_t c = code_of(bases[o + i]);
      codes[o + i] = c;
      codes_rc[o + len - 1 - i] = (c & 4) ? c : (uint8_t)(3 - c);
    }
  }
}

// The reads of one round of a block of pairs, taken from the two batches that are resident in HBM (reads and mates): read i of
// the round is read ids[i] >> 1 of batch ids[i] & 1.  One workgroup per read, bytes copied 16 at a time where both sides
// allow it (HBM to HBM, coalesced on both ends).
__global__ void __launch_bounds__(64) k_gather_reads(uint8_t *dst_bases, uint8_t *dst_quals, const uint64_t *dst_off, uint32_t n, const uint32_t *ids,
                                                     const uint8_t *b0, const uint8_t *b1, const uint8_t *q0, const uint8_t *q1, const uint64_t *o0, const uint64_t *o1) {
  for (uint32_t r = blockIdx.x; r < n; r += gridDim.x) {
    const uint32_t id = ids[r], w = id & 1u, i = id >> 1;
    const uint64_t so = w ? o1[i] : o0[i], d = dst_off[r];
    const uint32_t len = (uint32_t)(dst_off[r + 1] - d);
    const uint8_t *sb = (w ? b1 : b0) + so, *sq = dst_quals ? (w ? q1 : q0) + so : nullptr;
    if ((((uintptr_t)sb | (uintptr_t)(dst_bases + d)) & 15u) == 0 && (!sq || (((uintptr_t)sq | (uintptr_t)(dst_quals + d)) & 15u) == 0)) {
      const uint32_t nv = len >> 4;
      for (uint32_t v = threadIdx.x; v < nv; v += 64) {
        ((uint4 *)(dst_bases + d))[v] = ((const uint4 *)sb)[v];
        if (sq) ((uint4 *)(dst_quals + d))[v] = ((const uint4 *)sq)[v];
      }
      for (uint32_t t = (nv << 4) + threadIdx.x; t < len; t += 64) { dst_bases[d + t] = sb[t]; if (sq) dst_quals[d + t] = sq[t]; }
    } else
      for (uint32_t t = threadIdx.x; t < len; t += 64) { dst_bases[d + t] = sb[t]; if (sq) dst_quals[d + t] = sq[t]; }
  }
}

// ---------------------------------------------------------------------------------------
// S1 + S2: one wave per (read, strand); scratch in LDS when it fits, else in HBM slots
// ---------------------------------------------------------------------------------------
// Stage code reaches its LDS arrays through generic pointers (the same source serves HBM slots).  A flat
// access whose register address lies below the LDS aperture faults even if the instruction offset
// brings it back inside (e.g. a[i + 1] with i == -1 compiled as base a + 4*i, offset 4), so no array
// starts at LDS offset 0.
enum : uint32_t { LDS_GUARD = 64 };

// Persistent workgroups take the next item from a batch-wide cursor: per-read cost is heavy-tailed (repeat-rich
// reads), a static stride leaves most of the chip idle behind the slowest workgroup.
__device__ inline uint32_t next_item(uint32_t *cursor, uint32_t *lds_slot) {
  __syncthreads();
  if (threadIdx.x == 0) *lds_slot = atomicAdd(cursor, 1u);
  __syncthreads();
  return *lds_slot;
}

__global__ void __launch_bounds__(64) k_seed(Batch b, DevIndex ix, MapPar p, uint8_t *gscratch, size_t gbytes, int use_lds) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint8_t *base = use_lds ? lds + LDS_GUARD : gscratch + gbytes * blockIdx.x;
  SeedScratch x = seed_scratch_carve(base, b.qmax, b.fine_idx ? (int)FINE_S : ix.s);
  unsigned long long nlook = 0;
  __shared__ uint32_t qslot;
  for (uint32_t rs = next_item(b.next_item + 0, &qslot); rs < 2 * b.nreads; rs = next_item(b.next_item + 0, &qslot)) {
    nlook += stage_seed(b, read_index(b, ix, rs >> 1), p, rs >> 1, rs & 1, x);
    __syncthreads();
  }
  if (threadIdx.x == 0 && nlook) atomicAdd(b.work + WK_LOOKUPS, nlook);
}

// The on-the-fly index of rmapPair's rescue round (rmap.c:495-517 setupFineHashTable -> hashTableSetUp with an interval set,
// hashidx.c:549-575): perfect type, k = 5, s = 1, over the windows of ONE read's intervals.  One wave per read: key
// histogram in LDS, exclusive scan = idx, positions scattered behind per-key cursors, then every key's (short) list put
// in ascending order -- the reference fills the lists in ascending order because its intervals are sorted and disjoint.
__global__ void __launch_bounds__(64) k_fine_index(Batch b, DevIndex ix) {
  __shared__ uint32_t cnt[FINE_NKEYS + 1];
  for (uint32_t r = blockIdx.x; r < b.nreads; r += gridDim.x) {
    uint32_t *idx = b.fine_idx + (size_t)r * FINE_IDX_STRIDE;
    uint32_t *pos = b.fine_pos + b.fine_off[r];
    const uint32_t cap = b.fine_off[r + 1] - b.fine_off[r];
    const uint32_t niv = b.iv_off[r + 1] - b.iv_off[r];
    const IvRec *iv = b.iv + b.iv_off[r];
    for (uint32_t i = threadIdx.x; i <= (uint32_t)FINE_NKEYS; i += 64) cnt[i] = 0;
    __syncthreads();
    for (int pass = 0; pass < 2; pass++) {
      for (uint32_t v = 0; v < niv; v++) {
        const uint32_t sl = iv[v].hi - iv[v].lo + 1;
        if (sl < (uint32_t)FINE_K) continue;                        // hashidx.c:561-563
        const uint64_t g0 = ix.sop[iv[v].sx] + iv[v].lo;
        const uint32_t nk = sl - FINE_K + 1;
        for (uint32_t j = threadIdx.x; j < nk; j += 64) {
          uint32_t w = 0;
          bool ok = true;
          for (int t = 0; t < FINE_K; t++) { const uint32_t c = ref_code(ix.packed, g0 + j + (uint32_t)t); if (c & 4u) ok = false; w = (w << 2) | (c & 3u); }
          if (!ok) continue;                                         // words with non-standard bases are not indexed (hashidx.c:497-503)
          if (pass == 0) atomicAdd(&cnt[w], 1u);
          else { const uint32_t at = atomicAdd(&cnt[w], 1u); if (at < cap) pos[at] = (uint32_t)(g0 + j); }     // serial = base offset (s = 1)
        }
      }
      __syncthreads();
      if (pass == 0) {
        if (threadIdx.x == 0) { uint32_t c = 0; for (uint32_t i = 0; i < (uint32_t)FINE_NKEYS; i++) { const uint32_t t = cnt[i]; cnt[i] = c; c += t; } cnt[FINE_NKEYS] = c; }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i <= (uint32_t)FINE_NKEYS; i += 64) idx[i] = cnt[i];
        __syncthreads();
      }
    }
    for (uint32_t key = threadIdx.x; key < (uint32_t)FINE_NKEYS; key += 64) {      // ascending serials per key
      const uint32_t a = idx[key], e = idx[key + 1] < cap ? idx[key + 1] : cap;
      for (uint32_t i = a + 1; i < e; i++) { const uint32_t v = pos[i]; uint32_t j = i; while (j > a && pos[j - 1] > v) { pos[j] = pos[j - 1]; j--; } pos[j] = v; }
    }
    __syncthreads();
  }
}

int launch_fine_index(hipStream_t s, const Batch &b, const DevIndex &ix) {
  if (!b.nreads) return 0;
  hipLaunchKernelGGL(k_fine_index, dim3(b.nreads < 65535u ? b.nreads : 65535u), dim3(64), 0, s, b, ix);
  return (int)hipGetLastError();
}

// S3 ahead of the candidate stage: one wave per read strand (stage_hits, smg_cands.hpp); 4 waves per SIMD by registers,
// the LDS block (window keys + list tables) sized by the launcher for as many
template <int WAVES>
__global__ void __launch_bounds__(64, WAVES) k_hits(Batch b, DevIndex ix, MapPar p, uint32_t W, uint32_t tab, uint32_t lds_bytes) {
  extern __shared__ __align__(16) uint8_t lds[];
  unsigned long long ph[4] = {0, 0, 0, 0};
  __shared__ uint32_t qslot;
  if (ix.nseq > 0 && ix.nseq < 512) {               // first k-mer serial of every sequence: looked up per hit
    uint32_t *sl = (uint32_t *)(lds + LDS_GUARD + lds_bytes);
    for (int i = (int)threadIdx.x; i <= ix.nseq; i += 64) sl[i] = ix.seqlo[i];
    ix.seqlo = sl;
    __syncthreads();
  }
  HitsScratch x;
  x.lds = lds + LDS_GUARD; x.lds_bytes = lds_bytes; x.W = W; x.tab = tab;
  const uint32_t nitem = 2 * b.nreads;
  for (uint32_t it = next_item(b.hits_cursor, &qslot); it < nitem; it = next_item(b.hits_cursor, &qslot)) {
    stage_hits(b, ix, p, it >> 1, it & 1u, x, ph);
    __syncthreads();
  }
  if (threadIdx.x == 0) for (int i = 0; i < 3; i++) if (ph[i]) atomicAdd(b.work + WK_PHASE0 + i, ph[i]);
}

// S3 - S7: one wave per read.  Reads the wave-parallel form covers (smg_cands.hpp) keep their
// per-strand working set in LDS; everything else takes the sequential restatement on the HBM slot.
// LONGK: the mapper takes reads of 256 bases and more; every read then goes through the general instance of the
// wave-parallel form (mappers for short reads keep the lean one: fewer registers, more resident waves).
template <bool LONGK, bool SPLIT = false, int WAVES = 2>
__global__ void __launch_bounds__(64, WAVES) k_cands(Batch b, DevIndex ix, MapPar p, uint8_t *gscratch, CandGeom g, uint32_t lds_bytes) {
  extern __shared__ __align__(16) uint8_t lds[];
  unsigned long long nhit = 0;
  unsigned long long ph[16] = {0};
  __shared__ uint32_t qslot;
  // the first k-mer serial of every sequence goes to LDS (behind the working set): each hit looks its sequence up
  // there (a binary search = five reads) instead of in global memory
  if (lds_bytes && ix.nseq > 0 && ix.nseq < 512) {
    uint32_t *sl = (uint32_t *)(lds + LDS_GUARD + lds_bytes);
    for (int i = (int)threadIdx.x; i <= ix.nseq; i += 64) sl[i] = ix.seqlo[i];
    ix.seqlo = sl;
    __syncthreads();
  }
  uint32_t *cursor = b.next_item + NEXT_ITEM_STRIDE * (g.pass == 2 ? 4 : 1);
  // second pass: only the reads the first pass deferred (usually none: the launch ends at once)
  const uint32_t nitem = g.pass == 2 ? *b.cands_retry_n : b.nreads;
  for (uint32_t it = next_item(cursor, &qslot); it < nitem; it = next_item(cursor, &qslot)) {
    const uint32_t r = g.pass == 2 ? b.cands_retry[it] : it;
    uint8_t *base = gscratch + g.slot_bytes * (g.debug ? r : blockIdx.x);
    const DevIndex rix = read_index(b, ix, r);
    if (cands_v2_applicable(p, rix.k, rix.s, read_len(b, r), b.iv_off != nullptr)) {
      CandsV2Scratch x = cands_v2_carve(lds + LDS_GUARD, lds_bytes, base, b.qmax, rix.s, g.hcap_strand, g.ngrp, g.candcap, g.debug != 0);
      x.window = g.window; x.lds_hits = g.lds_hits; x.tab = g.tab; x.pass = g.pass;
      nhit += stage_cands_v2<LONGK, SPLIT>(b, rix, p, r, x, ph);
    } else if (b.iv_off) {                                           // interval-restricted calls exist in the wave-parallel form only
      if (threadIdx.x == 0) { CandHdr &ch = b.ch[r]; ch.ncand = ch.n_sort = ch.n_mincover = ch.n_reserved = 0; ch.rc_off = 0; ch.err = SMG_ERR_ASSERT; ch.err_site = __LINE__; ch.max_cover = ch.max2nd_cover = 0; }
    } else if (g.pass == 1) {                                        // the sequential form waits for the full-size slots
      if (threadIdx.x == 0) { CandHdr &ch = b.ch[r]; ch.ncand = ch.n_sort = ch.n_mincover = ch.n_reserved = 0; ch.rc_off = 0; ch.err = SMG_ERR_RETRY; ch.err_site = __LINE__; }
    } else {
      CandScratch x = cand_scratch_carve(base, b.qmax, ix.s, g.hcap, g.ngrp, g.segcap, g.candcap);
      nhit += stage_cands(b, ix, p, r, x);
    }
    __syncthreads();
    if (g.pass == 1 && threadIdx.x == 0 && b.ch[r].err == SMG_ERR_RETRY) b.cands_retry[atomicAdd(b.cands_retry_n, 1u)] = r;
  }
  if (threadIdx.x == 0) {
    if (nhit) atomicAdd(b.work + WK_HITS, nhit);
    for (int i = 0; i < 9; i++) if (ph[i]) atomicAdd(b.work + WK_PHASE0 + i, ph[i]);
    if (ph[9]) { atomicAdd(b.work + WK_NCAND, ph[9]); atomicAdd(b.work + WK_NKEPT, ph[10]); }
    for (int i = 11; i < 16; i++) if (ph[i]) atomicAdd(b.work + WK_PHASE0 + i, ph[i]);
  }
}

// O1: one thread per read
__global__ void __launch_bounds__(256) k_replay(Batch b, DevIndex ix, MapPar p) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long scored = 0;
  if (r < b.nreads) {
    stage_replay(b, read_index(b, ix, r), p, r);
    scored = (unsigned long long)b.ctl[r].n_scored;                               // candidates the reference would have scored
  }
  for (int o = 32; o > 0; o >>= 1) scored += __shfl_xor(scored, o);               // one atomic per wave, not per read
  if ((threadIdx.x & 63) == 0 && scored) atomicAdd(b.work + WK_CELLS_BAND, scored);
}

// K3: one wave per read; hot arrays in LDS, results and oversized direction matrices in the HBM slot
template <bool WIDE>
__global__ void __launch_bounds__(64) k_align(Batch b, DevIndex ix, MapPar p, uint8_t *gscratch, size_t gbytes, uint32_t wincap,
                                              uint64_t dircap, uint32_t rescap, uint32_t dstrcap, uint32_t lds_bytes, int pass) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint8_t *base = gscratch + gbytes * blockIdx.x;
  AlignScratch x = align_scratch_carve_lds(lds_bytes ? lds + LDS_GUARD : nullptr, lds_bytes, base, b.qmax, wincap, dircap, rescap, dstrcap);
  x.rows_form = !(pass & 0x100);          // bit 8 of the launch's pass word: anti-diagonal form for narrow bands (test hook)
  pass &= 0xff;
  x.pass = pass;
  __shared__ int2 strip_ring[WIDE ? 256 : 1];
  x.ring = WIDE ? (void *)strip_ring : nullptr;
  __shared__ uint32_t qslot;
  uint32_t *cursor = b.next_item + NEXT_ITEM_STRIDE * (pass == 2 ? 3 : 2);
  if (pass == 2) {                        // only the reads the first pass deferred (usually none: the launch ends at once)
    const uint32_t n = *b.align_retry_n;
    for (uint32_t i = next_item(cursor, &qslot); i < n; i = next_item(cursor, &qslot)) {
      stage_align<WIDE>(b, ix, p, b.align_retry[i], x);
      __syncthreads();
    }
    align_tally_flush(b, x);
    return;
  }
  for (uint32_t r = next_item(cursor, &qslot); r < b.nreads; r = next_item(cursor, &qslot)) {
    stage_align<WIDE>(b, ix, p, r, x);
    __syncthreads();
  }
  align_tally_flush(b, x);
}

// ---------------------------------------------------------------------------------------
// K2a: un-banded Smith-Waterman score pass (swsimd.c:868 -- textbook Gotoh maximum).
//
// A task = one ranked candidate (read x reference window).  G adjacent lanes own one task;
// lane g keeps C consecutive query columns (H, E and a byte selector per column) in
// registers and sweeps the window row by row, one row behind lane g-1 (anti-diagonal skew
// ACROSS lanes only).  Per row a lane receives H[row][first-1] and the running F from its
// left neighbour with DPP row_shr:1 -- no LDS traffic in the recurrence.  Substitution
// scores come from one v_perm_b32 per cell on an 8-byte row of the biased score matrix.
// Window bases are decoded once per task into LDS.  Rows beyond a task's window and columns
// beyond the read are fed 'N' (score 0), which can never raise the maximum, so lanes need no
// predication.  int32 arithmetic equals the reference's 8-bit pass, and its 16-bit re-run on
// saturation, for any score < 65535.
// ---------------------------------------------------------------------------------------
template <int G>
__device__ inline int shr1(int v) {          // value of the lane to the left inside a 16-lane row
  return __builtin_amdgcn_update_dpp(0, v, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
}

template <int G, int C>
__global__ void __launch_bounds__(64) k_sw_full(Batch b, DevIndex ix, MapPar p, uint32_t ntask_cap, int after16) {
  // after16: the packed 16-bit kernel ran first; what is left are the candidates of reads with non-ACGT codes
  if (after16 && b.work[WK_QN_TASKS] == 0) return;
  constexpr int NG = 64 / G;                 // tasks in flight per wave
  constexpr int WMAX = SW_FULL_WMAX;         // longest window handled here
  __shared__ uint8_t win[NG][WMAX + 8];
  __shared__ uint2 tab[8];
  const int lane = threadIdx.x, g = lane % G, grp = lane / G;
  const uint32_t ntask = min(*b.rc_count, ntask_cap);
  const int bias = (p.mismatch < p.mismatch - p.match ? -p.mismatch : -(p.mismatch - p.match));   // >= -min(M)
  const int gi = -p.gap_init, ge = -p.gap_ext;
  if (lane < 8) {                            // biased score matrix rows (score.c:138-173; rows 4,6 = N, 7 = A)
    int rb = lane == 7 ? 0 : ((lane == 6 || lane == 4) ? 5 : lane);
    uint32_t w[2] = {0, 0};
    for (int qc = 0; qc < 8; qc++) {
      int v = (rb == 5 || qc >= 4) ? 0 : ((rb == qc) ? p.match : p.mismatch);
      w[qc >> 2] |= (uint32_t)((v + bias) & 0xff) << (8 * (qc & 3));
    }
    tab[lane] = make_uint2(w[0], w[1]);
  }
  __syncthreads();
  const uint32_t ngroups = gridDim.x * NG;
  unsigned long long cells = 0, ntasks_done = 0;
  for (uint32_t t0 = blockIdx.x * NG; t0 < ntask; t0 += ngroups) {
    const uint32_t t = t0 + grp;
    RCand c;
    bool live = false;
    uint32_t qlen = 0, wlen = 0;
    uint64_t gbase = 0;
    if (t < ntask) {
      c = b.rcpool[t];
      qlen = read_len(b, c.rid);
      wlen = (uint32_t)(c.re - c.rs + 1);
      live = !(c.flags & (RCF_BANDED | RCF_ERR | RCF_SCORED)) && wlen <= (uint32_t)WMAX && qlen <= (uint32_t)(G * C);
      gbase = (c.sqidx < 0 ? 0ull : ix.sop[c.sqidx]) + c.rs;
    }
    if (!live) { qlen = 0; wlen = 0; }
    // window -> LDS (sequence.c:1499 decode folded into the fetch)
    for (uint32_t i = g; i < wlen; i += G) win[grp][i] = (uint8_t)ref_code(ix.packed, gbase + i);
    // query columns of this lane: selector byte = code (0..3 standard, 5 = N -> score 0 column)
    uint32_t sel[C];
    {
      const uint8_t *q = ((c.flags & RCF_REVERSE) ? b.codes_rc : b.codes) + (live ? b.read_off[c.rid] : 0);
#pragma unroll
      for (int cc = 0; cc < C; cc++) {
        uint32_t j = (uint32_t)(g * C + cc);
        uint32_t qc = (live && j < qlen) ? q[j] : 5u;
        sel[cc] = 0x0c0c0c00u | qc;
      }
    }
    int H[C], E[C];
#pragma unroll
    for (int cc = 0; cc < C; cc++) { H[cc] = 0; E[cc] = 0; }
    int best = 0, F = 0, prev_hl = 0;
    // wave-uniform number of steps
    int nstep = (int)wlen + G - 1;
    for (int o = 32; o > 0; o >>= 1) nstep = max(nstep, __shfl_xor(nstep, o));
    __syncthreads();
    for (int step = 0; step < nstep; step++) {
      const int row = step - g;
      const uint32_t rb = (row >= 0 && row < (int)wlen) ? win[grp][row] : 5u;
      const uint2 tr = tab[rb];
      int hl = shr1<G>(H[C - 1]);
      int fin = shr1<G>(F);
      if (g == 0) { hl = 0; fin = 0; }
      int diag = prev_hl;
      prev_hl = hl;
      F = fin;
#pragma unroll
      for (int cc = 0; cc < C; cc++) {
        const int w = (int)__builtin_amdgcn_perm(tr.y, tr.x, sel[cc]);
        const int h = diag + w - bias;
        const int hh = max(max(h, E[cc]), F);       // v_max3_i32; E, F >= 0 keep H >= 0
        best = max(best, hh);
        diag = H[cc];
        H[cc] = hh;
        const int tt = hh - gi;
        E[cc] = max(max(E[cc] - ge, tt), 0);
        F = max(max(F - ge, tt), 0);
      }
    }
    for (int o = G / 2; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
    if (live && g == 0) {
      b.rcpool[t].swscor = best;
      b.rcpool[t].flags = c.flags | RCF_SCORED | (best >= 65535 ? RCF_BANDED : 0u);   // ERRCODE_SWATEXCEED -> K2b
      cells += (unsigned long long)qlen * wlen;
      ntasks_done += best < 65535 ? 1 : 0;      /* a score that left 16 bits waits for K2b */
    }
    __syncthreads();
  }
  for (int o = 32; o > 0; o >>= 1) { cells += __shfl_xor(cells, o); ntasks_done += __shfl_xor(ntasks_done, o); }
  if (lane == 0 && cells) { atomicAdd(b.work + WK_CELLS_FULL, cells); atomicAdd(b.work + WK_TASKS_FULL, ntasks_done); }
}

// ---------------------------------------------------------------------------------------
// K2a, two tasks per lane group in packed 16-bit halves (v_pk_*_u16).
//
// Same tiling as k_sw_full; every register holds the cell of task A in its low half and the cell of
// task B (the next ranked candidate) in its high half.  All quantities are kept non-negative and
// subtractions saturate at 0, which is the recurrence of the 32-bit kernel with its max(.., 0)
// folded into the subtract: H = max(sat(Hdiag + (w + bias) - bias), E, F), E = max(sat(E - ge),
// sat(H - gi)).  One v_perm_b32 yields both substitution scores from an 8-byte source holding the
// ACGT row of either task (selector bytes {qA, 0x0c, 4 + qB, 0x0c}).  That leaves no byte for a
// query 'N' (score 0), so reads with non-ACGT codes stay with the 32-bit kernel (RCF_QN); columns
// beyond the read take the constant 0 (score -bias <= 0) and rows beyond the window an all-'N' row,
// neither of which can raise a maximum.  Requires match * 512 + bias < 65535 (checked by the launcher).
// ---------------------------------------------------------------------------------------
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ inline us2 as_us2(uint32_t v) { return __builtin_bit_cast(us2, v); }
__device__ inline uint32_t as_u32(us2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ inline us2 pk_max(us2 a, us2 b) { return __builtin_elementwise_max(a, b); }
__device__ inline us2 pk_subs(us2 a, us2 b) { return __builtin_elementwise_sub_sat(a, b); }
// sum of two packed pairs whose halves cannot overflow (score + biased substitution score < 65536, checked by the
// launcher): one full-rate 32-bit add instead of a packed add at half the issue rate
__device__ inline us2 pk_add_nc(us2 a, us2 b) { return as_us2(as_u32(a) + as_u32(b)); }
// Packed maximum of three: gfx950's v_pk_maximum3_f16 on the u16 bit patterns.  Non-negative half floats order like
// their bit patterns (denormals are kept in the default mode), and the values here stay far below 0x7C00 (infinity):
// scores of reads up to 512 bases with match <= 15 plus the bias.  Checked bit for bit against the integer maximum
// on 10^6 random triples including the denormal range.
typedef _Float16 hf2 __attribute__((ext_vector_type(2)));
__device__ inline us2 pk_max3(us2 a, us2 b, us2 c) {
  const hf2 r = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(hf2, a), __builtin_bit_cast(hf2, b)), __builtin_bit_cast(hf2, c));
  return __builtin_bit_cast(us2, r);
}
__device__ inline uint32_t shr1_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, true); }

enum : int { SW16_BLK = 5 };
struct Sw16Par { uint32_t mm4, dlt, n4; us2 bias, gi, ge; int fp; uint32_t hi_match, hi_mismatch; us2 ngi, nge; };
// half-float bit pattern of a small integer (|n| <= 2048: exact)
__device__ inline uint32_t f16_bits_of_int(int n) {
  if (n == 0) return 0;
  const uint32_t sign = n < 0 ? 0x8000u : 0u, a = (uint32_t)(n < 0 ? -n : n);
  const int e = 31 - __clz((int)a);
  return sign | ((uint32_t)(e + 15) << 10) | ((e <= 10 ? a << (10 - e) : a >> (e - 10)) & 0x3ffu);
}
// ncols: columns of the register tile (longest read of this instance)
__device__ inline Sw16Par sw16_par(const MapPar &p, int ncols = 512) {
  Sw16Par s;
  // Half-float form (sw16f_core): scores as half floats -- integers up to 2048 are exact -- so that the substitution
  // score is signed (no bias to take off again) and gfx950's packed maximum3 floors the gap scores at 0 in the same
  // instruction.  One v_perm fetches the HIGH bytes of both tasks' scores: match and mismatch must be half floats with a
  // zero low byte (integers with at most three significant bits: 1, -2, 3, -4, 5, -6, ...).
  const uint32_t fm = f16_bits_of_int(p.match), fx = f16_bits_of_int(p.mismatch);
  s.fp = (fm & 0xffu) == 0 && (fx & 0xffu) == 0 && p.match > 0 && p.match * ncols <= 2040 && -p.gap_init >= 0 && -p.gap_init <= 2040 &&
         -p.gap_ext >= 0 && -p.gap_ext <= 2040 && p.mismatch >= -1024;
  s.hi_match = fm >> 8; s.hi_mismatch = fx >> 8;
  { const uint32_t a = f16_bits_of_int(p.gap_init), e = f16_bits_of_int(p.gap_ext);      // gap_init, gap_ext are negative: added
    s.ngi = us2{(unsigned short)a, (unsigned short)a}; s.nge = us2{(unsigned short)e, (unsigned short)e}; }
  const int bias = (p.mismatch < p.mismatch - p.match ? -p.mismatch : -(p.mismatch - p.match));
  s.mm4 = (uint32_t)((p.mismatch + bias) & 0xff) * 0x01010101u;      // a row of mismatches ...
  s.dlt = (uint32_t)(p.match - p.mismatch);                          // ... plus this at the byte of the matching base
  s.n4 = (uint32_t)(bias & 0xff) * 0x01010101u;                      // reference 'N': score 0 against everything
  s.bias = us2{(unsigned short)bias, (unsigned short)bias};
  s.gi = us2{(unsigned short)(-p.gap_init), (unsigned short)(-p.gap_init)};
  s.ge = us2{(unsigned short)(-p.gap_ext), (unsigned short)(-p.gap_ext)};
  return s;
}

// wrow: per-row code pairs (code of task A | code of task B << 3) of this lane group, entry r + G - 1 = row r; the
// entries before row 0 and from the last row up to nstep + G - 1 hold the N pair, so the sweep reads without bounds checks.
// rowtab2[pair]: the biased ACGT score rows of both codes (one 8-byte LDS read per row).
template <int G, int C>
__device__ inline uint32_t sw16_core(const uint16_t *wrow, int nstep, const uint32_t (&sel)[C], int g, const Sw16Par &sp, const uint2 *rowtab2) {
  us2 H[C], E[C];
#pragma unroll
  for (int cc = 0; cc < C; cc++) { H[cc] = us2{0, 0}; E[cc] = us2{0, 0}; }
  us2 best = us2{0, 0}, F = us2{0, 0}, prev_hl = us2{0, 0};
  const uint32_t gmask = g == 0 ? 0u : 0xffffffffu;       // the first lane of a group has no left neighbour
  for (int step = 0; step < nstep; step++) {
    const uint2 rr = rowtab2[wrow[step - g + (G - 1)]];
    const uint32_t rowA = rr.x, rowB = rr.y;                              // ACGT scores (+ bias) against this reference base
    const uint32_t hl = shr1_u32(as_u32(H[C - 1])) & gmask;
    const uint32_t fin = shr1_u32(as_u32(F)) & gmask;
    us2 carry = prev_hl;                       // H[row-1] of the column left of the block
    prev_hl = as_us2(hl);
    F = as_us2(fin);
    // Blocks of SW16_BLK columns: first everything that does not depend on the running F (independent across the
    // columns, so the packed-op forwarding hazards are filled with useful work), then the F chain.
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += SW16_BLK) {
      us2 t3[SW16_BLK];
      const us2 last_old = H[(c0 + SW16_BLK - 1 < C) ? c0 + SW16_BLK - 1 : C - 1];
#pragma unroll
      for (int u = 0; u < SW16_BLK; u++) {
        const int cc = c0 + u;
        if (cc < C) {
          const us2 w = as_us2(__builtin_amdgcn_perm(rowB, rowA, sel[cc]));
          const us2 dg = u == 0 ? carry : H[cc - 1];
          t3[u] = pk_subs(pk_add_nc(dg, w), sp.bias);
        }
      }
#pragma unroll
      for (int u = 0; u < SW16_BLK; u++) {
        const int cc = c0 + u;
        if (cc < C) {
          const us2 hh = pk_max3(t3[u], E[cc], F);
          H[cc] = hh;
          const us2 tt = pk_subs(hh, sp.gi);
          E[cc] = pk_max(pk_subs(E[cc], sp.ge), tt);
          F = pk_max(pk_subs(F, sp.ge), tt);
        }
      }
#pragma unroll
      for (int u = 0; u < SW16_BLK; u += 2) {            // the running maximum takes the block's cells two at a time
        const int cc = c0 + u;
        if (cc + 1 < C && u + 1 < SW16_BLK) best = pk_max3(best, H[cc], H[cc + 1]);
        else if (cc < C) best = pk_max(best, H[cc]);
      }
      carry = last_old;
    }
  }
  uint32_t bb = as_u32(best);
  for (int o = G / 2; o > 0; o >>= 1) bb = as_u32(pk_max(as_us2(bb), as_us2((uint32_t)__shfl_xor((int)bb, o))));
  return bb;
}

// The same sweep in half floats (Sw16Par::fp).  Per cell pair: perm (high bytes of both signed scores), add, max3 (value,
// E, F -- both kept >= 0, which floors the value too), add (gap open), 2 x (add, max3 with 0) and 0.6 for the running
// maximum: 8.6 instructions.  Every value is an integer of magnitude <= 2048, so the arithmetic is exact.
__device__ inline hf2 hf_max3(hf2 a, hf2 b, hf2 c) { return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c); }
template <int G, int C>
__device__ inline uint32_t sw16f_core(const uint16_t *wrow, int nstep, const uint32_t (&sel)[C], int g, const Sw16Par &sp, const uint2 *rowtab2) {
  hf2 H[C], E[C];
  const hf2 zero = hf2{(_Float16)0, (_Float16)0};
#pragma unroll
  for (int cc = 0; cc < C; cc++) { H[cc] = zero; E[cc] = zero; }
  hf2 best = zero, F = zero, prev_hl = zero;
  const hf2 ngi = __builtin_bit_cast(hf2, sp.ngi), nge = __builtin_bit_cast(hf2, sp.nge);
  const uint32_t gmask = g == 0 ? 0u : 0xffffffffu;
  for (int step = 0; step < nstep; step++) {
    const uint2 rr = rowtab2[wrow[step - g + (G - 1)]];
    const uint32_t hl = shr1_u32(__builtin_bit_cast(uint32_t, H[C - 1])) & gmask;
    const uint32_t fin = shr1_u32(__builtin_bit_cast(uint32_t, F)) & gmask;
    hf2 carry = prev_hl;
    prev_hl = __builtin_bit_cast(hf2, hl);
    F = __builtin_bit_cast(hf2, fin);
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += SW16_BLK) {
      hf2 t3[SW16_BLK];
      const hf2 last_old = H[(c0 + SW16_BLK - 1 < C) ? c0 + SW16_BLK - 1 : C - 1];
#pragma unroll
      for (int u = 0; u < SW16_BLK; u++) {
        const int cc = c0 + u;
        if (cc < C) {
          const hf2 w = __builtin_bit_cast(hf2, __builtin_amdgcn_perm(rr.y, rr.x, sel[cc]));
          const hf2 dg = u == 0 ? carry : H[cc - 1];
          t3[u] = dg + w;
        }
      }
#pragma unroll
      for (int u = 0; u < SW16_BLK; u++) {
        const int cc = c0 + u;
        if (cc < C) {
          const hf2 hh = hf_max3(t3[u], E[cc], F);
          H[cc] = hh;
          const hf2 tt = hh + ngi;
          E[cc] = hf_max3(E[cc] + nge, tt, zero);
          F = hf_max3(F + nge, tt, zero);
        }
      }
#pragma unroll
      for (int u = 0; u < SW16_BLK; u += 2) {
        const int cc = c0 + u;
        if (cc + 1 < C && u + 1 < SW16_BLK) best = hf_max3(best, H[cc], H[cc + 1]);
        else if (cc < C) best = __builtin_elementwise_maximum(best, H[cc]);
      }
      carry = last_old;
    }
  }
  for (int o = G / 2; o > 0; o >>= 1)
    best = __builtin_elementwise_maximum(best, __builtin_bit_cast(hf2, (uint32_t)__shfl_xor((int)__builtin_bit_cast(uint32_t, best), o)));
  return (uint32_t)(int)(float)best.x | ((uint32_t)(int)(float)best.y << 16);
}

// rowtab[code]: the four biased ACGT scores against reference code 0..7 (4, 5, 6: 'N' -> score 0; 7 decodes as A upstream)
__device__ inline void sw16_rowtab(uint32_t *rowtab, const Sw16Par &sp) {
  if (threadIdx.x < 8) rowtab[threadIdx.x] = threadIdx.x < 4 ? sp.mm4 + (sp.dlt << (8 * threadIdx.x)) : sp.n4;
}
// the same for a pair of codes: entry a | b << 3
__device__ inline void sw16_rowtab2(uint2 *rowtab2, const Sw16Par &sp) {
  const uint32_t a = threadIdx.x & 7u, bq = threadIdx.x >> 3;
  if (sp.fp) {                                  // high bytes of the half-float scores against A, C, G, T; N rows: 0
    const uint32_t x4 = sp.hi_mismatch * 0x01010101u;
    rowtab2[threadIdx.x] = make_uint2(a < 4 ? (x4 & ~(0xffu << (8 * a))) | (sp.hi_match << (8 * a)) : 0u,
                                      bq < 4 ? (x4 & ~(0xffu << (8 * bq))) | (sp.hi_match << (8 * bq)) : 0u);
    return;
  }
  rowtab2[threadIdx.x] = make_uint2(a < 4 ? sp.mm4 + (sp.dlt << (8 * a)) : sp.n4, bq < 4 ? sp.mm4 + (sp.dlt << (8 * bq)) : sp.n4);
}
enum : uint32_t { SW16_NPAIR = 5u | (5u << 3) };

template <int G, int C, int WMAX>
__global__ void __launch_bounds__(64) k_sw_full16(Batch b, DevIndex ix, MapPar p, uint32_t ntask_cap) {
  // the small-LDS instance (WMAX = SW_SHORT_WMAX) runs first; the large one only sees what is left
  // (through the list S7 made of them; if the list overflowed, by scanning all candidates)
  const unsigned long long nlong = WMAX > SW_SHORT_WMAX ? b.work[WK_LONG_TASKS] : 0;
  if (WMAX > SW_SHORT_WMAX && nlong == 0) return;
  const uint32_t *list = (WMAX > SW_SHORT_WMAX && b.long_list && nlong <= b.long_cap) ? b.long_list : nullptr;
  constexpr int NG = 64 / G;
  __shared__ uint16_t win[NG][WMAX + 2 * G];
  __shared__ uint2 rowtab2[64];
  const int lane = threadIdx.x, g = lane % G, grp = lane / G;
  const uint32_t ntask = list ? (uint32_t)nlong : min(*b.rc_count, ntask_cap), npair = (ntask + 1) / 2;
  const Sw16Par sp = sw16_par(p, G * C);
  sw16_rowtab2(rowtab2, sp);
  const uint32_t ngroups = gridDim.x * NG;
  unsigned long long cells = 0, ntasks_done = 0;
  for (uint32_t t0 = blockIdx.x * NG; t0 < npair; t0 += ngroups) {
    const uint32_t tp = t0 + grp;
    RCand c[2];
    bool live[2] = {false, false};
    uint32_t qlen[2] = {0, 0}, wlen[2] = {0, 0}, tix[2] = {0, 0};
    uint64_t gbase[2] = {0, 0};
    const uint8_t *q[2] = {b.codes, b.codes};
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const uint32_t tl = 2 * tp + (uint32_t)u;
      if (tp < npair && tl < ntask) {
        tix[u] = list ? list[tl] : tl;
        c[u] = b.rcpool[tix[u]];
        qlen[u] = read_len(b, c[u].rid);
        wlen[u] = (uint32_t)(c[u].re - c[u].rs + 1);
        live[u] = !(c[u].flags & (RCF_BANDED | RCF_ERR | RCF_QN | RCF_SCORED)) && wlen[u] <= (uint32_t)WMAX && qlen[u] <= (uint32_t)(G * C);
        gbase[u] = (c[u].sqidx < 0 ? 0ull : ix.sop[c[u].sqidx]) + c[u].rs;
        q[u] = ((c[u].flags & RCF_REVERSE) ? b.codes_rc : b.codes) + b.read_off[c[u].rid];
      }
      if (!live[u]) { qlen[u] = 0; wlen[u] = 0; }
    }
    const uint32_t wmax = wlen[0] > wlen[1] ? wlen[0] : wlen[1];
    int nstep = (int)wmax + G - 1;
    for (int o = 32; o > 0; o >>= 1) nstep = max(nstep, __shfl_xor(nstep, o));
    nstep = __builtin_amdgcn_readfirstlane(nstep);                         // wave-uniform: a scalar loop bound for the sweep
    {
      // entry e = row e - (G - 1); N pairs around the window.  Every lane decodes a run of consecutive entries: one
      // division per task to find the first packed word (10 bases per word), then shifts; the next word is loaded ahead.
      const uint32_t nent = (uint32_t)(nstep + G - 1), per = (nent + G - 1) / G;
      const uint32_t e0 = (uint32_t)g * per, e1 = e0 + per < nent ? e0 + per : nent;
      const uint64_t lastw = ix.totlen / 10;
      uint64_t wi[2] = {0, 0};
      uint32_t wd[2] = {0, 0}, nx[2] = {0, 0}, off[2] = {0, 0};
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int i0 = (int)e0 - (G - 1);
        const uint64_t pos = gbase[u] + (uint64_t)(i0 > 0 ? i0 : 0);
        wi[u] = pos / 10; off[u] = (uint32_t)(pos - wi[u] * 10);
        if (wlen[u] && e0 < e1) { wd[u] = ix.packed[wi[u] < lastw ? wi[u] : lastw]; nx[u] = ix.packed[wi[u] + 1 < lastw ? wi[u] + 1 : lastw]; }
      }
      for (uint32_t e = e0; e < e1; e++) {
        const uint32_t i = e - (uint32_t)(G - 1);                         // (wraps for the leading pad: fails both tests below)
        uint32_t cd[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          cd[u] = 5u;
          if (i < wlen[u]) {
            const uint32_t c = (wd[u] >> (3 * (9 - off[u]))) & 7u;
            cd[u] = (c == 7) ? 0u : ((c == 6 || c == 4) ? 5u : c);
            if (++off[u] == 10) { off[u] = 0; wi[u]++; wd[u] = nx[u]; nx[u] = ix.packed[wi[u] + 1 < lastw ? wi[u] + 1 : lastw]; }
          }
        }
        win[grp][e] = (uint16_t)(cd[0] | (cd[1] << 3));
      }
    }
    uint32_t sel[C];
#pragma unroll
    for (int cc = 0; cc < C; cc++) {
      const uint32_t j = (uint32_t)(g * C + cc);
      const uint32_t sa = j < qlen[0] ? (uint32_t)q[0][j] : 0x0cu, sb = j < qlen[1] ? 4u + (uint32_t)q[1][j] : 0x0cu;
      sel[cc] = sp.fp ? (0x000c000cu | (sa << 8) | (sb << 24)) : (0x0c000c00u | sa | (sb << 16));      // half floats: the score byte is the high byte
    }
    __syncthreads();
    const uint32_t bb = sp.fp ? sw16f_core<G, C>(win[grp], nstep, sel, g, sp, rowtab2) : sw16_core<G, C>(win[grp], nstep, sel, g, sp, rowtab2);
    if (g == 0) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (live[u]) {
          const int best = (int)((bb >> (16 * u)) & 0xffffu);
          const uint32_t t = tix[u];
          b.rcpool[t].swscor = best;
          b.rcpool[t].flags = c[u].flags | RCF_SCORED | (best >= 65535 ? RCF_BANDED : 0u);   // ERRCODE_SWATEXCEED -> K2b
          cells += (unsigned long long)qlen[u] * wlen[u];
          ntasks_done += best < 65535 ? 1 : 0;      /* a score that left 16 bits waits for K2b */
        }
      }
    }
    __syncthreads();
  }
  for (int o = 32; o > 0; o >>= 1) { cells += __shfl_xor(cells, o); ntasks_done += __shfl_xor(ntasks_done, o); }
  if (lane == 0 && cells) { atomicAdd(b.work + WK_CELLS_FULL, cells); atomicAdd(b.work + WK_TASKS_FULL, ntasks_done); }
}

// stand-alone form over explicit code arrays (tasks with non-ACGT query codes report -2: not handled here)
template <int G, int C>
__global__ void __launch_bounds__(64) k_sw_full16_raw(const uint8_t *qcodes, const uint32_t *q_off, const uint8_t *rcodes,
                                                       const uint32_t *r_off, uint32_t ntask, MapPar p, int32_t *scores) {
  constexpr int NG = 64 / G;
  constexpr int WMAX = SW_FULL_WMAX;
  __shared__ uint16_t win[NG][WMAX + 2 * G];
  __shared__ uint2 rowtab2[64];
  const int lane = threadIdx.x, g = lane % G, grp = lane / G;
  const uint32_t npair = (ntask + 1) / 2;
  const Sw16Par sp = sw16_par(p, G * C);
  sw16_rowtab2(rowtab2, sp);
  const uint32_t ngroups = gridDim.x * NG;
  for (uint32_t t0 = blockIdx.x * NG; t0 < npair; t0 += ngroups) {
    const uint32_t tp = t0 + grp;
    bool live[2] = {false, false};
    uint32_t qlen[2] = {0, 0}, wlen[2] = {0, 0};
    const uint8_t *q[2] = {qcodes, qcodes}, *r[2] = {rcodes, rcodes};
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const uint32_t t = 2 * tp + (uint32_t)u;
      if (tp < npair && t < ntask) {
        qlen[u] = q_off[t + 1] - q_off[t]; wlen[u] = r_off[t + 1] - r_off[t];
        q[u] += q_off[t]; r[u] += r_off[t];
        live[u] = wlen[u] <= (uint32_t)WMAX && qlen[u] <= (uint32_t)(G * C);
      }
      if (!live[u]) { qlen[u] = 0; wlen[u] = 0; }
    }
    const uint32_t wmax = wlen[0] > wlen[1] ? wlen[0] : wlen[1];
    int nstep = (int)wmax + G - 1;
    for (int o = 32; o > 0; o >>= 1) nstep = max(nstep, __shfl_xor(nstep, o));
    nstep = __builtin_amdgcn_readfirstlane(nstep);                         // wave-uniform: a scalar loop bound for the sweep
    for (uint32_t e = g; e < (uint32_t)(nstep + G - 1); e += G) {
      const uint32_t i = e - (uint32_t)(G - 1);
      uint32_t cd[2];
#pragma unroll
      for (int u = 0; u < 2; u++) { const uint32_t x = i < wlen[u] ? (r[u][i] & 7u) : 5u; cd[u] = x == 7 ? 0 : ((x == 6 || x == 4) ? 5 : x); }
      win[grp][e] = (uint16_t)(cd[0] | (cd[1] << 3));
    }
    uint32_t sel[C];
    bool qn[2] = {false, false};
#pragma unroll
    for (int cc = 0; cc < C; cc++) {
      const uint32_t j = (uint32_t)(g * C + cc);
      uint32_t s2[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const uint32_t qc = j < qlen[u] ? (q[u][j] & 7u) : 0x0cu;
        if (j < qlen[u] && qc >= 4) qn[u] = true;
        s2[u] = qc == 0x0cu ? 0x0cu : (uint32_t)(4 * u) + (qc & 3u);
      }
      sel[cc] = sp.fp ? (0x000c000cu | (s2[0] << 8) | (s2[1] << 24)) : (0x0c000c00u | s2[0] | (s2[1] << 16));
    }
#pragma unroll
    for (int u = 0; u < 2; u++) for (int o = G / 2; o > 0; o >>= 1) { const int other = __shfl_xor((int)qn[u], o); qn[u] = qn[u] || other != 0; }
    __syncthreads();
    const uint32_t bb = sp.fp ? sw16f_core<G, C>(win[grp], nstep, sel, g, sp, rowtab2) : sw16_core<G, C>(win[grp], nstep, sel, g, sp, rowtab2);
    if (g == 0) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const uint32_t t = 2 * tp + (uint32_t)u;
        if (tp < npair && t < ntask) scores[t] = !live[u] ? -1 : (qn[u] ? -2 : (int)((bb >> (16 * u)) & 0xffffu));
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// K2a for reads or windows beyond the register tiling (long reads): one wave per task.  The read is cut
// into strips of 64 x SW_STRIP_C columns (lane g owns SW_STRIP_C consecutive columns of the strip) and the
// window is swept once per strip, lane g one row behind lane g-1 (DPP wave_shr:1 hands over H of the last
// column and the running F).  H and F of a strip's last column go, row by row, through an LDS ring into a
// boundary buffer in HBM that feeds lane 0 of the next strip (read back 64 rows at a time, ahead of use).
// 32-bit lanes, full 8-byte score rows: any code (N) and any score below 2^31.
// ---------------------------------------------------------------------------------------
enum : int { SW_STRIP_C = 16 };

// win: window codes (HBM, wlen bytes), bnd: 2 x wcap (H, F) pairs of this workgroup
__device__ inline int sw_strip_core(const uint8_t *q, uint32_t qlen, const uint8_t *win, uint32_t wlen, int2 *bnd, uint32_t wcap,
                                    const uint2 *tab, int2 *ring /* LDS [256] */, int bias, int gi, int ge) {
  constexpr int C = SW_STRIP_C;
  const int g = (int)threadIdx.x;
  int2 *ring_in = ring, *ring_out = ring + 128;
  int best = 0;
  const uint32_t nstrip = (qlen + 64 * C - 1) / (64 * C);
  const int nstep = (int)wlen + 63;
  for (uint32_t sidx = 0; sidx < nstrip; sidx++) {
    const int2 *bprev = bnd + (size_t)((sidx + 1) & 1) * wcap;
    int2 *bnext = bnd + (size_t)(sidx & 1) * wcap;
    uint32_t sel[C];
#pragma unroll
    for (int cc = 0; cc < C; cc++) {
      const uint32_t j = sidx * 64 * C + (uint32_t)(g * C + cc);
      sel[cc] = 0x0c0c0c00u | (j < qlen ? (uint32_t)(q[j] & 7) : 5u);
    }
    int H[C], E[C];
#pragma unroll
    for (int cc = 0; cc < C; cc++) { H[cc] = 0; E[cc] = 0; }
    int F = 0, prev_hl = 0;
    __syncthreads();
    if (sidx > 0) {                                  // first 64 boundary rows of the previous strip
      const int r0 = g;
      ring_in[g] = r0 < (int)wlen ? bprev[r0] : make_int2(0, 0);
    }
    __syncthreads();
    for (int step = 0; step < nstep; step++) {
      const int row = step - g;
      if (sidx > 0 && (step & 63) == 0) {            // read ahead: boundary rows step+64 .. step+127
        const int r1 = step + 64 + g;
        ring_in[((step >> 6) + 1) % 2 * 64 + g] = r1 < (int)wlen ? bprev[r1] : make_int2(0, 0);
      }
      const uint32_t rb = (row >= 0 && row < (int)wlen) ? win[row] : 5u;
      const uint2 tr = tab[rb];
      int hl = wave_shr1(H[C - 1]);
      int fin = wave_shr1(F);
      if (g == 0) {
        if (sidx > 0) { const int2 v = ring_in[(step >> 6) % 2 * 64 + (step & 63)]; hl = step < (int)wlen ? v.x : 0; fin = step < (int)wlen ? v.y : 0; }
        else { hl = 0; fin = 0; }
      }
      int diag = prev_hl;
      prev_hl = hl;
      F = fin;
#pragma unroll
      for (int cc = 0; cc < C; cc++) {
        const int w = (int)__builtin_amdgcn_perm(tr.y, tr.x, sel[cc]);
        const int h = diag + w - bias;
        const int hh = max(max(h, E[cc]), F);
        best = max(best, hh);
        diag = H[cc];
        H[cc] = hh;
        const int tt = hh - gi;
        E[cc] = max(max(E[cc] - ge, tt), 0);
        F = max(max(F - ge, tt), 0);
      }
      if (sidx + 1 < nstrip) {                       // hand the last column to the next strip
        if (g == 63 && row >= 0 && row < (int)wlen) ring_out[row & 127] = make_int2(H[C - 1], F);
        const int rdone = step - 63;                 // row lane 63 has just finished
        if (rdone >= 0 && ((rdone & 63) == 63 || rdone == (int)wlen - 1)) {
          __syncthreads();
          const int base = rdone & ~63, r2 = base + g;
          if (r2 <= rdone && r2 < (int)wlen) bnext[r2] = ring_out[r2 & 127];
        }
      }
      if (sidx > 0 && (step & 63) == 63) __syncthreads();   // the read-ahead chunk is in place before lane 0 turns to it
    }
    __threadfence();
  }
  for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
  return best;
}

__device__ inline void sw_tab8(uint2 *tab, const MapPar &p, int bias) {
  if (threadIdx.x < 8) {                     // biased score matrix rows (score.c:138-173; rows 4,6 = N, 7 = A)
    const int lane = (int)threadIdx.x;
    const int rb = lane == 7 ? 0 : ((lane == 6 || lane == 4) ? 5 : lane);
    uint32_t w[2] = {0, 0};
    for (int qc = 0; qc < 8; qc++) {
      const int v = (rb == 5 || qc >= 4) ? 0 : ((rb == qc) ? p.match : p.mismatch);
      w[qc >> 2] |= (uint32_t)((v + bias) & 0xff) << (8 * (qc & 3));
    }
    tab[lane] = make_uint2(w[0], w[1]);
  }
}

// The same in packed 16-bit halves: two tasks per wave (as k_sw_full16; reads without non-ACGT codes, scores < 65535).
// win2: per-row code pairs (low byte task A, high byte task B); bnd: 2 x wcap packed (H pair, F pair).
// M3: every value stays below 0x7C00, so the packed maximum of three (pk_max3) applies
template <bool M3>
__device__ inline uint32_t sw_strip16_core(const uint8_t *qa, uint32_t qlen_a, const uint8_t *qb, uint32_t qlen_b, const uint16_t *win2, uint32_t wmax,
                                           uint2 *bnd, uint32_t wcap, const uint32_t *rowtab, uint2 *ring /* LDS [256] */, const Sw16Par &sp) {
  constexpr int C = SW_STRIP_C;
  const int g = (int)threadIdx.x;
  uint2 *ring_in = ring, *ring_out = ring + 128;
  us2 best = us2{0, 0};
  const uint32_t qmaxlen = qlen_a > qlen_b ? qlen_a : qlen_b;
  const uint32_t nstrip = (qmaxlen + 64 * C - 1) / (64 * C);
  const int nstep = (int)wmax + 63;
  for (uint32_t sidx = 0; sidx < nstrip; sidx++) {
    const uint2 *bprev = bnd + (size_t)((sidx + 1) & 1) * wcap;
    uint2 *bnext = bnd + (size_t)(sidx & 1) * wcap;
    uint32_t sel[C];
#pragma unroll
    for (int cc = 0; cc < C; cc++) {
      const uint32_t j = sidx * 64 * C + (uint32_t)(g * C + cc);
      const uint32_t sa = j < qlen_a ? (uint32_t)(qa[j] & 3) : 0x0cu, sb = j < qlen_b ? 4u + (uint32_t)(qb[j] & 3) : 0x0cu;
      sel[cc] = 0x0c000c00u | sa | (sb << 16);
    }
    us2 H[C], E[C];
#pragma unroll
    for (int cc = 0; cc < C; cc++) { H[cc] = us2{0, 0}; E[cc] = us2{0, 0}; }
    us2 F = us2{0, 0}, prev_hl = us2{0, 0};
    __syncthreads();
    if (sidx > 0) ring_in[g] = g < (int)wmax ? bprev[g] : make_uint2(0, 0);
    __syncthreads();
    for (int step = 0; step < nstep; step++) {
      const int row = step - g;
      if (sidx > 0 && (step & 63) == 0) {
        const int r1 = step + 64 + g;
        ring_in[((step >> 6) + 1) % 2 * 64 + g] = r1 < (int)wmax ? bprev[r1] : make_uint2(0, 0);
      }
      const uint32_t rp = (row >= 0 && row < (int)wmax) ? win2[row] : 0x0505u;
      const uint32_t rowA = rowtab[rp & 0xffu], rowB = rowtab[rp >> 8];
      uint32_t hl = (uint32_t)wave_shr1((int)as_u32(H[C - 1]));
      uint32_t fin = (uint32_t)wave_shr1((int)as_u32(F));
      if (g == 0) {
        if (sidx > 0) { const uint2 v = ring_in[(step >> 6) % 2 * 64 + (step & 63)]; hl = step < (int)wmax ? v.x : 0u; fin = step < (int)wmax ? v.y : 0u; }
        else { hl = 0; fin = 0; }
      }
      us2 carry = prev_hl;
      prev_hl = as_us2(hl);
      F = as_us2(fin);
#pragma unroll
      for (int c0 = 0; c0 < C; c0 += 4) {
        us2 t3[4];
        const us2 last_old = H[c0 + 3];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const us2 w = as_us2(__builtin_amdgcn_perm(rowB, rowA, sel[c0 + u]));
          const us2 dg = u == 0 ? carry : H[c0 + u - 1];
          t3[u] = pk_subs(pk_add_nc(dg, w), sp.bias);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const us2 hh = M3 ? pk_max3(t3[u], E[c0 + u], F) : pk_max(pk_max(t3[u], E[c0 + u]), F);
          if (!M3) best = pk_max(best, hh);
          H[c0 + u] = hh;
          const us2 tt = pk_subs(hh, sp.gi);
          E[c0 + u] = pk_max(pk_subs(E[c0 + u], sp.ge), tt);
          F = pk_max(pk_subs(F, sp.ge), tt);
        }
        if (M3) { best = pk_max3(best, H[c0], H[c0 + 1]); best = pk_max3(best, H[c0 + 2], H[c0 + 3]); }
        carry = last_old;
      }
      if (sidx + 1 < nstrip) {
        if (g == 63 && row >= 0 && row < (int)wmax) ring_out[row & 127] = make_uint2(as_u32(H[C - 1]), as_u32(F));
        const int rdone = step - 63;
        if (rdone >= 0 && ((rdone & 63) == 63 || rdone == (int)wmax - 1)) {
          __syncthreads();
          const int r2 = (rdone & ~63) + g;
          if (r2 <= rdone && r2 < (int)wmax) bnext[r2] = ring_out[r2 & 127];
        }
      }
      if (sidx > 0 && (step & 63) == 63) __syncthreads();
    }
    __threadfence();
  }
  uint32_t bb = as_u32(best);
  for (int o = 32; o > 0; o >>= 1) bb = as_u32(pk_max(as_us2(bb), as_us2((uint32_t)__shfl_xor((int)bb, o))));
  return bb;
}

// packed strip kernel over the strip list: two consecutive list entries per wave; reads with non-ACGT codes are left
// to k_sw_strip (32-bit lanes), which runs afterwards on whatever is still unscored
template <bool M3>
__global__ void __launch_bounds__(64) k_sw_strip16(Batch b, DevIndex ix, MapPar p, uint2 *bnd_all, uint16_t *win_all, uint32_t wcap) {
  const unsigned long long nlist = b.work[WK_STRIP_TASKS];
  if (nlist == 0 || !b.strip_list || nlist > b.strip_cap) return;
  __shared__ uint32_t rowtab[8];
  __shared__ uint2 ring[256];
  const Sw16Par sp = sw16_par(p);
  sw16_rowtab(rowtab, sp);
  __syncthreads();
  const uint32_t npair = ((uint32_t)nlist + 1) / 2;
  uint2 *bnd = bnd_all + (size_t)blockIdx.x * 2 * wcap;
  uint16_t *win = win_all + (size_t)blockIdx.x * wcap;
  unsigned long long cells = 0, ntasks_done = 0;
  __shared__ uint32_t qslot;
  // pairs from a cursor, not by a static stride: windows of 8 kbp reads differ in length, and the launch has more workgroups
  // than are resident at a time
  for (uint32_t tp = next_item(b.strip_cursor, &qslot); tp < npair; tp = next_item(b.strip_cursor, &qslot)) {
    RCand c[2];
    bool live[2] = {false, false};
    uint32_t qlen[2] = {0, 0}, wlen[2] = {0, 0}, tix[2] = {0, 0};
    uint64_t gbase[2] = {0, 0};
    const uint8_t *q[2] = {b.codes, b.codes};
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const uint32_t tl = 2 * tp + (uint32_t)u;
      if (tl < (uint32_t)nlist) {
        tix[u] = b.strip_list[tl];
        c[u] = b.rcpool[tix[u]];
        qlen[u] = read_len(b, c[u].rid);
        wlen[u] = (uint32_t)(c[u].re - c[u].rs + 1);
        live[u] = !(c[u].flags & (RCF_BANDED | RCF_ERR | RCF_QN | RCF_SCORED)) && wlen[u] <= wcap;
        gbase[u] = (c[u].sqidx < 0 ? 0ull : ix.sop[c[u].sqidx]) + c[u].rs;
        q[u] = ((c[u].flags & RCF_REVERSE) ? b.codes_rc : b.codes) + b.read_off[c[u].rid];
      }
      if (!live[u]) { qlen[u] = 0; wlen[u] = 0; }
    }
    const uint32_t wmax = wlen[0] > wlen[1] ? wlen[0] : wlen[1];
    if (wmax == 0) continue;                                   // wave-uniform
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < wmax; i += 64) {
      const uint32_t a = i < wlen[0] ? ref_code(ix.packed, gbase[0] + i) : 5u, bb = i < wlen[1] ? ref_code(ix.packed, gbase[1] + i) : 5u;
      win[i] = (uint16_t)(a | (bb << 8));
    }
    __threadfence();
    __syncthreads();
    const uint32_t bb = sw_strip16_core<M3>(q[0], qlen[0], q[1], qlen[1], win, wmax, bnd, wcap, rowtab, ring, sp);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (live[u]) {
          const int best = (int)((bb >> (16 * u)) & 0xffffu);
          b.rcpool[tix[u]].swscor = best;
          b.rcpool[tix[u]].flags = c[u].flags | RCF_SCORED | (best >= 65535 ? RCF_BANDED : 0u);   // ERRCODE_SWATEXCEED -> K2b
          cells += (unsigned long long)qlen[u] * wlen[u];
          ntasks_done += best < 65535 ? 1 : 0;      /* a score that left 16 bits waits for K2b */
        }
      }
    }
  }
  if (threadIdx.x == 0 && cells) { atomicAdd(b.work + WK_CELLS_FULL, cells); atomicAdd(b.work + WK_TASKS_FULL, ntasks_done); }
}

// tasks: the strip list S7 made (ranked candidates whose read or window exceeds the register tiling)
__global__ void __launch_bounds__(64) k_sw_strip(Batch b, DevIndex ix, MapPar p, int2 *bnd_all, uint8_t *win_all, uint32_t wcap) {
  const unsigned long long nlist = b.work[WK_STRIP_TASKS];
  if (nlist == 0) return;
  __shared__ uint2 tab[8];
  __shared__ int2 ring[256];
  const int bias = (p.mismatch < p.mismatch - p.match ? -p.mismatch : -(p.mismatch - p.match));
  sw_tab8(tab, p, bias);
  __syncthreads();
  const bool listed = b.strip_list && nlist <= b.strip_cap;
  const uint32_t ntask = listed ? (uint32_t)nlist : min(*b.rc_count, b.rccap);
  int2 *bnd = bnd_all + (size_t)blockIdx.x * 2 * wcap;
  uint8_t *win = win_all + (size_t)blockIdx.x * wcap;
  unsigned long long cells = 0, ntasks_done = 0;
  for (uint32_t tl = blockIdx.x; tl < ntask; tl += gridDim.x) {
    const uint32_t t = listed ? b.strip_list[tl] : tl;
    const RCand c = b.rcpool[t];
    const uint32_t qlen = read_len(b, c.rid), wlen = (uint32_t)(c.re - c.rs + 1);
    if ((c.flags & (RCF_BANDED | RCF_ERR | RCF_SCORED)) || wlen > wcap) continue;     // wave-uniform
    const uint64_t gbase = (c.sqidx < 0 ? 0ull : ix.sop[c.sqidx]) + c.rs;
    const uint8_t *q = ((c.flags & RCF_REVERSE) ? b.codes_rc : b.codes) + b.read_off[c.rid];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < wlen; i += 64) win[i] = (uint8_t)ref_code(ix.packed, gbase + i);
    __threadfence();
    __syncthreads();
    const int best = sw_strip_core(q, qlen, win, wlen, bnd, wcap, tab, ring, bias, -p.gap_init, -p.gap_ext);
    if (threadIdx.x == 0) {
      b.rcpool[t].swscor = best;
      b.rcpool[t].flags = c.flags | RCF_SCORED | (best >= 65535 ? RCF_BANDED : 0u);   // ERRCODE_SWATEXCEED -> K2b
      cells += (unsigned long long)qlen * wlen;
      ntasks_done += best < 65535 ? 1 : 0;      /* a score that left 16 bits waits for K2b */
    }
  }
  if (threadIdx.x == 0 && cells) { atomicAdd(b.work + WK_CELLS_FULL, cells); atomicAdd(b.work + WK_TASKS_FULL, ntasks_done); }
}

// stand-alone form over explicit code arrays (parity tests)
__global__ void __launch_bounds__(64) k_sw_strip_raw(const uint8_t *qcodes, const uint32_t *q_off, const uint8_t *rcodes, const uint32_t *r_off,
                                                     uint32_t ntask, MapPar p, int32_t *scores, int2 *bnd_all, uint8_t *win_all, uint32_t wcap) {
  __shared__ uint2 tab[8];
  __shared__ int2 ring[256];
  const int bias = (p.mismatch < p.mismatch - p.match ? -p.mismatch : -(p.mismatch - p.match));
  sw_tab8(tab, p, bias);
  __syncthreads();
  int2 *bnd = bnd_all + (size_t)blockIdx.x * 2 * wcap;
  uint8_t *win = win_all + (size_t)blockIdx.x * wcap;
  for (uint32_t t = blockIdx.x; t < ntask; t += gridDim.x) {
    const uint32_t qlen = q_off[t + 1] - q_off[t], wlen = r_off[t + 1] - r_off[t];
    if (wlen > wcap) { if (threadIdx.x == 0) scores[t] = -1; continue; }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < wlen; i += 64) { const uint32_t cd = rcodes[r_off[t] + i] & 7; win[i] = (uint8_t)(cd == 7 ? 0 : (cd == 6 || cd == 4) ? 5 : cd); }
    __threadfence();
    __syncthreads();
    const int best = sw_strip_core(qcodes + q_off[t], qlen, win, wlen, bnd, wcap, tab, ring, bias, -p.gap_init, -p.gap_ext);
    if (threadIdx.x == 0) scores[t] = best;
  }
}

// ---------------------------------------------------------------------------------------
// K2b (alignSmiWatBandFast, alignment.c:1029-1233) for long reads: one wave per task, the strip scheme of
// sw_strip_core with the restricted cell update and the band as an activity mask.  Row r of the band visits the
// columns [js(r), jl(r)): js(r) = q_left for ever when the band starts clipped at the read's left end (the
// reference never advances it then), l_edge + r otherwise; jl(r) = min(r_edge + 1 + r, q_len).  Cells outside keep
// H and E (the reference's row buffers persist) and hand F = 0 to the right; every value a visited cell reads
// from an unvisited neighbour is then what the reference reads (0 for a cell never visited, else the stale
// buffer value, which only cells on the same diagonal -- visited or never visited together -- can reach).
// A strip sweeps only the rows in which it has visited cells.
// ---------------------------------------------------------------------------------------
__device__ inline int band_fast_wave(const Band &bp, const uint8_t *q, const uint8_t *win /* row i of the window */, int2 *bnd, uint32_t wcap,
                                     const uint2 *tab, int2 *ring /* LDS [256] */, int bias, int gi, int ge) {
  constexpr int C = SW_STRIP_C;
  const int g = (int)threadIdx.x;
  const bool clipped = bp.q_left > bp.l_edge;
  const int j0 = clipped ? bp.q_left : bp.l_edge;
  const int nrows = bp.s_len - bp.s_left;
  if (nrows <= 0) return 0;
  const int jmax = min(bp.r_edge + nrows, bp.q_len);          // jl(nrows - 1)
  if (jmax <= j0 || (uint32_t)nrows > wcap) return 0;
  const int nstrip = (jmax - j0 + 64 * C - 1) / (64 * C);
  int2 *ring_in = ring, *ring_out = ring + 128;
  int best = 0, plo = 0, phi = 0;                             // [plo, phi): rows the previous strip swept
  for (int sidx = 0; sidx < nstrip; sidx++) {
    const int c_lo = j0 + sidx * 64 * C, c_hi = c_lo + 64 * C;
    const int r_lo = max(c_lo - bp.r_edge, 0);
    const int r_hi = clipped ? nrows : min(nrows, c_hi - bp.l_edge);
    const int nr = r_hi - r_lo;
    const int2 *bprev = bnd + (size_t)((sidx + 1) & 1) * wcap;
    int2 *bnext = bnd + (size_t)(sidx & 1) * wcap;
    const int jb = c_lo + g * C;
    uint32_t sel[C];
#pragma unroll
    for (int cc = 0; cc < C; cc++) sel[cc] = 0x0c0c0c00u | (jb + cc < bp.q_len ? (uint32_t)(q[jb + cc] & 7) : 5u);
    int H[C], E[C];
#pragma unroll
    for (int cc = 0; cc < C; cc++) { H[cc] = 0; E[cc] = 0; }
    int F = 0, prev_hl = 0;
#define SMG_BGET(r) ((sidx > 0 && (r) >= plo && (r) < phi) ? bprev[(r)] : make_int2(0, 0))
    __syncthreads();
    if (sidx > 0) {
      ring_in[g] = SMG_BGET(r_lo + g);
      if (g == 0) prev_hl = SMG_BGET(r_lo - 1).x;
    }
    __syncthreads();
    const int nstep = nr + 63;
    for (int step = 0; step < nstep; step++) {
      const int rel = step - g, row = r_lo + rel;
      if (sidx > 0 && (step & 63) == 0) ring_in[((step >> 6) + 1) % 2 * 64 + g] = SMG_BGET(r_lo + step + 64 + g);
      const bool rowok = rel >= 0 && rel < nr;
      const uint32_t rb = rowok ? win[bp.s_left + row] : 5u;
      const uint2 tr = tab[rb];
      int hl = wave_shr1(H[C - 1]);
      int fin = wave_shr1(F);
      if (g == 0) {
        if (sidx > 0) { const int2 v = ring_in[(step >> 6) % 2 * 64 + (step & 63)]; hl = v.x; fin = v.y; }
        else { hl = 0; fin = 0; }
      }
      int diag = prev_hl;
      prev_hl = hl;
      F = fin;
      if (rowok) {
        const int jsr = clipped ? bp.q_left : bp.l_edge + row;
        const int jlr = min(bp.r_edge + 1 + row, bp.q_len);
#pragma unroll
        for (int cc = 0; cc < C; cc++) {
          const int hold = H[cc];
          const int j = jb + cc;
          if (j >= jsr && j < jlr) {
            const int hin = diag + (int)__builtin_amdgcn_perm(tr.y, tr.x, sel[cc]) - bias;
            bool cand;
            (void)cell_update(H[cc], E[cc], F, hin, gi, ge, cand);
            if (cand && hin > best) best = hin;
          } else F = 0;
          diag = hold;
        }
      } else F = 0;
      if (sidx + 1 < nstrip) {                       // hand the last column to the next strip
        if (g == 63 && rowok) ring_out[rel & 127] = make_int2(H[C - 1], F);
        const int rdone = step - 63;                 // row lane 63 has just finished
        if (rdone >= 0 && ((rdone & 63) == 63 || rdone == nr - 1)) {
          __syncthreads();
          const int base = rdone & ~63, r2 = base + g;
          if (r2 <= rdone && r2 < nr) bnext[r_lo + r2] = ring_out[r2 & 127];
        }
      }
      if (sidx > 0 && (step & 63) == 63) __syncthreads();   // the read-ahead chunk is in place before lane 0 turns to it
    }
#undef SMG_BGET
    plo = r_lo; phi = r_hi;
    __threadfence();
  }
  for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
  return best;
}

// banded tasks of long reads (S7 lists them with the strip tasks) and K2a tasks whose score left 16 bits
__global__ void __launch_bounds__(64) k_sw_band(Batch b, DevIndex ix, MapPar p, int2 *bnd_all, uint8_t *win_all, uint32_t wcap) {
  const unsigned long long nlist = b.work[WK_STRIP_TASKS];
  if (nlist == 0 || !b.strip_list || nlist > b.strip_cap) return;       // k_sw_scalar takes whatever is left
  __shared__ uint2 tab[8];
  __shared__ int2 ring[256];
  const int bias = (p.mismatch < p.mismatch - p.match ? -p.mismatch : -(p.mismatch - p.match));
  sw_tab8(tab, p, bias);
  __syncthreads();
  int2 *bnd = bnd_all + (size_t)blockIdx.x * 2 * wcap;
  uint8_t *win = win_all + (size_t)blockIdx.x * wcap;
  unsigned long long nband = 0;
  for (uint32_t tl = blockIdx.x; tl < (uint32_t)nlist; tl += gridDim.x) {
    const uint32_t t = b.strip_list[tl];
    const RCand c = b.rcpool[t];
    const uint32_t qlen = read_len(b, c.rid), wlen = (uint32_t)(c.re - c.rs + 1);
    if (!(c.flags & RCF_BANDED) || (c.flags & (RCF_ERR | RCF_BSCORED)) || wlen > wcap) continue;     // wave-uniform
    Band bd;
    if (band_init(bd, c.band_l, c.band_r, (int)c.qs, (int)c.qe, (int)qlen, 0, (int)wlen - 1, (int)wlen)) {
      if (threadIdx.x == 0) b.rcpool[t].flags = c.flags | RCF_ERR;
      continue;
    }
    const uint64_t gbase = (c.sqidx < 0 ? 0ull : ix.sop[c.sqidx]) + c.rs;
    const uint8_t *q = ((c.flags & RCF_REVERSE) ? b.codes_rc : b.codes) + b.read_off[c.rid];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < wlen; i += 64) win[i] = (uint8_t)ref_code(ix.packed, gbase + i);
    __threadfence();
    __syncthreads();
    const int best = band_fast_wave(bd, q, win, bnd, wcap, tab, ring, bias, -p.gap_init, -p.gap_ext);
    if (threadIdx.x == 0) {
      b.rcpool[t].swscor = best;
      b.rcpool[t].flags = c.flags | RCF_SCORED | RCF_BSCORED;
      nband++;
    }
  }
  if (threadIdx.x == 0 && nband) atomicAdd(b.work + WK_TASKS_FULL, nband);
}

// K2a for tasks the register-tiled kernel does not cover (long reads / long windows) and
// K2b (banded score-only pass, alignment.c:1029) -- one lane per task, rows in HBM scratch.
__global__ void __launch_bounds__(64) k_sw_scalar(Batch b, DevIndex ix, MapPar p, int *rows, uint32_t rowlen, int full_gc) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nthr = gridDim.x * blockDim.x;
  const uint32_t ntask = min(*b.rc_count, b.rccap);      // the cursor runs past the pool when it overflows (reads keep SMG_ERR_CAP)
  if (b.work[WK_TASKS_FULL] >= (unsigned long long)ntask) return;      // every ranked candidate has its score already (the usual case)
  int *Hp = rows + (size_t)tid * 2 * rowlen, *Ep = Hp + rowlen;
  int8_t M[64];
  score_matrix(M, p.match, p.mismatch);
  for (uint32_t t = tid; t < ntask; t += nthr) {
    RCand c = b.rcpool[t];
    if (c.flags & (RCF_ERR | RCF_SCORED)) { if (!(c.flags & RCF_BANDED) || (c.flags & (RCF_ERR | RCF_BSCORED))) continue; }
    const uint32_t qlen = read_len(b, c.rid), wlen = (uint32_t)(c.re - c.rs + 1);
    const uint8_t *q = ((c.flags & RCF_REVERSE) ? b.codes_rc : b.codes) + b.read_off[c.rid];
    const uint64_t gbase = (c.sqidx < 0 ? 0ull : ix.sop[c.sqidx]) + c.rs;
    bool banded = (c.flags & RCF_BANDED) != 0;
    if (!banded) {
      if (wlen <= (uint32_t)SW_FULL_WMAX && qlen <= (uint32_t)full_gc) continue;   // k_sw_full did / will do it
      c.swscor = sw_full_scalar(q, qlen, ix.packed, gbase, wlen, M, -p.gap_init, -p.gap_ext, Hp, Ep);
      if (c.swscor >= 65535) banded = true;
    }
    if (banded) {
      Band bd;
      if (band_init(bd, c.band_l, c.band_r, (int)c.qs, (int)c.qe, (int)qlen, 0, (int)wlen - 1, (int)wlen)) { b.rcpool[t].flags = c.flags | RCF_ERR; continue; }
      c.swscor = band_fast_scalar(bd, q, ix.packed, gbase, M, -p.gap_init, -p.gap_ext, Hp, Ep);
    }
    b.rcpool[t].swscor = c.swscor;
    b.rcpool[t].flags = c.flags | RCF_SCORED;
  }
}

// stand-alone K2a over explicit (query, window) code arrays -- parity tests of the kernel
template <int G, int C>
__global__ void __launch_bounds__(64) k_sw_full_raw(const uint8_t *qcodes, const uint32_t *q_off, const uint8_t *rcodes,
                                                     const uint32_t *r_off, uint32_t ntask, MapPar p, int32_t *scores) {
  constexpr int NG = 64 / G;
  constexpr int WMAX = SW_FULL_WMAX;
  __shared__ uint8_t win[NG][WMAX + 8];
  __shared__ uint2 tab[8];
  const int lane = threadIdx.x, g = lane % G, grp = lane / G;
  const int bias = (p.mismatch < p.mismatch - p.match ? -p.mismatch : -(p.mismatch - p.match));
  const int gi = -p.gap_init, ge = -p.gap_ext;
  if (lane < 8) {
    int rb = lane == 7 ? 0 : ((lane == 6 || lane == 4) ? 5 : lane);
    uint32_t w[2] = {0, 0};
    for (int qc = 0; qc < 8; qc++) {
      int v = (rb == 5 || qc >= 4) ? 0 : ((rb == qc) ? p.match : p.mismatch);
      w[qc >> 2] |= (uint32_t)((v + bias) & 0xff) << (8 * (qc & 3));
    }
    tab[lane] = make_uint2(w[0], w[1]);
  }
  __syncthreads();
  const uint32_t ngroups = gridDim.x * NG;
  for (uint32_t t0 = blockIdx.x * NG; t0 < ntask; t0 += ngroups) {
    const uint32_t t = t0 + grp;
    uint32_t qlen = 0, wlen = 0;
    const uint8_t *q = qcodes, *r = rcodes;
    bool live = false;
    if (t < ntask) {
      qlen = q_off[t + 1] - q_off[t]; wlen = r_off[t + 1] - r_off[t];
      q += q_off[t]; r += r_off[t];
      live = wlen <= (uint32_t)WMAX && qlen <= (uint32_t)(G * C);
    }
    if (!live) { qlen = 0; wlen = 0; }
    for (uint32_t i = g; i < wlen; i += G) { uint32_t cd = r[i] & 7; win[grp][i] = (uint8_t)(cd == 7 ? 0 : (cd == 6 || cd == 4) ? 5 : cd); }
    uint32_t sel[C];
#pragma unroll
    for (int cc = 0; cc < C; cc++) {
      uint32_t j = (uint32_t)(g * C + cc);
      uint32_t qc = (j < qlen) ? (q[j] & 7u) : 5u;
      sel[cc] = 0x0c0c0c00u | qc;
    }
    int H[C], E[C];
#pragma unroll
    for (int cc = 0; cc < C; cc++) { H[cc] = 0; E[cc] = 0; }
    int best = 0, F = 0, prev_hl = 0;
    int nstep = (int)wlen + G - 1;
    for (int o = 32; o > 0; o >>= 1) nstep = max(nstep, __shfl_xor(nstep, o));
    __syncthreads();
    for (int step = 0; step < nstep; step++) {
      const int row = step - g;
      const uint32_t rb = (row >= 0 && row < (int)wlen) ? win[grp][row] : 5u;
      const uint2 tr = tab[rb];
      int hl = shr1<G>(H[C - 1]);
      int fin = shr1<G>(F);
      if (g == 0) { hl = 0; fin = 0; }
      int diag = prev_hl;
      prev_hl = hl;
      F = fin;
#pragma unroll
      for (int cc = 0; cc < C; cc++) {
        const int w = (int)__builtin_amdgcn_perm(tr.y, tr.x, sel[cc]);
        const int h = diag + w - bias;
        const int hh = max(max(h, E[cc]), F);
        best = max(best, hh);
        diag = H[cc];
        H[cc] = hh;
        const int tt = hh - gi;
        E[cc] = max(max(E[cc] - ge, tt), 0);
        F = max(max(F - ge, tt), 0);
      }
    }
    for (int o = G / 2; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
    if (t < ntask && g == 0) scores[t] = live ? best : -1;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// host-side launchers (declared in smg_kernels.h)
// ---------------------------------------------------------------------------------------
#define SMG_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

int launch_encode(hipStream_t s, const uint8_t *bases, const uint64_t *off, uint32_t n, uint8_t *codes, uint8_t *codes_rc) {
  if (!n) return 0;
  uint32_t grid = n < 16384u ? n : 16384u;
  hipLaunchKernelGGL(k_encode, dim3(grid), dim3(64), 0, s, bases, off, n, codes, codes_rc);
  SMG_LAUNCH_CHECK();
  return 0;
}

int launch_gather_reads(hipStream_t s, uint8_t *dst_bases, uint8_t *dst_quals, const uint64_t *dst_off, uint32_t n, const uint32_t *ids,
                        const uint8_t *const src_bases[2], const uint8_t *const src_quals[2], const uint64_t *const src_off[2]) {
  if (!n) return 0;
  const uint32_t grid = n < 32768u ? n : 32768u;
  hipLaunchKernelGGL(k_gather_reads, dim3(grid), dim3(64), 0, s, dst_bases, dst_quals, dst_off, n, ids, src_bases[0], src_bases[1],
                     dst_quals ? src_quals[0] : nullptr, dst_quals ? src_quals[1] : nullptr, src_off[0], src_off[1]);
  SMG_LAUNCH_CHECK();
  return 0;
}

int launch_seed(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, uint8_t *scratch, size_t sbytes, uint32_t nslots) {
  if (!b.nreads) return 0;
  const int use_lds = sbytes <= 48 * 1024;
  uint32_t items = 2 * b.nreads;
  uint32_t grid = use_lds ? (items < 32768u ? items : 32768u) : (items < nslots ? items : nslots);
  hipLaunchKernelGGL(k_seed, dim3(grid), dim3(64), use_lds ? sbytes + LDS_GUARD : 0, s, b, ix, p, scratch, sbytes, use_lds);
  SMG_LAUNCH_CHECK();
  return 0;
}

int launch_hits(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, uint32_t W, uint32_t tab, uint32_t nwg) {
  if (!b.nreads || !b.hitrun) return 0;
  const uint32_t lds_bytes = (uint32_t)((hits_lds_bytes(W, tab) + 15) & ~(size_t)15);
  const size_t seq_bytes = (ix.nseq > 0 && ix.nseq < 512) ? (((size_t)ix.nseq + 1) * 4 + 15) & ~(size_t)15 : 0;
  uint32_t grid = 2 * b.nreads < nwg ? 2 * b.nreads : nwg;
  static const int waves = getenv("SMALTGPU_HITS_WAVES") ? atoi(getenv("SMALTGPU_HITS_WAVES")) : 3;      // tuning hook (3 waves per SIMD: 148 registers, no spills; 2 and 4 measured the same)
  if (waves == 3) hipLaunchKernelGGL(k_hits<3>, dim3(grid), dim3(64), lds_bytes + LDS_GUARD + seq_bytes, s, b, ix, p, W, tab, lds_bytes);
  else if (waves == 2) hipLaunchKernelGGL(k_hits<2>, dim3(grid), dim3(64), lds_bytes + LDS_GUARD + seq_bytes, s, b, ix, p, W, tab, lds_bytes);
  else hipLaunchKernelGGL(k_hits<4>, dim3(grid), dim3(64), lds_bytes + LDS_GUARD + seq_bytes, s, b, ix, p, W, tab, lds_bytes);
  SMG_LAUNCH_CHECK();
  return 0;
}

int launch_cands(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, uint8_t *scratch, uint32_t nslots, const CandGeom &g) {
  if (!b.nreads) return 0;
  uint32_t grid = b.nreads < nslots ? b.nreads : nslots;
  const uint32_t lds_bytes = (uint32_t)(((strand_work_bytes<uint16_t>(g.lds_hits) + 15) & ~(size_t)15) + (size_t)5 * g.tab * 4);
  const size_t seq_bytes = (ix.nseq > 0 && ix.nseq < 512) ? (((size_t)ix.nseq + 1) * 4 + 15) & ~(size_t)15 : 0;     // LDS copy of seqlo
  if (b.qmax > 255) hipLaunchKernelGGL(k_cands<true>, dim3(grid), dim3(64), lds_bytes + LDS_GUARD + seq_bytes, s, b, ix, p, scratch, g, lds_bytes);
  else if (b.hitrun) {
    // S3 ran ahead (k_hits): no per-list tables in this kernel's LDS block, and (tuning hook) three waves per SIMD
    static const int waves3 = getenv("SMALTGPU_CANDS_WAVES") && atoi(getenv("SMALTGPU_CANDS_WAVES")) == 3;
    const uint32_t lb = (uint32_t)((strand_work_bytes<uint16_t>(g.lds_hits) + 15) & ~(size_t)15);
    if (waves3) hipLaunchKernelGGL((k_cands<false, true, 3>), dim3(grid), dim3(64), lb + LDS_GUARD + seq_bytes, s, b, ix, p, scratch, g, lb);
    else hipLaunchKernelGGL((k_cands<false, true, 2>), dim3(grid), dim3(64), lb + LDS_GUARD + seq_bytes, s, b, ix, p, scratch, g, lb);
  }
  else hipLaunchKernelGGL(k_cands<false>, dim3(grid), dim3(64), lds_bytes + LDS_GUARD + seq_bytes, s, b, ix, p, scratch, g, lds_bytes);
  SMG_LAUNCH_CHECK();
  return 0;
}

int launch_replay(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p) {
  if (!b.nreads) return 0;
  hipLaunchKernelGGL(k_replay, dim3((b.nreads + 255) / 256), dim3(256), 0, s, b, ix, p);
  SMG_LAUNCH_CHECK();
  return 0;
}

int launch_align(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, uint8_t *scratch, size_t sbytes, uint32_t nslots,
                 uint32_t wincap, uint64_t dircap, uint32_t rescap, uint32_t dstrcap, int pass) {
  if (!b.nreads) return 0;
  uint32_t grid = b.nreads < nslots ? b.nreads : nslots;
  size_t small = align_lds_small_bytes(b.qmax, wincap);
  static const uint32_t lds_kb = getenv("SMALTGPU_ALIGN_LDS_KB") ? (uint32_t)atoi(getenv("SMALTGPU_ALIGN_LDS_KB")) : 8u;   // tuning hook; 8 KB = 20 workgroups per CU
  uint32_t kb = lds_kb;
  if (b.qmax > 256 && small + 4096 > (size_t)kb * 1024 && small + 8192 <= 64 * 1024) kb = (uint32_t)((small + 8192 + 1023) / 1024);   // long reads: read, window and traceback string stay in LDS
  uint32_t lds_bytes = small + 4096 <= (size_t)kb * 1024 ? kb * 1024 - LDS_GUARD : 0;      // rows + window + direction bytes
  static const int antidiag = getenv("SMALTGPU_ALIGN_ANTIDIAG") ? 0x100 : 0;      // test hook: narrow bands in the anti-diagonal form (band_track_wave)
  pass |= antidiag;
  if (b.qmax > 256) hipLaunchKernelGGL(k_align<true>, dim3(grid), dim3(64), lds_bytes ? lds_bytes + LDS_GUARD : 0, s, b, ix, p, scratch, sbytes, wincap, dircap, rescap, dstrcap, lds_bytes, pass);
  else hipLaunchKernelGGL(k_align<false>, dim3(grid), dim3(64), lds_bytes ? lds_bytes + LDS_GUARD : 0, s, b, ix, p, scratch, sbytes, wincap, dircap, rescap, dstrcap, lds_bytes, pass);
  SMG_LAUNCH_CHECK();
  return 0;
}

int sw_full_geometry(uint32_t qmax_len, int *G, int *C) {
  if (qmax_len <= 64) { *G = 4; *C = 16; }
  else if (qmax_len <= 104) { *G = 8; *C = 13; }
  else if (qmax_len <= 152) { *G = 8; *C = 19; }
  else if (qmax_len <= 160) { *G = 8; *C = 20; }
  else if (qmax_len <= 256) { *G = 16; *C = 16; }
  else if (qmax_len <= 512) { *G = 16; *C = 32; }
  else { *G = 0; *C = 0; return -1; }
  return 0;
}

// the packed kernel's 16-bit lanes hold any score of a read of up to 512 bases
static bool sw16_ok(const MapPar &p) {
  const int bias = (p.mismatch < p.mismatch - p.match ? -p.mismatch : -(p.mismatch - p.match));
  return p.match > 0 && p.match * 512 + bias + 256 < 0x7C00 /* pk_max3 */ && bias >= 0 && p.match + bias < 256 && p.mismatch + bias >= 0 &&
         -p.gap_init >= 0 && -p.gap_init < 30000 && -p.gap_ext >= 0 && -p.gap_ext < 30000;
}

template <int G, int C>
static void launch_sw_full_t(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, uint32_t ntask_cap, uint32_t grid) {
  const int use16 = sw16_ok(p);
  if (use16) {
    hipLaunchKernelGGL((k_sw_full16<G, C, SW_SHORT_WMAX>), dim3(grid), dim3(64), 0, s, b, ix, p, ntask_cap);
    hipLaunchKernelGGL((k_sw_full16<G, C, SW_FULL_WMAX>), dim3(grid), dim3(64), 0, s, b, ix, p, ntask_cap);
  }
  hipLaunchKernelGGL((k_sw_full<G, C>), dim3(grid), dim3(64), 0, s, b, ix, p, ntask_cap, use16);
}

int launch_sw_full(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, uint32_t qmax_len, uint32_t ntask_cap, uint32_t grid) {
  int G, C;
  if (sw_full_geometry(qmax_len, &G, &C)) return 0;    // nothing for the register-tiled kernel: k_sw_scalar takes all
  if (G == 4) launch_sw_full_t<4, 16>(s, b, ix, p, ntask_cap, grid);
  else if (G == 8 && C == 13) launch_sw_full_t<8, 13>(s, b, ix, p, ntask_cap, grid);
  else if (G == 8 && C == 19) launch_sw_full_t<8, 19>(s, b, ix, p, ntask_cap, grid);
  else if (G == 8) launch_sw_full_t<8, 20>(s, b, ix, p, ntask_cap, grid);
  else if (C == 16) launch_sw_full_t<16, 16>(s, b, ix, p, ntask_cap, grid);
  else launch_sw_full_t<16, 32>(s, b, ix, p, ntask_cap, grid);
  SMG_LAUNCH_CHECK();
  return 0;
}

int launch_sw_strip(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, void *bnd, uint8_t *win, uint32_t wcap, uint32_t grid) {
  if (!b.nreads || !grid) return 0;
  if (sw16_ok(p) && (uint64_t)b.qmax * (uint64_t)p.match < 60000ull)          // 16-bit halves hold every score of these reads
  {
    if ((uint64_t)b.qmax * (uint64_t)p.match + 512 < 0x7C00ull)                 // every value below the half-float infinity pattern: max3 form
      hipLaunchKernelGGL(k_sw_strip16<true>, dim3(grid), dim3(64), 0, s, b, ix, p, (uint2 *)bnd, (uint16_t *)win, wcap);
    else hipLaunchKernelGGL(k_sw_strip16<false>, dim3(grid), dim3(64), 0, s, b, ix, p, (uint2 *)bnd, (uint16_t *)win, wcap);
  }
  hipLaunchKernelGGL(k_sw_strip, dim3(grid), dim3(64), 0, s, b, ix, p, (int2 *)bnd, win, wcap);
  hipLaunchKernelGGL(k_sw_band, dim3(grid), dim3(64), 0, s, b, ix, p, (int2 *)bnd, win, wcap);
  SMG_LAUNCH_CHECK();
  return 0;
}

int launch_sw_strip_raw(hipStream_t s, const uint8_t *q, const uint32_t *qo, const uint8_t *r, const uint32_t *ro, uint32_t n, const MapPar &p,
                        int32_t *sc, void *bnd, uint8_t *win, uint32_t wcap, uint32_t grid) {
  if (!n) return 0;
  hipLaunchKernelGGL(k_sw_strip_raw, dim3(n < grid ? n : grid), dim3(64), 0, s, q, qo, r, ro, n, p, sc, (int2 *)bnd, win, wcap);
  SMG_LAUNCH_CHECK();
  return 0;
}

int launch_sw_scalar(hipStream_t s, const Batch &b, const DevIndex &ix, const MapPar &p, int *rows, uint32_t rowlen, uint32_t nthreads,
                     uint32_t qmax_len) {
  int G, C;
  int full_gc = sw_full_geometry(qmax_len, &G, &C) ? 0 : G * C;
  hipLaunchKernelGGL(k_sw_scalar, dim3(nthreads / 64), dim3(64), 0, s, b, ix, p, rows, rowlen, full_gc);
  SMG_LAUNCH_CHECK();
  return 0;
}

template <int G, int C>
static void launch_sw_raw_t(hipStream_t s, const uint8_t *q, const uint32_t *qo, const uint8_t *r, const uint32_t *ro, uint32_t n,
                            const MapPar &p, int32_t *sc, uint32_t grid, int packed16) {
  if (packed16) hipLaunchKernelGGL((k_sw_full16_raw<G, C>), dim3(grid), dim3(64), 0, s, q, qo, r, ro, n, p, sc);
  else hipLaunchKernelGGL((k_sw_full_raw<G, C>), dim3(grid), dim3(64), 0, s, q, qo, r, ro, n, p, sc);
}

// test entry of the candidate ranking sort (smg_wsort.hpp): one wave per array of keys
__global__ void __launch_bounds__(64) k_rank_sort_raw(const uint32_t *keys, const uint32_t *off, uint32_t narr, int nneed, int in_lds,
                                                      uint32_t *kv, uint32_t *out_key, uint32_t *out_idx) {
  __shared__ uint32_t sh[16 + WSORT_WORDS + 8192];
  uint32_t *wk = sh + 16, *arr = wk + WSORT_WORDS;
  for (uint32_t t = blockIdx.x; t < narr; t += gridDim.x) {
    const uint32_t o = off[t], n = off[t + 1] - o;
    uint32_t *a = (in_lds && n <= 8192) ? arr : kv + o;
    for (uint32_t i = threadIdx.x; i < n; i += 64) a[i] = (keys[o + i] << WSORT_IDXBITS) | i;
    __syncthreads();
    wave_sort_kv(a, (int)n, nneed < 0 ? (int)n : nneed, wk);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 64) { out_key[o + i] = a[i] >> WSORT_IDXBITS; out_idx[o + i] = a[i] & ((1u << WSORT_IDXBITS) - 1u); }
    __syncthreads();
  }
}

int launch_rank_sort_raw(hipStream_t s, const uint32_t *keys, const uint32_t *off, uint32_t narr, int nneed, int in_lds, uint32_t *kv,
                         uint32_t *out_key, uint32_t *out_idx) {
  if (!narr) return 0;
  hipLaunchKernelGGL(k_rank_sort_raw, dim3(narr < 4096 ? narr : 4096), dim3(64), 0, s, keys, off, narr, nneed, in_lds, kv, out_key, out_idx);
  SMG_LAUNCH_CHECK();
  return 0;
}

int launch_sw_full_raw(hipStream_t s, const uint8_t *q, const uint32_t *qo, const uint8_t *r, const uint32_t *ro, uint32_t n,
                       const MapPar &p, int32_t *sc, uint32_t qmax_len, int packed16) {
  int G, C;
  if (!n) return 0;
  if (sw_full_geometry(qmax_len, &G, &C)) return -1;
  if (packed16 && !sw16_ok(p)) return -2;
  uint32_t grid = (n + (64 / G) - 1) / (64 / G);
  if (grid > 8192) grid = 8192;
  if (G == 4) launch_sw_raw_t<4, 16>(s, q, qo, r, ro, n, p, sc, grid, packed16);
  else if (G == 8 && C == 13) launch_sw_raw_t<8, 13>(s, q, qo, r, ro, n, p, sc, grid, packed16);
  else if (G == 8 && C == 19) launch_sw_raw_t<8, 19>(s, q, qo, r, ro, n, p, sc, grid, packed16);
  else if (G == 8) launch_sw_raw_t<8, 20>(s, q, qo, r, ro, n, p, sc, grid, packed16);
  else if (C == 16) launch_sw_raw_t<16, 16>(s, q, qo, r, ro, n, p, sc, grid, packed16);
  else launch_sw_raw_t<16, 32>(s, q, qo, r, ro, n, p, sc, grid, packed16);
  SMG_LAUNCH_CHECK();
  return 0;
}

}  // namespace smg
