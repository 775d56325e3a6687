// smg_exec.h -- the execution model the per-read stage functions are written against.
//
// A read (or read-strand) is owned by ONE 64-lane wavefront (a 64-thread workgroup).  Stage
// code is written once with the macros below:
//   SMG_PAR_CHUNKS(base, n)  wave-uniform loop over chunks of SMG_NLANES items; inside, the
//                            item of this lane is  base + SMG_LANE  (may be >= n)
//   SMG_LANE0                the sequential sections (one lane; the reference's order matters)
//   SMG_SYNC()               workgroup barrier + memory visibility between the two
// On the device SMG_NLANES = 64.  The same source also compiles for the host with
// SMG_NLANES = 1 (tests/hostemu): that build exists to unit-test this logic on a machine
// without a GPU and is never part of libsmaltgpu.so.
#pragma once
#include <stdint.h>
#if !defined(__HIP_DEVICE_COMPILE__)
#include <algorithm>
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define SMG_LANE ((uint32_t)threadIdx.x)
#define SMG_NLANES 64u
#define SMG_SYNC() __syncthreads()
#else
#define SMG_LANE 0u
#define SMG_NLANES 1u
#define SMG_SYNC() do {} while (0)
#endif

// Pointers that are known to address the workgroup's LDS block carry the LDS address space on the device, so that
// the compiler emits ds_* instructions (32-bit addresses) instead of flat ones; plain pointers on the host build.
#if defined(__HIP_DEVICE_COMPILE__)
#define SMG_LDSQ __attribute__((address_space(3)))
#else
#define SMG_LDSQ
#endif
namespace smg {
template <class T, bool IN_LDS> struct ptr_of { typedef T *type; };
template <class T> struct ptr_of<T, true> { typedef SMG_LDSQ T *type; };
}

#define SMG_PAR_CHUNKS(base, n) for (uint32_t base = 0; base < (uint32_t)(n); base += SMG_NLANES)
#define SMG_LANE0 if (SMG_LANE == 0)

namespace smg {

// phase clock for the diagnostic counters (shader clock on the device, nothing on the host)
SMG_HD inline unsigned long long phase_clock() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (unsigned long long)__builtin_readcyclecounter();
#else
  return 0;
#endif
}

// Ordered stream compaction: lanes with flag get consecutive slots in lane order.
// `counter` is wave-uniform.
SMG_HD inline uint32_t compact_slot(bool flag, uint32_t &counter) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned long long m = __ballot(flag);
  uint32_t slot = counter + (uint32_t)__popcll(m & ((1ull << SMG_LANE) - 1ull));
  counter += (uint32_t)__popcll(m);
  return slot;
#else
  uint32_t slot = counter;
  if (flag) counter++;
  return slot;
#endif
}

SMG_HD inline uint32_t wave_sum_u32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
#else
  return v;
#endif
}

SMG_HD inline uint32_t wave_max_u32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  for (int o = 32; o > 0; o >>= 1) { const uint32_t u = (uint32_t)__shfl_xor((int)v, o); if (u > v) v = u; }
#endif
  return v;
}

SMG_HD inline uint64_t wave_min_u64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o);
    const uint64_t u = ((uint64_t)hi << 32) | lo;
    if (u < v) v = u;
  }
#endif
  return v;
}

SMG_HD inline bool wave_any(bool f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __ballot(f) != 0ull;
#else
  return f;
#endif
}

// inclusive running maximum over the lanes, continued from `carry` (the maximum of everything before lane 0); returns the
// lane's value, *carry_out = the wave's maximum (wave-uniform)
SMG_HD inline uint32_t wave_scan_max_u32(uint32_t v, uint32_t carry, uint32_t *carry_out) {
#if defined(__HIP_DEVICE_COMPILE__)
  int x = (int)v;                                  // values are small non-negative numbers: 0 is the identity
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111 /* row_shr:1 */, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112 /* row_shr:2 */, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114 /* row_shr:4 */, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118 /* row_shr:8 */, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142 /* row_bcast:15 */, 0xa, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143 /* row_bcast:31 */, 0xc, 0xf, false));
  x = max(x, (int)carry);
  *carry_out = (uint32_t)__builtin_amdgcn_readlane(x, 63);
  return (uint32_t)x;
#else
  const uint32_t r = v > carry ? v : carry;
  *carry_out = r;
  return r;
#endif
}

SMG_HD inline uint32_t bcast_lane0(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#else
  return v;
#endif
}

// In-place ascending sort of n 64-bit keys by the whole wave: bitonic network in its normalised form (every
// comparator puts the smaller key at the lower index: the first step of a merge pairs i with its mirror
// image in the block, the others pair i with i + j).  Keys beyond n are virtual +inf: a comparator that
// touches one never swaps, so no padding is stored and only comparators below n are visited.
template <class P>
SMG_HD inline void wave_sort_u64(P a, uint32_t n) {
  if (n < 2) return;
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t np = 1;
  while (np < n) np <<= 1;
  // Each step loads the operands of up to four comparators per lane before it stores any: the step is bound by
  // LDS latency, and four independent round trips overlap.
  for (uint32_t k = 2; k <= np; k <<= 1) {
    const uint32_t h = k >> 1;
    {                                                           // mirror step
      const uint32_t tmax = (n / k) * h + ((n % k) < h ? (n % k) : h);
      for (uint32_t t0 = 0; t0 < tmax; t0 += 4 * SMG_NLANES) {
        uint32_t ii[4], pp[4];
        uint64_t x[4], y[4];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t t = t0 + (uint32_t)u * SMG_NLANES + SMG_LANE;
          const uint32_t blk = t / h, off = t % h;
          ii[u] = blk * k + off; pp[u] = blk * k + (k - 1 - off);
          on[u] = t < tmax && pp[u] < n;
          if (on[u]) { x[u] = a[ii[u]]; y[u] = a[pp[u]]; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) if (on[u] && x[u] > y[u]) { a[ii[u]] = y[u]; a[pp[u]] = x[u]; }
      }
      SMG_SYNC();
    }
    for (uint32_t j = h >> 1; j > 0; j >>= 1) {
      const uint32_t tmax = (n / (2 * j)) * j + ((n % (2 * j)) < j ? (n % (2 * j)) : j);
      for (uint32_t t0 = 0; t0 < tmax; t0 += 4 * SMG_NLANES) {
        uint32_t ii[4];
        uint64_t x[4], y[4];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t t = t0 + (uint32_t)u * SMG_NLANES + SMG_LANE;
          ii[u] = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          on[u] = t < tmax && (ii[u] | j) < n;
          if (on[u]) { x[u] = a[ii[u]]; y[u] = a[ii[u] | j]; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) if (on[u] && x[u] > y[u]) { a[ii[u]] = y[u]; a[ii[u] | j] = x[u]; }
      }
      SMG_SYNC();
    }
  }
#else
  // host build: any correct sort gives the same array (keys are plain integers)
  std::sort((uint64_t *)a, (uint64_t *)a + n);
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
// The same network with the keys in registers, for arrays of at most 64 * E keys in LDS.  Lane l holds the logical
// elements E * l .. E * l + E - 1 (which unsorted key starts where is free, so the load is coalesced); comparators
// whose span stays below E are compare-exchanges between a lane's own registers, the others pair register r of a lane
// with register r (mirror step: E - 1 - r) of the lane at distance span / E and keep the smaller or the larger key
// by side.  34 of the 55 steps of a 1024-key sort touch neither LDS nor another lane.
__device__ inline uint64_t shfl_xor_u64(uint64_t v, int mask) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask);
  return ((uint64_t)hi << 32) | lo;
}
template <int E, class P>
__device__ inline void wave_sort_u64_reg(P a, uint32_t n) {
  if (n < 2) return;
  const uint32_t lane = threadIdx.x;
  uint64_t v[E];
#pragma unroll
  for (int r = 0; r < E; r++) { const uint32_t idx = (uint32_t)r * 64u + lane; v[r] = idx < n ? a[idx] : ~0ull; }
#pragma unroll
  for (int k = 2; k <= 64 * E; k <<= 1) {              // the whole network: the coalesced load spreads the keys over all lanes
    if (k <= E) {                                      // mirror step inside the lane
#pragma unroll
      for (int r = 0; r < E; r++) {
        const int q = r ^ (k - 1);
        if (r < q) { const uint64_t x = v[r], y = v[q]; const bool sw = x > y; v[r] = sw ? y : x; v[q] = sw ? x : y; }
      }
    } else {                                           // mirror step across lanes
      const int m = k / E - 1;
      const bool lower = (lane & (uint32_t)(k / (2 * E))) == 0;
      uint64_t pv[E];
#pragma unroll
      for (int r = 0; r < E; r++) pv[r] = shfl_xor_u64(v[E - 1 - r], m);
#pragma unroll
      for (int r = 0; r < E; r++) { const bool take = lower ? (pv[r] < v[r]) : (pv[r] > v[r]); v[r] = take ? pv[r] : v[r]; }
    }
#pragma unroll
    for (int j = k / 4; j >= 1; j >>= 1) {
      if (j < E) {
#pragma unroll
        for (int r = 0; r < E; r++) {
          if (!(r & j)) { const uint64_t x = v[r], y = v[r | j]; const bool sw = x > y; v[r] = sw ? y : x; v[r | j] = sw ? x : y; }
        }
      } else {
        const int m = j / E;
        const bool lower = (lane & (uint32_t)m) == 0;
#pragma unroll
        for (int r = 0; r < E; r++) { const uint64_t o = shfl_xor_u64(v[r], m); const bool take = lower ? (o < v[r]) : (o > v[r]); v[r] = take ? o : v[r]; }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < E; r++) { const uint32_t i = (uint32_t)E * lane + (uint32_t)r; if (i < n) a[i] = v[r]; }
  __syncthreads();
}

// Runs of more than 1024 keys in HBM (the exhaustive search: every seed of a strand in a repeat): wave_sort_u64 over global
// memory pays a round trip per stage -- 105 of them for 16 384 keys.  Here every chunk of 1024 keys is sorted in registers, and of
// the later merges only the comparators that span chunks run in memory (the mirror step and the half-cleaners with j >= 1024);
// the rest of each merge, j = 512 .. 1, stays inside a chunk and runs in registers again: about ten passes over the memory
// and five register passes per chunk for 16 384 keys.  Same network, same +inf convention for the keys beyond n.
// MERGE = false: the chunk's keys in any order -> sorted (the full network; loads in the order that coalesces, lane l ends up
// with the logical elements 16 l .. 16 l + 15).  MERGE = true: only the half-cleaners j = 512 .. 1 of a larger merge; here register
// r of lane l holds element 64 r + l, so that loads and stores coalesce: j >= 64 pairs registers of one lane, j < 64 lanes.
template <bool MERGE, class P>
__device__ inline void sort_chunk_1024(P a, uint32_t base, uint32_t n) {
  constexpr int E = 16;
  const uint32_t lane = threadIdx.x;
  uint64_t v[E];
#pragma unroll
  for (int r = 0; r < E; r++) { const uint32_t idx = base + (uint32_t)r * 64u + lane; v[r] = idx < n ? a[idx] : ~0ull; }
  if (MERGE) {
#pragma unroll
    for (int j = 512; j >= 64; j >>= 1) {
#pragma unroll
      for (int r = 0; r < E; r++) {
        if (!(r & (j / 64))) { const uint64_t x = v[r], y = v[r | (j / 64)]; const bool sw = x > y; v[r] = sw ? y : x; v[r | (j / 64)] = sw ? x : y; }
      }
    }
#pragma unroll
    for (int j = 32; j >= 1; j >>= 1) {
      const bool lower = (lane & (uint32_t)j) == 0;
#pragma unroll
      for (int r = 0; r < E; r++) { const uint64_t o = shfl_xor_u64(v[r], j); const bool take = lower ? (o < v[r]) : (o > v[r]); v[r] = take ? o : v[r]; }
    }
#pragma unroll
    for (int r = 0; r < E; r++) { const uint32_t idx = base + (uint32_t)r * 64u + lane; if (idx < n) a[idx] = v[r]; }
    return;
  }
#pragma unroll
  for (int k = 2; k <= 1024; k <<= 1) {
    if (k <= E) {
#pragma unroll
      for (int r = 0; r < E; r++) {
        const int q = r ^ (k - 1);
        if (r < q) { const uint64_t x = v[r], y = v[q]; const bool sw = x > y; v[r] = sw ? y : x; v[q] = sw ? x : y; }
      }
    } else {
      const int m = k / E - 1;
      const bool lower = (lane & (uint32_t)(k / (2 * E))) == 0;
      uint64_t pv[E];
#pragma unroll
      for (int r = 0; r < E; r++) pv[r] = shfl_xor_u64(v[E - 1 - r], m);
#pragma unroll
      for (int r = 0; r < E; r++) { const bool take = lower ? (pv[r] < v[r]) : (pv[r] > v[r]); v[r] = take ? pv[r] : v[r]; }
    }
#pragma unroll
    for (int j = k / 4; j >= 1; j >>= 1) {
      if (j < E) {
#pragma unroll
        for (int r = 0; r < E; r++) {
          if (!(r & j)) { const uint64_t x = v[r], y = v[r | j]; const bool sw = x > y; v[r] = sw ? y : x; v[r | j] = sw ? x : y; }
        }
      } else {
        const int m = j / E;
        const bool lower = (lane & (uint32_t)m) == 0;
#pragma unroll
        for (int r = 0; r < E; r++) { const uint64_t o = shfl_xor_u64(v[r], m); const bool take = lower ? (o < v[r]) : (o > v[r]); v[r] = take ? o : v[r]; }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < E; r++) { const uint32_t idx = base + (uint32_t)E * lane + (uint32_t)r; if (idx < n) a[idx] = v[r]; }
}

template <class P>
__device__ __noinline__ void wave_sort_u64_chunked(P a, uint32_t n) {
  constexpr uint32_t C = 1024;
  for (uint32_t base = 0; base < n; base += C) sort_chunk_1024<false>(a, base, n);
  __threadfence_block();
  __syncthreads();
  uint32_t np = C;
  while (np < n) np <<= 1;
  for (uint32_t k = 2 * C; k <= np; k <<= 1) {
    const uint32_t h = k >> 1;
    {                                                           // mirror step (pairs i with its mirror image in the block of k)
      const uint32_t tmax = (n / k) * h + ((n % k) < h ? (n % k) : h);
      for (uint32_t t0 = 0; t0 < tmax; t0 += 4 * SMG_NLANES) {
        uint32_t ii[4], pp[4];
        uint64_t x[4], y[4];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t t = t0 + (uint32_t)u * SMG_NLANES + SMG_LANE;
          const uint32_t blk = t / h, off = t % h;
          ii[u] = blk * k + off; pp[u] = blk * k + (k - 1 - off);
          on[u] = t < tmax && pp[u] < n;
          if (on[u]) { x[u] = a[ii[u]]; y[u] = a[pp[u]]; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) if (on[u] && x[u] > y[u]) { a[ii[u]] = y[u]; a[pp[u]] = x[u]; }
      }
      __threadfence_block();
      __syncthreads();
    }
    for (uint32_t j = h >> 1; j >= C; j >>= 1) {                // half-cleaners that span chunks
      const uint32_t tmax = (n / (2 * j)) * j + ((n % (2 * j)) < j ? (n % (2 * j)) : j);
      for (uint32_t t0 = 0; t0 < tmax; t0 += 4 * SMG_NLANES) {
        uint32_t ii[4];
        uint64_t x[4], y[4];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t t = t0 + (uint32_t)u * SMG_NLANES + SMG_LANE;
          ii[u] = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          on[u] = t < tmax && (ii[u] | j) < n;
          if (on[u]) { x[u] = a[ii[u]]; y[u] = a[ii[u] | j]; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) if (on[u] && x[u] > y[u]) { a[ii[u]] = y[u]; a[ii[u] | j] = x[u]; }
      }
      __threadfence_block();
      __syncthreads();
    }
    for (uint32_t base = 0; base < n; base += C) sort_chunk_1024<true>(a, base, n);      // j = 512 .. 1 inside the chunks
    __threadfence_block();
    __syncthreads();
  }
}
#endif

}  // namespace smg
