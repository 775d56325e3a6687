// smg_indexbuild.hip -- the k-mer index and the packed reference built in HBM.
//
// What the reference does on one host thread (hashTableSetUp, hashidx.c:829-998: count words, allocate, fill the
// position lists sequence by sequence; seqSetCompress, sequence.c:1360-1424) is here a stream compaction, one stable
// radix sort and two prefix sums over the sampled k-mers:
//   * k-mer serial t covers the bases [t*s, t*s + k) of the concatenated reference; it is indexed iff it lies inside
//     one sequence and holds no non-ACGT base (doWordsInSeq, hashidx.c:465-531: the sampling grid is global, serial
//     numbers run on across sequence boundaries);
//   * PERFECT: key = the 2k-bit word; pos lists = serials sorted by (key, serial); idx = prefix sums of key counts;
//   * HASH32MIX (hashidx.c:155-172): key = hash32mix(word >> nbits_lo) % 2^(nbits_key - nbits_lo) << nbits_lo | low
//     bits; serials sorted by (key, word_hi, serial); one (wordidx, posidx) entry per distinct (key, word_hi); idx =
//     prefix sums of the number of distinct words per key.
// Serials are generated in ascending order and rocPRIM's radix sort is stable, so sorting by the key alone leaves the
// serials of equal keys ascending.  All of it is HBM-bound integer work: no LDS tiling to speak of, rocPRIM's device
// scan / sort do the passes.
#include <cstring>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "smg_common.h"
#include "smg_indexbuild.h"

namespace smg {

#define IB_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(err, errlen, "%s: %s", #x, hipGetErrorString(e_)); goto fail; } } while (0)

// sequence.c:287-322: upper-case; U -> T; A C G T -> 0..3; everything else -> 5
__device__ inline uint8_t ref_code_of_ascii(uint8_t c) {
  c &= 0xDF;                                   // upper case (letters)
  return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : (c == 'T' || c == 'U') ? 3 : 5;
}

// letters [g0, g0 + span) of the reference as codes into LDS; beyond the end: `fill` (and 7 right at the end if term).
// Four letters per load when the source is word-aligned (g0 is a multiple of 4 for every caller).
__device__ inline void ib_stage_codes(uint8_t *cd, const uint8_t *ascii, uint64_t g0, uint32_t span, uint64_t tot, bool term, uint8_t fill) {
  if ((((uintptr_t)ascii + g0) & 3) == 0) {
    for (uint32_t i = threadIdx.x * 4; i < span; i += 256 * 4) {
      const uint64_t o = g0 + i;
      uint32_t w = 0;
      if (o + 4 <= tot) w = *(const uint32_t *)(ascii + o);
      else for (int u = 0; u < 4; u++) if (o + u < tot) w |= (uint32_t)ascii[o + u] << (8 * u);
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (i + u < span) cd[i + u] = o + u < tot ? ref_code_of_ascii((uint8_t)(w >> (8 * u))) : ((term && o + u == tot) ? 7 : fill);
    }
  } else {
    for (uint32_t i = threadIdx.x; i < span; i += 256) { const uint64_t o = g0 + i; cd[i] = o < tot ? ref_code_of_ascii(ascii[o]) : ((term && o == tot) ? 7 : fill); }
  }
}

// compressSeq (sequence.c:1360-1424): 10 codes per word, first base in the highest bits, terminator 7 after the last base.
// A block stages the 2560 letters of its 256 words through LDS (coalesced reads), then each thread packs one word.
__global__ void __launch_bounds__(256) k_ib_pack(const uint8_t *ascii, uint64_t tot, uint64_t nwords, uint32_t *packed) {
  __shared__ uint8_t cd[2560];
  for (uint64_t w0 = (uint64_t)blockIdx.x * 256; w0 < nwords; w0 += (uint64_t)gridDim.x * 256) {
    __syncthreads();
    ib_stage_codes(cd, ascii, w0 * 10, 2560, tot, true, 0);
    __syncthreads();
    const uint64_t w = w0 + threadIdx.x;
    if (w < nwords) {
      uint32_t v = 0;
      for (int i = 0; i < 10; i++) v |= (uint32_t)cd[threadIdx.x * 10 + i] << (3 * (9 - i));
      packed[w] = v;
    }
  }
}

__device__ inline uint32_t ib_hash32mix(uint32_t a) {        // hashidx.c:163-172
  a = (a + 0x7ed55d16) + (a << 12);
  a = (a ^ 0xc761c23c) ^ (a >> 19);
  a = (a + 0x165667b1) + (a << 5);
  a = (a + 0xd3a2646c) ^ (a << 9);
  a = (a + 0xfd7046c5) + (a << 3);
  a = (a ^ 0xb55a4f09) ^ (a >> 16);
  return a;
}

// one thread per k-mer serial.  K32 (PERFECT with 2k <= 30): 32-bit sort key, the sentinel 2^2k for serials that are not
// indexed (they sort behind every key), value = serial; the number of indexed serials goes to *nvalid.  Otherwise:
// validity flag and 64-bit sort key ((key << 32) | word_hi; PERFECT: the word) for the compaction path.
template <bool K32>
__global__ void __launch_bounds__(256) k_ib_kmers(const uint8_t *ascii, const uint64_t *sop, int nseq, uint64_t ntup, int k, int s, int typ,
                                                  int nbits_key, int nbits_lo, uint32_t *flag, uint64_t *key64, uint32_t *key32, uint32_t *pos32,
                                                  unsigned long long *nvalid) {
  unsigned long long mine = 0;
  // the letters of the block's 256 serials go through LDS as codes (coalesced reads; a base is in k / s words)
  extern __shared__ uint8_t cd[];                      // 256 * s + k codes
  const uint32_t span = 256u * (uint32_t)s + (uint32_t)k;
  for (uint64_t t0 = (uint64_t)blockIdx.x * 256; t0 < ntup; t0 += (uint64_t)gridDim.x * 256) {
    const uint64_t g0 = t0 * (uint64_t)s, tot = sop[nseq];
    __syncthreads();
    ib_stage_codes(cd, ascii, g0, span, tot, false, 5);
    __syncthreads();
    const uint64_t t = t0 + threadIdx.x;
    if (t >= ntup) continue;
    const uint64_t g = t * (uint64_t)s;
    int lo = 0, hi = nseq;                       // sequence of base g: last i with sop[i] <= g
    {                                            // block-uniform search for the block's first base; nearly always the answer
      int ulo = 0, uhi = nseq;
      while (uhi - ulo > 1) { const int mid = (ulo + uhi) >> 1; if (sop[mid] <= g0) ulo = mid; else uhi = mid; }
      if (g < sop[ulo + 1]) { lo = ulo; hi = ulo + 1; } else lo = ulo;
    }
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sop[mid] <= g) lo = mid; else hi = mid; }
    bool ok = g + (uint64_t)k <= sop[lo + 1];
    uint64_t word = 0;
    if (ok) {
      const uint8_t *c0 = cd + threadIdx.x * (uint32_t)s;
      for (int j = 0; j < k; j++) { const uint32_t c = c0[j]; if (c & 4) ok = false; word = (word << 2) | (c & 3); }
    }
    if (K32) {
      key32[t] = ok ? (uint32_t)word : (1u << (2 * k));
      pos32[t] = (uint32_t)t;
      mine += ok ? 1ull : 0ull;
    } else {
      uint64_t kv = 0;
      if (ok) {
        if (typ == IDX_PERFECT) kv = word;
        else {
          const uint64_t mask_lo = (1ull << nbits_lo) - 1ull;
          const uint32_t whi = (uint32_t)(word >> nbits_lo);
          const uint32_t keymod = 1u << (nbits_key - nbits_lo);
          const uint32_t key = ((ib_hash32mix(whi) % keymod) << nbits_lo) + (uint32_t)(word & mask_lo);
          kv = ((uint64_t)key << 32) | whi;
        }
      }
      flag[t] = ok ? 1u : 0u;
      key64[t] = kv;
    }
  }
  if (K32) {
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(nvalid, mine);
  }
}

__global__ void __launch_bounds__(256) k_ib_scatter(const uint32_t *flag, const uint32_t *slot, const uint64_t *key64, uint64_t ntup, uint64_t *okey, uint32_t *opos) {
  for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < ntup; t += (uint64_t)gridDim.x * 256)
    if (flag[t]) { okey[slot[t]] = key64[t]; opos[slot[t]] = (uint32_t)t; }
}

// idx[key + 1] = number of sorted elements with a key <= key: written where a run of equal keys ends (no atomics;
// keys without elements keep 0 and are filled by the running maximum afterwards -- the sequence is non-decreasing)
__global__ void __launch_bounds__(256) k_ib_runend32(const uint32_t *skey, uint64_t n, uint32_t *idx) {
  for (uint64_t a = (uint64_t)blockIdx.x * 256 + threadIdx.x; a < n; a += (uint64_t)gridDim.x * 256)
    if (a + 1 == n || skey[a] != skey[a + 1]) idx[(uint64_t)skey[a] + 1] = (uint32_t)(a + 1);
}
__global__ void __launch_bounds__(256) k_ib_runend64(const uint64_t *skey, uint64_t n, uint32_t *idx) {
  for (uint64_t a = (uint64_t)blockIdx.x * 256 + threadIdx.x; a < n; a += (uint64_t)gridDim.x * 256)
    if (a + 1 == n || skey[a] != skey[a + 1]) idx[skey[a] + 1] = (uint32_t)(a + 1);
}

// HASH32MIX: heads of runs of equal (key, word_hi)
__global__ void __launch_bounds__(256) k_ib_heads(const uint64_t *skey, uint64_t n, uint32_t *head) {
  for (uint64_t a = (uint64_t)blockIdx.x * 256 + threadIdx.x; a < n; a += (uint64_t)gridDim.x * 256) head[a] = (a == 0 || skey[a] != skey[a - 1]) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_ib_words(const uint64_t *skey, const uint32_t *head, const uint32_t *slot, uint64_t n, uint32_t *wordidx, uint32_t *posidx, uint32_t *idx) {
  for (uint64_t a = (uint64_t)blockIdx.x * 256 + threadIdx.x; a < n; a += (uint64_t)gridDim.x * 256) {
    if (head[a]) { wordidx[slot[a]] = (uint32_t)skey[a]; posidx[slot[a]] = (uint32_t)a; }
    const uint32_t key = (uint32_t)(skey[a] >> 32);
    if (a + 1 == n || key != (uint32_t)(skey[a + 1] >> 32)) idx[(uint64_t)key + 1] = slot[a] + head[a];   // distinct words with a key <= key
  }
}

static inline unsigned ib_grid(uint64_t n) { uint64_t g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 65536 ? 65536 : g)); }

// selectHashTyp (smalt.c:268-332)
int index_geometry(int k, int s, uint64_t totlen, int *typ, int *nbits_key, int *nbits_lo) {
  const int nbk = 2 * k;
  *typ = IDX_PERFECT; *nbits_key = nbk; *nbits_lo = 0;
  if (nbk > 63 || s < 1) return -1;
  const uint64_t ntup = totlen / (uint64_t)s, nkey = 1ull << nbk;
  if (ntup > 0xFFFFFFFFull) return -1;
  if (nkey > 2 * ntup) {
    int last_b = (ntup & 1) ? 1 : 0, nk, np = 0;
    uint32_t t = (uint32_t)ntup;
    for (int i = 0; i < 32; i++) { t >>= 1; if (t & 1) last_b = i; }
    nk = (last_b & 1) ? last_b + 1 : last_b;
    if (nbk > 32) { np = nbk - 32; if (np > 10) return -1; }
    if (nk + np > 26) nk = 26 - np;
    if (nk < np + 1) nk = np + 1;
    if (nk > 26) nk = 26;
    *typ = IDX_HASH32MIX; *nbits_key = nk; *nbits_lo = np;
  }
  return 0;
}

int build_index_device(const uint8_t *d_ascii, uint64_t tot, const uint64_t *h_sop, int nseq, int k, int s, BuiltIndex *out, char *err, size_t errlen) {
  BuiltIndex b;
  memset(&b, 0, sizeof(b));
  uint64_t *d_sop = nullptr, *key64 = nullptr, *okey = nullptr, *skey = nullptr;
  uint32_t *flag = nullptr, *slot = nullptr, *opos = nullptr, *head = nullptr, *key32 = nullptr, *skey32 = nullptr;
  unsigned long long *d_nvalid = nullptr, h_nvalid = 0;
  void *tmp = nullptr;
  size_t tmp_bytes = 0, need = 0;
  uint32_t last_slot = 0, last_flag = 0;
  uint64_t n = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (index_geometry(k, s, tot, &b.typ, &b.nbits_key, &b.nbits_lo) || b.nbits_key > 31 || nseq < 1 || s > 200) { snprintf(err, errlen, "unsupported index geometry (k=%d s=%d)", k, s); return -1; }
  for (int i = 0; i < nseq; i++) if (h_sop[i + 1] - h_sop[i] < (uint64_t)k) { snprintf(err, errlen, "sequence %d is shorter than the word length (hashidx.c:499)", i); return -1; }
  {
  b.nkeys = 1u << b.nbits_key;
  const uint64_t ntup = (tot + (uint64_t)s - 1) / (uint64_t)s;           // serials 0 .. ntup-1 start inside the reference
  const uint64_t nwords = tot / 10 + 1;
  const bool k32 = b.typ == IDX_PERFECT && 2 * k <= 30;                  // 32-bit keys with a sentinel behind the largest word
#define IB_TMP(call_null, call_real) { IB_HIP(call_null); if (need > tmp_bytes) { if (tmp) IB_HIP(hipFree(tmp)); tmp = nullptr; tmp_bytes = need; IB_HIP(hipMalloc(&tmp, tmp_bytes)); } IB_HIP(call_real); }
  b.maxpos = ntup > 0 ? (uint32_t)(ntup - 1) : 0;                        // hashidx.c:992
  IB_HIP(hipEventCreate(&e0)); IB_HIP(hipEventCreate(&e1));
  IB_HIP(hipMalloc((void **)&d_sop, ((size_t)nseq + 1) * 8));
  IB_HIP(hipMemcpy(d_sop, h_sop, ((size_t)nseq + 1) * 8, hipMemcpyHostToDevice));
  IB_HIP(hipMalloc((void **)&b.packed, nwords * 4));
  IB_HIP(hipMalloc((void **)&b.idx, ((size_t)b.nkeys + 2) * 4));
  IB_HIP(hipMalloc((void **)&d_nvalid, 8));
  if (k32) {
    IB_HIP(hipMalloc((void **)&key32, (ntup + 1) * 4));
    IB_HIP(hipMalloc((void **)&skey32, (ntup + 1) * 4));
    IB_HIP(hipMalloc((void **)&opos, (ntup + 1) * 4));
    IB_HIP(hipMalloc((void **)&b.pos, (ntup + 1) * 4));
  } else {
    IB_HIP(hipMalloc((void **)&flag, (ntup + 1) * 4));
    IB_HIP(hipMalloc((void **)&slot, (ntup + 1) * 4));
    IB_HIP(hipMalloc((void **)&key64, (ntup + 1) * 8));
  }
  IB_HIP(hipEventRecord(e0, 0));
  IB_HIP(hipMemsetAsync(d_nvalid, 0, 8, 0));
  IB_HIP(hipMemsetAsync(b.idx, 0, ((size_t)b.nkeys + 2) * 4, 0));
  hipLaunchKernelGGL(k_ib_pack, dim3(ib_grid(nwords)), dim3(256), 0, 0, d_ascii, tot, nwords, b.packed);
  if (k32) {
    hipLaunchKernelGGL(k_ib_kmers<true>, dim3(ib_grid(ntup)), dim3(256), (size_t)(256 * s + k + 16), 0, d_ascii, d_sop, nseq, ntup, k, s, b.typ, b.nbits_key, b.nbits_lo,
                       (uint32_t *)nullptr, (uint64_t *)nullptr, key32, opos, d_nvalid);
    IB_HIP(hipGetLastError());
    // stable sort by key: serials of equal keys stay ascending, serials that are not indexed end up behind
    IB_TMP(rocprim::radix_sort_pairs(nullptr, need, key32, skey32, opos, b.pos, (size_t)ntup, 0u, (unsigned)(2 * k + 1), 0),
           rocprim::radix_sort_pairs(tmp, need, key32, skey32, opos, b.pos, (size_t)ntup, 0u, (unsigned)(2 * k + 1), 0));
    IB_HIP(hipMemcpy(&h_nvalid, d_nvalid, 8, hipMemcpyDeviceToHost));
    n = h_nvalid;
    b.npos = (uint32_t)n;
    hipLaunchKernelGGL(k_ib_runend32, dim3(ib_grid(n)), dim3(256), 0, 0, skey32, n, b.idx);
    IB_HIP(hipGetLastError());
  } else {
    hipLaunchKernelGGL(k_ib_kmers<false>, dim3(ib_grid(ntup)), dim3(256), (size_t)(256 * s + k + 16), 0, d_ascii, d_sop, nseq, ntup, k, s, b.typ, b.nbits_key, b.nbits_lo,
                       flag, key64, (uint32_t *)nullptr, (uint32_t *)nullptr, d_nvalid);
    IB_HIP(hipGetLastError());
    // ordered compaction of the indexed serials
    IB_TMP(rocprim::exclusive_scan(nullptr, need, flag, slot, 0u, (size_t)ntup, rocprim::plus<uint32_t>(), 0),
           rocprim::exclusive_scan(tmp, need, flag, slot, 0u, (size_t)ntup, rocprim::plus<uint32_t>(), 0));
    if (ntup) {
      IB_HIP(hipMemcpy(&last_slot, slot + (ntup - 1), 4, hipMemcpyDeviceToHost));
      IB_HIP(hipMemcpy(&last_flag, flag + (ntup - 1), 4, hipMemcpyDeviceToHost));
    }
    n = (uint64_t)last_slot + last_flag;
    b.npos = (uint32_t)n;
    IB_HIP(hipMalloc((void **)&okey, (n + 1) * 8));
    IB_HIP(hipMalloc((void **)&skey, (n + 2) * 8));
    IB_HIP(hipMalloc((void **)&opos, (n + 1) * 4));
    IB_HIP(hipMalloc((void **)&b.pos, (n + 1) * 4));
    hipLaunchKernelGGL(k_ib_scatter, dim3(ib_grid(ntup)), dim3(256), 0, 0, flag, slot, key64, ntup, okey, opos);
    IB_HIP(hipGetLastError());
    {
      const unsigned end_bit = b.typ == IDX_PERFECT ? (unsigned)(2 * k) : (unsigned)(32 + b.nbits_key);
      IB_TMP(rocprim::radix_sort_pairs(nullptr, need, okey, skey, opos, b.pos, (size_t)n, 0u, end_bit, 0),
             rocprim::radix_sort_pairs(tmp, need, okey, skey, opos, b.pos, (size_t)n, 0u, end_bit, 0));
    }
    if (b.typ == IDX_PERFECT) {
      hipLaunchKernelGGL(k_ib_runend64, dim3(ib_grid(n)), dim3(256), 0, 0, skey, n, b.idx);
      IB_HIP(hipGetLastError());
    } else {
      uint32_t lh = 0, ls = 0;
      IB_HIP(hipMalloc((void **)&head, (n + 1) * 4));
      hipLaunchKernelGGL(k_ib_heads, dim3(ib_grid(n)), dim3(256), 0, 0, skey, n, head);
      IB_HIP(hipGetLastError());
      IB_TMP(rocprim::exclusive_scan(nullptr, need, head, slot, 0u, (size_t)n, rocprim::plus<uint32_t>(), 0),
             rocprim::exclusive_scan(tmp, need, head, slot, 0u, (size_t)n, rocprim::plus<uint32_t>(), 0));
      if (n) {
        IB_HIP(hipMemcpy(&ls, slot + (n - 1), 4, hipMemcpyDeviceToHost));
        IB_HIP(hipMemcpy(&lh, head + (n - 1), 4, hipMemcpyDeviceToHost));
      }
      b.nwords = ls + lh;
      IB_HIP(hipMalloc((void **)&b.wordidx, ((size_t)b.nwords + 2) * 4));
      IB_HIP(hipMalloc((void **)&b.posidx, ((size_t)b.nwords + 2) * 4));
      IB_HIP(hipMemsetAsync(b.wordidx, 0, ((size_t)b.nwords + 2) * 4, 0));
      IB_HIP(hipMemsetAsync(b.posidx, 0, ((size_t)b.nwords + 2) * 4, 0));
      hipLaunchKernelGGL(k_ib_words, dim3(ib_grid(n)), dim3(256), 0, 0, skey, head, slot, n, b.wordidx, b.posidx, b.idx);
      IB_HIP(hipGetLastError());
      IB_HIP(hipMemcpyAsync(b.posidx + b.nwords, &b.npos, 4, hipMemcpyHostToDevice, 0));       // posidx[nwords] = npos
    }
  }
  // run ends -> prefix sums: keys without elements take the count of their predecessor (running maximum)
  IB_TMP(rocprim::inclusive_scan(nullptr, need, b.idx, b.idx, (size_t)b.nkeys + 1, rocprim::maximum<uint32_t>(), 0),
         rocprim::inclusive_scan(tmp, need, b.idx, b.idx, (size_t)b.nkeys + 1, rocprim::maximum<uint32_t>(), 0));
  IB_HIP(hipEventRecord(e1, 0));
  IB_HIP(hipEventSynchronize(e1));
  IB_HIP(hipEventElapsedTime(&b.build_ms, e0, e1));
#undef IB_TMP
  }
  (void)hipFree(d_sop); (void)hipFree(flag); (void)hipFree(slot); (void)hipFree(key64); (void)hipFree(okey); (void)hipFree(skey);
  (void)hipFree(opos); (void)hipFree(head); (void)hipFree(tmp); (void)hipFree(key32); (void)hipFree(skey32); (void)hipFree(d_nvalid);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *out = b;
  return 0;
fail:
  (void)hipFree(d_sop); (void)hipFree(flag); (void)hipFree(slot); (void)hipFree(key64); (void)hipFree(okey); (void)hipFree(skey);
  (void)hipFree(opos); (void)hipFree(head); (void)hipFree(tmp); (void)hipFree(key32); (void)hipFree(skey32); (void)hipFree(d_nvalid);
  (void)hipFree(b.packed); (void)hipFree(b.pos); (void)hipFree(b.idx); (void)hipFree(b.wordidx); (void)hipFree(b.posidx);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  return -1;
}

}  // namespace smg
