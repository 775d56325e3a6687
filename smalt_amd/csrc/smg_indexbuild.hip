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

__global__ void __launch_bounds__(256) k_ib_codes(const uint8_t *ascii, uint64_t tot, uint8_t *codes, uint64_t ncodes) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < ncodes; i += (uint64_t)gridDim.x * 256)
    codes[i] = i < tot ? ref_code_of_ascii(ascii[i]) : (i == tot ? 7 : 0);      // terminator 7 after the last base
}

// compressSeq (sequence.c:1360-1424): 10 codes per word, first base in the highest bits
__global__ void __launch_bounds__(256) k_ib_pack(const uint8_t *codes, uint64_t nwords, uint32_t *packed) {
  for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < nwords; w += (uint64_t)gridDim.x * 256) {
    uint32_t v = 0;
    for (int i = 0; i < 10; i++) v |= (uint32_t)codes[w * 10 + (uint64_t)i] << (3 * (9 - i));
    packed[w] = v;
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

// one thread per k-mer serial: validity flag and sort key ((key << 32) | word_hi; PERFECT: key only)
__global__ void __launch_bounds__(256) k_ib_kmers(const uint8_t *codes, const uint64_t *sop, int nseq, uint64_t ntup, int k, int s, int typ,
                                                  int nbits_key, int nbits_lo, uint32_t *flag, uint64_t *key64) {
  for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < ntup; t += (uint64_t)gridDim.x * 256) {
    const uint64_t g = t * (uint64_t)s;
    int lo = 0, hi = nseq;                       // sequence of base g: last i with sop[i] <= g
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sop[mid] <= g) lo = mid; else hi = mid; }
    bool ok = g + (uint64_t)k <= sop[lo + 1];
    uint64_t word = 0;
    if (ok) {
      for (int j = 0; j < k; j++) { const uint32_t c = codes[g + (uint64_t)j]; if (c & 4) ok = false; word = (word << 2) | (c & 3); }
    }
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

__global__ void __launch_bounds__(256) k_ib_scatter(const uint32_t *flag, const uint32_t *slot, const uint64_t *key64, uint64_t ntup, uint64_t *okey, uint32_t *opos) {
  for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < ntup; t += (uint64_t)gridDim.x * 256)
    if (flag[t]) { okey[slot[t]] = key64[t]; opos[slot[t]] = (uint32_t)t; }
}

__global__ void __launch_bounds__(256) k_ib_count_perfect(const uint64_t *skey, uint64_t n, uint32_t *idx) {
  for (uint64_t a = (uint64_t)blockIdx.x * 256 + threadIdx.x; a < n; a += (uint64_t)gridDim.x * 256) atomicAdd(&idx[skey[a] + 1], 1u);
}

// HASH32MIX: heads of runs of equal (key, word_hi)
__global__ void __launch_bounds__(256) k_ib_heads(const uint64_t *skey, uint64_t n, uint32_t *head) {
  for (uint64_t a = (uint64_t)blockIdx.x * 256 + threadIdx.x; a < n; a += (uint64_t)gridDim.x * 256) head[a] = (a == 0 || skey[a] != skey[a - 1]) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_ib_words(const uint64_t *skey, const uint32_t *head, const uint32_t *slot, uint64_t n, uint32_t *wordidx, uint32_t *posidx, uint32_t *idx) {
  for (uint64_t a = (uint64_t)blockIdx.x * 256 + threadIdx.x; a < n; a += (uint64_t)gridDim.x * 256)
    if (head[a]) {
      wordidx[slot[a]] = (uint32_t)skey[a];
      posidx[slot[a]] = (uint32_t)a;
      atomicAdd(&idx[(uint32_t)(skey[a] >> 32) + 1], 1u);
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
  uint8_t *codes = nullptr;
  uint64_t *d_sop = nullptr, *key64 = nullptr, *okey = nullptr, *skey = nullptr;
  uint32_t *flag = nullptr, *slot = nullptr, *opos = nullptr, *head = nullptr;
  void *tmp = nullptr;
  size_t tmp_bytes = 0, need = 0;
  uint32_t last_slot = 0, last_flag = 0;
  uint64_t n = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (index_geometry(k, s, tot, &b.typ, &b.nbits_key, &b.nbits_lo) || b.nbits_key > 31 || nseq < 1) { snprintf(err, errlen, "unsupported index geometry (k=%d s=%d)", k, s); return -1; }
  for (int i = 0; i < nseq; i++) if (h_sop[i + 1] - h_sop[i] < (uint64_t)k) { snprintf(err, errlen, "sequence %d is shorter than the word length (hashidx.c:499)", i); return -1; }
  {
  b.nkeys = 1u << b.nbits_key;
  const uint64_t ntup = (tot + (uint64_t)s - 1) / (uint64_t)s;           // serials 0 .. ntup-1 start inside the reference
  const uint64_t nwords = tot / 10 + 1, ncodes = nwords * 10 + 32;
  b.maxpos = ntup > 0 ? (uint32_t)(ntup - 1) : 0;                        // hashidx.c:992
  IB_HIP(hipEventCreate(&e0)); IB_HIP(hipEventCreate(&e1));
  IB_HIP(hipMalloc((void **)&codes, ncodes));
  IB_HIP(hipMalloc((void **)&d_sop, ((size_t)nseq + 1) * 8));
  IB_HIP(hipMemcpy(d_sop, h_sop, ((size_t)nseq + 1) * 8, hipMemcpyHostToDevice));
  IB_HIP(hipMalloc((void **)&b.packed, nwords * 4));
  IB_HIP(hipMalloc((void **)&flag, (ntup + 1) * 4));
  IB_HIP(hipMalloc((void **)&slot, (ntup + 1) * 4));
  IB_HIP(hipMalloc((void **)&key64, (ntup + 1) * 8));
  IB_HIP(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_ib_codes, dim3(ib_grid(ncodes)), dim3(256), 0, 0, d_ascii, tot, codes, ncodes);
  hipLaunchKernelGGL(k_ib_pack, dim3(ib_grid(nwords)), dim3(256), 0, 0, codes, nwords, b.packed);
  hipLaunchKernelGGL(k_ib_kmers, dim3(ib_grid(ntup)), dim3(256), 0, 0, codes, d_sop, nseq, ntup, k, s, b.typ, b.nbits_key, b.nbits_lo, flag, key64);
  IB_HIP(hipGetLastError());
  // ordered compaction of the indexed serials
  IB_HIP(rocprim::exclusive_scan(nullptr, need, flag, slot, 0u, (size_t)ntup, rocprim::plus<uint32_t>(), 0));
  tmp_bytes = need; IB_HIP(hipMalloc(&tmp, tmp_bytes));
  IB_HIP(rocprim::exclusive_scan(tmp, need, flag, slot, 0u, (size_t)ntup, rocprim::plus<uint32_t>(), 0));
  if (ntup) {
    IB_HIP(hipMemcpy(&last_slot, slot + (ntup - 1), 4, hipMemcpyDeviceToHost));
    IB_HIP(hipMemcpy(&last_flag, flag + (ntup - 1), 4, hipMemcpyDeviceToHost));
  }
  n = (uint64_t)last_slot + last_flag;
  b.npos = (uint32_t)n;
  IB_HIP(hipMalloc((void **)&okey, (n + 1) * 8));
  IB_HIP(hipMalloc((void **)&skey, (n + 1) * 8));
  IB_HIP(hipMalloc((void **)&opos, (n + 1) * 4));
  IB_HIP(hipMalloc((void **)&b.pos, (n + 1) * 4));
  hipLaunchKernelGGL(k_ib_scatter, dim3(ib_grid(ntup)), dim3(256), 0, 0, flag, slot, key64, ntup, okey, opos);
  IB_HIP(hipGetLastError());
  IB_HIP(hipFree(key64)); key64 = nullptr;
  // stable sort by key: serials of equal keys stay ascending
  {
    const unsigned end_bit = b.typ == IDX_PERFECT ? (unsigned)(2 * k) : (unsigned)(32 + b.nbits_key);
    IB_HIP(rocprim::radix_sort_pairs(nullptr, need, okey, skey, opos, b.pos, (size_t)n, 0u, end_bit, 0));
    if (need > tmp_bytes) { IB_HIP(hipFree(tmp)); tmp = nullptr; tmp_bytes = need; IB_HIP(hipMalloc(&tmp, tmp_bytes)); }
    if (n) IB_HIP(rocprim::radix_sort_pairs(tmp, need, okey, skey, opos, b.pos, (size_t)n, 0u, end_bit, 0));
  }
  IB_HIP(hipMalloc((void **)&b.idx, ((size_t)b.nkeys + 2) * 4));
  IB_HIP(hipMemsetAsync(b.idx, 0, ((size_t)b.nkeys + 2) * 4, 0));
  if (b.typ == IDX_PERFECT) {
    hipLaunchKernelGGL(k_ib_count_perfect, dim3(ib_grid(n)), dim3(256), 0, 0, skey, n, b.idx);
    IB_HIP(hipGetLastError());
  } else {
    uint32_t lh = 0, ls = 0;
    IB_HIP(hipMalloc((void **)&head, (n + 1) * 4));
    hipLaunchKernelGGL(k_ib_heads, dim3(ib_grid(n)), dim3(256), 0, 0, skey, n, head);
    IB_HIP(hipGetLastError());
    IB_HIP(rocprim::exclusive_scan(nullptr, need, head, slot, 0u, (size_t)n, rocprim::plus<uint32_t>(), 0));
    if (need > tmp_bytes) { IB_HIP(hipFree(tmp)); tmp = nullptr; tmp_bytes = need; IB_HIP(hipMalloc(&tmp, tmp_bytes)); }
    if (n) {
      IB_HIP(rocprim::exclusive_scan(tmp, need, head, slot, 0u, (size_t)n, rocprim::plus<uint32_t>(), 0));
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
  // counts -> prefix sums: idx[j+1] += idx[j]
  IB_HIP(rocprim::inclusive_scan(nullptr, need, b.idx, b.idx, (size_t)b.nkeys + 1, rocprim::plus<uint32_t>(), 0));
  if (need > tmp_bytes) { IB_HIP(hipFree(tmp)); tmp = nullptr; tmp_bytes = need; IB_HIP(hipMalloc(&tmp, tmp_bytes)); }
  IB_HIP(rocprim::inclusive_scan(tmp, need, b.idx, b.idx, (size_t)b.nkeys + 1, rocprim::plus<uint32_t>(), 0));
  IB_HIP(hipEventRecord(e1, 0));
  IB_HIP(hipEventSynchronize(e1));
  IB_HIP(hipEventElapsedTime(&b.build_ms, e0, e1));
  }
  (void)hipFree(codes); (void)hipFree(d_sop); (void)hipFree(flag); (void)hipFree(slot); (void)hipFree(okey); (void)hipFree(skey);
  (void)hipFree(opos); (void)hipFree(head); (void)hipFree(tmp);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *out = b;
  return 0;
fail:
  (void)hipFree(codes); (void)hipFree(d_sop); (void)hipFree(flag); (void)hipFree(slot); (void)hipFree(key64); (void)hipFree(okey); (void)hipFree(skey);
  (void)hipFree(opos); (void)hipFree(head); (void)hipFree(tmp);
  (void)hipFree(b.packed); (void)hipFree(b.pos); (void)hipFree(b.idx); (void)hipFree(b.wordidx); (void)hipFree(b.posidx);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  return -1;
}

}  // namespace smg
