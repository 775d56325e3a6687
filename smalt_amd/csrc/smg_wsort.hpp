// smg_wsort.hpp -- the candidate ranking sort of segAliCandsStats (segment.c:1733, sort.c:233
// sortUINT32andINTarraysByQuickSort) executed by a whole wave with the reference's tie order.
//
// The reference ranks candidates with an unstable median-of-3 quicksort; the order it leaves equal
// keys in decides which candidates survive the depth cut, so it has to be reproduced exactly.  The
// permutation a Hoare partition produces is a function of the array only: pointer i stops at the
// successive positions holding a key >= pivot, pointer j at the successive positions (from the
// right) holding a key <= pivot, the t-th stops are swapped until the pointers cross.  A wave
// therefore reads 64 positions from either end, pairs the flagged positions by rank and swaps all
// pairs of a round at once; sub-ranges are independent, so ranges of at most WSORT_SMALL elements are
// finished one per lane with the sequential routine.  Ranges that start at or beyond `nneed`
// (everything behind the depth cut) are left unsorted: no later stage reads them.
//
// Elements are packed (key << IB) | index: 32-bit words with IB = 22 for reads up to 255 bases (keys are cover
// differences < 1024), 64-bit words with IB = 32 for longer reads.
#pragma once
#include "smg_exec.h"

namespace smg {

enum : int { WSORT_IDXBITS = 22, WSORT_SMALL = 32, WSORT_LISTCAP = 256, WSORT_LSTK = 16,
             WSORT_WORDS = 128 + 128 + 2 * WSORT_LISTCAP + 64 * WSORT_LSTK,   // LDS words of the wave sort
             WSORT_NBINS = 320 };
#define SMG_KVKEY(v) ((v) >> IB)       // IB: index bits of the packed element (template parameter of the routines below)

// the reference's routine on the sub-range [lo, hi] of a packed array; stk: >= 2*log2(hi-lo+1)+2 ints
template <int IB = WSORT_IDXBITS, class T = uint32_t>
SMG_HD inline void qsort_kv_range(T *a, int lo, int hi, int *stk) {
  int i, j, mid, sp = 0;
  T pv, t;
#define SMG_SWP(x, y) { t = a[x]; a[x] = a[y]; a[y] = t; }
  for (;;) {
    if (hi - lo < 7) {                                   // sort.c:240-251 insertion sort of short ranges
      for (j = lo + 1; j <= hi; j++) {
        pv = a[j];
        for (i = j - 1; i >= lo && SMG_KVKEY(a[i]) > SMG_KVKEY(pv); i--) a[i + 1] = a[i];
        a[i + 1] = pv;
      }
      if (!sp) return;
      hi = stk[--sp]; lo = stk[--sp];
    } else {
      mid = (lo + hi) >> 1;
      SMG_SWP(mid, lo + 1)
      if (SMG_KVKEY(a[lo]) > SMG_KVKEY(a[hi])) SMG_SWP(lo, hi)
      if (SMG_KVKEY(a[lo + 1]) > SMG_KVKEY(a[hi])) SMG_SWP(lo + 1, hi)
      if (SMG_KVKEY(a[lo]) > SMG_KVKEY(a[lo + 1])) SMG_SWP(lo, lo + 1)
      i = lo + 1; j = hi;
      pv = a[lo + 1];
      const T pk = SMG_KVKEY(pv);
      for (;;) {
        do i++; while (SMG_KVKEY(a[i]) < pk);
        do j--; while (SMG_KVKEY(a[j]) > pk);
        if (j < i) break;
        SMG_SWP(i, j)
      }
      a[lo + 1] = a[j]; a[j] = pv;
      if (hi - i + 1 >= j - lo) { stk[sp++] = i; stk[sp++] = hi; hi = j - 1; }     // larger part waits
      else { stk[sp++] = lo; stk[sp++] = j - 1; lo = i; }
    }
  }
#undef SMG_SWP
}

#if defined(__HIP_DEVICE_COMPILE__)
// One partition step of [lo, hi] (hi - lo >= 7) by the wave; returns the reference's final i and j.
template <int IB = WSORT_IDXBITS, class T = uint32_t>
__device__ inline void wave_partition_kv(T *a, int lo, int hi, uint32_t *pairs, int &out_i, int &out_j) {
  const int lane = (int)threadIdx.x;
  const uint64_t lt = (1ull << lane) - 1ull;
  if (lane == 0) {                                        // median of three to lo+1 (sort.c:252-262)
    T t;
    const int mid = (lo + hi) >> 1;
#define SMG_SWP(x, y) { t = a[x]; a[x] = a[y]; a[y] = t; }
    SMG_SWP(mid, lo + 1)
    if (SMG_KVKEY(a[lo]) > SMG_KVKEY(a[hi])) SMG_SWP(lo, hi)
    if (SMG_KVKEY(a[lo + 1]) > SMG_KVKEY(a[hi])) SMG_SWP(lo + 1, hi)
    if (SMG_KVKEY(a[lo]) > SMG_KVKEY(a[lo + 1])) SMG_SWP(lo, lo + 1)
#undef SMG_SWP
  }
  __syncthreads();
  const T pv = a[lo + 1], pk = SMG_KVKEY(pv);
  int L = lo + 2, R = hi - 1;                             // next unread position of either pointer
  uint64_t mA = 0, mB = 0;                                // unconsumed stops of i (bit l = baseA + l) and j (baseB - l)
  int baseA = 0, baseB = 0, prevA = -1, prevB = -1, fi = 0, fj = 0;
  for (;;) {
    while (!mA) {                                         // a[hi] >= pivot: a stop always exists
      const int p = L + lane;
      baseA = L; L += 64;
      mA = __ballot(p <= hi && SMG_KVKEY(a[p <= hi ? p : hi]) >= pk);
    }
    while (!mB) {                                         // a[lo+1] == pivot
      const int p = R - lane;
      baseB = R; R -= 64;
      mB = __ballot(p >= lo + 1 && SMG_KVKEY(a[p >= lo + 1 ? p : lo + 1]) <= pk);
    }
    const int cA = __popcll(mA), cB = __popcll(mB), m = cA < cB ? cA : cB;
    if ((mA >> lane) & 1ull) pairs[__popcll(mA & lt)] = (uint32_t)lane;
    if ((mB >> lane) & 1ull) pairs[64 + __popcll(mB & lt)] = (uint32_t)lane;
    __syncthreads();
    const int la = (int)pairs[lane], lb = (int)pairs[64 + lane];
    const int posA = baseA + la, posB = baseB - lb;
    const bool inpair = lane < m, valid = inpair && posA <= posB;
    const uint64_t minv = __ballot(inpair && !valid);
    uint64_t mstrict = __ballot(valid && posA < posB);    // real swaps (a pair with posA == posB swaps nothing)
    if (valid && posA != posB) { const T u = a[posA], w = a[posB]; a[posA] = w; a[posB] = u; }
    __syncthreads();
    if (minv) mstrict &= (1ull << __builtin_ctzll(minv)) - 1ull;
    if (mstrict) { const int ls = 63 - __builtin_clzll(mstrict); prevA = __shfl(posA, ls); prevB = __shfl(posB, ls); }
    if (minv) {                                           // pointers crossed at pair f
      const int f = __builtin_ctzll(minv);
      fi = __shfl(posA, f); fj = __shfl(posB, f);
      if (prevA >= 0) {                                   // the last real swap left a stop for either pointer
        if (prevB < fi) fi = prevB;
        if (prevA > fj) fj = prevA;
      }
      break;
    }
    const int cutA = __shfl(la, m - 1), cutB = __shfl(lb, m - 1);
    mA = (m == cA) ? 0ull : (mA & ~((2ull << cutA) - 1ull));
    mB = (m == cB) ? 0ull : (mB & ~((2ull << cutB) - 1ull));
  }
  if (lane == 0) { a[lo + 1] = a[fj]; a[fj] = pv; }
  __syncthreads();
  out_i = fi; out_j = fj;
}
#endif

// Sort a[0..n) like the reference does, at least up to position nneed.  wk: WSORT_WORDS words (LDS).
// wk: 256 + 2 * LISTCAP + 64 * LSTK words (WSORT_WORDS with the defaults); LSTK >= 2 * log2(WSORT_SMALL) + 2
template <int IB = WSORT_IDXBITS, int LISTCAP = WSORT_LISTCAP, int LSTK = WSORT_LSTK, class T = uint32_t>
SMG_HD inline void wave_sort_kv(T *a, int n, int nneed, uint32_t *wk) {
  if (n < 2) return;
#if defined(__HIP_DEVICE_COMPILE__)
  const int lane = (int)threadIdx.x;
  uint32_t *pairs = wk;
  int *bstk = (int *)(wk + 128), *slist = (int *)(wk + 256), *lstk = (int *)(wk + 256 + 2 * LISTCAP) + lane * LSTK;
  int nb = 0, ns = 0;
  int lo = 0, hi = n - 1;
  bool have = true;
  for (;;) {
    if (!have) {
      if (!nb) break;
      __syncthreads();
      nb--; lo = bstk[2 * nb]; hi = bstk[2 * nb + 1];
      __syncthreads();
    }
    have = false;
    if (lo >= nneed || hi <= lo) continue;
    if (hi - lo + 1 <= WSORT_SMALL) {
      slist[2 * ns] = lo; slist[2 * ns + 1] = hi; ns++;   // every lane writes the same words
      if (ns == LISTCAP) {
        __syncthreads();
        for (int t = lane; t < ns; t += 64) qsort_kv_range<IB, T>(a, slist[2 * t], slist[2 * t + 1], lstk);
        __syncthreads();
        ns = 0;
      }
      continue;
    }
    int i, j;
    wave_partition_kv<IB, T>(a, lo, hi, pairs, i, j);
    // [lo, j-1] and [i, hi]: keep the smaller one, park the larger (bounded stack)
    int plo, phi;
    if (hi - i + 1 >= j - lo) { plo = i; phi = hi; hi = j - 1; }
    else { plo = lo; phi = j - 1; lo = i; }
    if (nb < 64) { bstk[2 * nb] = plo; bstk[2 * nb + 1] = phi; nb++; }
    have = true;
  }
  __syncthreads();
  for (int t = lane; t < ns; t += 64) qsort_kv_range<IB, T>(a, slist[2 * t], slist[2 * t + 1], lstk);
  __syncthreads();
#else
  (void)nneed; (void)wk;
  int stk[128];
  qsort_kv_range<IB, T>(a, 0, n - 1, stk);
#endif
}

}  // namespace smg
