// smg_logic.hpp -- per-read sequential logic of the seed-and-extend path, written once as
// __host__ __device__ functions over flat, fixed-capacity arrays.  On the GPU each function
// runs on ONE lane of the wave that owns the read (the reference is sequential here and the
// outcome depends on its exact order of operations); the wide, order-free parts (k-mer
// lookups, hit gather, sorting of hit words, window scoring) live in the kernels.
// `file:line` citations refer to the reference tree (SMALT 0.7.6, src/).
#pragma once
#include "smg_common.h"

namespace smg {

template <class T> SMG_HD inline T tmin(T a, T b) { return a < b ? a : b; }
template <class T> SMG_HD inline T tmax(T a, T b) { return a > b ? a : b; }

// ---------------------------------------------------------------------------------------
// I1: k-mer lookup (hashidx.c:155-172, 1146-1191)
// ---------------------------------------------------------------------------------------
SMG_HD inline uint32_t hash32mix(uint32_t a) {
  a = (a + 0x7ed55d16u) + (a << 12);
  a = (a ^ 0xc761c23cu) ^ (a >> 19);
  a = (a + 0x165667b1u) + (a << 5);
  a = (a + 0xd3a2646cu) ^ (a << 9);
  a = (a + 0xfd7046c5u) + (a << 3);
  a = (a ^ 0xb55a4f09u) ^ (a >> 16);
  return a;
}

// returns number of hits; *posidx identifies the position block
SMG_HD inline uint32_t index_lookup(const DevIndex &ix, uint64_t word, uint32_t *posidx) {
  const uint64_t wordmask = (1ull << (2 * ix.k)) - 1;
  if (ix.typ == IDX_PERFECT) {
    uint32_t key = (uint32_t)(word & wordmask);
    *posidx = key;
    return ix.idx[key + 1] - ix.idx[key];
  }
  const uint64_t mask_lo = (1ull << ix.nbits_lo) - 1;
  const uint32_t keymod = 1u << (ix.nbits_key - ix.nbits_lo);
  uint32_t word_hi = (uint32_t)((word & wordmask & ~mask_lo) >> ix.nbits_lo);
  uint32_t key = ((hash32mix(word_hi) % keymod) << ix.nbits_lo) + (uint32_t)(word & mask_lo);
  uint32_t b = ix.idx[key + 1];
  if (b < 1) return 0;
  uint32_t a = ix.idx[key];
  b--;
  while (a < b) {
    uint32_t pivot = (a + b) >> 1;
    if (ix.wordidx[pivot] < word_hi) a = pivot + 1; else b = pivot;
  }
  if (a == b && ix.wordidx[b] == word_hi) {
    *posidx = b;
    return ix.posidx[b + 1] - ix.posidx[b];
  }
  return 0;
}

// hashidx.c:1193-1212
SMG_HD inline uint32_t index_positions(const DevIndex &ix, uint32_t posidx, const uint32_t **posp) {
  if (ix.typ == IDX_PERFECT) {
    *posp = ix.pos + ix.idx[posidx];
    return ix.idx[posidx + 1] - ix.idx[posidx];
  }
  *posp = ix.pos + ix.posidx[posidx];
  return ix.posidx[posidx + 1] - ix.posidx[posidx];
}

// I2: 3-bit code of reference base at concatenated offset o, as the scalar path sees it after
// uncompressSeq + seqFastqEncode (sequence.c:1499-1550): 6/4 -> N(5), terminator 7 -> 0.
SMG_HD inline uint32_t ref_code(const uint32_t *packed, uint64_t o) {
  uint32_t w = packed[o / 10];
  uint32_t c = (w >> (3 * (9 - (uint32_t)(o % 10)))) & 7u;
  return (c == 7) ? 0u : ((c == 6 || c == 4) ? 5u : c);
}

// ---------------------------------------------------------------------------------------
// sort.c:233-330 -- two-array quicksort, median of three, insertion below 7; the order of
// equal keys is a property of this exact exchange sequence.
// ---------------------------------------------------------------------------------------
SMG_HD inline void sort2_u32(int n, uint32_t *key, uint32_t *val) {
  int lo = 0, hi = n - 1, i, j, mid, sp = 0;
  int stk[64];
  uint32_t pk, pv, t;
#define SMG_SWP(a, b) { t = key[a]; key[a] = key[b]; key[b] = t; t = val[a]; val[a] = val[b]; val[b] = t; }
  for (;;) {
    if (hi - lo < 7) {
      for (j = lo + 1; j <= hi; j++) {
        pk = key[j]; pv = val[j];
        for (i = j - 1; i >= lo && key[i] > pk; i--) { key[i + 1] = key[i]; val[i + 1] = val[i]; }
        key[i + 1] = pk; val[i + 1] = pv;
      }
      if (!sp) return;
      hi = stk[sp--]; lo = stk[sp--];
    } else {
      mid = (lo + hi) >> 1;
      SMG_SWP(mid, lo + 1)
      if (key[lo] > key[hi]) SMG_SWP(lo, hi)
      if (key[lo + 1] > key[hi]) SMG_SWP(lo + 1, hi)
      if (key[lo] > key[lo + 1]) SMG_SWP(lo, lo + 1)
      i = lo + 1; j = hi;
      pk = key[lo + 1]; pv = val[lo + 1];
      for (;;) {
        do i++; while (key[i] < pk);
        do j--; while (key[j] > pk);
        if (j < i) break;
        SMG_SWP(i, j)
      }
      key[lo + 1] = key[j]; key[j] = pk;
      val[lo + 1] = val[j]; val[j] = pv;
      sp += 2;
      if (sp > 60) return;
      if (hi - i + 1 >= j - lo) { stk[sp] = hi; stk[sp - 1] = i; hi = j - 1; }
      else { stk[sp] = j - 1; stk[sp - 1] = lo; lo = i; }
    }
  }
#undef SMG_SWP
}

// ---------------------------------------------------------------------------------------
// S2: seed budget (hashhit.c:769-891 getHitInfoMaxRank, 1028-1079)
//   sortkey[n_seeds] (= nhits, rarity-sorted), qoffs_by_rank[i] = read offset of rank i.
//   frame_cnt[s], frame_rank[s*stride]: scratch; qbuf[qlen]: scratch bytes.
// ---------------------------------------------------------------------------------------
SMG_HD inline void build_frames(uint32_t n_seeds, const uint32_t *qoffs_by_rank, int s, uint32_t *frame_cnt,
                                uint32_t *frame_rank, uint32_t stride) {
  for (int f = 0; f < s; f++) frame_cnt[f] = 0;
  for (uint32_t i = 0; i < n_seeds; i++) {
    uint32_t f = qoffs_by_rank[i] % (uint32_t)s;
    frame_rank[f * stride + frame_cnt[f]++] = i;
  }
}

SMG_HD inline uint32_t seed_max_rank(uint32_t n_seeds, const uint32_t *sortkey, const uint32_t *qoffs_by_rank, int k, int s,
                                     uint32_t qlen, uint32_t mincover, uint32_t maxcover, uint32_t maxhit,
                                     const uint32_t *frame_cnt, const uint32_t *frame_rank, uint32_t stride, uint8_t *qbuf) {
  uint32_t i, ntot = sortkey[0];
  for (i = 1; i <= n_seeds && ntot <= maxhit; i++)
    if (i < n_seeds) ntot += sortkey[i];
  uint32_t n = i - 1, nmax = n;
  for (int f = 0; f < s; f++) {
    uint32_t imax = frame_cnt[f];
    if (!imax) continue;
    const uint32_t *ixp = frame_rank + (uint32_t)f * stride;
    for (uint32_t q = 0; q < qlen; q++) qbuf[q] = 0;
    uint32_t cover = 0;
    for (i = 0; i < imax && cover <= maxcover && (cover < mincover || ixp[i] <= n); i++) {
      uint32_t q0 = qoffs_by_rank[ixp[i]];
      for (uint32_t q = q0; q < q0 + (uint32_t)k - 1; q++)        // k-1 bases (hashhit.c:873)
        if (!qbuf[q]) { qbuf[q] = 1; cover++; }
    }
    if (i > 0 && ixp[i - 1] > nmax) nmax = ixp[i - 1];
  }
  if (nmax < (uint32_t)HITINFO_MINSEEDNUM) return ((uint32_t)HITINFO_MINSEEDNUM < n_seeds) ? (uint32_t)HITINFO_MINSEEDNUM : n_seeds;
  return nmax;
}

// hashhit.c:1096-1169 (hashCalcHitInfoCoverDeficit)
SMG_HD inline uint32_t cover_deficit(uint32_t status, uint32_t seed_rank, uint32_t qlen, const uint8_t *qmask,
                                     const uint32_t *qoffs_by_rank, int k, int s, const uint32_t *frame_cnt,
                                     const uint32_t *frame_rank, uint32_t stride, uint8_t *qbuf) {
  uint32_t deficit, d, i;
  if (status & HI_RANK) {
    uint32_t maxcover = 0;
    d = qlen;
    for (int f = 0; f < s; f++) {
      uint32_t imax = frame_cnt[f];
      if (!imax) continue;
      const uint32_t *ixp = frame_rank + (uint32_t)f * stride;
      for (uint32_t q = 0; q < qlen; q++) qbuf[q] = 0;
      uint32_t cover = 0;
      for (i = 0; i < imax && ixp[i] < seed_rank; i++) {
        uint32_t q0 = qoffs_by_rank[ixp[i]];
        for (uint32_t q = q0; q < q0 + (uint32_t)k; q++)
          if (!qbuf[q]) { qbuf[q] = 1; cover++; }
      }
      if (cover < d) d = cover;
      if (cover > maxcover) maxcover = cover;
    }
    deficit = maxcover - d + 1;
  } else {
    uint8_t kk = (uint8_t)(k / s);
    if (kk > 0) kk--;
    deficit = 0;
    for (int f = 0; f < s; f++) {
      uint8_t ctr = 0;
      d = 0;
      for (i = (uint32_t)f; i < qlen; i += (uint32_t)s) {
        if (qmask[i] == HQ_NORMHIT) ctr = kk;
        else if (ctr) ctr--;
        else d += (uint32_t)s;
      }
      if (d > deficit) deficit = d;
    }
  }
  return deficit;
}

// SET_NEXT_SHIFT (hashhit.c:283-288): diagonal part of the packed hit word
SMG_HD inline uint64_t hit_diag(bool is_reverse, uint32_t pos, uint32_t q, int s) {
  const uint64_t offbit = 1ull << 32;
  return is_reverse ? ((uint64_t)pos + q / (uint32_t)s) : ((((uint64_t)pos) | offbit) - q / (uint32_t)s);
}

// the same with the division by the stride as a multiplication: magic = ceil(2^32 / s) (0 for s == 1), exact for
// read offsets below 2^20 (the error of q * magic / 2^32 is below 2^-12 < 1 / s)
SMG_HD inline uint32_t div_magic(int s) { return s > 1 ? (uint32_t)((0x100000000ull + (uint32_t)s - 1) / (uint32_t)s) : 0u; }
SMG_HD inline uint64_t hit_diag_m(bool is_reverse, uint32_t pos, uint32_t q, uint32_t magic) {
  const uint64_t offbit = 1ull << 32;
  const uint32_t qs = magic ? (uint32_t)(((uint64_t)q * magic) >> 32) : q;
  return is_reverse ? ((uint64_t)pos + qs) : ((((uint64_t)pos) | offbit) - qs);
}

SMG_HD inline uint32_t lower_bound_u32(const uint32_t *a, uint32_t n, uint32_t v) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (a[m] < v) lo = m + 1; else hi = m; }
  return lo;
}

// both lower bounds of a half-open range [lo, hi) in a sorted position list.  Lists of up to 32 positions (nearly all: a k-mer of
// a 3 Gbp reference has a dozen positions) are counted through with independent loads -- one round trip instead of the eight to
// ten dependent ones of two binary searches, which is what an interval-restricted call spends its time on
SMG_HD inline void lower_bounds2_u32(const uint32_t *a, uint32_t n, uint32_t lo, uint32_t hi, uint32_t *at_lo, uint32_t *at_hi) {
  if (n <= 32) {
    uint32_t below_lo = 0, below_hi = 0;
    for (uint32_t j0 = 0; j0 < n; j0 += 16) {          // sixteen loads in flight
      uint32_t v[16];
#pragma unroll
      for (uint32_t u = 0; u < 16; u++) v[u] = j0 + u < n ? a[j0 + u] : 0xffffffffu;
#pragma unroll
      for (uint32_t u = 0; u < 16; u++) { below_lo += v[u] < lo ? 1u : 0u; below_hi += v[u] < hi ? 1u : 0u; }
    }
    *at_lo = below_lo; *at_hi = below_hi;
    return;
  }
  *at_lo = lower_bound_u32(a, n, lo);
  *at_hi = *at_lo + lower_bound_u32(a + *at_lo, n - *at_lo, hi);
}

// ---------------------------------------------------------------------------------------
// S3 slow path: exact replay of hashCollectHitsForSegment's retry protocol
// (hashhit.c:1416-1546, 1730-1741) for one (strand, sequence): decides which ranked seeds
// contribute hits.  Returns the number of leading ranks used (abort point) and *m_final, the
// per-seed hit ceiling of the final attempt (0 = none).
// ---------------------------------------------------------------------------------------
struct FillDecision { uint32_t n_used; uint32_t m_final; };

SMG_HD inline FillDecision fill_decide(const DevIndex &ix, const SeedRec *seeds, uint32_t n_seeds, uint32_t lo, uint32_t hi,
                                       uint32_t nhit_max, int nhits_alloc, uint8_t *qmask) {
  FillDecision d;
  uint32_t m = nhit_max;
  for (;;) {
    uint32_t total = 0, n;
    bool aborted = false;
    for (n = 0; n < n_seeds; n++) {
      const SeedRec &sp = seeds[n];
      if (m > 0 && sp.nhits > m) { qmask[sp.qoffs] = HQ_MULTIHIT; continue; }
      const uint32_t *posp;
      uint32_t nhits = index_positions(ix, sp.posidx, &posp);
      uint32_t a = lower_bound_u32(posp, nhits, lo);
      uint32_t nh = nhits - a;
      if (nh == 0) continue;                         // "posp[nhits-1] < segpos_lo -> continue"
      if (total + nh > (uint32_t)nhits_alloc) {
        if (m > 0) { aborted = true; break; }
        qmask[sp.qoffs] = HQ_MULTIHIT;
        continue;
      }
      uint32_t b = lower_bound_u32(posp, nhits, hi);
      total += b - a;
    }
    d.n_used = n;
    d.m_final = m;
    m /= 2;
    if (!(aborted && m > (uint32_t)MINHIT_PER_TUPLE)) break;
  }
  return d;
}

// ---------------------------------------------------------------------------------------
// S4: sorted hit words -> hit regions -> seeds -> constant-shift segments
// (segment.c:396-584, 763-810).  `dat[nhits]`: packed words (diagonal<<31 | q), ascending.
// ---------------------------------------------------------------------------------------
struct SegLst {
  HitRegion *hreg; SegSeed *seed; Segment *segm;
  uint32_t nhreg, nseed, nsegm, cap;
};

SMG_HD inline int seglst_fill(SegLst &sl, uint32_t min_ktup, const uint64_t *dat, int nhits, uint32_t qlen,
                              const uint8_t *hl_qmask /* may be null: all NOHIT */, int k, int s) {
  sl.nhreg = sl.nseed = sl.nsegm = 0;
  // segment.c:781-788: min_ktup is reduced once per non-NORMHIT offset of the HIT LIST's mask
  if (hl_qmask) {
    for (const uint8_t *qm = hl_qmask; *qm; qm++) {
      if (*qm == HQ_NORMHIT) continue;
      if (min_ktup < 2) break;
      min_ktup--;
    }
  } else {
    for (uint32_t q = 0; q < qlen; q++) { if (min_ktup < 2) break; min_ktup--; }
  }
  if (nhits >= 1) {
    uint32_t max_dshift = (uint32_t)(k * SEGMENTING_DIFFSHIFT / s) & 0xffffu;
    uint32_t ds = (qlen - (uint32_t)k) / (uint32_t)s + 1;
    if (ds < max_dshift) max_dshift = ds & 0xffffu;
    const uint64_t dsthresh = ((uint64_t)max_dshift) << HALFBIT;
    for (int i = 0; i < nhits;) {
      int j;
      for (j = i + 1; j < nhits; j++)
        if (dat[j] - dat[j - 1] >= dsthresh) break;
      if ((uint32_t)(j - i) >= min_ktup) {
        if (sl.nhreg >= sl.cap) return -1;
        sl.hreg[sl.nhreg].idx = (uint32_t)i;
        sl.hreg[sl.nhreg].num = j - i;
        sl.nhreg++;
      }
      i = j;
    }
  }
  for (uint32_t r = 0; r < sl.nhreg; r++) {              // makeSeedsFromHits
    uint32_t a = sl.hreg[r].idx, b, end = a + (uint32_t)sl.hreg[r].num;
    sl.hreg[r].idx = sl.nseed;
    while (a < end) {
      uint64_t shift = dat[a] & ~HALFMASK;
      uint32_t qoffs = (uint32_t)(dat[a] & HALFMASK), lastq = qoffs + (uint32_t)k, qo;
      for (b = a + 1; b < end; b++) {
        if ((dat[b] & ~HALFMASK) != shift) break;
        qo = (uint32_t)(dat[b] & HALFMASK);
        if (qo > lastq || ((qo - qoffs) % (uint32_t)s)) break;
        lastq = qo + (uint32_t)k;
      }
      if (sl.nseed >= sl.cap) return -1;
      sl.seed[sl.nseed].sqo = dat[a];
      sl.seed[sl.nseed].len = (int32_t)(lastq - qoffs);
      sl.nseed++;
      a = b;
    }
    sl.hreg[r].num = (int32_t)(sl.nseed - sl.hreg[r].idx);
  }
  for (uint32_t r = 0; r < sl.nhreg; r++) {              // makeSegmentsFromSeeds
    uint32_t a = sl.hreg[r].idx, b, end = a + (uint32_t)sl.hreg[r].num;
    sl.hreg[r].idx = sl.nsegm;
    sl.hreg[r].num = 0;
    while (a < end) {
      uint64_t shift = sl.seed[a].sqo & ~HALFMASK;
      uint32_t qoffs = (uint32_t)(sl.seed[a].sqo & HALFMASK);
      uint32_t cover = (uint32_t)sl.seed[a].len;
      for (b = a + 1; b < end; b++) {
        if ((sl.seed[b].sqo & ~HALFMASK) != shift || (((uint32_t)(sl.seed[b].sqo & HALFMASK)) - qoffs) % (uint32_t)s) break;
        cover += (uint32_t)sl.seed[b].len;
      }
      if (sl.nsegm >= sl.cap) return -1;
      sl.segm[sl.nsegm].ix = a;
      sl.segm[sl.nsegm].nseed = (int32_t)(b - a);
      sl.segm[sl.nsegm].cover = cover;
      sl.nsegm++;
      sl.hreg[r].num++;
      a = b;
    }
  }
  return 0;
}

// calcSegmentBoundaries (segment.c:635-668)
SMG_HD inline void segment_bounds(uint32_t *qs, uint32_t *qe, uint32_t *rs, uint32_t *re, const Segment &sg,
                                  const SegSeed *seedr, int k, int s, bool is_reverse) {
  const SegSeed &a = seedr[sg.ix];
  const SegSeed &b = seedr[sg.ix + (uint32_t)sg.nseed - 1];
  *qs = (uint32_t)(a.sqo & HALFMASK);
  *qe = (uint32_t)(b.sqo & HALFMASK) + (uint32_t)b.len - 1;
  if (is_reverse) {
    *rs = (uint32_t)(((b.sqo >> HALFBIT) - (b.sqo & HALFMASK) / (uint64_t)s) & SOFFSMASK);
    *rs -= (uint32_t)((b.len - k) / s);
    *re = (uint32_t)(((a.sqo >> HALFBIT) - (uint64_t)(*qs) / (uint64_t)s) & SOFFSMASK);
  } else {
    *rs = (uint32_t)(((a.sqo >> HALFBIT) + (uint64_t)(*qs) / (uint64_t)s) & SOFFSMASK);
    *re = (uint32_t)(((b.sqo >> HALFBIT) + (b.sqo & HALFMASK) / (uint64_t)s) & SOFFSMASK);
    *re += (uint32_t)((b.len - k) / s);
  }
}

// derriveSEGCAND (segment.c:929-1059)
SMG_HD inline int derive_cand(SegCand &c, int first, int nseg, Segment *segbase, const SegSeed *seedr, int k, int s,
                              uint32_t cover, uint32_t mincover_noindel, uint32_t hregix, bool is_reverse) {
  const uint64_t offbit = 1ull << (HALFBIT + 1);
  Segment *sg0 = segbase + first, *sg = sg0 + 1;
  if (sg0->nseed < 0) return -1;
  segment_bounds(&c.qs, &c.qe, &c.rs, &c.re, *sg0, seedr, k, s, is_reverse);
  sg0->nseed *= -1;
  int64_t shift_min = (int64_t)(seedr[sg0->ix].sqo >> HALFBIT), shift_2mm = shift_min, shift_start;
  uint32_t maxcover = sg0->cover, qs, qe, rs, re;
  for (int n = 1; n < nseg; n++, sg++) {
    if (sg->nseed < 0) return -1;
    segment_bounds(&qs, &qe, &rs, &re, *sg, seedr, k, s, is_reverse);
    if (sg->cover > maxcover) { shift_2mm = (int64_t)(seedr[sg->ix].sqo >> HALFBIT); maxcover = sg->cover; }
    sg->nseed *= -1;
    if (qs < c.qs) c.qs = qs;
    if (qe > c.qe) c.qe = qe;
    if (rs < c.rs) c.rs = rs;
    if (re > c.re) c.re = re;
  }
  sg--;
  uint8_t flag = 0;
  if (is_reverse) {
    flag |= CANDFLG_REVERSE;
    shift_start = ((int64_t)c.rs) + (int64_t)((c.qe - (uint32_t)k + 1) / (uint32_t)s);
  } else {
    shift_start = (int64_t)((((uint64_t)c.rs) | offbit) - (uint64_t)(c.qs / (uint32_t)s));
  }
  uint64_t shift_range = (uint64_t)(((int64_t)(seedr[sg->ix].sqo >> HALFBIT)) - shift_min);
  int64_t diff_shift = shift_min - shift_start;
  if (shift_range > 32767) return -1;
  if (diff_shift < -32768 || diff_shift > 32767) return -1;
  c.shiftoffs = (int16_t)diff_shift;
  if (maxcover >= mincover_noindel) {
    int64_t ds = shift_2mm - shift_start;
    flag |= CANDFLG_MMALI;
    if (ds < -32768 || ds > 32767) return -1;
    c.shift2mm = (int16_t)ds;
  } else {
    c.shift2mm = 0;
  }
  c.flag = flag; c.pad = 0;
  c.srange = (int16_t)shift_range;
  c.cover = cover;
  c.nseg = nseg;
  c.hregix = hregix;
  c.seqidx = -1;
  return 0;
}

struct CandSet { SegCand *cand; uint32_t ncand, cap; uint32_t max_cover, max2nd_cover; };

// S5: segAliCandsAddFast -> addCandsFast (segment.c:1140-1223); mask: qlen scratch bytes
SMG_HD inline int cands_add_fast(CandSet &cs, uint8_t *mask, SegLst &sl, uint32_t qlen, int k, int s, bool is_reverse,
                                 uint32_t mincover, int32_t seqidx) {
  for (uint32_t r = 0; r < sl.nhreg; r++) {
    const HitRegion hr = sl.hreg[r];
    Segment *base = sl.segm + hr.idx;
    for (int i = 0; i < hr.num;) {
      Segment *sg = base + i;
      for (uint32_t q = 0; q < qlen; q++) mask[q] = 0;         // INIT_COVERAGE_CALC
      for (int l = 0; l < sg->nseed; l++) {
        const SegSeed &sd = sl.seed[sg->ix + (uint32_t)l];
        uint8_t *u = mask + (sd.sqo & HALFMASK);
        for (int q = 0; q < sd.len; q++) u[q] = 1;
      }
      uint32_t cover = sg->cover;
      int j;
      sg++;
      for (j = i + 1; j < hr.num; j++, sg++) {
        if (sg->nseed < 0) break;
        uint32_t cover_new = 0;                                  // CALC_COVERAGE
        for (int l = 0; l < sg->nseed; l++) {
          const SegSeed &sd = sl.seed[sg->ix + (uint32_t)l];
          uint8_t *u = mask + (sd.sqo & HALFMASK);
          for (int q = 0; q < sd.len; q++) if (!u[q]) { cover_new++; u[q] = 1; }
        }
        if ((cover_new << 1) < sg->cover && cover >= mincover) break;
        cover += cover_new;
      }
      if (cover >= mincover) {
        if (cs.ncand >= cs.cap) return -2;
        SegCand &c = cs.cand[cs.ncand];
        if (derive_cand(c, i, j - i, base, sl.seed, k, s, cover, mincover, r, is_reverse)) return -1;
        cs.ncand++;
        c.seqidx = seqidx;
        if (cover > cs.max2nd_cover) {
          if (cover > cs.max_cover) { cs.max2nd_cover = cs.max_cover; cs.max_cover = cover; }
          else if (cover != cs.max_cover) cs.max2nd_cover = cover;
        }
      }
      i = j;
    }
  }
  return 0;
}

// S6: segAliCandsStats (segment.c:1616-1785).  sort_keys/sort_idx: capacity >= ncand.
SMG_HD inline int cands_stats(const CandSet &cs, uint32_t cdf_fwd /* cover_deficit[0] */, int s, uint32_t min_cover_below_max,
                              uint32_t target_depth, uint32_t max_depth, bool is_sensitive, uint32_t *sort_keys,
                              uint32_t *sort_idx, uint32_t *n_mincover, uint32_t *n_sort) {
  const SegCand *scp = cs.cand;
  if (max_depth < 1 || max_depth > (uint32_t)MAXIMUM_DEPTH) max_depth = MAXIMUM_DEPTH;
  if (target_depth < 1) target_depth = DEFAULT_TARGET_DEPTH;
  if (target_depth > max_depth) target_depth = max_depth;
  uint32_t min_cover = (min_cover_below_max > cs.max_cover) ? 0 : cs.max_cover - min_cover_below_max;
  uint32_t cdf = 0;
  if (min_cover > cs.max2nd_cover) { cdf = min_cover - cs.max2nd_cover; min_cover = cs.max2nd_cover; }
  uint32_t adj = (cdf_fwd > cdf) ? cdf_fwd - cdf : 0;   // deficit of strand [0] for both strands (:1676)
  uint32_t i, j;
  for (i = j = 0; i < cs.ncand; i++) {
    if (scp[i].cover + adj < min_cover) continue;
    if (scp[i].cover > cs.max_cover) return -1;
    sort_keys[j] = cs.max_cover - scp[i].cover;
    sort_idx[j] = i;
    j++;
  }
  sort2_u32((int)j, sort_keys, sort_idx);
  *n_mincover = j;
  if (j > target_depth) {
    uint32_t maxj = (j < max_depth) ? j : max_depth;
    if (is_sensitive) {
      for (j = target_depth; j < maxj; j++)      // scp[j] rather than scp[sort_idx[j]] (:1761-1762)
        if (sort_keys[j] >= adj) break;
      for (; j < *n_mincover && sort_keys[j] < (uint32_t)s; j++) {}
    } else {
      uint32_t cov = sort_keys[j / 2];
      if (cov < (uint32_t)s) cov = (uint32_t)s;
      for (j = target_depth; j < maxj && sort_keys[j] < cov; j++) {}
    }
  }
  *n_sort = j;
  return 0;
}

// S7: segAliCandsCalcSegmentOffsets (segment.c:1861-1985) with edgelen = 0 (rmap.c:548-552),
// plus the kernel-selection predicate of scoreRMAPCAND (rmap.c:715-718).
SMG_HD inline int cand_offsets(RCand &c, const SegCand &p, const DevIndex &ix, uint32_t qlen) {
  const int s = ix.s, k = ix.k;
  uint64_t roffs, rlen;
  c.sqidx = p.seqidx;
  c.flags = (p.flag & CANDFLG_REVERSE) ? RCF_REVERSE : 0;
  c.cover = p.cover;
  c.swscor = 0;
  if (p.seqidx < 0 || p.seqidx >= ix.nseq) { roffs = 0; rlen = ix.sop[ix.nseq]; }
  else { roffs = ix.sop[p.seqidx]; rlen = ix.sop[p.seqidx + 1] - roffs; }
  uint64_t rs = ((uint64_t)p.rs) * (uint64_t)s;
  uint64_t re = ((uint64_t)p.re) * (uint64_t)s + (uint64_t)k - 1;
  if (rs < roffs || re < rs) return -1;
  rs -= roffs; re -= roffs;
  if (re >= rlen) return -1;
  if (p.qe < p.qs || p.qs >= qlen) return -1;
  uint32_t qs, qe;
  if (p.flag & CANDFLG_REVERSE) { qs = qlen - p.qe - 1; qe = qlen - p.qs - 1; }
  else { qs = p.qs; qe = p.qe; }
  int edge_band = (int)(qlen - p.cover) / EDGE_BAND_FACTOR;
  if (edge_band > s) {
    if (edge_band > (int)(qlen >> MAX_BANDEDGE_2POW)) edge_band = (int)(qlen >> MAX_BANDEDGE_2POW);
    edge_band -= s - 1;
  } else edge_band = 0;
  int br = (-p.shiftoffs + 1) * s + edge_band + 1;
  int bl = br - (p.srange + 2) * s - 2 * edge_band - 2;
  int q_edge_l = (int)qs, q_edge_r = (int)(qlen - qe - 1);
  qs -= (uint32_t)q_edge_l;
  qe += (uint32_t)q_edge_r;
  int r_edge_l = q_edge_l + br, r_edge_r = q_edge_r - bl;
  if (r_edge_l > 0 && rs < (uint64_t)r_edge_l) { r_edge_l = (int)rs; rs = 0; }
  else rs -= (uint64_t)(int64_t)r_edge_l;
  if (re + (uint64_t)(int64_t)r_edge_r >= rlen) { re = rlen - 1; }
  else re += (uint64_t)(int64_t)r_edge_r;
  if (re < rs) return -1;
  int band_offs = q_edge_l - r_edge_l;
  c.band_l = bl + band_offs + (int)qs;
  c.band_r = br + band_offs + (int)qs;
  c.qs = qs; c.qe = qe; c.rs = rs; c.re = re;
  if (c.qe > 0x7fffffffu || c.re - c.rs > 0x7fffffffull) return -1;
  bool simd = qlen >= (uint32_t)MINLEN_QUERY_STRIPED && ((uint32_t)(c.band_r - c.band_l) * (uint32_t)BWSCAL_QLEN) > qlen &&
              c.qs == 0 && c.qe >= qlen - 1;
  if (!simd) c.flags |= RCF_BANDED;
  return 0;
}

// ---------------------------------------------------------------------------------------
// O1: the sequential control of scoreRMAPCAND (rmap.c:756-785) and the threshold block of
// mapSingleRead (rmap.c:1373-1400), replayed over the scores the GPU computed for all
// ranked candidates.
// ---------------------------------------------------------------------------------------
SMG_HD inline void replay_scores(ReadCtl &ctl, const RCand *rc, uint32_t n_candseg, const uint32_t cover_deficit[2],
                                 const MapPar &p, int s, int k, uint32_t qlen) {
  const int mmscordiff = p.match - p.mismatch;
  uint32_t max_cover = 0, min_cover = 0, i;
  int max1 = 0, max2 = 0;
  for (i = 0; i < n_candseg; i++) {
    const RCand &c = rc[i];
    uint32_t cover = c.cover, cdf = cover_deficit[c.flags & RCF_REVERSE];
    if ((p.flags & FLG_BEST) && cover + cdf < min_cover) break;
    if (c.swscor > max2) {
      if (c.swscor > max1) {
        max2 = max1; max1 = c.swscor;
        if (cover + cdf > max_cover) max_cover = (cover > cdf) ? cover - cdf : 0;
      } else max2 = c.swscor;
      uint32_t dcov = (uint32_t)(((max1 - max2) / mmscordiff + 1) * s);
      if (dcov + cdf + min_cover < max_cover) min_cover = max_cover - dcov;
    }
  }
  ctl.n_scored = (int32_t)i;
  ctl.max1 = max1; ctl.max2 = max2;
  ctl.go = 0;
  ctl.bandwidth_min = ctl.min_swatscor = ctl.scorlen_min = 0;
  const int maxscor_perfect = (int)qlen * p.match;
  if (max1 > maxscor_perfect || max1 < 1) return;
  int min_swatscor = p.min_swatscor, below_max = p.below_max, scorlen_min = k + s;
  ctl.bandwidth_min = (maxscor_perfect - max1) / (-1 * p.gap_ext);
  if (below_max >= max1) below_max = max1;
  if (min_swatscor > max2 && max2 > 0) min_swatscor = max2;
  if (below_max >= 0) {
    int minswc = (max2 > 0) ? max2 : max1;
    if (p.flags & FLG_BEST) { if (minswc > min_swatscor) min_swatscor = minswc; }
    else if (min_swatscor + below_max < max1) {
      min_swatscor = max1 - below_max;
      if (min_swatscor > minswc) min_swatscor = minswc;
    }
  }
  if (min_swatscor > scorlen_min * p.match && p.match > 0) scorlen_min = min_swatscor / p.match;
  ctl.min_swatscor = min_swatscor;
  ctl.scorlen_min = scorlen_min;
  ctl.go = 1;
}

// ---------------------------------------------------------------------------------------
// K2b / K3: band geometry (alignment.c:310-396) and the restricted, branch-ordered cell
// update shared by both (alignment.c:884-983, 1109-1197).
// ---------------------------------------------------------------------------------------
struct Band {
  int band_width, l_edge_orig, r_edge_orig, l_edge, r_edge;
  int s_left_orig, s_left, s_len, s_totlen, q_left_orig, q_left, q_len, q_totlen;
};

SMG_HD inline int band_init(Band &b, int l_edge, int r_edge, int q_left, int q_right, int q_len, int s_left, int s_right,
                            int s_len) {
  b.s_len = (s_right < 0 || s_right >= s_len) ? s_len : s_right + 1;
  b.q_len = (q_right < 0 || q_right >= q_len) ? q_len : q_right + 1;
  b.s_totlen = s_len;
  b.q_totlen = q_len;
  b.s_left = b.s_left_orig = (s_left > 0 && s_left < b.s_len) ? s_left : 0;
  b.q_left = b.q_left_orig = (q_left > 0 && q_left < b.q_len) ? q_left : 0;
  b.l_edge_orig = b.l_edge = l_edge;
  b.r_edge_orig = b.r_edge = r_edge;
  b.band_width = r_edge - l_edge + 1;
  if (b.band_width <= 0) {
    b.band_width = 0;
    b.l_edge = b.q_left;
    b.r_edge = b.q_len - 1;
  } else {
    if (b.l_edge_orig + b.s_len > b.q_len) b.s_len = b.q_len - b.l_edge_orig;
    b.l_edge += b.s_left;
    if (b.l_edge >= b.q_len || b.r_edge_orig + b.s_len <= b.q_left) return -1;
    b.r_edge += b.s_left;
    if (b.r_edge < b.q_left) {
      b.s_left += b.q_left - b.r_edge;
      b.l_edge += b.q_left - b.r_edge;
      b.r_edge = b.q_left;
    }
    if (b.r_edge > b.q_len - 1) b.r_edge = b.q_len - 1;
  }
  b.band_width = b.r_edge - b.l_edge + 1;
  return (b.band_width >= 0) ? 0 : -1;
}

// returns direction; cand: whether the cell may raise the running maximum.
// The reference spells the update out as a tree of cases on the signs of E and F (alignment.c:884-983).  With
// e = max(E, 0), f = max(F, 0) and m = max(e, f) the tree is: the diagonal wins iff H > m (strictly); otherwise the
// cell takes m, from E when e >= f (DIR_COL) else from F (DIR_ROW), or direction 0 when m == 0; a positive gap
// score is extended (minus ge) whatever the outcome, a non-positive one is left alone; and when the diagonal wins
// with H > gi both gap scores are raised to H - gi.  Branch-free, which is what 64 lanes in lock step need.
SMG_HD inline int cell_update(int &Hj, int &E, int &F, int H, int gi, int ge, bool &cand) {
  const int e = E > 0 ? E : 0, f = F > 0 ? F : 0;
  const int m = e > f ? e : f;
  const bool dia = H > m;
  Hj = dia ? H : m;
  const int dir = dia ? (int)DIR_DIA : (m == 0 ? 0 : (e >= f ? (int)DIR_COL : (int)DIR_ROW));
  const int t = H - gi;
  cand = dia && H > gi;
  const int Ed = E - (E > 0 ? ge : 0), Fd = F - (F > 0 ? ge : 0);
  E = (cand && t > Ed) ? t : Ed;
  F = (cand && t > Fd) ? t : Fd;
  return dir;
}

// diffStrReverse (diffstr.c:850-896): `in` is the reversed DiffStr of length n INCLUDING its
// terminating 0; writes the forward string (with terminator) to out, returns its length.
SMG_HD inline int diffstr_reverse(uint8_t *out, const uint8_t *in, int n) {
  int l = n - 1, u = 0;              // in[l] == 0
  l--;
  uint8_t count_prev = in[l] & 0x3f, typ = in[l] >> DIFF_TYPSHIFT, count;
  if (typ != DIFF_S) return -1;
  for (l--; l >= 0; l--) {
    count = in[l] & 0x3f; typ = in[l] >> DIFF_TYPSHIFT;
    if (typ == DIFF_M) {
      count_prev = (uint8_t)(count_prev + count + 1);
      if (count_prev > DIFF_MAXMISMATCH) { out[u++] = (uint8_t)(DIFF_MAXMISMATCH + (DIFF_M << DIFF_TYPSHIFT)); count_prev -= DIFF_MAXMISMATCH + 1; }
    } else {
      out[u++] = (uint8_t)(count_prev + (typ << DIFF_TYPSHIFT));
      count_prev = count;
    }
  }
  out[u++] = (uint8_t)(count_prev + (DIFF_S << DIFF_TYPSHIFT));
  out[u++] = 0;
  return u;
}

// 256-bit read-coverage mask in registers
struct QMask256 { uint32_t w[8]; };
SMG_HD inline void qm_clear(QMask256 &m) { for (int i = 0; i < 8; i++) m.w[i] = 0; }
// add [q, q+len) ; returns the number of newly covered bases
SMG_HD inline uint32_t qm_add(QMask256 &m, uint32_t q, uint32_t len) {
  uint32_t added = 0;
  const uint32_t e = q + len;
  for (int i = 0; i < 8; i++) {
    const uint32_t lo = (uint32_t)i * 32, hi = lo + 32;
    if (e <= lo || q >= hi) continue;
    const uint32_t a = q > lo ? q - lo : 0, b = e < hi ? e - lo : 32;
    const uint32_t bits = (b >= 32 ? 0xFFFFFFFFu : ((1u << b) - 1)) & ~((1u << a) - 1);
    added += (uint32_t)__builtin_popcount(bits & ~m.w[i]);
    m.w[i] |= bits;
  }
  return added;
}


}  // namespace smg
