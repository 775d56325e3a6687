// smg_cands.hpp -- wave-parallel form of the candidate stage (S3-S7) for the common case
// (min_ktup == 1, reads up to 256 bases): the same results as the sequential restatement in
// smg_stages.hpp (stage_cands), which stays as the path for everything else.
//
// What makes the parallel form possible (all derived from the reference's loops, see comments):
//   * every boundary in segment.c's three scans (hit region / seed / constant-shift segment) is
//     a LOCAL predicate on two neighbouring sorted hits, so each scan is a flag + ordered
//     stream compaction;
//   * addCandsFast's greedy absorption runs independently inside each hit region, so regions
//     go to different lanes; the read-coverage mask of a region fits in 8 registers (256 bits);
//   * candidates are emitted in (strand, sequence, region, segment) order, which an ordered
//     compaction over "first segment of a candidate" flags reproduces exactly;
//   * max_cover / max2nd_cover are order-free (largest and second largest DISTINCT cover).
// The per-strand working set (hit words, seeds, segments, regions) lives in LDS when the strand
// has at most CANDS_LDS_HITS hits, else in the HBM slot.
// `file:line` citations refer to the reference tree (SMALT 0.7.6, src/).
#pragma once
#include "smg_stages.hpp"
#include "smg_wsort.hpp"

namespace smg {

enum : uint32_t { CANDS_LDS_HITS = 640,        // hits of the LDS working set (one strand, or one window of a strand)
                  CANDS_TAB = 264,             // per-list tables: a read of the wave-parallel form has <= 256 seeds
                  CANDS_TAB_BYTES = 5 * CANDS_TAB * 4 };
enum : int { SMG_WINDOW_FALLBACK = 1000 };      // internal: a hit region does not fit a window

template <class IT>
struct StrandWork {           // one strand's working set; IT = uint16_t (LDS, LDS-typed pointers) or uint32_t (HBM)
  static constexpr bool L = sizeof(IT) == 2;
  typename ptr_of<uint64_t, L>::type dat;              // [cap]  sorted keys: seq(10) | diagonal(33) | q(20)  (strand bit cleared)
  typename ptr_of<IT, L>::type seed_first, seed_len;   // [cap]
  typename ptr_of<IT, L>::type segm_first, segm_nseed, segm_cover;   // [cap]
  typename ptr_of<IT, L>::type reg_first, reg_num;     // [cap]
  typename ptr_of<uint8_t, L>::type cflag;             // [cap]  1: a candidate starts at this segment
  uint32_t cap;
};

template <class IT>
SMG_HD inline size_t strand_work_bytes(uint32_t cap) { return (size_t)cap * (8 + 7 * sizeof(IT) + 1) + 64; }

template <class IT>
SMG_HD inline StrandWork<IT> strand_work_carve(uint8_t *base, uint32_t cap) {
  StrandWork<IT> w;
  typedef typename ptr_of<uint64_t, StrandWork<IT>::L>::type P64;
  typedef typename ptr_of<IT, StrandWork<IT>::L>::type PIT;
  typedef typename ptr_of<uint8_t, StrandWork<IT>::L>::type P8;
  w.cap = cap;
  w.dat = (P64)base; base += (size_t)cap * 8;
  w.seed_first = (PIT)base; base += (size_t)cap * sizeof(IT);
  w.seed_len = (PIT)base; base += (size_t)cap * sizeof(IT);
  w.segm_first = (PIT)base; base += (size_t)cap * sizeof(IT);
  w.segm_nseed = (PIT)base; base += (size_t)cap * sizeof(IT);
  w.segm_cover = (PIT)base; base += (size_t)cap * sizeof(IT);
  w.reg_first = (PIT)base; base += (size_t)cap * sizeof(IT);
  w.reg_num = (PIT)base; base += (size_t)cap * sizeof(IT);
  w.cflag = (P8)base;
  return w;
}

SMG_HD inline uint32_t key_q(uint64_t key) { return (uint32_t)(key & KEY_QMASK); }
SMG_HD inline uint64_t key_diag(uint64_t key) { return (key >> KEY_QBITS) & KEY_DIAGMASK; }
SMG_HD inline uint32_t key_grp(uint64_t key) { return (uint32_t)(key >> (KEY_QBITS + KEY_DIAGBITS)); }
SMG_HD inline uint64_t key_packed(uint64_t key) { return (key_diag(key) << HALFBIT) | key_q(key); }   // hashhit.h:67-72

// defineHitRegions (segment.c:431-444): region boundary between sorted hits a (previous) and b
SMG_HD inline bool region_break(uint64_t a, uint64_t b, uint64_t dsthresh) {
  return key_grp(a) != key_grp(b) || (key_packed(b) - key_packed(a)) >= dsthresh;
}
// makeSeedsFromHits (segment.c:493-510): hit b starts a new seed after hit a.  Inside a seed all
// offsets are congruent to the first one modulo s, so "(qo - qoffs) % s" is local.
SMG_HD inline bool seed_break(uint64_t a, uint64_t b, int k, int s) {
  const uint32_t qa = key_q(a), qb = key_q(b);
  return key_diag(a) != key_diag(b) || qb > qa + (uint32_t)k || ((qb - qa) % (uint32_t)s) != 0;
}

// calcSegmentBoundaries (segment.c:635-668) from the compact arrays
// x / s for offsets below 2^20 (magic = div_magic(s): a multiplication instead of a division)
SMG_HD inline uint32_t div_s(uint32_t x, uint32_t magic) { return magic ? (uint32_t)(((uint64_t)x * magic) >> 32) : x; }
template <class IT>
SMG_HD inline void segm_bounds(const StrandWork<IT> &w, uint32_t m, int k, uint32_t mg /* div_magic(s) */, bool is_reverse, uint32_t *qs, uint32_t *qe,
                               uint32_t *rs, uint32_t *re) {
  const uint32_t sa = w.segm_first[m], sb = sa + w.segm_nseed[m] - 1;
  const uint64_t ka = w.dat[w.seed_first[sa]], kb = w.dat[w.seed_first[sb]];
  const uint32_t qa = key_q(ka), qb = key_q(kb), lb = w.seed_len[sb];
  *qs = qa;
  *qe = qb + lb - 1;
  if (is_reverse) {
    *rs = (uint32_t)((key_diag(kb) - div_s(qb, mg)) & SOFFSMASK);
    *rs -= div_s(lb - (uint32_t)k, mg);
    *re = (uint32_t)((key_diag(ka) - div_s(qa, mg)) & SOFFSMASK);
  } else {
    *rs = (uint32_t)((key_diag(ka) + div_s(qa, mg)) & SOFFSMASK);
    *re = (uint32_t)((key_diag(kb) + div_s(qb, mg)) & SOFFSMASK);
    *re += div_s(lb - (uint32_t)k, mg);
  }
}

// derriveSEGCAND (segment.c:929-1059) over segments [m0, m0 + nseg)
template <class IT>
SMG_HD inline int derive_cand_c(SegCand &c, const StrandWork<IT> &w, uint32_t m0, int nseg, int k, uint32_t mg /* div_magic(s) */, uint32_t cover,
                                uint32_t mincover_noindel, uint32_t hregix, bool is_reverse, int32_t seqidx) {
  const uint64_t offbit = 1ull << (HALFBIT + 1);
  segm_bounds(w, m0, k, mg, is_reverse, &c.qs, &c.qe, &c.rs, &c.re);
  int64_t shift_min = (int64_t)key_diag(w.dat[w.seed_first[w.segm_first[m0]]]), shift_2mm = shift_min, shift_start, shift_last = shift_min;
  uint32_t maxcover = w.segm_cover[m0], qs, qe, rs, re;
  for (int n = 1; n < nseg; n++) {
    const uint32_t m = m0 + (uint32_t)n;
    segm_bounds(w, m, k, mg, is_reverse, &qs, &qe, &rs, &re);
    shift_last = (int64_t)key_diag(w.dat[w.seed_first[w.segm_first[m]]]);
    if (w.segm_cover[m] > maxcover) { shift_2mm = shift_last; maxcover = w.segm_cover[m]; }
    if (qs < c.qs) c.qs = qs;
    if (qe > c.qe) c.qe = qe;
    if (rs < c.rs) c.rs = rs;
    if (re > c.re) c.re = re;
  }
  uint8_t flag = 0;
  if (is_reverse) { flag |= CANDFLG_REVERSE; shift_start = ((int64_t)c.rs) + (int64_t)div_s(c.qe - (uint32_t)k + 1, mg); }
  else shift_start = (int64_t)((((uint64_t)c.rs) | offbit) - (uint64_t)div_s(c.qs, mg));
  const uint64_t shift_range = (uint64_t)(shift_last - shift_min);
  const int64_t diff_shift = shift_min - shift_start;
  if (shift_range > 32767 || diff_shift < -32768 || diff_shift > 32767) return -1;
  c.shiftoffs = (int16_t)diff_shift;
  if (maxcover >= mincover_noindel) {
    const int64_t ds = shift_2mm - shift_start;
    flag |= CANDFLG_MMALI;
    if (ds < -32768 || ds > 32767) return -1;
    c.shift2mm = (int16_t)ds;
  } else c.shift2mm = 0;
  c.flag = flag; c.pad = 0;
  c.srange = (int16_t)shift_range;
  c.cover = cover; c.nseg = nseg; c.hregix = hregix; c.seqidx = seqidx;
  return 0;
}

// One strand: hits in w.dat[0..n) (unsorted keys on entry) -> candidates appended to cand[*ncand..].
// cover8: byte array (capacity >= candcap) that receives the cover of every candidate.
// Returns 0 or an SMG_ERR_* code (wave-uniform).
//
// LONG (reads of 256 bases and more): covers do not fit a byte and the coverage mask of a region does not fit
// registers.  Regions of one segment (nearly all of them) need no mask; the others are taken one after the other by
// the whole wave, segment by segment as the reference does, the seeds of a segment spread over the lanes and the
// mask (one bit per read base) in memory.  lw: covers per first segment, list of multi-segment regions, mask words.
struct LongWork {
  uint32_t *ccov, *mlist, *mask;
  // Early drop of candidates the cover filter of S6 can never keep: its threshold only grows with the largest and second
  // largest cover seen so far, and its allowance is at most the cover deficit of strand [0] (prune_cdf0).
  int prune_on; uint32_t prune_mcbm, prune_cdf0;
  uint16_t *dbg_nseg; int32_t *dbg_seq;      // debug slots only: what the packed records of short reads leave out (dumps)
};

// bits [q, q + len) of a mask in memory; returns how many were clear (addCandsFast's cover_new, segment.c:1185-1200)
SMG_HD inline uint32_t mask_add_mem(uint32_t *mask, uint32_t q, uint32_t len) {
  uint32_t added = 0;
  const uint32_t e = q + len;
  for (uint32_t w = q >> 5; (w << 5) < e; w++) {
    const uint32_t lo = w << 5, a = q > lo ? q - lo : 0, z = e - lo;
    const uint32_t bits = (z >= 32 ? 0xFFFFFFFFu : ((1u << z) - 1u)) & ~((1u << a) - 1u);
    const uint32_t old = atomic_or_u32(&mask[w], bits);
    added += (uint32_t)__builtin_popcount(bits & ~old);
  }
  return added;
}

template <bool LONG, class IT>
SMG_HD inline int strand_cands(StrandWork<IT> &w, uint32_t n, bool is_reverse, bool seqbyseq, uint32_t qlen, int k, int s,
                               uint32_t mincover, uint8_t *cover8, SegCand *cand, uint32_t candcap, uint32_t *ncand_io,
                               uint32_t *max_cover_io, uint32_t *max2nd_io, unsigned long long *ph,
                               bool hold_tail, uint32_t *nproc_out, uint32_t *reg_base_io, const LongWork &lw, const IvRec *ivmap = nullptr,
                               bool presorted = false) {
  *nproc_out = n;
  if (!n) return 0;
  unsigned long long t0 = phase_clock(), t1;
#define SMG_PH(i) { t1 = phase_clock(); ph[i] += t1 - t0; t0 = t1; }
  if (!presorted) {                              // (k_hits delivers the keys of a strand in order)
#if defined(__HIP_DEVICE_COMPILE__)
  if (n <= 1024 && (StrandWork<IT>::L || n <= 256)) {   // keys sorted in registers: the LDS working set, and short runs of the HBM one (the
    if (n <= 256) wave_sort_u64_reg<4>(w.dat, n);       // few hits of a restricted call: ten round trips to HBM with the network in memory)
    else if (n <= 512) wave_sort_u64_reg<8>(w.dat, n);
    else wave_sort_u64_reg<16>(w.dat, n);
  } else if (!StrandWork<IT>::L) wave_sort_u64_chunked(w.dat, n);                  // longer runs in HBM (exhaustive search): chunks of 1024 keys in registers
  else wave_sort_u64(w.dat, n);
#else
  wave_sort_u64(w.dat, n);
#endif
  }
  SMG_SYNC();
  SMG_PH(2)
  uint32_t max_dshift = (uint32_t)(k * SEGMENTING_DIFFSHIFT / s) & 0xffffu;     // segment.c:426-429
  { uint32_t ds = (qlen - (uint32_t)k) / (uint32_t)s + 1; if (ds < max_dshift) max_dshift = ds & 0xffffu; }
  const uint64_t dsthresh = ((uint64_t)max_dshift) << HALFBIT;
  if (hold_tail) {           // more hits follow: stop at the last region boundary, the caller keeps the rest
    uint32_t last = 0;
    SMG_PAR_CHUNKS(base, n) {
      const uint32_t i = base + SMG_LANE;
      if (i > 0 && i < n && region_break(w.dat[i - 1], w.dat[i], dsthresh)) last = i;
    }
    last = wave_max_u32(last);
    if (!last) return SMG_WINDOW_FALLBACK;
    n = last;
    *nproc_out = n;
  }
  const uint32_t reg_base = *reg_base_io;

  // hits -> seeds, seeds -> constant-shift segments (makeSegmentsFromSeeds, segment.c:558-580), segments -> hit regions in
  // one sweep over the hits.  All three boundaries are predicates on two neighbouring sorted hits: a seed ends at a
  // region boundary, a change of diagonal, a gap of more than k bases or a change of the offset's residue modulo s; a
  // segment ends where a seed ends for any reason but the gap (all hits of a seed share diagonal and residue, so the
  // reference's comparison with the previous seed's first hit is a comparison with the previous hit); a region ends
  // at a region boundary.
  uint32_t nseed = 0, nsegm = 0, nreg = 0;
  SMG_PAR_CHUNKS(base, n) {
    const uint32_t i = base + SMG_LANE;
    bool sb = false, gb = false, rb = false;
    if (i < n) {
      if (i == 0) sb = gb = rb = true;
      else {
        const uint64_t ka = w.dat[i - 1], kb = w.dat[i];
        rb = region_break(ka, kb, dsthresh);
        gb = rb || key_diag(ka) != key_diag(kb) || ((key_q(kb) - key_q(ka)) % (uint32_t)s) != 0;
        sb = gb || key_q(kb) > key_q(ka) + (uint32_t)k;
      }
    }
    const uint32_t sslot = compact_slot(sb, nseed);
    const uint32_t gslot = compact_slot(gb, nsegm);
    const uint32_t rslot = compact_slot(rb, nreg);
    if (sb) w.seed_first[sslot] = (IT)i;
    if (gb) w.segm_first[gslot] = (IT)sslot;
    if (rb) w.reg_first[rslot] = (IT)gslot;
  }
  SMG_SYNC();
  SMG_PAR_CHUNKS(base, nseed) {
    const uint32_t j = base + SMG_LANE;
    if (j < nseed) {
      const uint32_t last = (j + 1 < nseed ? (uint32_t)w.seed_first[j + 1] : n) - 1;
      w.seed_len[j] = (IT)(key_q(w.dat[last]) + (uint32_t)k - key_q(w.dat[w.seed_first[j]]));
    }
  }
  SMG_SYNC();
  SMG_PAR_CHUNKS(base, nsegm) {
    const uint32_t m = base + SMG_LANE;
    if (m < nsegm) {
      const uint32_t a = w.segm_first[m], e = (m + 1 < nsegm ? (uint32_t)w.segm_first[m + 1] : nseed);
      uint32_t cov = 0;
      for (uint32_t j = a; j < e; j++) cov += w.seed_len[j];
      w.segm_nseed[m] = (IT)(e - a);
      w.segm_cover[m] = (IT)cov;
      w.cflag[m] = 0;
    }
  }
  SMG_SYNC();
  SMG_PH(3)
  // S5: one lane per hit region (addCandsFast, segment.c:1169-1217)
  uint32_t mx = *max_cover_io, mx2 = *max2nd_io;
  int err = 0;
  uint32_t nmulti = 0;
  SMG_PAR_CHUNKS(base, nreg) {
    const uint32_t r = base + SMG_LANE;
    bool multi = false;
    if (r < nreg) {
      const uint32_t first = w.reg_first[r], num = (r + 1 < nreg ? (uint32_t)w.reg_first[r + 1] : nsegm) - first;
      if (LONG && num > 1) multi = true;
      else for (uint32_t i = 0; i < num;) {
        const uint32_t m0 = first + i;
        uint32_t cover = w.segm_cover[m0], j = i + 1;
        if (j < num) {
          QMask256 mk;
          qm_clear(mk);
          for (uint32_t t = 0; t < (uint32_t)w.segm_nseed[m0]; t++) { const uint32_t sd = (uint32_t)w.segm_first[m0] + t; (void)qm_add(mk, key_q(w.dat[w.seed_first[sd]]), w.seed_len[sd]); }
          for (; j < num; j++) {
            const uint32_t m = first + j;
            uint32_t cover_new = 0;
            for (uint32_t t = 0; t < (uint32_t)w.segm_nseed[m]; t++) { const uint32_t sd = (uint32_t)w.segm_first[m] + t; cover_new += qm_add(mk, key_q(w.dat[w.seed_first[sd]]), w.seed_len[sd]); }
            if ((cover_new << 1) < (uint32_t)w.segm_cover[m] && cover >= mincover) break;
            cover += cover_new;
          }
        }
        if (cover >= mincover) {                 // the candidate record itself is derived below, one lane per candidate
          w.reg_num[m0] = (IT)(j - i);
          if (LONG) { w.cflag[m0] = 1; lw.ccov[m0] = cover; }
          else w.cflag[m0] = (uint8_t)cover;     // reads of this form have < 256 bases and cover >= mincover > 0
          if (cover > mx2) { if (cover > mx) { mx2 = mx; mx = cover; } else if (cover != mx) mx2 = cover; }
        }
        i = j;
      }
    }
    if (LONG) { const uint32_t slot = compact_slot(multi, nmulti); if (multi) lw.mlist[slot] = r; }
  }
  if (LONG && nmulti) {
    SMG_SYNC();
    const uint32_t nw = (qlen + 31) >> 5;
    for (uint32_t t = 0; t < nmulti; t++) {      // wave-uniform from here
      const uint32_t r = lw.mlist[t];
      const uint32_t first = w.reg_first[r], num = (r + 1 < nreg ? (uint32_t)w.reg_first[r + 1] : nsegm) - first;
      for (uint32_t i = 0; i < num;) {
        const uint32_t m0 = first + i;
        uint32_t cover = w.segm_cover[m0], j = i + 1;
        if (j < num) {
          SMG_SYNC();
          SMG_PAR_CHUNKS(base, nw) { const uint32_t x = base + SMG_LANE; if (x < nw) lw.mask[x] = 0; }
          SMG_SYNC();
          { const uint32_t ns0 = w.segm_nseed[m0], sf = w.segm_first[m0];
            SMG_PAR_CHUNKS(base, ns0) { const uint32_t x = base + SMG_LANE; if (x < ns0) (void)mask_add_mem(lw.mask, key_q(w.dat[w.seed_first[sf + x]]), w.seed_len[sf + x]); } }
          for (; j < num; j++) {
            const uint32_t m = first + j, nsm = w.segm_nseed[m], sf = w.segm_first[m];
            uint32_t cover_new = 0;
            SMG_SYNC();
            SMG_PAR_CHUNKS(base, nsm) { const uint32_t x = base + SMG_LANE; if (x < nsm) cover_new += mask_add_mem(lw.mask, key_q(w.dat[w.seed_first[sf + x]]), w.seed_len[sf + x]); }
            cover_new = wave_sum_u32(cover_new);
            if ((cover_new << 1) < (uint32_t)w.segm_cover[m] && cover >= mincover) break;
            cover += cover_new;
          }
        }
        if (cover >= mincover) {
          SMG_LANE0 { w.reg_num[m0] = (IT)(j - i); w.cflag[m0] = 1; lw.ccov[m0] = cover; }
          if (cover > mx2) { if (cover > mx) { mx2 = mx; mx = cover; } else if (cover != mx) mx2 = cover; }
        }
        i = j;
      }
    }
    SMG_SYNC();
  }
#if defined(__HIP_DEVICE_COMPILE__)
  // (largest, second largest distinct) over the lanes
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t omx = (uint32_t)__shfl_xor((int)mx, o), omx2 = (uint32_t)__shfl_xor((int)mx2, o);
    const uint32_t hi = mx > omx ? mx : omx;
    uint32_t lo = mx2 > omx2 ? mx2 : omx2;
    const uint32_t mn = mx < omx ? mx : omx;
    if (mn != hi && mn > lo) lo = mn;
    mx = hi; mx2 = lo;
  }
#endif
  *max_cover_io = mx; *max2nd_io = mx2;
  *reg_base_io = reg_base + nreg;
  SMG_SYNC();
  SMG_PH(4)
  // candidates in segment order: one lane per candidate derives the record (derriveSEGCAND) and writes it once
  uint32_t nc = *ncand_io;
  bool ovf = false;
  const uint32_t smg = div_magic(s);
  uint32_t pthr = 0;                             // lower bound of the final cover threshold (segment.c:1700-1730)
  if (lw.prune_on) { pthr = lw.prune_mcbm > mx ? 0 : mx - lw.prune_mcbm; if (pthr > mx2) pthr = mx2; }
  SMG_PAR_CHUNKS(base, nsegm) {
    const uint32_t m = base + SMG_LANE;
    bool f = m < nsegm && w.cflag[m];
    if (f && lw.prune_on && (LONG ? lw.ccov[m] : (uint32_t)w.cflag[m]) + lw.prune_cdf0 < pthr) f = false;
    const uint32_t slot = compact_slot(f, nc);
    if (f) {
      uint32_t lo = 0, hi = nreg;                // hit region of segment m (kept in the 40-byte record only: short reads skip the search)
      if (LONG) while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)w.reg_first[mid] <= m) lo = mid; else hi = mid; }
      const uint32_t grp = seqbyseq ? (key_grp(w.dat[w.seed_first[w.segm_first[m]]]) & ((2u << KEY_SEQBITS) - 1)) : 0u;     // (11 bits: interval numbers of restricted calls)
      int32_t seqidx = seqbyseq ? (int32_t)grp : -1;
      if (ivmap) seqidx = ivmap[seqidx].sx;       // interval-restricted call: the key's group is the interval number (rmap.c:486-490)
      SegCand c;
      const uint32_t ccover = LONG ? lw.ccov[m] : (uint32_t)w.cflag[m];
      if (derive_cand_c(c, w, m, (int)(uint32_t)w.reg_num[m], k, smg, ccover, mincover, reg_base + lo, is_reverse, seqidx)) err = SMG_ERR_ASSERT;
      if (slot >= candcap) ovf = true;
      else if (LONG) cand[slot] = c;
      else {
        // 16-byte record (the group as in the key: S7 translates interval numbers again) + the cover as a byte array,
        // which is all the S6 filter reads
        SegCandP pc;
        if (!segcand_pack(pc, c, grp, seqbyseq)) err = SMG_ERR_ASSERT;
        ((SegCandP *)cand)[slot] = pc;
        cover8[slot] = w.cflag[m];
        if (lw.dbg_nseg) { lw.dbg_nseg[slot] = (uint16_t)c.nseg; lw.dbg_seq[slot] = c.seqidx; }
      }
    }
  }
  *ncand_io = nc;
  SMG_PH(5)
#undef SMG_PH
  if (wave_any(err != 0)) return SMG_ERR_ASSERT;
  if (wave_any(ovf)) return SMG_ERR_CAP;
  return 0;
}

// hashCalcHitInfoCoverDeficit (hashhit.c:1096-1169) by the wave.  With a seed-rank cut the per-frame
// coverage is a union of k-base intervals (order-free): lanes OR the intervals of the seeds below the
// rank into one mask of nw words per sampling frame.  wk: >= (nw + 1) * s words of scratch.
SMG_HD inline uint32_t wave_cover_deficit(const HitInfoHdr &hdr, const SeedRec *seeds, const uint8_t *qmask, uint32_t qlen, int k, int s, uint32_t *wk,
                                          uint32_t nw = 8) {
  uint32_t *mask = wk, *val = wk + nw * (uint32_t)s;       // mask[s][nw] (nw * 32 >= qlen); val[s]: frame has seeds / per-frame result
  SMG_SYNC();
  SMG_PAR_CHUNKS(base, (nw + 1) * (uint32_t)s) { const uint32_t i = base + SMG_LANE; if (i < (nw + 1) * (uint32_t)s) wk[i] = 0; }
  SMG_SYNC();
  uint32_t deficit = 0;
  if (hdr.status & HI_RANK) {
    SMG_PAR_CHUNKS(base, hdr.n_seeds) {
      const uint32_t i = base + SMG_LANE;
      if (i < hdr.n_seeds) {
        const uint32_t q0 = seeds[i].qoffs, f = q0 % (uint32_t)s;
        (void)atomic_or_u32(&val[f], 1u);
        if (i < hdr.seed_rank) {
          for (uint32_t w = q0 >> 5; w <= (q0 + (uint32_t)k - 1) >> 5; w++) {
            const uint32_t lo = w << 5, a = q0 > lo ? q0 - lo : 0, e = q0 + (uint32_t)k - lo;
            const uint32_t bits = (e >= 32 ? 0xFFFFFFFFu : ((1u << e) - 1u)) & ~((1u << a) - 1u);
            if (w < nw) (void)atomic_or_u32(&mask[f * nw + w], bits);
          }
        }
      }
    }
    SMG_SYNC();
    uint32_t d = qlen, maxcover = 0;
    SMG_PAR_CHUNKS(base, (uint32_t)s) {
      const uint32_t f = base + SMG_LANE;
      if (f < (uint32_t)s && val[f]) {
        uint32_t cover = 0;
        for (uint32_t w = 0; w < nw; w++) cover += (uint32_t)__builtin_popcount(mask[f * nw + w]);
        if (cover < d) d = cover;
        if (cover > maxcover) maxcover = cover;
      }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t od = (uint32_t)__shfl_xor((int)d, o), om = (uint32_t)__shfl_xor((int)maxcover, o);
      if (od < d) d = od;
      if (om > maxcover) maxcover = om;
    }
#endif
    deficit = maxcover - d + 1;
  } else {
    uint8_t kk = (uint8_t)(k / s);
    if (kk > 0) kk--;
    SMG_PAR_CHUNKS(base, (uint32_t)s) {
      const uint32_t f = base + SMG_LANE;
      if (f < (uint32_t)s) {
        uint8_t ctr = 0;
        uint32_t d = 0;
        for (uint32_t i = f; i < qlen; i += (uint32_t)s) {
          if (qmask[i] == HQ_NORMHIT) ctr = kk;
          else if (ctr) ctr--;
          else d += (uint32_t)s;
        }
        if (d > deficit) deficit = d;
      }
    }
    deficit = wave_max_u32(deficit);
  }
  SMG_SYNC();
  return deficit;
}

struct CandsV2Scratch {
  uint8_t *lds; size_t lds_bytes;        // per-workgroup LDS block (the host build passes plain memory)
  uint32_t window;                       // test hook: hits per window (0: lds_hits)
  uint32_t lds_hits, tab;                // geometry of the LDS block: hits of the working set, entries of the per-list tables
  uint8_t *hbm;                          // HBM slot: strand work for large strands, candidates, sort arrays
  uint32_t hcap_strand;                  // capacity of the HBM strand work (hits per strand)
  uint8_t *cover8;                       // [candcap] cover of every candidate (what the S6 filter reads)
  SegCand *cand; uint32_t candcap;
  uint32_t *sort_keys, *sort_idx;
  FillDecision *dec; uint32_t ngrp;
  uint64_t *dbg_words; uint32_t *dbg_first, *dbg_cnt;   // debug: packed hit words grouped as the dump expects (or null)
  LongWork lw;                           // reads of 256 bases and more
  int pass;                              // 1: first of two passes -- a read that overflows this slot is deferred (SMG_ERR_RETRY)
};

SMG_HD inline size_t cands_v2_hbm_bytes(uint32_t qmax, int s, uint32_t hcap_strand, uint32_t ngrp, uint32_t candcap, bool debug) {
  (void)s;
  size_t n = strand_work_bytes<uint32_t>(hcap_strand) + (size_t)candcap * (sizeof(SegCand) + 8 + 1) + (size_t)ngrp * 2 * sizeof(FillDecision) + 512;
  if (qmax > 255) n += (size_t)hcap_strand * 8 + (size_t)(qmax / 32 + 2) * 4 + 64;
  if (debug) n += (size_t)hcap_strand * 2 * 8 + (size_t)ngrp * 2 * 8;
  return (n + 255) & ~(size_t)255;
}

SMG_HD inline CandsV2Scratch cands_v2_carve(uint8_t *lds, size_t lds_bytes, uint8_t *hbm, uint32_t qmax, int s, uint32_t hcap_strand,
                                            uint32_t ngrp, uint32_t candcap, bool debug) {
  CandsV2Scratch x;
  x.lds = lds; x.lds_bytes = lds_bytes; x.window = 0; x.lds_hits = CANDS_LDS_HITS; x.tab = CANDS_TAB; x.hcap_strand = hcap_strand; x.candcap = candcap; x.ngrp = ngrp;
  uint8_t *b = hbm;
  x.hbm = b; b += (strand_work_bytes<uint32_t>(hcap_strand) + 63) & ~(size_t)63;
  x.cand = (SegCand *)b; b += (size_t)candcap * sizeof(SegCand);
  x.sort_keys = (uint32_t *)b; b += (size_t)candcap * 4;
  x.sort_idx = (uint32_t *)b; b += (size_t)candcap * 4;
  x.dec = (FillDecision *)b; b += (size_t)ngrp * 2 * sizeof(FillDecision);
  x.cover8 = b; b += candcap;
  b = (uint8_t *)(((uintptr_t)b + 15) & ~(uintptr_t)15);
  if (debug) {
    x.dbg_words = (uint64_t *)b; b += (size_t)hcap_strand * 2 * 8;
    x.dbg_first = (uint32_t *)b; b += (size_t)ngrp * 2 * 4;
    x.dbg_cnt = (uint32_t *)b;
  } else { x.dbg_words = nullptr; x.dbg_first = x.dbg_cnt = nullptr; }
  x.pass = 0;
  x.lw.ccov = x.lw.mlist = x.lw.mask = nullptr;
  x.lw.prune_on = 0; x.lw.prune_mcbm = x.lw.prune_cdf0 = 0;
  x.lw.dbg_nseg = nullptr; x.lw.dbg_seq = nullptr;
  if (debug) {        // short reads write 16-byte records into the candidate array: its tail holds what the dumps print beside them
    x.lw.dbg_nseg = (uint16_t *)((uint8_t *)x.cand + (size_t)candcap * sizeof(SegCandP));
    x.lw.dbg_seq = (int32_t *)((uint8_t *)x.cand + (size_t)candcap * (sizeof(SegCandP) + 4));
  }
  if (qmax > 255) {
    b = (uint8_t *)(((uintptr_t)b + 15) & ~(uintptr_t)15);
    if (debug) b += (size_t)ngrp * 2 * 4;
    x.lw.ccov = (uint32_t *)b; b += (size_t)hcap_strand * 4;
    x.lw.mlist = (uint32_t *)b; b += (size_t)hcap_strand * 4;
    x.lw.mask = (uint32_t *)b;
  }
  return x;
}

// debug: processed hits in the layout of the reference's per-sequence hit lists
SMG_HD inline void dbg_hits(const CandsV2Scratch &x, const uint64_t *dat, uint32_t n, uint32_t gbase, uint32_t st, uint32_t ngrp, bool seqbyseq,
                            uint64_t *dw, uint32_t &last_grp) {
  SMG_PAR_CHUNKS(base, n) {
    const uint32_t i = base + SMG_LANE;
    if (i < n) {
      const uint64_t key = dat[i];
      const uint32_t grp = key_grp(key), prev = i ? key_grp(dat[i - 1]) : last_grp;
      const uint32_t gi = st * ngrp + (seqbyseq ? (grp & ((2u << KEY_SEQBITS) - 1)) : 0u);
      if (prev != grp) x.dbg_first[gi] = gbase + i + st * x.hcap_strand;
      (void)atomic_add_u32(&x.dbg_cnt[gi], 1u);
      dw[gbase + i] = key_packed(key);
    }
  }
  SMG_SYNC();
  if (n) last_grp = key_grp(dat[n - 1]);
  SMG_SYNC();
}

// hashCollectHitsForSegment's retry protocol (hashhit.c:1416-1546, 1730-1741) for one interval of a restricted call
// (collectHitsFromInterVal, rmap.c:438-492: use_short_hitinfo = 0).  All n_seeds seeds are visited in read-offset order
// (seedp + n), but the per-seed ceiling is tested on nhitqual_sortkeyp[n], the n-th SMALLEST hit count when the hit info was
// collected in its short form (hashhit.c:1478-1481) -- seeds[] is kept in rank order, so seeds[n].nhits is that key and
// ord[n] the seed at offset rank n (ord == nullptr: the info is unsorted and both orders coincide).
SMG_HD inline FillDecision fill_decide_iv(const DevIndex &ix, const SeedRec *seeds, const uint32_t *ord, uint32_t n_seeds, uint32_t lo, uint32_t hi,
                                          uint32_t nhit_max, int nhits_alloc, uint8_t *qmask) {
  FillDecision d;
  uint32_t m = nhit_max;
  for (;;) {
    uint32_t total = 0, n;
    bool aborted = false;
    for (n = 0; n < n_seeds; n++) {
      const SeedRec &sp = seeds[ord ? ord[n] : n];
      if (m > 0 && seeds[n].nhits > m) { qmask[sp.qoffs] = HQ_MULTIHIT; continue; }
      const uint32_t *posp;
      const uint32_t nhits = index_positions(ix, sp.posidx, &posp);
      const uint32_t a = lower_bound_u32(posp, nhits, lo);
      const uint32_t nh = nhits - a;
      if (nh == 0) continue;
      if (total + nh > (uint32_t)nhits_alloc) {
        if (m > 0) { aborted = true; break; }
        qmask[sp.qoffs] = HQ_MULTIHIT;
        continue;
      }
      total += lower_bound_u32(posp, nhits, hi) - a;
    }
    d.n_used = n;
    d.m_final = m;
    m /= 2;
    if (!(aborted && m > (uint32_t)MINHIT_PER_TUPLE)) break;
  }
  return d;
}

// The same decision from counts that were taken for all (seed, interval) pairs at once: tail[n] = positions of the n-th seed (in
// visiting order) at or behind the interval's first serial, inside[n] = those of them inside the interval.  The protocol's
// retries then cost no index reads (each retry of fill_decide_iv searches every position list again: 165 of the 240
// microseconds a read of a restricted round spent in the candidate stage, whenever a repeat seed made the sum of all hit
// counts exceed the list's capacity).
SMG_HD inline FillDecision fill_decide_counted(const SeedRec *seeds, const uint32_t *ord, uint32_t n_seeds, const uint32_t *tail, const uint32_t *inside,
                                               uint32_t nhit_max, int nhits_alloc, uint8_t *qmask) {
  FillDecision d;
  uint32_t m = nhit_max;
  for (;;) {
    uint32_t total = 0, n;
    bool aborted = false;
    for (n = 0; n < n_seeds; n++) {
      const uint32_t q = seeds[ord ? ord[n] : n].qoffs;
      if (m > 0 && seeds[n].nhits > m) { qmask[q] = HQ_MULTIHIT; continue; }
      const uint32_t nh = tail[n];
      if (nh == 0) continue;
      if (total + nh > (uint32_t)nhits_alloc) {
        if (m > 0) { aborted = true; break; }
        qmask[q] = HQ_MULTIHIT;
        continue;
      }
      total += inside[n];
    }
    d.n_used = n;
    d.m_final = m;
    m /= 2;
    if (!(aborted && m > (uint32_t)MINHIT_PER_TUPLE)) break;
  }
  return d;
}

// k-mer serial range [plo, phi) of interval v (hashCollectHitsForSegment, hashhit.c:1711-1717, called with
// sop[sx] + lo and sop[sx] + hi + 1, rmap.c:462-474)
SMG_HD inline void iv_serials(const DevIndex &ix, const IvRec &v, uint32_t *plo, uint32_t *phi) {
  const uint64_t offs = ix.sop[v.sx];
  const uint64_t a = (offs + v.lo) / (uint64_t)ix.s;
  uint64_t b = (offs + v.hi + 1) / (uint64_t)ix.s;
  if (b > 0xFFFFFFFFull) b = 0xFFFFFFFFull;
  *plo = (uint32_t)a; *phi = (uint32_t)b;
}

// true when the parallel form applies to this read.  It has no filter on the number of hits of a hit region (min_ktup,
// segment.c:781-800), so it needs that filter to be void: either the cover threshold is below k + s (calcMinKtup, rmap.c:240-247,
// gives min_ktup == 1), or the hit list is filled per sequence or per search interval (`restricted`) -- those lists keep their
// read-offset mask blank (hashBlankHitList before every fill, rmap.c:294, :461), and segLstFillHits takes one off min_ktup for
// every offset that is not marked as hit (segment.c:781-788), which brings any min_ktup <= read length down to 1.  The cover
// threshold itself, (min_ktup - 1) * s + k, applies in every case.
SMG_HD inline bool cands_v2_applicable(const MapPar &p, int k, int s, uint32_t qlen, bool restricted) {
  return qlen < (1u << KEY_QBITS) && (read_min_cover(p, qlen) < (uint32_t)(k + s) || (p.flags & FLG_SEQBYSEQ) != 0 || restricted);
}

// ---------------------------------------------------------------------------------------------------------------
// S3 on its own (kernel k_hits): the hits of one read strand are gathered from the position lists of its seeds, packed,
// put in order and written ONCE to the batch-wide pool; the candidate stage then streams them (stage_cands_v2, HITRUN_SORTED).
// Same lists and the same windows of ascending diagonal as the fused form further down -- but nothing of S4-S7 lives in this
// kernel, so it takes a third of the registers and half the LDS, and two to three times as many waves stay resident to hide
// the dependent index reads and LDS round trips the stage waits for.  What the fused form does not treat as a plain strand
// (allocation-boundary protocol, hashhit.c:1497), restricted calls (rmapPair) and reads of 256 bases and more keep S3 inside
// the candidate stage (HITRUN_NONE); so does any strand on which this function meets something it does not expect.
struct HitsScratch {
  uint8_t *lds; size_t lds_bytes;
  uint32_t W, tab;            // keys per window; entries of the per-list tables
};
SMG_HD inline size_t hits_lds_bytes(uint32_t W, uint32_t tab) { return (size_t)W * 8 + (((size_t)W + 64) * 2 + 15 & ~(size_t)15) + (size_t)5 * tab * 4 + 64; }

SMG_HD inline void stage_hits(const Batch &b, const DevIndex &ix, const MapPar &p, uint32_t r, uint32_t st, const HitsScratch &x, unsigned long long *ph) {
  unsigned long long t0 = phase_clock(), t1;
#define SMG_PH(i) { t1 = phase_clock(); ph[i] += t1 - t0; t0 = t1; }
  const uint32_t rs = 2 * r + st;
  const uint32_t qlen = read_len(b, r);
  const int k = ix.k, s = ix.s;
  const bool seqbyseq = (p.flags & FLG_SEQBYSEQ) != 0;
  HitRun run;
  run.off = 0; run.n = 0; run.mode = HITRUN_NONE;
  typedef typename ptr_of<uint64_t, true>::type P64;
  typedef typename ptr_of<uint16_t, true>::type P16;
  typedef typename ptr_of<uint32_t, true>::type P32;
  const uint32_t W = x.W;
  P64 dat = (P64)x.lds;
  P16 marks = (P16)(x.lds + (size_t)W * 8);
  P32 gt = (P32)(x.lds + (size_t)W * 8 + ((((size_t)W + 64) * 2 + 15) & ~(size_t)15));
  const uint32_t tabn = x.tab;
  P32 g_pfx = gt, g_poff = gt + tabn, g_qo = gt + 2 * tabn, g_len = gt + 3 * tabn, g_cur = gt + 4 * tabn;
  const HitInfoHdr hdr = b.hi[rs];
  const uint32_t n_use = hdr.seed_rank > 0 ? hdr.seed_rank : hdr.n_seeds;
  bool mine = qlen >= (uint32_t)k && qlen < 256u && !b.iv_off && cands_v2_applicable(p, k, s, qlen, false) && n_use < tabn && W > tabn + 128u &&
              hits_lds_bytes(W, tabn) <= x.lds_bytes;
  if (!mine) { SMG_LANE0 { b.hitrun[rs] = run; } return; }
  int nhits_alloc, nhits_max;
  hitlist_caps(qlen, b.alloc_len ? b.alloc_len[r] : qlen, &nhits_alloc, &nhits_max);
  const uint32_t ncut = (uint32_t)(p.ncut > 0 ? p.ncut : 0);
  const uint32_t smagic = div_magic(s);
  const SeedRec *seeds = b.seeds + (size_t)rs * b.qmax;
  uint8_t *qmask = b.qmask + (size_t)rs * b.qmax;
  uint32_t tot = 0;
  SMG_PAR_CHUNKS(base, n_use) { const uint32_t n = base + SMG_LANE; if (n < n_use && !(ncut > 0 && seeds[n].nhits > ncut)) tot += seeds[n].nhits; }
  tot = wave_sum_u32(tot);
  uint32_t d_used = n_use, d_mfinal = 0;                      // concatenated mode: hashCollectHitsUsingCutoff's outcome (hashhit.c:1593-1689)
  if (seqbyseq) {
    if (tot > (uint32_t)nhits_alloc) { SMG_LANE0 { b.hitrun[rs] = run; } return; }       // allocation-boundary protocol: the candidate stage's business
    SMG_PAR_CHUNKS(base, n_use) { const uint32_t n = base + SMG_LANE; if (n < n_use && ncut > 0 && seeds[n].nhits > ncut) qmask[seeds[n].qoffs] = HQ_MULTIHIT; }
  } else {
    SMG_LANE0 {
      uint32_t m = ncut;
      for (;;) {
        uint32_t total = 0, i;
        bool ceiling = false;
        for (i = 0; i < n_use; i++) {
          const uint32_t nh = seeds[i].nhits;
          if (nh < 1) continue;
          if (m > 0 && nh > m) continue;
          if ((int)(total + nh) > nhits_max) { ceiling = true; break; }
          total += nh;
        }
        const uint32_t mf = m;
        m /= 2;
        if (!(ceiling && m > (uint32_t)MINHIT_PER_TUPLE)) { d_used = i; d_mfinal = mf; break; }
      }
    }
    d_used = bcast_lane0(d_used); d_mfinal = bcast_lane0(d_mfinal);
  }
  // position lists that contribute, with the exclusive prefix of their lengths
  uint32_t nlist = 0, total = 0;
  SMG_PAR_CHUNKS(base, n_use) {
    const uint32_t n = base + SMG_LANE;
    bool take = false;
    uint32_t nh = 0, poff = 0, qo = 0;
    if (n < n_use) {
      const SeedRec sd = seeds[n];
      if (seqbyseq) take = !(ncut > 0 && sd.nhits > ncut);
      else take = !(n >= d_used || (d_mfinal > 0 && sd.nhits > d_mfinal) || sd.nhits < 1);
      if (take) {
        const uint32_t *posp;
        nh = index_positions(ix, sd.posidx, &posp);
        poff = (uint32_t)(posp - ix.pos); qo = sd.qoffs;
        take = nh > 0;
      }
    }
    uint32_t incl = nh;
#if defined(__HIP_DEVICE_COMPILE__)
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if ((int)SMG_LANE >= o) incl += v; }
#endif
    const uint32_t slot = compact_slot(take, nlist);
    if (take) { g_pfx[slot] = total + incl - nh; g_poff[slot] = poff; g_qo[slot] = qo; g_len[slot] = nh; g_cur[slot] = 0; }
#if defined(__HIP_DEVICE_COMPILE__)
    total += (uint32_t)__shfl((int)incl, 63);
#else
    total += incl;
#endif
  }
  SMG_SYNC();
  if (!total) { run.mode = HITRUN_SORTED; SMG_LANE0 { b.hitrun[rs] = run; } return; }
  // a place in the pool
  {
    unsigned long long at = 0;
    SMG_LANE0 { at = atomic_add_u64(b.hit_count, (unsigned long long)total); }
    run.off = (unsigned long long)bcast_lane0((uint32_t)at) | (unsigned long long)bcast_lane0((uint32_t)(at >> 32)) << 32;
  }
  run.n = total;
  if (run.off + total > b.hitpool_cap) { run.mode = HITRUN_OVERFLOW; SMG_LANE0 { b.hitrun[rs] = run; } return; }
  uint64_t *out = b.hitpool + run.off;
  SMG_PH(0)
  auto sort_and_write = [&](uint32_t n, uint32_t at) {
    const unsigned long long ts0 = phase_clock();
#if defined(__HIP_DEVICE_COMPILE__)
    if (n <= 256) wave_sort_u64_reg<4>(dat, n);
    else if (n <= 512) wave_sort_u64_reg<8>(dat, n);
    else if (n <= 1024) wave_sort_u64_reg<16>(dat, n);
    else wave_sort_u64(dat, n);
#else
    wave_sort_u64(dat, n);
#endif
    SMG_SYNC();
    SMG_PAR_CHUNKS(base, n) { const uint32_t i = base + SMG_LANE; if (i < n) out[at + i] = dat[i]; }
    SMG_SYNC();
    ph[2] += phase_clock() - ts0;
  };
  bool bad = false;
  if (total <= W) {
    for (uint32_t base = 0; base < total; base += 4 * SMG_NLANES) {        // four independent index reads in flight per lane
      uint32_t pos[4], qo[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t h = base + (uint32_t)u * SMG_NLANES + SMG_LANE;
        if (h < total) {
          uint32_t lo = 0, hi = nlist;                     // last list with prefix <= h
          while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (g_pfx[mid] <= h) lo = mid; else hi = mid; }
          pos[u] = ix.pos[g_poff[lo] + (h - g_pfx[lo])]; qo[u] = g_qo[lo];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t h = base + (uint32_t)u * SMG_NLANES + SMG_LANE;
        if (h < total) {
          uint64_t key = (hit_diag_m(st != 0, pos[u], qo[u], smagic) << KEY_QBITS) | qo[u];
          if (seqbyseq) key |= (uint64_t)seq_of_pos(ix.seqlo, ix.nseq, pos[u]) << (KEY_DIAGBITS + KEY_QBITS);
          dat[h] = key;
        }
      }
    }
    SMG_SYNC();
    SMG_PH(1)
    sort_and_write(total, 0);
    t0 = phase_clock();
  } else {
    // windows of ascending diagonal: every list contributes what lies below the window's bound (stage_cands_v2, mode 1)
    uint32_t remaining = total, written = 0;
    uint64_t prev_bound = 0;
    const uint32_t room = W - nlist;
    while (remaining > 0) {
      uint64_t bound = ~0ull;
      SMG_PAR_CHUNKS(base, nlist) {
        const uint32_t l = base + SMG_LANE;
        if (l < nlist) {
          const uint32_t cur = g_cur[l], rem = g_len[l] - cur;
          if (rem) {
            const uint32_t t = (uint32_t)(((uint64_t)room * rem) / remaining) + 1;
            if (t < rem) {
              const uint32_t pos = ix.pos[g_poff[l] + cur + t];
              uint64_t kh = hit_diag_m(st != 0, pos, g_qo[l], smagic);
              if (seqbyseq) kh |= (uint64_t)seq_of_pos(ix.seqlo, ix.nseq, pos) << KEY_DIAGBITS;
              if (kh < bound) bound = kh;
            }
          }
        }
      }
      bound = wave_min_u64(bound);
      SMG_PAR_CHUNKS(base, room + nlist) { const uint32_t i = base + SMG_LANE; if (i < room + nlist) marks[i] = 0; }
      SMG_SYNC();
      uint32_t cnt_tot = 0;
      SMG_PAR_CHUNKS(base, nlist) {
        const uint32_t l = base + SMG_LANE;
        uint32_t cnt = 0;
        if (l < nlist) {
          const uint32_t cur = g_cur[l], rem = g_len[l] - cur;
          if (rem) {
            const uint32_t t = (uint32_t)(((uint64_t)room * rem) / remaining) + 1;
            uint32_t lo = 0, hi = t < rem ? t : rem;
            const uint32_t *pp = ix.pos + g_poff[l] + cur;
            const int64_t qs = (int64_t)(g_qo[l] / (uint32_t)s);
            const uint64_t db = bound & ((1ull << KEY_DIAGBITS) - 1ull);
            int64_t plim = st != 0 ? (int64_t)db - qs : (int64_t)db - (int64_t)(1ull << 32) + qs;
            if (seqbyseq) {
              const uint32_t sb = (uint32_t)(bound >> KEY_DIAGBITS);
              if (sb >= (uint32_t)ix.nseq) plim = (int64_t)1 << 33;
              else {
                const int64_t slo = (int64_t)ix.seqlo[sb], shi = sb + 1 < (uint32_t)ix.nseq ? (int64_t)ix.seqlo[sb + 1] : ((int64_t)1 << 33);
                plim = plim < slo ? slo : (plim > shi ? shi : plim);
              }
            } else if (bound == ~0ull) plim = (int64_t)1 << 33;
            while (lo < hi) {                                  // eight-way search: seven independent index reads per round
              const uint32_t n = hi - lo, step = (n + 7) >> 3;
              uint32_t pv[7];
#pragma unroll
              for (int u = 0; u < 7; u++) { const uint32_t ip_ = lo + (uint32_t)(u + 1) * step - 1; pv[u] = ip_ < hi ? pp[ip_] : 0xffffffffu; }
              uint32_t c = 0;
#pragma unroll
              for (int u = 0; u < 7; u++) c += ((lo + (uint32_t)(u + 1) * step - 1 < hi) && (int64_t)pv[u] < plim) ? 1u : 0u;
              const uint32_t nlo = lo + c * step, piv = nlo + step - 1;
              if (c < 7 && piv < hi) hi = piv;
              lo = nlo;
              if (step == 1 && c < 7) break;
            }
            cnt = lo < hi ? lo : hi;
          }
        }
        uint32_t incl = cnt;
#if defined(__HIP_DEVICE_COMPILE__)
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if ((int)SMG_LANE >= o) incl += v; }
#endif
        if (l < nlist) { g_pfx[l] = cnt_tot + incl - cnt; if (cnt && cnt_tot + incl - cnt < room + nlist) marks[cnt_tot + incl - cnt] = (uint16_t)(l + 1); }
#if defined(__HIP_DEVICE_COMPILE__)
        cnt_tot += (uint32_t)__shfl((int)incl, 63);
#else
        cnt_tot += incl;
#endif
      }
      SMG_SYNC();
      if (cnt_tot > W || cnt_tot == 0) { bad = true; break; }
      const uint32_t sq_lo = seqbyseq ? (uint32_t)(prev_bound >> KEY_DIAGBITS) : 0u;
      uint32_t sq_hi = seqbyseq ? (uint32_t)(bound >> KEY_DIAGBITS) : 0u;
      if (sq_hi >= (uint32_t)ix.nseq) sq_hi = (uint32_t)ix.nseq - 1;
      prev_bound = bound;
      uint32_t id_carry = 0;
      for (uint32_t base = 0; base < cnt_tot; base += 4 * SMG_NLANES) {
        uint32_t pos[4], qo[4], mk[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t h = base + (uint32_t)u * SMG_NLANES + SMG_LANE; mk[u] = h < cnt_tot ? (uint32_t)marks[h] : 0u; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t h = base + (uint32_t)u * SMG_NLANES + SMG_LANE;
          const uint32_t id = wave_scan_max_u32(mk[u], id_carry, &id_carry);
          if (h < cnt_tot) {
            const uint32_t lo = id - 1;
            pos[u] = ix.pos[g_poff[lo] + g_cur[lo] + (h - g_pfx[lo])]; qo[u] = g_qo[lo];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t h = base + (uint32_t)u * SMG_NLANES + SMG_LANE;
          if (h < cnt_tot) {
            uint64_t key = (hit_diag_m(st != 0, pos[u], qo[u], smagic) << KEY_QBITS) | qo[u];
            if (seqbyseq) {
              uint32_t lo = sq_lo, hi = sq_hi + 1;
              while (hi - lo > 1) { const uint32_t m = (lo + hi) >> 1; if (ix.seqlo[m] <= pos[u]) lo = m; else hi = m; }
              key |= (uint64_t)lo << (KEY_DIAGBITS + KEY_QBITS);
            }
            dat[h] = key;
          }
        }
      }
      SMG_SYNC();
      SMG_PAR_CHUNKS(base, nlist) {
        const uint32_t l = base + SMG_LANE;
        if (l < nlist) g_cur[l] += (l + 1 < nlist ? g_pfx[l + 1] : cnt_tot) - g_pfx[l];
      }
      remaining -= cnt_tot;
      SMG_SYNC();
      SMG_PH(1)
      sort_and_write(cnt_tot, written);
      t0 = phase_clock();
      written += cnt_tot;
    }
    if (!bad && written != total) bad = true;
  }
  if (!bad) run.mode = HITRUN_SORTED;          // (else the candidate stage does this strand itself; the pool space stays unused)
  SMG_LANE0 { b.hitrun[rs] = run; }
#undef SMG_PH
}

// LONG: reads of 256 bases and more (wide covers, coverage masks in memory, 64-bit ranking words)
// SPLIT: S3 runs ahead in k_hits; a strand it left alone (HITRUN_NONE) takes the HBM working set here, so the windowed gather is
// not part of this instance (fewer live registers in the kernel)
template <bool LONG, bool SPLIT = false>
SMG_HD inline uint32_t stage_cands_v2(const Batch &b, const DevIndex &ix, const MapPar &p, uint32_t r, CandsV2Scratch &x, unsigned long long *ph) {
  unsigned long long t0 = phase_clock(), t1;
#define SMG_PH(i) { t1 = phase_clock(); ph[i] += t1 - t0; t0 = t1; }
  const uint32_t qlen = read_len(b, r);
  CandHdr &ch = b.ch[r];
  const int k = ix.k, s = ix.s;
  const bool seqbyseq = (p.flags & FLG_SEQBYSEQ) != 0;
  const uint32_t ngrp = x.ngrp;
  if (qlen < (uint32_t)k) {
    SMG_LANE0 { ch.ncand = ch.n_sort = ch.n_mincover = ch.max_cover = ch.max2nd_cover = 0; ch.cover_deficit[0] = ch.cover_deficit[1] = 0; ch.rc_off = 0; ch.err = 0; ch.err_site = 0; ch.nhits[0] = ch.nhits[1] = 0; ch.n_reserved = 0; }
    return 0;
  }
  uint32_t min_cover = read_min_cover(p, qlen);                   // calcMinKtup (rmap.c:240-247)
  { const uint32_t min_ktup = (min_cover >= (uint32_t)(k + s)) ? (min_cover - (uint32_t)k) / (uint32_t)s : 1u; min_cover = (min_ktup - 1) * (uint32_t)s + (uint32_t)k; }
  const int mismatchdiff = p.match - p.mismatch;
  uint32_t mincov_below_max;
  if (p.below_max < 0) mincov_below_max = qlen - 1;
  else {
    mincov_below_max = ((uint32_t)(p.below_max / mismatchdiff)) * (uint32_t)s;
    if (mincov_below_max < (uint32_t)k || (p.flags & FLG_BEST)) mincov_below_max = (uint32_t)(k + 2 * (s - 1));
  }
  int nhits_alloc, nhits_max;
  hitlist_caps(qlen, b.alloc_len ? b.alloc_len[r] : qlen, &nhits_alloc, &nhits_max);
  const uint32_t ncut = (uint32_t)(p.ncut > 0 ? p.ncut : 0);
  const uint32_t smagic = div_magic(s);
  int err = 0, site = 0;
  uint32_t ncand = 0, max_cover = 0, max2nd = 0, nhits_total = 0;
  if (x.dbg_first) { SMG_PAR_CHUNKS(base, 2 * ngrp) { uint32_t g = base + SMG_LANE; if (g < 2 * ngrp) { x.dbg_first[g] = 0; x.dbg_cnt[g] = 0; } } }

  // rmapPair's restricted calls (rmap.c:1940-1954, :2010-2039): one hit list per interval instead of per sequence
  const bool ivmode = !SPLIT && b.iv_off != nullptr;          // (restricted calls never run with S3 split off: the launcher sees to that)
  const uint32_t niv = ivmode ? b.iv_off[r + 1] - b.iv_off[r] : 0u;
  const IvRec *ivr = ivmode ? b.iv + b.iv_off[r] : nullptr;
  if (ivmode && (niv > (uint32_t)IV_MAX || (uint64_t)2 * niv > x.candcap)) { err = SMG_ERR_CAP; site = __LINE__; }

  for (uint32_t st = 0; st < 2 && !err; st++) {
    const uint32_t rs = 2 * r + st;
    const unsigned long long ts = phase_clock();
    const HitInfoHdr hdr = b.hi[rs];
    const SeedRec *seeds = b.seeds + (size_t)rs * b.qmax;
    uint8_t *qmask = b.qmask + (size_t)rs * b.qmax;
    if (ivmode) {
      SMG_PH(0)
      // All seeds take part (use_short_hitinfo = 0), in read-offset order.  Hits are few (the intervals are a few hundred
      // bases), so the strand always takes the HBM working set; what costs is finding each position list's slice.
      const uint32_t n_all = hdr.n_seeds;
      uint32_t tot = 0;
      SMG_PAR_CHUNKS(base, n_all) { const uint32_t n = base + SMG_LANE; if (n < n_all && !(ncut > 0 && seeds[n].nhits > ncut)) tot += seeds[n].nhits; }
      tot = wave_sum_u32(tot);
      const bool all_in = tot <= (uint32_t)nhits_alloc;        // the allocation boundary cannot be reached: every seed below the ceiling contributes
      FillDecision *dec = (FillDecision *)x.sort_keys;         // per interval (the ranking arrays are dead until S6)
      uint32_t *ord = nullptr;
      if (all_in) {
        SMG_PAR_CHUNKS(base, n_all) { const uint32_t n = base + SMG_LANE; if (n < n_all && ncut > 0 && seeds[n].nhits > ncut) qmask[seeds[n].qoffs] = HQ_MULTIHIT; }
      } else {
        if (hdr.status & HI_SORTED) {                           // rank order -> read-offset order
          ord = x.sort_idx;
          SMG_PAR_CHUNKS(base, n_all) {
            const uint32_t i = base + SMG_LANE;
            if (i < n_all) { const uint32_t q = seeds[i].qoffs; uint32_t rk = 0; for (uint32_t j = 0; j < n_all; j++) rk += seeds[j].qoffs < q; ord[rk] = i; }
          }
          SMG_SYNC();
        }
        // counts of every (seed, interval) pair first, all lanes busy, when the sort arrays have room for them behind the decisions
        // and the offset order; then one lane per interval runs the protocol on the counts
        const uint64_t ncnt = (uint64_t)n_all * niv;
        if (ncnt + 2ull * niv <= (uint64_t)x.candcap && ncnt + n_all <= (uint64_t)x.candcap) {
          uint32_t *tailc = x.sort_keys + 2 * niv, *insc = x.sort_idx + n_all;
          if (!LONG && x.lds && ncnt * 8 <= (uint64_t)x.lds_bytes) { tailc = (uint32_t *)x.lds; insc = tailc + ncnt; }   // (the LDS working set is idle in a restricted call)
          SMG_PAR_CHUNKS(base, (uint32_t)ncnt) {
            const uint32_t pi = base + SMG_LANE;
            if (pi < (uint32_t)ncnt) {
              const uint32_t g = pi / n_all, n = pi - g * n_all;
              const SeedRec sd = seeds[ord ? ord[n] : n];
              uint32_t plo, phi, a, e;
              const uint32_t *posp;
              iv_serials(ix, ivr[g], &plo, &phi);
              const uint32_t nh = index_positions(ix, sd.posidx, &posp);
              lower_bounds2_u32(posp, nh, plo, phi, &a, &e);
              tailc[pi] = nh - a; insc[pi] = e - a;
            }
          }
          SMG_SYNC();
          SMG_PAR_CHUNKS(base, niv) {
            const uint32_t g = base + SMG_LANE;
            if (g < niv) dec[g] = fill_decide_counted(seeds, ord, n_all, tailc + (size_t)g * n_all, insc + (size_t)g * n_all, ncut, nhits_alloc, qmask);
          }
        } else {
          SMG_PAR_CHUNKS(base, niv) {
            const uint32_t g = base + SMG_LANE;
            if (g < niv) { uint32_t plo, phi; iv_serials(ix, ivr[g], &plo, &phi); dec[g] = fill_decide_iv(ix, seeds, ord, n_all, plo, phi, ncut, nhits_alloc, qmask); }
          }
        }
      }
      SMG_SYNC();
      { t1 = phase_clock(); ph[13] += t1 - t0; t0 = t1; }     // (diagnostic: the decisions ahead of the gather)
      StrandWork<uint32_t> wg = strand_work_carve<uint32_t>(x.hbm, x.hcap_strand);
      uint32_t nkeys = 0;
      const uint32_t npairs = n_all * niv;
      bool ovf = false;
      SMG_PAR_CHUNKS(base, npairs) {
        const uint32_t pi = base + SMG_LANE;
        uint32_t cnt = 0, a = 0, g = 0, qo = 0;
        const uint32_t *posp = nullptr;
        if (pi < npairs) {
          const uint32_t n = pi / niv;
          g = pi - n * niv;
          const SeedRec sd = seeds[ord ? ord[n] : n];
          bool take = all_in ? !(ncut > 0 && sd.nhits > ncut) : (n < dec[g].n_used && !(dec[g].m_final > 0 && seeds[n].nhits > dec[g].m_final));
          if (take) {
            uint32_t plo, phi;
            iv_serials(ix, ivr[g], &plo, &phi);
            const uint32_t nh = index_positions(ix, sd.posidx, &posp);
            uint32_t e;
            lower_bounds2_u32(posp, nh, plo, phi, &a, &e);
            cnt = e - a;
            qo = sd.qoffs;
          }
        }
        uint32_t incl = cnt;
#if defined(__HIP_DEVICE_COMPILE__)
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if ((int)SMG_LANE >= o) incl += v; }
#endif
        const uint32_t at = nkeys + incl - cnt;
        if (at + cnt <= x.hcap_strand) {
          for (uint32_t j = 0; j < cnt; j++)
            wg.dat[at + j] = ((uint64_t)g << (KEY_DIAGBITS + KEY_QBITS)) | (hit_diag_m(st != 0, posp[a + j], qo, smagic) << KEY_QBITS) | qo;
        } else ovf = true;
#if defined(__HIP_DEVICE_COMPILE__)
        nkeys += (uint32_t)__shfl((int)incl, 63);
#else
        nkeys += incl;
#endif
      }
      if (wave_any(ovf)) { err = SMG_ERR_CAP; site = __LINE__; break; }
      SMG_SYNC();
      SMG_PH(1)
      uint32_t nproc = nkeys, reg_base = 0;
      const int rv = strand_cands<LONG>(wg, nkeys, st != 0, true, qlen, k, s, min_cover, x.cover8, x.cand, x.candcap, &ncand, &max_cover, &max2nd, ph,
                                        false, &nproc, &reg_base, x.lw, ivr);
      t0 = phase_clock();
      if (rv) { err = rv; site = __LINE__; break; }
      SMG_LANE0 { ch.nhits[st] = nkeys; }
      nhits_total += nkeys;
      SMG_SYNC();
      continue;
    }
    const uint32_t n_use = hdr.seed_rank > 0 ? hdr.seed_rank : hdr.n_seeds;
    uint32_t tot = 0;
    SMG_PAR_CHUNKS(base, n_use) {
      uint32_t n = base + SMG_LANE;
      if (n < n_use && !(ncut > 0 && seeds[n].nhits > ncut)) tot += seeds[n].nhits;
    }
    tot = wave_sum_u32(tot);
    FillDecision *dec = x.dec + st * ngrp;
    bool all_in = false;                     // every usable seed contributes everywhere (the common case)
    if (seqbyseq) {
      if (tot <= (uint32_t)nhits_alloc) {
        all_in = true;
        SMG_PAR_CHUNKS(base, n_use) {        // hashhit.c:1472-1481: over-cut seeds are flagged (-x mode only)
          uint32_t n = base + SMG_LANE;
          if (n < n_use && ncut > 0 && seeds[n].nhits > ncut) qmask[seeds[n].qoffs] = HQ_MULTIHIT;
        }
      } else {
        SMG_PAR_CHUNKS(base, ngrp) {         // rare: allocation-boundary retry protocol per sequence
          uint32_t g = base + SMG_LANE;
          if (g < ngrp) {
            uint64_t lo = ix.sop[g] / (uint64_t)s, hi = ix.sop[g + 1] / (uint64_t)s;
            if (hi > 0xFFFFFFFFull) hi = 0xFFFFFFFFull;
            dec[g] = fill_decide(ix, seeds, n_use, (uint32_t)lo, (uint32_t)hi, ncut, nhits_alloc, qmask);
          }
        }
      }
    } else {
      SMG_LANE0 {                            // hashCollectHitsUsingCutoff (hashhit.c:1593-1689)
        uint32_t m = ncut;
        for (;;) {
          uint32_t total = 0, i;
          bool ceiling = false;
          for (i = 0; i < n_use; i++) {
            uint32_t nh = seeds[i].nhits;
            if (nh < 1) continue;
            if (m > 0 && nh > m) continue;
            if ((int)(total + nh) > nhits_max) { ceiling = true; break; }
            total += nh;
          }
          uint32_t mf = m;
          m /= 2;
          if (!(ceiling && m > (uint32_t)MINHIT_PER_TUPLE)) { dec[0].n_used = i; dec[0].m_final = mf; break; }
        }
      }
    }
    SMG_SYNC();
    if (st == 0 && !x.dbg_first) {           // strand [0]'s read-offset flags are final now: its cover deficit bounds the S6 allowance
      uint32_t *cw = (LONG || !x.lds) ? (uint32_t *)x.hbm : (uint32_t *)x.lds;
      const uint32_t nwq0 = LONG ? (qlen + 31) >> 5 : 8u;
      x.lw.prune_cdf0 = wave_cover_deficit(hdr, seeds, qmask, qlen, k, s, cw, nwq0);
      x.lw.prune_mcbm = mincov_below_max;
      x.lw.prune_on = 1;
    }
    SMG_PH(0)
    // ---- the strand's keys already lie sorted in the batch-wide pool (k_hits): stream them through the LDS working set in
    //      chunks that end at a hit-region boundary, the unfinished region carried into the next chunk ----
    if (SPLIT && b.hitrun && !ivmode && b.hitrun[rs].mode != HITRUN_NONE) {
      const HitRun run = b.hitrun[rs];
      if (run.mode != HITRUN_SORTED) { err = SMG_ERR_CAP; site = __LINE__; break; }
      const uint64_t *src = b.hitpool + run.off;
      uint64_t *dbg_w = x.dbg_words ? x.dbg_words + (size_t)st * x.hcap_strand : nullptr;
      const size_t wl_bytes = (strand_work_bytes<uint16_t>(x.lds_hits) + 15) & ~(size_t)15;
      uint32_t W = x.lds_hits;
      if (x.window && x.window < W) W = x.window;
      const uint32_t ncand0 = ncand, mx0 = max_cover, mx20 = max2nd;
      int rv = SMG_WINDOW_FALLBACK;
      if (x.lds && wl_bytes <= x.lds_bytes && W >= 128) {
        StrandWork<uint16_t> wl = strand_work_carve<uint16_t>(x.lds, x.lds_hits);
        uint32_t carry = 0, done = 0, reg_base = 0, gproc = 0, last_grp = ~0u;
        rv = 0;
        while (done < run.n) {
          if (carry + 64 > W) { rv = SMG_WINDOW_FALLBACK; break; }
          const uint32_t take = (run.n - done) < (W - carry) ? (run.n - done) : (W - carry);
          SMG_PAR_CHUNKS(base, take) { const uint32_t i = base + SMG_LANE; if (i < take) wl.dat[carry + i] = src[done + i]; }
          done += take;
          const uint32_t n = carry + take;
          SMG_SYNC();
          SMG_PH(1)
          uint32_t nproc = n;
          rv = strand_cands<LONG>(wl, n, st != 0, seqbyseq, qlen, k, s, min_cover, x.cover8, x.cand, x.candcap, &ncand, &max_cover, &max2nd, ph,
                                  done < run.n, &nproc, &reg_base, x.lw, nullptr, true);
          t0 = phase_clock();
          if (rv) break;
          if (dbg_w) dbg_hits(x, (const uint64_t *)wl.dat, nproc, gproc, st, ngrp, seqbyseq, dbg_w, last_grp);
          gproc += nproc;
          carry = n - nproc;
          for (uint32_t base = 0; base < carry; base += SMG_NLANES) {
            const uint32_t i = base + SMG_LANE;
            uint64_t v = 0;
            if (i < carry) v = wl.dat[nproc + i];
            SMG_SYNC();
            if (i < carry) wl.dat[i] = v;
            SMG_SYNC();
          }
        }
        ph[12] += run.n;
      }
      if (rv == SMG_WINDOW_FALLBACK) {                // a hit region larger than a chunk (or no LDS block): the whole strand on the HBM working set
        ncand = ncand0; max_cover = mx0; max2nd = mx20;
        SMG_SYNC();
        if (x.dbg_first) { SMG_PAR_CHUNKS(base, ngrp) { uint32_t g = base + SMG_LANE; if (g < ngrp) { x.dbg_first[st * ngrp + g] = 0; x.dbg_cnt[st * ngrp + g] = 0; } } SMG_SYNC(); }
        if (run.n > x.hcap_strand) { err = SMG_ERR_CAP; site = __LINE__; break; }
        StrandWork<uint32_t> wg = strand_work_carve<uint32_t>(x.hbm, x.hcap_strand);
        wg.dat = const_cast<uint64_t *>(src);         // in order already; the index arrays of S4 live in the slot
        uint32_t nproc = run.n, reg_base = 0, last_grp = ~0u;
        rv = strand_cands<LONG>(wg, run.n, st != 0, seqbyseq, qlen, k, s, min_cover, x.cover8, x.cand, x.candcap, &ncand, &max_cover, &max2nd, ph,
                                false, &nproc, &reg_base, x.lw, nullptr, true);
        t0 = phase_clock();
        ph[14]++; ph[15] += t0 - ts;
        if (!rv && dbg_w) dbg_hits(x, src, run.n, 0, st, ngrp, seqbyseq, dbg_w, last_grp);
      }
      if (rv) { err = rv; site = __LINE__; break; }
      SMG_LANE0 { ch.nhits[st] = run.n; }
      nhits_total += run.n;
      SMG_SYNC();
      continue;
    }
    // ---- working set --------------------------------------------------------------------------
    //  mode 0: the whole strand fits the LDS working set
    //  mode 1: larger strands are streamed through the LDS working set in windows of ascending diagonal;
    //          a window ends at a hit-region boundary, so every later stage sees complete regions
    //  mode 2: HBM working set (allocation-boundary protocol active, or a region larger than a window)
    const bool simple = all_in || !seqbyseq;
    const size_t wl_bytes = (strand_work_bytes<uint16_t>(x.lds_hits) + 15) & ~(size_t)15;
    const bool lds_ok = x.lds && wl_bytes + (size_t)5 * x.tab * 4 <= x.lds_bytes && n_use < x.tab;
    const uint32_t tabn = lds_ok ? x.tab : ((n_use + 8) & ~3u);      // tables in the sort arrays (2 * candcap words) otherwise
    if (!lds_ok && (uint64_t)5 * tabn > (uint64_t)2 * x.candcap) { err = SMG_ERR_CAP; site = __LINE__; break; }
    uint32_t W = x.lds_hits;
    if (x.window && x.window < W) W = x.window;
    StrandWork<uint16_t> wl = strand_work_carve<uint16_t>(x.lds, x.lds_hits);
    StrandWork<uint32_t> wg = strand_work_carve<uint32_t>(x.hbm, x.hcap_strand);
    uint32_t *gt = lds_ok ? (uint32_t *)(x.lds + wl_bytes) : x.sort_keys;        // per-list tables (sort arrays are dead here)
    uint32_t *g_pfx = gt, *g_poff = gt + tabn, *g_qo = gt + 2 * tabn, *g_len = gt + 3 * tabn;
    uint32_t nlist = 0, total = 0;
    if (simple) {
      // Seeds that contribute become "lists" (position lists of the index, ascending); hit h of the strand
      // belongs to the list whose exclusive length prefix holds h.
      SMG_PAR_CHUNKS(base, n_use) {
        const uint32_t n = base + SMG_LANE;
        bool take = false;
        uint32_t nh = 0, poff = 0, qo = 0;
        if (n < n_use) {
          const SeedRec sd = seeds[n];
          if (seqbyseq) take = !(ncut > 0 && sd.nhits > ncut);
          else take = !(n >= dec[0].n_used || (dec[0].m_final > 0 && sd.nhits > dec[0].m_final) || sd.nhits < 1);
          if (take) {
            const uint32_t *posp;
            nh = index_positions(ix, sd.posidx, &posp);
            poff = (uint32_t)(posp - ix.pos); qo = sd.qoffs;
            take = nh > 0;
          }
        }
        uint32_t incl = nh;
#if defined(__HIP_DEVICE_COMPILE__)
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if ((int)SMG_LANE >= o) incl += v; }
#endif
        const uint32_t slot = compact_slot(take, nlist);
        if (take) { g_pfx[slot] = total + incl - nh; g_poff[slot] = poff; g_qo[slot] = qo; g_len[slot] = nh; }
#if defined(__HIP_DEVICE_COMPILE__)
        total += (uint32_t)__shfl((int)incl, 63);
#else
        total += incl;
#endif
      }
      SMG_SYNC();
    }
    { const unsigned long long tq = phase_clock(); ph[11] += tq - t0; }
    int mode = (!simple || SPLIT) ? 2 : ((lds_ok && total <= W) ? 0 : (lds_ok ? 1 : 2));
    const uint32_t ncand0 = ncand, mx0 = max_cover, mx20 = max2nd;
    uint32_t nkeys = 0;
    uint64_t *dbg_w = x.dbg_words ? x.dbg_words + (size_t)st * x.hcap_strand : nullptr;

    if (!SPLIT && mode == 1) {
      uint32_t *g_cur = gt + 4 * tabn;                 // per-list cursor
      auto marks = wl.reg_num;                          // [W] list starts among the hits of the window under construction
      SMG_PAR_CHUNKS(base, nlist) { const uint32_t l = base + SMG_LANE; if (l < nlist) g_cur[l] = 0; }
      SMG_SYNC();
      uint32_t carry = 0, remaining = total, reg_base = 0, gproc = 0, last_grp = ~0u;
      uint64_t prev_bound = 0;
      int rv = 0;
      while (remaining > 0 || carry > 0) {
        if (carry + nlist + 64 > W) { rv = SMG_WINDOW_FALLBACK; break; }
        const uint32_t room = W - carry - nlist;
        // window end: smallest (sequence, diagonal) reached by any list after its proportional share
        uint64_t bound = ~0ull;
        SMG_PAR_CHUNKS(base, nlist) {
          const uint32_t l = base + SMG_LANE;
          if (l < nlist) {
            const uint32_t cur = g_cur[l], rem = g_len[l] - cur;
            if (rem) {
              const uint32_t t = (uint32_t)(((uint64_t)room * rem) / remaining) + 1;
              if (t < rem) {
                const uint32_t pos = ix.pos[g_poff[l] + cur + t];
                uint64_t kh = hit_diag_m(st != 0, pos, g_qo[l], smagic);
                if (seqbyseq) kh |= (uint64_t)seq_of_pos(ix.seqlo, ix.nseq, pos) << KEY_DIAGBITS;
                if (kh < bound) bound = kh;
              }
            }
          }
        }
        bound = wave_min_u64(bound);
        // marks[h] = 1 + the list whose first hit of this window is hit h (0 elsewhere): a running maximum over the hits
        // then gives every hit its list without a search (the array is one of the working set's that is idle until S4)
        SMG_PAR_CHUNKS(base, room + nlist) { const uint32_t i = base + SMG_LANE; if (i < room + nlist) marks[i] = 0; }      // (a window takes at most room + nlist hits)
        SMG_SYNC();
        // per list: elements below the bound
        uint32_t cnt_tot = 0;
        SMG_PAR_CHUNKS(base, nlist) {
          const uint32_t l = base + SMG_LANE;
          uint32_t cnt = 0;
          if (l < nlist) {
            const uint32_t cur = g_cur[l], rem = g_len[l] - cur;
            if (rem) {
              const uint32_t t = (uint32_t)(((uint64_t)room * rem) / remaining) + 1;
              uint32_t lo = 0, hi = t < rem ? t : rem;       // first offset whose key is not below the bound
              const uint32_t *pp = ix.pos + g_poff[l] + cur;
              // Along one list the key grows with the position, so the bound turns into a position once per list:
              // key(pos) < bound  <=>  pos < plim (the diagonal bound shifted by the list's read offset, clamped to
              // the bound's sequence).
              const int64_t qs = (int64_t)(g_qo[l] / (uint32_t)s);
              const uint64_t db = bound & ((1ull << KEY_DIAGBITS) - 1ull);
              int64_t plim = st != 0 ? (int64_t)db - qs : (int64_t)db - (int64_t)(1ull << 32) + qs;
              if (seqbyseq) {
                const uint32_t sb = (uint32_t)(bound >> KEY_DIAGBITS);
                if (sb >= (uint32_t)ix.nseq) plim = (int64_t)1 << 33;                 // no bound: every position is below it
                else {
                  const int64_t slo = (int64_t)ix.seqlo[sb], shi = sb + 1 < (uint32_t)ix.nseq ? (int64_t)ix.seqlo[sb + 1] : ((int64_t)1 << 33);
                  plim = plim < slo ? slo : (plim > shi ? shi : plim);
                }
              } else if (bound == ~0ull) plim = (int64_t)1 << 33;
              // eight-way search: the seven pivots of a round are independent index reads (one round trip to HBM
              // instead of three); invariant: positions below lo are below the limit, those from hi on are not
              while (lo < hi) {
                const uint32_t n = hi - lo, step = (n + 7) >> 3;
                uint32_t pv[7];
#pragma unroll
                for (int u = 0; u < 7; u++) { const uint32_t ip_ = lo + (uint32_t)(u + 1) * step - 1; pv[u] = ip_ < hi ? pp[ip_] : 0xffffffffu; }
                uint32_t c = 0;
#pragma unroll
                for (int u = 0; u < 7; u++) c += ((lo + (uint32_t)(u + 1) * step - 1 < hi) && (int64_t)pv[u] < plim) ? 1u : 0u;
                const uint32_t nlo = lo + c * step, piv = nlo + step - 1;      // piv: the first pivot that is not below the limit
                if (c < 7 && piv < hi) hi = piv;
                lo = nlo;
                if (step == 1 && c < 7) break;                                 // every element of [lo, hi) was a pivot
              }
              cnt = lo < hi ? lo : hi;
            }
          }
          uint32_t incl = cnt;
#if defined(__HIP_DEVICE_COMPILE__)
          for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if ((int)SMG_LANE >= o) incl += v; }
#endif
          if (l < nlist) { g_pfx[l] = cnt_tot + incl - cnt; if (cnt && cnt_tot + incl - cnt < room + nlist) marks[cnt_tot + incl - cnt] = (uint16_t)(l + 1); }
#if defined(__HIP_DEVICE_COMPILE__)
          cnt_tot += (uint32_t)__shfl((int)incl, 63);
#else
          cnt_tot += incl;
#endif
        }
        SMG_SYNC();
        const unsigned long long tb0 = phase_clock();
        if (carry + cnt_tot > W || (cnt_tot == 0 && remaining > 0)) { rv = SMG_ERR_ASSERT; break; }
        const uint32_t sq_lo = seqbyseq ? (uint32_t)(prev_bound >> KEY_DIAGBITS) : 0u;
        uint32_t sq_hi = seqbyseq ? (uint32_t)(bound >> KEY_DIAGBITS) : 0u;
        if (sq_hi >= (uint32_t)ix.nseq) sq_hi = (uint32_t)ix.nseq - 1;
        prev_bound = bound;
        uint32_t id_carry = 0;
        for (uint32_t base = 0; base < cnt_tot; base += 4 * SMG_NLANES) {      // four independent index reads in flight per lane
          uint32_t pos[4], qo[4], mk[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { const uint32_t h = base + (uint32_t)u * SMG_NLANES + SMG_LANE; mk[u] = h < cnt_tot ? (uint32_t)marks[h] : 0u; }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const uint32_t h = base + (uint32_t)u * SMG_NLANES + SMG_LANE;
            const uint32_t id = wave_scan_max_u32(mk[u], id_carry, &id_carry);
            if (h < cnt_tot) {
              const uint32_t lo = id - 1;
              pos[u] = ix.pos[g_poff[lo] + g_cur[lo] + (h - g_pfx[lo])]; qo[u] = g_qo[lo];
            }
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const uint32_t h = base + (uint32_t)u * SMG_NLANES + SMG_LANE;
            if (h < cnt_tot) {
              uint64_t key = (hit_diag_m(st != 0, pos[u], qo[u], smagic) << KEY_QBITS) | qo[u];
              if (seqbyseq) {                      // the window's keys lie between the previous bound and this one: so do their sequences
                uint32_t lo = sq_lo, hi = sq_hi + 1;
                while (hi - lo > 1) { const uint32_t m = (lo + hi) >> 1; if (ix.seqlo[m] <= pos[u]) lo = m; else hi = m; }
                key |= (uint64_t)lo << (KEY_DIAGBITS + KEY_QBITS);
              }
              wl.dat[carry + h] = key;
            }
          }
        }
        SMG_SYNC();
        SMG_PAR_CHUNKS(base, nlist) {
          const uint32_t l = base + SMG_LANE;
          if (l < nlist) g_cur[l] += (l + 1 < nlist ? g_pfx[l + 1] : cnt_tot) - g_pfx[l];
        }
        remaining -= cnt_tot;
        const uint32_t n = carry + cnt_tot;
        SMG_SYNC();
        ph[13] += phase_clock() - tb0;
        SMG_PH(1)
        uint32_t nproc = n;
        rv = strand_cands<LONG>(wl, n, st != 0, seqbyseq, qlen, k, s, min_cover, x.cover8, x.cand, x.candcap, &ncand, &max_cover, &max2nd, ph,
                          remaining > 0, &nproc, &reg_base, x.lw);
        t0 = phase_clock();
        if (rv) break;
        if (dbg_w) dbg_hits(x, (const uint64_t *)wl.dat, nproc, gproc, st, ngrp, seqbyseq, dbg_w, last_grp);
        gproc += nproc;
        // the unfinished last region opens the next window
        carry = n - nproc;
        for (uint32_t base = 0; base < carry; base += SMG_NLANES) {
          const uint32_t i = base + SMG_LANE;
          uint64_t v = 0;
          if (i < carry) v = wl.dat[nproc + i];
          SMG_SYNC();
          if (i < carry) wl.dat[i] = v;
          SMG_SYNC();
        }
        if (remaining == 0) carry = 0;
      }
      if (rv == SMG_WINDOW_FALLBACK) {           // a hit region larger than the window: redo the strand in HBM
        ncand = ncand0; max_cover = mx0; max2nd = mx20;
        mode = 2;
        SMG_SYNC();
        SMG_LANE0 { uint32_t c = 0; for (uint32_t l = 0; l < nlist; l++) { g_pfx[l] = c; c += g_len[l]; } }   // the windows reused the prefix table
        SMG_SYNC();
        if (x.dbg_first) { SMG_SYNC(); SMG_PAR_CHUNKS(base, ngrp) { uint32_t g = base + SMG_LANE; if (g < ngrp) { x.dbg_first[st * ngrp + g] = 0; x.dbg_cnt[st * ngrp + g] = 0; } } SMG_SYNC(); }
      } else if (rv) { err = rv; site = __LINE__; break; }
      else nkeys = total;
      ph[12] += total;
    }

    if (mode != 1) {
      const bool in_lds = mode == 0;
      const uint32_t gcap = in_lds ? x.lds_hits : x.hcap_strand;
      uint64_t *dat = in_lds ? (uint64_t *)wl.dat : wg.dat;
      if (simple) {
        if (total > gcap) { err = SMG_ERR_CAP; site = __LINE__; break; }
        nkeys = total;
        for (uint32_t base = 0; base < nkeys; base += 4 * SMG_NLANES) {        // four independent index reads in flight per lane
          uint32_t pos[4], qo[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const uint32_t h = base + (uint32_t)u * SMG_NLANES + SMG_LANE;
            if (h < nkeys) {
              uint32_t lo = 0, hi = nlist;                     // last list with prefix <= h
              while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (g_pfx[mid] <= h) lo = mid; else hi = mid; }
              pos[u] = ix.pos[g_poff[lo] + (h - g_pfx[lo])]; qo[u] = g_qo[lo];
            }
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const uint32_t h = base + (uint32_t)u * SMG_NLANES + SMG_LANE;
            if (h < nkeys) {
              uint64_t key = (hit_diag_m(st != 0, pos[u], qo[u], smagic) << KEY_QBITS) | qo[u];
              if (seqbyseq) key |= (uint64_t)seq_of_pos(ix.seqlo, ix.nseq, pos[u]) << (KEY_DIAGBITS + KEY_QBITS);
              dat[h] = key;
            }
          }
        }
      } else {
        // gather under the allocation-boundary protocol (hashhit.c:1416-1546): per-sequence decisions
        for (uint32_t n = 0; n < n_use; n++) {
          const SeedRec sd = seeds[n];
          if (ncut > 0 && sd.nhits > ncut) continue;
          const uint32_t *posp;
          const uint32_t nh = index_positions(ix, sd.posidx, &posp);
          uint32_t cnt = 0;
          SMG_PAR_CHUNKS(base, nh) {
            uint32_t i = base + SMG_LANE;
            bool take = false;
            uint64_t key = 0;
            if (i < nh) {
              const uint32_t pos = posp[i];
              const uint32_t g = seq_of_pos(ix.seqlo, ix.nseq, pos);
              const FillDecision d = dec[g];
              take = n < d.n_used && !(d.m_final > 0 && sd.nhits > d.m_final);
              key = ((uint64_t)g << (KEY_DIAGBITS + KEY_QBITS)) | (hit_diag(st != 0, pos, sd.qoffs, s) << KEY_QBITS) | sd.qoffs;
            }
            const uint32_t slot = compact_slot(take, cnt);
            if (take && nkeys + slot < gcap) dat[nkeys + slot] = key;
          }
          nkeys += cnt;
          if (nkeys > gcap) { err = SMG_ERR_CAP; site = __LINE__; break; }
        }
        if (err) break;
      }
      SMG_SYNC();
      SMG_PH(1)
      int rv;
      uint32_t nproc = nkeys, reg_base = 0, last_grp = ~0u;
      if (in_lds) rv = strand_cands<LONG>(wl, nkeys, st != 0, seqbyseq, qlen, k, s, min_cover, x.cover8, x.cand, x.candcap, &ncand, &max_cover, &max2nd, ph, false, &nproc, &reg_base, x.lw);
      else rv = strand_cands<LONG>(wg, nkeys, st != 0, seqbyseq, qlen, k, s, min_cover, x.cover8, x.cand, x.candcap, &ncand, &max_cover, &max2nd, ph, false, &nproc, &reg_base, x.lw);
      t0 = phase_clock();
      if (!in_lds) { ph[14]++; ph[15] += t0 - ts; }
      if (rv) { err = rv; site = __LINE__; break; }
      if (dbg_w) dbg_hits(x, dat, nkeys, 0, st, ngrp, seqbyseq, dbg_w, last_grp);
    }
    SMG_LANE0 { ch.nhits[st] = nkeys; }
    nhits_total += nkeys;
    SMG_SYNC();
  }

  // ---- S6: cover deficits (hashhit.c:1096), threshold, ranking (segment.c:1616-1785) ----
  // The strand working set is dead now: its LDS block hosts the work words of the ranking sort, the key
  // histogram and, when the candidates fit, the packed sort array itself.
  uint32_t *kv = x.sort_keys;             // packed (key << 22) | candidate index
  uint32_t *wk = x.sort_idx;              // work words of the wave sort + key histogram
  uint32_t lds_sort_cap = 0;
  if (x.lds) {
    const size_t wkb = ((size_t)WSORT_WORDS + WSORT_NBINS) * 4;
    if (wkb + 4096 <= x.lds_bytes) {
      wk = (uint32_t *)x.lds;
      lds_sort_cap = (uint32_t)((x.lds_bytes - wkb) / 4);
      kv = wk + WSORT_WORDS + WSORT_NBINS;
    }
  }
  uint32_t *hist = wk + WSORT_WORDS;
  uint64_t *kv64 = nullptr;
  uint32_t nbins = WSORT_NBINS;
  const uint32_t nwq = LONG ? (qlen + 31) >> 5 : 8u;
  if (LONG) {                               // the HBM strand work is dead too: cover-deficit masks, then key histogram | ranking words
    nbins = max_cover + 2;
    hist = (uint32_t *)x.hbm;
    size_t hb = (size_t)(nwq + 1) * (uint32_t)s;
    if (hb < nbins) hb = nbins;
    kv64 = (uint64_t *)(x.hbm + ((hb * 4 + 15) & ~(size_t)15));
    if ((size_t)((uint8_t *)kv64 - x.hbm) + (size_t)ncand * 8 > strand_work_bytes<uint32_t>(x.hcap_strand)) { if (!err) site = __LINE__; err = err ? err : SMG_ERR_CAP; }
  }
  SMG_SYNC();
  uint32_t cdf[2] = {0, 0};
  for (uint32_t st = 0; st < 2; st++) {
    const uint32_t rs = 2 * r + st;
    if (st == 0 && x.lw.prune_on) { cdf[0] = x.lw.prune_cdf0; continue; }      // computed when strand [0]'s flags became final
    cdf[st] = wave_cover_deficit(b.hi[rs], b.seeds + (size_t)rs * b.qmax, b.qmask + (size_t)rs * b.qmax, qlen, k, s, LONG ? hist : wk, nwq);
  }
  SMG_LANE0 { ch.cover_deficit[0] = cdf[0]; ch.cover_deficit[1] = cdf[1]; }
  uint32_t target_depth = (uint32_t)p.target_depth, max_depth = (uint32_t)p.max_depth;
  if (max_depth < 1 || max_depth > (uint32_t)MAXIMUM_DEPTH) max_depth = MAXIMUM_DEPTH;
  if (target_depth < 1) target_depth = DEFAULT_TARGET_DEPTH;
  if (target_depth > max_depth) target_depth = max_depth;
  uint32_t min_cov_thr = (mincov_below_max > max_cover) ? 0 : max_cover - mincov_below_max, cdfx = 0;
  if (min_cov_thr > max2nd) { cdfx = min_cov_thr - max2nd; min_cov_thr = max2nd; }
  const uint32_t adj = (cdf[0] > cdfx) ? cdf[0] - cdfx : 0;       // deficit of strand [0] for both strands (:1676)
  uint32_t nmin = 0;
  if (!LONG && !err && (ncand > (1u << WSORT_IDXBITS) || max_cover >= (uint32_t)WSORT_NBINS)) { err = SMG_ERR_CAP; site = __LINE__; }
  if (!err) {
    if (ncand > lds_sort_cap) kv = x.sort_keys;
    SMG_PAR_CHUNKS(base, nbins) { const uint32_t i = base + SMG_LANE; if (i < nbins) hist[i] = 0; }
    SMG_SYNC();
    // candidates that pass the cover threshold, in candidate order (:1700-1730); four independent loads per lane
    for (uint32_t base = 0; base < ncand; base += 4 * SMG_NLANES) {
      uint32_t cov[4];
      for (int u = 0; u < 4; u++) { const uint32_t i = base + (uint32_t)u * SMG_NLANES + SMG_LANE; cov[u] = i < ncand ? (LONG ? x.cand[i].cover : (uint32_t)x.cover8[i]) : 0; }
      for (int u = 0; u < 4; u++) {
        const uint32_t i = base + (uint32_t)u * SMG_NLANES + SMG_LANE;
        const bool keep = i < ncand && !(cov[u] + adj < min_cov_thr);
        const uint32_t slot = compact_slot(keep, nmin);
        if (keep) {
          const uint32_t key = max_cover - cov[u];
          if (LONG) kv64[slot] = ((uint64_t)key << 32) | i;
          else kv[slot] = (key << WSORT_IDXBITS) | i;
          atomic_add_u32(&hist[key < nbins ? key : nbins - 1], 1u);
        }
      }
    }
  }
  SMG_SYNC();
  SMG_PH(6)
  // The sorted key sequence does not depend on the tie order: the depth cut (:1745-1775) follows from
  // the key histogram, and only ranks below it have to be brought into the reference's order.
  uint32_t nrank = 0;
  if (!err) {
    if (LONG) {                                                                   // hist[v] = #keys <= v
      uint32_t run = 0;
      SMG_PAR_CHUNKS(base, nbins) {
        const uint32_t i = base + SMG_LANE;
        uint32_t incl = i < nbins ? hist[i] : 0;
#if defined(__HIP_DEVICE_COMPILE__)
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if ((int)SMG_LANE >= o) incl += v; }
#endif
        if (i < nbins) hist[i] = run + incl;
#if defined(__HIP_DEVICE_COMPILE__)
        run += (uint32_t)__shfl((int)incl, 63);
#else
        run += incl;
#endif
      }
    } else SMG_LANE0 { uint32_t c = 0; for (int v = 0; v < WSORT_NBINS; v++) { c += hist[v]; hist[v] = c; } }
    SMG_SYNC();
#define SMG_CLT(v) ((v) == 0 ? 0u : hist[((v) > nbins ? nbins : (v)) - 1])
    uint32_t j = nmin;
    if (j > target_depth) {
      const uint32_t maxj = (j < max_depth) ? j : max_depth;
      if (p.flags & FLG_SENSITIVE) {
        const uint32_t c1 = SMG_CLT(adj), c2 = SMG_CLT((uint32_t)s);
        j = c1 > target_depth ? c1 : target_depth;                     // :1760-1764
        if (j > maxj) j = maxj;
        if (c2 > j) j = c2;
      } else {
        const uint32_t rank = j / 2;
        uint32_t lo = 0, hi = nbins - 1;                               // key of rank j/2
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (hist[mid] > rank) hi = mid; else lo = mid + 1; }
        uint32_t cov = lo;
        if (cov < (uint32_t)s) cov = (uint32_t)s;
        const uint32_t c1 = SMG_CLT(cov);
        j = c1 > target_depth ? c1 : target_depth;
        if (j > maxj) j = maxj;
      }
    }
#undef SMG_CLT
    nrank = j;
    SMG_SYNC();
    if (LONG) wave_sort_kv<32, WSORT_LISTCAP, WSORT_LSTK, uint64_t>(kv64, (int)nmin, (int)nrank, wk);
    else wave_sort_kv(kv, (int)nmin, (int)nrank, wk);    // sort.c:233 tie order
  }
  if (x.pass == 1 && err == SMG_ERR_CAP) err = SMG_ERR_RETRY;      // slot capacity: the second pass has full-size slots
  SMG_LANE0 {
    ch.ncand = ncand; ch.n_sort = nrank; ch.n_mincover = nmin; ch.max_cover = max_cover; ch.max2nd_cover = max2nd;
    ch.err = err; ch.err_site = site;
    ch.n_reserved = ch.n_sort;
    ch.rc_off = atomic_add_u32(b.rc_count, ch.n_sort);
    if ((uint64_t)ch.rc_off + ch.n_sort > b.rccap) { ch.err = SMG_ERR_CAP; ch.err_site = __LINE__; ch.n_sort = 0; }
  }
  SMG_SYNC();
  {                                         // ranked part to the slot (S7 below, diagnostics)
    const uint32_t ns = ch.n_sort;
    if (LONG) {
      SMG_PAR_CHUNKS(base, ns) { const uint32_t i = base + SMG_LANE; if (i < ns) { const uint64_t v = kv64[i]; x.sort_idx[i] = (uint32_t)v; x.sort_keys[i] = (uint32_t)(v >> 32); } }
    } else {
      SMG_PAR_CHUNKS(base, ns) { const uint32_t i = base + SMG_LANE; if (i < ns) { const uint32_t v = kv[i]; x.sort_idx[i] = v & ((1u << WSORT_IDXBITS) - 1u); } }
      SMG_SYNC();
      SMG_PAR_CHUNKS(base, ns) { const uint32_t i = base + SMG_LANE; if (i < ns) x.sort_keys[i] = kv[i] >> WSORT_IDXBITS; }
    }
  }
  SMG_SYNC();
  SMG_PH(7)
  ph[9] += ncand; ph[10] += nmin;
  // ---- S7 ----
  const uint32_t n_sort = ch.n_sort, rc_off = ch.rc_off;
  if (ch.err == SMG_ERR_CAP && ch.n_reserved > n_sort) rc_pool_fill_inert(b, rc_off, ch.n_reserved, r);
  bool qn = false;                          // reads with non-ACGT codes are scored in 32-bit lanes (k_sw_full)
  SMG_PAR_CHUNKS(base, qlen) { const uint32_t i = base + SMG_LANE; if (i < qlen && b.codes[b.read_off[r] + i] >= 4) qn = true; }
  qn = wave_any(qn);
  SMG_LANE0 { if (qn && n_sort) (void)atomic_add_u64(b.work + WK_QN_TASKS, n_sort); }
  SMG_PAR_CHUNKS(base, n_sort) {
    uint32_t i = base + SMG_LANE;
    if (i < n_sort) {
      RCand c;
      SegCand sc;
      if (LONG) sc = x.cand[x.sort_idx[i]];
      else {
        const uint32_t ci = x.sort_idx[i];
        segcand_unpack(sc, ((const SegCandP *)x.cand)[ci], (uint32_t)x.cover8[ci]);
        if (ivmode && sc.seqidx >= 0) sc.seqidx = ivr[sc.seqidx].sx;      // the key's group is the interval number (rmap.c:486-490)
      }
      if (cand_offsets(c, sc, ix, qlen)) { c.flags |= RCF_ERR; c.rs = c.re = 0; c.qs = c.qe = 0; c.band_l = c.band_r = 0; }
      if (qn) c.flags |= RCF_QN;
      if (!(c.flags & (RCF_ERR | RCF_BANDED))) {          // K2a task: which instance scores it (smg_kernels.hip)
        const uint32_t wl_ = (uint32_t)(c.re - c.rs + 1);
        if (qlen > b.tile_qmax || wl_ > 1016u) {           // SW_FULL_WMAX: strip kernel
          const unsigned long long li = atomic_add_u64(b.work + WK_STRIP_TASKS, 1ull);
          if (b.strip_list && li < b.strip_cap) b.strip_list[li] = rc_off + i;
        } else if (wl_ > 248u) {                            // SW_SHORT_WMAX: large-LDS instance of the packed kernel
          const unsigned long long li = atomic_add_u64(b.work + WK_LONG_TASKS, 1ull);
          if (b.long_list && li < b.long_cap) b.long_list[li] = rc_off + i;
        }
      } else if ((c.flags & RCF_BANDED) && !(c.flags & RCF_ERR) && qlen > 255) {   // K2b of a long read: wave kernel
        const unsigned long long li = atomic_add_u64(b.work + WK_STRIP_TASKS, 1ull);
        if (b.strip_list && li < b.strip_cap) b.strip_list[li] = rc_off + i;
      }
      c.rid = r; c.pad = 0;
      b.rcpool[rc_off + i] = c;
    }
  }
  SMG_PH(8)
#undef SMG_PH
  return nhits_total;
}

}  // namespace smg
