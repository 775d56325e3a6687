// smg_stages.hpp -- the per-read stages of the seed-and-extend path, each executed by the
// wavefront that owns the read (see smg_exec.h for the execution model).
//   stage_seed   S1 + S2   k-mer words of both strands, index lookups, rarity ranking, budget
//   stage_cands  S3 - S7   hit gather + sort, binning into candidates, ranking, windows/bands
//   stage_replay O1        sequential control over the scores of the ranked candidates
//   stage_align  K3        banded Smith-Waterman with traceback, recursive split, result list
// The wide Smith-Waterman score pass (K2a) is a kernel of its own (smg_sw.hip).
// `file:line` citations refer to the reference tree (SMALT 0.7.6, src/).
#pragma once
#include <math.h>
#include "smg_common.h"
#include "smg_exec.h"
#include "smg_logic.hpp"
#include "smg_wsort.hpp"

namespace smg {

// S3 ahead of the candidate stage (k_hits): the sorted hit keys of one read strand in the batch-wide pool
struct HitRun { unsigned long long off; uint32_t n; uint32_t mode; };
enum : uint32_t { HITRUN_NONE = 0,      // the candidate stage gathers and sorts this strand itself
                  HITRUN_SORTED = 1,    // hitpool[off .. off + n) holds the strand's keys in ascending order
                  HITRUN_OVERFLOW = 2 };// the pool was full: the read takes SMG_ERR_CAP (re-mapped in a smaller batch)

struct Batch {                          // one block of reads resident in HBM
  uint32_t nreads, qmax;                // qmax: stride of the per-read-strand arrays (>= longest read + 1)
  const uint8_t *codes;                 // 3-bit codes of all reads, forward orientation, concatenated
  const uint8_t *codes_rc;              // reverse complement of each read, same offsets
  const uint8_t *qual;                  // phred+33 or null
  const uint64_t *read_off;             // nreads + 1
  // S1/S2 outputs, indexed by rs = 2*read + strand
  HitInfoHdr *hi;                       // [2*nreads]
  SeedRec *seeds;                       // [2*nreads][qmax], rank order
  uint8_t *qmask;                       // [2*nreads][qmax]
  // S3-S7 outputs
  CandHdr *ch;                          // [nreads]
  RCand *rcpool; uint32_t rccap; uint32_t *rc_count;    // bump-allocated ranked candidates
  uint32_t *long_list; uint32_t long_cap;               // ranked candidates with windows > SW_SHORT_WMAX (count: work[WK_LONG_TASKS])
  uint32_t *strip_list; uint32_t strip_cap;             // ... whose read or window exceeds the register tiling (work[WK_STRIP_TASKS])
  uint32_t tile_qmax;                                   // longest read the register-tiled K2a kernels of this mapper take (0: none)
  // O1
  ReadCtl *ctl;                         // [nreads]
  // K3 outputs
  ReadStat *stat;                       // [nreads]
  Result *respool; uint64_t rescap; unsigned long long *res_count;
  uint8_t *dstrpool; uint64_t dstrcap; unsigned long long *dstr_count;
  int32_t *err_flag;                    // batch-wide first error
  uint32_t *strip_cursor;               // work-queue cursor of the packed strip kernel (pairs of strip-list entries)
  uint32_t *next_item;                  // work-queue cursors of the persistent kernels (seed, cands, align, align pass 2, cands pass 2), NEXT_ITEM_STRIDE words apart
  uint32_t *align_retry, *align_retry_n; // reads the first K3 pass deferred to the second one (SMG_ERR_RETRY), and how many
  uint32_t *cands_retry, *cands_retry_n; // the same for the candidate stage (reads whose hits overflow a first-pass slot)
  unsigned long long *work;             // [WK_NWORK] work counters (WK_*), one atomic per workgroup
  // ---- per-read context of the mapSingleRead calls rmapPair makes (rmap.c:1744-2112); all null for plain batches ----
  const uint32_t *iv_off; const IvRec *iv;   // seeding restricted to intervals iv[iv_off[r] .. iv_off[r+1]) (collectHitsFromInterVal, rmap.c:438-492)
  const int32_t *min_sw;                // [nreads] min_swatscor of the call (rmap.c:2031: the first mate's second-best score)
  const int32_t *prevmax;               // [2 * nreads] running score maxima of the ResultSet the call appends to (rmap.c:881-885)
  uint32_t *fine_idx, *fine_pos;        // on-the-fly k=5 s=1 index of each read over its intervals (rmap.c:495-517): idx[r][FINE_IDX_STRIDE], pos
  const uint32_t *fine_off;             // [nreads + 1] first position of read r in fine_pos
  uint32_t totals_only;                 // != 0: the batch stops behind the k-mer lookups (smaltgpu_hit_totals): stage_seed leaves the ranking out
  uint32_t raw_results;                 // != 0: every alignment of a call is returned (no duplicate handling: the caller holds the set the call appends to)
  const uint32_t *seed_range;           // [2 * nreads] or null: k-mer words are taken from bases [first, last] of the read only (mapSecondary, rmap.c:1435-1505)
  const uint32_t *alloc_len;            // [nreads] or null: length of the longest read the reference's one hit list has held up to read r (serial-order mode)
  // ---- S3 split off the candidate stage (k_hits): null when the mapper keeps S3 inside k_cands ----
  HitRun *hitrun;                       // [2 * nreads]
  uint64_t *hitpool; uint64_t hitpool_cap; unsigned long long *hit_count;    // bump-allocated sorted keys
  uint32_t *hits_cursor;                // work-queue cursor of k_hits
};

// Capacity and ceiling of the reference's hit list for a read of qlen bases (initHitList, hashhit.c:1262-1288).  The ceiling
// follows the read; the capacity only grows (blocks of 16384, from hashCreateHitList's 16384): it is set by the longest read the
// list has held so far -- `longest` -- which is the read itself unless the caller runs the serial-order mode (Batch::alloc_len).
SMG_HD inline void hitlist_caps(uint32_t qlen, uint32_t longest, int *nhits_alloc, int *nhits_max) {
  auto target_of = [](uint32_t len) {
    const double t = (double)len * log((double)len) * HITLST_LOGQLEN_FACT;   // hashhit.c:1266
    long long target = (long long)t;
    if (target > 0x7fffffffLL) target = 0x7fffffffLL; else if (target < HITLST_MINSIZ) target = HITLST_MINSIZ;
    return target;
  };
  const long long own = target_of(qlen), widest = longest > qlen ? target_of(longest) : own;
  long long alloc = HITLST_BLKSZ;
  if (widest > alloc) alloc = ((widest + HITLST_BLKSZ - 1) / HITLST_BLKSZ) * HITLST_BLKSZ;
  *nhits_alloc = (int)alloc; *nhits_max = (int)own;
}

// the index a read is seeded against: the mapper's, or the read's own on-the-fly index
SMG_HD inline DevIndex read_index(const Batch &b, const DevIndex &ix, uint32_t r) {
  if (!b.fine_idx) return ix;
  DevIndex f = ix;
  f.k = FINE_K; f.s = FINE_S; f.typ = IDX_PERFECT; f.nbits_key = 2 * FINE_K; f.nbits_lo = 0; f.nkeys = FINE_NKEYS; f.nwords = 0;
  f.idx = b.fine_idx + (size_t)r * FINE_IDX_STRIDE;
  f.pos = b.fine_pos + b.fine_off[r];
  f.npos = b.fine_off[r + 1] - b.fine_off[r];
  f.wordidx = f.posidx = nullptr;
  return f;
}
enum : int { NEXT_ITEM_STRIDE = 32 };          // a 128-byte line per cursor
enum : int { WK_LOOKUPS = 0, WK_HITS = 1, WK_CELLS_FULL = 2, WK_TASKS_FULL = 3, WK_CELLS_BAND = 4, WK_NCAND = 5, WK_NKEPT = 6,
              WK_QN_TASKS = 7 /* ranked candidates of reads with non-ACGT codes */, WK_LONG_TASKS = 17 /* windows > SW_SHORT_WMAX */, WK_STRIP_TASKS = 18 /* beyond the register tiling */,
              WK_PHASE0 = 8 /* .. 23: shader-clock ticks per phase of k_cands (diagnostic) */,
              WK_ALIGN0 = 24 /* .. 31: k_align ticks (window fetch, band pass, traceback + results), band passes, aligned candidates, band steps, sequential passes, sum of band widths */, WK_NWORK = 32 };

// The score kernels walk the candidate pool linearly.  A read whose ranked candidates do not fit keeps none
// (SMG_ERR_CAP); the part of its reservation that still lies inside the pool is filled with inert entries.
SMG_HD inline void rc_pool_fill_inert(const Batch &b, uint32_t rc_off, uint32_t n_reserved, uint32_t r) {
  const uint64_t lo = rc_off < b.rccap ? rc_off : b.rccap;
  const uint64_t hi = (uint64_t)rc_off + n_reserved < b.rccap ? (uint64_t)rc_off + n_reserved : b.rccap;
  for (uint64_t i = lo + SMG_LANE; i < hi; i += SMG_NLANES) {
    RCand c;
    c.flags = RCF_ERR | RCF_SCORED; c.qs = c.qe = 0; c.rs = c.re = 0; c.band_l = c.band_r = 0; c.sqidx = 0; c.swscor = 0; c.cover = 0;
    c.rid = r; c.pad = 0;
    b.rcpool[i] = c;
  }
}

// covermin_tuple of this read (smalt.c:1113-1126): a fraction of the read length (-c below 1.01) or an absolute number
SMG_HD inline uint32_t read_min_cover(const MapPar &p, uint32_t qlen) {
  if (p.cov_frac > 0.0) { uint32_t c = (uint32_t)(p.cov_frac * qlen); return c > qlen ? qlen : c; }
  return p.min_cover;
}

SMG_HD inline uint32_t read_len(const Batch &b, uint32_t r) { return (uint32_t)(b.read_off[r + 1] - b.read_off[r]); }

SMG_HD inline uint32_t atomic_add_u32(uint32_t *p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicAdd(p, v);
#else
  uint32_t o = *p; *p += v; return o;
#endif
}
SMG_HD inline uint32_t atomic_or_u32(uint32_t *p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicOr(p, v);
#else
  uint32_t o = *p; *p |= v; return o;
#endif
}
SMG_HD inline unsigned long long atomic_add_u64(unsigned long long *p, unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicAdd(p, v);
#else
  unsigned long long o = *p; *p += v; return o;
#endif
}

// =======================================================================================
// stage_seed: S1 (hashhit.c:480-657 collectHitInfo) + S2 (hashhit.c:1007-1080)
// =======================================================================================
struct SeedScratch {          // per wave; capacity qmax each unless noted
  uint32_t *vt;               // offsets of k-mers without non-standard / low-quality bases
  uint64_t *vw;               // their words
  uint32_t *key, *sidx;       // nhits per seed (offset order, then rarity-sorted) + companion
  uint32_t *sposidx, *sqoffs; // seeds in offset order
  uint32_t *qbr;              // read offset by rank
  uint32_t *frame_cnt;        // [s]
  uint32_t *frame_rank;       // [s * stride]
  uint8_t *qbuf;              // [qmax]
  uint32_t stride;
  uint32_t *kv;               // [qmax] packed (nhits << 12 | seed) for the wave form of the rarity sort
  uint32_t *wk;               // [SEED_WSORT_WORDS] its work words; also the prefix sums of the seed budget
};
enum : int { SEED_IDXBITS = 12, SEED_LISTCAP = 32, SEED_LSTK = 12, SEED_WSORT_WORDS = 256 + 2 * SEED_LISTCAP + 64 * SEED_LSTK };

SMG_HD inline size_t seed_scratch_bytes(uint32_t qmax, int s) {
  uint32_t stride = qmax / (uint32_t)s + 2;
  return (size_t)qmax * (4 + 8 + 4 + 4 + 4 + 4 + 4 + 1 + 4) + (size_t)SEED_WSORT_WORDS * 4 + (size_t)s * 4 + (size_t)s * stride * 4 + 64;
}

SMG_HD inline SeedScratch seed_scratch_carve(uint8_t *base, uint32_t qmax, int s) {
  SeedScratch x;
  x.vw = (uint64_t *)base; base += (size_t)qmax * 8;
  x.vt = (uint32_t *)base; base += (size_t)qmax * 4;
  x.key = (uint32_t *)base; base += (size_t)qmax * 4;
  x.sidx = (uint32_t *)base; base += (size_t)qmax * 4;
  x.sposidx = (uint32_t *)base; base += (size_t)qmax * 4;
  x.sqoffs = (uint32_t *)base; base += (size_t)qmax * 4;
  x.qbr = (uint32_t *)base; base += (size_t)qmax * 4;
  x.stride = qmax / (uint32_t)s + 2;
  x.frame_cnt = (uint32_t *)base; base += (size_t)s * 4;
  x.frame_rank = (uint32_t *)base; base += (size_t)s * x.stride * 4;
  x.kv = (uint32_t *)base; base += (size_t)qmax * 4;
  x.wk = (uint32_t *)base; base += (size_t)SEED_WSORT_WORDS * 4;
  x.qbuf = base;
  return x;
}

// returns the number of index lookups made (wave-uniform)
SMG_HD inline uint32_t stage_seed(const Batch &b, const DevIndex &ix, const MapPar &p, uint32_t r, uint32_t st, SeedScratch &x) {
  const uint32_t rs = 2 * r + st;
  const uint32_t qlen = read_len(b, r);
  const uint8_t *codes = b.codes + b.read_off[r];
  const uint8_t *qual = b.qual ? b.qual + b.read_off[r] : nullptr;
  uint8_t *qmask = b.qmask + (size_t)rs * b.qmax;
  SeedRec *out = b.seeds + (size_t)rs * b.qmax;
  HitInfoHdr &hdr = b.hi[rs];
  const int k = ix.k, s = ix.s;
  const bool noshort = (p.flags & FLG_NOSHRTINFO) != 0;
  const uint32_t ncut = noshort ? 0u : (uint32_t)(p.ncut > 0 ? p.ncut : 0);
  const int minqval = p.min_basq + 33;

  if (qlen < (uint32_t)k) {          // ERRCODE_SHORTSEQ, swallowed by rmapSingle (rmap.c:1736)
    SMG_LANE0 { hdr.n_seeds = 0; hdr.seed_rank = 0; hdr.status = st ? HI_REVERSE : 0; hdr.qlen = qlen; hdr.nhit_rank = hdr.nhit_tot = 0; hdr.nhit_cut = 0; }
    return 0;
  }
  const uint32_t nk = qlen - (uint32_t)k + 1;
  const uint64_t wordmask = (1ull << (2 * k)) - 1;
  // words start at offsets [t_lo, t_hi): the whole read, or the stretch a split-read call names (collectHitInfo, hashhit.c:536-551:
  // offsets in front of it count as without hits, the word window and the repeat filter start afresh at its first base)
  uint32_t t_lo = 0, t_hi = nk;
  if (b.seed_range) {
    const uint32_t q0 = b.seed_range[2 * r];
    uint32_t q1 = b.seed_range[2 * r + 1];
    if (q1 >= qlen) q1 = qlen - 1;
    if (q0 <= q1 && q1 - q0 + 1 >= (uint32_t)k) { t_lo = q0; t_hi = q1 - (uint32_t)k + 2; }
  }

  // (1) words + validity of every k-mer start t; ordered compaction of the valid ones.  The bases are staged once -- one byte per
  // base in the stage's scratch: the code, plus 4 if the base cannot be part of a word (non-ACGT, low quality) -- so that a lane
  // reads its k bases from there and not k code bytes and k quality bytes from memory (the word loop was 46 % of the kernel)
  SMG_PAR_CHUNKS(base, qlen) {
    const uint32_t i = base + SMG_LANE;
    if (i < qlen) { const uint32_t c = codes[i]; x.qbuf[i] = (uint8_t)((c & 3u) | (((c & 4u) || (qual && qual[i] < minqval)) ? 4u : 0u)); }
  }
  SMG_SYNC();
  uint32_t nvalid = 0;
  SMG_PAR_CHUNKS(base, t_hi) {
    uint32_t t = base + SMG_LANE;
    bool valid = false;
    uint64_t w = 0;
    if (t < t_lo) qmask[t] = HQ_NOHIT;
    else if (t < t_hi) {
      valid = true;
      for (int i = 0; i < k; i++) {
        const uint32_t c = x.qbuf[t + (uint32_t)i];
        if (c & 4) valid = false;
        // forward: first base in the top bits; reverse strand: complement of base t+i at bit 2i
        if (st) w |= ((uint64_t)((c ^ 3u) & 3u)) << (2 * i);
        else w = (w << 2) | (c & 3u);
      }
      w &= wordmask;
      if (!valid) qmask[t] = HQ_NONSTDNT;
    }
    uint32_t slot = compact_slot(valid, nvalid);
    if (valid) { x.vt[slot] = t; x.vw[slot] = w; }
  }
  SMG_PAR_CHUNKS(base, qlen - t_hi + 1) {         // tail offsets + terminator (hashhit.c:652-653)
    uint32_t t = t_hi + base + SMG_LANE;
    if (t <= qlen && t < b.qmax) qmask[t] = HQ_TERM;
  }
  SMG_SYNC();

  // (2) repeat filter over the previous NREPEATS looked-at words (hashhit.c:325-340), then lookup
  uint32_t nseeds = 0, nlook = 0;
  SMG_PAR_CHUNKS(base, nvalid) {
    uint32_t j = base + SMG_LANE;
    bool hit = false, looked = false;
    uint32_t t = 0, nh = 0, posidx = 0;
    if (j < nvalid) {
      t = x.vt[j];
      uint64_t w = x.vw[j];
      bool rep = false;
      for (uint32_t d = 1; d <= (uint32_t)NREPEATS && d <= j; d++) if (x.vw[j - d] == w) rep = true;
      if (rep) qmask[t] = HQ_REPEAT;
      else {
        nh = index_lookup(ix, w, &posidx);
        looked = true;
        if (nh < 1) qmask[t] = HQ_NOHIT;
        else if (ncut > 0 && nh > ncut) qmask[t] = HQ_MULTIHIT;
        else { qmask[t] = HQ_NORMHIT; hit = true; }
      }
    }
    (void)compact_slot(looked, nlook);
    uint32_t slot = compact_slot(hit, nseeds);
    if (hit) { x.key[slot] = nh; x.sidx[slot] = slot; x.sposidx[slot] = posidx; x.sqoffs[slot] = t; }
  }
  SMG_SYNC();

  if (b.totals_only) {            // calcTotalNumberOfHits (rmap.c:1076) needs the hit counts only: no ranking, no seed budget
    uint32_t tot = 0;
    const uint32_t hc = p.ncut > 0 ? (uint32_t)p.ncut : 0u;
    SMG_PAR_CHUNKS(base, nseeds) { const uint32_t i = base + SMG_LANE; if (i < nseeds && (!hc || x.key[i] <= hc)) tot += x.key[i]; }
    tot = wave_sum_u32(tot);
    SMG_LANE0 { hdr.n_seeds = nseeds; hdr.seed_rank = 0; hdr.status = st ? HI_REVERSE : 0; hdr.qlen = qlen; hdr.nhit_rank = hdr.nhit_tot = 0; hdr.nhit_cut = tot; }
    return nlook;
  }
  // (3) rarity ranking + seed budget (sort.c:233 tie order; getHitInfoMaxRank, hashhit.c:769-891).
  // Wave form for reads of up to 256 bases whose seeds and hit counts fit the packed sort element: the unstable
  // quicksort is emulated exactly (smg_wsort.hpp); the budget is a prefix sum over the sorted hit counts; the
  // per-frame coverage loops run one lane per sampling frame with the read coverage as a 256-bit register mask.
  uint32_t maxkey = 0;
  SMG_PAR_CHUNKS(base, nseeds) { const uint32_t i = base + SMG_LANE; if (i < nseeds && x.key[i] > maxkey) maxkey = x.key[i]; }
  maxkey = wave_max_u32(maxkey);
  const bool wave_form = !noshort && nseeds > 1 && qlen <= 256 && nseeds < (1u << SEED_IDXBITS) && maxkey < (1u << (32 - SEED_IDXBITS));
  if (wave_form) {
    SMG_PAR_CHUNKS(base, nseeds) { const uint32_t i = base + SMG_LANE; if (i < nseeds) x.kv[i] = (x.key[i] << SEED_IDXBITS) | i; }
    SMG_SYNC();
    wave_sort_kv<SEED_IDXBITS, SEED_LISTCAP, SEED_LSTK>(x.kv, (int)nseeds, (int)nseeds, x.wk);
    SMG_SYNC();
    uint32_t run = 0;                                   // inclusive prefix sums of the sorted hit counts -> x.wk
    SMG_PAR_CHUNKS(base, nseeds) {
      const uint32_t i = base + SMG_LANE;
      uint32_t v = 0;
      if (i < nseeds) { v = x.kv[i] >> SEED_IDXBITS; x.key[i] = v; x.sidx[i] = x.kv[i] & ((1u << SEED_IDXBITS) - 1u); x.qbr[i] = x.sqoffs[x.sidx[i]]; }
      uint32_t incl = v;
#if defined(__HIP_DEVICE_COMPILE__)
      for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, o); if ((int)SMG_LANE >= o) incl += u; }
#endif
      if (i < nseeds) x.wk[i] = run + incl;
#if defined(__HIP_DEVICE_COMPILE__)
      run += (uint32_t)__shfl((int)incl, 63);
#else
      run += incl;
#endif
    }
    SMG_SYNC();
    uint32_t mincover = (uint32_t)(HITINFO_MINCOVER_KMER * k + s);
    uint32_t maxcover = qlen * HITINFO_MAXCOVER_PERCENT / 100;
    if (maxcover < (uint32_t)(k + s)) maxcover = (uint32_t)(k + s);
    else if (maxcover > qlen - (uint32_t)s) maxcover = qlen - (uint32_t)s;
    if (mincover > maxcover) { mincover = 0; maxcover = qlen; }
    // n = first rank whose inclusive prefix exceeds HASH_MAXNHITS (hashhit.c:822-827), n_seeds if none
    uint32_t nbud = nseeds;
    SMG_PAR_CHUNKS(base, nseeds) { const uint32_t i = base + SMG_LANE; if (i < nseeds && x.wk[i] > (uint32_t)HASH_MAXNHITS && i < nbud) nbud = i; }
#if defined(__HIP_DEVICE_COMPILE__)
    for (int o = 32; o > 0; o >>= 1) { const uint32_t u = (uint32_t)__shfl_xor((int)nbud, o); if (u < nbud) nbud = u; }
#endif
    uint32_t nmax = nbud;
    // the ranks of every sampling frame as a list of its own (all lanes: a seed's frame, its place in the frame's list by
    // ballot), so that the lane of a frame walks its ~ n/s seeds and not all n (the walk was 40 % of the kernel)
    const uint32_t smagic = div_magic(s);               // q0 % s without a division (exact for offsets below 2^20)
    const bool listed = s <= 16;
    uint32_t fcnt[16];
#pragma unroll
    for (int ff = 0; ff < 16; ff++) fcnt[ff] = 0;
    if (listed) {
      SMG_PAR_CHUNKS(base, nseeds) {
        const uint32_t rk = base + SMG_LANE;
        uint32_t f = 0xffffffffu;
        if (rk < nseeds) { const uint32_t q0 = x.qbr[rk]; const uint32_t qd = smagic ? (uint32_t)(((uint64_t)q0 * smagic) >> 32) : q0; f = q0 - qd * (uint32_t)s; }
#pragma unroll
        for (int ff = 0; ff < 16; ff++) {
          if (ff < s) {                                  // (wave-uniform)
            const uint32_t slot = compact_slot(f == (uint32_t)ff, fcnt[ff]);
            if (f == (uint32_t)ff) x.frame_rank[(uint32_t)ff * x.stride + slot] = rk;
          }
        }
      }
      SMG_SYNC();
    }
    SMG_PAR_CHUNKS(base, (uint32_t)s) {                // one lane per sampling frame (:860-882)
      const uint32_t f = base + SMG_LANE;
      if (f < (uint32_t)s) {
        QMask256 mk;
        qm_clear(mk);
        uint32_t cover = 0;
        int last = -1;
        if (listed) {
          uint32_t nf = 0;
#pragma unroll
          for (int ff = 0; ff < 16; ff++) nf = f == (uint32_t)ff ? fcnt[ff] : nf;
          const uint32_t *mine = x.frame_rank + (size_t)f * x.stride;
          for (uint32_t i = 0; i < nf; i++) {
            const uint32_t rk = mine[i];
            if (!(cover <= maxcover && (cover < mincover || rk <= nbud))) break;
            cover += qm_add(mk, x.qbr[rk], (uint32_t)k - 1);      // k-1 bases (hashhit.c:873)
            last = (int)rk;
          }
        } else {
          for (uint32_t rk = 0; rk < nseeds; rk++) {
            const uint32_t q0 = x.qbr[rk];
            const uint32_t qd = smagic ? (uint32_t)(((uint64_t)q0 * smagic) >> 32) : q0;
            if (q0 - qd * (uint32_t)s != f) continue;
            if (!(cover <= maxcover && (cover < mincover || rk <= nbud))) break;
            cover += qm_add(mk, q0, (uint32_t)k - 1);
            last = (int)rk;
          }
        }
        if (last >= 0 && (uint32_t)last > nmax) nmax = (uint32_t)last;
      }
    }
    nmax = wave_max_u32(nmax);
    uint32_t seed_rank = nmax;
    if (nmax < (uint32_t)HITINFO_MINSEEDNUM) seed_rank = ((uint32_t)HITINFO_MINSEEDNUM < nseeds) ? (uint32_t)HITINFO_MINSEEDNUM : nseeds;
    SMG_LANE0 {
      const uint32_t ns = seed_rank > 0 ? seed_rank : nseeds;          // hashhit.c:1200-1219
      hdr.nhit_rank = ns ? x.wk[ns - 1] : 0;
      hdr.nhit_tot = x.wk[nseeds - 1];
      hdr.n_seeds = nseeds; hdr.seed_rank = seed_rank; hdr.status = (st ? HI_REVERSE : 0) | HI_SORTED | HI_RANK; hdr.qlen = qlen;
    }
  } else
  SMG_LANE0 {
    uint32_t status = st ? HI_REVERSE : 0, seed_rank = 0;
    if (!noshort) {
      if (nseeds <= 1) { status |= HI_SORTED; seed_rank = nseeds; }
      else {
        sort2_u32((int)nseeds, x.key, x.sidx);
        status |= HI_SORTED;
        for (uint32_t i = 0; i < nseeds; i++) x.qbr[i] = x.sqoffs[x.sidx[i]];
        uint32_t mincover = (uint32_t)(HITINFO_MINCOVER_KMER * k + s);
        uint32_t maxcover = qlen * HITINFO_MAXCOVER_PERCENT / 100;
        if (maxcover < (uint32_t)(k + s)) maxcover = (uint32_t)(k + s);
        else if (maxcover > qlen - (uint32_t)s) maxcover = qlen - (uint32_t)s;
        if (mincover > maxcover) { mincover = 0; maxcover = qlen; }
        build_frames(nseeds, x.qbr, s, x.frame_cnt, x.frame_rank, x.stride);
        seed_rank = seed_max_rank(nseeds, x.key, x.qbr, k, s, qlen, mincover, maxcover, (uint32_t)HASH_MAXNHITS,
                                  x.frame_cnt, x.frame_rank, x.stride, x.qbuf);
        status |= HI_RANK;
      }
    }
    uint32_t ns = seed_rank > 0 ? seed_rank : nseeds, nr = 0, i;   // hashhit.c:1200-1219
    for (i = 0; i < ns; i++) nr += x.key[i];
    hdr.nhit_rank = nr;
    for (; i < nseeds; i++) nr += x.key[i];
    hdr.nhit_tot = nr;
    hdr.n_seeds = nseeds; hdr.seed_rank = seed_rank; hdr.status = status; hdr.qlen = qlen;
  }
  SMG_SYNC();
  uint32_t ncutsum = 0;
  const uint32_t hcut = p.ncut > 0 ? (uint32_t)p.ncut : 0u;
  SMG_PAR_CHUNKS(base, nseeds) {
    uint32_t i = base + SMG_LANE;
    if (i < nseeds) {
      uint32_t src = x.sidx[i];
      SeedRec sr; sr.posidx = x.sposidx[src]; sr.nhits = x.key[i]; sr.qoffs = x.sqoffs[src];
      out[i] = sr;
      if (!hcut || sr.nhits <= hcut) ncutsum += sr.nhits;
    }
  }
  ncutsum = wave_sum_u32(ncutsum);
  SMG_LANE0 { hdr.nhit_cut = ncutsum; }
  return nlook;
}

// =======================================================================================
// stage_cands: S3 (hashhit.c:1416-1769), S4/S5 (segment.c:396-1223), S6 (:1616), S7 (:1861)
// =======================================================================================
struct CandScratch {          // per slot, in HBM
  uint64_t *keys;             // [hcap]  hit sort keys, then packed hit words grouped by (strand, seq)
  uint32_t hcap;              // power of two
  uint32_t *grp_first, *grp_cnt;   // [2 * ngrp]  group = strand * ngrp + seq
  uint32_t ngrp;              // nseq in sequence-by-sequence mode, else 1
  FillDecision *dec;          // [2 * ngrp]
  HitRegion *hreg; SegSeed *sseed; Segment *segm; uint32_t segcap;
  SegCand *cand; uint32_t candcap;
  uint32_t *sort_keys, *sort_idx;  // [candcap]
  uint8_t *mask;              // [qmax]
  uint8_t *hlmask;            // [2][qmax + 8] hit-list masks of the two strands (concatenated mode)
  uint32_t *qbr;              // [qmax]
  uint32_t *frame_cnt, *frame_rank; uint32_t stride;
  uint8_t *qbuf;              // [qmax]
};

SMG_HD inline size_t cand_scratch_bytes(uint32_t qmax, int s, uint32_t hcap, uint32_t ngrp, uint32_t segcap, uint32_t candcap) {
  uint32_t stride = qmax / (uint32_t)s + 2;
  size_t n = (size_t)hcap * 8 + (size_t)ngrp * 2 * (4 + 4 + sizeof(FillDecision)) + (size_t)segcap * (sizeof(HitRegion) + sizeof(SegSeed) + sizeof(Segment)) +
             (size_t)candcap * (sizeof(SegCand) + 8) + (size_t)qmax * (1 + 2 + 4 + 1) + 32 + (size_t)s * 4 + (size_t)s * stride * 4;
  return (n + 255) & ~(size_t)255;
}

SMG_HD inline CandScratch cand_scratch_carve(uint8_t *base, uint32_t qmax, int s, uint32_t hcap, uint32_t ngrp, uint32_t segcap, uint32_t candcap) {
  CandScratch x;
  x.hcap = hcap; x.ngrp = ngrp; x.segcap = segcap; x.candcap = candcap;
  x.keys = (uint64_t *)base; base += (size_t)hcap * 8;
  x.sseed = (SegSeed *)base; base += (size_t)segcap * sizeof(SegSeed);
  x.dec = (FillDecision *)base; base += (size_t)ngrp * 2 * sizeof(FillDecision);
  x.hreg = (HitRegion *)base; base += (size_t)segcap * sizeof(HitRegion);
  x.segm = (Segment *)base; base += (size_t)segcap * sizeof(Segment);
  x.cand = (SegCand *)base; base += (size_t)candcap * sizeof(SegCand);
  x.sort_keys = (uint32_t *)base; base += (size_t)candcap * 4;
  x.sort_idx = (uint32_t *)base; base += (size_t)candcap * 4;
  x.grp_first = (uint32_t *)base; base += (size_t)ngrp * 2 * 4;
  x.grp_cnt = (uint32_t *)base; base += (size_t)ngrp * 2 * 4;
  x.qbr = (uint32_t *)base; base += (size_t)qmax * 4;
  x.stride = qmax / (uint32_t)s + 2;
  x.frame_cnt = (uint32_t *)base; base += (size_t)s * 4;
  x.frame_rank = (uint32_t *)base; base += (size_t)s * x.stride * 4;
  x.mask = base; base += qmax;
  x.hlmask = base; base += 2 * ((size_t)qmax + 8);
  x.qbuf = base;
  return x;
}

// sequence of k-mer serial number `pos`: largest j with seqlo[j] <= pos (ties: last)
SMG_HD inline uint32_t seq_of_pos(const uint32_t *seqlo, int nseq, uint32_t pos) {
  uint32_t lo = 0, hi = (uint32_t)nseq;            // answer in [0, nseq-1]
  while (hi - lo > 1) { uint32_t m = (lo + hi) >> 1; if (seqlo[m] <= pos) lo = m; else hi = m; }
  return lo;
}

// returns the number of hits gathered (wave-uniform)
SMG_HD inline uint32_t stage_cands(const Batch &b, const DevIndex &ix, const MapPar &p, uint32_t r, CandScratch &x) {
  const uint32_t qlen = read_len(b, r);
  CandHdr &ch = b.ch[r];
  const int k = ix.k, s = ix.s;
  const bool seqbyseq = (p.flags & FLG_SEQBYSEQ) != 0;
  const uint32_t ngrp = x.ngrp;

  if (qlen < (uint32_t)k) {
    SMG_LANE0 { ch.ncand = ch.n_sort = ch.n_mincover = ch.max_cover = ch.max2nd_cover = 0; ch.cover_deficit[0] = ch.cover_deficit[1] = 0; ch.rc_off = 0; ch.err = 0; ch.err_site = 0; ch.nhits[0] = ch.nhits[1] = 0; ch.n_reserved = 0; }
    return 0;
  }
  // calcMinKtup (rmap.c:240-247) and the coverage threshold of mapSingleRead (:1283-1289)
  uint32_t min_cover = read_min_cover(p, qlen);
  const uint32_t min_ktup = (min_cover >= (uint32_t)(k + s)) ? (min_cover - (uint32_t)k) / (uint32_t)s : 1u;
  min_cover = (min_ktup - 1) * (uint32_t)s + (uint32_t)k;
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
  int err = 0;
  uint32_t nkeys = 0;

  // ---- S3: decide which ranked seeds contribute per (strand, sequence), then gather ----
  for (uint32_t st = 0; st < 2; st++) {
    const uint32_t rs = 2 * r + st;
    const HitInfoHdr hdr = b.hi[rs];
    const SeedRec *seeds = b.seeds + (size_t)rs * b.qmax;
    uint8_t *qmask = b.qmask + (size_t)rs * b.qmax;
    const uint32_t n_use = hdr.seed_rank > 0 ? hdr.seed_rank : hdr.n_seeds;
    // total hits of the usable seeds
    uint32_t tot = 0;
    SMG_PAR_CHUNKS(base, n_use) {
      uint32_t n = base + SMG_LANE;
      if (n < n_use && !(ncut > 0 && seeds[n].nhits > ncut)) tot += seeds[n].nhits;
    }
    tot = wave_sum_u32(tot);
    FillDecision *dec = x.dec + st * ngrp;
    if (seqbyseq) {
      if (tot <= (uint32_t)nhits_alloc) {
        SMG_PAR_CHUNKS(base, ngrp) { uint32_t g = base + SMG_LANE; if (g < ngrp) { dec[g].n_used = n_use; dec[g].m_final = ncut; } }
        SMG_PAR_CHUNKS(base, n_use) {     // hashhit.c:1472-1481: over-cut seeds are flagged (-x mode only)
          uint32_t n = base + SMG_LANE;
          if (n < n_use && ncut > 0 && seeds[n].nhits > ncut) qmask[seeds[n].qoffs] = HQ_MULTIHIT;
        }
      } else {
        // rare: replay the allocation-boundary retry protocol per sequence (one lane each)
        SMG_PAR_CHUNKS(base, ngrp) {
          uint32_t g = base + SMG_LANE;
          if (g < ngrp) {
            uint64_t lo = ix.sop[g] / (uint64_t)s, hi = ix.sop[g + 1] / (uint64_t)s;
            if (hi > 0xFFFFFFFFull) hi = 0xFFFFFFFFull;
            dec[g] = fill_decide(ix, seeds, n_use, (uint32_t)lo, (uint32_t)hi, ncut, nhits_alloc, qmask);
          }
        }
      }
    } else {
      // hashCollectHitsUsingCutoff (hashhit.c:1593-1689): ceiling retry on the summed hits
      SMG_LANE0 {
        uint32_t m = ncut, n_used = 0;
        for (;;) {
          uint32_t total = 0, i;
          bool ceiling = false;
          uint8_t *hlmask = x.hlmask + (size_t)st * (b.qmax + 8);
          for (uint32_t q = 0; q < qlen; q++) hlmask[q] = HQ_NOHIT;
          hlmask[qlen] = 0;
          for (i = 0; i < n_use; i++) {
            uint32_t nh = seeds[i].nhits;
            if (nh < 1) continue;
            if (m > 0 && nh > m) { hlmask[seeds[i].qoffs] = HQ_MULTIHIT; continue; }
            if ((int)(total + nh) > nhits_max) { ceiling = true; break; }
            hlmask[seeds[i].qoffs] = HQ_NORMHIT;
            total += nh;
          }
          n_used = i;
          uint32_t mf = m;
          m /= 2;
          if (!(ceiling && m > (uint32_t)MINHIT_PER_TUPLE)) { dec[0].n_used = n_used; dec[0].m_final = mf; break; }
        }
      }
    }
    SMG_SYNC();
    // gather: lanes stride over the position list of one seed at a time (coalesced pos[] reads)
    const uint32_t key_first = nkeys;
    for (uint32_t n = 0; n < n_use; n++) {
      const SeedRec sd = seeds[n];
      const uint32_t *posp;
      const uint32_t nh = index_positions(ix, sd.posidx, &posp);
      if (!seqbyseq) {
        if (n >= dec[0].n_used || (dec[0].m_final > 0 && sd.nhits > dec[0].m_final) || sd.nhits < 1) continue;
        if (nkeys + nh > x.hcap) { err = SMG_ERR_CAP; break; }
        SMG_PAR_CHUNKS(base, nh) {
          uint32_t i = base + SMG_LANE;
          if (i < nh) x.keys[nkeys + i] = ((uint64_t)st << 63) | (hit_diag(st != 0, posp[i], sd.qoffs, s) << KEY_QBITS) | sd.qoffs;
        }
        nkeys += nh;
      } else {
        if (ncut > 0 && sd.nhits > ncut) continue;
        uint32_t cnt = 0;
        SMG_PAR_CHUNKS(base, nh) {
          uint32_t i = base + SMG_LANE;
          bool take = false;
          uint64_t key = 0;
          if (i < nh) {
            uint32_t pos = posp[i];
            uint32_t g = seq_of_pos(ix.seqlo, ix.nseq, pos);
            const FillDecision d = dec[g];
            take = n < d.n_used && !(d.m_final > 0 && sd.nhits > d.m_final) && pos < ix.seqlo[ix.nseq];
            key = ((uint64_t)st << 63) | ((uint64_t)g << (KEY_DIAGBITS + KEY_QBITS)) | (hit_diag(st != 0, pos, sd.qoffs, s) << KEY_QBITS) | sd.qoffs;
          }
          uint32_t slot = compact_slot(take && nkeys + cnt < x.hcap, cnt);
          if (take && nkeys + slot < x.hcap) x.keys[nkeys + slot] = key;
          else if (take) err = SMG_ERR_CAP;
        }
        nkeys += cnt;
      }
    }
    SMG_LANE0 { ch.nhits[st] = nkeys - key_first; }
    SMG_SYNC();
  }
  if (wave_any(err != 0)) err = SMG_ERR_CAP;
  // ---- sort by (strand, seq, diagonal, q): only the multiset matters (sort.c:415 sorts plain words)
  wave_sort_u64(x.keys, nkeys);
  SMG_SYNC();
  // group table + conversion to the reference's packed hit word (diagonal << 31 | q)
  SMG_PAR_CHUNKS(base, 2 * ngrp) { uint32_t g = base + SMG_LANE; if (g < 2 * ngrp) { x.grp_first[g] = 0; x.grp_cnt[g] = 0; } }
  SMG_SYNC();
  SMG_PAR_CHUNKS(base, nkeys) {
    uint32_t i = base + SMG_LANE;
    if (i < nkeys) {
      uint64_t key = x.keys[i];
      uint32_t g = (uint32_t)(key >> (KEY_DIAGBITS + KEY_QBITS));     // strand * 1024 + seq
      uint32_t gi = (g >> KEY_SEQBITS) * ngrp + (g & ((1u << KEY_SEQBITS) - 1));
      bool first = (i == 0) || ((uint32_t)(x.keys[i - 1] >> (KEY_DIAGBITS + KEY_QBITS)) != g);
      bool last = (i + 1 == nkeys) || ((uint32_t)(x.keys[i + 1] >> (KEY_DIAGBITS + KEY_QBITS)) != g);
      if (first) x.grp_first[gi] = i;
      if (last) x.grp_cnt[gi] = i + 1;     // end index for now
    }
  }
  SMG_SYNC();
  SMG_PAR_CHUNKS(base, nkeys) {
    uint32_t i = base + SMG_LANE;
    if (i < nkeys) {
      uint64_t key = x.keys[i];
      x.keys[i] = (((key >> KEY_QBITS) & KEY_DIAGMASK) << HALFBIT) | (key & KEY_QMASK);
    }
  }
  SMG_PAR_CHUNKS(base, 2 * ngrp) {
    uint32_t g = base + SMG_LANE;
    if (g < 2 * ngrp && x.grp_cnt[g]) x.grp_cnt[g] -= x.grp_first[g];
  }
  SMG_SYNC();

  // ---- S4-S6 on one lane: the candidate order feeds an unstable sort and must be the reference's ----
  SMG_LANE0 {
    CandSet cs; cs.cand = x.cand; cs.ncand = 0; cs.cap = x.candcap; cs.max_cover = cs.max2nd_cover = 0;
    SegLst sl; sl.hreg = x.hreg; sl.seed = x.sseed; sl.segm = x.segm; sl.cap = x.segcap; sl.nhreg = sl.nseed = sl.nsegm = 0;
    for (uint32_t st = 0; st < 2 && !err; st++) {
      for (uint32_t g = 0; g < ngrp && !err; g++) {
        uint32_t cnt = x.grp_cnt[st * ngrp + g];
        if (!cnt && seqbyseq) continue;
        const uint64_t *dat = x.keys + x.grp_first[st * ngrp + g];
        if (seglst_fill(sl, min_ktup, dat, (int)cnt, qlen, seqbyseq ? nullptr : x.hlmask + (size_t)st * (b.qmax + 8), k, s)) { err = SMG_ERR_CAP; break; }
        int rv = cands_add_fast(cs, x.mask, sl, qlen, k, s, st != 0, min_cover, seqbyseq ? (int32_t)g : -1);
        if (rv) { err = (rv == -2) ? SMG_ERR_CAP : SMG_ERR_ASSERT; break; }
      }
    }
    // cover deficits (hashhit.c:1096) need the frames of each strand
    uint32_t cdf[2] = {0, 0};
    for (uint32_t st = 0; st < 2; st++) {
      const uint32_t rs = 2 * r + st;
      const HitInfoHdr hdr = b.hi[rs];
      const SeedRec *seeds = b.seeds + (size_t)rs * b.qmax;
      if (hdr.status & HI_RANK) {
        for (uint32_t i = 0; i < hdr.n_seeds; i++) x.qbr[i] = seeds[i].qoffs;
        build_frames(hdr.n_seeds, x.qbr, s, x.frame_cnt, x.frame_rank, x.stride);
      }
      cdf[st] = cover_deficit(hdr.status, hdr.seed_rank, qlen, b.qmask + (size_t)rs * b.qmax, x.qbr, k, s, x.frame_cnt,
                              x.frame_rank, x.stride, x.qbuf);
    }
    uint32_t n_mincover = 0, n_sort = 0;
    if (!err && cands_stats(cs, cdf[0], s, mincov_below_max, (uint32_t)p.target_depth, (uint32_t)p.max_depth,
                            (p.flags & FLG_SENSITIVE) != 0, x.sort_keys, x.sort_idx, &n_mincover, &n_sort)) err = SMG_ERR_ASSERT;
    ch.ncand = cs.ncand; ch.n_sort = err ? 0 : n_sort; ch.n_mincover = n_mincover;
    ch.max_cover = cs.max_cover; ch.max2nd_cover = cs.max2nd_cover;
    ch.cover_deficit[0] = cdf[0]; ch.cover_deficit[1] = cdf[1];
    ch.err = err; ch.err_site = err ? __LINE__ : 0;
    ch.n_reserved = ch.n_sort;
    ch.rc_off = atomic_add_u32(b.rc_count, ch.n_sort);
    if ((uint64_t)ch.rc_off + ch.n_sort > b.rccap) { ch.err = SMG_ERR_CAP; ch.err_site = __LINE__; ch.n_sort = 0; }
  }
  SMG_SYNC();
  // ---- S7: windows and bands of the ranked candidates ----
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
      if (cand_offsets(c, x.cand[x.sort_idx[i]], ix, qlen)) { c.flags |= RCF_ERR; c.rs = c.re = 0; c.qs = c.qe = 0; c.band_l = c.band_r = 0; }
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
  return nkeys;
}

// =======================================================================================
// K2b on one lane: alignSmiWatBandFast (alignment.c:1029-1233) over a window read straight
// from the packed reference.  Hp/Ep: q_totlen + 2 ints each.  NB unlike K3 the left band
// edge never advances once it starts clipped at q_left (delta_band_start is not decremented).
// =======================================================================================
SMG_HD inline int band_fast_scalar(const Band &bp, const uint8_t *q, const uint32_t *packed, uint64_t rbase,
                                   const int8_t *M /* [8][8] */, int gi, int ge, int *Hp, int *Ep) {
  int delta_start, j_start, j_len, currH = 0, best = 0;
  if (bp.q_left > bp.l_edge) { delta_start = bp.q_left - bp.l_edge; j_start = bp.q_left; }
  else { delta_start = 0; j_start = bp.l_edge; }
  j_len = bp.r_edge + 1;
  for (int j = j_start; j < bp.q_len; j++) Hp[j] = Ep[j] = 0;
  for (int i = bp.s_left; i < bp.s_len; i++) {
    const int8_t *w = M + 8 * ref_code(packed, rbase + (uint64_t)i);
    int F = 0;
    for (int j = j_start; j < j_len; j++) {
      int H = currH + w[q[j] & 7];
      currH = Hp[j];
      bool cand;
      cell_update(Hp[j], Ep[j], F, H, gi, ge, cand);
      if (cand && H > best) best = H;
    }
    if (delta_start > 0) currH = 0;
    else { currH = Hp[j_start]; j_start++; }
    if (j_len < bp.q_len) j_len++;
  }
  return best;
}

// K2a on one lane (fallback for reads the DPP kernel does not cover): textbook Gotoh maximum
// (swsimd.c:868).  H/E: qlen ints each.
SMG_HD inline int sw_full_scalar(const uint8_t *q, uint32_t qlen, const uint32_t *packed, uint64_t rbase, uint32_t rlen,
                                 const int8_t *M, int gi, int ge, int *H, int *E) {
  int best = 0;
  for (uint32_t j = 0; j < qlen; j++) H[j] = E[j] = 0;
  for (uint32_t i = 0; i < rlen; i++) {
    const int8_t *w = M + 8 * ref_code(packed, rbase + i);
    int diag = 0, F = 0;
    for (uint32_t j = 0; j < qlen; j++) {
      int h = diag + w[q[j] & 7];
      if (h < 0) h = 0;
      if (h > best) best = h;
      if (E[j] > h) h = E[j];
      if (F > h) h = F;
      diag = H[j];
      H[j] = h;
      int t = h - gi;
      int e = E[j] - ge; if (e < t) e = t; if (e < 0) e = 0; E[j] = e;
      F -= ge; if (F < t) F = t; if (F < 0) F = 0;
    }
  }
  return best;
}

SMG_HD inline void score_matrix(int8_t *M /* [8][8] */, int match, int mismatch) {   // score.c:138-173
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) {
      int v;
      if (i >= 6 || j >= 6 || i == 5 || j == 5) v = 0;          // alphabet "ACGTXN": N rows/cols 0
      else if (i == 4 || j == 4) v = mismatch - match;          // X
      else v = (i == j) ? match : mismatch;
      M[i * 8 + j] = (int8_t)v;
    }
}

// =======================================================================================
// stage_replay: O1 for one read (one THREAD per read)
// =======================================================================================
SMG_HD inline void stage_replay(const Batch &b, const DevIndex &ix, const MapPar &p, uint32_t r) {
  const CandHdr ch = b.ch[r];
  ReadCtl ctl;
  ctl.pad = 0;
  MapPar q = p;
  if (b.min_sw) q.min_swatscor = b.min_sw[r];          // rmapPair passes its own threshold to some calls (rmap.c:2031)
  replay_scores(ctl, b.rcpool + ch.rc_off, ch.n_sort, ch.cover_deficit, q, ix.s, ix.k, read_len(b, r));
  if (ch.err) ctl.go = 0;
  b.ctl[r] = ctl;
}

// =======================================================================================
// stage_align: alignRMAPCANDFull (rmap.c:790-928) -> aliSmiWatInBand (alignment.c:1548) ->
// alignSmiWatBandRecursive (:1300) -> alignSmiWatBand (:788) + makeMetaFromTrack (:628) ->
// resultSetAddFromAli (results.c:1852).
// =======================================================================================
struct AlignScratch {
  int *Hp, *Ep;               // [qmax + 2]
  uint8_t *win; uint32_t wincap;
  uint8_t *dir; uint64_t dircap;        // direction matrix in the HBM slot
  uint8_t *dir_lds; uint32_t dir_lds_cap;   // ... and its LDS home for bands that fit (0 on the host build)
  uint8_t *dtmp; uint32_t dtmpcap;      // reversed DiffStr of the current traceback
  unsigned long long tally[8];          // phase ticks and counts of the reads this workgroup has aligned (flushed to Batch::work once, by the kernel)
  Result *res; uint32_t rescap;
  uint8_t *dstr; uint32_t dstrcap;
  int *ivstack;               // [2 * 64] pending reference intervals
  int32_t *state;             // [16] lane-0 state visible to the wave
  uint8_t *qcodes; uint32_t qstride;   // [2][qstride] the read in both orientations
  uint8_t *win_lds, *dtmp_lds; uint32_t win_lds_cap;   // LDS copies for windows of ordinary length (else the HBM arrays)
  int pass;                   // 0: only pass; 1: first of two (a band that does not fit defers the read); 2: second (deferred reads only)
  int rows_form;              // 1: bands of up to 64 diagonals row by row (band_track_rows); 0: anti-diagonal form (SMALTGPU_ALIGN_ANTIDIAG, tests)
  void *ring;                 // LDS [256] (H, F) pairs of band_track_strip (set by k_align<true>)
  void *bnd; uint32_t bndcap; // [2 * bndcap] (H, F) pairs: hand-over between the strips of band_track_strip (long reads only)
};

SMG_HD inline size_t align_scratch_bytes(uint32_t qmax, uint32_t wincap, uint64_t dircap, uint32_t rescap, uint32_t dstrcap) {
  size_t n = ((size_t)qmax + 2) * 8 + wincap + dircap + ((size_t)qmax + wincap + 16) + (size_t)rescap * sizeof(Result) + dstrcap + 128 * 4 + 64 + 256 + 2 * ((size_t)qmax + 8);
  if (qmax > 256) n += (size_t)wincap * 16 + 64;
  return (n + 255) & ~(size_t)255;
}

SMG_HD inline AlignScratch align_scratch_carve(uint8_t *base, uint32_t qmax, uint32_t wincap, uint64_t dircap, uint32_t rescap, uint32_t dstrcap) {
  AlignScratch x;
  x.rows_form = 0;
  x.res = (Result *)base; base += (size_t)rescap * sizeof(Result); x.rescap = rescap;
  x.Hp = (int *)base; base += ((size_t)qmax + 2) * 4;
  x.Ep = (int *)base; base += ((size_t)qmax + 2) * 4;
  x.ivstack = (int *)base; base += 128 * 4;
  x.state = (int32_t *)base; base += 64;
  x.qstride = qmax + 8;
  x.qcodes = base; base += 2 * (size_t)x.qstride;
  x.win = base; base += wincap; x.wincap = wincap;
  x.dtmpcap = qmax + wincap + 16;
  x.dtmp = base; base += x.dtmpcap;
  x.dstr = base; base += dstrcap; x.dstrcap = dstrcap;
  x.bnd = nullptr; x.bndcap = 0; x.ring = nullptr;
  if (qmax > 256) { base = (uint8_t *)(((uintptr_t)base + 15) & ~(uintptr_t)15); x.bnd = (void *)base; x.bndcap = wincap; base += (size_t)wincap * 16; }
  base = (uint8_t *)(((uintptr_t)base + 15) & ~(uintptr_t)15);
  x.dir = base; x.dircap = dircap;
  x.dir_lds = nullptr; x.dir_lds_cap = 0;
  x.win_lds = x.dtmp_lds = nullptr; x.win_lds_cap = 0;
  for (int i = 0; i < 8; i++) x.tally[i] = 0;
  x.pass = 0;
  return x;
}

// Same, with the small hot arrays (window and reversed DiffStr of ordinary-sized candidates, interval stack, state,
// read codes) and the direction matrix of ordinary-sized bands in the workgroup's LDS block.  The block is kept
// small (launch_align): the kernel is LDS-latency bound and gains from every additional resident wave.
SMG_HD inline uint32_t align_lds_wincap(uint32_t qmax, uint32_t wincap) {
  uint32_t w = qmax + qmax / 2 + 96;                 // read + band + edges of an ordinary candidate
  return w < wincap ? w : wincap;
}
SMG_HD inline size_t align_lds_small_bytes(uint32_t qmax, uint32_t wincap) {
  const uint32_t wl = align_lds_wincap(qmax, wincap);
  return (128 * 4 + 64 + 2 * ((size_t)qmax + 8) + wl + ((size_t)qmax + wl + 16) + 63) & ~(size_t)63;
}
SMG_HD inline AlignScratch align_scratch_carve_lds(uint8_t *lds, size_t lds_bytes, uint8_t *base, uint32_t qmax, uint32_t wincap,
                                                   uint64_t dircap, uint32_t rescap, uint32_t dstrcap) {
  AlignScratch x = align_scratch_carve(base, qmax, wincap, dircap, rescap, dstrcap);
  const size_t small = align_lds_small_bytes(qmax, wincap);
  if (lds && small + 1024 <= lds_bytes) {
    uint8_t *l = lds;
    x.ivstack = (int *)l; l += 128 * 4;
    x.state = (int32_t *)l; l += 64;
    x.qcodes = l; l += 2 * (size_t)x.qstride;
    x.win_lds_cap = align_lds_wincap(qmax, wincap);
    x.win_lds = l; l += x.win_lds_cap;
    x.dtmp_lds = l;
    x.dir_lds = lds + small;
    x.dir_lds_cap = (uint32_t)(lds_bytes - small);
  }
  return x;
}

// alignSmiWatBand (alignment.c:788-1027) on one lane; returns max score, sets max_i/max_j
SMG_HD inline int band_track_scalar(const Band &bp, const uint8_t *q, const uint8_t *win, const int8_t *M, int gi, int ge,
                                    int *Hp, int *Ep, uint8_t *dir, int *max_i, int *max_j) {
  int delta_start, delta_end = 0, j_start, j_len, currH = 0, best = 0, mi = 0, mj = 0;
  if (bp.q_left > bp.l_edge) { delta_start = bp.q_left - bp.l_edge; j_start = bp.q_left; }
  else { delta_start = 0; j_start = bp.l_edge; }
  j_len = bp.r_edge + 1;
  uint8_t *dirp = dir + delta_start;
  for (int j = j_start; j < bp.q_len; j++) Hp[j] = Ep[j] = 0;
  for (int i = bp.s_left; i < bp.s_len; i++) {
    const int8_t *w = M + 8 * (win[i] & 7);
    int F = 0;
    for (int j = j_start; j < j_len; j++, dirp++) {
      int H = currH + w[q[j] & 7];
      currH = Hp[j];
      bool cand;
      *dirp = (uint8_t)cell_update(Hp[j], Ep[j], F, H, gi, ge, cand);
      if (cand && H > best) { mi = i; mj = j; best = H; }
    }
    if (delta_start > 0) { currH = 0; dirp += --delta_start; }
    else { currH = Hp[j_start]; j_start++; }
    if (j_len < bp.q_len) j_len++;
    else dirp += delta_end++;
  }
  *max_i = mi; *max_j = mj;
  return best;
}

// makeMetaFromTrack (alignment.c:628-781): traceback into a REVERSED DiffStr; returns its
// length (with terminator) or < 0
// ---- strip form of the band pass (wide bands of long reads) ---------------------------------------------------
// Columns are cut into strips of 64 x 16 (origin j0: the band's first column rounded down to 16); lane g owns 16
// consecutive columns and sweeps the strip's rows one row behind lane g-1.  Row r visits [js(r), jl(r)) with
// js(r) = max(q_left, l_edge + r) and jl(r) = min(r_edge + 1 + r, q_len); strip s sweeps the rows [r_lo, r_hi) in which
// it has visited cells.  Directions take 2 bits: a lane packs its 16 cells of a row into one word and the wave
// writes word (t, g) of the strip at step t -- 256 contiguous bytes per step.
struct StripGeom { int j0, l, r, nrows, nstrip, cbits; };   // cbits: log2 of the columns per lane (3 or 4)
SMG_HD inline StripGeom strip_geom(const Band &bp, int cbits) {
  StripGeom sg;
  sg.cbits = cbits;
  const int cw = 64 << cbits;                  // columns per strip
  const int jmin = bp.q_left > bp.l_edge ? bp.q_left : bp.l_edge;
  sg.l = bp.l_edge; sg.r = bp.r_edge; sg.nrows = bp.s_len - bp.s_left;
  int jmax = bp.r_edge + sg.nrows; if (jmax > bp.q_len) jmax = bp.q_len;      // one past the last column visited
  sg.j0 = jmin & ~((1 << cbits) - 1);
  sg.nstrip = (sg.nrows > 0 && jmax > sg.j0) ? (jmax - sg.j0 + cw - 1) / cw : 0;
  return sg;
}
SMG_HD inline void strip_rows(const StripGeom &sg, int sidx, int *r_lo, int *r_hi) {
  const int c_lo = sg.j0 + sidx * (64 << sg.cbits), c_hi = c_lo + (64 << sg.cbits);
  int lo = c_lo - sg.r; if (lo < 0) lo = 0;
  int hi = c_hi - sg.l; if (hi > sg.nrows) hi = sg.nrows;
  *r_lo = lo; *r_hi = hi;
}
SMG_HD inline uint64_t strip_words(const StripGeom &sg, int upto) {          // direction words of the strips [0, upto)
  uint64_t n = 0;
  for (int sidx = 0; sidx < upto; sidx++) { int lo, hi; strip_rows(sg, sidx, &lo, &hi); if (hi > lo) n += (uint64_t)(hi - lo + 63) * 64; }
  return n;
}
// direction of cell (ip, j); cache: {strip, its first word, its first row} of the previous call
SMG_HD inline int strip_dir(const StripGeom &sg, const uint32_t *dirw, int ip, int j, int *cache_sidx, uint64_t *cache_base, int *cache_rlo) {
  const int o = j - sg.j0, sidx = o >> (6 + sg.cbits), g = (o >> sg.cbits) & 63, cc = o & ((1 << sg.cbits) - 1);
  if (sidx != *cache_sidx) { int hi; *cache_sidx = sidx; *cache_base = strip_words(sg, sidx); strip_rows(sg, sidx, cache_rlo, &hi); }
  const uint32_t w = dirw[*cache_base + (uint64_t)(ip - *cache_rlo + g) * 64 + (uint64_t)g];
  return (int)((w >> (2 * cc)) & 3u);
}

// Direction bytes of a band pass.  tW == 0: the reference's layout (row-major, band_width - 1 bytes per row).
// tW > 0: anti-diagonal-major, tW bytes per step t = row + column -- the cells the wave computes in one step are
// neighbours in memory (at most band_width / 2 + 1 columns are live per step), which is what a direction matrix
// in HBM needs; only cells the pass has computed are ever read back, so the layout is private to the two routines.
SMG_HD inline size_t dir_index(const Band &bp, int tW, int ip, int j) {
  if (!tW) return (size_t)ip * (size_t)(bp.band_width - 1) + (size_t)(j - bp.l_edge);
  const int jmin = bp.q_left > bp.l_edge ? bp.q_left : bp.l_edge;
  return (size_t)(ip + (j - jmin)) * (size_t)tW + (size_t)((j - jmin) & (tW - 1));     // tW: a power of two
}

SMG_HD inline int traceback_scalar(uint8_t *ds, uint32_t dscap, int *qs, int *rs, const Band &bp, const uint8_t *dir,
                                   int max_i, int max_j, int max_scor, const uint8_t *q, const uint8_t *win,
                                   const int8_t *M, int gi, int ge, int tW = 0, const StripGeom *sg = nullptr) {
  uint32_t n = 0;
  const int tb_match = M[0], tb_mismatch = M[1];   // (the matrix is read twice here, not once per step: it lives in scratch memory on the device)
  int c_sidx = -1, c_rlo = 0;                  // strip layout (tW < 0): directions come from strip_dir
  uint64_t c_base = 0;
  int i, j, checksum = 0;
  bool gap_open = false;
  uint8_t nmatch = 0;
  const uint8_t *dp = dir + (tW < 0 ? 0 : dir_index(bp, tW, max_i - bp.s_left, max_j));
#define SMG_PUT(cnt, typ) { if (n + 2 >= dscap) return -2; ds[n++] = (uint8_t)((cnt) + ((typ) << DIFF_TYPSHIFT)); }
  if (tW) {
    for (i = max_i, j = max_j; i >= bp.s_left && j >= bp.q_left;) {
      const uint8_t d = tW < 0 ? (uint8_t)strip_dir(*sg, (const uint32_t *)dir, i - bp.s_left, j, &c_sidx, &c_base, &c_rlo) : dir[dir_index(bp, tW, i - bp.s_left, j)];
      if (!d) break;
      if (d == DIR_DIA) {
        const int rbc = win[i] & 7, qcc = q[j] & 7;
        int s = (rbc >= 4 || qcc >= 4) ? 0 : (rbc == qcc ? tb_match : tb_mismatch);     // score.c:138-173 on the codes that occur (0-3, 5 = N)
        if (s > 0) {
          if (nmatch > DIFF_MAXMISMATCH) { SMG_PUT(DIFF_MAXMISMATCH, DIFF_M) nmatch -= DIFF_MAXMISMATCH; }
          else nmatch++;
        } else { SMG_PUT(nmatch, DIFF_S) nmatch = 0; }
        checksum += s;
        gap_open = false;
        i--; j--;
        continue;
      }
      if (gap_open) checksum -= ge; else { checksum -= gi; gap_open = true; }
      if (d & DIR_COL) { SMG_PUT(nmatch, DIFF_D) nmatch = 0; i--; continue; }
      if (!(d & DIR_ROW)) return -1;
      SMG_PUT(nmatch, DIFF_I) nmatch = 0; j--;
    }
  } else
  for (i = max_i, j = max_j; i >= bp.s_left && j >= bp.q_left && *dp;) {
    if (*dp == DIR_DIA) {
      const int rbc = win[i] & 7, qcc = q[j] & 7;
        int s = (rbc >= 4 || qcc >= 4) ? 0 : (rbc == qcc ? tb_match : tb_mismatch);     // score.c:138-173 on the codes that occur (0-3, 5 = N)
      if (s > 0) {
        if (nmatch > DIFF_MAXMISMATCH) { SMG_PUT(DIFF_MAXMISMATCH, DIFF_M) nmatch -= DIFF_MAXMISMATCH; }
        else nmatch++;
      } else { SMG_PUT(nmatch, DIFF_S) nmatch = 0; }
      checksum += s;
      gap_open = false;
      dp -= bp.band_width; i--; j--;
      continue;
    }
    if (gap_open) checksum -= ge; else { checksum -= gi; gap_open = true; }
    if (*dp & DIR_COL) { SMG_PUT(nmatch, DIFF_D) nmatch = 0; dp -= bp.band_width - 1; i--; continue; }
    if (!(*dp & DIR_ROW)) return -1;
    SMG_PUT(nmatch, DIFF_I) nmatch = 0; dp--; j--;
  }
  SMG_PUT(nmatch, DIFF_S)
  SMG_PUT(0, DIFF_M)
#undef SMG_PUT
  *rs = i + 1; *qs = j + 1;
  return (checksum != max_scor) ? -3 : (int)n;      // ERRCODE_SWATSCOR (alignment.c:767): the path's score is not the pass's maximum
}

#if defined(__HIPCC__)
__device__ inline int wave_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true); }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline int wave_ror1(int v) {      // value of lane-1 (lane 0 takes lane 63)
  return __builtin_amdgcn_update_dpp(v, v, 0x13C /* wave_ror:1 */, 0xf, 0xf, false);
}

// alignSmiWatBand (alignment.c:788-1027) in strip form; bnd: 2 x wcap (H, F) pairs in HBM for the hand-over between
// strips (as sw_strip_core, smg_kernels.hip).  Cells outside the band keep H and E and pass F = 0: what a visited cell
// reads from an unvisited neighbour is then what the reference's row buffers hold (see band_fast_wave).
template <class PW, int C>
__device__ inline int band_track_strip(const Band &bp, const StripGeom &sg, PW q, PW win, int match, int mismatch, int gi, int ge,
                                       uint32_t *dirw, int2 *bnd, uint32_t wcap, int2 *ring /* LDS [256] */, int *max_i, int *max_j) {
  const int g = (int)threadIdx.x;
  int2 *ring_in = ring, *ring_out = ring + 128;
  int best = 0, bi = 0, bj = 0, plo = 0, phi = 0;
  uint64_t sbase = 0;
  for (int sidx = 0; sidx < sg.nstrip; sidx++) {
    int r_lo, r_hi;
    strip_rows(sg, sidx, &r_lo, &r_hi);
    const int nr = r_hi - r_lo;
    if (nr <= 0) continue;
    const int2 *bprev = bnd + (size_t)((sidx + 1) & 1) * wcap;
    int2 *bnext = bnd + (size_t)(sidx & 1) * wcap;
    const int jb = sg.j0 + sidx * (64 * C) + g * C;
    int qc[C];
#pragma unroll
    for (int cc = 0; cc < C; cc++) qc[cc] = (jb + cc >= 0 && jb + cc < bp.q_len) ? (int)(q[jb + cc] & 7) : 5;
    int H[C], E[C];
#pragma unroll
    for (int cc = 0; cc < C; cc++) { H[cc] = 0; E[cc] = 0; }
    int F = 0, prev_hl = 0, sbest = 0, sbi = 0, sbj = 0;
#define SMG_BGET(rr) ((sidx > 0 && (rr) >= plo && (rr) < phi) ? bprev[(rr)] : make_int2(0, 0))
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
      const int rb = rowok ? (int)(win[bp.s_left + row] & 7) : 5;
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
        const int jsr = max(bp.q_left, sg.l + row);
        const int jlr = min(sg.r + 1 + row, bp.q_len);
        const int wm = rb >= 4 ? 0 : match, wx = rb >= 4 ? 0 : mismatch;
        uint32_t dw = 0;
#pragma unroll
        for (int cc = 0; cc < C; cc++) {
          const int hold = H[cc];
          const int j = jb + cc;
          if (j >= jsr && j < jlr) {
            const int hin = diag + (qc[cc] >= 4 ? 0 : (qc[cc] == rb ? wm : wx));
            bool cand;
            const int d = cell_update(H[cc], E[cc], F, hin, gi, ge, cand);
            dw |= (uint32_t)d << (2 * cc);
            if (cand && hin > sbest) { sbest = hin; sbi = row; sbj = j; }
          } else F = 0;
          diag = hold;
        }
        dirw[sbase + (uint64_t)step * 64 + (uint64_t)g] = dw;
      } else F = 0;
      if (sidx + 1 < sg.nstrip) {                    // hand the last column to the next strip
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
    if (sbest > best || (sbest == best && sbest > 0 && (sbi < bi || (sbi == bi && sbj < bj)))) { best = sbest; bi = sbi; bj = sbj; }
    plo = r_lo; phi = r_hi;
    sbase += (uint64_t)(nr + 63) * 64;
    __threadfence();
  }
  for (int o = 32; o > 0; o >>= 1) {                 // ties: first cell in row-major order (alignment.c:826-830)
    const int ob = __shfl_xor(best, o), oi = __shfl_xor(bi, o), oj = __shfl_xor(bj, o);
    if (ob > best || (ob == best && (oi < bi || (oi == bi && oj < bj)))) { best = ob; bi = oi; bj = oj; }
  }
  *max_i = best > 0 ? bp.s_left + bi : 0;
  *max_j = best > 0 ? bj : 0;
  return best;
}

// alignSmiWatBand (alignment.c:788-1027) by the whole wave for bands up to 64 columns wide.
// Lane c owns the columns jmin + c + 64m; cell (i', j) is computed at step t = i' + (j - jmin), so
// a column advances one row per step and its left neighbour column (lane c-1) is always one step
// ahead on the same row: the diagonal H and the running F arrive by a wave rotate, the column's own
// H/E stay in registers.  A column that enters the band starts from H = E = 0 (:871-872), exactly
// as the reference's row buffers do.  Ties of the maximum resolve to the first cell in row-major
// order (:826-830) by reducing (score, row, column).
// PW: pointer type of the read and window codes -- LDS-typed where they live in LDS, so that their loads (ds_read, not
// flat) never wait for the direction bytes a long read streams to HBM.
template <class PW = const uint8_t *>
__device__ inline int band_track_wave(const Band &bp, PW q, PW win, int match, int mismatch, int gi, int ge,
                                      uint8_t *dir, int *max_i, int *max_j, int tW = 0) {
  const int lane = (int)threadIdx.x;
  const int nrows = bp.s_len - bp.s_left, l = bp.l_edge, r = bp.r_edge, bw = bp.band_width;
  const int jmin = bp.q_left > l ? bp.q_left : l;
  int jlast = r + nrows - 1; if (jlast > bp.q_len - 1) jlast = bp.q_len - 1;
  const int tmax = __builtin_amdgcn_readfirstlane((nrows - 1) + (jlast - jmin));      // wave-uniform: scalar loop bound
  const int tWm = tW - 1;                                                              // tW is a power of two
  int Hcol = 0, Ecol = 0, Hprev = 0, Fout = 0, lastrow = -2, lastcol = -1, qc = 5;
  int best = 0, bi = 0, bj = 0;
  for (int t = 0; t <= tmax; t++) {
    const int nH = wave_ror1(Hcol), nHp = wave_ror1(Hprev), nF = wave_ror1(Fout), nrow = wave_ror1(lastrow), ncol = wave_ror1(lastcol);
    const int x0 = t + l - jmin - 2 * lane;
    const int m = x0 <= 0 ? 0 : (x0 + 127) >> 7;
    const int j = jmin + lane + 64 * m;
    const int ip = t - (j - jmin);
    const bool act = ip >= 0 && ip < nrows && j <= jlast && (j - ip) <= r && (j - ip) >= l;
    if (act) {
      if (j != lastcol) { Hcol = 0; Ecol = 0; Hprev = 0; qc = q[j] & 7; }
      int diag = 0, F = 0;
      if (ncol == j - 1) {
        if (nrow == ip) { diag = nHp; F = nF; }
        else if (nrow == ip - 1) diag = nH;
      }
      const int rb = win[bp.s_left + ip] & 7;
      const int w = (rb >= 4 || qc >= 4) ? 0 : (rb == qc ? match : mismatch);      // score.c:138-173 (codes 0-3, 5 = N)
      const int Hin = diag + w;
      int Hnew;
      bool cand;
      const int hb = Hcol;
      const int d = cell_update(Hnew, Ecol, F, Hin, gi, ge, cand);
      Hprev = hb; Hcol = Hnew; Fout = F; lastrow = ip; lastcol = j;
      dir[tW ? (size_t)t * (size_t)tW + (size_t)((j - jmin) & tWm) : (size_t)ip * (size_t)(bw - 1) + (size_t)(j - l)] = (uint8_t)d;
      if (cand && Hin > best) { best = Hin; bi = ip; bj = j; }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const int ob = __shfl_xor(best, o), oi = __shfl_xor(bi, o), oj = __shfl_xor(bj, o);
    if (ob > best || (ob == best && (oi < bi || (oi == bi && oj < bj)))) { best = ob; bi = oi; bj = oj; }
  }
  *max_i = best > 0 ? bp.s_left + bi : 0;
  *max_j = best > 0 ? bj : 0;
  return best;
}

// Bands of up to 64 diagonals, row by row: lane c owns band diagonal l + c, i.e. cell (i', j = l + c + i') of row i'.
// The diagonal predecessor (i'-1, j-1) lies on the same diagonal (own register), the cell above (i'-1, j) on diagonal
// c + 1 (one DPP shift), and the horizontal gap score reaches a cell through a prefix maximum over the lanes to its left:
// in a row F only ever comes from cells whose diagonal won with H > gi (cell_update raises F to H - gi there and nowhere
// else), decays by ge per column while positive, and is used as max(F, 0) -- so F_in(c) = max(0, max_{i<c} (H_i - gi
// + ge i) - ge (c - 1)), exact as long as gi >= ge (a cell that loses to F cannot raise it: H - gi <= F - ge).  One step
// per row (16 VALU for the scan) instead of two anti-diagonal steps with half the lanes idle: 2.4 x fewer instructions
// per band cell than band_track_wave.  Results (scores, direction bytes, first maximum) are those of band_track_wave.
// exclusive prefix maximum over the wave; INT_MIN is the identity, so the DPP moves fold into the v_max instructions
// LANES: how many lanes can hold a live value (16, 32 or 64): a band of up to 16 diagonals needs no step across DPP rows
template <int LANES = 64>
__device__ inline int dpp_scan_max_excl(int x) {
  const int neg = (int)0x80000000;
  x = max(x, __builtin_amdgcn_update_dpp(neg, x, 0x111 /* row_shr:1 */, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(neg, x, 0x112 /* row_shr:2 */, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(neg, x, 0x114 /* row_shr:4 */, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(neg, x, 0x118 /* row_shr:8 */, 0xf, 0xf, false));
  if (LANES > 16) x = max(x, __builtin_amdgcn_update_dpp(neg, x, 0x142 /* row_bcast:15 */, 0xa, 0xf, false));
  if (LANES > 32) x = max(x, __builtin_amdgcn_update_dpp(neg, x, 0x143 /* row_bcast:31 */, 0xc, 0xf, false));
  return __builtin_amdgcn_update_dpp(neg, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
template <class PW, class PD, int LANES>
__device__ inline int band_track_rows_n(const Band &bp, PW q, PW win, int match, int mismatch, int gi, int ge, PD dir, int *max_i, int *max_j);
template <class PW = const uint8_t *, class PD = uint8_t *>
__device__ inline int band_track_rows(const Band &bp, PW q, PW win, int match, int mismatch, int gi, int ge,
                                      PD dir, int *max_i, int *max_j) {
  const int bw = __builtin_amdgcn_readfirstlane(bp.band_width);        // (wave-uniform: one instance of the row loop runs)
  if (bw <= 16) return band_track_rows_n<PW, PD, 16>(bp, q, win, match, mismatch, gi, ge, dir, max_i, max_j);
  if (bw <= 32) return band_track_rows_n<PW, PD, 32>(bp, q, win, match, mismatch, gi, ge, dir, max_i, max_j);
  return band_track_rows_n<PW, PD, 64>(bp, q, win, match, mismatch, gi, ge, dir, max_i, max_j);
}
template <class PW, class PD, int LANES>
__device__ inline int band_track_rows_n(const Band &bp, PW q, PW win, int match, int mismatch, int gi, int ge,
                                      PD dir, int *max_i, int *max_j) {
  const int lane = (int)threadIdx.x;
  const int nrows = __builtin_amdgcn_readfirstlane(bp.s_len - bp.s_left), l = bp.l_edge, bw = bp.band_width;
  const int d = l + lane, qhi = bp.q_len - 1;
  const bool inband = lane < bw;
  const int NEG = -(1 << 28);
  const int gel = ge * lane - gi, gel1 = ge * (lane - 1);
  int Hd = 0, Eown = 0;
  int best = 0, bi = 0, bj = 0;
  // the read code of the next row is loaded a row ahead (the read offset clamped into the read: inactive cells do not use it); the
  // reference codes of 64 rows sit in one register, lane r holding row (ip & ~63) + r, and reach the row's cells by v_readlane
  int wrow = 0;
  int qn = q[d < 0 ? 0 : (d > qhi ? qhi : d)] & 7;
  PD dp = dir + lane;
  for (int ip = 0; ip < nrows; ip++, dp += bw) {
    const int j = d + ip;
    const bool act = inband && j >= bp.q_left && j <= qhi;
    if (!(ip & 63)) { const int r = ip + lane; wrow = win[bp.s_left + (r < nrows ? r : nrows - 1)] & 7; }
    const int rb = __builtin_amdgcn_readlane(wrow, ip & 63), qc = qn;
    {
      const int jn = j + 1;
      qn = q[jn < 0 ? 0 : (jn > qhi ? qhi : jn)] & 7;
    }
    const int Ein = __builtin_amdgcn_update_dpp(0, Eown, 0x130 /* wave_shl:1: the lane above */, 0xf, 0xf, false);
    const int w = (rb >= 4 || qc >= 4) ? 0 : (rb == qc ? match : mismatch);      // score.c:138-173 (codes 0-3, 5 = N)
    const int Hin = Hd + w;
    const int e = Ein > 0 ? Ein : 0;
    const int src = (act && Hin > e && Hin > gi) ? Hin + gel : (int)0x80000000;
    int px = dpp_scan_max_excl<LANES>(src);
    px = px > NEG ? px : NEG;
    int F = px - gel1;
    F = F > 0 ? F : 0;
    int Hnew, E = Ein;
    bool cand;
    const int dc = cell_update(Hnew, E, F, Hin, gi, ge, cand);
    if (act) *dp = (uint8_t)dc;
    const bool better = act && cand && Hin > best;
    best = better ? Hin : best; bi = better ? ip : bi; bj = better ? j : bj;
    Hd = act ? Hnew : 0;
    Eown = act ? E : 0;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const int ob = __shfl_xor(best, o), oi = __shfl_xor(bi, o), oj = __shfl_xor(bj, o);
    if (ob > best || (ob == best && (oi < bi || (oi == bi && oj < bj)))) { best = ob; bi = oi; bj = oj; }
  }
  *max_i = best > 0 ? bp.s_left + bi : 0;
  *max_j = best > 0 ? bj : 0;
  return best;
}

// The traceback of traceback_scalar (row-major directions) run by the whole wave: the walk's state is the same in all lanes and
// lives on the scalar unit (readfirstlane); the alignment string is written by lane 0.  A stretch of diagonal steps is taken in
// ONE round: lane k looks at the cell k steps up the diagonal (its direction byte, reference and read base), a ballot finds where
// the stretch ends and where it holds non-matches, and the string operations of the whole stretch are put together from those two
// masks -- an alignment of a 150-base read with three mismatches takes about eight rounds instead of 150 dependent steps.
__device__ inline int traceback_uniform(uint8_t *ds, uint32_t dscap, int *qs, int *rs, const Band &bp, const uint8_t *dir,
                                        int max_i, int max_j, int max_scor, const uint8_t *q, const uint8_t *win,
                                        int tb_match, int tb_mismatch, int gi, int ge) {
#define SMG_U(v) __builtin_amdgcn_readfirstlane((int)(v))
  const int bw = SMG_U(bp.band_width), s_left = SMG_U(bp.s_left), q_left = SMG_U(bp.q_left);
  int i = SMG_U(max_i), j = SMG_U(max_j);
  int off = (i - s_left) * (bw - 1) + (j - SMG_U(bp.l_edge));       // dir_index with tW == 0
  int n = 0, checksum = 0, nmatch = 0, rv = 0;
  bool gap_open = false;
  const bool writer = threadIdx.x == 0;
  const int cap = (int)dscap, lane = (int)threadIdx.x;
  // (on overflow the walk goes on without writing: the caller only sees the code)
#define SMG_PUTU(cnt, typ) { if (n + 2 >= cap) rv = -2; else { if (writer) ds[n] = (uint8_t)((cnt) + ((typ) << DIFF_TYPSHIFT)); n++; } }
  while (i >= s_left && j >= q_left && !rv) {
    const int room = (i - s_left < j - q_left ? i - s_left : j - q_left) + 1;       // cells up this diagonal inside the matrix
    const bool in = lane < room;
    uint8_t dv = 0, wv = 0, qv = 0;
    if (in) { dv = dir[off - lane * bw]; wv = win[i - lane]; qv = q[j - lane]; }
    const unsigned long long off_dia = __ballot(!in || dv != (uint8_t)DIR_DIA);
    const int run = off_dia ? (int)__builtin_ctzll(off_dia) : 64;
    if (run == 0) {                                    // the cell itself: a gap step, or the end of the path
      const int d = SMG_U(dv);
      if (!d) break;
      if (gap_open) checksum -= ge; else { checksum -= gi; gap_open = true; }
      if (d & (int)DIR_COL) { SMG_PUTU(nmatch, DIFF_D) nmatch = 0; off -= bw - 1; i--; continue; }
      if (!(d & (int)DIR_ROW)) { rv = -1; break; }
      SMG_PUTU(nmatch, DIFF_I) nmatch = 0; off--; j--;
      continue;
    }
    const int rbc = wv & 7, qcc = qv & 7;
    const int sc = (rbc >= 4 || qcc >= 4) ? 0 : (rbc == qcc ? tb_match : tb_mismatch);
    const unsigned long long inrun = run == 64 ? ~0ull : ((1ull << run) - 1ull);
    unsigned long long stops = __ballot(sc <= 0) & inrun;            // steps that close a run of matches (substitution or a code without score)
    const unsigned long long minus = __ballot(sc < 0) & inrun;
    checksum += (run - (int)__builtin_popcountll(stops)) * tb_match + (int)__builtin_popcountll(minus) * tb_mismatch;
    int pos = 0;
    for (;;) {
      const int nxt = stops ? (int)__builtin_ctzll(stops) : run;
      int m = nxt - pos;                               // matching steps up to the next stop: the counter saturates as in the step-by-step walk
      while (m > 0) {
        const int space = (int)DIFF_MAXMISMATCH + 1 - nmatch;
        if (m <= space) { nmatch += m; m = 0; }
        else { m -= space + 1; SMG_PUTU(DIFF_MAXMISMATCH, DIFF_M) nmatch = 1; }
      }
      if (nxt == run) break;
      SMG_PUTU(nmatch, DIFF_S) nmatch = 0;
      stops &= stops - 1ull;
      pos = nxt + 1;
    }
    gap_open = false;
    off -= run * bw; i -= run; j -= run;
  }
  if (!rv) { SMG_PUTU(nmatch, DIFF_S) }
  if (!rv) { SMG_PUTU(0, DIFF_M) }
#undef SMG_PUTU
#undef SMG_U
  if (rv) return rv;
  *rs = i + 1; *qs = j + 1;
  return (checksum != max_scor) ? -3 : n;
}

// The same for wider bands.  Columns jmin + c + 64m of lane c that are inside the band at step t differ by 128
// diagonals, so a lane has at most (r - l) / 128 + 1 live columns; column m keeps its state in register slot m % NS.
// The left neighbour of (lane c, slot s) is (lane c - 1, slot s), for lane 0 (lane 63, slot s - 1).
template <int NS, class PW = const uint8_t *>
__device__ inline int band_track_wave_n(const Band &bp, PW q, PW win, int match, int mismatch, int gi, int ge,
                                        uint8_t *dir, int *max_i, int *max_j, int tW = 0) {
  const int lane = (int)threadIdx.x;
  const int nrows = bp.s_len - bp.s_left, l = bp.l_edge, r = bp.r_edge, bw = bp.band_width;
  const int jmin = bp.q_left > l ? bp.q_left : l;
  int jlast = r + nrows - 1; if (jlast > bp.q_len - 1) jlast = bp.q_len - 1;
  const int tmax = __builtin_amdgcn_readfirstlane((nrows - 1) + (jlast - jmin));      // wave-uniform: scalar loop bound
  const int tWm = tW - 1;                                                              // tW is a power of two
  int Hcol[NS], Ecol[NS], Hprev[NS], Fout[NS], lastrow[NS], lastcol[NS], qc[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) { Hcol[s] = Ecol[s] = Hprev[s] = Fout[s] = 0; lastrow[s] = -2; lastcol[s] = -1; qc[s] = 5; }
  int best = 0, bi = 0, bj = 0;
  for (int t = 0; t <= tmax; t++) {
    int nH[NS], nHp[NS], nF[NS], nrow[NS], ncol[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) { nH[s] = wave_ror1(Hcol[s]); nHp[s] = wave_ror1(Hprev[s]); nF[s] = wave_ror1(Fout[s]); nrow[s] = wave_ror1(lastrow[s]); ncol[s] = wave_ror1(lastcol[s]); }
    const int x0 = t + l - jmin - 2 * lane, x1 = t + r - jmin - 2 * lane;
    const int mlo = x0 <= 0 ? 0 : (x0 + 127) >> 7;
    const int mhi = x1 < 0 ? -1 : x1 >> 7;
    const int mlo_s = mlo % NS;
    int rbv[NS];                              // window codes of all live columns first: the loads overlap
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int m = mlo + (s - mlo_s + NS) % NS;
      const int ip = t - lane - 64 * m;
      rbv[s] = (m <= mhi && ip >= 0 && ip < nrows) ? win[bp.s_left + ip] & 7 : 5;
    }
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int m = mlo + (s - mlo_s + NS) % NS;
      const int j = jmin + lane + 64 * m;
      const int ip = t - (j - jmin);
      const bool act = m <= mhi && ip >= 0 && ip < nrows && j <= jlast && (j - ip) <= r && (j - ip) >= l;
      if (act) {
        if (j != lastcol[s]) { Hcol[s] = 0; Ecol[s] = 0; Hprev[s] = 0; qc[s] = q[j] & 7; }
        const int sp = (s + NS - 1) % NS;
        const int vH = lane ? nH[s] : nH[sp], vHp = lane ? nHp[s] : nHp[sp], vF = lane ? nF[s] : nF[sp];
        const int vrow = lane ? nrow[s] : nrow[sp], vcol = lane ? ncol[s] : ncol[sp];
        int diag = 0, F = 0;
        if (vcol == j - 1) {
          if (vrow == ip) { diag = vHp; F = vF; }
          else if (vrow == ip - 1) diag = vH;
        }
        const int rb = rbv[s];
        const int w = (rb >= 4 || qc[s] >= 4) ? 0 : (rb == qc[s] ? match : mismatch);
        const int Hin = diag + w;
        int Hnew;
        bool cand;
        const int hb = Hcol[s];
        const int d = cell_update(Hnew, Ecol[s], F, Hin, gi, ge, cand);
        Hprev[s] = hb; Hcol[s] = Hnew; Fout[s] = F; lastrow[s] = ip; lastcol[s] = j;
        dir[tW ? (size_t)t * (size_t)tW + (size_t)((j - jmin) & tWm) : (size_t)ip * (size_t)(bw - 1) + (size_t)(j - l)] = (uint8_t)d;
        if (cand && (Hin > best || (Hin == best && (ip < bi || (ip == bi && j < bj))))) { best = Hin; bi = ip; bj = j; }
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const int ob = __shfl_xor(best, o), oi = __shfl_xor(bi, o), oj = __shfl_xor(bj, o);
    if (ob > best || (ob == best && (oi < bi || (oi == bi && oj < bj)))) { best = ob; bi = oi; bj = oj; }
  }
  *max_i = best > 0 ? bp.s_left + bi : 0;
  *max_j = best > 0 ? bj : 0;
  return best;
}
#endif

// WIDE: also instantiate the multi-column wave form for bands of more than 64 columns (long reads); it needs many
// registers, so mappers for short reads use the lean instance and leave the rare wide band to the sequential form.
template <bool WIDE = false>
SMG_HD inline void stage_align(const Batch &b, const DevIndex &ix, const MapPar &p, uint32_t r, AlignScratch &x) {
  const uint32_t qlen = read_len(b, r);
  const CandHdr ch = b.ch[r];
  const ReadCtl ctl = b.ctl[r];
  ReadStat &st = b.stat[r];
  const RCand *rc = b.rcpool + ch.rc_off;
  enum { S_MINSW = 0, S_SWMAX = 1, S_SW2ND = 2, S_NRES = 3, S_NDSTR = 4, S_ERR = 5, S_SP = 6, S_NALI = 7, S_SITE = 8 };
  int8_t M[64];
  score_matrix(M, p.match, p.mismatch);
  const int gi = -p.gap_init, ge = -p.gap_ext;

  unsigned long long aph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (x.pass == 2 && st.err != SMG_ERR_RETRY) return;       // wave-uniform: the first pass finished this read (the kernel walks the retry list)
  SMG_LANE0 {
    // a call that appends to a ResultSet continues that set's running score maxima (rmap.c:881-885 reads them)
    x.state[S_MINSW] = ctl.min_swatscor; x.state[S_SWMAX] = b.prevmax ? b.prevmax[2 * r] : 0; x.state[S_SW2ND] = b.prevmax ? b.prevmax[2 * r + 1] : 0;
    x.state[S_NRES] = 0; x.state[S_NDSTR] = 0; x.state[S_ERR] = ch.err; x.state[S_SP] = 0; x.state[S_NALI] = 0; x.state[S_SITE] = ch.err ? ch.err_site : 0;
  }
  // both orientations of the read next to the DP rows (codes are read once per column)
  SMG_PAR_CHUNKS(base, qlen) {
    uint32_t i = base + SMG_LANE;
    if (i < qlen) { x.qcodes[i] = b.codes[b.read_off[r] + i]; x.qcodes[x.qstride + i] = b.codes_rc[b.read_off[r] + i]; }
  }
  SMG_SYNC();
  const uint32_t ncand = (ctl.go && qlen >= (uint32_t)(b.fine_idx ? (int)FINE_K : ix.k)) ? (uint32_t)ctl.n_scored : 0u;
  // Candidates are visited in rank order, but only those whose first-pass score reaches the threshold are aligned
  // (rmap.c:826-828), typically one to three of a few hundred, and the threshold only rises while the read is
  // processed (rmap.c:881-885).  64 scores are tested at once against the initial threshold; the survivors are
  // taken in order and tested again against the current one.
  for (uint32_t cbase = 0; cbase < ncand && !x.state[S_ERR]; cbase += SMG_NLANES) {
   const uint32_t cmy = cbase + SMG_LANE;
   const bool cpass = cmy < ncand && rc[cmy].swscor >= ctl.min_swatscor;
#if defined(__HIP_DEVICE_COMPILE__)
   unsigned long long cmask = __ballot(cpass);
#else
   unsigned long long cmask = cpass ? 1ull : 0ull;
#endif
   while (cmask) {
    const uint32_t ci = cbase + (uint32_t)__builtin_ctzll(cmask);
    cmask &= cmask - 1ull;
    const RCand c = rc[ci];
    if (x.state[S_ERR]) break;
    if (c.swscor < x.state[S_MINSW]) continue;            // rmap.c:826-828 (all ranked candidates are scored)
    const uint32_t wlen = (uint32_t)(c.re - c.rs + 1);
    if (wlen > x.wincap || c.qs > c.qe || c.qe >= qlen || (c.flags & RCF_ERR)) { SMG_SYNC(); SMG_LANE0 { x.state[S_ERR] = (wlen > x.wincap) ? SMG_ERR_CAP : SMG_ERR_ASSERT; x.state[S_SITE] = __LINE__; } SMG_SYNC(); break; }
    const uint64_t gbase = (c.sqidx < 0 ? 0ull : ix.sop[c.sqidx]) + c.rs;
    uint8_t *const win = wlen <= x.win_lds_cap ? x.win_lds : x.win;
    uint8_t *const dtmp = wlen <= x.win_lds_cap ? x.dtmp_lds : x.dtmp;
    const uint32_t dtmpcap = wlen <= x.win_lds_cap ? qlen + x.win_lds_cap + 16 : x.dtmpcap;
    unsigned long long tq0 = phase_clock(), tq1;
    SMG_PAR_CHUNKS(base, wlen) {                           // fetch + decode the reference window (rmap.c:831-845)
      uint32_t i = base + SMG_LANE;
      if (i < wlen) win[i] = (uint8_t)ref_code(ix.packed, gbase + i);
    }
    SMG_SYNC();
    tq1 = phase_clock(); aph[0] += tq1 - tq0; tq0 = tq1; aph[4]++;
    // scalars of this candidate: every lane computes the same values from shared state
    const uint8_t *q = x.qcodes + ((c.flags & RCF_REVERSE) ? x.qstride : 0);
    int min_swatscor = x.state[S_MINSW];
    if ((p.flags & FLG_BEST) && x.state[S_SW2ND] > min_swatscor) min_swatscor = x.state[S_SW2ND];   // rmap.c:881-885
    int bwc = c.band_r - c.band_l, band_l, band_r;
    if (bwc < ctl.bandwidth_min) { bwc = (ctl.bandwidth_min - bwc + 1) / 2; band_l = c.band_l - bwc; band_r = c.band_r + bwc; }
    else { band_l = c.band_l; band_r = c.band_r; }
    // aliSmiWatInBand (alignment.c:1548-1601)
    const int minscore = min_swatscor;
    int minscorlen = ctl.scorlen_min, err0 = 0;
    if (minscore < 1 || p.match <= 0) err0 = SMG_ERR_ASSERT;
    if (minscorlen * p.match < minscore) minscorlen = minscore / p.match;
    if (minscorlen < ALILEN_MIN) err0 = SMG_ERR_ASSERT;
    const int res_first = x.state[S_NRES];
    SMG_SYNC();
    SMG_LANE0 {
      x.state[S_MINSW] = min_swatscor;
      x.state[S_NALI] = 0;
      if (err0) { x.state[S_ERR] = err0; x.state[S_SITE] = __LINE__; x.state[S_SP] = 0; }
      else { x.ivstack[0] = 0; x.ivstack[1] = (int)wlen - 1; x.state[S_SP] = 1; }
    }
    SMG_SYNC();
    // alignSmiWatBandRecursive (alignment.c:1300-1434) with an explicit interval stack: node, then the
    // reference range left of its alignment, then the range to the right
    for (;;) {
      const int sp = x.state[S_SP];
      if (sp <= 0 || x.state[S_ERR]) break;
      const int s_left = x.ivstack[2 * (sp - 1)], s_right = x.ivstack[2 * (sp - 1) + 1];
      Band band;
      int nerr = 0, nsite = 0;
      bool skip = false;
      if (minscorlen < 2) { nerr = SMG_ERR_ASSERT; nsite = __LINE__; }
      else if (band_init(band, band_l, band_r, (int)c.qs, (int)c.qe, (int)qlen, s_left, s_right, (int)wlen)) skip = true;
      else if (band.s_left >= band.s_len || band.band_width < 0) { nerr = SMG_ERR_ASSERT; nsite = __LINE__; }
      uint64_t dneed = 0;
      uint8_t *dirm = x.dir;
      if (!nerr && !skip) {
        dneed = (uint64_t)band.band_width * (uint64_t)(band.s_len - band.s_left) + (uint64_t)band.band_width + 8;
        if (dneed <= x.dir_lds_cap) dirm = x.dir_lds;
      }
      int max_i = 0, max_j = 0, max_scor = 0, tW = 0;
#if defined(__HIP_DEVICE_COMPILE__)
      // bands of up to 64 diagonals take the row form (band_track_rows; its horizontal gap scan needs gi >= ge), row-major directions
      const bool rows_form = gi >= ge && ge >= 0 && band.band_width <= 64 && x.rows_form;
      if (!nerr && !skip && dirm == x.dir && band.band_width >= 1 && !rows_form && (band.band_width <= 64 || (WIDE && (band.r_edge - band.l_edge) / 128 + 1 <= 12))) {
        // direction matrix in HBM and a wave form: anti-diagonal-major layout (dir_index) if it fits
        const int jmin = band.q_left > band.l_edge ? band.q_left : band.l_edge;
        int jlast = band.r_edge + (band.s_len - band.s_left) - 1; if (jlast > band.q_len - 1) jlast = band.q_len - 1;
        int w = 64;                              // power of two >= band_width / 2 + 2 (the live columns of one step)
        while (w < band.band_width / 2 + 2) w <<= 1;
        const int64_t tmax = (int64_t)(band.s_len - band.s_left - 1) + (jlast - jmin);
        if (tmax >= 0 && (uint64_t)(tmax + 1) * (uint64_t)w + 8 <= x.dircap) tW = w;
      }
      StripGeom sg;
      if (WIDE && !nerr && !skip && dirm == x.dir && band.band_width >= 256 && x.bnd && x.ring && (uint32_t)(band.s_len - band.s_left) <= x.bndcap) {
        sg = strip_geom(band, band.band_width < 2048 ? 3 : 4);      // wide band: strip form, 2-bit directions; narrower strips fit the band better
        if (sg.nstrip > 0 && strip_words(sg, sg.nstrip) * 4 + 16 <= x.dircap) tW = -1;
      }
#endif
      if (!nerr && !skip && !tW && dirm == x.dir && dneed > x.dircap) { nerr = x.pass == 1 ? SMG_ERR_RETRY : SMG_ERR_CAP; nsite = __LINE__; }
      tq0 = phase_clock();
      if (!nerr && !skip) { aph[3]++; aph[7] += (unsigned long long)band.band_width; aph[5] += (unsigned long long)(band.s_len - band.s_left) + (unsigned long long)(band.q_len - band.q_left); }
      if (!nerr && !skip) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int nslot = band.band_width >= 1 ? (band.r_edge - band.l_edge) / 128 + 1 : 0;    // live columns per lane
        typedef SMG_LDSQ const uint8_t *PL;
        const bool in_lds = WIDE && x.win_lds && win == x.win_lds;      // read and window both in the LDS block
        if (WIDE && tW < 0) {
          if (sg.cbits == 3) {
            if (in_lds) max_scor = band_track_strip<PL, 8>(band, sg, (PL)q, (PL)win, p.match, p.mismatch, gi, ge, (uint32_t *)dirm, (int2 *)x.bnd, x.bndcap, (int2 *)x.ring, &max_i, &max_j);
            else max_scor = band_track_strip<const uint8_t *, 8>(band, sg, q, win, p.match, p.mismatch, gi, ge, (uint32_t *)dirm, (int2 *)x.bnd, x.bndcap, (int2 *)x.ring, &max_i, &max_j);
          } else {
            if (in_lds) max_scor = band_track_strip<PL, 16>(band, sg, (PL)q, (PL)win, p.match, p.mismatch, gi, ge, (uint32_t *)dirm, (int2 *)x.bnd, x.bndcap, (int2 *)x.ring, &max_i, &max_j);
            else max_scor = band_track_strip<const uint8_t *, 16>(band, sg, q, win, p.match, p.mismatch, gi, ge, (uint32_t *)dirm, (int2 *)x.bnd, x.bndcap, (int2 *)x.ring, &max_i, &max_j);
          }
        } else if (band.band_width >= 1 && band.band_width <= 64 && rows_form) {
          // read, window and (usually) the direction bytes live in the LDS block: LDS-typed pointers (ds_read / ds_write with
          // 32-bit addresses instead of flat accesses with 64-bit address arithmetic)
          typedef SMG_LDSQ uint8_t *PLD;
          const bool codes_lds = x.win_lds && win == x.win_lds;
          if (codes_lds && dirm == x.dir_lds) max_scor = band_track_rows<PL, PLD>(band, (PL)q, (PL)win, p.match, p.mismatch, gi, ge, (PLD)dirm, &max_i, &max_j);
          else if (codes_lds) max_scor = band_track_rows<PL, uint8_t *>(band, (PL)q, (PL)win, p.match, p.mismatch, gi, ge, dirm, &max_i, &max_j);
          else max_scor = band_track_rows<const uint8_t *, uint8_t *>(band, q, win, p.match, p.mismatch, gi, ge, dirm, &max_i, &max_j);
        } else if (band.band_width >= 1 && band.band_width <= 64) {
          if (in_lds) max_scor = band_track_wave<PL>(band, (PL)q, (PL)win, p.match, p.mismatch, gi, ge, dirm, &max_i, &max_j, tW);
          else max_scor = band_track_wave<const uint8_t *>(band, q, win, p.match, p.mismatch, gi, ge, dirm, &max_i, &max_j, tW);
        } else if (WIDE && nslot >= 1 && nslot <= 2) {
          if (in_lds) max_scor = band_track_wave_n<2, PL>(band, (PL)q, (PL)win, p.match, p.mismatch, gi, ge, dirm, &max_i, &max_j, tW);
          else max_scor = band_track_wave_n<2, const uint8_t *>(band, q, win, p.match, p.mismatch, gi, ge, dirm, &max_i, &max_j, tW);
        } else if (WIDE && nslot >= 1 && nslot <= 4) {
          if (in_lds) max_scor = band_track_wave_n<4, PL>(band, (PL)q, (PL)win, p.match, p.mismatch, gi, ge, dirm, &max_i, &max_j, tW);
          else max_scor = band_track_wave_n<4, const uint8_t *>(band, q, win, p.match, p.mismatch, gi, ge, dirm, &max_i, &max_j, tW);
        } else if (WIDE && nslot >= 1 && nslot <= 7) {
          if (in_lds) max_scor = band_track_wave_n<7, PL>(band, (PL)q, (PL)win, p.match, p.mismatch, gi, ge, dirm, &max_i, &max_j, tW);
          else max_scor = band_track_wave_n<7, const uint8_t *>(band, q, win, p.match, p.mismatch, gi, ge, dirm, &max_i, &max_j, tW);
        } else if (WIDE && nslot >= 1 && nslot <= 12) {
          if (in_lds) max_scor = band_track_wave_n<12, PL>(band, (PL)q, (PL)win, p.match, p.mismatch, gi, ge, dirm, &max_i, &max_j, tW);
          else max_scor = band_track_wave_n<12, const uint8_t *>(band, q, win, p.match, p.mismatch, gi, ge, dirm, &max_i, &max_j, tW);
        } else {
          aph[6]++;
          SMG_LANE0 { x.state[8] = band_track_scalar(band, q, win, M, gi, ge, x.Hp, x.Ep, dirm, &max_i, &max_j); x.state[9] = max_i; x.state[10] = max_j; }
          SMG_SYNC();
          max_scor = x.state[8]; max_i = x.state[9]; max_j = x.state[10];
        }
#else
        max_scor = band_track_scalar(band, q, win, M, gi, ge, x.Hp, x.Ep, dirm, &max_i, &max_j);
#endif
      }
      SMG_SYNC();
      tq1 = phase_clock(); aph[1] += tq1 - tq0; tq0 = tq1;
      int tb_dn = 0, tb_qs = 0, tb_rs = 0;
      bool tb_done = false;
#if defined(__HIP_DEVICE_COMPILE__)
      if (!nerr && !skip && max_scor >= minscore && tW == 0 && x.rows_form) {      // row-major directions: traceback by the wave, on the scalar unit
        tb_dn = traceback_uniform(dtmp, dtmpcap, &tb_qs, &tb_rs, band, dirm, max_i, max_j, max_scor, q, win, (int)M[0], (int)M[1], gi, ge);
        tb_done = true;
      }
#else
      (void)tb_dn; (void)tb_done;
#endif
      SMG_LANE0 {
        int nsp = sp - 1, err = nerr;
        if (nerr) x.state[S_SITE] = nsite;
        if (!err && !skip && max_scor >= minscore) {
          int qs = tb_qs, rs = tb_rs;
          #if defined(__HIP_DEVICE_COMPILE__)
          const int dn = tb_done ? tb_dn : traceback_scalar(dtmp, dtmpcap, &qs, &rs, band, dirm, max_i, max_j, max_scor, q, win, M, gi, ge, tW, &sg);
#else
          const int dn = traceback_scalar(dtmp, dtmpcap, &qs, &rs, band, dirm, max_i, max_j, max_scor, q, win, M, gi, ge, tW);
#endif
          if (dn < 0) { err = (dn == -2) ? SMG_ERR_CAP : (dn == -3 ? SMG_ERR_SCORE : SMG_ERR_ASSERT); x.state[S_SITE] = __LINE__; }
          const int qe = max_j, re = max_i;
          if (!err && !(qs + minscorlen > qe + 1)) {
            // addALIMETAtoRsltSet (alignment.c:1277): forward DiffStr appended to the scratch string pool
            const int nali = x.state[S_NALI];
            if ((uint32_t)(res_first + nali) >= x.rescap || (uint32_t)(x.state[S_NDSTR] + dn + 2) > x.dstrcap) { err = x.pass == 1 ? SMG_ERR_RETRY : SMG_ERR_CAP; x.state[S_SITE] = __LINE__; }   // the second pass has large result slots
            else {
              Result &a = x.res[res_first + nali];
              a.swatscor = max_scor; a.q_start = (uint32_t)qs; a.q_end = (uint32_t)qe; a.s_start = (uint64_t)rs; a.s_end = (uint64_t)re;
              a.stroffs = (uint32_t)x.state[S_NDSTR];
              const int fl = diffstr_reverse(x.dstr + a.stroffs, dtmp, dn);
              if (fl < 0) { err = SMG_ERR_ASSERT; x.state[S_SITE] = __LINE__; }
              else {
                a.strlen = (uint32_t)fl;
                x.state[S_NDSTR] += fl;
                x.state[S_NALI] = nali + 1;
                // right interval is pushed first so that the left one is aligned first (alignment.c:1389-1431)
                if (s_right > re + minscorlen) { if (nsp >= 62) { err = SMG_ERR_CAP; x.state[S_SITE] = __LINE__; } else { x.ivstack[2 * nsp] = re + 1; x.ivstack[2 * nsp + 1] = s_right; nsp++; } }
                if (!err && s_left + minscorlen < rs) { if (nsp >= 62) { err = SMG_ERR_CAP; x.state[S_SITE] = __LINE__; } else { x.ivstack[2 * nsp] = s_left; x.ivstack[2 * nsp + 1] = rs - 1; nsp++; } }
              }
            }
          }
        }
        if (err) x.state[S_ERR] = err;
        x.state[S_SP] = nsp;
      }
      SMG_SYNC();
      tq1 = phase_clock(); aph[2] += tq1 - tq0; tq0 = tq1;
    }
    SMG_LANE0 {
      const int nali = x.state[S_NALI];
      if (!x.state[S_ERR] && nali > 0) {
        // resultSetAddFromAli (results.c:1852-1942) incl. its duplicate handling: a result equal to
        // its predecessor is popped; what is written after a popped slot within one call is lost
        // from the array but still raises the score maxima.
        const bool is_rev = (c.flags & RCF_REVERSE) != 0;
        uint32_t arrlen = (uint32_t)res_first;
        uint32_t dkeep = x.res[res_first].stroffs;       // compacted string pool write position
        Result *rp = &x.res[arrlen++];
        bool is_new = false;
        Result cur;
        for (int i = 0; i < nali; i++) {
          cur = x.res[res_first + i];                    // raw alignment i (slots are consumed in order, never ahead of i)
          if (is_new) { rp = &x.res[arrlen++]; is_new = false; }
          Result nr;
          nr.swatscor = cur.swatscor;
          if (is_rev) { nr.q_start = qlen - cur.q_end; nr.q_end = qlen - cur.q_start; }
          else { nr.q_start = cur.q_start + 1; nr.q_end = cur.q_end + 1; }
          nr.s_start = (uint64_t)((uint32_t)c.rs + (uint32_t)cur.s_start + 1u);   // soffs is passed as SEQLEN_t
          nr.s_end = (uint64_t)((uint32_t)c.rs + (uint32_t)cur.s_end + 1u);
          nr.sidx = c.sqidx; nr.reverse = is_rev ? 1u : 0u; nr.pad = (i == 0) ? 1u : 0u;
          nr.stroffs = cur.stroffs; nr.strlen = cur.strlen;
          const Result *pp = rp - 1;
          is_new = (arrlen < 2) || !(nr.s_start == pp->s_start && nr.s_end == pp->s_end && nr.q_start == pp->q_start &&
                                     nr.q_end == pp->q_end && nr.swatscor == pp->swatscor && nr.sidx == pp->sidx);
          // (a call that appends to a ResultSet returns everything: what repeats the set's last alignment, and what is lost behind
          //  a repeat, is decided where the set is -- smgpost::Table::take_call; a repeat carries the score of an alignment that is
          //  in the maxima already, so they come out the same)
          if (b.raw_results) is_new = true;
          if (is_new) {
            for (uint32_t t = 0; t < nr.strlen; t++) x.dstr[dkeep + t] = x.dstr[nr.stroffs + t];
            nr.stroffs = dkeep; dkeep += nr.strlen;
            if (nr.swatscor > x.state[S_SW2ND]) {        // UPDATE_SWATSCORMAX (results.c:1013)
              if (nr.swatscor > x.state[S_SWMAX]) { x.state[S_SW2ND] = x.state[S_SWMAX]; x.state[S_SWMAX] = nr.swatscor; }
              else if (nr.swatscor < x.state[S_SWMAX]) x.state[S_SW2ND] = nr.swatscor;
            }
            *rp = nr;
          } else {
            *rp = nr;
            arrlen--;
          }
        }
        x.state[S_NRES] = (int32_t)arrlen;
        x.state[S_NDSTR] = (int32_t)dkeep;
      }
    }
    SMG_SYNC();
  }
  }
  // ---- publish: exact-size slices of the result and string pools ----
  SMG_LANE0 {
    const uint32_t nres = (uint32_t)x.state[S_NRES], nd = (uint32_t)x.state[S_NDSTR];
    st.swmax = x.state[S_SWMAX]; st.sw2nd = x.state[S_SW2ND];
    st.nseg = (int32_t)ch.n_sort; st.nseg_tot = (int32_t)ch.n_mincover;
    st.nhit = b.hi[2 * r].nhit_rank + b.hi[2 * r + 1].nhit_rank;
    st.nhit_tot = b.hi[2 * r].nhit_tot + b.hi[2 * r + 1].nhit_tot;
    st.err = x.state[S_ERR];
    st.max1 = ctl.max1; st.err_site = st.err ? x.state[S_SITE] : 0;
    st.nres = st.err ? 0 : nres;
    st.res_off = atomic_add_u64(b.res_count, st.nres);
    st.dstr_off = atomic_add_u64(b.dstr_count, st.err ? 0 : nd);
    if (st.res_off + st.nres > b.rescap || st.dstr_off + nd > b.dstrcap) { st.err = SMG_ERR_CAP; st.err_site = __LINE__; st.nres = 0; }
    if (st.err && st.err != SMG_ERR_RETRY) atomic_add_u32((uint32_t *)b.err_flag, 1u);
    if (st.err == SMG_ERR_RETRY && b.align_retry) b.align_retry[atomic_add_u32(b.align_retry_n, 1u)] = r;
    x.state[S_NRES] = (int32_t)st.nres; x.state[S_NDSTR] = st.err ? 0 : (int32_t)nd;
  }
  SMG_SYNC();
  {
    const uint32_t nres = (uint32_t)x.state[S_NRES], nd = (uint32_t)x.state[S_NDSTR];
    const uint64_t ro = st.res_off, dof = st.dstr_off;
    SMG_PAR_CHUNKS(base, nres) { uint32_t i = base + SMG_LANE; if (i < nres) b.respool[ro + i] = x.res[i]; }
    SMG_PAR_CHUNKS(base, nd) { uint32_t i = base + SMG_LANE; if (i < nd) b.dstrpool[dof + i] = x.dstr[i]; }
  }
  for (int i = 0; i < 8; i++) x.tally[i] += aph[i];        // (eight atomics per read on eight fixed addresses before: the kernel flushes once)
}
SMG_HD inline void align_tally_flush(const Batch &b, AlignScratch &x) {
  SMG_LANE0 { if (x.tally[4]) for (int i = 0; i < 8; i++) (void)atomic_add_u64(b.work + WK_ALIGN0 + i, x.tally[i]); }
  for (int i = 0; i < 8; i++) x.tally[i] = 0;
}

}  // namespace smg
