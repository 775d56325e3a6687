// smg_common.h -- data layout shared by the host side and the HIP kernels of libsmaltgpu.
// Names follow the reference's domain (seeds, hits, segments, candidates); `file:line`
// citations refer to the reference tree (SMALT 0.7.6, src/).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SMG_HD __host__ __device__
#else
#define SMG_HD
#endif

namespace smg {

enum : int { IDX_PERFECT = 0, IDX_HASH32MIX = 1 };
enum : uint32_t { FLG_BEST = 0x02, FLG_SEQBYSEQ = 0x10, FLG_NOSHRTINFO = 0x20, FLG_SENSITIVE = 0x80 };
// hit qualifiers per read offset (hashhit.h:57-65)
enum : uint8_t { HQ_TERM = 0, HQ_NORMHIT = 1, HQ_MULTIHIT = 2, HQ_REPEAT = 3, HQ_NOHIT = 4, HQ_NONSTDNT = 5 };
// HashHitInfo status bits (hashhit.c:84-92)
enum : uint32_t { HI_REVERSE = 1, HI_SORTED = 2, HI_RANK = 4 };
enum : uint8_t { CANDFLG_REVERSE = 1, CANDFLG_MMALI = 4 };                 // segment.h:50-58
enum : int { SMG_ERR_CAP = -5, SMG_ERR_ASSERT = -6,   // same values as SMALTGPU_ECAP / EINTERNAL
             SMG_ERR_RETRY = -7,     // internal: the read waits for the second K3 pass (large direction-matrix slots)
             SMG_ERR_SCORE = -8 };   // = SMALTGPU_ESCORE: a traceback whose score is not the pass's maximum (the reference's ERRCODE_SWATSCOR)
enum : uint32_t { RCF_REVERSE = 1, RCF_SCORED = 2, RCF_BANDED = 4, RCF_ERR = 8,
                  RCF_QN = 16 /* the read holds non-ACGT codes: K2a in 32-bit lanes */,
                  RCF_BSCORED = 32 /* swscor is the banded (K2b) score */ };

enum : int {
  HASH_MAXNHITS = 16 * 1024,       // rmap.c:50
  NREPEATS = 4,                    // hashhit.c:42
  MINHIT_PER_TUPLE = 16,           // hashhit.c:43
  HITLST_MINSIZ = 8192, HITLST_BLKSZ = 16384, HITLST_LOGQLEN_FACT = 32,   // hashhit.c:45-48
  HITINFO_MINSEEDNUM = 3, HITINFO_MAXCOVER_PERCENT = 80, HITINFO_MINCOVER_KMER = 2, // hashhit.c:53-56
  HALFBIT = 31,                    // hashhit.h:68
  SEGMENTING_DIFFSHIFT = 3, MAXIMUM_DEPTH = 8000, DEFAULT_TARGET_DEPTH = 200,
  EDGE_BAND_FACTOR = 4, MAX_BANDEDGE_2POW = 4,                              // segment.c:119-141
  MINLEN_QUERY_STRIPED = 32, BWSCAL_QLEN = 48,                              // rmap.c:83-86
  DIR_COL = 1, DIR_ROW = 2, DIR_DIA = 3,                                    // alignment.c:53-59
  DIFF_M = 0, DIFF_D = 1, DIFF_I = 2, DIFF_S = 3, DIFF_MAXMISMATCH = 61, DIFF_TYPSHIFT = 6, // diffstr.h:90-107
  ALILEN_MIN = 5                                                            // alignment.c:50
};

// Sort key of one k-mer hit: strand(1) | seq(10) | diagonal(33) | q(20).  The low 53 bits are
// the reference's packed hit word (hashhit.h:67-72: diagonal<<31 | q) with q narrowed to 20 bits.
enum : int { KEY_QBITS = 20, KEY_DIAGBITS = 33, KEY_SEQBITS = 10 };
constexpr uint64_t KEY_QMASK = (1ull << KEY_QBITS) - 1;
constexpr uint64_t KEY_DIAGMASK = (1ull << KEY_DIAGBITS) - 1;
constexpr uint64_t HALFMASK = 0x7FFFFFFFull;
constexpr uint64_t SOFFSMASK = 0xFFFFFFFFull;

struct DevIndex {                 // I1 + I2 image in HBM (hashidx.c:105-146, sequence.c:148-171)
  int32_t k, s, typ, nbits_key, nbits_lo;
  uint32_t nkeys, npos, nwords;
  const uint32_t *idx, *pos, *wordidx, *posidx, *packed;
  const uint64_t *sop;            // nseq+1 base offsets (device copy)
  const uint32_t *seqlo;          // nseq+1: sop[i]/s, the k-mer serial where sequence i starts
  int32_t nseq;
  uint64_t totlen;
};

struct MapPar {                   // scalar arguments of rmapSingle (rmap.h:127-145)
  int32_t ncut;
  uint32_t min_cover;
  int32_t min_swatscor, below_max, min_basq, target_depth, max_depth;
  uint32_t flags;
  int32_t match, mismatch, gap_init, gap_ext;   // signed scores (-4/-3 for gaps)
  double cov_frac;                              // > 0: min_cover = (uint32_t)(cov_frac * read length) per read (smalt.c:1113-1122)
};

struct SeedRec { uint32_t posidx, nhits, qoffs; };                 // hashhit.c:148-162
struct HitInfoHdr {                                                 // hashhit.c:164-213
  uint32_t n_seeds, seed_rank, status, qlen;
  uint32_t nhit_rank, nhit_tot;
  uint32_t nhit_cut;        // hashCalcHitInfoNumberOfHits(ktuple_maxhit), hashhit.c:1171: what rmapPair compares to pick the rarer mate (rmap.c:1868)
  uint32_t pad1;
};

struct SegSeed { uint64_t sqo; int32_t len; int32_t pad; };        // segment.c:162-192
struct Segment { uint32_t ix; int32_t nseed; uint32_t cover; };    // segment.c:206-215
struct HitRegion { uint32_t idx; int32_t num; };                   // segment.c:195-203

struct SegCand {                                                    // segment.c:239-263
  uint32_t qs, qe, rs, re;
  int16_t shiftoffs, shift2mm, srange;
  uint8_t flag, pad;
  uint32_t cover;
  int32_t nseg;
  uint32_t hregix;
  int32_t seqidx;
};

// The same for reads below 256 bases in 16 bytes (the candidate stage writes one record per candidate, 2000 per read on
// a repeat-rich genome, of which a few hundred are ranked): read offsets fit a byte, the reference range 16 bits
// (srange <= 32767 is checked), the cover lives in the stage's byte array, nseg / hregix are not needed downstream.
struct SegCandP { uint32_t rs, w1, w2, w3; };
SMG_HD inline bool segcand_pack(SegCandP &p, const SegCand &c, uint32_t grp, bool hasgrp) {
  const uint32_t span = c.re - c.rs;
  p.rs = c.rs;
  p.w1 = (span << 16) | ((c.qs & 0xffu) << 8) | (c.qe & 0xffu);
  p.w2 = (uint32_t)(uint16_t)c.shiftoffs | ((uint32_t)(uint16_t)c.shift2mm << 16);
  p.w3 = ((uint32_t)c.srange & 0x7fffu) | ((uint32_t)((c.flag & CANDFLG_REVERSE) ? 1u : 0u) << 15) | ((uint32_t)((c.flag & CANDFLG_MMALI) ? 1u : 0u) << 16) | (hasgrp ? (1u << 17) : 0u) | ((grp & 0x3fffu) << 18);
  return span <= 0xffffu && c.qs <= 0xffu && c.qe <= 0xffu && c.srange >= 0 && !(c.flag & ~(uint8_t)(CANDFLG_REVERSE | CANDFLG_MMALI));
}
SMG_HD inline void segcand_unpack(SegCand &c, const SegCandP &p, uint32_t cover) {
  c.rs = p.rs; c.re = p.rs + (p.w1 >> 16); c.qs = (p.w1 >> 8) & 0xffu; c.qe = p.w1 & 0xffu;
  c.shiftoffs = (int16_t)(uint16_t)(p.w2 & 0xffffu); c.shift2mm = (int16_t)(uint16_t)(p.w2 >> 16);
  c.srange = (int16_t)(p.w3 & 0x7fffu); c.flag = (uint8_t)(((p.w3 >> 15) & 1u ? CANDFLG_REVERSE : 0) | ((p.w3 >> 16) & 1u ? CANDFLG_MMALI : 0)); c.pad = 0;
  c.cover = cover; c.nseg = 0; c.hregix = 0;
  c.seqidx = (p.w3 & (1u << 17)) ? (int32_t)((p.w3 >> 18) & 0x3fffu) : -1;
}

struct RCand {                                                      // rmap.c:111-126
  uint64_t rs, re;          // window in sequence sqidx (or concatenated set if sqidx < 0)
  uint32_t qs, qe;
  int32_t band_l, band_r;
  int32_t sqidx;
  uint32_t flags;           // RCF_*
  uint32_t cover;
  int32_t swscor;
  uint32_t rid;             // read index in the batch
  uint32_t pad;
};

struct CandHdr {                                                    // segment.c:267-284 + rmap.c:1333-1338
  uint32_t ncand, n_sort, n_mincover, max_cover, max2nd_cover;
  uint32_t cover_deficit[2];
  uint32_t rc_off;          // first RCand of this read in the pool
  int32_t err;
  uint32_t nhits[2];        // collected hits per strand (diagnostic)
  uint32_t n_reserved;      // pool entries reserved for this read (= n_sort unless the pool overflowed)
  int32_t err_site;         // source line of the limit or assertion behind `err` (diagnostic; 0: none recorded)
};

struct ReadCtl {                                                    // scalars of mapSingleRead (rmap.c:1373-1400)
  int32_t max1, max2, n_scored;
  int32_t bandwidth_min, min_swatscor, scorlen_min;
  int32_t go;               // 1: run the traceback pass
  int32_t pad;
};

struct Result {                                                     // results.c:121-160
  int32_t swatscor;
  uint32_t q_start, q_end;
  uint32_t reverse;
  uint64_t s_start, s_end;
  int32_t sidx;
  uint32_t stroffs, strlen;
  uint32_t pad;             // 1: first alignment of its candidate (of one resultSetAddFromAli call, results.c:1852)
};

struct ReadStat {
  int32_t swmax, sw2nd, nseg, nseg_tot;
  uint32_t nhit, nhit_tot;
  int32_t err;
  uint32_t nres;
  uint64_t res_off;         // first Result of this read in the pool
  uint64_t dstr_off;
  int32_t max1;             // best first-pass score (max1scor, rmap.c:1355): < 1 means mapSingleRead returned before the traceback pass
  int32_t err_site;         // source line of the limit or assertion behind `err` (diagnostic; 0: none recorded)
};

// rmapPair (rmap.c:1744-2112): one search interval of a read, [lo, hi] 0-based inclusive in sequence sx (interval.c:44-49)
struct IvRec { int32_t sx; uint32_t lo, hi; };
enum : int { FINE_K = 5, FINE_S = 1,              // the on-the-fly index of the rescue round (rmap.c:91-92)
             FINE_NKEYS = 1 << (2 * FINE_K), FINE_IDX_STRIDE = FINE_NKEYS + 8,
             IV_MAX = 1 << 11 };                  // intervals per read: the interval number (0 .. 2047) takes the sequence field of the hit sort key and the bit above it;
                                                   // rmapPair makes one interval per alignment of the mate mapped first (rmap.c:354-436), i.e. up to max_depth = 2048
                                                  // (KEY_SEQBITS + 1 = 11 bits: the wave-parallel candidate stage keeps the strands apart, so bit 63 is free there)

}  // namespace smg
