// smg_results.cpp -- result post-processing of the mapped reads of a batch: what the reference's
// resultSetSortAndAssignSequence (results.c:2022) does to the raw alignments of a read --
//   assignSequenceIndex            results.c:1695   concatenated mode: which sequence, offsets relative to it
//   sortAndPrune                   results.c:759    duplicates / contained alignments out, order for output, score ranks
//   labelComplementarySegments     results.c:707    groups of alignments that cover the same part of the read
//   calcPhredScaledMappingQuality  results.c:1143   mapping quality per group (+ propagateMapQualAsProb :1343)
// -- for all reads of a batch, on the host (worker threads over the reads).  Host code on purpose: the mapping quality is
// double arithmetic through libm's log/exp and the orders are libc qsort's on comparators that are not total orders
// (cmpRes compares a query length with a subject length, results.c:466-470), so bit-identical results need the very same
// libm / libc the reference runs on.  Compiled with g++ and -ffp-contract=off (the reference is plain gcc -O2).
//
// Alignments that span several reference sequences (concatenated mode) are cut at the junctions as splitMultiSpan
// (results.c:1472-1648) does -- diffStrSegment (diffstr.c:1370) on the alignment string, the fragment re-scored against the
// read's score profile (aliScoreDiffStr, alignment.c:179) -- when the caller hands in the read bases, the packed reference
// and the penalties; without them such a read is flagged `needs_reference` and left to the caller.
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include "../../include/smaltgpu.h"

namespace {

enum {                                   // results.c:49-91
  MAPSCOR_MAX = 60, MAPSCOR_DUMMY_COUNT = 3, MAPSCOR_MAX_RANDOM = 3, MAPSCOR_MIN_UNIQ = MAPSCOR_MAX_RANDOM + 1,
  MAPSCOR_EXPFAC = 10,                   // results_mapscor_exp is defined for the build (results.h / resultpairs.h:40)
  QUALSCOR_SCAL = 10, RSLTX_INITVAL = -1, QSEGX_INITVAL = -1, N100PERCENT = 100, MIN_QSEGOVERLAP_PERCENT = 80,
  SEQCOD_QVAL_OFFS = 33                  // sequence.h: phred + 33
};
enum { F_SELECT = 0x01, F_REVERSE = 0x04, F_NOSEQID = 0x08, F_SINGLE = 0x100 };                        // results.h:67-78
enum { S_SEQX = 0x01, S_SERIALNO = 0x02, S_SWSORT = 0x04, S_SEGIDX = 0x08, S_MAPQ = 0x10 };            // results.c:93-100
const double MINLOGARG = 1E-7;           // results.c:102
const float QUALSCOR_LOGBASE = 2.30259;  // results.c:103 (a float in the reference: the arithmetic below depends on it)

struct Res {                             // struct _RESULT, results.c:121-160
  short serialno;
  uint32_t status;
  int swatscor, mapscor;
  double prob;
  uint32_t q_start, q_end;
  uint64_t s_start, s_end;
  int64_t sidx;
  uint32_t stroffs, strlen;
  short rsltx, qsegx, swrank;
};

int cmpRes(const void *p1, const void *p2) {           // results.c:450-476, including its mixed comparison of lengths
  const Res *ap = *(Res *const *)p1, *bp = *(Res *const *)p2;
  if (ap->sidx < bp->sidx) return -1;
  if (ap->sidx > bp->sidx) return 1;
  if ((ap->status & F_REVERSE) < (bp->status & F_REVERSE)) return -1;
  if ((ap->status & F_REVERSE) > (bp->status & F_REVERSE)) return 1;
  if (ap->s_start < bp->s_start) return -1;
  if (ap->s_start > bp->s_start) return 1;
  const uint32_t da = ap->q_end - ap->q_start, db = (uint32_t)(bp->s_end - bp->s_start);
  if (da > db) return -1;
  if (da < db) return 1;
  return 0;
}
int cmpResOutput(const void *p1, const void *p2) {     // results.c:478-507
  const Res *ap = *(Res *const *)p1, *bp = *(Res *const *)p2;
  if (ap->swatscor > bp->swatscor) return -1;
  if (ap->swatscor < bp->swatscor) return 1;
  if ((ap->status & F_REVERSE) < (bp->status & F_REVERSE)) return -1;
  if ((ap->status & F_REVERSE) > (bp->status & F_REVERSE)) return 1;
  if (ap->sidx < bp->sidx) return -1;
  if (ap->sidx > bp->sidx) return 1;
  if (ap->s_start < bp->s_start) return -1;
  if (ap->s_start > bp->s_start) return 1;
  const uint32_t da = ap->q_end - ap->q_start, db = bp->q_end - bp->q_start;
  if (da > db) return -1;
  if (da < db) return 1;
  return 0;
}
int cmpResSegSW(const void *p1, const void *p2) {      // results.c:509-523
  const Res *ap = *(Res *const *)p1, *bp = *(Res *const *)p2;
  if (ap->qsegx < bp->qsegx) return -1;
  if (ap->qsegx > bp->qsegx) return 1;
  if (ap->swatscor > bp->swatscor) return -1;
  if (ap->swatscor < bp->swatscor) return 1;
  return 0;
}
int cmpResSegLen(const void *p1, const void *p2) {     // results.c:525-554
  const Res *ap = *(Res *const *)p1, *bp = *(Res *const *)p2;
  if (ap->swatscor > bp->swatscor) return -1;
  if (ap->swatscor < bp->swatscor) return 1;
  const uint32_t da = ap->q_end - ap->q_start, db = bp->q_end - bp->q_start;
  if (da > db) return -1;
  if (da < db) return 1;
  if ((ap->status & F_REVERSE) < (bp->status & F_REVERSE)) return -1;
  if ((ap->status & F_REVERSE) > (bp->status & F_REVERSE)) return 1;
  if (ap->sidx < bp->sidx) return -1;
  if (ap->sidx > bp->sidx) return 1;
  if (ap->s_start < bp->s_start) return -1;
  if (ap->s_start > bp->s_start) return 1;
  return 0;
}

// sumQualOverMisMatch (results.c:232-285) with with_nonali == 0: base qualities summed over the substitutions of an alignment
int sum_qual_mismatch(int *sum, const uint8_t *qual, uint32_t slen, uint32_t pos_start, uint32_t pos_end, const uint8_t *dstr) {
  uint32_t qs = 0, spos;
  if (pos_end < pos_start) return -1;
  *sum = 0;
  spos = pos_start > 0 ? pos_start - 1 : 0;
  for (const uint8_t *dp = dstr; *dp; dp++) {
    const uint32_t gap = *dp & 0x3F, typ = *dp >> 6;                 // diffstr.h:90-107
    spos += gap;
    if (typ == 1) continue;                                          // DIFFCOD_D
    if (typ == 3) {                                                  // DIFFCOD_S
      if (!dp[1]) continue;
      if (spos < 1 || spos >= slen) return -1;
      const uint8_t q = qual[spos];
      if (q < SEQCOD_QVAL_OFFS) return -1;
      qs += (uint32_t)q - SEQCOD_QVAL_OFFS;
      if (qs > (uint32_t)INT_MAX) return -1;
    }
    spos++;
  }
  if (spos != pos_end) return -1;
  *sum = (int)qs;
  return 0;
}

// ---- diffstr.c: the segment of an alignment string between two positions of the unprofiled (reference) sequence ----
enum { DIFFCOD_M = 0, DIFFCOD_D = 1, DIFFCOD_I = 2, DIFFCOD_S = 3, DIFFSTR_MAXMISMATCH = 61, DIFFSTR_TYPSHIFT = 6, DIFFSTR_COUNTMASK = 0x3F };
enum { DS_OK = 0, DS_ERR = -1, DS_NOMATCH = 1 };
inline uint8_t setdiff(int gap, int typ) { return (uint8_t)(gap + (((unsigned char)typ) << DIFFSTR_TYPSHIFT)); }

// scrollDIFFSTRStartEnd (diffstr.c:413-591)
int scroll_start_end(int *start_unprof, int *end_unprof, int *start_prof, int *end_prof, uint8_t *count_start, uint8_t *count_end,
                     uint8_t *typ_start, int *idx_start, int *idx_end, int start_unprof_target, int end_unprof_target, const uint8_t *diffstrp) {
  int i, idx_last, shift = 0, shift_last = 0, pos = 0, pos_last;
  uint8_t count = 0, count_add = 0, typ = 0;
  for (i = 0; diffstrp[i] != 0 && i < INT_MAX; i++) {
    typ = (uint8_t)(diffstrp[i] >> DIFFSTR_TYPSHIFT); count = (uint8_t)(diffstrp[i] & DIFFSTR_COUNTMASK);
    shift_last = shift;
    if (typ == DIFFCOD_M) { count++; count_add = 0; }
    else if (typ == DIFFCOD_S) count_add = 1;
    else if (typ == DIFFCOD_I) { shift++; count_add = 0; }
    else { count_add = 1; shift--; }
    pos += count;
    if (pos > start_unprof_target && count > 0) break;
    pos += count_add;
  }
  if (i >= INT_MAX) return DS_ERR;
  if (diffstrp[i] == 0) return DS_ERR;
  idx_last = i;
  *count_start = (uint8_t)(pos - start_unprof_target);
  if (*count_start > count) *count_start = count;
  *start_unprof = pos - *count_start;
  *start_prof = *start_unprof + shift_last;
  pos_last = pos;
  pos += count_add;
  *idx_start = i;
  *typ_start = typ;
  if (*start_unprof > end_unprof_target) return DS_NOMATCH;
  if (pos <= end_unprof_target) {
    for (i++; diffstrp[i] != 0 && i < INT_MAX; i++) {
      typ = (uint8_t)(diffstrp[i] >> DIFFSTR_TYPSHIFT); count = (uint8_t)(diffstrp[i] & DIFFSTR_COUNTMASK);
      if (count > 0) shift_last = shift;
      if (typ == DIFFCOD_M) { count++; count_add = 0; }
      else if (typ == DIFFCOD_S) count_add = 1;
      else if (typ == DIFFCOD_I) { count_add = 0; shift++; }
      else { count_add = 1; shift--; }
      pos += count;
      if (count > 0) { pos_last = pos; idx_last = i; }
      pos += count_add;
      if (pos > end_unprof_target) break;
    }
    if (diffstrp[i] == 0) i--;
    else if (i >= INT_MAX) return DS_ERR;
  }
  if (pos_last > end_unprof_target) {
    *count_end = (uint8_t)(pos_last - end_unprof_target - 1);
    if (*count_end > count) return DS_ERR;
    *count_end = (uint8_t)(count - *count_end);
    *end_unprof = end_unprof_target;
    *idx_end = i;
  } else {
    typ = (uint8_t)(diffstrp[idx_last] >> DIFFSTR_TYPSHIFT); count = (uint8_t)(diffstrp[idx_last] & DIFFSTR_COUNTMASK);
    if (typ == DIFFCOD_M) count++;
    *count_end = count;
    *end_unprof = pos_last - 1;
    *idx_end = idx_last;
  }
  *end_prof = *end_unprof + shift_last;
  return DS_OK;
}

// diffStrSegment (diffstr.c:1370-1456): the new string goes to `out` (terminating M:0 included)
int diffstr_segment(std::vector<uint8_t> &out, const uint8_t *diffstrp, int start_unprof_target, int end_unprof_target, int *start_unprof,
                    int *end_unprof, int *start_prof, int *end_prof) {
  int i, idx_start, idx_end, nmatch;
  uint8_t count, nmatch_start = 0, nmatch_end = 0, typ, typ_start;
  out.clear();
  const int rv = scroll_start_end(start_unprof, end_unprof, start_prof, end_prof, &nmatch_start, &nmatch_end, &typ_start, &idx_start, &idx_end,
                                  start_unprof_target, end_unprof_target, diffstrp);
  if (rv) return rv;
  nmatch = 0;
  if (idx_start == idx_end) {
    typ = (uint8_t)(diffstrp[idx_start] >> DIFFSTR_TYPSHIFT); count = (uint8_t)(diffstrp[idx_start] & DIFFSTR_COUNTMASK);
    if (typ == DIFFCOD_M) count++;
    nmatch_end = (uint8_t)(nmatch_end + nmatch_start - count);
  } else {
    if (typ_start == DIFFCOD_M) nmatch = nmatch_start;
    else if (nmatch_start > 0) { out.push_back(setdiff(nmatch_start, typ_start)); nmatch = 0; }
    if (idx_end > idx_start + 1) {
      for (i = idx_start + 1; i < idx_end && diffstrp[i] != 0; i++) {
        typ = (uint8_t)(diffstrp[i] >> DIFFSTR_TYPSHIFT); count = (uint8_t)(diffstrp[i] & DIFFSTR_COUNTMASK);
        nmatch += count;
        if (typ == DIFFCOD_M) { nmatch++; continue; }
        for (; nmatch > DIFFSTR_MAXMISMATCH; nmatch -= DIFFSTR_MAXMISMATCH + 1) out.push_back(setdiff(DIFFSTR_MAXMISMATCH, DIFFCOD_M));
        out.push_back(setdiff(nmatch, typ));
        nmatch = 0;
      }
    }
  }
  nmatch += nmatch_end;
  for (; nmatch > DIFFSTR_MAXMISMATCH + 1; nmatch -= DIFFSTR_MAXMISMATCH + 1) out.push_back(setdiff(DIFFSTR_MAXMISMATCH, DIFFCOD_M));
  out.push_back(setdiff(nmatch, DIFFCOD_S));
  out.push_back(setdiff(0, DIFFCOD_M));
  return DS_OK;
}

// what splitMultiSpan needs beside the alignments: the read (for its score profile), the packed reference, the penalties
struct SplitCtx {
  const uint8_t *bases;          // the read, ASCII
  uint32_t qlen;
  const uint32_t *packed;        // 10 bases per word, 3 bits each, first base in bits 29-27 (sequence.c:1360)
  int match, mismatch, gap_init, gap_ext;
};
inline uint32_t read_code(uint8_t c) {                          // make3BitMangledCodec & 7 (sequence.c:287): ACGT(U) 0-3, anything else N
  switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': case 'U': case 'u': return 3; default: return 5; }
}
inline uint32_t ref_code_host(const uint32_t *packed, uint64_t o) {      // as the scalar path sees the reference (smg_logic.hpp ref_code)
  const uint32_t c = (packed[o / 10] >> (3 * (9 - (uint32_t)(o % 10)))) & 7u;
  return (c == 7) ? 0u : ((c == 6 || c == 4) ? 5u : c);
}
inline int subst_score(uint32_t a, uint32_t b, int match, int mismatch) {   // setScoreMatrix (score.c:138-173), alphabet ACGTXN
  if (a >= 6 || b >= 6 || a == 5 || b == 5) return 0;
  if (a == 4 || b == 4) return mismatch - match;
  return a == b ? match : mismatch;
}

struct Set {                             // the parts of struct _ResultSet this path touches
  std::vector<Res> resr;
  std::vector<Res *> sortr, segsrtr;
  std::vector<int> segnor;
  uint32_t status = 0;
  int qsegno = 0;
  int n_ali_done = 0, n_ali_tot = 0;
  uint32_t n_hits_used = 0, n_hits_tot = 0;
  std::vector<uint8_t> newstr;          // alignment strings of fragments: stroffs = STR_NEW | offset into newstr until the batch is merged
  uint32_t str_base = 0x80000000u;
};
enum : uint32_t { STR_NEW = 0x80000000u };
inline const uint8_t *str_of(const Set &rs, const uint8_t *dstr, uint32_t stroffs) {
  return (stroffs & STR_NEW) ? rs.newstr.data() + (stroffs & ~STR_NEW) : dstr + stroffs;
}

// splitMultiSpan (results.c:1472-1648): the alignment residx spans sequences so .. eo-1; its fragments are appended
int split_multi_span(Set &rs, uint32_t residx, int64_t so, int64_t eo, const uint64_t *ofp, int64_t nseq, const uint8_t *dstr, const SplitCtx &cx) {
  if (rs.resr.size() <= residx || so < 0 || eo <= so || eo > nseq || rs.resr[residx].s_start <= ofp[so]) return -1;
  const bool rc = (rs.resr[residx].status & F_REVERSE) != 0;
  const int n = (int)(eo - so);
  std::vector<uint8_t> seg;
  for (int i = 0; i < n; i++) {
    const Res r0 = rs.resr[residx];
    const int64_t idx = so + i;
    int curr_start, curr_end, s_start, s_end, q_start, q_end;
    curr_start = (r0.s_start > ofp[idx]) ? 0 : (int)(ofp[idx] - r0.s_start + 1);
    curr_end = (int)(((r0.s_end <= ofp[idx + 1]) ? r0.s_end : ofp[idx + 1]) - r0.s_start);
    const uint8_t *src = str_of(rs, dstr, r0.stroffs);
    const int rv = diffstr_segment(seg, src, curr_start, curr_end, &s_start, &s_end, &q_start, &q_end);
    if (rv == DS_NOMATCH) continue;
    if (rv) return -1;
    Res h = r0;
    h.stroffs = rs.str_base + (uint32_t)rs.newstr.size();
    h.strlen = (uint32_t)seg.size();
    rs.newstr.insert(rs.newstr.end(), seg.begin(), seg.end());
    uint32_t q0;                                        // 0-based start in the profiled sequence (the read or its reverse complement)
    if (rc) { h.q_start = r0.q_end - (uint32_t)q_end; h.q_end = r0.q_end - (uint32_t)q_start; q0 = cx.qlen - h.q_end; }
    else { h.q_start = r0.q_start + (uint32_t)q_start; h.q_end = r0.q_start + (uint32_t)q_end; q0 = h.q_start - 1; }
    if (h.q_start > h.q_end || h.q_end > cx.qlen) return -1;
    h.s_start = r0.s_start + (uint64_t)s_start - ofp[idx];
    h.s_end = r0.s_start + (uint64_t)s_end - ofp[idx];
    if (h.s_end < h.s_start || h.s_end - h.s_start >= (uint64_t)INT_MAX) return -1;
    h.sidx = idx;
    h.status &= ~(uint32_t)F_NOSEQID;
    h.status |= F_SELECT;
    // aliScoreDiffStr (alignment.c:179-225) over the fragment: reference bases sidx:[s_start-1, s_end-1], read from q0
    {
      const uint64_t g0 = ofp[idx] + h.s_start - 1;
      const int ulen = (int)(h.s_end - h.s_start + 1);
      const uint8_t *d = str_of(rs, dstr, h.stroffs);
      int sw = 0, rsi = 0;
      uint32_t po = q0;
      bool is_open = false;
      uint32_t k;
      for (k = 0; k < h.strlen && d[k]; k++) {
        uint32_t count = d[k] & DIFFSTR_COUNTMASK;
        const uint32_t typ = d[k] >> DIFFSTR_TYPSHIFT;
        if (typ == DIFFCOD_M || (typ == DIFFCOD_S && d[k + 1])) count++;
        if (count > 0) {
          is_open = false;
          for (uint32_t j = 0; j < count; j++) {
            const uint32_t rb = ref_code_host(cx.packed, g0 + (uint64_t)rsi);
            uint32_t qb;
            if (po >= cx.qlen) return -1;
            if (rc) { const uint32_t c = read_code(cx.bases[cx.qlen - 1 - po]); qb = (c & 4) ? c : 3 - c; } else qb = read_code(cx.bases[po]);
            sw += subst_score(rb, qb, cx.match, cx.mismatch);
            rsi++; po++;
            if (po > cx.qlen || rsi > ulen) return -1;
          }
        }
        if (typ == DIFFCOD_I || typ == DIFFCOD_D) {
          if (is_open) sw += cx.gap_ext; else { sw += cx.gap_init; is_open = true; }     // (negative numbers here; scoreGetProfile hands them out positive and subtracts)
          if (typ == DIFFCOD_I) { po++; if (po > cx.qlen) return -1; }
          else { rsi++; if (rsi > ulen) return -1; }
        }
      }
      if (k < h.strlen && d[k]) return -1;
      h.swatscor = sw;
    }
    rs.resr.push_back(h);
  }
  return 0;
}

// assignSequenceIndex (results.c:1695-1781); returns 1 when an alignment spans several sequences and cannot be split here
int assign_sequence_index(Set &rs, const uint64_t *ofp, int64_t nseq, const uint8_t *dstr, const SplitCtx *cx) {
  std::vector<uint32_t> idx;
  for (size_t i = 0; i < rs.resr.size(); i++) if ((rs.resr[i].status & F_SELECT) && rs.resr[i].sidx < 0) idx.push_back((uint32_t)i);
  // ascending s_start; the order of equal keys is immaterial here (each alignment is placed on its own)
  for (size_t a = 1; a < idx.size(); a++) { const uint32_t v = idx[a]; size_t b = a; while (b > 0 && rs.resr[idx[b - 1]].s_start > rs.resr[v].s_start) { idx[b] = idx[b - 1]; b--; } idx[b] = v; }
  int64_t s = 0, e;
  for (size_t i = 0; i < idx.size() && s < nseq; i++) {
    Res *rp = &rs.resr[idx[i]];
    if (rp->status & (F_NOSEQID | F_SELECT)) {
      for (; s < nseq && rp->s_start > ofp[s + 1]; s++);
      if (s >= nseq) return -1;
      for (e = s + 1; e < nseq && rp->s_end > ofp[e]; e++);
      if (rp->s_end > ofp[e]) return -1;
      if (e > s + 1) {
        if (!cx) return 1;
        if (split_multi_span(rs, idx[i], s, e, ofp, nseq, dstr, *cx)) return -1;
        rs.resr[idx[i]].status &= ~(uint32_t)F_SELECT;       // (the array may have moved)
        continue;
      }
      rp->sidx = s;
      rp->s_start -= ofp[s];
      rp->s_end -= ofp[s];
      rp->status &= ~(uint32_t)F_NOSEQID;
    }
  }
  rs.status &= ~(uint32_t)S_SWSORT;
  rs.status |= S_SEQX;
  return 0;
}

// sortAndPrune (results.c:759-837)
int sort_and_prune(Set &rs) {
  rs.sortr.clear();
  for (size_t i = 0; i < rs.resr.size(); i++) {
    Res *rp = &rs.resr[i];
    rp->serialno = (short)i;
    rp->swrank = 0;
    if (rp->status & F_SELECT) rs.sortr.push_back(rp);
  }
  rs.status |= S_SERIALNO;
  size_t nres = rs.sortr.size();
  if (nres < 2) { rs.status |= S_SWSORT; return 0; }
  qsort(rs.sortr.data(), nres, sizeof(Res *), cmpRes);
  Res **prevpp = rs.sortr.data(), **endpp = rs.sortr.data() + nres;
  nres = 1;
  for (Res **dpp = rs.sortr.data() + 1; dpp < endpp; dpp++) {
    if ((*dpp)->s_end > (*prevpp)->s_end || (*dpp)->swatscor > (*prevpp)->swatscor || (*dpp)->q_start < (*prevpp)->q_start ||
        (*dpp)->q_end > (*prevpp)->q_end || (*dpp)->sidx != (*prevpp)->sidx || (((*dpp)->status) & F_REVERSE) != (((*prevpp)->status) & F_REVERSE)) {
      if (nres == (size_t)SHRT_MAX) return -1;
      nres++;
      if ((++prevpp) < dpp) *prevpp = *dpp;
    } else (*dpp)->status &= ~(uint32_t)F_SELECT;
  }
  qsort(rs.sortr.data(), nres, sizeof(Res *), cmpResOutput);
  rs.sortr.resize(nres);
  rs.sortr[0]->swrank = 0;
  for (size_t i = 1; i < nres; i++) {
    if (rs.sortr[i]->swatscor > rs.sortr[i - 1]->swatscor) return -1;
    rs.sortr[i]->swrank = (short)(rs.sortr[i]->swatscor < rs.sortr[i - 1]->swatscor ? rs.sortr[i - 1]->swrank + 1 : rs.sortr[i - 1]->swrank);
  }
  rs.status |= S_SWSORT;
  return 0;
}

// labelComplementarySegments (results.c:707-757) + sortBySegmentAndSWscor (:669-705)
int label_segments(Set &rs) {
  const short n = (short)rs.sortr.size();
  const double min_overlap_frac = ((double)MIN_QSEGOVERLAP_PERCENT) / N100PERCENT;
  if (n < 1) return 0;
  if (n > 1 && !(rs.status & S_SWSORT)) return -1;
  for (short i = 0; i < n; i++) rs.sortr[i]->qsegx = QSEGX_INITVAL;
  short i_start = 0;
  rs.qsegno = 0;
  do {
    Res *r1p = rs.sortr[i_start];
    const uint32_t l1 = r1p->q_end - r1p->q_start;
    r1p->qsegx = (short)rs.qsegno;
    short i = (short)(i_start + 1);
    i_start = 0;
    for (; i < n; i++) {
      Res *r2p = rs.sortr[i];
      if (r2p->qsegx < 0) {
        const uint32_t l2 = r2p->q_end - r2p->q_start;
        const uint32_t min_overlap = (uint32_t)(((l1 < l2) ? l1 : l2) * min_overlap_frac);
        if (r1p->q_start + min_overlap < r2p->q_end && r2p->q_start + min_overlap < r1p->q_end) r2p->qsegx = (short)rs.qsegno;   // TEST_RESULT_OVERLAP
        else if (i_start == 0) i_start = i;
      }
    }
    if (rs.qsegno == SHRT_MAX) return -1;
    rs.qsegno++;
  } while (i_start != 0);
  rs.segsrtr = rs.sortr;
  if (n > 1) qsort(rs.segsrtr.data(), (size_t)n, sizeof(Res *), cmpResSegSW);
  rs.segnor.clear();
  rs.segnor.push_back(0);
  for (short i = 1; i < n; i++) {
    if (rs.segsrtr[i]->qsegx < rs.segsrtr[i - 1]->qsegx) return -1;
    if (rs.segsrtr[i]->qsegx > rs.segsrtr[i - 1]->qsegx) rs.segnor.push_back(i);
  }
  rs.segnor.push_back(n);
  if ((int)rs.segnor.size() != rs.qsegno + 1) return -1;
  rs.status |= S_SEGIDX;
  return 0;
}

// calcPhredScaledMappingQuality (results.c:1143-1341; the build defines results_mapscor_exp, not results_loscor_capped)
int mapping_quality(Set &rs, short qsegx, const uint8_t *qual, uint32_t qlen, const uint8_t *dstr0) {
  struct { const Set &rs; const uint8_t *d; const uint8_t *operator+(uint32_t o) const { return str_of(rs, d, o); } } dstr = {rs, dstr0};
  if (!(rs.status & S_SEGIDX) || qsegx < 0 || qsegx >= rs.qsegno) return -1;
  Res **rspp = rs.segsrtr.data() + rs.segnor[(size_t)qsegx];
  const short n = (short)(rs.segnor[(size_t)qsegx + 1] - rs.segnor[(size_t)qsegx]);
  short i, i_min, n_swatscor_2nd = 0;
  int qn, swatscor_2nd, mapscor, maxmapscor, qvalsum_1st = 0, qvalsum_2nd = 0, qvalsum_ali;
  double fs, fa;
  if (n < 1) return 0;
  const int swatscor_1st = rspp[0]->swatscor;
  if (swatscor_1st < 1) { rspp[0]->mapscor = 0; return 0; }
  fs = ((double)rs.n_hits_used) / (rs.n_hits_tot + MAPSCOR_DUMMY_COUNT);
  fa = ((double)rs.n_ali_done) / (rs.n_ali_tot + MAPSCOR_DUMMY_COUNT);
  if (fs > fa) fs = fa;
  fs = (fs > MINLOGARG) ? -QUALSCOR_SCAL * log(fs) / QUALSCOR_LOGBASE : MAPSCOR_MAX;
  maxmapscor = (fs < MAPSCOR_MAX) ? MAPSCOR_MAX - (int)fs : 0;
  if (n > 1) {
    swatscor_2nd = rspp[1]->swatscor;
    for (i = 2; i < n && rspp[i]->swatscor == swatscor_2nd; i++);
    n_swatscor_2nd = (short)(i - 1);
    qn = (int)(QUALSCOR_SCAL * log((double)(n_swatscor_2nd)) / QUALSCOR_LOGBASE);
  } else { swatscor_2nd = 0; n_swatscor_2nd = 0; qn = 0; }
  if (swatscor_2nd == swatscor_1st && n > 1) {
    qsort(rspp, (size_t)n_swatscor_2nd + 1, sizeof(Res *), cmpResSegLen);
    const uint32_t seglen_1st = rspp[0]->q_end - rspp[0]->q_start;
    uint32_t seglen = rspp[1]->q_end - rspp[1]->q_start;
    if (seglen_1st == seglen) {
      if (qual) {
        if (sum_qual_mismatch(&qvalsum_1st, qual, qlen, rspp[0]->q_start, rspp[0]->q_end, dstr + rspp[0]->stroffs)) return -1;
        if (sum_qual_mismatch(&qvalsum_2nd, qual, qlen, rspp[1]->q_start, rspp[1]->q_end, dstr + rspp[1]->stroffs)) return -1;
        i_min = 1;
        for (i = 2; i < n && rspp[i]->swatscor == swatscor_1st; i++) {
          seglen = rspp[i]->q_end - rspp[i]->q_start;
          if (seglen < seglen_1st) break;
          if (sum_qual_mismatch(&qvalsum_ali, qual, qlen, rspp[i]->q_start, rspp[i]->q_end, dstr + rspp[i]->stroffs)) return -1;
          if (qvalsum_ali < qvalsum_2nd) { qvalsum_2nd = qvalsum_ali; i_min = i; }
        }
        if (qvalsum_1st > qvalsum_2nd) { Res *t = rspp[i_min]; rspp[i_min] = rspp[0]; rspp[0] = t; mapscor = MAPSCOR_MIN_UNIQ; }
        else mapscor = (qvalsum_1st == qvalsum_2nd) ? 0 : MAPSCOR_MIN_UNIQ;
      } else mapscor = 0;
    } else mapscor = MAPSCOR_MIN_UNIQ;
    if (mapscor < 1) qsort(rspp, (size_t)n_swatscor_2nd + 1, sizeof(Res *), cmpResOutput);
  } else {
    // results.c:1299-1303: exponential scaling of the score difference (double * int / uint32_t, as written there)
    mapscor = (int)(MAPSCOR_MAX * (1 - exp(((double)(swatscor_2nd - swatscor_1st)) * MAPSCOR_EXPFAC / qlen)) - qn);
    if (mapscor >= 0) mapscor += MAPSCOR_MIN_UNIQ;
    if (mapscor > maxmapscor) mapscor = maxmapscor;
  }
  if (mapscor > MAPSCOR_MAX) mapscor = MAPSCOR_MAX;
  else if (mapscor < 0) mapscor = 0;
  rspp[0]->mapscor = mapscor;
  for (i = 1; i < n; i++) rspp[i]->mapscor = 0;
  return 0;
}

// propagateMapQualAsProb (results.c:1343-1398)
int propagate_prob(Set &rs, short qsegx) {
  Res **rspp = rs.segsrtr.data() + rs.segnor[(size_t)qsegx];
  const short nn = (short)(rs.segnor[(size_t)qsegx + 1] - rs.segnor[(size_t)qsegx]);
  short i, ns, n1 = 0, n2 = 0;
  double p1 = 0.0, p2 = 0.0;
  if (nn < 1) return 0;
  for (i = 1; i < nn && rspp[i]->swatscor == rspp[0]->swatscor; i++);
  n1 = i;
  if (i < nn) { for (++i; i < nn && rspp[i]->swatscor == rspp[n1]->swatscor; i++); n2 = (short)(i - n1); }
  if (n1 == 1) {
    int isc = rspp[0]->mapscor;
    if (isc < 0) isc = 0;
    p2 = exp(((double)(-QUALSCOR_LOGBASE * isc)) / QUALSCOR_SCAL);       // float * int, then double: as written in the reference
    p1 = 1.0 - p2;
    if (n2 > 1) p2 /= n2;
  } else if (n1 > 1) { p1 = 1.0 / n1; p2 = p1; }
  for (i = 0; i < n1; i++) rspp[i]->prob = p1;
  if (n1 + n2 > SHRT_MAX) return -1;
  ns = (short)(n1 + n2);
  for (; i < ns; i++) rspp[i]->prob = p2;
  for (; i < nn; i++) rspp[i]->prob = 0.0;
  if (1 == n1 && 0 == n2) rspp[0]->status |= F_SINGLE;
  return 0;
}

// resultSetSortAndAssignSequence (results.c:2022-2064) without the split-read search (search_split == 0 on this path)
int post_one(Set &rs, const uint64_t *sop, int64_t nseq, const uint8_t *qual, uint32_t qlen, const uint8_t *dstr, const SplitCtx *cx) {
  int rv = assign_sequence_index(rs, sop, nseq, dstr, cx);
  if (rv) return rv;
  if (sort_and_prune(rs)) return -1;
  rs.qsegno = 0;
  if (!rs.sortr.empty()) {
    if (label_segments(rs)) return -1;
    for (short q = 0; q < rs.qsegno; q++) { if (mapping_quality(rs, q, qual, qlen, dstr)) return -1; if (propagate_prob(rs, q)) return -1; }
    rs.status |= S_MAPQ;
  }
  return 0;
}

}  // namespace

struct smaltgpu_post {
  std::vector<uint64_t> res_off, sort_off, seg_off;
  std::vector<smaltgpu_post_result> res;
  std::vector<int32_t> sortr, segsrtr, segnor, qsegno, needs_reference;
  std::vector<uint32_t> setstatus;
  std::vector<uint8_t> dstr;             // the batch's string pool + the strings of split fragments (only when there are any)
};

extern "C" smaltgpu_post *smaltgpu_post_create(void) { return new smaltgpu_post(); }
extern "C" void smaltgpu_post_free(smaltgpu_post *p) { delete p; }

extern "C" int smaltgpu_postprocess(smaltgpu_post *pp, const uint64_t *sop, int64_t nseq, const smaltgpu_batch_out *raw, const uint8_t *bases,
                                    const uint8_t *quals, const uint64_t *read_off, const uint32_t *packed_host, const smaltgpu_params *par, int nthreads,
                                    smaltgpu_post_out *out) {
  if (!pp || !sop || !raw || !read_off || !out || nseq < 1) return SMALTGPU_EARG;
  const uint32_t n = raw->nreads;
  const bool can_split = bases && packed_host && par;
  struct PerRead { std::vector<smaltgpu_post_result> res; std::vector<int32_t> sortr, segsrtr, segnor; std::vector<uint8_t> newstr; int qsegno = 0, needs = 0, err = 0; uint32_t status = 0; };
  std::vector<PerRead> per(n);
  auto work = [&](uint32_t lo, uint32_t hi) {
    Set rs;
    for (uint32_t r = lo; r < hi; r++) {
      PerRead &pr = per[r];
      const uint64_t a = raw->res_off[r], b = raw->res_off[r + 1];
      const smaltgpu_readstat &st = raw->stat[r];
      rs.resr.clear(); rs.sortr.clear(); rs.segsrtr.clear(); rs.segnor.clear(); rs.newstr.clear(); rs.status = 0; rs.qsegno = 0;
      rs.n_ali_done = st.n_ali_done; rs.n_ali_tot = st.n_ali_tot; rs.n_hits_used = st.n_hits_used; rs.n_hits_tot = st.n_hits_tot;
      for (uint64_t j = a; j < b; j++) {                                   // as resultSetAddFromAli leaves them (results.c:1885-1922)
        const smaltgpu_result &x = raw->res[j];
        Res q;
        memset(&q, 0, sizeof(q));
        q.swatscor = x.swatscor; q.q_start = x.q_start; q.q_end = x.q_end; q.s_start = x.s_start; q.s_end = x.s_end; q.sidx = x.sidx;
        q.status = F_SELECT | ((x.reverse & SMALTGPU_RES_REVERSE) ? F_REVERSE : 0u) | (x.sidx < 0 ? F_NOSEQID : 0u);
        q.stroffs = x.stroffs; q.strlen = x.strlen; q.rsltx = RSLTX_INITVAL; q.qsegx = QSEGX_INITVAL;
        rs.resr.push_back(q);
      }
      const uint32_t qlen = (uint32_t)(read_off[r + 1] - read_off[r]);
      int rv = 0;
      // mapSingleRead sorts only when the score pass found something (rmap.c:1376); an unmapped read keeps a blank set
      SplitCtx cx;
      if (can_split) { cx.bases = bases + read_off[r]; cx.qlen = qlen; cx.packed = packed_host; cx.match = par->match; cx.mismatch = par->mismatch; cx.gap_init = par->gap_init; cx.gap_ext = par->gap_ext; }
      if (st.max1scor >= 1 && !st.errcode) rv = post_one(rs, sop, nseq, quals ? quals + read_off[r] : nullptr, qlen, raw->diffstr, can_split ? &cx : nullptr);
      pr.newstr.swap(rs.newstr);
      if (rv > 0) pr.needs = 1; else if (rv < 0) pr.err = 1;
      pr.status = rs.status; pr.qsegno = rs.qsegno;
      for (const Res &q : rs.resr) {
        smaltgpu_post_result o;
        memset(&o, 0, sizeof(o));
        o.swatscor = q.swatscor; o.q_start = q.q_start; o.q_end = q.q_end; o.s_start = q.s_start; o.s_end = q.s_end; o.sidx = (int32_t)q.sidx;
        o.status = q.status; o.mapscor = q.mapscor; o.prob = q.prob; o.rsltx = q.rsltx; o.qsegx = q.qsegx; o.swrank = q.swrank; o.stroffs = q.stroffs; o.strlen = q.strlen;
        pr.res.push_back(o);
      }
      if (!pr.needs && !pr.err) {
        for (Res *p : rs.sortr) pr.sortr.push_back((int32_t)(p - rs.resr.data()));
        if (rs.status & S_SEGIDX) { for (Res *p : rs.segsrtr) pr.segsrtr.push_back((int32_t)(p - rs.resr.data())); for (int v : rs.segnor) pr.segnor.push_back(v); }
      }
    }
  };
  if (nthreads < 1) nthreads = 1;
  if ((uint32_t)nthreads > n / 256 + 1) nthreads = (int)(n / 256 + 1);
  if (nthreads == 1) work(0, n);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(work, (uint32_t)((uint64_t)n * t / nthreads), (uint32_t)((uint64_t)n * (t + 1) / nthreads));
    for (std::thread &t : th) t.join();
  }
  smaltgpu_post &P = *pp;
  P.res_off.assign((size_t)n + 1, 0); P.sort_off.assign((size_t)n + 1, 0); P.seg_off.assign((size_t)n + 1, 0);
  P.res.clear(); P.sortr.clear(); P.segsrtr.clear(); P.segnor.clear(); P.qsegno.assign(n ? n : 1, 0); P.needs_reference.assign(n ? n : 1, 0); P.setstatus.assign(n ? n : 1, 0);
  int nerr = 0;
  size_t nnew = 0, nraw = 0;
  for (uint32_t r = 0; r < n; r++) nnew += per[r].newstr.size();
  P.dstr.clear();
  if (nnew) {                              // fragments of split alignments: their strings go behind a copy of the batch's pool
    for (uint64_t j = 0; j < raw->res_off[n]; j++) { const size_t e = (size_t)raw->res[j].stroffs + raw->res[j].strlen; if (e > nraw) nraw = e; }
    if (nraw + nnew >= (size_t)STR_NEW) return SMALTGPU_ECAP;
    P.dstr.assign(raw->diffstr, raw->diffstr + nraw);
  }
  for (uint32_t r = 0; r < n; r++) {
    PerRead &pr = per[r];
    P.res_off[r] = P.res.size(); P.sort_off[r] = P.sortr.size(); P.seg_off[r] = P.segnor.size();
    if (!pr.newstr.empty()) {
      const uint32_t base = (uint32_t)P.dstr.size();
      for (smaltgpu_post_result &o : pr.res) if (o.stroffs & STR_NEW) o.stroffs = base + (o.stroffs & ~(uint32_t)STR_NEW);
      P.dstr.insert(P.dstr.end(), pr.newstr.begin(), pr.newstr.end());
    }
    P.res.insert(P.res.end(), pr.res.begin(), pr.res.end());
    P.sortr.insert(P.sortr.end(), pr.sortr.begin(), pr.sortr.end());
    P.segsrtr.insert(P.segsrtr.end(), pr.segsrtr.begin(), pr.segsrtr.end());
    P.segsrtr.resize(P.sortr.size(), -1);                                  // parallel to sortr (empty when the set has no segment index)
    P.segnor.insert(P.segnor.end(), pr.segnor.begin(), pr.segnor.end());
    P.qsegno[r] = pr.qsegno; P.needs_reference[r] = pr.needs; P.setstatus[r] = pr.status;
    nerr += pr.err;
  }
  P.res_off[n] = P.res.size(); P.sort_off[n] = P.sortr.size(); P.seg_off[n] = P.segnor.size();
  if (P.res.empty()) P.res.resize(1);
  if (P.sortr.empty()) { P.sortr.resize(1); P.segsrtr.resize(1); }
  if (P.segnor.empty()) P.segnor.resize(1);
  out->nreads = n; out->res_off = P.res_off.data(); out->res = P.res.data(); out->diffstr = nnew ? P.dstr.data() : raw->diffstr; out->sort_off = P.sort_off.data();
  out->sortr = P.sortr.data(); out->segsrtr = P.segsrtr.data(); out->seg_off = P.seg_off.data(); out->segnor = P.segnor.data();
  out->qsegno = P.qsegno.data(); out->setstatus = P.setstatus.data(); out->needs_reference = P.needs_reference.data();
  return nerr ? SMALTGPU_EINTERNAL : SMALTGPU_OK;
}
