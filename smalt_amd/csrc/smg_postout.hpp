// smg_postout.hpp -- from settled alignment tables (smg_post.hpp) to the arrays of smaltgpu_post_out: what one worker thread
// produced for a range of reads (Chunk), and the stitching of the chunks in read order.  Shared by smaltgpu_postprocess
// (smg_post.cpp) and the split-read runner (smg_split.cpp).
#ifndef SMG_POSTOUT_HPP
#define SMG_POSTOUT_HPP
#include <stdio.h>
#include "../../include/smaltgpu.h"
#include "smg_post.hpp"

extern "C" int smaltgpu_set_error(int code, const char *msg);   // smaltgpu.cpp: the per-thread message of smaltgpu_last_error()

struct smaltgpu_post {
  std::vector<uint64_t> res_off, sort_off, seg_off;
  std::vector<smaltgpu_post_result> res;
  std::vector<int32_t> sortr, segsrtr, segnor, qsegno, needs_reference;
  std::vector<uint32_t> setstatus;
  std::vector<uint8_t> dstr;
};

namespace smgpostout {

struct Chunk {                          // what one worker produced for reads [lo, hi)
  uint32_t lo = 0, hi = 0;
  std::vector<smaltgpu_post_result> rows;
  std::vector<int32_t> by_score, by_segment, segment_begin;
  std::vector<uint8_t> strings;
  std::vector<uint32_t> nrows, nlive, nbegin, set_bits;      // per read
  std::vector<int32_t> nsegments, wants_reference;
  int64_t bad_read = -1;
  const char *bad_why = "";
};

// the rows of one read behind the chunk's rows (oc: what the table's pass returned)
inline void emit_table(Chunk &ck, uint32_t r, const smgpost::Table &tb, smgpost::Outcome oc) {
    if (oc == smgpost::BROKEN && ck.bad_read < 0) { ck.bad_read = r; ck.bad_why = tb.why; }
    const uint32_t base = (uint32_t)ck.strings.size();
    ck.strings.insert(ck.strings.end(), tb.strings.begin(), tb.strings.end());
    for (uint32_t i = 0; i < tb.rows(); i++) {
      smaltgpu_post_result o;
      memset(&o, 0, sizeof(o));
      o.swatscor = tb.score[i]; o.q_start = tb.q_lo[i]; o.q_end = tb.q_hi[i]; o.s_start = tb.r_lo[i]; o.s_end = tb.r_hi[i]; o.sidx = (int32_t)tb.seq[i];
      o.status = tb.bits[i]; o.mapscor = tb.quality[i]; o.prob = tb.prob[i]; o.rsltx = tb.primary[i]; o.qsegx = tb.segment[i]; o.swrank = tb.rank[i];
      o.stroffs = base + tb.str_at[i]; o.strlen = tb.str_len[i];
      ck.rows.push_back(o);
    }
    const bool whole = oc == smgpost::DONE;
    const bool segmented = whole && (tb.set_bits & smgpost::SET_SEGMENTED);
    if (whole) {
      ck.by_score.insert(ck.by_score.end(), tb.by_score.begin(), tb.by_score.end());
      if (segmented) { ck.by_segment.insert(ck.by_segment.end(), tb.by_segment.begin(), tb.by_segment.end()); ck.segment_begin.insert(ck.segment_begin.end(), tb.segment_begin.begin(), tb.segment_begin.end()); }
      else ck.by_segment.insert(ck.by_segment.end(), tb.by_score.size(), -1);
    }
    ck.nrows.push_back(tb.rows());
    ck.nlive.push_back(whole ? (uint32_t)tb.by_score.size() : 0u);
    ck.nbegin.push_back(segmented ? (uint32_t)tb.segment_begin.size() : 0u);
    ck.set_bits.push_back(tb.set_bits);
    ck.nsegments.push_back(tb.nsegments);
    ck.wants_reference.push_back(oc == smgpost::WANTS_REFERENCE);
}

// offsets of every read, string offsets moved behind the strings of the chunks in front; what: the caller's name for messages
inline int stitch(smaltgpu_post *pp, std::vector<Chunk> &chunks, uint32_t n, smaltgpu_post_out *out, const char *what) {
  // stitch the chunks: offsets of every read, string offsets moved behind the strings of the chunks in front
  smaltgpu_post &P = *pp;
  size_t nrow = 0, nlive = 0, nbeg = 0, nstr = 0;
  for (const Chunk &ck : chunks) { nrow += ck.rows.size(); nlive += ck.by_score.size(); nbeg += ck.segment_begin.size(); nstr += ck.strings.size(); }
  P.res_off.assign((size_t)n + 1, 0); P.sort_off.assign((size_t)n + 1, 0); P.seg_off.assign((size_t)n + 1, 0);
  P.res.clear(); P.res.reserve(nrow + 1); P.sortr.clear(); P.sortr.reserve(nlive + 1); P.segsrtr.clear(); P.segsrtr.reserve(nlive + 1);
  P.segnor.clear(); P.segnor.reserve(nbeg + 1); P.dstr.clear(); P.dstr.reserve(nstr + 1);
  P.qsegno.assign(n ? n : 1, 0); P.needs_reference.assign(n ? n : 1, 0); P.setstatus.assign(n ? n : 1, 0);
  int64_t bad_read = -1;
  const char *bad_why = "";
  bool too_long = nstr >= (size_t)UINT32_MAX;
  for (Chunk &ck : chunks) {
    const uint32_t shift = (uint32_t)P.dstr.size();
    if (!too_long && shift) for (smaltgpu_post_result &o : ck.rows) o.stroffs += shift;
    P.dstr.insert(P.dstr.end(), ck.strings.begin(), ck.strings.end());
    uint64_t a = P.res.size(), b = P.sortr.size(), c = P.segnor.size();
    for (uint32_t r = ck.lo, i = 0; r < ck.hi; r++, i++) {
      P.res_off[r] = a; P.sort_off[r] = b; P.seg_off[r] = c;
      a += ck.nrows[i]; b += ck.nlive[i]; c += ck.nbegin[i];
      P.qsegno[r] = ck.nsegments[i]; P.needs_reference[r] = ck.wants_reference[i]; P.setstatus[r] = ck.set_bits[i];
    }
    P.res.insert(P.res.end(), ck.rows.begin(), ck.rows.end());
    P.sortr.insert(P.sortr.end(), ck.by_score.begin(), ck.by_score.end());
    P.segsrtr.insert(P.segsrtr.end(), ck.by_segment.begin(), ck.by_segment.end());
    P.segnor.insert(P.segnor.end(), ck.segment_begin.begin(), ck.segment_begin.end());
    if (ck.bad_read >= 0 && bad_read < 0) { bad_read = ck.bad_read; bad_why = ck.bad_why; }
  }
  P.res_off[n] = P.res.size(); P.sort_off[n] = P.sortr.size(); P.seg_off[n] = P.segnor.size();
  if (P.res.empty()) P.res.resize(1);
  if (P.sortr.empty()) { P.sortr.resize(1); P.segsrtr.resize(1); }
  if (P.segnor.empty()) P.segnor.resize(1);
  if (P.dstr.empty()) P.dstr.resize(1);
  out->nreads = n; out->res_off = P.res_off.data(); out->res = P.res.data(); out->diffstr = P.dstr.data(); out->sort_off = P.sort_off.data();
  out->sortr = P.sortr.data(); out->segsrtr = P.segsrtr.data(); out->seg_off = P.seg_off.data(); out->segnor = P.segnor.data();
  out->qsegno = P.qsegno.data(); out->setstatus = P.setstatus.data(); out->needs_reference = P.needs_reference.data();
  if (too_long) { char m[160]; snprintf(m, sizeof(m), "%s: the alignment strings of the batch exceed 4 GB; map smaller batches", what); return smaltgpu_set_error(SMALTGPU_ECAP, m); }
  if (bad_read >= 0) {
    char m[192];
    snprintf(m, sizeof(m), "%s: read %lld: %s", what, (long long)bad_read, bad_why);
    return smaltgpu_set_error(SMALTGPU_EINTERNAL, m);
  }
  return SMALTGPU_OK;
}

}  // namespace smgpostout
#endif
