// smg_post.cpp -- smaltgpu_postprocess: the post-call pass of smg_post.hpp over every read of a batch (SURVEY 8f N1;
// the reference runs resultSetSortAndAssignSequence, results.c:2022, read by read behind each mapSingleRead).
// Host code on worker threads: each worker owns one alignment table, fills it from the raw alignments of a read, runs the
// pass and appends the read's rows to its own output chunk; the chunks are stitched together in read order at the end.
// Compiled with g++ -ffp-contract=off (the mapping quality is double arithmetic that has to match the reference's gcc -O2).
#include <stdio.h>
#include <thread>
#include "../../include/smaltgpu.h"
#include "smg_post.hpp"

extern "C" int smaltgpu_set_error(int code, const char *msg);   // smaltgpu.cpp: the per-thread message of smaltgpu_last_error()

namespace {

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

void run_range(Chunk &ck, const smaltgpu_batch_out &raw, const smgpost::Reference &ref, const uint8_t *bases, const uint8_t *quals,
               const uint64_t *read_off, const smgpost::Penalties *pen) {
  smgpost::Table tb;
  for (uint32_t r = ck.lo; r < ck.hi; r++) {
    const smaltgpu_readstat &st = raw.stat[r];
    tb.clear();
    tb.n_ali_done = st.n_ali_done; tb.n_ali_tot = st.n_ali_tot; tb.n_hits_used = st.n_hits_used; tb.n_hits_tot = st.n_hits_tot;
    for (uint64_t j = raw.res_off[r]; j < raw.res_off[r + 1]; j++) {
      const smaltgpu_result &x = raw.res[j];
      tb.add(x.swatscor, x.q_start, x.q_end, x.s_start, x.s_end, x.sidx, (x.reverse & SMALTGPU_RES_REVERSE) != 0, raw.diffstr + x.stroffs, x.strlen);
    }
    smgpost::Read rd;
    rd.len = (uint32_t)(read_off[r + 1] - read_off[r]);
    rd.bases = bases ? bases + read_off[r] : nullptr;
    rd.quals = quals ? quals + read_off[r] : nullptr;
    // a call that found nothing in its score pass returns before this pass (rmap.c:1376): the set stays as filled
    smgpost::Outcome oc = smgpost::DONE;
    if (st.max1scor >= 1 && !st.errcode) oc = tb.settle(ref, rd, pen);
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
}

}  // namespace

struct smaltgpu_post {
  std::vector<uint64_t> res_off, sort_off, seg_off;
  std::vector<smaltgpu_post_result> res;
  std::vector<int32_t> sortr, segsrtr, segnor, qsegno, needs_reference;
  std::vector<uint32_t> setstatus;
  std::vector<uint8_t> dstr;
};

extern "C" smaltgpu_post *smaltgpu_post_create(void) { return new smaltgpu_post(); }
extern "C" void smaltgpu_post_free(smaltgpu_post *p) { delete p; }

extern "C" int smaltgpu_postprocess(smaltgpu_post *pp, const uint64_t *sop, int64_t nseq, const smaltgpu_batch_out *raw, const uint8_t *bases,
                                    const uint8_t *quals, const uint64_t *read_off, const uint32_t *packed_host, const smaltgpu_params *par, int nthreads,
                                    smaltgpu_post_out *out) {
  if (out) memset(out, 0, sizeof(*out));
  if (!pp || !sop || !raw || !read_off || !out || nseq < 1) return smaltgpu_set_error(SMALTGPU_EARG, "smaltgpu_postprocess: null argument or no reference sequence");
  const uint32_t n = raw->nreads;
  const bool can_cut = bases && packed_host && par;
  smgpost::Reference ref{sop, nseq, can_cut ? packed_host : nullptr};
  smgpost::Penalties pen{0, 0, 0, 0};
  if (par) { pen.match = par->match; pen.mismatch = par->mismatch; pen.gap_open = par->gap_init; pen.gap_ext = par->gap_ext; }

  if (nthreads < 1) nthreads = 1;
  if ((uint32_t)nthreads > n / 256 + 1) nthreads = (int)(n / 256 + 1);
  std::vector<Chunk> chunks((size_t)nthreads);
  for (int t = 0; t < nthreads; t++) { chunks[(size_t)t].lo = (uint32_t)((uint64_t)n * t / nthreads); chunks[(size_t)t].hi = (uint32_t)((uint64_t)n * (t + 1) / nthreads); }
  if (nthreads == 1) run_range(chunks[0], *raw, ref, can_cut ? bases : nullptr, quals, read_off, can_cut ? &pen : nullptr);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(run_range, std::ref(chunks[(size_t)t]), std::cref(*raw), std::cref(ref), can_cut ? bases : nullptr, quals, read_off, can_cut ? &pen : nullptr);
    for (std::thread &t : th) t.join();
  }

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
  if (too_long) return smaltgpu_set_error(SMALTGPU_ECAP, "smaltgpu_postprocess: the alignment strings of the batch exceed 4 GB; map smaller batches");
  if (bad_read >= 0) {
    char m[192];
    snprintf(m, sizeof(m), "smaltgpu_postprocess: read %lld: %s", (long long)bad_read, bad_why);
    return smaltgpu_set_error(SMALTGPU_EINTERNAL, m);
  }
  return SMALTGPU_OK;
}
