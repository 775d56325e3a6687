// smg_split.hpp -- split reads for a whole batch: what rmapSingle does with RMAPFLG_SPLIT (rmap.c:1716-1728), read by read, done
// here as two device batches.  The reference maps the read, takes the best alignment of the read's first segment (the alignment
// tables are sorted and grouped by then, results.c:2022), and -- if the larger of the two stretches that alignment leaves
// uncovered can hold a word and a step -- maps once more with k-mer words from that stretch only (mapSecondary, rmap.c:1435-1505:
// hashCollectHitInfo with a range), appending to the same set and running the post-call pass again.
//   round 1: every read, blank sets                         -> tables settled, at rest as byte runs (Table::pack)
//   plan:    per read the stretch of the second call, or none (second_call_range below)
//   round 2: the reads with a stretch, seeds from it, the set's score maxima into the call, every alignment back
//            (Table::take_call puts them behind the set as resultSetAddFromAli would) -> settled again
// The mapping is behind `Exec` (two calls), as in smg_pairrun.hpp: the product's Exec goes through smaltgpu_map_batch /
// smaltgpu_map_batch_ctx (smg_split.cpp), a test's Exec can replay recorded calls.
#ifndef SMG_SPLIT_HPP
#define SMG_SPLIT_HPP
#include <string>
#include <thread>
#include "smg_pairrun.hpp"        // RestStore
#include "smg_postout.hpp"

namespace smgsplit {

using smgpost::Table;

// mapSecondary's choice (rmap.c:1459-1481).  lo/hi: the best alignment of the first read segment on the read, 1-based and
// inclusive; qlen, k, s.  -> true and the stretch [*first, *last] (0-based, inclusive) the second call seeds from
inline bool second_call_range(uint32_t lo, uint32_t hi, uint32_t qlen, int k, int s, uint32_t *first, uint32_t *last) {
  if (hi > qlen || lo > hi) return false;                                       // (the reference: ERRCODE_ASSERT)
  uint32_t a, b;
  if ((uint64_t)lo + hi > qlen) { a = 0; b = lo > 1 ? lo - 2 : 0; }              // the alignment sits towards the end: the front is left
  else { a = hi; b = qlen - 1; }
  if ((uint64_t)a + (uint32_t)k + (uint32_t)s > (uint64_t)b + 1) return false;   // no room for a word and a step
  *first = a; *last = b;
  return true;
}

struct Input { const uint8_t *bases, *quals; const uint64_t *off; uint32_t n; };
struct Setup {
  smaltgpu_params map;
  const uint64_t *sop; int64_t nseq; const uint32_t *packed_host;
  int k, s, nthreads;
};

struct Runner {
  smgpairs::RestStore rest;
  std::vector<uint32_t> ids, range;        // round 2: read numbers, (first, last) per call
  std::vector<int32_t> prev_max;
  std::vector<uint8_t> has_table;          // 1: the read's table is at rest in `rest`
  std::string error;
  int error_code = SMALTGPU_EINTERNAL;
  uint32_t n_second = 0;

  template <class Fn> static void spread(uint32_t n, int nthreads, Fn fn) { smgpairs::PairBlock::spread(n, nthreads, fn); }

  // Exec: bool first(const Input &, smaltgpu_batch_out *), bool second(const Input &, const uint32_t *ids, uint32_t n, const uint32_t *range,
  //                                                                    const int32_t *prev_max, smaltgpu_batch_out *); both leave a message in .err
  template <class Exec> bool run(Exec &ex, const Input &in, const Setup &su, smaltgpu_post *post, smaltgpu_post_out *out) {
    const int nt = su.nthreads < 1 ? 1 : su.nthreads;
    const smgpost::Reference ref{su.sop, su.nseq, su.packed_host};
    const smgpost::Penalties pen{su.map.match, su.map.mismatch, su.map.gap_init, su.map.gap_ext};
    const bool can_cut = su.packed_host != nullptr;
    auto read_of = [&](uint32_t r) { smgpost::Read rd; rd.len = (uint32_t)(in.off[r + 1] - in.off[r]); rd.bases = can_cut ? in.bases + in.off[r] : nullptr;
                                     rd.quals = in.quals ? in.quals + in.off[r] : nullptr; return rd; };
    std::vector<std::string> bad((size_t)nt);
    auto complain = [&](int t, uint32_t r, const char *what, int code, int site) {
      if (!bad[(size_t)t].empty()) return;
      char m[256];
      if (code) snprintf(m, sizeof(m), "read %u: %s (code %d, site %d)", r, what, code, site); else snprintf(m, sizeof(m), "read %u: %s", r, what);
      bad[(size_t)t] = m;
    };
    auto first_complaint = [&]() { for (const std::string &b : bad) if (!b.empty()) { error = b; return true; } return false; };

    // ---- round 1 ----
    smaltgpu_batch_out o;
    if (!ex.first(in, &o)) { error = ex.err; error_code = ex.rc; return false; }
    if (o.nreads != in.n) { error = "the mapping call returned a different number of reads"; return false; }
    rest.reset(in.n);
    has_table.assign(in.n ? in.n : 1, 0);
    std::vector<uint32_t> want((size_t)in.n * 2 + 2, 0);
    std::vector<uint8_t> wants(in.n ? in.n : 1, 0);
    std::vector<int32_t> pm((size_t)in.n * 2 + 2, 0);
    int code_seen = 0;
    const size_t arena1 = rest.open_pass(nt);
    spread(in.n, nt, [&](uint32_t lo, uint32_t hi, int t) {
      Table tb;
      for (uint32_t r = lo; r < hi; r++) {
        const smaltgpu_readstat &st = o.stat[r];
        if (st.errcode) { complain(t, r, st.errcode == SMALTGPU_ESCORE ? "inconsistency when calculating Smith-Waterman scores: the reference stops at this read (ERRCODE_SWATSCOR)" : "the mapping call failed on the device", st.errcode, st.errsite); if (st.errcode == SMALTGPU_ESCORE) code_seen = SMALTGPU_ESCORE; continue; }
        tb.clear();
        tb.n_ali_done = st.n_ali_done; tb.n_ali_tot = st.n_ali_tot; tb.n_hits_used = st.n_hits_used; tb.n_hits_tot = st.n_hits_tot;
        tb.take_call(o.res + o.res_off[r], (uint32_t)(o.res_off[r + 1] - o.res_off[r]), o.diffstr, st.swatscor_max, st.swatscor_2ndmax);
        if (st.max1scor >= 1) {                                  // a call without a score-pass hit returns before the pass (rmap.c:1376)
          const smgpost::Outcome oc = tb.settle(ref, read_of(r), can_cut ? &pen : nullptr);
          if (oc != smgpost::DONE) { complain(t, r, oc == smgpost::WANTS_REFERENCE ? "an alignment crosses reference sequences and no host copy of the reference was given" : tb.why, 0, 0); continue; }
        }
        // resultSetGetResultInSegment(.., 0, 0, ..) (results.c:2118): nothing sorted -> no second call
        if (!tb.by_score.empty() && (tb.set_bits & smgpost::SET_SEGMENTED) && tb.segment_begin.size() >= 2 && tb.segment_begin[1] > tb.segment_begin[0]) {
          const int32_t top = tb.by_segment[(size_t)tb.segment_begin[0]];
          uint32_t a, b;
          if (second_call_range(tb.q_lo[(size_t)top], tb.q_hi[(size_t)top], (uint32_t)(in.off[r + 1] - in.off[r]), su.k, su.s, &a, &b) &&
              (uint32_t)(in.off[r + 1] - in.off[r]) >= (uint32_t)su.k) {
            wants[r] = 1; want[2 * (size_t)r] = a; want[2 * (size_t)r + 1] = b;
            pm[2 * (size_t)r] = tb.score_max; pm[2 * (size_t)r + 1] = tb.score_2nd;
          }
        }
        rest.put(r, arena1 + (size_t)t, tb);
        has_table[r] = 1;
      }
    });
    if (first_complaint()) { if (code_seen) error_code = code_seen; return false; }

    // ---- round 2 ----
    ids.clear(); range.clear(); prev_max.clear();
    for (uint32_t r = 0; r < in.n; r++) if (wants[r]) {
      ids.push_back(r); range.push_back(want[2 * (size_t)r]); range.push_back(want[2 * (size_t)r + 1]);
      prev_max.push_back(pm[2 * (size_t)r]); prev_max.push_back(pm[2 * (size_t)r + 1]);
    }
    n_second = (uint32_t)ids.size();
    if (n_second) {
      smaltgpu_batch_out o2;
      if (!ex.second(in, ids.data(), n_second, range.data(), prev_max.data(), &o2)) { error = ex.err; error_code = ex.rc; return false; }
      if (o2.nreads != n_second) { error = "the second mapping call returned a different number of reads"; return false; }
      const size_t arena2 = rest.open_pass(nt);
      spread(n_second, nt, [&](uint32_t lo, uint32_t hi, int t) {
        Table tb;
        for (uint32_t i = lo; i < hi; i++) {
          const uint32_t r = ids[i];
          const smaltgpu_readstat &st = o2.stat[i];
          if (st.errcode) { complain(t, r, st.errcode == SMALTGPU_ESCORE ? "second call: inconsistency when calculating Smith-Waterman scores: the reference stops at this read (ERRCODE_SWATSCOR)" : "the second mapping call failed on the device", st.errcode, st.errsite); if (st.errcode == SMALTGPU_ESCORE) code_seen = SMALTGPU_ESCORE; continue; }
          tb.unpack(rest.data(r), rest.size(r));
          tb.n_ali_done = st.n_ali_done; tb.n_ali_tot = st.n_ali_tot; tb.n_hits_used = st.n_hits_used; tb.n_hits_tot = st.n_hits_tot;     // rmap.c:1337
          tb.take_call(o2.res + o2.res_off[i], (uint32_t)(o2.res_off[i + 1] - o2.res_off[i]), o2.diffstr, st.swatscor_max, st.swatscor_2ndmax);
          if (st.max1scor >= 1) {
            const smgpost::Outcome oc = tb.settle(ref, read_of(r), can_cut ? &pen : nullptr);
            if (oc != smgpost::DONE) { complain(t, r, oc == smgpost::WANTS_REFERENCE ? "an alignment crosses reference sequences and no host copy of the reference was given" : tb.why, 0, 0); continue; }
          }
          rest.put(r, arena2 + (size_t)t, tb);
        }
      });
      if (first_complaint()) { if (code_seen) error_code = code_seen; return false; }
    }

    // ---- the tables in read order -> smaltgpu_post_out ----
    int nc = nt;
    if ((uint32_t)nc > in.n / 256 + 1) nc = (int)(in.n / 256 + 1);
    std::vector<smgpostout::Chunk> chunks((size_t)nc);
    for (int t = 0; t < nc; t++) { chunks[(size_t)t].lo = (uint32_t)((uint64_t)in.n * t / nc); chunks[(size_t)t].hi = (uint32_t)((uint64_t)in.n * (t + 1) / nc); }
    auto emit = [&](int t) {
      smgpostout::Chunk &ck = chunks[(size_t)t];
      Table tb;
      for (uint32_t r = ck.lo; r < ck.hi; r++) {
        tb.clear();
        if (has_table[r]) tb.unpack(rest.data(r), rest.size(r));
        smgpostout::emit_table(ck, r, tb, smgpost::DONE);
      }
    };
    if (nc == 1) emit(0);
    else { std::vector<std::thread> th; for (int t = 0; t < nc; t++) th.emplace_back(emit, t); for (std::thread &x : th) x.join(); }
    const int rv = smgpostout::stitch(post, chunks, in.n, out, "smaltgpu_map_split");
    if (rv) { error = "assembling the alignment tables failed"; error_code = rv; return false; }
    return true;
  }
};

}  // namespace smgsplit
#endif
