// smg_pairrun.hpp -- a block of read pairs through the rounds of smg_pairs.hpp (PairPlan): what rmapPair (rmap.c:1744-2112)
// does pair by pair, done round by round for the whole block so that every round is ONE batch for the device.
// The mapping itself is behind `Exec` (two calls: k-mer hit totals of reads, one round of mapping calls with per-read
// context); the product's Exec sends them through smaltgpu_hit_totals / smaltgpu_map_batch_ctx (smg_pairs.cpp), the
// parity test's Exec replays the calls the reference recorded (tests/hostemu/pair_check.cpp).
// Between rounds a pair's two alignment tables rest as packed byte runs (Table::pack); worker threads unpack a table,
// append the round's alignments, run the post-call pass (smg_post.hpp) and pack it again.
#ifndef SMG_PAIRRUN_HPP
#define SMG_PAIRRUN_HPP
#include <stdio.h>
#include <chrono>
#include <string>
#include <thread>
#include "../../include/smaltgpu.h"
#include "smg_pairs.hpp"

namespace smgpairs {

struct BlockInput {                         // two batches in the layout of smaltgpu_map_batch: [0] reads, [1] mates
  const uint8_t *bases[2], *quals[2];       // quals[w] may be null; bases[w] may be null when the reads are resident on the device
  const uint64_t *off[2];
  uint32_t npairs;
};
struct BlockParams {
  smaltgpu_params map;                      // per-call parameters (best-only mapping is forced, as rmapPair does: rmap.c:1848)
  int d_min, d_max, lib;
  bool every_pair;                          // -x: the unrestricted round for every pair (RMAPFLG_ALLPAIR, rmap.c:1965)
  int k;                                    // word length of the index
  const uint64_t *sop; int64_t nseq;        // sequence offsets
  const uint32_t *packed_host;              // host copy of the packed reference, or null when alignments cannot cross sequences
  int nthreads;
  bool split = false;                       // RMAPFLG_SPLIT: mapSecondary for both mates ahead of the pairing (rmap.c:2073-2097)
  int s = 0;                                // sampling step of the index (the stretch of a second call holds a word and a step)
};
enum RoundKind { ROUND_PLAIN = 0, ROUND_RESTRICTED = 1, ROUND_APPEND = 2, ROUND_FINE = 3 };
struct Round {                              // one batch of mapping calls: read ids are 2 * pair + mate
  int kind;
  const uint32_t *ids; uint32_t n;
  const uint64_t *iv_off; const smaltgpu_interval *iv;      // ROUND_RESTRICTED, ROUND_FINE
  const int32_t *min_score;                                  // ROUND_FINE
  const int32_t *prev_max;                                   // ROUND_APPEND, ROUND_FINE: (max, second) per call
  const uint32_t *seed_range = nullptr;                      // split reads (a ROUND_APPEND): (first, last) base the k-mer words of a call come from
};

// The tables of a block at rest: byte runs (Table::pack) in arenas, one arena per worker thread and post-call pass -- a pass only
// appends to its own arenas and reads those of earlier passes, and a block re-uses the arenas of the block before it: no
// allocation per read (two vectors per pair cost 50 ms per block of 262 144 pairs in malloc/free alone).
struct RestStore {
  struct Ref { uint32_t arena, len; uint64_t at; };
  std::vector<Ref> ref;                       // per read id (2 * pair + mate)
  std::vector<std::vector<uint8_t>> arena;
  size_t used = 0;                            // arenas handed out in this block
  void reset(size_t nreads) { ref.assign(nreads, Ref{0, 0, 0}); used = 0; }
  size_t open_pass(int nthreads) {            // -> first arena of the pass; call before its threads start
    const size_t first = used;
    used += (size_t)nthreads;
    if (arena.size() < used) arena.resize(used);
    for (size_t a = first; a < used; a++) arena[a].clear();
    return first;
  }
  const uint8_t *data(size_t id) const { return ref[id].len ? arena[ref[id].arena].data() + ref[id].at : nullptr; }
  size_t size(size_t id) const { return ref[id].len; }
  void clear(size_t id) { ref[id].len = 0; }
  void put(size_t id, size_t a, const smgpost::Table &tb) {
    std::vector<uint8_t> &buf = arena[a];
    const size_t at = buf.size();
    tb.pack(buf, true);
    ref[id] = Ref{(uint32_t)a, (uint32_t)(buf.size() - at), (uint64_t)at};
  }
};

struct PairBlock {
  RestStore packed;                           // 2 * npairs tables at rest
  std::vector<PairPlan> plan;
  std::vector<uint32_t> nrounds;              // mapping calls per round kind, for the caller's statistics
  std::string error;
  uint32_t npairs = 0;
  // host wall time of the work between the rounds [ms]: passes after round A (with the intervals), B, C, D; the proper-pair
  // probe; the decisions behind round C
  enum { H_AFTER_A, H_AFTER_B, H_PROBE, H_AFTER_C, H_PLAN_D, H_AFTER_D, H_NUM };
  double host_ms[H_NUM] = {0, 0, 0, 0, 0, 0};
  struct Clock { std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
                 double lap() { const auto n = std::chrono::steady_clock::now(); const double d = std::chrono::duration<double, std::milli>(n - t).count(); t = n; return d; } };

  template <class Fn> static void spread(uint32_t n, int nthreads, Fn fn) {      // fn(lo, hi, thread)
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > n / 64 + 1) nthreads = (int)(n / 64 + 1);
    if (nthreads == 1) { fn(0u, n, 0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(fn, (uint32_t)((uint64_t)n * t / nthreads), (uint32_t)((uint64_t)n * (t + 1) / nthreads), t);
    for (std::thread &x : th) x.join();
  }
  static uint32_t len_of(const BlockInput &in, uint32_t id) { return (uint32_t)(in.off[id & 1][(id >> 1) + 1] - in.off[id & 1][id >> 1]); }

  // a round's results into the tables of its reads; `after(slot, pair, table)` runs per call with the settled table
  template <class After> bool take(const Round &rd, const smaltgpu_batch_out &o, const BlockInput &in, const BlockParams &bp, After after) {
    if (o.nreads != rd.n) { error = "a mapping round returned a different number of reads"; return false; }
    std::vector<std::string> bad((size_t)(bp.nthreads < 1 ? 1 : bp.nthreads));
    const smgpost::Reference ref{bp.sop, bp.nseq, bp.packed_host};
    const smgpost::Penalties pen{bp.map.match, bp.map.mismatch, bp.map.gap_init, bp.map.gap_ext};
    const size_t arena0 = packed.open_pass(bp.nthreads < 1 ? 1 : bp.nthreads);
    spread(rd.n, bp.nthreads, [&](uint32_t lo, uint32_t hi, int t) {
      Table tb;
      char msg[256];
      for (uint32_t i = lo; i < hi; i++) {
        const uint32_t id = rd.ids[i], w = id & 1, pr = id >> 1;
        const smaltgpu_readstat &st = o.stat[i];
        if (st.errcode) { snprintf(msg, sizeof(msg), "pair %u, mate %u, round %d: the mapping call failed on the device (code %d, site %d)", pr, w + 1, (int)rd.kind, st.errcode, st.errsite); if (bad[(size_t)t].empty()) bad[(size_t)t] = msg; continue; }
        tb.unpack(packed.data(id), packed.size(id));
        tb.n_ali_done = st.n_ali_done; tb.n_ali_tot = st.n_ali_tot; tb.n_hits_used = st.n_hits_used; tb.n_hits_tot = st.n_hits_tot;     // rmap.c:1337
        tb.take_call(o.res + o.res_off[i], (uint32_t)(o.res_off[i + 1] - o.res_off[i]), o.diffstr, st.swatscor_max, st.swatscor_2ndmax);
        if (st.max1scor >= 1) {                                        // a call without a score-pass hit returns before the pass (rmap.c:1376)
          smgpost::Read r;
          r.len = len_of(in, id);
          r.bases = in.bases[w] ? in.bases[w] + in.off[w][pr] : nullptr;
          r.quals = in.quals[w] ? in.quals[w] + in.off[w][pr] : nullptr;
          const smgpost::Outcome oc = tb.settle(ref, r, bp.packed_host && r.bases ? &pen : nullptr);
          if (oc != smgpost::DONE) {
            snprintf(msg, sizeof(msg), "pair %u, mate %u: %s", pr, w + 1, oc == smgpost::WANTS_REFERENCE ? "an alignment crosses reference sequences and no host copy of the reference was given" : tb.why);
            if (bad[(size_t)t].empty()) bad[(size_t)t] = msg;
            continue;
          }
        }
        if (!after(i, pr, tb, t)) { snprintf(msg, sizeof(msg), "pair %u, mate %u: alignment set is inconsistent", pr, w + 1); if (bad[(size_t)t].empty()) bad[(size_t)t] = msg; }
        packed.put(id, arena0 + (size_t)t, tb);
      }
    });
    for (const std::string &b : bad) if (!b.empty()) { error = b; return false; }
    return true;
  }

  template <class Exec> bool run(Exec &exec, const BlockInput &in, const BlockParams &bp) {
    const uint32_t n = in.npairs;
    npairs = n;
    error.clear();
    packed.reset((size_t)2 * n);
    plan.assign(n, PairPlan());
    nrounds.assign(4, 0);
    for (double &h : host_ms) h = 0;
    Clock ck;
    if (!n) return true;
    const int nt = bp.nthreads < 1 ? 1 : bp.nthreads;
    std::vector<uint32_t> ids((size_t)2 * n), hits((size_t)2 * n, 0);
    smaltgpu_batch_out o;

    // which mate first: k-mer hit totals of all reads (rmap.c:1866-1905)
    for (uint32_t i = 0; i < 2 * n; i++) ids[i] = i;
    if (!exec.totals(ids.data(), 2 * n, hits.data(), error)) return false;
    for (uint32_t p = 0; p < n; p++) plan_start(plan[p], len_of(in, 2 * p), len_of(in, 2 * p + 1), hits[2 * p], hits[2 * p + 1], bp.k);

    // per-thread interval lists of a round, stitched in call order afterwards
    std::vector<std::vector<smaltgpu_interval>> iv_part((size_t)nt);
    std::vector<uint32_t> iv_count;
    std::vector<uint64_t> iv_off;
    std::vector<smaltgpu_interval> iv;
    std::vector<int32_t> prev_max, min_score;
    auto stitch = [&](uint32_t ncalls) {              // iv_count per call + the parts in thread order (threads own ascending call ranges) -> iv_off, iv
      iv_off.assign((size_t)ncalls + 1, 0);
      for (uint32_t i = 0; i < ncalls; i++) iv_off[(size_t)i + 1] = iv_off[i] + iv_count[i];
      iv.clear();
      for (const auto &part : iv_part) iv.insert(iv.end(), part.begin(), part.end());
      if (iv.empty()) iv.resize(1);
    };
    std::vector<Interval> tmp_iv;

    // ---- round A: the first mate ----
    uint32_t na = 0;
    for (uint32_t p = 0; p < n; p++) if (!plan[p].idle && !plan[p].lone) ids[na++] = 2 * p + plan[p].first;
    std::vector<uint32_t> ids_b(na);
    if (na) {
      Round rd{ROUND_PLAIN, ids.data(), na, nullptr, nullptr, nullptr, nullptr};
      if (!exec.map(rd, &o, error)) return false;
      ck.lap();
      nrounds[ROUND_PLAIN] += na;
      iv_count.assign(na, 0);
      for (auto &part : iv_part) part.clear();
      std::vector<std::vector<Interval>> scratch((size_t)nt);
      if (!take(rd, o, in, bp, [&](uint32_t i, uint32_t p, Table &tb, int t) {
            plan_after_a(plan[p], tb);
            const uint32_t me = 2 * p + plan[p].first, other = me ^ 1u;
            if (!search_intervals(scratch[(size_t)t], tb, len_of(in, me), len_of(in, other), bp.d_min, bp.d_max, bp.k, bp.sop, bp.nseq)) return false;
            iv_count[i] = (uint32_t)scratch[(size_t)t].size();
            for (const Interval &v : scratch[(size_t)t]) iv_part[(size_t)t].push_back(smaltgpu_interval{v.seq, v.lo, v.hi});
            return true;
          })) return false;
      // ---- round B: the second mate inside the intervals of the first ----
      stitch(na);
      host_ms[H_AFTER_A] += ck.lap();
      for (uint32_t i = 0; i < na; i++) ids_b[i] = ids[i] ^ 1u;
      Round rb{ROUND_RESTRICTED, ids_b.data(), na, iv_off.data(), iv.data(), nullptr, nullptr};
      if (!exec.map(rb, &o, error)) return false;
      ck.lap();
      nrounds[ROUND_RESTRICTED] += na;
      if (!take(rb, o, in, bp, [&](uint32_t, uint32_t, Table &, int) { return true; })) return false;
      host_ms[H_AFTER_B] += ck.lap();
      // ---- proper pairs so far; who needs the unrestricted round ----
      std::vector<int> broken((size_t)nt, 0);
      spread(na, nt, [&](uint32_t lo, uint32_t hi, int t) {
        Table A, B;
        Probe probe;
        for (uint32_t i = lo; i < hi; i++) {
          const uint32_t p = ids_b[i] >> 1;
          A.unpack(packed.data(2 * p), packed.size(2 * p));
          B.unpack(packed.data(2 * p + 1), packed.size(2 * p + 1));
          if (plan_after_b(plan[p], probe, A, B, len_of(in, 2 * p), len_of(in, 2 * p + 1), bp.d_min, bp.d_max, bp.lib, bp.every_pair) < 0) { broken[(size_t)t] = 1; continue; }
          if (plan[p].wants_c && plan[p].proper_found < 1) packed.clear(2 * p + (plan[p].first ^ 1u));
        }
      });
      for (int b : broken) if (b) { error = "a pair's alignment sets are not in order for pairing, or the insert range is empty"; return false; }
      host_ms[H_PROBE] += ck.lap();
    }

    // ---- round C: the second mate without restriction; lone mates join here ----
    uint32_t nc = 0;
    for (uint32_t p = 0; p < n; p++) {
      const PairPlan &pl = plan[p];
      if (pl.idle || !(pl.wants_c || pl.lone)) continue;
      ids[nc++] = 2 * p + (pl.lone ? pl.lone_which : (uint8_t)(pl.first ^ 1u));
    }
    if (nc) {
      prev_max.assign((size_t)2 * nc, 0);
      spread(nc, nt, [&](uint32_t lo, uint32_t hi, int) {
        Table tb;
        for (uint32_t i = lo; i < hi; i++) { tb.unpack(packed.data(ids[i]), packed.size(ids[i])); prev_max[2 * (size_t)i] = tb.score_max; prev_max[2 * (size_t)i + 1] = tb.score_2nd; }
      });
      Round rc{ROUND_APPEND, ids.data(), nc, nullptr, nullptr, nullptr, prev_max.data()};
      if (!exec.map(rc, &o, error)) return false;
      ck.lap();
      nrounds[ROUND_APPEND] += nc;
      if (!take(rc, o, in, bp, [&](uint32_t, uint32_t, Table &, int) { return true; })) return false;
      host_ms[H_AFTER_C] += ck.lap();
      // ---- who maps the first mate again, inside the intervals of the second one's results ----
      iv_count.assign(nc, 0);
      for (auto &part : iv_part) part.clear();
      min_score.assign(nc, 0);
      std::vector<int32_t> pm((size_t)2 * nc, 0);
      std::vector<int> broken((size_t)nt, 0);
      spread(nc, nt, [&](uint32_t lo, uint32_t hi, int t) {
        Table A, B;
        std::vector<Interval> mine;
        for (uint32_t i = lo; i < hi; i++) {
          const uint32_t p = ids[i] >> 1;
          PairPlan &pl = plan[p];
          if (pl.lone) continue;
          A.unpack(packed.data(2 * p), packed.size(2 * p));
          B.unpack(packed.data(2 * p + 1), packed.size(2 * p + 1));
          plan_after_c(pl, A, B, len_of(in, 2 * p), len_of(in, 2 * p + 1), bp.k);
          if (!pl.wants_d) continue;
          const Table &first = pl.first ? B : A, &second = pl.first ? A : B;
          const uint32_t id1 = 2 * p + pl.first, id2 = id1 ^ 1u;
          if (!search_intervals(mine, second, len_of(in, id2), len_of(in, id1), bp.d_min, bp.d_max, bp.k, bp.sop, bp.nseq)) { broken[(size_t)t] = 1; continue; }
          iv_count[i] = (uint32_t)mine.size();
          for (const Interval &v : mine) iv_part[(size_t)t].push_back(smaltgpu_interval{v.seq, v.lo, v.hi});
          min_score[i] = pl.threshold_d;
          pm[2 * (size_t)i] = first.score_max; pm[2 * (size_t)i + 1] = first.score_2nd;
        }
      });
      for (int b : broken) if (b) { error = "a pair's alignment set is not in order for the search intervals"; return false; }
      // ---- round D: compact the calls that take part ----
      uint32_t nd = 0;
      std::vector<uint32_t> ids_d;
      std::vector<uint32_t> cnt_d;
      std::vector<int32_t> ms_d, pm_d;
      for (uint32_t i = 0; i < nc; i++) {
        const PairPlan &pl = plan[ids[i] >> 1];
        if (pl.lone || !pl.wants_d) continue;
        ids_d.push_back(2 * (ids[i] >> 1) + pl.first); cnt_d.push_back(iv_count[i]); ms_d.push_back(min_score[i]); pm_d.push_back(pm[2 * (size_t)i]); pm_d.push_back(pm[2 * (size_t)i + 1]);
        nd++;
      }
      if (nd) {
        iv_count = cnt_d;
        stitch(nd);
        host_ms[H_PLAN_D] += ck.lap();
        Round rdd{ROUND_FINE, ids_d.data(), nd, iv_off.data(), iv.data(), ms_d.data(), pm_d.data()};
        if (!exec.map(rdd, &o, error)) return false;
        ck.lap();
        nrounds[ROUND_FINE] += nd;
        if (!take(rdd, o, in, bp, [&](uint32_t, uint32_t, Table &, int) { return true; })) return false;
        host_ms[H_AFTER_D] += ck.lap();
      }
    }

    // ---- split reads: a second call for the read and for the mate, k-mer words from the stretch the best alignment of the first read
    //      segment leaves uncovered (mapSecondary, rmap.c:1435-1505; rmapPair calls it for both sets ahead of the pairing, :2073-2097)
    if (bp.split) {
      std::vector<uint8_t> wants((size_t)2 * n, 0);
      std::vector<uint32_t> stretch((size_t)4 * n, 0);
      std::vector<int32_t> pm((size_t)4 * n, 0);
      spread(2 * n, nt, [&](uint32_t lo, uint32_t hi, int) {
        Table tb;
        for (uint32_t id = lo; id < hi; id++) {
          if (plan[id >> 1].idle || !packed.size(id)) continue;
          tb.unpack(packed.data(id), packed.size(id));
          if (tb.by_score.empty() || !(tb.set_bits & smgpost::SET_SEGMENTED) || tb.segment_begin.size() < 2 || tb.segment_begin[1] <= tb.segment_begin[0]) continue;
          const int32_t top = tb.by_segment[(size_t)tb.segment_begin[0]];
          const uint32_t lo_q = tb.q_lo[(size_t)top], hi_q = tb.q_hi[(size_t)top], qlen = len_of(in, id);
          if (hi_q > qlen || lo_q > hi_q || qlen < (uint32_t)bp.k) continue;
          uint32_t a, b;
          if ((uint64_t)lo_q + hi_q > qlen) { a = 0; b = lo_q > 1 ? lo_q - 2 : 0; } else { a = hi_q; b = qlen - 1; }
          if ((uint64_t)a + (uint32_t)bp.k + (uint32_t)bp.s > (uint64_t)b + 1) continue;
          wants[id] = 1; stretch[2 * (size_t)id] = a; stretch[2 * (size_t)id + 1] = b;
          pm[2 * (size_t)id] = tb.score_max; pm[2 * (size_t)id + 1] = tb.score_2nd;
        }
      });
      std::vector<uint32_t> ids_e, range_e;
      std::vector<int32_t> pm_e;
      for (uint32_t w = 0; w < 2; w++) {              // the reads, then the mates: a round holds at most one call per pair (the mapper's batch size)
      ids_e.clear(); range_e.clear(); pm_e.clear();
      for (uint32_t id = w; id < 2 * n; id += 2) if (wants[id]) {
        ids_e.push_back(id); range_e.push_back(stretch[2 * (size_t)id]); range_e.push_back(stretch[2 * (size_t)id + 1]);
        pm_e.push_back(pm[2 * (size_t)id]); pm_e.push_back(pm[2 * (size_t)id + 1]);
      }
      if (!ids_e.empty()) {
        Round re{ROUND_APPEND, ids_e.data(), (uint32_t)ids_e.size(), nullptr, nullptr, nullptr, pm_e.data(), range_e.data()};
        ck.lap();
        if (!exec.map(re, &o, error)) return false;
        ck.lap();
        nrounds[ROUND_APPEND] += (uint32_t)ids_e.size();
        if (!take(re, o, in, bp, [&](uint32_t, uint32_t, Table &, int) { return true; })) return false;
        host_ms[H_AFTER_C] += ck.lap();
      }
      }
    }
    return true;
  }
};

}  // namespace smgpairs

// the C ABI's handle on a mapped block (include/smaltgpu.h)
struct smaltgpu_pairs {
  smgpairs::PairBlock blk;
  std::vector<smaltgpu_pair_info> info;
  uint64_t calls[4] = {0, 0, 0, 0};
  double round_ms[4] = {0, 0, 0, 0};
  double totals_ms = 0;                       // host wall time of the hit-totals batches
  double wall_ms = 0;                         // host wall time of the whole call
  double kernel_ms[5][16];                    // per round (4 = hit totals) and kernel: device time of the block
  uint64_t work[5][32];
};
#endif
