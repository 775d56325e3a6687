// smg_pairs.hpp -- read pairs above the mapping calls: what the reference does in resultpairs.c, in the
// decisions of rmapPair between its mapSingleRead calls (rmap.c:1744-2112) and in resultSetAddPairToReport
// (resultpairs.c:1222) -- stated here as rules over two alignment tables (smg_post.hpp), the read's (A) and the mate's (B).
//
//   LAYOUT      two alignments, one of each mate, have a template length (outermost start to outermost end, negative when
//               the mate's alignment starts first) and an orientation; a library type says which orientation is the
//               expected one, the insert range [d_lo, d_hi] which lengths are (resultCalcInsertSize results.c:992,
//               testProperPair resultpairs.c:140).
//   PROBE       after the second mate has been mapped inside the search intervals of the first: is there any alignment
//               pair on opposite strands whose anchors lie an insert apart (resultSetFindProperPairs, resultpairs.c:1162)?
//               Only the answer (and whether the probe ran into its ceiling) is used.
//   INTERVALS   where the other mate may lie, given the best alignments of one mate (setupInterValFromResultSet
//               rmap.c:354 + interValPrune interval.c:121).
//   ROUNDS      which of a pair's mates is mapped when: see PairPlan below.
//   JOIN        all pairings of the top score classes of both mates (resultSetFindPairs, resultpairs.c:1116).
//   CHOICE      a weight per pairing from the two mapping probabilities and a prior for the layout; the heaviest pairing
//               is reported with mapping qualities from the weight its alignments hold among all pairings; when it holds
//               no more than 60 % the pair is ambiguous and -r decides what is printed (scorePairsSimple resultpairs.c:830,
//               resultSetAddPairToReport :1222).
// Walks over a table go segment by segment in score order (resultSetDo, results.c:2184): a visitor may end the current
// segment or the whole walk.  Several rules only hold in that order -- in particular the PROBE's interval cursor, which
// only moves forward and starts over when it has run off the end -- so the order is part of the rule.
#ifndef SMG_PAIRS_HPP
#define SMG_PAIRS_HPP
#include <limits.h>
#include "smg_post.hpp"

namespace smgpairs {

using smgpost::Table;

// pair state bits; values are ABI (RSLTPAIRFLG_*, resultpairs.h:52-66)
enum : uint8_t { PAIR_IS_PAIR = 0x01, PAIR_MATE_FIRST = 0x02, PAIR_READ_RESTRICTED = 0x04, PAIR_MATE_RESTRICTED = 0x08 };
enum Library { LIB_PAIRED_END = 1, LIB_MATE_PAIR = 2, LIB_SAME_STRAND = 3, LIB_ANY = 4 };       // RSLTPAIRLIB_*, resultpairs.h:68-82
// what is known about a pairing (MAP_FLAGS, resultpairs.c:43-54)
enum : uint8_t { PM_PAIRED = 0x01, PM_SAME_SEQUENCE = 0x02, PM_ORIENTED = 0x04, PM_IN_RANGE = 0x08, PM_READ_AMBIGUOUS = 0x20, PM_MATE_AMBIGUOUS = 0x40 };
// output policy (RESULTFLG_*, results.h:55-63)
enum : uint32_t { OUT_BEST = 0x01, OUT_ONE_ONLY = 0x02, OUT_DRAW = 0x08 };
enum { PROBE_CEILING = 1028, JOIN_CEILING = 8192, QUALITY_CONFIDENT_FIRST = 20, INTERVAL_SLACK_PERCENT = 30 };   // rmap.c:66-74, resultpairs.c:39

enum Walk { GO_ON = 0, NEXT_SEGMENT = 1, STOP = 2 };
template <class Visit> inline bool walk(const Table &t, Visit visit) {     // false: the table is not in walking order
  if (t.by_score.empty()) return true;
  if (!(t.set_bits & smgpost::SET_ORDERED) || !(t.set_bits & smgpost::SET_SEGMENTED)) return false;
  for (int g = 0; g < t.nsegments; g++)
    for (int i = t.segment_begin[(size_t)g]; i < t.segment_begin[(size_t)g + 1]; i++) {
      const int v = visit((uint32_t)t.by_segment[(size_t)i]);
      if (v == STOP) return true;
      if (v == NEXT_SEGMENT) break;
    }
  return true;
}

// ---- LAYOUT -------------------------------------------------------------------------------------------------------
struct Layout { int tlen; bool read_reversed, mate_reversed, mate_leftmost, same_sequence; };
inline Layout layout_of(const Table &A, uint32_t a, const Table &B, uint32_t b) {
  Layout y;
  y.read_reversed = (A.bits[a] & smgpost::REVERSED) != 0;
  y.mate_reversed = (B.bits[b] & smgpost::REVERSED) != 0;
  y.mate_leftmost = B.r_lo[b] < A.r_lo[a];
  y.same_sequence = A.seq[a] >= 0 && B.seq[b] >= 0 && A.seq[a] == B.seq[b];
  const uint64_t left = std::min(A.r_lo[a], B.r_lo[b]), right = std::max(A.r_hi[a], B.r_hi[b]);
  y.tlen = (left + INT_MAX > right || left < right + INT_MAX) ? (int)(right - left + 1) : 0;      // results.c:1013
  if (y.mate_leftmost) y.tlen = -y.tlen;
  return y;
}
// -> PM_IN_RANGE | PM_ORIENTED as they apply.  Expected orientations, leftmost alignment named first: paired-end forward
// then reverse, mate-pair reverse then forward, same-strand read ahead of mate on their common strand.
inline uint8_t judge_layout(const Layout &y, int d_lo, int d_hi, int lib) {
  const bool mate_first = y.tlen < 0;
  uint8_t v = 0;
  if (mate_first ? (y.tlen <= -d_lo && y.tlen >= -d_hi) : (y.tlen >= d_lo && y.tlen <= d_hi)) v |= PM_IN_RANGE;
  if (lib == LIB_ANY) return v | PM_ORIENTED;
  if (mate_first != y.mate_leftmost) return v;                         // a template of length 0: never oriented
  const bool left_rev = mate_first ? y.mate_reversed : y.read_reversed, right_rev = mate_first ? y.read_reversed : y.mate_reversed;
  bool ok = false;
  if (lib == LIB_PAIRED_END) ok = !left_rev && right_rev;
  else if (lib == LIB_MATE_PAIR) ok = left_rev && !right_rev;
  else if (lib == LIB_SAME_STRAND) ok = y.read_reversed == y.mate_reversed && y.read_reversed == mate_first;
  return ok ? (uint8_t)(v | PM_ORIENTED) : v;
}

// forward-strand offset of the read's first sequenced base implied by an alignment (32-bit, wraps like the reference's)
inline uint32_t anchor_of(const Table &t, uint32_t r) {
  return (t.bits[r] & smgpost::REVERSED) ? (uint32_t)t.r_hi[r] + t.q_lo[r] - 2u : (uint32_t)t.r_lo[r] - t.q_lo[r];
}

// ---- PROBE --------------------------------------------------------------------------------------------------------
struct Probe {
  struct Window { uint64_t key; uint32_t lo, hi; int64_t seq; bool reversed; uint32_t row; };
  std::vector<Window> windows;
  int found = 0;
  bool hit_ceiling = false;
  // -1: broken (table not walkable, or d_lo > d_hi after clamping at 0), else the number of proper pairs found
  int run(const Table &A, const Table &B, int d_min, int d_max, int lib) {
    found = 0; hit_ceiling = false;
    windows.clear();
    if (A.by_score.empty() || B.by_score.empty()) return 0;
    // windows around the best alignments of A: an insert below and an insert above the anchor, joined when they touch
    if (A.segment_begin.size() >= 2 && A.segment_begin[1] - A.segment_begin[0] >= 1) {
      const uint32_t dl = d_min < 0 ? 0u : (uint32_t)d_min, dh = d_max < 0 ? 0u : (uint32_t)d_max;
      if (dl > dh) return -1;
      if (!walk(A, [&](uint32_t r) {
            if (A.rank[r] > 0) return NEXT_SEGMENT;
            const uint32_t at = anchor_of(A, r);
            Window below, above;
            below.seq = above.seq = A.seq[r]; below.reversed = above.reversed = (A.bits[r] & smgpost::REVERSED) != 0; below.row = above.row = r;
            if (at >= dh) { below.hi = at - dl; below.lo = at - dh; } else { below.hi = at > dl ? at - dl : 0; below.lo = 0; }
            above.hi = at + dh; above.lo = at + dl;
            if (above.lo <= below.hi) { below.hi = above.hi; windows.push_back(below); }
            else { windows.push_back(below); windows.push_back(above); }
            return GO_ON;
          })) return -1;
      // by sequence, reverse strand first, lower bound; equal keys keep their order (resultpairs.c:419-437)
      for (size_t i = 0; i < windows.size(); i++) windows[i].key = (uint64_t)windows[i].seq << 33 | (uint64_t)(!windows[i].reversed) << 32 | windows[i].lo;
      std::stable_sort(windows.begin(), windows.end(), [](const Window &x, const Window &y) { return x.key < y.key; });
    }
    const int floor_score = B.score_2nd > 0 ? B.score_2nd : B.score_max;              // resultpairs.c:1185-1189
    const int d_lo = std::min(d_min, d_max), d_hi = std::max(d_min, d_max);
    if (floor_score > B.score_max) return 0;
    const size_t nw = windows.size();
    size_t cursor = 0;
    if (!walk(B, [&](uint32_t r) {
          if (B.rank[r] > 0 || B.score[r] < floor_score) return NEXT_SEGMENT;
          if (cursor >= nw) cursor = 0;
          const bool rev = (B.bits[r] & smgpost::REVERSED) != 0;
          const uint32_t at = anchor_of(B, r);
          for (; cursor < nw; cursor++) {
            const Window &w = windows[cursor];
            if (B.seq[r] < w.seq) break;
            if (B.seq[r] > w.seq || rev == w.reversed || at > w.hi) continue;
            if (at < w.lo) break;
            const Layout y = layout_of(A, w.row, B, r);
            const int len = y.tlen < 0 ? -y.tlen : y.tlen;
            if (len >= d_lo && len <= d_hi) found++;
            if (found >= PROBE_CEILING) { hit_ceiling = true; return STOP; }
          }
          return GO_ON;
        })) return -1;
    return found;
  }
};

// ---- INTERVALS ----------------------------------------------------------------------------------------------------
struct Interval { int32_t seq; uint32_t lo, hi; };            // 0-based, inclusive, inside sequence seq
// `from` = the mate that is mapped, `other_len` = length of the mate to be searched; k = word length of the index.
// -> false when an alignment of the best class is not usable (not live, empty, without a sequence)
inline bool search_intervals(std::vector<Interval> &out, const Table &from, uint32_t from_len, uint32_t other_len, int d_min, int d_max, int k,
                             const uint64_t *sop, int64_t nseq) {
  out.clear();
  if (d_min > d_max) return false;
  int nbest;
  from.score_classes(&nbest, nullptr);
  const int64_t slack = ((int64_t)other_len * INTERVAL_SLACK_PERCENT) / 100;
  if (nbest > 0 && !(from.set_bits & smgpost::SET_ORDERED)) return false;
  for (int i = 0; i < nbest; i++) {
    const uint32_t r = (uint32_t)from.by_score[(size_t)i];
    const uint32_t rs = (uint32_t)from.r_lo[r], re = (uint32_t)from.r_hi[r];
    if (!(from.bits[r] & smgpost::LIVE) || re <= rs || from.seq[r] < 0 || from.seq[r] >= nseq) return false;
    const int64_t seqlen = (int64_t)(uint32_t)(sop[from.seq[r] + 1] - sop[from.seq[r]]);
    auto clamp = [&](int64_t v) { if (v >= seqlen) v = seqlen - 1; if (v < 1) v = 0; return v; };
    // towards smaller coordinates, measured from the alignment's end; towards larger ones, from its start (rmap.c:411-431)
    int64_t lo = clamp((int64_t)re + from_len - from.q_hi[r] - d_max);
    int64_t hi = clamp((int64_t)re + from_len + other_len + slack - from.q_hi[r] - d_min - k);
    if (lo <= hi) out.push_back(Interval{(int32_t)from.seq[r], (uint32_t)lo, (uint32_t)hi});
    lo = clamp((int64_t)rs - from.q_lo[r] + d_min - other_len);
    hi = clamp((int64_t)rs - from.q_lo[r] + d_max - k + slack);
    if (lo <= hi) out.push_back(Interval{(int32_t)from.seq[r], (uint32_t)lo, (uint32_t)hi});
  }
  if (out.empty()) return true;
  std::sort(out.begin(), out.end(), [](const Interval &x, const Interval &y) { return x.seq != y.seq ? x.seq < y.seq : (x.lo != y.lo ? x.lo < y.lo : x.hi < y.hi); });
  size_t w = 0;
  for (size_t j = 1; j < out.size(); j++) {
    if (out[j].seq == out[w].seq && out[j].lo <= out[w].hi) { if (out[j].hi > out[w].hi) out[w].hi = out[j].hi; }
    else out[++w] = out[j];
  }
  out.resize(w + 1);
  return true;
}

// is a score at least 80 % of another, scaled by the read lengths?  float arithmetic as in the reference (rmap.c:178-185)
inline bool score_holds_up(int score, uint32_t len, int against, uint32_t against_len) {
  static const float FRACTION = 0.8f;
  return score >= (unsigned)against * len * FRACTION / against_len;
}

// ---- ROUNDS -------------------------------------------------------------------------------------------------------
// rmapPair as a plan per pair.  `first` is the mate with fewer k-mer hits (the read on a tie), `second` the other.
//   round A   first, unrestricted.
//   round B   second, seeded inside the intervals round A implies.
//   after B   PROBE; the pair is settled (second counts as restricted) when a proper pair exists, first's quality is
//             at least 20 and second's restricted score holds up against first's.  Otherwise round C, into a blank set
//             when no proper pair exists, else on top of what round B found.
//   round C   second, unrestricted.
//   after C   when second's quality exceeds 20, or its score beats its restricted score or first's score: round D.
//   round D   first again, inside the intervals of second's results, seeded against an index of those intervals built on
//             the fly, with first's second-best score as threshold -- provided first is at least one word long.
// A pair with one mate shorter than a word maps the long mate alone (which is what the reference's rounds amount to,
// rmap.c:1836-1864 and the empty rounds behind); a pair of two short mates does nothing.
struct PairPlan {
  uint8_t state = PAIR_IS_PAIR;
  uint8_t first = 0;                 // 0 read, 1 mate
  bool idle = false, lone = false;
  uint8_t lone_which = 0;
  bool wants_c = false, wants_d = false;
  int first_quality = 0, first_score = 0, second_restricted_score = 0, proper_found = 0, threshold_d = 0;
};

inline void plan_start(PairPlan &p, uint32_t read_len, uint32_t mate_len, uint32_t read_hits, uint32_t mate_hits, int k) {
  p = PairPlan();
  if (read_len < (uint32_t)k && mate_len < (uint32_t)k) { p.idle = true; return; }
  if (read_len < (uint32_t)k || mate_len < (uint32_t)k) { p.lone = true; p.lone_which = read_len < (uint32_t)k ? 1 : 0; }
  if (read_hits > mate_hits) { p.state |= PAIR_MATE_FIRST; p.first = 1; }
}
inline void plan_after_a(PairPlan &p, const Table &first) { p.first_quality = first.top_quality(&p.first_score); }
// after round B.  -1: broken
inline int plan_after_b(PairPlan &p, Probe &probe, Table &A, Table &B, uint32_t read_len, uint32_t mate_len, int d_min, int d_max, int lib, bool all_pairs) {
  Table &second = p.first ? A : B;
  const uint32_t first_len = p.first ? mate_len : read_len, second_len = p.first ? read_len : mate_len;
  const int n = probe.run(A, B, d_min, d_max, lib);
  if (n < 0) return -1;
  p.proper_found = n;
  (void)second.top_quality(&p.second_restricted_score);
  if (all_pairs || n < 1 || p.first_quality < QUALITY_CONFIDENT_FIRST || !score_holds_up(p.second_restricted_score, second_len, p.first_score, first_len)) {
    p.wants_c = true;
    if (n < 1) second.clear();
  } else p.state |= p.first == 0 ? PAIR_MATE_RESTRICTED : PAIR_READ_RESTRICTED;
  return 0;
}
inline void plan_after_c(PairPlan &p, const Table &A, const Table &B, uint32_t read_len, uint32_t mate_len, int k) {
  const Table &first = p.first ? B : A, &second = p.first ? A : B;
  const uint32_t first_len = p.first ? mate_len : read_len;
  int second_score = 0;
  const int q = second.top_quality(&second_score);
  p.wants_d = false;
  if (!(q > QUALITY_CONFIDENT_FIRST || second_score > p.second_restricted_score || second_score > p.first_score)) return;
  p.threshold_d = first.score_2nd;
  if ((uint32_t)k > first_len) return;
  p.wants_d = true;
}

// ---- JOIN ---------------------------------------------------------------------------------------------------------
struct MatePair { uint32_t a, b; int tlen; uint8_t know; double weight; };
struct Join {
  std::vector<MatePair> pairs;
  int n_proper = 0, n_in_range = 0;
  bool run(const Table &A, const Table &B, uint8_t state, int lib, int d_min, int d_max) {
    pairs.clear(); n_proper = n_in_range = 0;
    const int d_lo = std::min(d_min, d_max), d_hi = std::max(d_min, d_max);
    int depth_a, depth_b;
    const bool one_a = A.rank_depth(&depth_a), one_b = B.rank_depth(&depth_b);
    if ((state & PAIR_MATE_RESTRICTED) && one_a) depth_a = 0;
    else if ((state & PAIR_READ_RESTRICTED) && one_b) depth_b = 0;
    bool ok = true;
    const bool walked = walk(A, [&](uint32_t a) {
      if (A.rank[a] > depth_a) return NEXT_SEGMENT;
      // once the list is full every further alignment of the read still contributes its first pairing (resultpairs.c:395-399)
      ok = walk(B, [&](uint32_t b) {
        if (B.rank[b] > depth_b) return NEXT_SEGMENT;
        MatePair m;
        m.a = a; m.b = b; m.weight = 0.0; m.know = PM_PAIRED;
        const Layout y = layout_of(A, a, B, b);
        m.tlen = y.tlen;
        if (y.same_sequence) {
          m.know |= judge_layout(y, d_lo, d_hi, lib);
          if (m.know & PM_IN_RANGE) { n_in_range++; if (m.know & PM_ORIENTED) n_proper++; }
          m.know |= PM_SAME_SEQUENCE;
        }
        pairs.push_back(m);
        return pairs.size() >= (size_t)JOIN_CEILING ? STOP : GO_ON;
      });
      return ok ? GO_ON : STOP;
    });
    return walked && ok;
  }
};

// ---- CHOICE -------------------------------------------------------------------------------------------------------
// uniform numbers in [0, 1) for the random choices, drawn by the caller in pair order (the reference draws from the C
// library's one drand48 sequence as it goes through the pairs); counting mode tells how many a pair needs
struct Draws {
  const double *values = nullptr;
  int have = 0, used = 0;
  double next() { const double v = (values && used < have) ? values[used] : 0.0; used++; return v; }
};

inline int quality_of_draw(int among) {                        // results.c:214-230
  if (among < 1 || among > 9) return 0;
  if (among == 1) return 4;
  int q = (int)(-10 * log10(((double)(among - 1)) / among) + .499);
  return q > 3 ? 3 : (q < 0 ? 0 : q);
}
// the alignment a mate is represented by when it stands alone: the best one; *ambiguous when it shares its score (or has
// quality 0); with OUT_DRAW an ambiguous one is drawn from the top class (resultSetGetTopResult, results.c:2499-2523)
inline int lone_top(Table &t, bool *ambiguous, bool draw, Draws &dr) {
  int ntop = 0;
  *ambiguous = false;
  if (t.by_score.empty()) return -1;
  const bool single = t.top_class(&ntop);
  int top = -1;
  if (ntop > 0) {
    if (single) { top = t.by_score[0]; if (t.quality[top] < 1) *ambiguous = true; }
    else *ambiguous = true;
    if (*ambiguous && draw) {
      top = t.by_score[(size_t)(short)(dr.next() * ntop)];
      t.quality[top] = quality_of_draw(ntop);
    }
  }
  return top;
}
inline int quality_of_probability(double p) {                  // results.c:292-305
  double wrong = 1.0 - p;
  if (wrong < 1E-7) wrong = 1E-7;
  const double m = -10 * log10(wrong);
  return m > 60 ? 60 : (m < 0 ? 0 : (int)(short)m);
}

struct Entry { int a, b; int quality_a, quality_b; uint8_t know; };     // one reported pairing: rows (-1: none) and what is printed with them

// -> the pairings to print, the chosen one first.  `join.pairs` is re-ordered by weight.
inline void choose(std::vector<Entry> &out, Join &join, Table &A, Table &B, uint8_t state, uint32_t policy, Draws &dr) {
  static const double TINY = 1E-7, P_DISORIENTED = 1e-4, P_OUT_OF_RANGE = 3e-3;            // resultpairs.c:134-136
  out.clear();
  std::vector<MatePair> &pr = join.pairs;
  const bool draw = (policy & OUT_DRAW) != 0;
  Entry e{-1, -1, 0, 0, 0};
  int ntied = 0;
  if (pr.empty()) {
    // no pairing (a mate without alignments): each mate by itself.  The ambiguity of the MATE lands in bit 0 of `know`
    // (resultpairs.c:877-878 hands the same byte to both calls)
    bool amb = false;
    e.a = lone_top(A, &amb, draw, dr);
    e.b = lone_top(B, &amb, draw, dr);
    e.know = amb ? 1 : 0;
  } else {
    const double p_oriented = 1.0 - P_DISORIENTED, p_inside = 1.0 - P_OUT_OF_RANGE, p_outside_all = P_DISORIENTED + p_oriented * P_OUT_OF_RANGE;
    double total = TINY, extra_a = 0.0, extra_b = 0.0;
    for (MatePair &m : pr) {
      double pa = A.prob[m.a], pb = B.prob[m.b];
      if (state & PAIR_READ_RESTRICTED) { if (pa > pb) pa = pb; }
      else if (state & PAIR_MATE_RESTRICTED) { if (pb > pa) pb = pa; }
      double prior;
      if (m.know & PM_ORIENTED) { prior = p_oriented; prior *= (m.know & PM_IN_RANGE) ? p_inside : P_OUT_OF_RANGE; }
      else prior = P_DISORIENTED;
      m.weight = pa * pb * prior;
      total += m.weight;
      // a mate with one alignment only could also be placed wrongly altogether
      if (A.bits[m.a] & smgpost::ONLY_ONE) { const double s = (1.0 - pa) * (p_outside_all) * pb; extra_b += s; total += s; }
      if (B.bits[m.b] & smgpost::ONLY_ONE) { const double s = pa * p_outside_all * (1.0 - pb); extra_a += s; total += s; }
    }
    if (total < TINY) total = TINY;
    std::stable_sort(pr.begin(), pr.end(), [](const MatePair &x, const MatePair &y) { return x.weight > y.weight; });
    ntied = 1;
    while (ntied < (int)pr.size() && !(pr[(size_t)ntied].weight + TINY < pr[0].weight)) ntied++;
    const MatePair *pick = &pr[0];
    uint8_t know = 0;
    if (pr[0].weight / total <= 0.6 && pr.size() > 1) {
      know = PM_READ_AMBIGUOUS | PM_MATE_AMBIGUOUS;
      if (draw) {
        double sum = 0.0;
        for (const MatePair &m : pr) sum += m.weight;
        const double mark = dr.next() * sum;
        sum = 0.0;
        pick = nullptr;
        for (const MatePair &m : pr) { sum += m.weight; if (sum + TINY > mark) { pick = &m; break; } }
        if (!pick) pick = &pr.back();
      } else if (policy & OUT_ONE_ONLY) pick = nullptr;
    }
    e.know = know;
    if (pick) {
      e.a = (int)pick->a; e.b = (int)pick->b; e.know |= pick->know;
      double share_a = extra_a, share_b = extra_b;
      for (const MatePair &m : pr) { if (m.a == pick->a) share_a += m.weight; if (m.b == pick->b) share_b += m.weight; }
      e.quality_a = quality_of_probability(share_a / total);
      e.quality_b = quality_of_probability(share_b / total);
    }
  }
  if (ntied > 1 && !draw && (policy & OUT_ONE_ONLY)) {
    // several equally heavy pairings and only one may be printed: a mate whose own best alignment is unambiguous is kept
    bool amb_a = false, amb_b = false;
    Draws none;
    e.a = lone_top(A, &amb_a, false, none);
    e.b = lone_top(B, &amb_b, false, none);
    if (!amb_a) { e.b = -1; e.know |= PM_MATE_AMBIGUOUS; }
    else if (!amb_b) { e.a = -1; e.know |= PM_READ_AMBIGUOUS; }
    else { e.know |= PM_READ_AMBIGUOUS | PM_MATE_AMBIGUOUS; e.a = e.b = -1; }
  }
  out.push_back(e);
  if ((e.know & (PM_READ_AMBIGUOUS | PM_MATE_AMBIGUOUS)) && !draw && !(policy & OUT_ONE_ONLY))
    for (int i = 0; i < ntied; i++) {
      const MatePair &m = pr[(size_t)i];
      if ((int)m.a != e.a || (int)m.b != e.b)
        out.push_back(Entry{(int)m.a, (int)m.b, e.quality_a, e.quality_b, (uint8_t)(m.know | (e.know & (PM_READ_AMBIGUOUS | PM_MATE_AMBIGUOUS)))});
    }
}

}  // namespace smgpairs
#endif
