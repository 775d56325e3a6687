// smg_post.hpp -- the alignment table of one read and the pass that runs over it after every mapping call
// (what the reference reaches through resultSetSortAndAssignSequence, results.c:2022-2064).
//
// Written from the rules, not from the reference's code.  The rules, in the order they apply:
//   1. PLACE   concatenated mode hands alignments over in coordinates of the whole reference; each one is given its
//              sequence and coordinates inside it.  One that runs across sequence junctions is replaced by one piece per
//              sequence, each trimmed to begin and end on a matched base and scored again (results.c:1472, :1695).
//   2. PRUNE   an alignment that lies inside another one of the same sequence and strand -- reference end, read
//              interval and score all within it -- is dropped (results.c:759-815).
//   3. ORDER   survivors by score (best first), forward strand first, sequence, start, longer read interval first;
//              equal scores share a rank (results.c:478-507, :817-834).
//   4. SEGMENT alignments are grouped by the part of the read they cover: the best alignment that has no group yet
//              founds one and takes every later one overlapping it by 80 % of the shorter interval (results.c:707-757).
//   5. QUALITY per group a PHRED-scaled mapping quality for the best alignment and probabilities for the top two score
//              classes (results.c:1143-1341 in the `results_mapscor_exp` build, :1343-1398).
// The table is column-wise: one vector per field, one entry per alignment in order of arrival, pieces behind.  It can
// live across several calls (rmapPair appends to a set, rmap.c:1976-2039): columns of rows that a later pass does not
// touch keep what the earlier pass wrote, as the reference's array does.
//
// Two places need more than the rules:
//   * Rule 2 walks the alignments ordered by (sequence, strand, start).  The reference orders them with libc qsort and
//     a predicate that, for equal (sequence, strand, start), compares the READ extent of its first argument with the
//     REFERENCE extent of its second (results.c:466-470) -- not an ordering, so the outcome for such ties is whatever
//     libc's algorithm makes of it.  Ties are rare; when one occurs the same libc qsort is asked, with a predicate of
//     equal truth table over our packed keys.  Without ties any sort yields the one possible order.
//   * Rule 5 is floating point.  Every formula appears once, with the operand order and the float constant of the
//     reference, so that the doubles come out bit for bit (compile with -ffp-contract=off).
#ifndef SMG_POST_HPP
#define SMG_POST_HPP
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace smgpost {

// per-alignment bits; the values are ABI (smaltgpu_post_result.status = the reference's RSLTFLAG_*, results.h:67-78)
enum : uint32_t { LIVE = 0x01, REVERSED = 0x04, UNPLACED = 0x08, WITHHELD = 0x10, BELOW_RELATIVE = 0x20, ONLY_ONE = 0x100, REPORTED = 0x200 };
// per-table bits (RSLTSETFLG_*, results.c:93-100)
enum : uint32_t { SET_PLACED = 0x01, SET_NUMBERED = 0x02, SET_ORDERED = 0x04, SET_SEGMENTED = 0x08, SET_QUALIFIED = 0x10 };
enum { QUALITY_TOP = 60, QUALITY_FLOOR_UNIQUE = 4, ROWS_MAX = 32767 };
enum Outcome { DONE = 0, WANTS_REFERENCE = 1, BROKEN = -1 };

// alignment string: one byte per event, op << 6 | n, n matched bases in front of the event (diffstr.h:28-76)
enum : unsigned { OP_MATCH = 0, OP_DELETE = 1, OP_INSERT = 2, OP_SUBST = 3 };
enum { RUN_FULL = 61 };                                      // a MATCH byte with n = 61 stands for 62 matched bases

struct Reference {
  const uint64_t *sop;           // nseq + 1 offsets of the sequences in the concatenated reference
  int64_t nseq;
  const uint32_t *packed;        // 10 bases per word, 3 bits each, first base in bits 29-27; may be null (no cutting then)
};
struct Read {
  const uint8_t *bases;          // ASCII; may be null (no cutting then)
  const uint8_t *quals;          // phred + 33; may be null
  uint32_t len;
};
struct Penalties { int match, mismatch, gap_open, gap_ext; };   // as smaltgpu_params carries them: +1 -2 -4 -3

// ---------------------------------------------------------------------------------------------------------------
// Cutting an alignment string to a window [lo, hi] of reference offsets (0 = first reference base of the alignment).
// One forward scan: `u`, `v` count reference and read bases passed; the piece opens at the first matched base at or
// behind `lo`; events are written as they come, and a snapshot taken at every matched base inside the window says how
// much of the output stands when the scan leaves the window -- events behind the last such base are dropped again.
// Runs are re-chunked: 62 matched bases per full MATCH byte in front of an event, and the closing byte may carry 62.
// (Equivalent to diffStrSegment, diffstr.c:1370, which finds both ends first and copies the bytes between them.)
struct Piece {
  std::vector<uint8_t> str;      // with the terminating 0
  int64_t ref_first, ref_last, read_first, read_last;     // offsets of the piece's end bases from the alignment's start
};
enum { CUT_OK = 0, CUT_EMPTY = 1, CUT_BAD = -1 };

inline void put_run_then(std::vector<uint8_t> &o, int64_t &run, unsigned op) {
  while (run > RUN_FULL) { o.push_back((uint8_t)(OP_MATCH << 6 | RUN_FULL)); run -= RUN_FULL + 1; }
  o.push_back((uint8_t)(op << 6 | (unsigned)run));
  run = 0;
}

inline int cut_window(const uint8_t *s, int64_t lo, int64_t hi, Piece &pc) {
  std::vector<uint8_t> &o = pc.str;
  o.clear();
  int64_t u = 0, v = 0, run = 0;
  bool open = false;
  size_t stand_len = 0;          // snapshot at the last matched base inside the window
  int64_t stand_run = 0;
  for (; *s; ++s) {
    const unsigned op = *s >> 6;
    const int64_t n = (*s & 63) + (op == OP_MATCH ? 1 : 0);          // matched bases of this byte: reference u .. u+n-1
    if (n > 0) {
      int64_t a = u;
      const int64_t b = u + n - 1;
      if (!open && b >= lo) {
        if (a < lo) a = lo;
        if (a > hi) return CUT_EMPTY;                                // the first matched base behind lo lies outside
        open = true;
        pc.ref_first = a;
        pc.read_first = v + (a - u);
      }
      if (open && a <= hi) {
        const int64_t last = b < hi ? b : hi;
        run += last - a + 1;
        stand_len = o.size(); stand_run = run;
        pc.ref_last = last;
        pc.read_last = v + (last - u);
      }
      u += n; v += n;
      if (open && u > hi) break;
    }
    if (op == OP_MATCH || (op == OP_SUBST && !s[1])) continue;       // no event column: a full run, or the closing byte
    if (open) put_run_then(o, run, op);
    if (op != OP_INSERT) u++;
    if (op != OP_DELETE) v++;
    if (open && u > hi) break;
  }
  if (!open) return CUT_BAD;                                          // no matched base at or behind lo at all
  o.resize(stand_len);
  run = stand_run;
  while (run > RUN_FULL + 1) { o.push_back((uint8_t)(OP_MATCH << 6 | RUN_FULL)); run -= RUN_FULL + 1; }
  o.push_back((uint8_t)(OP_SUBST << 6 | (unsigned)run));
  o.push_back(0);
  return CUT_OK;
}

// 3-bit codes: ACGT 0-3, 5 = N (scores 0 against everything), 4 = X (scores mismatch - match); score.c:138-173
inline unsigned base_code(uint8_t c) {
  switch (c | 0x20) { case 'a': return 0; case 'c': return 1; case 'g': return 2; case 't': case 'u': return 3; default: return 5; }
}
inline unsigned packed_code(const uint32_t *packed, uint64_t at) {   // as the device path reads the reference (smg_logic.hpp ref_code)
  const unsigned c = (packed[at / 10] >> (3 * (9 - (unsigned)(at % 10)))) & 7u;
  return c == 7 ? 0u : ((c == 6 || c == 4) ? 5u : c);
}
inline int pair_score(unsigned a, unsigned b, const Penalties &pen) {
  if (a > 4 || b > 4) return 0;
  if (a == 4 || b == 4) return pen.mismatch - pen.match;
  return a == b ? pen.match : pen.mismatch;
}

// Score of an alignment string laid over reference bases from `ref_at` (concatenated offset) and read bases from `read_at`
// (offset in the read as aligned, i.e. in its reverse complement for REVERSED): every aligned column scores by the
// base pair, a gap costs gap_open for its first base and gap_ext for each further one, and any aligned column closes
// an open gap (aliScoreDiffStr, alignment.c:179-225).  Returns false when the string runs off the read.
inline bool string_score(int &score, const uint8_t *s, const Reference &ref, uint64_t ref_at, const Read &rd, uint32_t read_at, bool reversed,
                         const Penalties &pen) {
  int sum = 0;
  bool in_gap = false;
  for (; *s; ++s) {
    const unsigned op = *s >> 6;
    unsigned cols = *s & 63;
    if (op == OP_MATCH || (op == OP_SUBST && s[1])) cols++;
    if (cols) in_gap = false;
    for (; cols; cols--, ref_at++, read_at++) {
      if (read_at >= rd.len) return false;
      unsigned q;
      if (reversed) { q = base_code(rd.bases[rd.len - 1 - read_at]); if (q < 4) q = 3 - q; }
      else q = base_code(rd.bases[read_at]);
      sum += pair_score(packed_code(ref.packed, ref_at), q, pen);
    }
    if (op == OP_INSERT || op == OP_DELETE) {
      sum += in_gap ? pen.gap_ext : pen.gap_open;
      in_gap = true;
      if (op == OP_INSERT) { if (++read_at > rd.len) return false; } else ref_at++;
    }
  }
  score = sum;
  return true;
}

// Base qualities summed over the substituted bases of an alignment (sumQualOverMisMatch without the unaligned flanks,
// results.c:232-285).  The walk is over read positions from q_lo on, whatever the strand, and must end on q_hi.
inline bool subst_quality_sum(int &sum, const uint8_t *quals, uint32_t read_len, uint32_t q_lo, uint32_t q_hi, const uint8_t *s) {
  if (q_hi < q_lo) return false;
  uint32_t at = q_lo ? q_lo - 1 : 0, acc = 0;
  for (; *s; ++s) {
    const unsigned op = *s >> 6;
    at += *s & 63;
    if (op == OP_DELETE) continue;
    if (op == OP_SUBST) {
      if (!s[1]) continue;
      if (at < 1 || at >= read_len || quals[at] < 33) return false;
      acc += (uint32_t)quals[at] - 33;
      if (acc > (uint32_t)INT32_MAX) return false;
    }
    at++;
  }
  if (at != q_hi) return false;
  sum = (int)acc;
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
struct Table {
  // columns, one entry per alignment
  std::vector<int32_t> score, quality;
  std::vector<uint32_t> q_lo, q_hi, bits, str_at, str_len;
  std::vector<uint64_t> r_lo, r_hi;
  std::vector<int64_t> seq;
  std::vector<double> prob;
  std::vector<int16_t> primary, segment, rank;         // `primary`: split-read link, always -1 on this path
  std::vector<uint8_t> strings;                         // the alignment strings of all rows, each with its terminating 0
  // orders over the live rows
  std::vector<int32_t> by_score, by_segment, segment_begin;
  uint32_t set_bits = 0;
  int nsegments = 0;
  // what the mapping call reported beside the alignments (rmap.c:1333-1338) and the running score maxima of the set
  int32_t n_ali_done = 0, n_ali_tot = 0, score_max = 0, score_2nd = 0;
  uint32_t n_hits_used = 0, n_hits_tot = 0;
  std::vector<uint32_t> slot_;               // take_call's view of the array's slots
  const char *why = "";                                 // set when a pass returns BROKEN

  uint32_t rows() const { return (uint32_t)score.size(); }
  void clear() {
    score.clear(); quality.clear(); q_lo.clear(); q_hi.clear(); bits.clear(); str_at.clear(); str_len.clear(); r_lo.clear(); r_hi.clear(); seq.clear();
    prob.clear(); primary.clear(); segment.clear(); rank.clear(); strings.clear(); by_score.clear(); by_segment.clear(); segment_begin.clear();
    set_bits = 0; nsegments = 0; n_ali_done = n_ali_tot = score_max = score_2nd = 0; n_hits_used = n_hits_tot = 0; why = "";
  }
  const uint8_t *str(uint32_t row) const { return strings.data() + str_at[row]; }
  uint32_t add(int32_t sc, uint32_t ql, uint32_t qh, uint64_t rl, uint64_t rh, int64_t sq, bool reversed, const uint8_t *s, uint32_t slen) {
    const uint32_t row = rows();
    score.push_back(sc); quality.push_back(0); q_lo.push_back(ql); q_hi.push_back(qh); r_lo.push_back(rl); r_hi.push_back(rh); seq.push_back(sq);
    bits.push_back(LIVE | (reversed ? (uint32_t)REVERSED : 0u) | (sq < 0 ? (uint32_t)UNPLACED : 0u));
    prob.push_back(0.0); primary.push_back(-1); segment.push_back(-1); rank.push_back(0);
    str_at.push_back((uint32_t)strings.size()); str_len.push_back(slen);
    strings.insert(strings.end(), s, s + slen);
    set_bits = 0;                                       // a table that takes a new row is unordered again (results.c:1883)
    return row;
  }
  uint32_t span_q(uint32_t row) const { return q_hi[row] - q_lo[row]; }

  // the whole pass; BROKEN leaves `why`
  Outcome settle(const Reference &ref, const Read &rd, const Penalties *pen) {
    Outcome oc = place(ref, rd, pen);
    if (oc != DONE) return oc;
    if (!prune_and_order()) return BROKEN;
    nsegments = 0;
    if (!by_score.empty()) {
      if (!group_by_read_interval()) return BROKEN;
      for (int g = 0; g < nsegments; g++) if (!qualify(g, rd)) return BROKEN;
      set_bits |= SET_QUALIFIED;
    }
    return DONE;
  }

  // ---- rule 1 ----
  Outcome place(const Reference &ref, const Read &rd, const Penalties *pen) {
    todo_.clear();
    for (uint32_t r = 0; r < rows(); r++) if ((bits[r] & LIVE) && seq[r] < 0) todo_.push_back(r);
    // by start; pieces are appended in this order (equal starts: by row -- the reference leaves that to an unstable sort)
    std::sort(todo_.begin(), todo_.end(), [&](uint32_t a, uint32_t b) { return r_lo[a] != r_lo[b] ? r_lo[a] < r_lo[b] : a < b; });
    const uint64_t *edge = ref.sop + 1, *edge_end = ref.sop + ref.nseq + 1;       // edge[s] = last 1-based coordinate of sequence s
    for (uint32_t r : todo_) {
      const int64_t first = std::lower_bound(edge, edge_end, r_lo[r]) - edge;     // the sequence that holds the first base
      if (first >= ref.nseq) { why = "alignment starts behind the last reference sequence"; return BROKEN; }
      int64_t past = std::lower_bound(edge + first, edge_end, r_hi[r]) - edge + 1; // one past the sequence that holds the last base
      if (past > ref.nseq) { why = "alignment ends behind the last reference sequence"; return BROKEN; }
      if (past > first + 1) {
        if (!ref.packed || !rd.bases || !pen) return WANTS_REFERENCE;
        if (!cut_at_junctions(r, first, past, ref, rd, *pen)) return BROKEN;
        bits[r] &= ~(uint32_t)LIVE;
        continue;
      }
      seq[r] = first;
      r_lo[r] -= ref.sop[first];
      r_hi[r] -= ref.sop[first];
      bits[r] &= ~(uint32_t)UNPLACED;
    }
    set_bits = (set_bits & ~(uint32_t)SET_ORDERED) | SET_PLACED;
    return DONE;
  }

  bool cut_at_junctions(uint32_t r, int64_t first, int64_t past, const Reference &ref, const Read &rd, const Penalties &pen) {
    const bool reversed = (bits[r] & REVERSED) != 0;
    if (r_lo[r] <= ref.sop[first]) { why = "alignment does not start in its first sequence"; return false; }
    for (int64_t s = first; s < past; s++) {
      // the window of sequence s in offsets from the alignment's first reference base (r_lo is 1-based, sop 0-based)
      const int64_t lo = r_lo[r] > ref.sop[s] ? 0 : (int64_t)(ref.sop[s] - r_lo[r] + 1);
      const int64_t hi = (int64_t)((r_hi[r] <= ref.sop[s + 1] ? r_hi[r] : ref.sop[s + 1]) - r_lo[r]);
      const int cut = cut_window(str(r), lo, hi, piece_);
      if (cut == CUT_EMPTY) continue;
      if (cut != CUT_OK) { why = "alignment string has no matched base in a sequence it spans"; return false; }
      if (rows() >= ROWS_MAX) { why = "more than 32767 alignments for one read"; return false; }
      uint32_t ql, qh, read_at;                          // read interval of the piece, and where it starts in the read as aligned
      if (reversed) { ql = q_hi[r] - (uint32_t)piece_.read_last; qh = q_hi[r] - (uint32_t)piece_.read_first; read_at = rd.len - qh; }
      else { ql = q_lo[r] + (uint32_t)piece_.read_first; qh = q_lo[r] + (uint32_t)piece_.read_last; read_at = ql - 1; }
      if (ql > qh || qh > rd.len) { why = "piece of a cut alignment falls outside the read"; return false; }
      const uint64_t rl = r_lo[r] + (uint64_t)piece_.ref_first - ref.sop[s], rh = r_lo[r] + (uint64_t)piece_.ref_last - ref.sop[s];
      if (rh < rl || rh - rl >= (uint64_t)INT32_MAX) { why = "piece of a cut alignment is too long"; return false; }
      int sc = 0;
      if (!string_score(sc, piece_.str.data(), ref, ref.sop[s] + rl - 1, rd, read_at, reversed, pen)) { why = "piece of a cut alignment runs off the read"; return false; }
      const uint32_t keep = set_bits;
      const uint32_t row = add(sc, ql, qh, rl, rh, s, reversed, piece_.str.data(), (uint32_t)piece_.str.size());
      set_bits = keep;
      // a piece inherits what the pass before may have left in the cut row (a table that lives across calls)
      quality[row] = quality[r]; prob[row] = prob[r]; primary[row] = primary[r]; segment[row] = segment[r]; rank[row] = rank[r];
      bits[row] = (bits[r] & ~(uint32_t)UNPLACED) | LIVE;
    }
    return true;
  }

  // ---- rules 2 and 3 ----
  struct Slot { uint64_t major; uint32_t read_span, ref_span, row; };
  static int tie_predicate(const void *x, const void *y) {
    const Slot *a = (const Slot *)x, *b = (const Slot *)y;
    if (a->major != b->major) return a->major < b->major ? -1 : 1;
    // first argument's read extent against second argument's reference extent, longer first (results.c:466-470)
    return a->read_span > b->ref_span ? -1 : (a->read_span < b->ref_span ? 1 : 0);
  }
  bool prune_and_order() {
    by_score.clear();
    slots_.clear();
    for (uint32_t r = 0; r < rows(); r++) {
      rank[r] = 0;
      if (!(bits[r] & LIVE)) continue;
      if (seq[r] < 0 || seq[r] >= INT32_MAX || r_lo[r] > UINT32_MAX) { why = "alignment without a sequence reached the ordering"; return false; }
      slots_.push_back(Slot{(uint64_t)seq[r] << 33 | (uint64_t)((bits[r] & REVERSED) != 0) << 32 | r_lo[r], span_q(r), (uint32_t)(r_hi[r] - r_lo[r]), r});
    }
    set_bits |= SET_NUMBERED;
    if (slots_.size() < 2) {
      for (const Slot &s : slots_) by_score.push_back((int32_t)s.row);
      set_bits |= SET_ORDERED;
      return true;
    }
    sorted_ = slots_;
    std::sort(sorted_.begin(), sorted_.end(), [](const Slot &a, const Slot &b) { return a.major != b.major ? a.major < b.major : a.row < b.row; });
    bool tie = false;
    for (size_t i = 1; i < sorted_.size() && !tie; i++) tie = sorted_[i].major == sorted_[i - 1].major;
    if (tie) { sorted_ = slots_; qsort(sorted_.data(), sorted_.size(), sizeof(Slot), tie_predicate); }      // see the note at the top
    // one survivor at a time: a row is dropped when the survivor in front of it contains it
    keys_.clear();
    uint32_t holder = sorted_[0].row;
    auto keep = [&](uint32_t r) {
      // score descending, forward strand first, sequence, start, read span descending, then the order found here
      keys_.push_back(Key{(uint64_t)(~((uint32_t)score[r] ^ 0x80000000u)) << 32 | (uint64_t)((bits[r] & REVERSED) != 0) << 31 | (uint64_t)seq[r],
                          (uint64_t)r_lo[r] << 32 | (uint32_t)~span_q(r), (uint32_t)keys_.size(), r});
    };
    keep(holder);
    for (size_t i = 1; i < sorted_.size(); i++) {
      const uint32_t r = sorted_[i].row;
      const bool inside = r_hi[r] <= r_hi[holder] && score[r] <= score[holder] && q_lo[r] >= q_lo[holder] && q_hi[r] <= q_hi[holder] &&
                          seq[r] == seq[holder] && ((bits[r] ^ bits[holder]) & REVERSED) == 0;
      if (inside) { bits[r] &= ~(uint32_t)LIVE; continue; }
      if (keys_.size() == ROWS_MAX) { why = "more than 32767 alignments for one read"; return false; }
      holder = r;
      keep(r);
    }
    std::sort(keys_.begin(), keys_.end(), [](const Key &a, const Key &b) { return a.hi != b.hi ? a.hi < b.hi : (a.lo != b.lo ? a.lo < b.lo : a.arrival < b.arrival); });
    int16_t level = 0;
    for (size_t i = 0; i < keys_.size(); i++) {
      const uint32_t r = keys_[i].row;
      if (i && score[r] != score[keys_[i - 1].row]) level++;
      rank[r] = level;
      by_score.push_back((int32_t)r);
    }
    set_bits |= SET_ORDERED;
    return true;
  }

  // ---- rule 4 ----
  bool group_by_read_interval() {
    const size_t n = by_score.size();
    for (size_t i = 0; i < n; i++) segment[by_score[i]] = -1;
    nsegments = 0;
    for (size_t founder = 0; founder < n;) {
      const uint32_t f = (uint32_t)by_score[founder];
      if (nsegments == ROWS_MAX) { why = "more than 32767 read segments"; return false; }
      segment[f] = (int16_t)nsegments;
      size_t next = n;
      for (size_t j = founder + 1; j < n; j++) {
        const uint32_t r = (uint32_t)by_score[j];
        if (segment[r] >= 0) continue;
        const uint32_t shorter = std::min(span_q(f), span_q(r));
        const uint32_t need = (uint32_t)(shorter * (80 / 100.0));                    // results.c:713, :733
        if (q_lo[f] + need < q_hi[r] && q_lo[r] + need < q_hi[f]) segment[r] = (int16_t)nsegments;
        else if (next == n) next = j;
      }
      nsegments++;
      founder = next;
    }
    // by segment, score order kept inside: by_score is sorted by score already, so this is a counting pass
    segment_begin.assign((size_t)nsegments + 1, 0);
    for (size_t i = 0; i < n; i++) segment_begin[(size_t)segment[by_score[i]] + 1]++;
    for (int g = 0; g < nsegments; g++) segment_begin[(size_t)g + 1] += segment_begin[(size_t)g];
    by_segment.assign(n, 0);
    fill_.assign(segment_begin.begin(), segment_begin.end() - 1);
    for (size_t i = 0; i < n; i++) by_segment[(size_t)fill_[(size_t)segment[by_score[i]]]++] = by_score[i];
    set_bits |= SET_SEGMENTED;
    return true;
  }

  // ---- rule 5 ----
  // Re-order rows [0, m) of a segment (all of one score) by the given keys, equal keys keeping their order.
  template <class KeyOf> void reorder_top(int32_t *rows_of, int m, KeyOf key_of) {
    top_.clear();
    for (int i = 0; i < m; i++) { Key k = key_of((uint32_t)rows_of[i]); k.arrival = (uint32_t)i; k.row = (uint32_t)rows_of[i]; top_.push_back(k); }
    std::sort(top_.begin(), top_.end(), [](const Key &a, const Key &b) { return a.hi != b.hi ? a.hi < b.hi : (a.lo != b.lo ? a.lo < b.lo : a.arrival < b.arrival); });
    for (int i = 0; i < m; i++) rows_of[i] = (int32_t)top_[(size_t)i].row;
  }
  bool qualify(int g, const Read &rd) {
    static const float LN10 = 2.30259f;                          // a float in the reference (results.c:103); the arithmetic below depends on it
    int32_t *rows_of = by_segment.data() + segment_begin[(size_t)g];
    const int n = segment_begin[(size_t)g + 1] - segment_begin[(size_t)g];
    if (n < 1) return true;
    const int best = score[rows_of[0]];
    if (best < 1) { quality[rows_of[0]] = 0; return assign_probabilities(rows_of, n); }
    // how much of the seeds and of the candidates was looked at caps the quality (results.c:1188-1192)
    double looked = ((double)n_hits_used) / (n_hits_tot + 3);
    const double aligned = ((double)n_ali_done) / (n_ali_tot + 3);
    if (looked > aligned) looked = aligned;
    looked = (looked > 1E-7) ? -10 * log(looked) / LN10 : QUALITY_TOP;
    const int cap = (looked < QUALITY_TOP) ? QUALITY_TOP - (int)looked : 0;
    int second = 0, nsecond = 0, crowd = 0;
    if (n > 1) {
      second = score[rows_of[1]];
      int i = 2;
      while (i < n && score[rows_of[i]] == second) i++;
      nsecond = i - 1;
      crowd = (int)(10 * log((double)nsecond) / LN10);                                // results.c:1223
    }
    int q;
    if (n > 1 && second == best) {
      // several alignments share the best score: the one covering most of the read leads; among equally long ones the one
      // whose substitutions sit on the worst base qualities (results.c:1230-1296)
      const int m = nsecond + 1;
      reorder_top(rows_of, m, [&](uint32_t r) {
        return Key{(uint64_t)(uint32_t)~span_q(r) << 32 | (uint64_t)((bits[r] & REVERSED) != 0) << 31 | (uint64_t)seq[r], r_lo[r], 0, 0}; });
      const uint32_t lead_span = span_q((uint32_t)rows_of[0]);
      if (lead_span != span_q((uint32_t)rows_of[1])) q = QUALITY_FLOOR_UNIQUE;
      else if (!rd.quals) q = 0;
      else {
        int lead_sum = 0, low_sum = 0, sum = 0, low_at = 1;
        if (!subst_quality_sum(lead_sum, rd.quals, rd.len, q_lo[rows_of[0]], q_hi[rows_of[0]], str((uint32_t)rows_of[0])) ||
            !subst_quality_sum(low_sum, rd.quals, rd.len, q_lo[rows_of[1]], q_hi[rows_of[1]], str((uint32_t)rows_of[1]))) { why = "alignment string and read interval disagree"; return false; }
        for (int i = 2; i < n && score[rows_of[i]] == best && span_q((uint32_t)rows_of[i]) >= lead_span; i++) {
          if (!subst_quality_sum(sum, rd.quals, rd.len, q_lo[rows_of[i]], q_hi[rows_of[i]], str((uint32_t)rows_of[i]))) { why = "alignment string and read interval disagree"; return false; }
          if (sum < low_sum) { low_sum = sum; low_at = i; }
        }
        if (lead_sum > low_sum) { std::swap(rows_of[0], rows_of[low_at]); q = QUALITY_FLOOR_UNIQUE; }
        else q = (lead_sum == low_sum) ? 0 : QUALITY_FLOOR_UNIQUE;
      }
      if (q < 1)                                            // nothing tells them apart: back to the output order
        reorder_top(rows_of, m, [&](uint32_t r) {
          return Key{(uint64_t)((bits[r] & REVERSED) != 0) << 63 | (uint64_t)seq[r], r_lo[r] << 32 | (uint32_t)~span_q(r), 0, 0}; });
    } else {
      q = (int)(QUALITY_TOP * (1 - exp(((double)(second - best)) * 10 / rd.len)) - crowd);    // results.c:1302-1303
      if (q >= 0) q += QUALITY_FLOOR_UNIQUE;
      if (q > cap) q = cap;
    }
    if (q > QUALITY_TOP) q = QUALITY_TOP; else if (q < 0) q = 0;
    quality[rows_of[0]] = q;
    for (int i = 1; i < n; i++) quality[rows_of[i]] = 0;
    return assign_probabilities(rows_of, n);
  }
  // the best score class shares 1 - P(wrong), the second one P(wrong), everything below gets 0 (results.c:1343-1398)
  bool assign_probabilities(const int32_t *rows_of, int n) {
    static const float LN10 = 2.30259f;
    int n1 = 1, n2 = 0;
    while (n1 < n && score[rows_of[n1]] == score[rows_of[0]]) n1++;
    if (n1 < n) { n2 = 1; while (n1 + n2 < n && score[rows_of[n1 + n2]] == score[rows_of[n1]]) n2++; }
    double p1, p2;
    if (n1 == 1) {
      const int q = quality[rows_of[0]] < 0 ? 0 : quality[rows_of[0]];
      p2 = exp(((double)(-LN10 * q)) / 10);                                            // float product first (results.c:1382)
      p1 = 1.0 - p2;
      if (n2 > 1) p2 /= n2;
    } else p1 = p2 = 1.0 / n1;
    if (n1 + n2 > ROWS_MAX) { why = "more than 32767 alignments for one read"; return false; }
    for (int i = 0; i < n; i++) prob[rows_of[i]] = i < n1 ? p1 : (i < n1 + n2 ? p2 : 0.0);
    if (n1 == 1 && n2 == 0) bits[rows_of[0]] |= ONLY_ONE;
    return true;
  }


  // ---- what callers above the pass ask of a table ----
  // quality and score of the best alignment, 0 / 0 for an empty table (resultSetGetMappingScore, results.c:2407)
  int top_quality(int *top_score) const {
    if (by_score.empty()) { if (top_score) *top_score = 0; return 0; }
    if (top_score) *top_score = score[by_score[0]];
    return quality[by_score[0]];
  }
  // How many ordered alignments carry the set's best score, and how many count as second class.  The maxima are the
  // RUNNING ones of the set (score_max / score_2nd: they have seen alignments that were dropped again), and the second
  // count is all-or-nothing: everything behind the first class if the first alignment there reaches score_2nd, else 0
  // (resultSetGetScorStats, results.c:2363-2386, tests element [first] throughout)
  void score_classes(int *nbest, int *nsecond) const {
    const int n = (int)by_score.size();
    int i = 0;
    while (i < n && score[by_score[i]] >= score_max) i++;
    if (nbest) *nbest = i;
    if (nsecond) *nsecond = (i < n && score[by_score[i]] >= score_2nd) ? n - i : 0;
  }
  // -> true when exactly one alignment has the best score; *deepest_rank: score ranks that take part in pairing
  bool rank_depth(int *deepest_rank) const {                      // resultSetGetRankDepth, results.c:2388-2405
    int nbest, nsecond;
    score_classes(&nbest, &nsecond);
    if (deepest_rank) *deepest_rank = nbest < 2 ? 1 : 0;
    return nbest == 1;
  }
  // the top of the order: rows that share the best score when there are several, else the best and the whole second class
  // (getNumberOfTopSwatRESULTs, results.c:838-866)
  bool top_class(int *ntop) const {
    const int n = (int)by_score.size();
    const bool single = n < 2 || score[by_score[1]] != score[by_score[0]];
    int k = n;
    if (n > 2) { k = 2; while (k < n && score[by_score[k]] == score[by_score[1]]) k++; }
    if (ntop) *ntop = k;
    return single;
  }
  // Output filter (resultSetFilterResults, results.c:2592-2626): absolute score, matched bases, distance from the best
  void apply_output_filter(int min_score, int below_best, double min_identity, uint32_t read_len) {
    if (by_score.empty()) return;
    const double idt = min_identity <= 1.0 ? min_identity * read_len : min_identity;
    const int need_matched = (int)(uint32_t)idt, best = score[by_score[0]];
    const int floor_rel = (below_best >= 0 && min_score + below_best < best) ? best - below_best : 0;
    for (int32_t r : by_score) {
      if (score[r] < min_score || matched_bases(str((uint32_t)r)) < need_matched) bits[r] |= WITHHELD;
      else if (score[r] < floor_rel) bits[r] |= BELOW_RELATIVE;
    }
  }
  static int matched_bases(const uint8_t *s) {                    // diffStrCalcAliLen's match count (diffstr.c:932-952)
    int m = 0;
    for (; *s; ++s) m += (*s & 63) + ((*s >> 6) == OP_MATCH ? 1 : 0);
    return m;
  }

  // One mapping call's alignments into the table: what resultSetAddFromAli (results.c:1852-1942), called once per candidate,
  // leaves in the set -- restated as a machine over the array's slots.  `res` holds EVERY alignment of the call, candidate by
  // candidate (`reverse & 2` marks the first of each; smaltgpu_callctx.raw_alignments when the table is not empty -- for an
  // empty table the library's own result is that already).  A candidate opens the slot behind the array's end; each of its
  // alignments is written into the open slot and compared with the slot before it (coordinates, score, sequence); one that
  // repeats it hands the slot back: the array shrinks by one and the slot stays open, now outside the array.  Anything else
  // stays, and the alignment after it opens the slot behind the array's end -- after a repeat that is the same slot again.  So
  // the alignment that follows a repeat is overwritten by its successor, or left outside the array when the candidate ends: it
  // is lost, although its score went through the set's maxima (which the call reports: max_after, second_after).
  // A set with fewer than two entries takes anything.  The running maxima and the counters are the call's.
  template <class Raw> void take_call(const Raw *res, uint32_t n, const uint8_t *pool, int32_t max_after, int32_t second_after) {
    if (n) {
      const uint32_t n_old = rows();
      // slots: [0, n_old) the table's rows, behind them indices into res (+ n_old); `length` is the array's length
      slot_.clear();
      for (uint32_t r = 0; r < n_old; r++) slot_.push_back(r);
      uint32_t length = n_old;
      auto same = [&](uint32_t x, uint32_t y) {            // isIdenticalResult (results.c:556-565) between two slot contents
        int64_t a[6], b[6];
        for (int w = 0; w < 2; w++) {
          const uint32_t v = w ? y : x;
          int64_t *o = w ? b : a;
          if (v < n_old) { o[0] = (int64_t)r_lo[v]; o[1] = (int64_t)r_hi[v]; o[2] = q_lo[v]; o[3] = q_hi[v]; o[4] = score[v]; o[5] = seq[v]; }
          else { const Raw &t = res[v - n_old]; o[0] = (int64_t)t.s_start; o[1] = (int64_t)t.s_end; o[2] = t.q_start; o[3] = t.q_end; o[4] = t.swatscor; o[5] = t.sidx; }
        }
        return a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3] && a[4] == b[4] && a[5] == b[5];
      };
      for (uint32_t i = 0; i < n;) {
        uint32_t j = i + 1;
        while (j < n && !(res[j].reverse & 2u)) j++;
        uint32_t s = length++;                              // the candidate's slot
        bool stays = false;
        for (uint32_t t = i; t < j; t++) {
          if (stays) { s = length++; stays = false; }
          if (s >= slot_.size()) slot_.resize((size_t)s + 1, 0);
          slot_[s] = n_old + t;
          stays = length < 2 || !same(slot_[s], slot_[s - 1]);
          if (!stays) length--;
        }
        i = j;
      }
      // rows that are no longer inside the array (a repeat of a repeat takes an older alignment with it) go, then the new ones come
      uint32_t keep = 0;
      while (keep < n_old && keep < length && slot_[keep] == keep) keep++;
      if (keep < n_old) truncate_rows(keep);
      for (uint32_t s = keep; s < length; s++) {
        const Raw &t = res[slot_[s] - n_old];               // (a slot below n_old that was written again holds an alignment of this call)
        add(t.swatscor, t.q_start, t.q_end, t.s_start, t.s_end, t.sidx, (t.reverse & 1u) != 0, pool + t.stroffs, t.strlen);
      }
      set_bits = 0;
    }
    score_max = max_after; score_2nd = second_after;
  }
  void truncate_rows(uint32_t nrows) {
    score.resize(nrows); quality.resize(nrows); q_lo.resize(nrows); q_hi.resize(nrows); bits.resize(nrows); str_at.resize(nrows); str_len.resize(nrows);
    r_lo.resize(nrows); r_hi.resize(nrows); seq.resize(nrows); prob.resize(nrows); primary.resize(nrows); segment.resize(nrows); rank.resize(nrows);
    by_score.clear(); by_segment.clear(); segment_begin.clear(); nsegments = 0;
  }

  // A table between two passes, as one run of bytes (a block of pairs keeps two of these per pair, not two Tables)
  void pack(std::vector<uint8_t> &o, bool append = false) const {
    if (!append) o.clear();
    const uint32_t head[13] = {rows(), (uint32_t)by_score.size(), (uint32_t)segment_begin.size(), (uint32_t)strings.size(), set_bits, (uint32_t)nsegments,
                               (uint32_t)n_ali_done, (uint32_t)n_ali_tot, (uint32_t)score_max, (uint32_t)score_2nd, n_hits_used, n_hits_tot, (uint32_t)by_segment.size()};
    put(o, head, 13);
    put(o, r_lo.data(), rows()); put(o, r_hi.data(), rows()); put(o, seq.data(), rows()); put(o, prob.data(), rows());
    put(o, score.data(), rows()); put(o, quality.data(), rows()); put(o, q_lo.data(), rows()); put(o, q_hi.data(), rows()); put(o, bits.data(), rows());
    put(o, str_at.data(), rows()); put(o, str_len.data(), rows()); put(o, by_score.data(), by_score.size()); put(o, by_segment.data(), by_segment.size());
    put(o, segment_begin.data(), segment_begin.size());
    put(o, primary.data(), rows()); put(o, segment.data(), rows()); put(o, rank.data(), rows());
    put(o, strings.data(), strings.size());
  }
  void unpack(const uint8_t *p, size_t len) {
    clear();
    if (!len) return;
    uint32_t head[13];
    get(p, head, 13);
    const size_t n = head[0], nl = head[1], nb = head[2], ns = head[3];
    set_bits = head[4]; nsegments = (int)head[5]; n_ali_done = (int32_t)head[6]; n_ali_tot = (int32_t)head[7]; score_max = (int32_t)head[8]; score_2nd = (int32_t)head[9];
    n_hits_used = head[10]; n_hits_tot = head[11];
    r_lo.resize(n); get(p, r_lo.data(), n); r_hi.resize(n); get(p, r_hi.data(), n); seq.resize(n); get(p, seq.data(), n); prob.resize(n); get(p, prob.data(), n);
    score.resize(n); get(p, score.data(), n); quality.resize(n); get(p, quality.data(), n); q_lo.resize(n); get(p, q_lo.data(), n); q_hi.resize(n); get(p, q_hi.data(), n);
    bits.resize(n); get(p, bits.data(), n); str_at.resize(n); get(p, str_at.data(), n); str_len.resize(n); get(p, str_len.data(), n);
    by_score.resize(nl); get(p, by_score.data(), nl); by_segment.resize(head[12]);
    get(p, by_segment.data(), by_segment.size()); segment_begin.resize(nb); get(p, segment_begin.data(), nb);
    primary.resize(n); get(p, primary.data(), n); segment.resize(n); get(p, segment.data(), n); rank.resize(n); get(p, rank.data(), n);
    strings.resize(ns); get(p, strings.data(), ns);
  }
  template <class T> static void put(std::vector<uint8_t> &o, const T *v, size_t n) { if (n) { const uint8_t *b = (const uint8_t *)v; o.insert(o.end(), b, b + n * sizeof(T)); } }
  template <class T> static void get(const uint8_t *&p, T *v, size_t n) { if (n) { memcpy(v, p, n * sizeof(T)); p += n * sizeof(T); } }

  struct Key { uint64_t hi, lo; uint32_t arrival, row; };

 private:
  std::vector<uint32_t> todo_;
  std::vector<Slot> slots_, sorted_;
  std::vector<Key> keys_, top_;
  std::vector<int32_t> fill_;
  Piece piece_;
};

}  // namespace smgpost
#endif
