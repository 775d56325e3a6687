// tests/hostemu/pair_check.cpp -- TEST DRIVER (no device): the product's pair logic (smalt_amd/csrc/smg_pairrun.hpp: rounds,
// post-call passes, proper-pair probe, search intervals; smaltgpu_report_emit_pairs: pairing, choice, lines) with the
// mapping calls REPLAYED from a `refdump -P` record of the reference (oracle/DUMPFORMAT.md).  Every call the product's plan
// asks for must be one the reference made for that pair -- same mate, same restriction (the search intervals line by line),
// same threshold and running maxima; the alignments the reference's call added go into the product's tables, and the
// report text comes out on stdout for comparison with what the reference program printed.
//
//   pair_check <refdump.txt> <reads_1.fq> <reads_2.fq> <seqinfo.txt> key=value ...
//   seqinfo: one line "<name> <length>" per reference sequence;  keys: k dmin dmax lib every fmt mod out minsw below minid seed threads
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fstream>
#include <map>
#include <sstream>
#include "../../smalt_amd/csrc/smg_pairrun.hpp"
#include "dump_record.hpp"

using namespace smgpairs;

struct ReplayExec {
  std::vector<RecPair> &rec;
  const BlockInput *in;
  int k;
  std::vector<smaltgpu_result> res;
  std::vector<uint64_t> res_off;
  std::vector<uint8_t> pool;
  std::vector<smaltgpu_readstat> stat;
  bool totals(const uint32_t *ids, uint32_t n, uint32_t *hits, std::string &) {
    // the reference maps the mate of its first call first: give that mate fewer hits (the read wins a tie)
    for (uint32_t i = 0; i < n; i++) {
      const RecPair &rp = rec[ids[i] >> 1];
      const int first = rp.calls.empty() ? 0 : rp.calls[0].mate;
      hits[i] = ((int)(ids[i] & 1) == first) ? 1u : 2u;
      // a pair with a mate shorter than a word starts with the call for the long mate (rmap.c:1836-1864); the short mate has no
      // hits, the long one what its call counted
      const uint32_t me = PairBlock::len_of(*in, ids[i]), other = PairBlock::len_of(*in, ids[i] ^ 1u);
      if (me < (uint32_t)k) hits[i] = 0;
      else if (other < (uint32_t)k) hits[i] = rp.calls.empty() ? 0u : (uint32_t)rp.calls[0].rx[6];
    }
    return true;
  }
  bool map(const Round &rd, smaltgpu_batch_out *o, std::string &err) {
    res.clear(); pool.clear(); stat.assign(rd.n, smaltgpu_readstat()); res_off.assign((size_t)rd.n + 1, 0);
    char msg[256];
    for (uint32_t i = 0; i < rd.n; i++) {
      RecPair &rp = rec[rd.ids[i] >> 1];
      const int mate = (int)(rd.ids[i] & 1);
      const bool restricted = rd.kind == ROUND_RESTRICTED || rd.kind == ROUND_FINE;
      RecCall *c = nullptr;
      for (RecCall &x : rp.calls) if (!x.used && x.mate == mate && (x.niv >= 0) == restricted && (x.fine != 0) == (rd.kind == ROUND_FINE)) { c = &x; break; }
      if (!c) { snprintf(msg, sizeof(msg), "pair %u: the plan asks for a call (mate %d, round kind %d) the reference did not make", rd.ids[i] >> 1, mate, rd.kind); err = msg; return false; }
      c->used = true;
      if (getenv("PAIR_TRACE") && (uint32_t)atoi(getenv("PAIR_TRACE")) == (rd.ids[i] >> 1)) fprintf(stderr, "TRACE pair %u: round kind %d, mate %d, %zu new alignments, max1 %d, stats %d %d %d %d\n", rd.ids[i] >> 1, (int)rd.kind, mate, c->res.size(), c->max1, c->rx[3], c->rx[4], c->rx[5], c->rx[6]);
      if (restricted) {
        const uint64_t a = rd.iv_off[i], b = rd.iv_off[i + 1];
        bool same = (int)(b - a) == c->niv;
        for (uint64_t j = a; same && j < b; j++) same = rd.iv[j].sidx == c->ivs[j - a].sidx && rd.iv[j].lo == c->ivs[j - a].lo && rd.iv[j].hi == c->ivs[j - a].hi;
        if (!same) { snprintf(msg, sizeof(msg), "pair %u mate %d: search intervals differ from the reference's", rd.ids[i] >> 1, mate); err = msg; return false; }
      }
      if (rd.kind == ROUND_FINE && rd.min_score[i] != c->minscor) { snprintf(msg, sizeof(msg), "pair %u: threshold %d, reference %d", rd.ids[i] >> 1, rd.min_score[i], c->minscor); err = msg; return false; }
      if (rd.prev_max && (rd.prev_max[2 * i] != c->prevmax[0] || rd.prev_max[2 * i + 1] != c->prevmax[1])) {
        snprintf(msg, sizeof(msg), "pair %u mate %d: running maxima %d,%d, reference %d,%d", rd.ids[i] >> 1, mate, rd.prev_max[2 * i], rd.prev_max[2 * i + 1], c->prevmax[0], c->prevmax[1]); err = msg; return false;
      }
      res_off[i] = res.size();
      for (size_t j = 0; j < c->res.size(); j++) {
        smaltgpu_result r = c->res[j];
        r.stroffs = (uint32_t)pool.size(); r.strlen = (uint32_t)c->strs[j].size();
        r.reverse |= 2u;                          // each recorded alignment stands for itself: what was dropped as a repeat is not in the record
        pool.insert(pool.end(), c->strs[j].begin(), c->strs[j].end());
        res.push_back(r);
      }
      smaltgpu_readstat &st = stat[i];
      st.swatscor_max = c->rx[1]; st.swatscor_2ndmax = c->rx[2]; st.n_ali_done = c->rx[3]; st.n_ali_tot = c->rx[4]; st.n_hits_used = (uint32_t)c->rx[5]; st.n_hits_tot = (uint32_t)c->rx[6];
      st.nres = (uint32_t)c->res.size(); st.max1scor = c->max1;
    }
    res_off[rd.n] = res.size();
    if (res.empty()) res.resize(1);
    if (pool.empty()) pool.resize(1);
    o->nreads = rd.n; o->res_off = res_off.data(); o->res = res.data(); o->diffstr = pool.data(); o->stat = stat.data();
    return true;
  }
};

static std::string slurp(const char *path) { std::ifstream f(path, std::ios::binary); std::stringstream ss; ss << f.rdbuf(); return ss.str(); }

int main(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "usage: pair_check refdump reads1 reads2 seqinfo key=value...\n"); return 2; }
  std::map<std::string, std::string> kv;
  for (int a = 5; a < argc; a++) { const char *e = strchr(argv[a], '='); if (e) kv[std::string((const char *)argv[a], (size_t)(e - argv[a]))] = e + 1; }
  auto geti = [&](const char *k, int dflt) { return kv.count(k) ? atoi(kv[k].c_str()) : dflt; };
  std::vector<RecPair> rec = load_dump(argv[1]);
  // reads through the library's own parser
  smaltgpu_reads *rs[2] = {smaltgpu_reads_create(), smaltgpu_reads_create()};
  smaltgpu_reads_view v[2];
  std::string text[2] = {slurp(argv[2]), slurp(argv[3])};
  for (int w = 0; w < 2; w++) if (smaltgpu_reads_parse(rs[w], text[w].data(), text[w].size(), 1, 0, 1, &v[w])) { fprintf(stderr, "parse: %s\n", smaltgpu_last_error()); return 1; }
  if (v[0].nreads != v[1].nreads || v[0].nreads != rec.size()) { fprintf(stderr, "pair counts differ: %u %u %zu\n", v[0].nreads, v[1].nreads, rec.size()); return 1; }
  std::vector<std::string> names;
  std::vector<uint64_t> sop(1, 0);
  { std::ifstream f(argv[4]); std::string nm; unsigned long long len; while (f >> nm >> len) { names.push_back(nm); sop.push_back(sop.back() + len); } }
  std::vector<const char *> name_ptr;
  for (const std::string &s : names) name_ptr.push_back(s.c_str());

  BlockInput in;
  for (int w = 0; w < 2; w++) { in.bases[w] = v[w].bases; in.quals[w] = v[w].has_qual ? v[w].quals : nullptr; in.off[w] = v[w].read_off; }
  in.npairs = v[0].nreads;
  BlockParams bp;
  memset(&bp.map, 0, sizeof(bp.map));
  bp.map.match = 1; bp.map.mismatch = -2; bp.map.gap_init = -4; bp.map.gap_ext = -3;
  bp.d_min = geti("dmin", 0); bp.d_max = geti("dmax", 500); bp.lib = geti("lib", 1); bp.every_pair = geti("every", 0) != 0; bp.k = geti("k", 13);
  bp.sop = sop.data(); bp.nseq = (int64_t)names.size(); bp.packed_host = nullptr; bp.nthreads = geti("threads", 1);
  smaltgpu_pairs *ps = smaltgpu_pairs_create();
  ReplayExec ex{rec, &in, bp.k};
  if (!ps->blk.run(ex, in, bp)) { fprintf(stderr, "pair_check: %s\n", ps->blk.error.c_str()); return 1; }
  // every unrestricted or restricted call of the reference that adds alignments must have been asked for (calls for a mate
  // shorter than a word, and the repeats of a lone mate's call, add nothing new and are not made)
  for (size_t p = 0; p < rec.size(); p++) {
    const PairPlan &pl = ps->blk.plan[p];
    for (const RecCall &c : rec[p].calls) if (!c.used && !c.res.empty() && !pl.lone) { fprintf(stderr, "pair_check: pair %zu: a call of the reference (mate %d, niv %d, fine %d) was not made\n", p, c.mate, c.niv, c.fine); return 1; }
  }
  if (getenv("PAIR_TRACE")) {
    const size_t p = (size_t)atoi(getenv("PAIR_TRACE"));
    for (int w = 0; w < 2 && p < rec.size(); w++) {
      smgpost::Table tb;
      tb.unpack(ps->blk.packed.data(2 * p + (size_t)w), ps->blk.packed.size(2 * p + (size_t)w));
      fprintf(stderr, "TRACE pair %zu mate %d at rest: %u rows, stats %d %d %u %u, max %d 2nd %d;", p, w, tb.rows(), tb.n_ali_done, tb.n_ali_tot, tb.n_hits_used, tb.n_hits_tot, tb.score_max, tb.score_2nd);
      for (uint32_t r = 0; r < tb.rows(); r++) fprintf(stderr, " [score %d quality %d bits %x prob %.6g]", tb.score[r], tb.quality[r], tb.bits[r], tb.prob[r]);
      fprintf(stderr, "\n");
    }
  }
  smaltgpu_report_opts ro;
  memset(&ro, 0, sizeof(ro));
  ro.format = geti("fmt", 0); ro.modflags = (uint32_t)geti("mod", 0); ro.outflags = (uint32_t)geti("out", 3); ro.min_swscor = geti("minsw", 18);
  ro.min_swscor_below_max = geti("below", 0); ro.min_identity = kv.count("minid") ? atof(kv["minid"].c_str()) : 0.0;
  smaltgpu_pair_opts po;
  po.insert_min = bp.d_min; po.insert_max = bp.d_max; po.library = bp.lib; po.every_pair = bp.every_pair; po.nthreads = bp.nthreads;
  if (ro.outflags & SMALTGPU_OUT_RANDSEL) srand48(geti("seed", 1));
  smaltgpu_report *rep = smaltgpu_report_create();
  const char *out; uint64_t len;
  {                                                  // the report learns the sequence lengths here (SSAHA lines); the header text itself is not compared
    smaltgpu_report_opts quiet = ro;
    quiet.modflags &= ~(uint32_t)SMALTGPU_REP_HEADER;
    if (smaltgpu_report_header(rep, name_ptr.data(), sop.data(), (int64_t)names.size(), &quiet, "pair_check", "0", 0, nullptr, &out, &len)) { fprintf(stderr, "header: %s\n", smaltgpu_last_error()); return 1; }
  }
  if (smaltgpu_report_emit_pairs(rep, ps, &v[0], &v[1], name_ptr.data(), (int64_t)names.size(), &ro, &po, bp.nthreads, &out, &len)) { fprintf(stderr, "emit: %s\n", smaltgpu_last_error()); return 1; }
  fwrite(out, 1, len, stdout);
  // pair flags for the caller to compare with the reference's (PE lines)
  for (size_t p = 0; p < rec.size(); p++) fprintf(stderr, "FLG %zu %u %d\n", p, (unsigned)ps->blk.plan[p].state, (int)(ps->blk.plan[p].lone || ps->blk.plan[p].idle));
  return 0;
}
