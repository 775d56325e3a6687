// smg_pairs.cpp -- smaltgpu_map_pairs: a block of read pairs through the rounds of smg_pairrun.hpp with the device behind
// them (replaces rmapPair, rmap.c:1744-2112, for a block).  Host code: every round gathers its reads into one
// batch for smaltgpu_map_batch_ctx, worker threads do the post-call passes and the decisions between the rounds.
// Compiled with g++ -ffp-contract=off like smg_post.cpp (mapping qualities are double arithmetic).
#include <chrono>
#include "smg_pairrun.hpp"

extern "C" int smaltgpu_set_error(int code, const char *msg);

namespace {

using namespace smgpairs;

struct DeviceExec {
  smaltgpu_mapper *m;
  const BlockInput &in;
  smaltgpu_params par;
  uint32_t cap_reads;
  uint64_t cap_bases;
  const smaltgpu_resident_reads *resident = nullptr;      // the block's reads in HBM: rounds are gathered on the device
  int nthreads = 1;                                       // host threads for the gather of a round's reads
  std::vector<uint8_t> bases, quals;
  std::vector<uint64_t> off;
  double ms[4] = {0, 0, 0, 0};
  double kernel_ms[5][16];                   // device time per kernel (smaltgpu_timer_name order), rounds 0-3 and [4] = the hit totals
  uint64_t work[5][32];                      // the mapper's work counters likewise
  int rc = SMALTGPU_OK;
  void tally(int slot) {
    double t[16] = {0}; uint64_t w[32] = {0};
    const int nt = smaltgpu_timers(m, t, w, 32);
    for (int i = 0; i < nt && i < 16; i++) kernel_ms[slot][i] += t[i];
    for (int i = 0; i < 32; i++) work[slot][i] += w[i];
  }
  bool with_quals() const { return in.quals[0] && in.quals[1]; }
  void gather(const uint32_t *ids, uint32_t n) {
    off.resize((size_t)n + 1);
    uint64_t tot = 0;
    for (uint32_t i = 0; i < n; i++) { off[i] = tot; tot += PairBlock::len_of(in, ids[i]); }
    off[n] = tot;
    bases.resize(tot + 1);
    if (with_quals()) quals.resize(tot + 1);
    PairBlock::spread(n, nthreads, [&](uint32_t lo, uint32_t hi, int) {
      for (uint32_t i = lo; i < hi; i++) {
        const uint32_t w = ids[i] & 1, p = ids[i] >> 1;
        const uint64_t len = off[i + 1] - off[i];
        memcpy(bases.data() + off[i], in.bases[w] + in.off[w][p], len);
        if (with_quals()) memcpy(quals.data() + off[i], in.quals[w] + in.off[w][p], len);
      }
    });
  }
  bool fail_with(std::string &err, int code) { rc = code; err = smaltgpu_last_error(); if (err.empty()) err = "the device call failed without a message"; return false; }
  double totals_wall = 0;
  bool totals(const uint32_t *ids, uint32_t n, uint32_t *hits, std::string &err) {
    const auto t0 = std::chrono::steady_clock::now();
    const bool ok = totals_inner(ids, n, hits, err);
    totals_wall += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return ok;
  }
  bool totals_inner(const uint32_t *ids, uint32_t n, uint32_t *hits, std::string &err) {
    for (uint32_t lo = 0; lo < n;) {                   // in pieces the mapper can take
      uint32_t hi = lo;
      uint64_t nb = 0;
      while (hi < n && hi - lo < cap_reads && nb + PairBlock::len_of(in, ids[hi]) <= cap_bases) nb += PairBlock::len_of(in, ids[hi++]);
      if (hi == lo) { rc = SMALTGPU_EARG; err = "a read is longer than the mapper was created for"; return false; }
      int rv;
      if (resident) rv = smaltgpu_hit_totals_resident(m, resident, ids + lo, hi - lo, &par, hits + lo);
      else {
        gather(ids + lo, hi - lo);
        rv = smaltgpu_hit_totals(m, bases.data(), with_quals() ? quals.data() : nullptr, off.data(), hi - lo, &par, hits + lo);
      }
      if (rv) return fail_with(err, rv);
      tally(4);
      lo = hi;
    }
    return true;
  }
  bool map(const Round &rd, smaltgpu_batch_out *o, std::string &err) {
    const auto t0 = std::chrono::steady_clock::now();
    if (rd.n > cap_reads) { rc = SMALTGPU_EARG; err = "the block holds more pairs than the mapper's batch size"; return false; }
    smaltgpu_callctx cx;
    memset(&cx, 0, sizeof(cx));
    cx.iv_off = rd.iv_off; cx.iv = rd.iv; cx.min_swatscor = rd.min_score; cx.prev_max = rd.prev_max; cx.fine_index = rd.kind == ROUND_FINE;
    cx.seed_range = rd.seed_range;
    cx.raw_alignments = rd.kind == ROUND_APPEND || rd.kind == ROUND_FINE;      // these rounds append to tables that may hold alignments: Table::take_call compares
    int rv;
    if (resident) rv = smaltgpu_map_batch_ctx_resident(m, resident, rd.ids, rd.n, &par, &cx, o);
    else {
      gather(rd.ids, rd.n);
      rv = smaltgpu_map_batch_ctx(m, bases.data(), with_quals() ? quals.data() : nullptr, off.data(), rd.n, &par, &cx, o);
    }
    // a read that failed on its own carries its code in stat[].errcode; the runner names it
    if (rv && !(SMALTGPU_IS_READ_ERROR(rv) && o->nreads == rd.n)) return fail_with(err, rv);
    tally(rd.kind);
    ms[rd.kind] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return true;
  }
};

}  // namespace

extern "C" smaltgpu_pairs *smaltgpu_pairs_create(void) { return new smaltgpu_pairs(); }
extern "C" void smaltgpu_pairs_free(smaltgpu_pairs *p) { delete p; }

static int map_pairs(smaltgpu_mapper *m, const smaltgpu_resident_reads *resident, const uint8_t *bases1, const uint8_t *quals1, const uint64_t *read_off1,
                     const uint8_t *bases2, const uint8_t *quals2, const uint64_t *read_off2, uint32_t npairs, const smaltgpu_params *par,
                     const smaltgpu_pair_opts *po, smaltgpu_pairs *out) {
  if (!m || !read_off1 || !read_off2 || !par || !po || !out || (!resident && (!bases1 || !bases2))) return smaltgpu_set_error(SMALTGPU_EARG, "smaltgpu_map_pairs: null argument");
  if (po->insert_min > po->insert_max) return smaltgpu_set_error(SMALTGPU_EARG, "smaltgpu_map_pairs: insert_min above insert_max");
  if (po->library < SMALTGPU_LIB_PE || po->library > SMALTGPU_LIB_ANY) return smaltgpu_set_error(SMALTGPU_EARG, "smaltgpu_map_pairs: unknown library type");
  const smaltgpu_index *ix = smaltgpu_mapper_index(m);
  smaltgpu_index_desc ds;
  if (!ix || smaltgpu_index_info(ix, &ds)) return smaltgpu_set_error(SMALTGPU_EARG, "smaltgpu_map_pairs: the mapper has no index");
  const auto wall0 = std::chrono::steady_clock::now();
  BlockInput in;
  in.bases[0] = bases1; in.bases[1] = bases2; in.quals[0] = quals1; in.quals[1] = quals2; in.off[0] = read_off1; in.off[1] = read_off2; in.npairs = npairs;
  BlockParams bp;
  bp.map = *par;
  bp.map.min_swatscor_below_max = 0;                       // MINSCOR_BELOW_MAX_BEST (rmap.c:87)
  bp.d_min = po->insert_min; bp.d_max = po->insert_max; bp.lib = po->library; bp.every_pair = po->every_pair != 0;
  bp.k = ds.k; bp.s = ds.s; bp.sop = ds.sop; bp.nseq = ds.nseq;
  bp.split = (par->rmapflg & SMALTGPU_FLG_SPLIT) != 0;
  // alignments can cross sequence junctions only in concatenated mode: the pieces are scored against a host copy of the reference
  bp.packed_host = (par->rmapflg & SMALTGPU_FLG_SEQBYSEQ) ? nullptr : smaltgpu_index_packed_host(ix);
  if (!(par->rmapflg & SMALTGPU_FLG_SEQBYSEQ) && !bp.packed_host) return SMALTGPU_ENODEV;
  if (bp.packed_host && (!bases1 || !bases2)) return smaltgpu_set_error(SMALTGPU_EARG, "smaltgpu_map_pairs: concatenated mode needs the read bases in host memory too (pieces of alignments across sequences are scored on the host)");
  bp.nthreads = po->nthreads < 1 ? 1 : po->nthreads;
  uint32_t cap_reads = 0, maxlen = 0;
  uint64_t cap_bases = 0;
  if (smaltgpu_mapper_capacity(m, &cap_reads, &maxlen, &cap_bases)) return SMALTGPU_EARG;
  // (the host phases between the rounds are not overlapped inside one call: a block cut into pieces that take turns on the mapper
  //  measured slower than the whole block; a caller hides them by running two blocks on two mappers, see bench.py's two_streams)
  memset(out->kernel_ms, 0, sizeof(out->kernel_ms)); memset(out->work, 0, sizeof(out->work));
  DeviceExec ex{m, in, bp.map, cap_reads, cap_bases};
  ex.resident = resident;
  ex.nthreads = bp.nthreads;
  smaltgpu_mapper_set_host_threads(m, bp.nthreads);        // the copy of a round's results into read order (smaltgpu_fetch_end)
  memset(ex.kernel_ms, 0, sizeof(ex.kernel_ms)); memset(ex.work, 0, sizeof(ex.work));
  const bool ok = out->blk.run(ex, in, bp);
  for (int r = 0; r < 4; r++) { out->round_ms[r] = ex.ms[r]; out->calls[r] = out->blk.nrounds.size() == 4 ? out->blk.nrounds[(size_t)r] : 0; }
  memcpy(out->kernel_ms, ex.kernel_ms, sizeof(out->kernel_ms)); memcpy(out->work, ex.work, sizeof(out->work));
  out->totals_ms = ex.totals_wall;
  smaltgpu_mapper_set_host_threads(m, 1);
  if (!ok) return smaltgpu_set_error(ex.rc != SMALTGPU_OK ? ex.rc : SMALTGPU_EINTERNAL, ("smaltgpu_map_pairs: " + out->blk.error).c_str());
  // the summary: flags, rounds, surviving alignments
  out->info.assign(npairs ? npairs : 1, smaltgpu_pair_info());
  PairBlock::spread(npairs, bp.nthreads, [&](uint32_t lo, uint32_t hi, int) {
    for (uint32_t p = lo; p < hi; p++) {
      const PairPlan &pl = out->blk.plan[p];
      smaltgpu_pair_info &f = out->info[p];
      f.pairflg = pl.state;
      f.rounds = (uint8_t)((!pl.idle && !pl.lone ? 3 : 0) | ((pl.wants_c || pl.lone) && !pl.idle ? 4 : 0) | (pl.wants_d && !pl.lone ? 8 : 0));
      for (int w = 0; w < 2; w++) {
        const size_t id = 2 * (size_t)p + (size_t)w;
        uint32_t nlive = 0;
        if (out->blk.packed.size(id) >= 8) memcpy(&nlive, out->blk.packed.data(id) + 4, 4);          // second header word of Table::pack
        f.nali[w] = (uint16_t)(nlive > 65535 ? 65535 : nlive);
      }
    }
  });
  out->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  return SMALTGPU_OK;
}

extern "C" int smaltgpu_map_pairs(smaltgpu_mapper *m, const uint8_t *bases1, const uint8_t *quals1, const uint64_t *read_off1, const uint8_t *bases2,
                                  const uint8_t *quals2, const uint64_t *read_off2, uint32_t npairs, const smaltgpu_params *par, const smaltgpu_pair_opts *po,
                                  smaltgpu_pairs *out) {
  return map_pairs(m, nullptr, bases1, quals1, read_off1, bases2, quals2, read_off2, npairs, par, po, out);
}

extern "C" int smaltgpu_map_pairs_resident(smaltgpu_mapper *m, const smaltgpu_resident_reads *src, const uint8_t *bases1, const uint8_t *quals1, const uint8_t *bases2,
                                           const uint8_t *quals2, uint32_t npairs, const smaltgpu_params *par, const smaltgpu_pair_opts *po, smaltgpu_pairs *out) {
  if (!src || !src->read_off[0] || !src->read_off[1] || src->nreads[0] < npairs || src->nreads[1] < npairs)
    return smaltgpu_set_error(SMALTGPU_EARG, "smaltgpu_map_pairs_resident: the resident batches do not hold the block");
  return map_pairs(m, src, bases1, quals1, src->read_off[0], bases2, quals2, src->read_off[1], npairs, par, po, out);
}

extern "C" int smaltgpu_pairs_host_times(const smaltgpu_pairs *p, double *ms, int n) {
  if (!p || !ms) return smaltgpu_set_error(SMALTGPU_EARG, "smaltgpu_pairs_host_times: null argument");
  for (int i = 0; i < n && i < smgpairs::PairBlock::H_NUM; i++) ms[i] = p->blk.host_ms[i];
  if (n > smgpairs::PairBlock::H_NUM) ms[smgpairs::PairBlock::H_NUM] = p->totals_ms;
  if (n > smgpairs::PairBlock::H_NUM + 1) ms[smgpairs::PairBlock::H_NUM + 1] = p->wall_ms;
  return smgpairs::PairBlock::H_NUM + 2;
}

extern "C" int smaltgpu_pairs_timers(const smaltgpu_pairs *p, double *kernel_ms, uint64_t *work) {
  if (!p) return smaltgpu_set_error(SMALTGPU_EARG, "smaltgpu_pairs_timers: null argument");
  if (kernel_ms) memcpy(kernel_ms, p->kernel_ms, sizeof(p->kernel_ms));
  if (work) memcpy(work, p->work, sizeof(p->work));
  return SMALTGPU_OK;
}

extern "C" int smaltgpu_pairs_info(const smaltgpu_pairs *p, uint32_t *npairs, const smaltgpu_pair_info **info, uint64_t *calls, double *round_ms) {
  if (!p) return smaltgpu_set_error(SMALTGPU_EARG, "smaltgpu_pairs_info: null argument");
  if (npairs) *npairs = p->blk.npairs;
  if (info) *info = p->info.data();
  for (int r = 0; r < 4; r++) { if (calls) calls[r] = p->calls[r]; if (round_ms) round_ms[r] = p->round_ms[r]; }
  return SMALTGPU_OK;
}
