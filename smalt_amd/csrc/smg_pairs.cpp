// smg_pairs.cpp -- smaltgpu_map_pairs: a block of read pairs through the rounds of smg_pairrun.hpp with the device behind
// them (replaces rmapPair, rmap.c:1744-2112, for a block).  Host code: every round gathers its reads into one
// batch for smaltgpu_map_batch_ctx, worker threads do the post-call passes and the decisions between the rounds.
// Compiled with g++ -ffp-contract=off like smg_post.cpp (mapping qualities are double arithmetic).
#include <atomic>
#include <chrono>
#include <mutex>
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
  // Several sub-blocks take turns on the one mapper (map_pairs): a device call holds the gate from its launch to the copy of its
  // results out of the mapper's buffers, so that the host work behind a round of one sub-block runs beside the device work of another
  std::mutex *gate = nullptr;
  std::vector<smaltgpu_result> k_res; std::vector<uint64_t> k_off; std::vector<uint8_t> k_str; std::vector<smaltgpu_readstat> k_stat;
  void keep(smaltgpu_batch_out *o) {
    const uint32_t n = o->nreads;
    const uint64_t nres = o->res_off[n];
    size_t nstr = 0;
    for (uint64_t j = 0; j < nres; j++) { const size_t e = (size_t)o->res[j].stroffs + o->res[j].strlen; if (e > nstr) nstr = e; }
    k_res.assign(o->res, o->res + nres); k_off.assign(o->res_off, o->res_off + n + 1); k_str.assign(o->diffstr, o->diffstr + nstr); k_stat.assign(o->stat, o->stat + n);
    if (k_res.empty()) k_res.resize(1);
    if (k_str.empty()) k_str.resize(1);
    o->res = k_res.data(); o->res_off = k_off.data(); o->diffstr = k_str.data(); o->stat = k_stat.data();
  }
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
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t w = ids[i] & 1, p = ids[i] >> 1;
      const uint64_t len = off[i + 1] - off[i];
      memcpy(bases.data() + off[i], in.bases[w] + in.off[w][p], len);
      if (with_quals()) memcpy(quals.data() + off[i], in.quals[w] + in.off[w][p], len);
    }
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
    std::unique_lock<std::mutex> turn;
    if (gate) turn = std::unique_lock<std::mutex>(*gate);
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
    std::unique_lock<std::mutex> turn;
    if (gate) turn = std::unique_lock<std::mutex>(*gate);
    const auto t0 = std::chrono::steady_clock::now();
    if (rd.n > cap_reads) { rc = SMALTGPU_EARG; err = "the block holds more pairs than the mapper's batch size"; return false; }
    smaltgpu_callctx cx;
    memset(&cx, 0, sizeof(cx));
    cx.iv_off = rd.iv_off; cx.iv = rd.iv; cx.min_swatscor = rd.min_score; cx.prev_max = rd.prev_max; cx.fine_index = rd.kind == ROUND_FINE;
    int rv;
    if (resident) rv = smaltgpu_map_batch_ctx_resident(m, resident, rd.ids, rd.n, &par, &cx, o);
    else {
      gather(rd.ids, rd.n);
      rv = smaltgpu_map_batch_ctx(m, bases.data(), with_quals() ? quals.data() : nullptr, off.data(), rd.n, &par, &cx, o);
    }
    // a read that failed on its own carries its code in stat[].errcode; the runner names it
    if (rv && !((rv == SMALTGPU_ECAP || rv == SMALTGPU_EINTERNAL) && o->nreads == rd.n)) return fail_with(err, rv);
    tally(rd.kind);
    if (gate) keep(o);
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
  bp.k = ds.k; bp.sop = ds.sop; bp.nseq = ds.nseq;
  // alignments can cross sequence junctions only in concatenated mode: the pieces are scored against a host copy of the reference
  bp.packed_host = (par->rmapflg & SMALTGPU_FLG_SEQBYSEQ) ? nullptr : smaltgpu_index_packed_host(ix);
  if (!(par->rmapflg & SMALTGPU_FLG_SEQBYSEQ) && !bp.packed_host) return SMALTGPU_ENODEV;
  if (bp.packed_host && (!bases1 || !bases2)) return smaltgpu_set_error(SMALTGPU_EARG, "smaltgpu_map_pairs: concatenated mode needs the read bases in host memory too (pieces of alignments across sequences are scored on the host)");
  bp.nthreads = po->nthreads < 1 ? 1 : po->nthreads;
  uint32_t cap_reads = 0, maxlen = 0;
  uint64_t cap_bases = 0;
  if (smaltgpu_mapper_capacity(m, &cap_reads, &maxlen, &cap_bases)) return SMALTGPU_EARG;
  // Sub-blocks: the block is cut into up to four pieces that two runners take in turn; the device calls of a piece hold the
  // mapper (DeviceExec::gate), its host work (post-call passes, probe, intervals) runs while the device works for another piece.
  // SMALTGPU_PAIR_SUBBLOCKS = 2 or 4 splits a large block into sub-blocks that take turns on the mapper (off by default: four
  // sub-blocks of a 2^20-pair block measured 7 % slower than the whole block -- smaller device calls cost more than the overlap
  // gains; a caller that wants the host phases hidden runs two blocks on two mappers, as bench.py's two_streams line does)
  static const uint32_t want_sub = getenv("SMALTGPU_PAIR_SUBBLOCKS") ? (uint32_t)atoi(getenv("SMALTGPU_PAIR_SUBBLOCKS")) : 1u;
  const uint32_t nsub = (want_sub >= 4 && npairs >= 65536) ? 4u : ((want_sub >= 2 && npairs >= 8192) ? 2u : 1u);
  std::vector<PairBlock> sub(nsub);
  std::vector<int> sub_rc(nsub, SMALTGPU_OK);
  std::mutex gate;
  std::mutex tally_mu;
  memset(out->kernel_ms, 0, sizeof(out->kernel_ms)); memset(out->work, 0, sizeof(out->work));
  for (int r = 0; r < 4; r++) { out->round_ms[r] = 0; out->calls[r] = 0; }
  out->totals_ms = 0;
  std::atomic<uint32_t> next(0);
  auto runner = [&]() {
    for (;;) {
      const uint32_t j = next.fetch_add(1);
      if (j >= nsub) return;
      const uint32_t lo = (uint32_t)((uint64_t)npairs * j / nsub), hi = (uint32_t)((uint64_t)npairs * (j + 1) / nsub);
      BlockInput sin = in;
      sin.off[0] = in.off[0] + lo; sin.off[1] = in.off[1] + lo; sin.npairs = hi - lo;
      smaltgpu_resident_reads sres;
      if (resident) {
        sres = *resident;
        for (int w = 0; w < 2; w++) { sres.d_read_off[w] = resident->d_read_off[w] + lo; sres.read_off[w] = resident->read_off[w] + lo; sres.nreads[w] = hi - lo; }
      }
      BlockParams sbp = bp;
      sbp.nthreads = nsub > 1 ? (bp.nthreads + 1) / 2 : bp.nthreads;
      DeviceExec ex{m, sin, sbp.map, cap_reads, cap_bases};
      ex.resident = resident ? &sres : nullptr;
      ex.gate = nsub > 1 ? &gate : nullptr;
      memset(ex.kernel_ms, 0, sizeof(ex.kernel_ms)); memset(ex.work, 0, sizeof(ex.work));
      const bool ok = sub[j].run(ex, sin, sbp);
      if (!ok) sub_rc[j] = ex.rc != SMALTGPU_OK ? ex.rc : SMALTGPU_EINTERNAL;
      std::lock_guard<std::mutex> lk(tally_mu);
      for (int r = 0; r < 4; r++) { out->round_ms[r] += ex.ms[r]; out->calls[r] += sub[j].nrounds.size() == 4 ? sub[j].nrounds[(size_t)r] : 0; }
      for (int r = 0; r < 5; r++) { for (int i = 0; i < 16; i++) out->kernel_ms[r][i] += ex.kernel_ms[r][i]; for (int i = 0; i < 32; i++) out->work[r][i] += ex.work[r][i]; }
      out->totals_ms += ex.totals_wall;
    }
  };
  if (nsub == 1) runner();
  else { std::thread t2(runner); runner(); t2.join(); }
  // one block again
  PairBlock &B = out->blk;
  B.npairs = npairs; B.error.clear();
  B.packed.assign((size_t)2 * npairs, std::vector<uint8_t>());
  B.plan.assign(npairs, PairPlan());
  B.nrounds.assign(4, 0);
  for (double &h : B.host_ms) h = 0;
  for (uint32_t j = 0; j < nsub; j++) {
    if (sub_rc[j] != SMALTGPU_OK) return smaltgpu_set_error(sub_rc[j], ("smaltgpu_map_pairs: " + sub[j].error).c_str());
    const uint32_t lo = (uint32_t)((uint64_t)npairs * j / nsub);
    for (size_t i = 0; i < sub[j].packed.size(); i++) B.packed[(size_t)2 * lo + i].swap(sub[j].packed[i]);
    for (size_t i = 0; i < sub[j].plan.size(); i++) B.plan[(size_t)lo + i] = sub[j].plan[i];
    for (int r = 0; r < 4 && sub[j].nrounds.size() == 4; r++) B.nrounds[(size_t)r] += sub[j].nrounds[(size_t)r];
    for (int h = 0; h < PairBlock::H_NUM; h++) B.host_ms[h] += sub[j].host_ms[h];
  }
  // the summary: flags, rounds, surviving alignments
  out->info.assign(npairs ? npairs : 1, smaltgpu_pair_info());
  PairBlock::spread(npairs, bp.nthreads, [&](uint32_t lo, uint32_t hi, int) {
    for (uint32_t p = lo; p < hi; p++) {
      const PairPlan &pl = out->blk.plan[p];
      smaltgpu_pair_info &f = out->info[p];
      f.pairflg = pl.state;
      f.rounds = (uint8_t)((!pl.idle && !pl.lone ? 3 : 0) | ((pl.wants_c || pl.lone) && !pl.idle ? 4 : 0) | (pl.wants_d && !pl.lone ? 8 : 0));
      for (int w = 0; w < 2; w++) {
        const std::vector<uint8_t> &rest = out->blk.packed[2 * (size_t)p + (size_t)w];
        uint32_t nlive = 0;
        if (rest.size() >= 8) memcpy(&nlive, rest.data() + 4, 4);          // second header word of Table::pack
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
