// smg_post.cpp -- smaltgpu_postprocess: the post-call pass of smg_post.hpp over every read of a batch (SURVEY 8f N1;
// the reference runs resultSetSortAndAssignSequence, results.c:2022, read by read behind each mapSingleRead).
// Host code on worker threads: each worker owns one alignment table, fills it from the raw alignments of a read, runs the
// pass and appends the read's rows to its own output chunk; the chunks are stitched together in read order at the end.
// Compiled with g++ -ffp-contract=off (the mapping quality is double arithmetic that has to match the reference's gcc -O2).
#include <stdio.h>
#include <thread>
#include "../../include/smaltgpu.h"
#include "smg_postout.hpp"

namespace {
using smgpostout::Chunk;

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
    smgpostout::emit_table(ck, r, tb, oc);
  }
}

}  // namespace

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

  return smgpostout::stitch(pp, chunks, n, out, "smaltgpu_postprocess");
}
