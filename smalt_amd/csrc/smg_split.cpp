// smg_split.cpp -- smaltgpu_map_split: split reads (smalt map -p) through the library.  The two rounds of smg_split.hpp run as
// smaltgpu_map_batch over the whole batch and smaltgpu_map_batch_ctx over the reads that get a second call (their bases gathered
// into a batch of their own, seeds from the stretch the first alignment leaves uncovered).
// Compiled like smg_post.cpp (g++ -ffp-contract=off: the mapping quality is double arithmetic).
#include <string.h>
#include "../../include/smaltgpu.h"
#include "smg_split.hpp"

namespace {

struct DeviceExec {
  smaltgpu_mapper *m;
  const smaltgpu_params *par;
  std::string err;
  int rc = SMALTGPU_EINTERNAL;
  std::vector<uint8_t> bases, quals;
  std::vector<uint64_t> off;

  bool fail(int rv) { rc = rv; const char *e = smaltgpu_last_error(); err = e && *e ? e : "a mapping call failed"; return false; }
  bool first(const smgsplit::Input &in, smaltgpu_batch_out *o) {
    const int rv = smaltgpu_map_batch(m, in.bases, in.quals, in.off, in.n, par, o);
    // a read that failed on its own carries its code in stat[].errcode: the runner names it
    if (rv && !(SMALTGPU_IS_READ_ERROR(rv) && o->nreads == in.n)) return fail(rv);
    return true;
  }
  bool second(const smgsplit::Input &in, const uint32_t *ids, uint32_t n, const uint32_t *range, const int32_t *prev_max, smaltgpu_batch_out *o) {
    bases.clear(); quals.clear(); off.assign(1, 0);
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t r = ids[i];
      bases.insert(bases.end(), in.bases + in.off[r], in.bases + in.off[r + 1]);
      if (in.quals) quals.insert(quals.end(), in.quals + in.off[r], in.quals + in.off[r + 1]);
      off.push_back(bases.size());
    }
    if (bases.empty()) bases.push_back(0);
    smaltgpu_callctx cx;
    memset(&cx, 0, sizeof(cx));
    cx.seed_range = range; cx.prev_max = prev_max; cx.raw_alignments = 1;      // the set the call appends to is here: Table::take_call compares
    const int rv = smaltgpu_map_batch_ctx(m, bases.data(), in.quals ? quals.data() : nullptr, off.data(), n, par, &cx, o);
    if (rv && !(SMALTGPU_IS_READ_ERROR(rv) && o->nreads == n)) return fail(rv);
    return true;
  }
};

}  // namespace

extern "C" int smaltgpu_map_split(smaltgpu_mapper *m, smaltgpu_post *post, const uint8_t *bases, const uint8_t *quals, const uint64_t *read_off, uint32_t nreads,
                                  const smaltgpu_params *par, const smaltgpu_index *ix, int nthreads, smaltgpu_post_out *out,
                                  uint32_t *n_second_calls) {
  if (out) memset(out, 0, sizeof(*out));
  if (n_second_calls) *n_second_calls = 0;
  if (!m || !post || !bases || !read_off || !par || !ix || !out) return smaltgpu_set_error(SMALTGPU_EARG, "smaltgpu_map_split: null argument");
  if (!(par->rmapflg & SMALTGPU_FLG_NOSHRTINFO)) return smaltgpu_set_error(SMALTGPU_EARG, "smaltgpu_map_split: split reads are mapped with the long hit info (SMALTGPU_FLG_NOSHRTINFO | SMALTGPU_FLG_SENSITIVE, smalt.c:508)");
  smaltgpu_index_desc ds;
  const char *const *names; const uint64_t *sop; int64_t nseq;
  if (smaltgpu_index_info(ix, &ds) || smaltgpu_index_seqnames(ix, &names, &sop, &nseq)) return smaltgpu_set_error(SMALTGPU_EARG, "smaltgpu_map_split: index");
  const uint32_t *packed_host = (par->rmapflg & SMALTGPU_FLG_SEQBYSEQ) ? nullptr : smaltgpu_index_packed_host(ix);     // concatenated mode: alignments across junctions are cut
  if (!(par->rmapflg & SMALTGPU_FLG_SEQBYSEQ) && !packed_host) return SMALTGPU_ENODEV;
  static thread_local smgsplit::Runner run;            // arenas are re-used from batch to batch
  DeviceExec ex{m, par};
  smgsplit::Input in{bases, quals, read_off, nreads};
  smgsplit::Setup su;
  su.map = *par; su.sop = sop; su.nseq = nseq; su.packed_host = packed_host;
  su.k = (int)ds.k; su.s = (int)ds.s; su.nthreads = nthreads;
  if (!run.run(ex, in, su, post, out)) return smaltgpu_set_error(run.error_code ? run.error_code : SMALTGPU_EINTERNAL, ("smaltgpu_map_split: " + run.error).c_str());
  if (n_second_calls) *n_second_calls = run.n_second;
  return SMALTGPU_OK;
}
