// tests/hostemu/split_check.cpp -- TEST DRIVER (no device): the product's split-read runner (smalt_amd/csrc/smg_split.hpp: the first
// call of every read, the stretch of the second call by mapSecondary's rule, the post-call passes behind both) with the mapping
// calls REPLAYED from a `refdump -s` record of the reference's own rmapSingle under RMAPFLG_SPLIT.  The runner must ask for a second
// call exactly where the reference made one, with the set's running score maxima the reference passed; the alignments the
// reference's calls added go into the product's tables, and the sets the runner leaves come out on stdout in the `PS`/`RF`/`SO`/
// `SS`/`SG` lines of the record for comparison with what the reference left (tests/test_hostemu.py).
//
//   split_check <refdump.txt> <reads.fq> <seqinfo.txt> k=<word length> s=<step> threads=<n>
#include <map>
#include "../../smalt_amd/csrc/smg_split.hpp"
#include "dump_record.hpp"

struct ReplayExec {
  std::vector<RecPair> &rec;
  std::string err;
  int rc = SMALTGPU_EINTERNAL;
  std::vector<smaltgpu_result> res;
  std::vector<uint64_t> res_off;
  std::vector<uint8_t> pool;
  std::vector<smaltgpu_readstat> stat;
  uint32_t second_calls = 0;

  void begin(uint32_t n) { res.clear(); pool.clear(); stat.assign(n ? n : 1, smaltgpu_readstat()); res_off.assign((size_t)n + 1, 0); }
  void give(uint32_t i, RecCall &c) {
    for (size_t j = 0; j < c.res.size(); j++) {
      smaltgpu_result r = c.res[j];
      r.stroffs = (uint32_t)pool.size(); r.strlen = (uint32_t)c.strs[j].size();
      r.reverse |= SMALTGPU_RES_CANDFIRST;                  // the record holds what the set kept: every alignment its own candidate
      pool.insert(pool.end(), c.strs[j].begin(), c.strs[j].end());
      res.push_back(r);
    }
    smaltgpu_readstat &st = stat[i];
    st.swatscor_max = c.rx[1]; st.swatscor_2ndmax = c.rx[2]; st.n_ali_done = c.rx[3]; st.n_ali_tot = c.rx[4]; st.n_hits_used = (uint32_t)c.rx[5]; st.n_hits_tot = (uint32_t)c.rx[6];
    st.max1scor = c.max1; st.nres = (uint32_t)c.res.size();
    c.used = true;
  }
  void end(uint32_t n, smaltgpu_batch_out *o) {
    if (res.empty()) res.resize(1);
    if (pool.empty()) pool.resize(1);
    o->nreads = n; o->res_off = res_off.data(); o->res = res.data(); o->diffstr = pool.data(); o->stat = stat.data();
  }
  bool first(const smgsplit::Input &in, smaltgpu_batch_out *o) {
    begin(in.n);
    for (uint32_t i = 0; i < in.n; i++) {
      if (i < rec.size() && !rec[i].calls.empty()) give(i, rec[i].calls[0]);
      res_off[(size_t)i + 1] = res.size();
    }
    end(in.n, o);
    return true;
  }
  bool second(const smgsplit::Input &, const uint32_t *ids, uint32_t n, const uint32_t *, const int32_t *prev_max, smaltgpu_batch_out *o) {
    begin(n);
    char msg[200];
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t r = ids[i];
      if (r >= rec.size() || rec[r].calls.size() < 2) { snprintf(msg, sizeof(msg), "read %u: the runner asks for a second call, the reference made none", r); err = msg; return false; }
      RecCall &c = rec[r].calls[1];
      if (c.prevmax[0] != prev_max[2 * i] || c.prevmax[1] != prev_max[2 * i + 1]) {
        snprintf(msg, sizeof(msg), "read %u: running maxima %d,%d into the second call, the reference passed %d,%d", r, prev_max[2 * i], prev_max[2 * i + 1], c.prevmax[0], c.prevmax[1]);
        err = msg; return false;
      }
      give(i, c);
      res_off[(size_t)i + 1] = res.size();
      second_calls++;
    }
    end(n, o);
    return true;
  }
};

int main(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "usage: split_check <refdump.txt> <reads.fq> <seqinfo.txt> key=value ...\n"); return 2; }
  std::map<std::string, std::string> kv;
  for (int a = 4; a < argc; a++) { const std::string s(argv[a]); const size_t e = s.find('='); if (e != std::string::npos) kv[s.substr(0, e)] = s.substr(e + 1); }
  auto geti = [&](const char *key, int dflt) { return kv.count(key) ? atoi(kv[key].c_str()) : dflt; };
  std::vector<RecPair> rec = load_dump(argv[1]);
  std::string text;
  { std::ifstream f(argv[2], std::ios::binary); std::stringstream ss; ss << f.rdbuf(); text = ss.str(); }
  smaltgpu_reads *rs = smaltgpu_reads_create();
  smaltgpu_reads_view v;
  if (smaltgpu_reads_parse(rs, text.data(), text.size(), 1, 0, 1, &v)) { fprintf(stderr, "split_check: %s\n", smaltgpu_last_error()); return 1; }
  if (v.nreads != rec.size()) { fprintf(stderr, "split_check: %u reads, %zu records\n", v.nreads, rec.size()); return 1; }
  std::vector<uint64_t> sop(1, 0);
  { std::ifstream f(argv[3]); std::string nm; unsigned long long len; while (f >> nm >> len) sop.push_back(sop.back() + len); }
  smgsplit::Input in{v.bases, v.has_qual ? v.quals : nullptr, v.read_off, v.nreads};
  smgsplit::Setup su;
  memset(&su.map, 0, sizeof(su.map));
  su.map.match = 1; su.map.mismatch = -2; su.map.gap_init = -4; su.map.gap_ext = -3;
  su.sop = sop.data(); su.nseq = (int64_t)sop.size() - 1; su.packed_host = nullptr; su.k = geti("k", 13); su.s = geti("s", 6); su.nthreads = geti("threads", 1);
  ReplayExec ex{rec};
  smgsplit::Runner run;
  smaltgpu_post *post = smaltgpu_post_create();
  smaltgpu_post_out out;
  if (!run.run(ex, in, su, post, &out)) { fprintf(stderr, "split_check: %s\n", run.error.c_str()); return 1; }
  uint32_t recorded = 0;
  for (size_t r = 0; r < rec.size(); r++) {
    if (rec[r].calls.size() > 1) recorded++;
    for (const RecCall &c : rec[r].calls) if (!c.used) { fprintf(stderr, "split_check: read %zu: a call of the reference was not made\n", r); return 1; }
  }
  if (recorded != ex.second_calls || run.n_second != recorded) { fprintf(stderr, "split_check: %u second calls, the reference made %u\n", ex.second_calls, recorded); return 1; }
  const unsigned mask = ~(0x10u | 0x20u | 0x200u);              // output filter and report bits (rd_results.c)
  for (uint32_t i = 0; i < out.nreads; i++) {
    const uint64_t a = out.res_off[i], e = out.res_off[i + 1];
    printf("PS %u %u %d %u\n", (unsigned)(e - a), (unsigned)(out.sort_off[i + 1] - out.sort_off[i]), out.qsegno[i], out.setstatus[i]);
    for (uint64_t j = a; j < e; j++) {
      const smaltgpu_post_result &r = out.res[j];
      printf("RF %u %u %d %d %.17g %u %u %llu %llu %lld %d %d %d ", (unsigned)(j - a), r.status & mask, r.swatscor, r.mapscor, r.prob, r.q_start, r.q_end,
             (unsigned long long)r.s_start, (unsigned long long)r.s_end, (long long)r.sidx, (int)r.rsltx, (int)r.qsegx, (int)r.swrank);
      for (uint32_t b = 0; b < r.strlen; b++) printf("%02x", (unsigned)out.diffstr[r.stroffs + b]);
      printf("\n");
    }
    printf("SO");
    for (uint64_t j = out.sort_off[i]; j < out.sort_off[i + 1]; j++) printf(" %d", out.sortr[j]);
    printf("\n");
    if (out.sort_off[i + 1] > out.sort_off[i] && out.seg_off[i + 1] > out.seg_off[i]) {
      printf("SS");
      for (uint64_t j = out.sort_off[i]; j < out.sort_off[i + 1]; j++) printf(" %d", out.segsrtr[j]);
      printf("\nSG");
      for (uint64_t j = out.seg_off[i]; j < out.seg_off[i + 1]; j++) printf(" %d", out.segnor[j]);
      printf("\n");
    }
  }
  smaltgpu_post_free(post);
  smaltgpu_reads_free(rs);
  return 0;
}
