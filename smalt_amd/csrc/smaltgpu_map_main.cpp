// smaltgpu-map -- `smalt map` for single reads and read pairs on top of the C ABI of include/smaltgpu.h, file to file:
//   FASTQ/FASTA text --smaltgpu_reads_parse--> batch --smaltgpu_map_batch (GPU)--> raw alignments
//   --smaltgpu_postprocess--> mapping qualities, order --smaltgpu_report_emit--> CIGAR / SAM text;
//   two files of mates: --smaltgpu_map_pairs (the rounds of rmapPair on the GPU)--> --smaltgpu_report_emit_pairs--> text.
// The option letters, defaults and derived flags follow the reference's `smalt map` (menu.c:1147-1160 defaults,
// :1340-1345 -d, :1487-1497 -r; smalt.c:209-245 output formats, :490-503 result flags, :608-615 default -m) so that the
// same command line prints the same lines (tests/test_gpu_report.py).  Host code only: it needs libsmaltgpu.so, not hipcc.
//
// Three stages run side by side: one thread parses the next block of the (memory-mapped) input, two threads own a
// mapper each and take blocks in turn (host copies of one block overlap the kernels of the other), the main thread
// formats and writes the blocks in input order.
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/smaltgpu.h"

namespace {

const char VERSION[] = "0.2";

[[noreturn]] void die(const char *what, const char *detail = nullptr) {
  fprintf(stderr, "smaltgpu-map: %s%s%s\n", what, detail ? ": " : "", detail ? detail : "");
  exit(1);
}

void usage() {
  fprintf(stderr,
          "usage: smaltgpu-map [options] <index prefix> <reads.fq|reads.fa>[.gz] [<mates.fq|mates.fa>[.gz]]\n"
          "  -f <fmt>   cigar (default) | ssaha | sam | samsoft, SAM modifiers behind a colon: nohead, clip, x (e.g. sam:nohead,x)\n"
          "  -o <file>  output file (default: standard output)\n"
          "  -m <int>   minimum Smith-Waterman score (default: word length + step - 1)\n"
          "  -d <int>   report alignments within <int> of the best score; -1: all (default 0: best only)\n"
          "  -r <int>   seed for the random choice among equally good alignments; < 0: such reads are reported unmapped;\n"
          "             0 (default): seeded by the clock\n"
          "  -y <num>   minimum identity (fraction of the read if <= 1, else bases)\n"
          "  -c <num>   minimum k-mer cover (fraction of the read if <= 1, else bases)\n"
          "  -x         more sensitive search (all seeds, deeper candidate lists)\n"
          "  -p         split reads: a second alignment for the part of a read (or mate) its best alignment leaves uncovered\n"
          "  -q <int>   base quality threshold for k-mer words\n"
          "  -S <spec>  alignment scores, e.g. match=1,subst=-2,gapopen=-4,gapext=-3 (the default; any subset)\n"
          "  -n <int>   host threads for parsing, post-processing and formatting (default: up to 16)\n"
          "  -B <int>   reads per GPU batch (default 262144)\n"
          "  -g <list>  devices, e.g. 0 or 0,1,2,3 (default 0): the index is read once and copied device to device, every device gets\n"
          "             two mappers (SMALTGPU_MAP_WORKERS: 1-4), blocks go to whichever is free\n"
          "  -i <int>   maximum insert size of read pairs (default 500); -j <int> minimum insert size (default 0)\n"
          "  -l <lib>   pair library: pe (default) | mp | pp\n"
          "with two read files the reads are mapped as pairs (read i of the first with read i of the second file);\n"
          "-w, -a and insert-size histograms (-g) go through the bound reference program (INTEGRATION.md)\n");
  exit(2);
}

struct Block {                                  // one block of reads (or pairs) on its way through the stages
  smaltgpu_reads *rs = nullptr, *rs2 = nullptr;
  smaltgpu_reads_view v, v2;                    // v2: the mates
  smaltgpu_pairs *pairs = nullptr;
  uint32_t maxlen = 0;
  std::vector<uint32_t> hitlen;                 // serial-order mode: per read the longest read (>= k bases) of the input so far
  int state = 0;                                // 0 free, 1 parsed, 2 mapped + post-processed, 3 end of input
  int worker = -1;
  smaltgpu_batch_out raw;
  smaltgpu_post_out post;
  std::string err;
};

// The text of the input: a memory-mapped file, or the inflated stream of a gzip file (the reference reads FASTQ through
// zlib's gzgets when it is built with it, sequence.c:1108).  window() hands out at least `want` bytes of text that has not
// been consumed yet (less at the end of the input).
struct Source {
  const char *map = nullptr; uint64_t maplen = 0, pos = 0;      // plain file
  bool gz = false, gz_end = false;
  z_stream zs;
  std::vector<char> buf; uint64_t boff = 0;                     // inflated text not consumed yet: buf[boff..)
  void open(const char *path) {
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) die("cannot open", path);
    struct stat sb;
    if (fstat(fd, &sb)) die("cannot stat", path);
    maplen = (uint64_t)sb.st_size;
    map = maplen ? (const char *)mmap(nullptr, maplen, PROT_READ, MAP_PRIVATE, fd, 0) : "";
    if (maplen && map == (const char *)MAP_FAILED) die("cannot map", path);
    if (maplen) (void)madvise((void *)map, maplen, MADV_SEQUENTIAL);
    gz = maplen >= 2 && (unsigned char)map[0] == 0x1f && (unsigned char)map[1] == 0x8b;
    if (gz) {
      memset(&zs, 0, sizeof(zs));
      if (inflateInit2(&zs, 15 + 32) != Z_OK) die("zlib");          // + 32: gzip or zlib header
      zs.next_in = (Bytef *)map; zs.avail_in = 0;
    }
  }
  const char *window(uint64_t want, uint64_t *got, bool *last) {
    if (!gz) { *got = want < maplen - pos ? want : maplen - pos; *last = pos + *got >= maplen; return map + pos; }
    if (boff > (64u << 20) || (boff && boff == buf.size())) { buf.erase(buf.begin(), buf.begin() + (ptrdiff_t)boff); boff = 0; }
    while (!gz_end && buf.size() - boff < want) {
      const size_t old = buf.size(), add = want > (8u << 20) ? (size_t)want : (size_t)(8u << 20);
      buf.resize(old + add);
      zs.next_out = (Bytef *)buf.data() + old; zs.avail_out = (uInt)add;
      while (zs.avail_out && !gz_end) {
        if (!zs.avail_in) {
          const uint64_t left = maplen - pos;
          if (!left) { gz_end = true; break; }
          zs.avail_in = (uInt)(left < (1u << 30) ? left : (1u << 30)); zs.next_in = (Bytef *)map + pos; pos += zs.avail_in;
        }
        const int rv = inflate(&zs, Z_NO_FLUSH);
        if (rv == Z_STREAM_END) {                                 // a gzip file may hold several members
          if (!zs.avail_in && pos >= maplen) gz_end = true; else if (inflateReset(&zs) != Z_OK) die("zlib");
        } else if (rv != Z_OK && rv != Z_BUF_ERROR) die("corrupt gzip input");
        else if (rv == Z_BUF_ERROR && !zs.avail_in && pos >= maplen) gz_end = true;
      }
      buf.resize(old + add - zs.avail_out);
    }
    *got = want < buf.size() - boff ? want : buf.size() - boff;
    *last = gz_end && *got == buf.size() - boff;
    return buf.data() + boff;
  }
  void consume(uint64_t n) { if (gz) boff += n; else pos += n; }
  bool done() const { return gz ? (gz_end && boff >= buf.size()) : pos >= maplen; }
};

}  // namespace

int main(int argc, char **argv) {
  const char *fmt = "cigar", *oufil = nullptr, *scorespec = nullptr;
  int m = -1, d = 0, seed = 0, q = 0, nthreads = 0, ins_max = 500, ins_min = 0, lib = SMALTGPU_LIB_PE;
  std::vector<int> devices;
  bool d_given = false, randrepeat = true, exhaustive = false, split = false;
  double minid = 0.0, mincover = 0.0;
  long batch = 262144;
  int a = 1;
  for (; a < argc && argv[a][0] == '-' && argv[a][1]; a++) {
    const char o = argv[a][1];
    if (o == 'x' && !argv[a][2]) { exhaustive = true; continue; }
    if (o == 'p' && !argv[a][2]) { split = true; continue; }
    if (argv[a][2] || !strchr("fomdrycqnBgijlS", o)) {
      if (strchr("wTFa", o) && !argv[a][2]) die("option not supported by this program (use the bound `smalt map`, INTEGRATION.md)", argv[a]);
      usage();
    }
    if (a + 1 >= argc) usage();
    const char *val = argv[++a];
    switch (o) {
      case 'f': fmt = val; break;
      case 'o': oufil = val; break;
      case 'm': m = atoi(val); if (m < 0) die("-m out of range"); break;
      case 'd': d = atoi(val); d_given = true; break;                                   // MENUFLAG_RELSCOR (menu.c:1343)
      case 'r': seed = atoi(val); randrepeat = seed >= 0; break;                        // menu.c:1487-1497
      case 'y': minid = atof(val); if (minid < 0) die("-y out of range"); break;
      case 'c': mincover = atof(val); if (mincover < 0) die("-c out of range"); break;
      case 'q': q = atoi(val); break;
      case 'n': nthreads = atoi(val); break;
      case 'i': ins_max = atoi(val); break;
      case 'j': ins_min = atoi(val); break;
      case 'l': lib = !strcmp(val, "pe") ? SMALTGPU_LIB_PE : !strcmp(val, "mp") ? SMALTGPU_LIB_MP : !strcmp(val, "pp") ? SMALTGPU_LIB_PP : 0; if (!lib) die("-l: pe, mp or pp"); break;
      case 'S': scorespec = val; break;
      case 'B': batch = atol(val); if (batch < 1 || batch > (1L << 20)) die("-B out of range (1 .. 1048576)"); break;
      case 'g': for (const char *c = val; *c;) { devices.push_back(atoi(c)); while (*c && *c != ',') c++; if (*c) c++; } break;
    }
  }
  if (argc - a != 2 && argc - a != 3) usage();
  const char *prefix = argv[a], *readfil = argv[a + 1], *matefil = argc - a == 3 ? argv[a + 2] : nullptr;
  const bool paired = matefil != nullptr;
  if (paired && ins_min > ins_max) die("-j above -i");
  if (nthreads < 1) { nthreads = (int)std::thread::hardware_concurrency(); if (nthreads > 16) nthreads = 16; if (nthreads < 1) nthreads = 1; }

  smaltgpu_report_opts ro;
  memset(&ro, 0, sizeof(ro));
  {                                                                                      // smalt.c:209-245, menu.c:940-1010
    std::string f(fmt), key = f.substr(0, f.find(':'));
    if (key == "cigar") ro.format = SMALTGPU_FMT_CIGAR;
    else if (key == "sam" || key == "samsoft") { ro.format = SMALTGPU_FMT_SAM; ro.modflags = SMALTGPU_REP_HEADER | SMALTGPU_REP_SOFTCLIP; }
    else if (key == "ssaha") ro.format = SMALTGPU_FMT_SSAHA;
    else die("output format not supported here (cigar, sam, samsoft, ssaha)", fmt);
    size_t p = f.find(':');
    while (p != std::string::npos && ro.format == SMALTGPU_FMT_SAM) {
      const size_t e = f.find(',', p + 1);
      const std::string mod = f.substr(p + 1, e == std::string::npos ? std::string::npos : e - p - 1);
      if (mod == "nohead") ro.modflags &= ~(uint32_t)SMALTGPU_REP_HEADER;
      else if (mod == "clip") ro.modflags &= ~(uint32_t)SMALTGPU_REP_SOFTCLIP;
      else if (mod == "x" || mod == "X") ro.modflags |= SMALTGPU_REP_XMISMATCH;
      else if (!mod.empty()) die("unknown SAM modifier", mod.c_str());
      p = e;
    }
  }
  ro.min_swscor = m >= 0 ? m : 18;                                                       // resultSetFilterData gets the menu's value (smalt.c:490, menu.c:599)
  ro.min_swscor_below_max = d;
  ro.min_identity = minid;
  if (!d) {                                                                              // smalt.c:495-504
    ro.outflags |= SMALTGPU_OUT_BEST;
    if (!d_given) { ro.outflags |= SMALTGPU_OUT_SINGLE; if (randrepeat) ro.outflags |= SMALTGPU_OUT_RANDSEL; }
  }
  if (ro.outflags & SMALTGPU_OUT_RANDSEL) srand48(seed <= 0 ? (long)time(nullptr) : (long)seed);     // RANSEED (randef.h:19)

  Source src, src2;                                                                      // input: plain or gzip text
  src.open(readfil);
  if (paired) src2.open(matefil);
  FILE *ou = oufil ? fopen(oufil, "w") : stdout;
  if (!ou) die("cannot write", oufil);
  static char oubuf[1 << 22];
  setvbuf(ou, oubuf, _IOFBF, sizeof(oubuf));

  const auto t_start = std::chrono::steady_clock::now();
  if (devices.empty()) devices.push_back(0);
  if (devices.size() > 16) die("-g: at most 16 devices");
  const int ndev = (int)devices.size();
  std::vector<smaltgpu_index *> ixs((size_t)ndev, nullptr);
  if (smaltgpu_index_load(&ixs[0], prefix, devices[0])) die("index", smaltgpu_last_error());
  for (int dv = 1; dv < ndev; dv++)                       // the reference's workers share one read-only index; here every device gets an image of it
    if (smaltgpu_index_clone(&ixs[(size_t)dv], ixs[0], devices[(size_t)dv])) die("index copy", smaltgpu_last_error());
  smaltgpu_index *ix = ixs[0];
  const auto t_index = std::chrono::steady_clock::now();
  const char *const *seqnames; const uint64_t *sop; int64_t nseq;
  if (smaltgpu_index_seqnames(ix, &seqnames, &sop, &nseq)) die("index", smaltgpu_last_error());
  smaltgpu_index_desc idesc;
  if (smaltgpu_index_info(ix, &idesc)) die("index", smaltgpu_last_error());
  smaltgpu_params par;
  smaltgpu_params_default(&par, ix);
  if (m >= 0) par.min_swatscor = m;
  if (scorespec) {               // -S match=1,subst=-2,gapopen=-4,gapext=-3: any subset, in any order (menu.c:671-701, :875-935, smalt.c:539-550)
    struct { const char *key; int32_t *to; int lo, hi; } slot[4] = {{"match", &par.match, 0, 127}, {"subst", &par.mismatch, -127, 0},
                                                                   {"gapopen", &par.gap_init, -127, 0}, {"gapext", &par.gap_ext, -127, 0}};
    const std::string spec(scorespec);
    for (size_t at = 0; at < spec.size();) {
      size_t end = spec.find(',', at);
      if (end == std::string::npos) end = spec.size();
      const std::string item = spec.substr(at, end - at);
      at = end + 1;
      if (item.empty()) continue;
      const size_t eq = item.find('=');
      const std::string key = item.substr(0, eq), num = eq == std::string::npos ? "" : item.substr(eq + 1);
      size_t dg = (!num.empty() && (num[0] == '+' || num[0] == '-')) ? 1 : 0;
      bool digits = !num.empty();
      for (size_t c = dg; c < num.size(); c++) if (!isdigit((unsigned char)num[c])) digits = false;
      if (!digits) die("-S: key=number expected", item.c_str());
      const int v = atoi(num.c_str());
      int hit = -1;
      for (int t = 0; t < 4; t++) if (key == slot[t].key) hit = t;
      if (hit < 0) die("-S: keys are match, subst, gapopen, gapext", item.c_str());
      if (v < slot[hit].lo || v > slot[hit].hi) die("-S: value out of range", item.c_str());
      *slot[hit].to = v;
    }
  }
  par.min_swatscor_below_max = d;
  if (d) par.rmapflg &= ~(uint32_t)SMALTGPU_FLG_BEST;
  if (exhaustive) par.rmapflg |= SMALTGPU_FLG_NOSHRTINFO | SMALTGPU_FLG_SENSITIVE;      // smalt.c:531-533
  if (split) { par.rmapflg |= SMALTGPU_FLG_SPLIT | SMALTGPU_FLG_NOSHRTINFO | SMALTGPU_FLG_SENSITIVE; ro.outflags |= SMALTGPU_OUT_SPLIT; }      // smalt.c:507-511 (RMAPFLG_SPLIT: smaltgpu_map_split)
  par.min_basqval = (uint8_t)q;
  if (mincover < 1.01) { par.min_cover = 0; par.min_cover_frac = mincover; } else { par.min_cover = (uint32_t)mincover; par.min_cover_frac = 0.0; }   // smalt.c:1113-1126
  smaltgpu_pair_opts po;
  po.insert_min = ins_min; po.insert_max = ins_max; po.library = lib; po.every_pair = exhaustive ? 1 : 0;                              // smalt.c:533 (-x: RMAPFLG_ALLPAIR)
  po.nthreads = nthreads > 2 ? nthreads / 2 : 1;
  const uint32_t *packed = (par.rmapflg & SMALTGPU_FLG_SEQBYSEQ) ? nullptr : smaltgpu_index_packed_host(ix);    // concatenated mode: alignments across junctions are cut
  if (!(par.rmapflg & SMALTGPU_FLG_SEQBYSEQ) && !packed) die("index", smaltgpu_last_error());

  smaltgpu_report *rep = smaltgpu_report_create();
  {
    const char *htxt; uint64_t hlen;
    if (smaltgpu_report_header(rep, seqnames, sop, nseq, &ro, "smaltgpu-map", VERSION, argc, (const char *const *)argv, &htxt, &hlen)) die("header", smaltgpu_last_error());
    if (hlen && fwrite(htxt, 1, hlen, ou) != hlen) die("write error");
  }

  enum { MAXWORK = 32 };
  // mappers per device: while one block is post-processed and waits for its turn to be printed, the others keep the GPU busy
  int per_dev = 2;
  if (const char *e = getenv("SMALTGPU_MAP_WORKERS")) { const int v = atoi(e); if (v >= 1 && v <= 4) per_dev = v; }
  if (per_dev * ndev > MAXWORK) per_dev = MAXWORK / ndev;
  const int NWORK = per_dev * ndev, NBLK = NWORK + 2;
  std::vector<Block> blk((size_t)NBLK);
  for (Block &b : blk) { b.rs = smaltgpu_reads_create(); if (paired) { b.rs2 = smaltgpu_reads_create(); b.pairs = smaltgpu_pairs_create(); } }
  // a failing call's text, never empty (the block must not pass for mapped when a call failed without a message)
  auto why = [](const char *what) { const char *e = smaltgpu_last_error(); return std::string(e && *e ? e : what); };
  std::mutex mu;
  std::condition_variable cv;
  uint64_t n_parsed = 0, n_taken = 0, n_written = 0;      // block serial numbers: block k lives in blk[k % NBLK]
  bool input_done = false, failed = false;
  uint64_t n_blocks_total = ~0ull;
  double t_parse = 0, t_create[MAXWORK] = {0}, t_map[MAXWORK] = {0}, t_post[MAXWORK] = {0};          // seconds per stage (SMALTGPU_MAP_VERBOSE)
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };

  // SMALTGPU_SERIAL_ORDER=1 (single reads): the output of a serial `smalt map -n 0` also where it depends on the order of reads of
  // different lengths (the capacity of the reference's hit list follows the longest read so far)
  const bool serial_order = getenv("SMALTGPU_SERIAL_ORDER") && atoi(getenv("SMALTGPU_SERIAL_ORDER")) != 0;
  uint32_t longest_so_far = 0;
  std::thread parser([&] {
    double bytes_per_read = 0.0;
    for (uint64_t k = 0;; k++) {
      Block &b = blk[k % NBLK];
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return failed || k < n_written + NBLK; }); if (failed) return; }
      uint64_t win = bytes_per_read > 0 ? (uint64_t)(bytes_per_read * (double)batch * 1.05) + 65536 : (1u << 20);
      const double tp0 = now();
      // up to `want` reads from a source; exact: the block must hold exactly that many unless the source ends (the mates of a block)
      auto parse_from = [&](Source &sc, smaltgpu_reads *rs, smaltgpu_reads_view *v, uint32_t want, bool exact) -> bool {
        bool last = false;
        uint64_t w = win;
        for (;;) {
          uint64_t got = 0;
          const char *text = sc.window(w, &got, &last);
          if (!got) { v->nreads = 0; return true; }
          if (smaltgpu_reads_parse(rs, text, got, last ? 1 : 0, want, nthreads, v)) {
            std::lock_guard<std::mutex> lk(mu); b.err = why("cannot parse the reads"); failed = true; cv.notify_all(); return false;
          }
          if (last || (exact ? v->nreads == want : v->nreads > 0)) return true;
          w *= 4;                                          // not enough complete records in the window
        }
      };
      if (!parse_from(src, b.rs, &b.v, (uint32_t)batch, false)) return;
      if (paired && b.v.nreads) {
        if (!parse_from(src2, b.rs2, &b.v2, b.v.nreads, true)) return;
        if (b.v2.nreads != b.v.nreads) { std::lock_guard<std::mutex> lk(mu); b.err = "the two read files hold different numbers of reads"; failed = true; cv.notify_all(); return; }
      } else if (paired) {
        bool last2 = false; uint64_t got2 = 0;
        (void)src2.window(1, &got2, &last2);
        if (got2) { std::lock_guard<std::mutex> lk(mu); b.err = "the two read files hold different numbers of reads"; failed = true; cv.notify_all(); return; }
      }
      if (!b.v.nreads) { std::lock_guard<std::mutex> lk(mu); n_blocks_total = k; input_done = true; cv.notify_all(); return; }
      t_parse += now() - tp0;
      bytes_per_read = (double)b.v.consumed / (double)b.v.nreads;
      src.consume(b.v.consumed);
      if (paired) src2.consume(b.v2.consumed);
      b.maxlen = 1;
      for (uint32_t i = 0; i < b.v.nreads; i++) { const uint32_t l = (uint32_t)(b.v.read_off[i + 1] - b.v.read_off[i]); if (l > b.maxlen) b.maxlen = l; }
      if (paired) for (uint32_t i = 0; i < b.v2.nreads; i++) { const uint32_t l = (uint32_t)(b.v2.read_off[i + 1] - b.v2.read_off[i]); if (l > b.maxlen) b.maxlen = l; }
      b.hitlen.clear();
      if (serial_order && !paired) {                       // the reference's one hit list only grows (hashhit.c:1280): smaltgpu_callctx.hitlist_len
        b.hitlen.resize(b.v.nreads);
        for (uint32_t i = 0; i < b.v.nreads; i++) {
          const uint32_t l = (uint32_t)(b.v.read_off[i + 1] - b.v.read_off[i]);
          if (l >= (uint32_t)idesc.k && l > longest_so_far) longest_so_far = l;
          b.hitlen[i] = longest_so_far;
        }
        if (longest_so_far > b.maxlen) b.maxlen = longest_so_far;
      }
      { std::lock_guard<std::mutex> lk(mu); b.state = 1; n_parsed = k + 1; cv.notify_all(); }
    }
  });

  struct Worker { smaltgpu_mapper *mp = nullptr; smaltgpu_post *post = nullptr; uint32_t cap_reads = 0, cap_len = 0; bool busy = false; };
  Worker wk[MAXWORK];
  auto work = [&](int w) {
    Worker &W = wk[w];
    W.post = smaltgpu_post_create();
    for (;;) {
      uint64_t k;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return failed || (!W.busy && (n_taken < n_parsed || (input_done && n_taken >= n_blocks_total))); });
        if (failed || (input_done && n_taken >= n_blocks_total)) return;
        k = n_taken++;
        W.busy = true;                                     // until the main thread has formatted this block (it reads the mapper's buffers)
      }
      Block &b = blk[k % NBLK];
      std::string err;
      double t0 = now(), t1;
      if (!W.mp || W.cap_reads < b.v.nreads || W.cap_len < b.maxlen) {
        if (W.mp) smaltgpu_mapper_free(W.mp);
        W.mp = nullptr;
        const uint32_t cr = b.v.nreads > (uint32_t)batch ? b.v.nreads : (uint32_t)batch, cl = (b.maxlen + 31u) & ~31u;
        smaltgpu_mapper_opts mo = {0, (uint32_t)(per_dev <= 2 ? 28 : 18)};      // the mappers of a device share its memory: candidate slots of 18-28 GB each
        if (smaltgpu_mapper_create_ex(&W.mp, ixs[(size_t)(w % ndev)], cr, cl > W.cap_len ? cl : W.cap_len, &mo)) err = why("cannot create a mapper");
        else { W.cap_reads = cr; W.cap_len = cl > W.cap_len ? cl : W.cap_len; if (!paired) smaltgpu_mapper_set_host_threads(W.mp, nthreads > 4 ? nthreads / 4 : 1); }
      }
      t1 = now(); t_create[w] += t1 - t0; t0 = t1;
      if (err.empty() && paired) {
        // the rounds of rmapPair for the block; the results rest in the block, the mapper is free for the next one
        const bool q2 = b.v.has_qual && b.v2.has_qual;
        if (smaltgpu_map_pairs(W.mp, b.v.bases, q2 ? b.v.quals : nullptr, b.v.read_off, b.v2.bases, q2 ? b.v2.quals : nullptr, b.v2.read_off, b.v.nreads, &par, &po, b.pairs))
          err = why("mapping the pairs failed");
        t1 = now(); t_map[w] += t1 - t0; t0 = t1;
      } else if (err.empty() && split) {
        // both calls of every read and the post-call passes between and behind them
        if (smaltgpu_map_split(W.mp, W.post, b.v.bases, b.v.has_qual ? b.v.quals : nullptr, b.v.read_off, b.v.nreads, &par, ixs[(size_t)(w % ndev)],
                               nthreads > 2 ? nthreads / 2 : 1, &b.post, nullptr)) err = why("mapping the split reads failed");
        t1 = now(); t_map[w] += t1 - t0; t0 = t1;
      } else if (err.empty()) {
        smaltgpu_callctx cx;
        memset(&cx, 0, sizeof(cx));
        cx.hitlist_len = b.hitlen.empty() ? nullptr : b.hitlen.data();
        const int rv = smaltgpu_map_batch_ctx(W.mp, b.v.bases, b.v.has_qual ? b.v.quals : nullptr, b.v.read_off, b.v.nreads, &par, cx.hitlist_len ? &cx : nullptr, &b.raw);
        if (rv && !(SMALTGPU_IS_READ_ERROR(rv) && b.raw.nreads == b.v.nreads)) err = why("mapping the reads failed");
        t1 = now(); t_map[w] += t1 - t0; t0 = t1;
        if (err.empty() && smaltgpu_postprocess(W.post, sop, nseq, &b.raw, b.v.bases, b.v.has_qual ? b.v.quals : nullptr, b.v.read_off, packed, &par,
                                                nthreads > 2 ? nthreads / 2 : 1, &b.post)) err = why("post-processing failed");
        t_post[w] += now() - t0;
      }
      { std::lock_guard<std::mutex> lk(mu); b.worker = w; if (paired) W.busy = false; if (!err.empty()) { b.err = err; failed = true; } b.state = 2; cv.notify_all(); }
    }
  };
  std::vector<std::thread> workers;
  for (int w = 0; w < NWORK; w++) workers.emplace_back(work, w);

  std::string failure;
  uint64_t nreads_total = 0;
  double t_emit = 0, t_write = 0, t_wait = 0;
  for (uint64_t k = 0;; k++) {
    Block &b = blk[k % NBLK];
    double t0 = now(), t1;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return failed || (input_done && k >= n_blocks_total) || (n_parsed > k && b.state == 2); });
      if (failed) { for (Block &x : blk) if (!x.err.empty()) failure = x.err; break; }
      if (input_done && k >= n_blocks_total) break;
    }
    const char *txt; uint64_t tl;
    t1 = now(); t_wait += t1 - t0; t0 = t1;
    if (paired ? smaltgpu_report_emit_pairs(rep, b.pairs, &b.v, &b.v2, seqnames, nseq, &ro, &po, nthreads, &txt, &tl)
               : smaltgpu_report_emit(rep, &b.post, split ? nullptr : &b.raw, &b.v, seqnames, nseq, &ro, nthreads, &txt, &tl)) failure = why("formatting the report failed");
    t1 = now(); t_emit += t1 - t0; t0 = t1;
    if (failure.empty() && tl && fwrite(txt, 1, tl, ou) != tl) failure = "write error";
    t_write += now() - t0;
    nreads_total += b.v.nreads;
    { std::lock_guard<std::mutex> lk(mu); b.state = 0; if (!paired) wk[b.worker].busy = false; n_written = k + 1; if (!failure.empty()) failed = true; cv.notify_all(); }
    if (!failure.empty()) break;
  }
  { std::lock_guard<std::mutex> lk(mu); if (!failure.empty()) failed = true; cv.notify_all(); }
  parser.join();
  for (std::thread &t : workers) t.join();
  if (failed && failure.empty()) for (Block &x : blk) if (!x.err.empty()) failure = x.err;
  for (Worker &W : wk) { if (W.mp) smaltgpu_mapper_free(W.mp); if (W.post) smaltgpu_post_free(W.post); }
  for (Block &b : blk) { smaltgpu_reads_free(b.rs); if (b.rs2) smaltgpu_reads_free(b.rs2); if (b.pairs) smaltgpu_pairs_free(b.pairs); }
  smaltgpu_report_free(rep);
  for (smaltgpu_index *x : ixs) smaltgpu_index_free(x);
  if (ou != stdout) { if (fclose(ou)) failure = "write error"; } else fflush(ou);
  if (failed || !failure.empty()) die("failed", failure.c_str());
  if (getenv("SMALTGPU_MAP_VERBOSE")) {
    const double ti = std::chrono::duration<double>(t_index - t_start).count(), tm = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_index).count();
    fprintf(stderr, "smaltgpu-map: %llu reads (or pairs), index load %.3f s, reads in to lines out %.3f s (%.0f reads/s)\n", (unsigned long long)nreads_total, ti, tm,
            tm > 0 ? (double)nreads_total / tm : 0.0);
    double tc = 0, tm_ = 0, tp = 0;
    for (int w = 0; w < NWORK; w++) { tc += t_create[w]; tm_ += t_map[w]; tp += t_post[w]; }
    fprintf(stderr, "smaltgpu-map: stages [s]: parse %.3f | %d workers on %d device(s), mean per worker: mapper set-up %.3f  map %.3f  post %.3f | wait %.3f emit %.3f write %.3f\n",
            t_parse, NWORK, ndev, tc / NWORK, tm_ / NWORK, tp / NWORK, t_wait, t_emit, t_write);
  }
  return 0;
}
