// smg_report.cpp -- SURVEY 8f N4: read ingest (FASTQ / FASTA text -> the batch layout of smaltgpu_map_batch) and report
// emit (post-processed alignments -> the CIGAR / SAM lines `smalt map` prints) on the host side of libsmaltgpu.
//
// Our restatement of what the reference does around the hot path for single reads:
//   ingest   readHeader / readSeqFast / seqFastqRead   (sequence.c:1056-1140, 1229-1290, 1960-1990): a header is the text
//            behind the prompt up to the end of the line (the name is its first word), a sequence or quality string is
//            every non-blank character up to the next prompt at the start of a line (quality: only once as many characters
//            as the sequence has were read); letters are kept upper case with U -> T, anything else reads as N
//            (make3BitMangledCodec, sequence.c:287-318)
//   select   resultSetFilterResults (results.c:2592-2626) and resultSetAddToReport (results.c:2282-2345): output filters,
//            the choice among several best alignments (-r: drand48, randef.h:19-20), reportAddMap's duplicate test
//            (report.c:545-578)
//   print    fprintREPALIcigar / fprintREPALIsam (report.c:711-906), the CIGAR strings of writeDiffStrCIGAR
//            (diffstr.c:298-363), the SAM header (report.c:1266-1300)
// Host code only: nothing here touches the device, so the parity tests of this file run without a GPU
// (tests/test_report.py against output of the reference program committed under tests/golden/).
#include <ctype.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/smaltgpu.h"

extern "C" int smaltgpu_set_error(int code, const char *msg);   // smaltgpu.cpp: the per-thread message of smaltgpu_last_error()

namespace {

inline bool is_prompt(int c) { return c == '>' || c == '@' || c == '+'; }

// one symbol as the reads' codec stores and prints it (codtab -> decodtab, sequence.c:298-313)
inline uint8_t canon_base(uint8_t c) {
  int cu = toupper(c);
  if (cu == 'U') cu = 'T';
  const int offs = cu - 'A' + 1;
  return (offs > 0 && offs < 32) ? (uint8_t)cu : (uint8_t)'N';
}

struct ReadBlock {                       // what one parser thread produced
  std::vector<uint8_t> bases, quals;
  std::vector<uint64_t> off, name_off, rec_end;      // rec_end: offset in the text behind each record
  std::vector<char> names;
  bool has_qual = true;
  void clear() { bases.clear(); quals.clear(); off.assign(1, 0); name_off.assign(1, 0); names.clear(); rec_end.clear(); has_qual = true; }
};

// ---- the plain four-line FASTQ record, one memchr per line --------------------------------------------------------
// Returns the number of bytes of the record at p, 0 if the text at p is not such a record (the general parser decides then),
// or -1 when the record is not complete inside [p, end) (a later chunk holds the rest).
long strict_record(const char *p, const char *end, bool is_last, ReadBlock &o) {
  if (p >= end || *p != '@') return 0;
  const char *l1e = (const char *)memchr(p, '\n', (size_t)(end - p));
  if (!l1e) return -1;
  const char *l2 = l1e + 1, *l2e = l2 < end ? (const char *)memchr(l2, '\n', (size_t)(end - l2)) : nullptr;
  if (!l2e) return -1;
  const char *l3 = l2e + 1, *l3e = l3 < end ? (const char *)memchr(l3, '\n', (size_t)(end - l3)) : nullptr;
  if (!l3e) return -1;
  if (*l3 != '+') return 0;
  const char *l4 = l3e + 1, *l4e = l4 < end ? (const char *)memchr(l4, '\n', (size_t)(end - l4)) : nullptr;
  const char *next;
  if (!l4e) { if (!is_last) return -1; l4e = end; next = end; }
  else next = l4e + 1;
  if (next < end) { if (*next != '@') return 0; }       // anything else behind the quality line: the general rules apply
  else if (!is_last) return -1;                           // the next line decides whether the quality string goes on
  size_t sl = (size_t)(l2e - l2), ql = (size_t)(l4e - l4);
  if (sl && l2[sl - 1] == '\r') sl--;
  if (ql && l4[ql - 1] == '\r') ql--;
  if (sl != ql || sl == 0) return 0;
  // (a quality line may open with a prompt character: the general rules only end a quality string once it has the sequence's length)
  const char *nm = p + 1, *nme = l1e;
  while (nm < nme && isspace((unsigned char)*nm)) nm++;
  const char *t = nm;
  while (t < nme && !isspace((unsigned char)*t)) t++;
  const size_t b0 = o.bases.size();
  o.bases.resize(b0 + sl); o.quals.resize(b0 + sl);
  for (size_t i = 0; i < sl; i++) {
    const unsigned char c = (unsigned char)l2[i], q = (unsigned char)l4[i];
    if (isspace(c) || isspace(q)) { o.bases.resize(b0); o.quals.resize(b0); return 0; }
    o.bases[b0 + i] = canon_base(c); o.quals[b0 + i] = q;
  }
  o.names.insert(o.names.end(), nm, t); o.names.push_back('\0');
  o.off.push_back(o.bases.size()); o.name_off.push_back(o.names.size());
  return (long)(next - p);
}

// ---- the general rules (sequential) ---------------------------------------------------------------------------------
struct Stream { const char *p, *end; };

// header line: -> 0 ok, 1 end of text before any prompt, -2 not a prompt.  name = first word behind the prompt.
int read_header(Stream &s, int *prompt, std::string &name) {
  bool was_space = true, in_name = true, eol = false;
  *prompt = 0; name.clear();
  while (s.p < s.end && !eol) {
    const unsigned char c = (unsigned char)*s.p++;
    if (was_space) {
      if (isspace(c)) { eol = (c == '\n' && *prompt); continue; }
      if (!*prompt) { if (!is_prompt(c)) return -2; *prompt = c; continue; }
      was_space = false;
    } else if (isspace(c)) {
      if (c == '\n' && *prompt) { eol = true; continue; }
      was_space = true;
      in_name = false;
      continue;
    }
    if (in_name) name.push_back((char)c);                // later words of the header are not used
  }
  if (!*prompt) return 1;
  return 0;
}

// sequence or quality characters up to the next prompt at the start of a line (left in place); minlen as in readSeqFast
void read_symbols(Stream &s, std::vector<uint8_t> &out, size_t minlen, int *prompt, bool bases) {
  bool was_newline = false;
  size_t n = 0;
  *prompt = 0;
  while (s.p < s.end) {
    const unsigned char c = (unsigned char)*s.p;
    if (isspace(c)) { was_newline = (c == '\n'); s.p++; continue; }
    if (was_newline) {
      if (n >= minlen && is_prompt(c)) { *prompt = c; return; }
      was_newline = false;
    }
    out.push_back(bases ? canon_base(c) : c);
    n++; s.p++;
  }
}

// -> 0, or -1 with msg.  Stops (without consuming) at a record that may go on behind `end` unless is_last.
int general_parse(const char *text, const char *end, bool is_last, uint32_t max_reads, ReadBlock &o, std::string &msg, bool *reached_end) {
  Stream s{text, end};
  std::string name;
  while (s.p < s.end && (!max_reads || o.off.size() - 1 < max_reads)) {
    const char *rec0 = s.p;
    const size_t b0 = o.bases.size(), n0 = o.names.size();
    int this_prompt = '+', next_prompt = 0;
    bool eof = false;
    while (this_prompt == '+') {                               // seqFastqRead skips stray quality blocks (sequence.c:1968-1977)
      const int rv = read_header(s, &this_prompt, name);
      if (rv == 1) { eof = true; break; }
      if (rv < 0) { msg = "not in FASTA/FASTQ format (a header line was expected)"; return -1; }
      o.bases.resize(b0);
      read_symbols(s, o.bases, 0, &next_prompt, true);
    }
    if (eof) { o.bases.resize(b0); break; }
    bool fastq = false;
    if (next_prompt == '+') {
      int qp = 0;
      std::string qname;
      std::vector<uint8_t> q;
      const int rv = read_header(s, &qp, qname);
      if (rv || qp != '+') { msg = "not in FASTA/FASTQ format (quality header)"; return -1; }
      int p2 = 0;
      read_symbols(s, q, o.bases.size() - b0, &p2, false);
      if (!p2 && !is_last) { o.bases.resize(b0); o.names.resize(n0); s.p = rec0; break; }     // may go on in the next chunk
      if (q.size() != o.bases.size() - b0) { msg = "sequence and quality string differ in length (read '" + name + "')"; return -1; }
      if (o.has_qual) { o.quals.resize(b0); o.quals.insert(o.quals.end(), q.begin(), q.end()); }
      fastq = true;
    } else if (!next_prompt && !is_last) { o.bases.resize(b0); o.names.resize(n0); s.p = rec0; break; }
    if (!fastq) o.has_qual = false;
    o.names.insert(o.names.end(), name.begin(), name.end()); o.names.push_back('\0');
    o.off.push_back(o.bases.size()); o.name_off.push_back(o.names.size());
    o.rec_end.push_back((uint64_t)(s.p - text));
  }
  if (!o.has_qual) o.quals.clear();
  *reached_end = s.p >= s.end;
  return 0;
}

inline bool only_blank(const char *p, const char *end) { for (; p < end; p++) if (!isspace((unsigned char)*p)) return false; return true; }

}  // namespace

struct smaltgpu_reads {
  ReadBlock all;
  std::vector<ReadBlock> part;
};

extern "C" smaltgpu_reads *smaltgpu_reads_create(void) { smaltgpu_reads *r = new smaltgpu_reads(); r->all.clear(); return r; }
extern "C" void smaltgpu_reads_free(smaltgpu_reads *r) { delete r; }

extern "C" int smaltgpu_reads_parse(smaltgpu_reads *rs, const char *text, uint64_t len, int is_last, uint32_t max_reads, int nthreads,
                                    smaltgpu_reads_view *view) {
  if (!rs || !view || (!text && len)) return smaltgpu_set_error(SMALTGPU_EARG, "null argument");
  memset(view, 0, sizeof(*view));
  const char *end = text + len;
  ReadBlock &A = rs->all;
  A.clear();
  if (nthreads < 1) nthreads = 1;
  if ((uint64_t)nthreads > len / (1u << 20) + 1) nthreads = (int)(len / (1u << 20) + 1);
  // plain four-line FASTQ: byte ranges in parallel, each from the first record that starts in it
  std::vector<const char *> start((size_t)nthreads + 1, end);
  start[0] = text;
  bool strict = len > 0 && text[0] == '@';
  const char *stop_last = nullptr;
  for (int t = 1; t < nthreads && strict; t++) {
    const char *p = text + len * (uint64_t)t / (uint64_t)nthreads;
    if (p < start[t - 1]) p = start[t - 1];
    const char *found = nullptr;
    // a record starts at a line that opens with '@' whose next-but-one line opens with '+' and whose sequence and quality
    // lines have the same length (a quality line may open with '@' too, but the line two below it is then a sequence)
    for (int tries = 0; tries < 64 && p < end; tries++) {
      const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
      if (!nl) break;
      const char *l = nl + 1;
      if (l < end && *l == '@') {
        ReadBlock probe; probe.clear();
        const long n = strict_record(l, end, is_last != 0, probe);
        if (n > 0 || n < 0) { found = l; break; }
      }
      p = l;
    }
    if (!found) { strict = false; break; }
    start[t] = found;
  }
  if (strict) {
    rs->part.resize((size_t)nthreads);
    std::vector<int> bad((size_t)nthreads, 0);
    std::vector<const char *> stop((size_t)nthreads, nullptr);
    auto work = [&](int t) {
      ReadBlock &o = rs->part[(size_t)t];
      o.clear();
      const char *p = start[t], *lim = start[t + 1];
      while (p < lim) {
        const long n = strict_record(p, end, is_last != 0, o);
        if (n == 0 && lim == end && only_blank(p, end)) { p = end; break; }      // blank lines behind the last record
        if (n == 0) { bad[(size_t)t] = 1; break; }
        if (n < 0) break;                                  // incomplete: only at the end of the text
        p += n;
        o.rec_end.push_back((uint64_t)(p - text));
      }
      stop[(size_t)t] = p;
    };
    if (nthreads == 1) work(0);
    else { std::vector<std::thread> th; for (int t = 0; t < nthreads; t++) th.emplace_back(work, t); for (auto &x : th) x.join(); }
    for (int t = 0; t < nthreads && strict; t++) {
      if (bad[(size_t)t]) strict = false;
      else if (t + 1 < nthreads && stop[(size_t)t] != start[t + 1]) strict = false;      // a range must end where the next one starts
    }
    if (strict) {
      stop_last = stop[(size_t)nthreads - 1];
      for (int t = 0; t < nthreads; t++) {
        const ReadBlock &o = rs->part[(size_t)t];
        const uint64_t b0 = A.bases.size(), n0 = A.names.size();
        A.bases.insert(A.bases.end(), o.bases.begin(), o.bases.end());
        A.quals.insert(A.quals.end(), o.quals.begin(), o.quals.end());
        A.names.insert(A.names.end(), o.names.begin(), o.names.end());
        for (size_t i = 1; i < o.off.size(); i++) { A.off.push_back(b0 + o.off[i]); A.name_off.push_back(n0 + o.name_off[i]); }
        A.rec_end.insert(A.rec_end.end(), o.rec_end.begin(), o.rec_end.end());
      }
    }
  }
  bool reached_end = strict && stop_last == end;
  if (!strict) {
    A.clear();
    std::string msg;
    if (general_parse(text, end, is_last != 0, max_reads, A, msg, &reached_end)) return smaltgpu_set_error(SMALTGPU_EFILE, msg.c_str());
  }
  size_t n = A.off.size() - 1;
  if (max_reads && n > max_reads) {
    reached_end = false;
    n = max_reads;
    A.off.resize(n + 1); A.name_off.resize(n + 1); A.rec_end.resize(n);
    A.bases.resize(A.off[n]); if (A.has_qual) A.quals.resize(A.off[n]); A.names.resize(A.name_off[n]);
  }
  if (A.bases.empty()) A.bases.push_back(0);
  if (A.names.empty()) A.names.push_back(0);
  view->nreads = (uint32_t)n;
  view->has_qual = (A.has_qual && n) ? 1u : 0u;
  view->bases = A.bases.data();
  view->quals = view->has_qual ? A.quals.data() : nullptr;
  view->read_off = A.off.data();
  view->names = A.names.data();
  view->name_off = A.name_off.data();
  view->consumed = (is_last && reached_end) ? len : (n ? A.rec_end[n - 1] : 0);
  return SMALTGPU_OK;
}

// =====================================================================================================================
// report
// =====================================================================================================================
namespace {

enum : uint32_t { RF_SELECT = 0x01, RF_REVERSE = 0x04, RF_NOOUTPUT = 0x10, RF_BELOWRELSW = 0x20, RF_REPORTED = 0x200 };   // results.h:67-82
enum : uint32_t { MF_MAPPED = 0x01, MF_REVERSE = 0x02, MF_PAIRED = 0x04, MF_2NDMATE = 0x08, MF_PRIMARY = 0x10, MF_PARTIAL = 0x20, MF_MULTI = 0x40 };   // report.h:66-74
enum : uint8_t { RP_MAPPED = 0x01, RP_CONTIG = 0x02, RP_PROPER = 0x04, RP_WITHIN = 0x08 };                                                        // report.h:76-81
enum { MAPSCOR_MAX_RANDOM = 3, SAMPLESIZ_MAPQ_RANDOM = 9, QUALSCOR_SCAL = 10, CIGAR_MAXTAG = 99 };   // results.c:57,73,80; report.c:73

struct Ali {                              // REPALI (report.c:130-145)
  uint32_t status;
  int swatscor, mapscor;
  uint32_t qs, qe;
  uint64_t ss, se;
  int32_t sidx;
  const uint8_t *dstr;
};

inline void dget(uint8_t b, unsigned *count, unsigned *typ) { *count = b & 0x3f; *typ = b >> 6; }   // diffstr.h:28-76: 0 M, 1 D, 2 I, 3 S

int diff_matches(const uint8_t *d) {                       // diffStrCalcAliLen's match count (diffstr.c:932-952)
  int m = 0;
  for (; *d; d++) { unsigned c, t; dget(*d, &c, &t); m += (int)c + (t == 0 ? 1 : 0); }
  return m;
}

int diff_columns(const uint8_t *d) {                       // diffStrCalcAliLen's return value: alignment columns, the closing code not counted
  int n = 0;
  unsigned c = 0, t = 0;
  for (; *d; d++) { dget(*d, &c, &t); n += (int)c + 1; }
  return t == 3 ? n - 1 : n;
}

int diff_edit_distance(const uint8_t *d) {                 // diffStrGetLevenshteinDistance (diffstr.c:1496-1510)
  int ed = 0;
  unsigned t = 0, c;
  for (; *d; d++) { dget(*d, &c, &t); if (t != 0) ed++; }
  if (ed > 0 && t == 3) ed--;
  return ed;
}

// writeDiffStrCIGAR (diffstr.c:298-363).  ext: "<n><op>" units with clipping, else "<op> <n> " units; silent: mismatches
// count as matches ('M') instead of 'X'.  A run of matches absorbs the matches in front of the next operation; equal gap
// operations that follow each other directly are counted together; the string ends with the terminating S.
bool put_cigar(std::string &o, const uint8_t *d, bool ext, bool silent, int clip_start, int clip_end, char clipchar) {
  static const char SYM[] = "MDIX";
  char buf[32];
  auto unit = [&](char ch, unsigned n) {
    if (n > 0) { if (ext) snprintf(buf, sizeof(buf), "%d%c", (int)n, ch); else snprintf(buf, sizeof(buf), "%c %d ", ch, (int)n); o += buf; }
    else o.push_back(ch);
  };
  if (!d) { unit('*', 0); return true; }
  if (!*d) return false;
  if (clip_start > 0 && ext) unit(clipchar, (unsigned)clip_start);
  unsigned run = 0, run_typ = 0, typ = 0, count;
  for (; *d; d++) {
    dget(*d, &count, &typ);
    const bool like_match = typ == 0 || (typ == 3 && silent);
    if (run_typ == 0) {
      run += count;
      if (like_match) { run++; continue; }
    } else if (typ == run_typ && count < 1) { run++; continue; }
    if (run > 0) unit(SYM[run_typ], run);
    if (like_match) { run = count + 1; run_typ = 0; }
    else {
      if (count > 0 && run_typ != 0) unit('M', count);
      run = 1; run_typ = typ;
    }
  }
  if (typ != 3) return false;
  if (run > 1) unit(silent ? 'M' : 'X', run - 1);
  if (clip_end > 0 && ext) unit(clipchar, (unsigned)clip_end);
  return true;
}

inline void first_word(std::string &o, const char *s, bool strip_mate) {        // copyReadNamStrToREPSTR (report.c:434-461)
  const size_t b = o.size();
  while (*s && !isspace((unsigned char)*s)) o.push_back(*s++);
  const size_t n = o.size() - b;
  if (strip_mate && n > 2 && o[o.size() - 2] == '/' && (o.back() == '1' || o.back() == '2')) o.resize(o.size() - 2);
}

int mapq_of_random_draw(int n) {                            // assignPhredScaledMappingScoreToRandomDraw (results.c:214-230)
  if (n < 1 || n > SAMPLESIZ_MAPQ_RANDOM) return 0;
  if (n == 1) return MAPSCOR_MAX_RANDOM + 1;
  int q = (int)(-QUALSCOR_SCAL * log10(((double)(n - 1)) / n) + .499);
  if (q > MAPSCOR_MAX_RANDOM) q = MAPSCOR_MAX_RANDOM;
  else if (q < 0) q = 0;
  return q;
}

struct ReadCtx {
  const smaltgpu_post_out *po;
  const smaltgpu_reads_view *rv;
  const smaltgpu_report_opts *op;
  const char *const *seqnames;
  int64_t nseq;
  const uint32_t *seqlen;                // lengths of the reference sequences (SSAHA lines print them): from smaltgpu_report_header
};

// the alignments of read i in the order the reference's report holds them; draw: the pre-drawn index for a random choice
// among equal best alignments (-1: no draw was due).  -> false on an inconsistency
bool select_read(const ReadCtx &cx, uint32_t i, int draw, std::vector<Ali> &out, std::vector<uint32_t> &st) {
  const smaltgpu_post_out &po = *cx.po;
  const smaltgpu_report_opts &op = *cx.op;
  const bool split = (op.outflags & SMALTGPU_OUT_SPLIT) != 0;
  const smaltgpu_post_result *res = po.res + po.res_off[i];
  const int32_t *sortr = po.sortr + po.sort_off[i];
  const int n = (int)(po.sort_off[i + 1] - po.sort_off[i]);
  const uint32_t nres = (uint32_t)(po.res_off[i + 1] - po.res_off[i]);
  const uint32_t rlen = (uint32_t)(cx.rv->read_off[i + 1] - cx.rv->read_off[i]);
  out.clear();
  st.assign(nres, 0);
  for (uint32_t j = 0; j < nres; j++) st[j] = res[j].status;
  // resultSetFilterResults (results.c:2592-2626)
  if (n > 0) {
    const double idt = op.min_identity <= 1.0 ? op.min_identity * rlen : op.min_identity;
    const int minid = (int)(uint32_t)idt;
    const int maxsw = res[sortr[0]].swatscor, minabs = op.min_swscor;
    int minrel = 0;
    if (op.min_swscor_below_max >= 0 && minabs + op.min_swscor_below_max < maxsw) minrel = maxsw - op.min_swscor_below_max;
    for (int k = 0; k < n; k++) {
      const smaltgpu_post_result &r = res[sortr[k]];
      if (r.swatscor < minabs || diff_matches(po.diffstr + r.stroffs) < minid) st[(size_t)sortr[k]] |= RF_NOOUTPUT;
      else if (r.swatscor < minrel) st[(size_t)sortr[k]] |= RF_BELOWRELSW;
    }
  }
  auto add = [&](int ridx, int mapscor_override, uint32_t mateflg) {     // resultSetAddResultToReport + reportAddMap for a single read
    Ali a;
    memset(&a, 0, sizeof(a));
    if (ridx < 0 || (st[(size_t)ridx] & RF_NOOUTPUT) || res[ridx].strlen < 1) a.status = mateflg & ~(uint32_t)MF_MAPPED;
    else {
      const smaltgpu_post_result &r = res[ridx];
      a.status = mateflg | MF_MAPPED | ((st[(size_t)ridx] & RF_REVERSE) ? (uint32_t)MF_REVERSE : 0u);
      a.swatscor = r.swatscor; a.mapscor = mapscor_override >= 0 ? mapscor_override : r.mapscor;
      a.qs = r.q_start; a.qe = r.q_end; a.ss = r.s_start; a.se = r.s_end; a.sidx = r.sidx; a.dstr = po.diffstr + r.stroffs;
    }
    for (size_t k = out.size(); k-- > 0;) {                              // findREPALI (report.c:545-578): known already -> ignored
      const Ali &b = out[k];
      if (a.ss == b.ss && a.se == b.se && a.sidx == b.sidx && a.qs == b.qs && a.qe == b.qe &&
          (a.status & (MF_REVERSE | MF_2NDMATE)) == (b.status & (MF_REVERSE | MF_2NDMATE))) return;
    }
    out.push_back(a);
  };
  // resultSetAddToReport (results.c:2282-2345)
  int top = n < 1 ? -1 : sortr[0], top_mapscor = -1;
  uint32_t mateflg = 0;
  if (top >= 0) {
    // getNumberOfTopSwatRESULTs (results.c:839-869)
    const bool is_single = n < 2 || res[sortr[1]].swatscor != res[sortr[0]].swatscor;
    int ns = n;
    if (n > 2) { const int thr = res[sortr[1]].swatscor; int k = 2; while (k < n && res[sortr[k]].swatscor == thr) k++; ns = k; }
    if (res[top].mapscor == 0 && !is_single && ns > 1 && (op.outflags & SMALTGPU_OUT_BEST) && !split) {
      mateflg |= MF_MULTI;
      if (op.outflags & SMALTGPU_OUT_RANDSEL) {
        if (draw < 0 || draw >= n) return false;
        top = sortr[draw];
        top_mapscor = mapq_of_random_draw(ns);
      } else if (op.outflags & SMALTGPU_OUT_SINGLE) top = -1;
    }
  }
  add(top, top_mapscor, mateflg | MF_PRIMARY);
  if (top >= 0) st[(size_t)top] |= RF_REPORTED;
  if ((op.outflags & SMALTGPU_OUT_SINGLE) && !split) return true;
  for (int k = 1; k < n; k++) {
    const int r = sortr[k];
    if ((op.outflags & SMALTGPU_OUT_BEST) && res[r].swatscor < res[sortr[k - 1]].swatscor) break;
    if (!(st[(size_t)r] & (RF_NOOUTPUT | RF_BELOWRELSW))) { add(r, (r == top) ? top_mapscor : -1, mateflg); st[(size_t)r] |= RF_REPORTED; }
  }
  // split reads: the best alignments of every read segment that are not out yet follow as partial alignments
  // (resultSetAdd2ndaryResultsToReport, results.c:2250-2278)
  if ((op.outflags & SMALTGPU_OUT_BEST) && split) {
    const int nseg = po.qsegno[i];
    const int32_t *by_segment = po.segsrtr + po.sort_off[i], *segment_begin = po.segnor + po.seg_off[i];
    if (nseg > 0 && (int64_t)(po.seg_off[i + 1] - po.seg_off[i]) != (int64_t)nseg + 1) return false;
    for (int g = 0; g < nseg; g++) {
      int last = 0;
      for (int k = segment_begin[g]; k < segment_begin[g + 1]; k++) {
        const int r = by_segment[k];
        if (r < 0 || (uint32_t)r >= nres) return false;
        if (st[(size_t)r] & RF_NOOUTPUT) continue;
        if ((st[(size_t)r] & RF_REPORTED) || res[r].swatscor < last) break;            // (best-only is on in this branch)
        add(r, -1, mateflg | MF_PARTIAL);
        st[(size_t)r] |= RF_REPORTED;
        last = res[r].swatscor;
      }
    }
  }
  return true;
}

// does read i need a random draw, and among how many (resultSetAddToReport, results.c:2293-2301)
int draw_range(const ReadCtx &cx, uint32_t i) {
  const smaltgpu_post_out &po = *cx.po;
  const smaltgpu_report_opts &op = *cx.op;
  if (!(op.outflags & SMALTGPU_OUT_RANDSEL) || !(op.outflags & SMALTGPU_OUT_BEST) || (op.outflags & SMALTGPU_OUT_SPLIT)) return 0;
  const smaltgpu_post_result *res = po.res + po.res_off[i];
  const int32_t *sortr = po.sortr + po.sort_off[i];
  const int n = (int)(po.sort_off[i + 1] - po.sort_off[i]);
  if (n < 2 || res[sortr[1]].swatscor != res[sortr[0]].swatscor || res[sortr[0]].mapscor != 0) return 0;
  int ns = n;
  if (n > 2) { const int thr = res[sortr[1]].swatscor; int k = 2; while (k < n && res[sortr[k]].swatscor == thr) k++; ns = k; }
  return ns > 1 ? ns : 0;
}

// the other mate's printed alignment, the template length and what is known about the pairing (REPPAIR_*) -- null for single reads
struct PairSide { const Ali *mate; int isize; uint8_t pairflg; };

// the letter behind "cigar:" / "alignment:" (getMapLabelFromFlag, report.c:217-246; unmapped: report.c:624, :746)
char map_label(const Ali &a, const PairSide *ps) {
  if (!(a.status & MF_MAPPED)) return (a.status & MF_MULTI) ? 'R' : 'N';
  uint8_t pf = ps ? ps->pairflg : 0;
  if (ps && ps->mate && a.sidx == ps->mate->sidx) pf |= RP_CONTIG;         // report.c:1173-1176 (an unprinted mate carries sequence 0)
  if (a.status & MF_PARTIAL) return 'P';
  if (pf & RP_MAPPED) return (pf & RP_CONTIG) ? ((pf & RP_PROPER) ? ((pf & RP_WITHIN) ? 'A' : 'B') : 'C') : 'D';
  return 'S';
}

bool print_cigar_line(std::string &o, const ReadCtx &cx, uint32_t i, const Ali &a, const PairSide *ps = nullptr) {      // fprintREPALIcigar (report.c:711-760)
  char buf[96];
  const bool mapped = (a.status & MF_MAPPED) != 0;
  const char flagchr = map_label(a, ps);
  const int mq = mapped ? (a.mapscor > CIGAR_MAXTAG ? CIGAR_MAXTAG : a.mapscor) : 0;
  snprintf(buf, sizeof(buf), "cigar:%c:%2.2d ", flagchr, mq);
  o += buf;
  first_word(o, cx.rv->names + cx.rv->name_off[i], false);
  uint32_t qs = 0, qe = 0;
  char sense = '*';
  if (mapped) { if (a.status & MF_REVERSE) { qs = a.qe; qe = a.qs; sense = '-'; } else { qs = a.qs; qe = a.qe; sense = '+'; } }
  snprintf(buf, sizeof(buf), " %u %u %c ", qs, qe, sense);
  o += buf;
  if (mapped) { if (a.sidx < 0 || a.sidx >= cx.nseq) return false; first_word(o, cx.seqnames[a.sidx], false); }
  else o.push_back('*');
  snprintf(buf, sizeof(buf), " %u %u + %d ", mapped ? (unsigned)a.ss : 0u, mapped ? (unsigned)a.se : 0u, mapped ? a.swatscor : 0);
  o += buf;
  if (!put_cigar(o, mapped ? a.dstr : nullptr, false, true, 0, 0, 'H')) return false;
  o.push_back('\n');
  return true;
}

// -f ssaha (fprintREPALIssaha, report.c:579-646; line layout report.c:206): label and mapping quality as in the CIGAR line, then
// score, read, sequence, the read range in the read's own direction, the reference range, F/C, matching columns, their share of
// the alignment's columns in per cent, read length, sequence length
bool print_ssaha_line(std::string &o, const ReadCtx &cx, uint32_t i, const Ali &a, const PairSide *ps = nullptr) {
  char buf[160];
  const bool mapped = (a.status & MF_MAPPED) != 0;
  const int mq = mapped ? (a.mapscor > CIGAR_MAXTAG ? CIGAR_MAXTAG : a.mapscor) : 0;
  snprintf(buf, sizeof(buf), "alignment:%c:%2.2d %-5d ", map_label(a, ps), mq, mapped ? a.swatscor : 0);
  o += buf;
  first_word(o, cx.rv->names + cx.rv->name_off[i], false);
  o.push_back(' ');
  uint32_t qs = 0, qe = 0, slen = 0;
  int same = 0;
  double share = .0;
  char sense = '*';
  if (mapped) {
    if (a.sidx < 0 || a.sidx >= cx.nseq || !a.dstr) return false;
    first_word(o, cx.seqnames[a.sidx], false);
    if (a.status & MF_REVERSE) { qs = a.qe; qe = a.qs; sense = 'C'; } else { qs = a.qs; qe = a.qe; sense = 'F'; }
    slen = cx.seqlen[a.sidx];
    same = diff_matches(a.dstr);
    const int cols = diff_columns(a.dstr);
    share = cols > 0 ? ((double)100 * same) / cols : .0;
  } else o.push_back('*');
  const uint32_t qlen = cx.rv->read_off[i + 1] - cx.rv->read_off[i];
  snprintf(buf, sizeof(buf), " %8u %8u %9u %9u   %c %7d %5.2f %u %u\n", qs, qe, mapped ? (unsigned)a.ss : 0u, mapped ? (unsigned)a.se : 0u, sense, same, share,
           qlen, slen);
  o += buf;
  return true;
}

bool print_sam_line(std::string &o, const ReadCtx &cx, uint32_t i, const Ali &a, const PairSide *ps = nullptr) {        // fprintREPALIsam (report.c:762-906)
  char buf[96];
  const smaltgpu_report_opts &op = *cx.op;
  const bool mapped = (a.status & MF_MAPPED) != 0, soft = (op.modflags & SMALTGPU_REP_SOFTCLIP) != 0;
  const uint64_t r0 = cx.rv->read_off[i];
  const uint32_t qlen = (uint32_t)(cx.rv->read_off[i + 1] - r0);
  const uint8_t *seq = cx.rv->bases + r0, *qual = cx.rv->has_qual ? cx.rv->quals + r0 : nullptr;
  unsigned flag = 0;
  first_word(o, cx.rv->names + cx.rv->name_off[i], true);
  int clip_start = 0, clip_end = 0;
  // mate fields (report.c:795-815)
  int isize = ps ? ps->isize : 0;
  uint32_t mpos = 0;
  const char *mate_seq = nullptr;
  if (a.status & MF_PAIRED) {
    flag |= 0x1;
    if (a.status & MF_2NDMATE) { flag |= 0x80; isize = -isize; } else flag |= 0x40;
    if (ps && ps->mate && (ps->mate->status & MF_MAPPED)) {
      mpos = (uint32_t)ps->mate->ss;
      if (ps->mate->status & MF_REVERSE) flag |= 0x20;
      if (ps->mate->sidx < 0 || ps->mate->sidx >= cx.nseq) return false;
      mate_seq = cx.seqnames[ps->mate->sidx];
    } else { flag |= 0x8; isize = 0; }
  }
  uint32_t seg0 = 0, segn = 0;
  bool rev = false;
  if (mapped) {
    rev = (a.status & MF_REVERSE) != 0;
    if (soft) { seg0 = 0; segn = qlen; } else { seg0 = a.qs - 1; segn = a.qe - a.qs + 1; }
    if (a.qe > qlen || seg0 > qlen) return false;
    if (!segn || seg0 + segn > qlen) segn = qlen - seg0;                 // appendSeqSegment (sequence.c:865-867)
    if (rev) { flag |= 0x10; clip_start = (int)(qlen - a.qe); clip_end = (int)a.qs - 1; }
    else { clip_start = (int)a.qs - 1; clip_end = (int)(qlen - a.qe); }
    if (ps && (ps->pairflg & RP_PROPER) && (ps->pairflg & RP_WITHIN)) flag |= 0x2;
    if (a.status & MF_PARTIAL) flag |= 0x100;
  } else { flag |= 0x4; isize = 0; }
  snprintf(buf, sizeof(buf), "\t%hu\t", (unsigned short)flag);
  o += buf;
  if (mapped) { if (a.sidx < 0 || a.sidx >= cx.nseq) return false; first_word(o, cx.seqnames[a.sidx], false); }
  else o.push_back('*');
  snprintf(buf, sizeof(buf), "\t%i\t%hi\t", mapped ? (int)(uint32_t)a.ss : 0, (short)(mapped ? a.mapscor : 0));
  o += buf;
  int nm = 0;
  if (mapped) {
    if (!put_cigar(o, a.dstr, true, !(op.modflags & SMALTGPU_REP_XMISMATCH), clip_start, clip_end, soft ? 'S' : 'H')) return false;
    nm = diff_edit_distance(a.dstr);
  } else o.push_back('*');
  o.push_back('\t');
  if (mate_seq) first_word(o, mate_seq, false); else o.push_back('*');
  snprintf(buf, sizeof(buf), "\t%i\t%i\t", (int)mpos, isize);
  o += buf;
  if (mapped || soft) {
    if (!mapped) { seg0 = 0; segn = qlen; }
    const size_t b = o.size();
    o.resize(b + segn);
    if (rev) {
      // reverse complement: the four standard letters are complemented, any other letter stays (appendSeqSegment, sequence.c:881-893)
      for (uint32_t k = 0; k < segn; k++) {
        const uint8_t c = seq[seg0 + segn - 1 - k];
        o[b + k] = (char)(c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c);
      }
    } else memcpy(&o[b], seq + seg0, segn);
    o.push_back('\t');
    if (qual && segn) {
      const size_t q = o.size();
      o.resize(q + segn);
      if (rev) for (uint32_t k = 0; k < segn; k++) o[q + k] = (char)qual[seg0 + segn - 1 - k];
      else memcpy(&o[q], qual + seg0, segn);
    } else o.push_back('*');
  } else o += "*\t*";
  snprintf(buf, sizeof(buf), "\tNM:i:%i\tAS:i:%i\n", nm, mapped ? a.swatscor : 0);
  o += buf;
  return true;
}

bool print_line(std::string &o, const ReadCtx &cx, uint32_t i, const Ali &a, const PairSide *ps) {       // writeREPALI's switch (report.c:1178-1200)
  switch (cx.op->format) {
    case SMALTGPU_FMT_SAM: return print_sam_line(o, cx, i, a, ps);
    case SMALTGPU_FMT_SSAHA: return print_ssaha_line(o, cx, i, a, ps);
    default: return print_cigar_line(o, cx, i, a, ps);
  }
}

}  // namespace

struct smaltgpu_report { std::string text; std::vector<std::string> part; std::vector<int> draw; std::vector<uint32_t> seqlen; };

namespace {
int check_format(const smaltgpu_report *rp, const smaltgpu_report_opts *op, int64_t nseq) {
  if (op->format != SMALTGPU_FMT_CIGAR && op->format != SMALTGPU_FMT_SAM && op->format != SMALTGPU_FMT_SSAHA) return smaltgpu_set_error(SMALTGPU_EARG, "unknown output format");
  if (op->format == SMALTGPU_FMT_SSAHA && (int64_t)rp->seqlen.size() != nseq)
    return smaltgpu_set_error(SMALTGPU_EARG, "SSAHA lines carry the sequence lengths: call smaltgpu_report_header on this report first");
  return SMALTGPU_OK;
}
}  // namespace

extern "C" smaltgpu_report *smaltgpu_report_create(void) { return new smaltgpu_report(); }
extern "C" void smaltgpu_report_free(smaltgpu_report *r) { delete r; }

extern "C" int smaltgpu_report_header(smaltgpu_report *rp, const char *const *seqnames, const uint64_t *sop, int64_t nseq, const smaltgpu_report_opts *op,
                                      const char *prognam, const char *version, int argc, const char *const *argv, const char **text, uint64_t *len) {
  if (!rp || !seqnames || !sop || !op || !text || !len || nseq < 1) return smaltgpu_set_error(SMALTGPU_EARG, "null argument");
  std::string &o = rp->text;
  o.clear();
  rp->seqlen.resize((size_t)nseq);
  for (int64_t s = 0; s < nseq; s++) rp->seqlen[(size_t)s] = (uint32_t)(sop[s + 1] - sop[s]);
  if (op->format == SMALTGPU_FMT_SAM && (op->modflags & SMALTGPU_REP_HEADER)) {       // writeSAMHeaderf (report.c:1266-1300)
    char buf[64];
    o += "@HD\tVN:1.3\tSO:unknown\n";
    for (int64_t s = 0; s < nseq; s++) {
      o += "@SQ\tSN:";
      const char *n = seqnames[s];
      for (int k = 0; k < 511 && n[k] && !isspace((unsigned char)n[k]); k++) o.push_back(n[k]);
      snprintf(buf, sizeof(buf), "\tLN:%u\n", (unsigned)(sop[s + 1] - sop[s]));
      o += buf;
    }
    o += "@PG\tID:"; o += prognam ? prognam : "smaltgpu"; o += "\tPN:"; o += prognam ? prognam : "smaltgpu"; o += "\tVN:"; o += version ? version : "0"; o += "\tCL:";
    if (argc > 0 && argv) { for (int k = 0; k < argc; k++) { if (k) o.push_back(' '); o += argv[k]; } o.push_back('\n'); }
  }
  *text = o.data(); *len = o.size();
  return SMALTGPU_OK;
}

extern "C" int smaltgpu_report_emit(smaltgpu_report *rp, const smaltgpu_post_out *post, const smaltgpu_batch_out *raw, const smaltgpu_reads_view *reads,
                                    const char *const *seqnames, int64_t nseq, const smaltgpu_report_opts *op, int nthreads, const char **text, uint64_t *len) {
  if (!rp || !post || !reads || !seqnames || !op || !text || !len) return smaltgpu_set_error(SMALTGPU_EARG, "null argument");
  if (post->nreads != reads->nreads) return smaltgpu_set_error(SMALTGPU_EARG, "results and reads differ in number");
  if (int e = check_format(rp, op, nseq)) return e;
  const uint32_t n = post->nreads;
  ReadCtx cx{post, reads, op, seqnames, nseq, rp->seqlen.data()};
  for (uint32_t i = 0; i < n; i++) {
    if (post->needs_reference[i]) return smaltgpu_set_error(SMALTGPU_EARG, "a read was left to the caller by smaltgpu_postprocess (needs_reference): give it the packed reference");
    if (raw && raw->stat[i].errcode) {               // the reference stops at a read that fails (rmap.c:1417 -> smalt.c: the message names the read)
      std::string m = "read '";
      first_word(m, reads->names + reads->name_off[i], false);
      char t[160];
      snprintf(t, sizeof(t), "' (%u of its block) carries error code %d%s", i, raw->stat[i].errcode,
               raw->stat[i].errcode == SMALTGPU_ESCORE ? ": inconsistency when calculating Smith-Waterman scores (the reference stops at this read with ERRCODE_SWATSCOR)" : "");
      m += t;
      return smaltgpu_set_error(raw->stat[i].errcode == SMALTGPU_ESCORE ? SMALTGPU_ESCORE : SMALTGPU_EINTERNAL, m.c_str());
    }
  }
  // the random choices in read order, as one thread of the reference makes them (drand48 is the C library's shared sequence)
  rp->draw.assign(n ? n : 1, -1);
  for (uint32_t i = 0; i < n; i++) { const int ns = draw_range(cx, i); if (ns) rp->draw[i] = (int)(short)(drand48() * ns); }
  if (nthreads < 1) nthreads = 1;
  if ((uint32_t)nthreads > n / 512 + 1) nthreads = (int)(n / 512 + 1);
  rp->part.resize((size_t)nthreads);
  std::vector<int> bad((size_t)nthreads, -1);
  auto work = [&](int t) {
    std::string &o = rp->part[(size_t)t];
    o.clear();
    std::vector<Ali> alis;
    std::vector<uint32_t> st;
    const uint32_t lo = (uint32_t)((uint64_t)n * (uint64_t)t / (uint64_t)nthreads), hi = (uint32_t)((uint64_t)n * (uint64_t)(t + 1) / (uint64_t)nthreads);
    o.reserve((size_t)(hi - lo) * (op->format == SMALTGPU_FMT_SAM ? 420 : 96));
    for (uint32_t i = lo; i < hi; i++) {
      if (!select_read(cx, i, rp->draw[i], alis, st)) { bad[(size_t)t] = (int)i; return; }
      for (const Ali &a : alis) {
        const bool ok = print_line(o, cx, i, a, nullptr);
        if (!ok) { bad[(size_t)t] = (int)i; return; }
      }
    }
  };
  if (nthreads == 1) work(0);
  else { std::vector<std::thread> th; for (int t = 0; t < nthreads; t++) th.emplace_back(work, t); for (auto &x : th) x.join(); }
  for (int t = 0; t < nthreads; t++) if (bad[(size_t)t] >= 0) { char m[96]; snprintf(m, sizeof(m), "inconsistent alignment of read %d (alignment string or sequence number)", bad[(size_t)t]); return smaltgpu_set_error(SMALTGPU_EINTERNAL, m); }
  if (nthreads == 1) { *text = rp->part[0].data(); *len = rp->part[0].size(); return SMALTGPU_OK; }
  rp->text.clear();
  size_t tot = 0;
  for (const std::string &s : rp->part) tot += s.size();
  rp->text.reserve(tot);
  for (const std::string &s : rp->part) rp->text += s;
  *text = rp->text.data(); *len = rp->text.size();
  return SMALTGPU_OK;
}

// =====================================================================================================================
// report of read pairs: the tail of rmapPair (resultSetFindPairs, output filters: rmap.c:2099-2109),
// resultSetAddPairToReport (resultpairs.c:1222) and reportWrite (report.c:1758) for every pair of a mapped block
// =====================================================================================================================
#include "smg_pairrun.hpp"

namespace {

using smgpairs::Entry;
using smgpost::Table;

// The printed alignments of one pair: the reference keeps a list per mate and a list of (read entry, mate entry) pairs; an
// alignment that is reported twice (two pairings share it) takes one slot, the later report overwriting the earlier one
// (reportAddMap / findREPALI, report.c:545-578, :1587-1720).
struct PairSheet {
  struct Link { int ia, ib, isize; uint8_t pairflg; };
  std::vector<Ali> side[2];
  std::vector<uint8_t> printed[2];
  std::vector<Link> links;
  void clear() { side[0].clear(); side[1].clear(); links.clear(); }
  int put(int w, const Ali &a) {
    std::vector<Ali> &v = side[w];
    for (size_t k = v.size(); k-- > 0;) {
      const Ali &b = v[k];
      if (a.ss == b.ss && a.se == b.se && a.sidx == b.sidx && a.qs == b.qs && a.qe == b.qe && (a.status & (MF_REVERSE | MF_2NDMATE)) == (b.status & (MF_REVERSE | MF_2NDMATE))) { v[k] = a; return (int)k; }
    }
    v.push_back(a);
    return (int)v.size() - 1;
  }
};

Ali ali_of(const Table &t, int row, int quality, uint32_t mateflg) {            // resultSetAddResultToReport (results.c:2216-2250)
  Ali a;
  memset(&a, 0, sizeof(a));
  if (row < 0 || (t.bits[(size_t)row] & smgpost::WITHHELD) || t.str_len[(size_t)row] < 1) { a.status = mateflg & ~(uint32_t)MF_MAPPED; return a; }
  a.status = mateflg | MF_MAPPED | ((t.bits[(size_t)row] & smgpost::REVERSED) ? (uint32_t)MF_REVERSE : 0u);
  a.swatscor = t.score[(size_t)row]; a.mapscor = quality;
  a.qs = t.q_lo[(size_t)row]; a.qe = t.q_hi[(size_t)row]; a.ss = t.r_lo[(size_t)row]; a.se = t.r_hi[(size_t)row]; a.sidx = (int32_t)t.seq[(size_t)row];
  a.dstr = t.str((uint32_t)row);
  return a;
}

// one reported pairing onto the sheet (addPairResultsToReport, resultpairs.c:1024-1086)
void sheet_add(PairSheet &sh, const Entry &e, const Table &A, const Table &B) {
  using namespace smgpairs;
  PairSheet::Link ln{-1, -1, 0, 0};
  const bool a_out = e.a >= 0 && !(A.bits[(size_t)e.a] & smgpost::WITHHELD), b_out = e.b >= 0 && !(B.bits[(size_t)e.b] & smgpost::WITHHELD);
  if ((e.know & PM_PAIRED) && a_out && b_out) {
    ln.pairflg |= RP_MAPPED;
    if (e.know & PM_SAME_SEQUENCE) {
      ln.pairflg |= RP_CONTIG;
      ln.isize = layout_of(A, (uint32_t)e.a, B, (uint32_t)e.b).tlen;
      if (e.know & PM_IN_RANGE) ln.pairflg |= RP_WITHIN;
      if (e.know & PM_ORIENTED) ln.pairflg |= RP_PROPER;
    }
  }
  const uint32_t base = MF_PAIRED | MF_PRIMARY;
  ln.ia = sh.put(0, ali_of(A, e.a, e.quality_a, base | ((e.know & PM_READ_AMBIGUOUS) ? (uint32_t)MF_MULTI : 0u)));
  ln.ib = sh.put(1, ali_of(B, e.b, e.quality_b, base | MF_2NDMATE | ((e.know & PM_MATE_AMBIGUOUS) ? (uint32_t)MF_MULTI : 0u)));
  sh.links.push_back(ln);
}

// split reads: the best alignments of every read segment of a mate join its list as partial alignments unless they are there already
// (resultSetAdd2ndaryResultsToReport, results.c:2250-2278, from resultSetAddPairToReport, resultpairs.c:1293-1311; an alignment the
// report knows is ignored: reportAddMap, report.c:1676-1679)
void sheet_add_partial(PairSheet &sh, int w, const Table &t) {
  if (t.by_score.empty() || !(t.set_bits & smgpost::SET_SEGMENTED)) return;
  std::vector<Ali> &v = sh.side[w];
  for (int g = 0; g < t.nsegments && (size_t)g + 1 < t.segment_begin.size(); g++) {
    int last = 0;
    for (int k = t.segment_begin[(size_t)g]; k < t.segment_begin[(size_t)g + 1]; k++) {
      const int row = t.by_segment[(size_t)k];
      if (t.bits[(size_t)row] & smgpost::WITHHELD) continue;
      if (t.score[(size_t)row] < last) break;
      last = t.score[(size_t)row];
      const Ali a = ali_of(t, row, t.quality[(size_t)row], MF_PAIRED | MF_PARTIAL | (w ? (uint32_t)MF_2NDMATE : 0u));
      bool known = false;
      for (const Ali &b : v)
        if (a.ss == b.ss && a.se == b.se && a.sidx == b.sidx && a.qs == b.qs && a.qe == b.qe && (a.status & (MF_REVERSE | MF_2NDMATE)) == (b.status & (MF_REVERSE | MF_2NDMATE))) { known = true; break; }
      if (!known) v.push_back(a);
    }
  }
}

struct PairJob {
  const smaltgpu_pairs *pairs;
  ReadCtx cx[2];
  const smaltgpu_report_opts *op;
  const smaltgpu_pair_opts *po;
};

// everything for pair p up to the choice; -> number of random draws the choice needs (counting mode: dr.values == nullptr)
bool pair_entries(const PairJob &jb, uint32_t p, Table &A, Table &B, smgpairs::Join &join, std::vector<Entry> &entries, smgpairs::Draws &dr) {
  const smgpairs::PairBlock &blk = jb.pairs->blk;
  A.unpack(blk.packed.data(2 * (size_t)p), blk.packed.size(2 * (size_t)p));
  B.unpack(blk.packed.data(2 * (size_t)p + 1), blk.packed.size(2 * (size_t)p + 1));
  const smgpairs::PairPlan &pl = blk.plan[p];
  join.pairs.clear();
  if (!pl.idle) {                         // a pair of two mates shorter than a word returns before the pairing (rmap.c:1833-1834)
    if (!join.run(A, B, pl.state, jb.po->library, jb.po->insert_min, jb.po->insert_max)) return false;
    A.apply_output_filter(jb.op->min_swscor, jb.op->min_swscor_below_max, jb.op->min_identity, (uint32_t)(jb.cx[0].rv->read_off[p + 1] - jb.cx[0].rv->read_off[p]));
    B.apply_output_filter(jb.op->min_swscor, jb.op->min_swscor_below_max, jb.op->min_identity, (uint32_t)(jb.cx[1].rv->read_off[p + 1] - jb.cx[1].rv->read_off[p]));
  }
  smgpairs::choose(entries, join, A, B, pl.state, jb.op->outflags, dr);
  return true;
}

bool pair_lines(std::string &o, const PairJob &jb, uint32_t p, const Table &A, const Table &B, const std::vector<Entry> &entries, PairSheet &sh) {
  sh.clear();
  for (const Entry &e : entries) sheet_add(sh, e, A, B);
  if ((jb.op->outflags & SMALTGPU_OUT_BEST) && (jb.op->outflags & SMALTGPU_OUT_SPLIT)) { sheet_add_partial(sh, 0, A); sheet_add_partial(sh, 1, B); }
  auto line = [&](int w, const Ali &a, const PairSide *ps) { return print_line(o, jb.cx[w], p, a, ps); };
  for (int w = 0; w < 2; w++) sh.printed[w].assign(sh.side[w].size(), 0);
  for (const PairSheet::Link &ln : sh.links) {                       // reportWrite (report.c:1758-1867): the pairs first ...
    const Ali &a = sh.side[0][(size_t)ln.ia], &b = sh.side[1][(size_t)ln.ib];
    sh.printed[0][(size_t)ln.ia] = sh.printed[1][(size_t)ln.ib] = 1;
    const PairSide pa{&b, ln.isize, ln.pairflg}, pb{&a, ln.isize, ln.pairflg};
    if (!line(0, a, &pa) || !line(1, b, &pb)) return false;
  }
  const PairSide rest{nullptr, 0, sh.links.empty() ? (uint8_t)0 : sh.links[0].pairflg};       // ... then what no pair printed
  for (int w = 0; w < 2; w++)
    for (size_t k = 0; k < sh.side[w].size(); k++) if (!sh.printed[w][k] && !line(w, sh.side[w][k], &rest)) return false;
  return true;
}

}  // namespace

extern "C" int smaltgpu_report_emit_pairs(smaltgpu_report *rp, const smaltgpu_pairs *pairs, const smaltgpu_reads_view *reads, const smaltgpu_reads_view *mates,
                                          const char *const *seqnames, int64_t nseq, const smaltgpu_report_opts *op, const smaltgpu_pair_opts *po, int nthreads,
                                          const char **text, uint64_t *len) {
  if (!rp || !pairs || !reads || !mates || !seqnames || !op || !po || !text || !len) return smaltgpu_set_error(SMALTGPU_EARG, "smaltgpu_report_emit_pairs: null argument");
  const uint32_t n = pairs->blk.npairs;
  if (reads->nreads != n || mates->nreads != n) return smaltgpu_set_error(SMALTGPU_EARG, "smaltgpu_report_emit_pairs: reads, mates and mapped pairs differ in number");
  if (int e = check_format(rp, op, nseq)) return e;
  PairJob jb{pairs, {ReadCtx{nullptr, reads, op, seqnames, nseq, rp->seqlen.data()}, ReadCtx{nullptr, mates, op, seqnames, nseq, rp->seqlen.data()}}, op, po};
  if (nthreads < 1) nthreads = 1;
  if ((uint32_t)nthreads > n / 256 + 1) nthreads = (int)(n / 256 + 1);
  const bool drawing = (op->outflags & SMALTGPU_OUT_RANDSEL) != 0;
  // pass 1: every pair that needs no random number is printed; the others say how many they need
  struct Slice { uint32_t part; uint64_t at, len; };
  std::vector<Slice> where(n ? n : 1);
  std::vector<uint8_t> need(n ? n : 1, 0);
  rp->part.assign((size_t)nthreads + 1, std::string());
  std::vector<int64_t> bad((size_t)nthreads, -1);
  auto pass = [&](int t, uint32_t lo, uint32_t hi, bool second, const std::vector<uint32_t> *todo, const std::vector<double> *values, const std::vector<uint64_t> *value_at) {
    std::string &o = rp->part[(size_t)t];
    Table A, B;
    smgpairs::Join join;
    std::vector<Entry> entries;
    PairSheet sh;
    for (uint32_t j = lo; j < hi; j++) {
      const uint32_t p = second ? (*todo)[j] : j;
      smgpairs::Draws dr;
      if (second) { dr.values = values->data() + (*value_at)[j]; dr.have = need[p]; }
      if (!pair_entries(jb, p, A, B, join, entries, dr)) { if (bad[(size_t)t] < 0) bad[(size_t)t] = p; continue; }
      if (!second && drawing && dr.used > 0) { need[p] = (uint8_t)dr.used; continue; }
      const uint64_t at = o.size();
      if (!pair_lines(o, jb, p, A, B, entries, sh)) { if (bad[(size_t)t] < 0) bad[(size_t)t] = p; continue; }
      where[p] = Slice{(uint32_t)t, at, o.size() - at};
    }
  };
  {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) {
      const uint32_t lo = (uint32_t)((uint64_t)n * t / nthreads), hi = (uint32_t)((uint64_t)n * (t + 1) / nthreads);
      rp->part[(size_t)t].reserve((size_t)(hi - lo) * (op->format == SMALTGPU_FMT_SAM ? 840 : 192));
      if (nthreads == 1) pass(0, lo, hi, false, nullptr, nullptr, nullptr); else th.emplace_back(pass, t, lo, hi, false, nullptr, nullptr, nullptr);
    }
    for (std::thread &x : th) x.join();
  }
  // pass 2: the random numbers in pair order, as one thread of the reference draws them; then the pairs that waited for them
  if (drawing) {
    std::vector<uint32_t> todo;
    std::vector<uint64_t> value_at;
    std::vector<double> values;
    for (uint32_t p = 0; p < n; p++) if (need[p]) { todo.push_back(p); value_at.push_back(values.size()); for (int k = 0; k < need[p]; k++) values.push_back(drand48()); }
    if (!todo.empty()) {
      const int t = nthreads;                               // the extra part
      const uint64_t before = rp->part[(size_t)t].size();
      (void)before;
      bad.push_back(-1);
      // few pairs wait: one thread
      std::string &o = rp->part[(size_t)t];
      o.clear();
      pass(t, 0, (uint32_t)todo.size(), true, &todo, &values, &value_at);
    }
  }
  for (int64_t b : bad) if (b >= 0) { char m[128]; snprintf(m, sizeof(m), "smaltgpu_report_emit_pairs: pair %lld: inconsistent alignment sets or alignment strings", (long long)b); return smaltgpu_set_error(SMALTGPU_EINTERNAL, m); }
  // stitch in pair order
  size_t tot = 0;
  for (const std::string &s_ : rp->part) tot += s_.size();
  rp->text.clear();
  rp->text.reserve(tot);
  for (uint32_t p = 0; p < n; p++) rp->text.append(rp->part[where[p].part], where[p].at, where[p].len);
  *text = rp->text.data(); *len = rp->text.size();
  return SMALTGPU_OK;
}
