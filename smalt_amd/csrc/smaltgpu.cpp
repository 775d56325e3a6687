// smaltgpu.cpp -- host side of libsmaltgpu.so: the C ABI of include/smaltgpu.h on top of the
// gfx950 kernels.  Device memory, one HIP stream and all scratch are owned by the mapper (the
// analogue of the reference's RMap, rmap.c:130-171, which owns every per-thread buffer).
// There is no CPU code path: every entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/smaltgpu.h"
#include "smg_dump.hpp"
#include "smg_indexfile.hpp"
#include "smg_indexbuild.h"
#include "smg_kernels.h"

using namespace smg;

static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(SMALTGPU_ENODEV, "%s: %s", #x, hipGetErrorString(e_)); } while (0)

extern "C" const char *smaltgpu_last_error(void) { return g_err.c_str(); }
extern "C" int smaltgpu_set_error(int code, const char *msg) { return fail(code, "%s", msg ? msg : ""); }   // for the library's other translation units
extern "C" int smaltgpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ------------------------------------------------------------------------------------------
struct smaltgpu_index {
  DevIndex d;
  int device = 0;
  bool owns = true;
  std::vector<uint64_t> sop;       // host copy
  void *bufs[8] = {nullptr};
  int nbufs = 0;
  std::vector<std::string> names;  // set by smaltgpu_index_build (what smaltgpu_index_save writes)
  uint32_t maxpos = 0;
  bool built = false;
  std::vector<const char *> name_ptrs;  // smaltgpu_index_seqnames
  std::vector<uint32_t> packed_host;   // host copy of the packed reference, fetched on first request (smaltgpu_index_packed_host)
  std::mutex packed_mu;
};

template <class T>
static int upload(smaltgpu_index *ix, const T *host, size_t n, const T **dev) {
  T *p = nullptr;
  HIPCHK(hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)));
  ix->bufs[ix->nbufs++] = p;
  if (n) HIPCHK(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
  *dev = p;
  return 0;
}

extern "C" int smaltgpu_index_create(smaltgpu_index **out, const smaltgpu_index_desc *ds, int device) {
  if (!out || !ds || !ds->idx || !ds->pos || !ds->sop || !ds->packed) return fail(SMALTGPU_EARG, "null argument");
  // Any number of sequences: only the sequence-by-sequence mode packs a sequence number into the hit sort key (KEY_SEQBITS),
  // and that mode is limited to fewer than 512 sequences as in the reference (smalt.c:599; check_par).
  if (ds->k < 1 || ds->k > 21 || ds->s < 1 || ds->nseq < 1 || ds->nseq > 0x7ffffffeLL)
    return fail(SMALTGPU_EARG, "unsupported index geometry (k=%d s=%d nseq=%lld)", ds->k, ds->s, (long long)ds->nseq);
  HIPCHK(hipSetDevice(device));
  smaltgpu_index *ix = new smaltgpu_index();
  ix->device = device;
  DevIndex &d = ix->d;
  d.k = ds->k; d.s = ds->s; d.typ = ds->typ;
  d.nbits_key = ds->typ == IDX_PERFECT ? 2 * ds->k : ds->nbits_key;
  d.nbits_lo = ds->typ == IDX_PERFECT ? 0 : ds->nbits_lo;
  d.nkeys = 1u << d.nbits_key; d.npos = ds->npos; d.nwords = ds->nwords; d.nseq = (int32_t)ds->nseq;
  ix->sop.assign(ds->sop, ds->sop + ds->nseq + 1);
  d.totlen = ix->sop.back();
  int rv = 0;
  if (ds->on_device) {
    ix->owns = false;
    d.idx = ds->idx; d.pos = ds->pos; d.wordidx = ds->wordidx; d.posidx = ds->posidx; d.packed = ds->packed;
  } else {
    rv = upload(ix, ds->idx, (size_t)d.nkeys + 1, &d.idx);
    if (!rv) rv = upload(ix, ds->pos, (size_t)d.npos, &d.pos);
    if (!rv && d.typ != IDX_PERFECT) { rv = upload(ix, ds->wordidx, (size_t)d.nwords + 1, &d.wordidx); if (!rv) rv = upload(ix, ds->posidx, (size_t)d.nwords + 1, &d.posidx); }
    if (!rv) rv = upload(ix, ds->packed, (size_t)(d.totlen / 10 + 1), &d.packed);
  }
  std::vector<uint32_t> seqlo((size_t)d.nseq + 1);
  for (int i = 0; i <= d.nseq; i++) seqlo[(size_t)i] = (uint32_t)(ix->sop[(size_t)i] / (uint64_t)d.s);
  if (!rv) rv = upload(ix, ix->sop.data(), ix->sop.size(), &d.sop);
  if (!rv) rv = upload(ix, seqlo.data(), seqlo.size(), &d.seqlo);
  if (rv) { smaltgpu_index_free(ix); return rv; }
  *out = ix;
  return SMALTGPU_OK;
}

extern "C" int smaltgpu_index_load(smaltgpu_index **out, const char *prefix, int device) {
  if (!out || !prefix) return fail(SMALTGPU_EARG, "null argument");
  HostIndex h;
  std::string err;
  if (!read_index_files(prefix, h, err)) return fail(SMALTGPU_EFILE, "%s", err.c_str());
  smaltgpu_index_desc ds;
  memset(&ds, 0, sizeof(ds));
  ds.k = h.k; ds.s = h.s; ds.typ = h.typ; ds.nbits_key = h.nbits_key; ds.nbits_lo = h.nbits_lo; ds.npos = h.npos; ds.nwords = h.nwords;
  ds.idx = h.idx.data(); ds.pos = h.pos.data(); ds.wordidx = h.wordidx.data(); ds.posidx = h.posidx.data();
  ds.nseq = h.nseq; ds.sop = h.sop.data(); ds.packed = h.packed.data(); ds.on_device = 0;
  const int rv = smaltgpu_index_create(out, &ds, device);
  if (!rv) (*out)->names = h.names;
  return rv;
}

extern "C" int smaltgpu_index_seqnames(const smaltgpu_index *cix, const char *const **names, const uint64_t **sop, int64_t *nseq) {
  smaltgpu_index *ix = const_cast<smaltgpu_index *>(cix);
  if (!ix || !names || !sop || !nseq) return fail(SMALTGPU_EARG, "null argument");
  if ((int64_t)ix->names.size() != ix->d.nseq) return fail(SMALTGPU_EARG, "the index was adopted from arrays without sequence names");
  std::lock_guard<std::mutex> lk(ix->packed_mu);
  if (ix->name_ptrs.empty()) for (const std::string &x : ix->names) ix->name_ptrs.push_back(x.c_str());
  *names = ix->name_ptrs.data(); *sop = ix->sop.data(); *nseq = ix->d.nseq;
  return SMALTGPU_OK;
}

// A second image of an index on another device, copied device to device (xGMI between the GPUs of a node) instead of being
// read from disk and uploaded again: the reference's worker threads share ONE read-only index (threads.c:793-985, rmapCreate
// receives the same HashTable/SeqSet pointers); with one image per GPU the images are filled from the first one.
extern "C" int smaltgpu_index_clone(smaltgpu_index **out, const smaltgpu_index *src, int device) {
  if (!out || !src) return fail(SMALTGPU_EARG, "null argument");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(SMALTGPU_EARG, "no device %d", device);
  const DevIndex &d = src->d;
  const uint32_t *sp[5] = {d.idx, d.pos, d.typ != IDX_PERFECT ? d.wordidx : nullptr, d.typ != IDX_PERFECT ? d.posidx : nullptr, d.packed};
  const size_t sn[5] = {(size_t)d.nkeys + 1, (size_t)d.npos, (size_t)d.nwords + 1, (size_t)d.nwords + 1, (size_t)(d.totlen / 10 + 1)};
  uint32_t *dp[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  HIPCHK(hipSetDevice(device));
  if (device != src->device) {                     // direct access between the two GPUs (ignored when already enabled or not available: the copy is then staged by the runtime)
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, device, src->device) == hipSuccess && can) { const hipError_t e = hipDeviceEnablePeerAccess(src->device, 0); if (e != hipSuccess) (void)hipGetLastError(); }
  }
  int rv = 0;
  for (int i = 0; i < 5 && !rv; i++) {
    if (!sp[i]) continue;
    if (hipMalloc((void **)&dp[i], (sn[i] ? sn[i] : 1) * 4) != hipSuccess) { rv = fail(SMALTGPU_ENOMEM, "device memory for the index copy"); break; }
    if (sn[i] && hipMemcpyPeer(dp[i], device, sp[i], src->device, sn[i] * 4) != hipSuccess) rv = fail(SMALTGPU_ENODEV, "device-to-device copy of the index failed");
  }
  smaltgpu_index *ix = nullptr;
  if (!rv) {
    smaltgpu_index_desc ds;
    memset(&ds, 0, sizeof(ds));
    ds.k = d.k; ds.s = d.s; ds.typ = d.typ; ds.nbits_key = d.nbits_key; ds.nbits_lo = d.nbits_lo; ds.npos = d.npos; ds.nwords = d.nwords;
    ds.idx = dp[0]; ds.pos = dp[1]; ds.wordidx = dp[2]; ds.posidx = dp[3]; ds.packed = dp[4]; ds.nseq = d.nseq; ds.sop = src->sop.data(); ds.on_device = 1;
    rv = smaltgpu_index_create(&ix, &ds, device);
  }
  if (rv) { for (uint32_t *p : dp) if (p) (void)hipFree(p); return rv; }
  for (uint32_t *p : dp) if (p && ix->nbufs < 8) ix->bufs[ix->nbufs++] = p;      // adopted: freed with the index
  ix->names = src->names; ix->maxpos = src->maxpos; ix->built = src->built;
  *out = ix;
  return SMALTGPU_OK;
}

// ---- index construction on the device (smg_indexbuild.hip) ----
extern "C" int smaltgpu_index_build_device(smaltgpu_index **out, int device, const uint8_t *d_bases, const uint64_t *seq_off, const char *const *names,
                                           int64_t nseq, int32_t k, int32_t s, float *build_ms) {
  if (!out || !d_bases || !seq_off || !names) return fail(SMALTGPU_EARG, "null argument");
  if (k < 1 || k > 21 || s < 1 || nseq < 1 || nseq > 0x7ffffffeLL) return fail(SMALTGPU_EARG, "unsupported index geometry (k=%d s=%d nseq=%lld)", k, s, (long long)nseq);
  for (int64_t i = 0; i < nseq; i++) if (seq_off[i + 1] - seq_off[i] > 0x7fffffffull) return fail(SMALTGPU_EARG, "sequence %lld is longer than 2^31-1 bases (hashidx.c:592)", (long long)i);
  HIPCHK(hipSetDevice(device));
  BuiltIndex b;
  char err[256] = "";
  if (build_index_device(d_bases, seq_off[nseq], seq_off, (int)nseq, k, s, &b, err, sizeof(err))) return fail(SMALTGPU_EARG, "index construction failed: %s", err);
  smaltgpu_index_desc ds;
  memset(&ds, 0, sizeof(ds));
  ds.k = k; ds.s = s; ds.typ = b.typ; ds.nbits_key = b.nbits_key; ds.nbits_lo = b.nbits_lo; ds.npos = b.npos; ds.nwords = b.nwords;
  ds.idx = b.idx; ds.pos = b.pos; ds.wordidx = b.wordidx; ds.posidx = b.posidx; ds.nseq = nseq; ds.sop = seq_off; ds.packed = b.packed; ds.on_device = 1;
  smaltgpu_index *ix = nullptr;
  const int rv = smaltgpu_index_create(&ix, &ds, device);
  if (rv) { (void)hipFree(b.idx); (void)hipFree(b.pos); (void)hipFree(b.wordidx); (void)hipFree(b.posidx); (void)hipFree(b.packed); return rv; }
  void *own[5] = {b.idx, b.pos, b.wordidx, b.posidx, b.packed};      // adopted: freed with the index
  for (void *p : own) if (p && ix->nbufs < 8) ix->bufs[ix->nbufs++] = p;
  // The reference's reader takes only 2 * nwords + 1 of the collision words (hashidx.c:1257; smg_indexfile.hpp), so a loaded
  // index runs with posidx[nwords] == 0; the built image must behave the same.  smaltgpu_index_save writes npos there.
  if (b.typ != IDX_PERFECT && b.posidx) HIPCHK(hipMemset(b.posidx + b.nwords, 0, 4));
  ix->maxpos = b.maxpos; ix->built = true;
  for (int64_t i = 0; i < nseq; i++) ix->names.emplace_back(names[i] ? names[i] : "");
  if (build_ms) *build_ms = b.build_ms;
  *out = ix;
  return SMALTGPU_OK;
}

extern "C" int smaltgpu_index_build(smaltgpu_index **out, int device, const uint8_t *bases, const uint64_t *seq_off, const char *const *names,
                                    int64_t nseq, int32_t k, int32_t s, float *build_ms) {
  if (!out || !bases || !seq_off || !names || nseq < 1) return fail(SMALTGPU_EARG, "null argument");
  HIPCHK(hipSetDevice(device));
  uint8_t *d = nullptr;
  const uint64_t tot = seq_off[nseq];
  HIPCHK(hipMalloc((void **)&d, tot ? tot : 1));
  if (hipMemcpy(d, bases, tot, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return fail(SMALTGPU_ENODEV, "copy of the reference to the device failed"); }
  const int rv = smaltgpu_index_build_device(out, device, d, seq_off, names, nseq, k, s, build_ms);
  (void)hipFree(d);
  return rv;
}

// seqSetWriteBinFil (sequence.c:2448-2519) + hashTableWrite (hashidx.c:1214-1255) in the container of filio.c:48-77
extern "C" int smaltgpu_index_save(const smaltgpu_index *ix, const char *prefix) {
  if (!ix || !prefix) return fail(SMALTGPU_EARG, "null argument");
  if (!ix->built) return fail(SMALTGPU_EARG, "only an index made by smaltgpu_index_build can be saved");
  HIPCHK(hipSetDevice(ix->device));
  const DevIndex &d = ix->d;
  auto container = [](FILE *fp, uint64_t siz, uint32_t typ, uint32_t version, const uint32_t *h8) {
    uint32_t f[12] = {0x73212173u, 0x6E378A19u, (uint32_t)(siz + 12), typ, version, 8, 0, 0, 0, 0, 0, 0};
    return fwrite(f, 4, 12, fp) == 12 && fwrite(h8, 4, 8, fp) == 8;
  };
  auto put_dev = [](FILE *fp, const uint32_t *dev, size_t n) {        // device words to the file, 64 MB at a time
    std::vector<uint32_t> buf(n < (16u << 20) ? n : (16u << 20));
    for (size_t o = 0; o < n;) {
      const size_t c = n - o < buf.size() ? n - o : buf.size();
      if (hipMemcpy(buf.data(), dev + o, c * 4, hipMemcpyDeviceToHost) != hipSuccess || fwrite(buf.data(), 4, c, fp) != c) return false;
      o += c;
    }
    return true;
  };
  std::string nm;
  for (const std::string &x : ix->names) { nm += x; nm.push_back('\0'); }
  const uint64_t namsiz = nm.size(), nseq = (uint64_t)d.nseq, seqsiz = d.totlen / 10 + 1;
  uint32_t h[8] = {(uint32_t)nseq, (uint32_t)(nseq >> 32), (uint32_t)namsiz, (uint32_t)(namsiz >> 32), (uint32_t)d.totlen, (uint32_t)(d.totlen >> 32), 2u /* compressed */, 0};
  FILE *fp = fopen((std::string(prefix) + ".sma").c_str(), "wb");
  if (!fp) return fail(SMALTGPU_EFILE, "cannot write %s.sma", prefix);
  std::vector<uint32_t> seqlen((size_t)nseq);
  for (uint64_t i = 0; i < nseq; i++) seqlen[(size_t)i] = (uint32_t)(ix->sop[(size_t)i + 1] - ix->sop[(size_t)i]);
  bool ok = container(fp, 8 + seqsiz + nseq + ((namsiz - 1) / 4 + 1), 1, 4, h) && fwrite(nm.data(), 1, namsiz, fp) == namsiz &&
            fwrite(seqlen.data(), 4, (size_t)nseq, fp) == (size_t)nseq && put_dev(fp, d.packed, (size_t)seqsiz);
  ok = (fclose(fp) == 0) && ok;
  if (!ok) return fail(SMALTGPU_EFILE, "short write to %s.sma", prefix);
  uint32_t g[8] = {(uint32_t)d.k, (uint32_t)d.s, d.npos, ix->maxpos, (uint32_t)d.typ, (uint32_t)d.nbits_key, (uint32_t)d.nbits_lo, d.nwords};
  uint64_t totsiz = (uint64_t)d.npos + d.nkeys + 1;
  if (d.typ != IDX_PERFECT) totsiz += ((uint64_t)d.nwords + 1) * 2;
  fp = fopen((std::string(prefix) + ".smi").c_str(), "wb");
  if (!fp) return fail(SMALTGPU_EFILE, "cannot write %s.smi", prefix);
  ok = container(fp, totsiz, 2, 3, g) && put_dev(fp, d.idx, (size_t)d.nkeys + 1) && put_dev(fp, d.pos, d.npos);
  if (ok && d.typ != IDX_PERFECT) ok = put_dev(fp, d.wordidx, (size_t)d.nwords + 1) && put_dev(fp, d.posidx, (size_t)d.nwords) && fwrite(&d.npos, 4, 1, fp) == 1;   // posidx[nwords] = npos in the file (hashidx.c:989)
  ok = (fclose(fp) == 0) && ok;
  if (!ok) return fail(SMALTGPU_EFILE, "short write to %s.smi", prefix);
  return SMALTGPU_OK;
}

extern "C" void smaltgpu_index_free(smaltgpu_index *ix) {
  if (!ix) return;
  (void)hipSetDevice(ix->device);
  for (int i = 0; i < ix->nbufs; i++) (void)hipFree(ix->bufs[i]);
  delete ix;
}

// A host copy of the packed reference (what smaltgpu_postprocess needs to cut alignments at sequence junctions): fetched from
// the device once, on first request.
extern "C" const uint32_t *smaltgpu_index_packed_host(const smaltgpu_index *cix) {
  smaltgpu_index *ix = const_cast<smaltgpu_index *>(cix);
  if (!ix) return nullptr;
  std::lock_guard<std::mutex> lk(ix->packed_mu);
  if (ix->packed_host.empty()) {
    const size_t nw = (size_t)(ix->d.totlen / 10 + 1);
    std::vector<uint32_t> h(nw);
    if (hipSetDevice(ix->device) != hipSuccess || hipMemcpy(h.data(), ix->d.packed, nw * 4, hipMemcpyDeviceToHost) != hipSuccess) { fail(SMALTGPU_ENODEV, "copy of the packed reference to the host failed"); return nullptr; }
    ix->packed_host.swap(h);
  }
  return ix->packed_host.data();
}

extern "C" int smaltgpu_index_info(const smaltgpu_index *ix, smaltgpu_index_desc *o) {
  if (!ix || !o) return fail(SMALTGPU_EARG, "null argument");
  memset(o, 0, sizeof(*o));
  o->k = ix->d.k; o->s = ix->d.s; o->typ = ix->d.typ; o->nbits_key = ix->d.nbits_key; o->nbits_lo = ix->d.nbits_lo;
  o->npos = ix->d.npos; o->nwords = ix->d.nwords; o->idx = ix->d.idx; o->pos = ix->d.pos; o->wordidx = ix->d.wordidx;
  o->posidx = ix->d.posidx; o->nseq = ix->d.nseq; o->sop = ix->sop.data(); o->packed = ix->d.packed; o->on_device = 1;
  return SMALTGPU_OK;
}

extern "C" void smaltgpu_params_default(smaltgpu_params *p, const smaltgpu_index *ix) {
  memset(p, 0, sizeof(*p));
  p->ktuple_maxhit = 10000;                         // menu.c:603
  p->min_cover = 0;
  p->min_swatscor = ix->d.k + ix->d.s - 1;          // smalt.c:608-615
  p->min_swatscor_below_max = 0;
  p->min_basqval = 0;
  p->target_depth = 512; p->max_depth = 2048;       // smalt.c:60-61
  p->rmapflg = SMALTGPU_FLG_BEST | (ix->d.nseq < 512 ? SMALTGPU_FLG_SEQBYSEQ : 0);   // smalt.c:495-497, 599
  p->match = 1; p->mismatch = -2; p->gap_init = -4; p->gap_ext = -3;                  // score.c:41-47
}

// ------------------------------------------------------------------------------------------
enum { T_ENCODE = 0, T_SEED, T_HITS, T_CANDS, T_SW_FULL, T_SW_SCALAR, T_REPLAY, T_ALIGN, T_NUM };
static const char *const kTimerNames[T_NUM] = {"encode", "seed", "hits", "cands", "sw_full", "sw_scalar", "replay", "align"};
extern "C" const char *smaltgpu_timer_name(int i) { return (i >= 0 && i < T_NUM) ? kTimerNames[i] : nullptr; }

struct smaltgpu_mapper {
  const smaltgpu_index *ix = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  uint32_t max_reads = 0, max_len = 0, qmax = 0;
  uint64_t max_bases = 0;
  // device buffers
  uint8_t *d_bases = nullptr, *d_quals = nullptr, *d_codes = nullptr, *d_codes_rc = nullptr;
  uint64_t *d_off = nullptr;
  uint32_t *d_ids = nullptr;
  uint32_t hits_W = 0, hits_wg = 0;          // k_hits: keys per window, workgroups (0: S3 stays inside k_cands)
  HitRun *b_hitrun = nullptr;                 // the run table (Batch::hitrun is set per call: restricted calls do not use it)                  // read ids of a round gathered from resident batches (smaltgpu_map_batch_ctx_resident)
  Batch b;
  // Counters of a batch, one 128-byte line each: atomics on one line are served one after the other (2 ns each), and the pool cursors,
  // the work-queue cursors of the persistent kernels and the retry counts used to share the first line.
  enum : size_t { CT_LINE = 128, CT_RC = 0, CT_RES = 1 * CT_LINE, CT_DSTR = 2 * CT_LINE, CT_ERR = 3 * CT_LINE, CT_NEXT = 4 * CT_LINE /* 5 lines */,
                  CT_ALIGN_RETRY = 9 * CT_LINE, CT_CANDS_RETRY = 10 * CT_LINE, CT_HITS = 11 * CT_LINE, CT_HITS_CURSOR = 12 * CT_LINE, CT_STRIP_CURSOR = 13 * CT_LINE,
                  CT_WORK = 14 * CT_LINE /* 32 x 8 bytes */, CT_BYTES = 16 * CT_LINE };
  uint8_t *d_counters = nullptr;
  uint8_t *seed_scr = nullptr; size_t seed_bytes = 0; uint32_t seed_slots = 0;
  uint8_t *cand_scr = nullptr; size_t cand_bytes = 0; uint32_t cand_slots = 0;
  CandGeom cg2; uint8_t *cand_scr2 = nullptr; size_t cand_bytes2 = 0; uint32_t cand_slots2 = 0;   // second pass of the candidate stage: full-size slots
  CandGeom cg;
  uint8_t *cand_scr_dbg = nullptr; uint32_t cand_dbg_reads = 0;
  int *sw_rows = nullptr; uint32_t sw_rowlen = 0, sw_threads = 0;
  void *strip_bnd = nullptr; uint8_t *strip_win = nullptr; uint32_t strip_grid = 0;   // k_sw_strip: boundary columns + decoded window per workgroup
  uint8_t *align_scr = nullptr; size_t align_bytes = 0; uint32_t align_slots = 0;
  uint8_t *align_scr2 = nullptr; size_t align_bytes2 = 0; uint32_t align_slots2 = 0; uint64_t dircap2 = 0;   // second K3 pass: few slots with full-size direction matrices
  uint32_t wincap = 0, rescap_slot = 0, dstrcap_slot = 0; uint64_t dircap = 0;
  uint32_t rescap_slot2 = 0, dstrcap_slot2 = 0;       // result slots of the second K3 pass (reads with more alignments than a first-pass slot holds)
  // host mirrors
  // results come back through pinned host memory: the copies are asynchronous, so the next batch can be launched
  // behind them (smaltgpu_fetch_begin / _end)
  template <class T> struct Pinned {
    T *p = nullptr; size_t cap = 0;
    int ensure(size_t n) {
      if (n <= cap) return 0;
      size_t c = cap ? cap : 1024;
      while (c < n) c += c / 2 + 1;
      T *q = nullptr;
      if (hipHostMalloc((void **)&q, c * sizeof(T), hipHostMallocDefault) != hipSuccess) return -1;
      if (p) (void)hipHostFree(p);
      p = q; cap = c;
      return 0;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    T *data() { return p; }
    T &operator[](size_t i) { return p[i]; }
  };
  Pinned<ReadStat> h_stat;
  Pinned<Result> h_res;
  Pinned<uint8_t> h_dstr;
  hipEvent_t ev_fetch = nullptr;
  uint32_t fetch_n = 0; uint64_t fetch_nres = 0; bool fetch_open = false, pool_overflow = false;
  // smaltgpu_map_batch after a pool overflow: results of the whole batch assembled from several device batches
  std::vector<smaltgpu_result> fin_res; std::vector<uint8_t> fin_dstr; std::vector<smaltgpu_readstat> fin_stat; std::vector<uint64_t> fin_off;
  std::vector<uint64_t> h_res_off;
  std::vector<smaltgpu_result> o_res;
  std::vector<smaltgpu_readstat> o_stat;
  std::vector<uint64_t> h_off;              // read offsets of the last batch (host copy)
  uint32_t last_n = 0;
  MapPar last_par;
  bool have_host_off = false;
  int debug = 0;
  // per-read context of rmapPair's rounds (smaltgpu_map_batch_ctx): device copies, grown on demand
  struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    int ensure(size_t n) { if (n <= cap) return 0; if (p) (void)hipFree(p); p = nullptr; cap = 0; size_t c = n + n / 2 + 256; if (hipMalloc(&p, c) != hipSuccess) return -1; cap = c; return 0; }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  };
  DevBuf cx_ivoff, cx_iv, cx_minsw, cx_prevmax, cx_fineidx, cx_finepos, cx_fineoff, cx_alloclen, cx_seedrange;
  int host_threads = 1;                      // smaltgpu_mapper_set_host_threads
  bool history = false;                      // serial-order mode (smaltgpu_mapper_set_history)
  uint32_t hist_longest = 0;                 // longest read of length >= k of the run so far
  std::vector<uint32_t> h_alloclen;
  std::vector<uint32_t> h_ivoff, h_fineoff;
  std::vector<HitInfoHdr> h_hi;
  bool last_fine = false;
  uint64_t remap_batches = 0;               // device batches smaltgpu_map_batch ran to recover from pool overflows (diagnostic)
  hipEvent_t ev[T_NUM + 1] = {nullptr};
  double ms[T_NUM] = {0};
  unsigned long long work[WK_NWORK] = {0};
};

template <class T>
static int dalloc(T **p, size_t n) {
  HIPCHK(hipMalloc((void **)p, (n ? n : 1) * sizeof(T)));
  return 0;
}

static uint64_t next_pow2(uint64_t v) { uint64_t p = 1; while (p < v && p < (1ull << 62)) p <<= 1; return p; }

extern "C" int smaltgpu_mapper_create(smaltgpu_mapper **out, const smaltgpu_index *ix, uint32_t max_batch_reads, uint32_t max_read_len) {
  return smaltgpu_mapper_create_ex(out, ix, max_batch_reads, max_read_len, nullptr);
}

extern "C" int smaltgpu_mapper_create_ex(smaltgpu_mapper **out, const smaltgpu_index *ix, uint32_t max_batch_reads, uint32_t max_read_len,
                                         const smaltgpu_mapper_opts *opts) {
  if (!out || !ix || !max_batch_reads || !max_read_len) return fail(SMALTGPU_EARG, "bad argument");
  const uint32_t opt_cands = opts ? opts->cands_per_read : 0, opt_budget = opts ? opts->slot_budget_gb : 0;
  if (max_read_len >= (1u << KEY_QBITS)) return fail(SMALTGPU_EARG, "reads longer than %u bases are not supported", (1u << KEY_QBITS) - 1);
  HIPCHK(hipSetDevice(ix->device));
  smaltgpu_mapper *m = new smaltgpu_mapper();
  m->ix = ix; m->device = ix->device; m->max_reads = max_batch_reads; m->max_len = max_read_len;
  m->qmax = (max_read_len + 8 + 7) & ~7u;
  m->max_bases = (uint64_t)max_batch_reads * max_read_len;
  HIPCHK(hipStreamCreate(&m->stream));
  for (int i = 0; i <= T_NUM; i++) HIPCHK(hipEventCreate(&m->ev[i]));
  const DevIndex &d = ix->d;
  Batch &b = m->b;
  memset(&b, 0, sizeof(b));
  b.qmax = m->qmax;
  int rv = 0;
#define DA(ptr, n) if (!rv) rv = dalloc(&(ptr), (size_t)(n))
  DA(m->d_bases, m->max_bases + 16); DA(m->d_quals, m->max_bases + 16); DA(m->d_codes, m->max_bases + 16); DA(m->d_codes_rc, m->max_bases + 16);
  DA(m->d_off, (size_t)max_batch_reads + 1);
  DA(m->d_ids, (size_t)max_batch_reads + 1);
  DA(b.hi, 2 * (size_t)max_batch_reads);
  DA(b.seeds, 2 * (size_t)max_batch_reads * m->qmax);
  DA(b.qmask, 2 * (size_t)max_batch_reads * m->qmax);
  DA(b.ch, max_batch_reads); DA(b.ctl, max_batch_reads); DA(b.stat, max_batch_reads);
  {
    const char *e = getenv("SMALTGPU_CANDS_PER_READ");        // the environment overrides the caller's option (test hook)
    const bool sized = e || opt_cands;
    uint64_t per = e ? strtoull(e, nullptr, 10) : (opt_cands ? opt_cands : 640);      // ranked candidates per read: the depth cut leaves 270 on average at 3 Gbp, at most 2048
    if (per < 8) per = 8;
    if (per > 2048) per = 2048;
    uint64_t cap = (uint64_t)max_batch_reads * per;
    if (max_batch_reads <= 4096 && !sized) cap = (uint64_t)max_batch_reads * 2048;
    if (cap < (uint64_t)MAXIMUM_DEPTH) cap = MAXIMUM_DEPTH;      // one read alone always fits: smaltgpu_map_batch re-maps the reads of an overflowing batch in smaller batches
    if (cap > 0xFFFFFFF0ull) cap = 0xFFFFFFF0ull;
    b.rccap = (uint32_t)cap;
  }
  DA(b.rcpool, b.rccap);
  b.long_cap = b.rccap / 8 + 1024;
  DA(b.long_list, b.long_cap);
  b.strip_cap = b.rccap / 8 + 1024;
  DA(b.strip_list, b.strip_cap);
  { int G, C; b.tile_qmax = sw_full_geometry(max_read_len, &G, &C) ? 0u : (uint32_t)(G * C); }
  b.rescap = (uint64_t)max_batch_reads * 8 + 4096;
  DA(b.respool, b.rescap);
  b.dstrcap = b.rescap * (uint64_t)(m->qmax / 4 + 48);
  if (b.dstrcap > 0xFFFFFFF0ull) b.dstrcap = 0xFFFFFFF0ull;    // smaltgpu_result.stroffs is 32 bits wide: a batch that needs more reports SMALTGPU_ECAP and is re-mapped in parts
  DA(b.dstrpool, b.dstrcap);
  DA(b.align_retry, max_batch_reads);
  DA(b.cands_retry, max_batch_reads);
  DA(m->d_counters, smaltgpu_mapper::CT_BYTES);
  // S3 as a kernel of its own (k_hits) for mappers of short reads: sorted hit keys of every strand in one pool, 8 bytes per hit.
  // SMALTGPU_HITS_SPLIT=0 keeps S3 inside k_cands; SMALTGPU_HITS_PER_READ sizes the pool (a batch that overflows it is re-mapped in
  // smaller batches like any other pool overflow); SMALTGPU_HITS_WINDOW the keys per window of the kernel.
  {
    const char *e = getenv("SMALTGPU_HITS_SPLIT");
    const bool split = (!e || atoi(e) != 0) && m->qmax <= 256;
    if (split && !rv) {
      const char *hp = getenv("SMALTGPU_HITS_PER_READ");
      uint64_t per = hp ? strtoull(hp, nullptr, 10) : 6144;
      if (per < 64) per = 64;
      uint64_t cap = (uint64_t)max_batch_reads * per;
      if (cap < (1ull << 20)) cap = 1ull << 20;                 // one read alone always fits (<= 2 x 4 x its hit-list allocation)
      b.hitpool_cap = cap;
      DA(b.hitpool, cap);
      DA(b.hitrun, 2 * (size_t)max_batch_reads);
      m->b_hitrun = b.hitrun;
      const char *hw = getenv("SMALTGPU_HITS_WINDOW");
      m->hits_W = hw ? (uint32_t)atoi(hw) : 1024u;       // measured (1 M reads of the bench): 512: 195 ms, 768: 124 ms, 1024: 112 ms
      if (m->hits_W < 256) m->hits_W = 256;
      if (m->hits_W > 1024) m->hits_W = 1024;
      m->hits_wg = 256 * 24;                                     // persistent workgroups: more than can be resident at the smallest LDS block
    }
  }
  if (!rv) {
    typedef smaltgpu_mapper M;
    b.rc_count = (uint32_t *)(m->d_counters + M::CT_RC);
    b.res_count = (unsigned long long *)(m->d_counters + M::CT_RES);
    b.dstr_count = (unsigned long long *)(m->d_counters + M::CT_DSTR);
    b.err_flag = (int32_t *)(m->d_counters + M::CT_ERR);
    b.next_item = (uint32_t *)(m->d_counters + M::CT_NEXT);           // cursor i at next_item + i * NEXT_ITEM_STRIDE
    b.align_retry_n = (uint32_t *)(m->d_counters + M::CT_ALIGN_RETRY);
    b.cands_retry_n = (uint32_t *)(m->d_counters + M::CT_CANDS_RETRY);
    b.work = (unsigned long long *)(m->d_counters + M::CT_WORK);
    b.hit_count = (unsigned long long *)(m->d_counters + M::CT_HITS);
    b.hits_cursor = (uint32_t *)(m->d_counters + M::CT_HITS_CURSOR);
    b.strip_cursor = (uint32_t *)(m->d_counters + M::CT_STRIP_CURSOR);
  }
  // scratch geometry -------------------------------------------------------------------
  m->seed_bytes = (seed_scratch_bytes(m->qmax, d.s) + 255) & ~(size_t)255;
  m->seed_slots = 8192;
  if (m->seed_bytes > 48 * 1024) DA(m->seed_scr, m->seed_bytes * m->seed_slots);
  {
    // hit list capacity per strand (hashhit.c:1262-1288) for the longest read
    double t = (double)max_read_len * log((double)(max_read_len > 1 ? max_read_len : 2)) * HITLST_LOGQLEN_FACT;
    uint64_t alloc = HITLST_BLKSZ;
    if (t > (double)alloc) alloc = (((uint64_t)t + HITLST_BLKSZ - 1) / HITLST_BLKSZ) * HITLST_BLKSZ;
    memset(&m->cg, 0, sizeof(m->cg));
    // A strand gathers at most `alloc` hits per reference sequence (hashhit.c:1497); reads that take
    // the allocation-boundary protocol can therefore exceed `alloc` over all sequences: 4x headroom.
    m->cg.hcap_strand = (uint32_t)std::min<uint64_t>(next_pow2(4 * alloc), 1ull << 30);
    m->cg.hcap = 2 * m->cg.hcap_strand;
    m->cg.ngrp = d.nseq < 512 ? (uint32_t)d.nseq : 1u;  // both modes fit: concatenated mode uses group 0
    m->cg.segcap = m->cg.hcap / 2;
    m->cg.candcap = m->cg.hcap;                         // every hit can be a candidate of its own (mincover = k)
    { const char *e = getenv("SMALTGPU_CANDS_WINDOW"); m->cg.window = e ? (uint32_t)atoi(e) : 0; }   // test hook (tests/test_gpu_large.py)
    m->cg.tab = m->qmax + 8 < (uint32_t)CANDS_TAB ? m->qmax + 8 : (uint32_t)CANDS_TAB;    // a read has fewer seeds than bases
    {
      // Hits of the LDS working set: as many as leave eight workgroups per CU (160 KB / 8 = 20 KB each, less a margin for
      // the allocation granularity) beside the per-list tables and the LDS copy of the sequence table -- more hits per
      // window mean fewer windows, fewer resident workgroups cost more (measured: 704-712 best, 736 = seven workgroups
      // 10 % slower for 150-base reads and 24 sequences).  SMALTGPU_CANDS_LDS_HITS overrides.
      const size_t seqb = (d.nseq > 0 && d.nseq < 512) ? (((size_t)d.nseq + 1) * 4 + 15) & ~(size_t)15 : 0;
      const size_t fixed = 64 /* guard */ + seqb + (size_t)5 * m->cg.tab * 4 + 64 + 15 + 256 /* margin */;
      uint32_t w = fixed + 23 * 256 < 20480 ? (uint32_t)((20480 - fixed) / 23) & ~7u : 256u;
      if (w < (uint32_t)CANDS_LDS_HITS / 2) w = CANDS_LDS_HITS;            // many sequences: the table does not leave room, accept fewer workgroups
      const char *e = getenv("SMALTGPU_CANDS_LDS_HITS");
      m->cg.lds_hits = e ? (uint32_t)atoi(e) : w;
    }
    if (m->cg.lds_hits < 256) m->cg.lds_hits = 256;
    if (m->cg.lds_hits > 2048) m->cg.lds_hits = 2048;
    m->cg.slot_bytes = m->cand_bytes = cand_slot_bytes(m->cg, m->qmax, d.s);
    memset(&m->cg2, 0, sizeof(m->cg2));
    // Two passes.  The slots of the first pass hold 4 x the reference's hit-list allocation per strand (long reads, whose
    // slots would be gigabytes that way: 24 hits per read base and strand); a read that overflows its slot is deferred
    // (SMG_ERR_RETRY) to a second launch over a few slots that hold what the reference's protocol can gather at most: `alloc`
    // hits per strand and reference sequence (hashhit.c:1497; 134 k hits on one strand of a 150-base read against 24
    // sequences were seen at 1 in 2 M reads of the bench's repeat-rich reference), every hit a candidate of its own.
    auto geom_for = [&](uint64_t hs) { CandGeom g = m->cg; g.hcap_strand = (uint32_t)hs; g.hcap = (uint32_t)(2 * hs); g.segcap = (uint32_t)hs; g.candcap = (uint32_t)(2 * hs);
                                       g.slot_bytes = cand_slot_bytes(g, m->qmax, d.s); return g; };
    const CandGeom full = m->cg;
    if (m->cand_bytes > (64ull << 20)) {
      uint64_t hs = next_pow2((uint64_t)24 * m->qmax);
      if (const char *e = getenv("SMALTGPU_CANDS_HCAP")) { const long v = atol(e); if (v >= 1024) hs = next_pow2((uint64_t)v); }   // test hook
      if (hs < m->cg.hcap_strand) { m->cg = geom_for(hs); m->cand_bytes = m->cg.slot_bytes; }
    } else if (const char *e = getenv("SMALTGPU_CANDS_HCAP")) {
      const long v = atol(e);
      if (v >= 1024 && next_pow2((uint64_t)v) < m->cg.hcap_strand) { m->cg = geom_for(next_pow2((uint64_t)v)); m->cand_bytes = m->cg.slot_bytes; }
    }
    {
      uint64_t hw = next_pow2((uint64_t)m->cg.ngrp * alloc);
      if (hw < full.hcap_strand) hw = full.hcap_strand;
      // rmapPair's restricted rounds keep one hit list per search interval, not per sequence (up to IV_MAX lists of up to `alloc`
      // hits; the k = 5 index of the last round finds a hit per read base in every interval of a repeat): room for 2 M hits
      if (hw < (2ull << 20)) hw = 2ull << 20;
      while (hw > full.hcap_strand && (hw > (1ull << 30) || geom_for(hw).slot_bytes > (4ull << 30))) hw >>= 1;      // at most 4 GB per slot
      if (hw > m->cg.hcap_strand) {
        m->cg2 = geom_for(hw); m->cg2.pass = 2; m->cand_bytes2 = m->cg2.slot_bytes;
        m->cg.pass = 1;
        uint64_t n2 = (opt_budget ? ((uint64_t)opt_budget << 30) / 4 : (8ull << 30)) / m->cand_bytes2;      // many mappers on one device: a quarter of their slot budget
        if (n2 > 8) n2 = 8;
        if (n2 < 1) n2 = 1;
        if (n2 > max_batch_reads) n2 = max_batch_reads;
        m->cand_slots2 = (uint32_t)n2;
        DA(m->cand_scr2, m->cand_bytes2 * m->cand_slots2);
      }
    }
    uint64_t budget = 64ull << 30;     // of 288 GB: more slots than resident workgroups lets the hardware balance uneven reads
    if (opt_budget) budget = (uint64_t)opt_budget << 30;              // many mappers on one device
    if (const char *e = getenv("SMALTGPU_SLOT_BUDGET_GB")) { const long g = atol(e); if (g > 0) budget = (uint64_t)g << 30; }
    uint64_t slots = budget / m->cand_bytes;
    if (slots > 2048) slots = 2048;                     // LDS admits 4 workgroups per CU: 1024 run at a time
    if (slots < 64) slots = 64;
    if (slots > max_batch_reads) slots = max_batch_reads;
    m->cand_slots = (uint32_t)slots;
    DA(m->cand_scr, m->cand_bytes * m->cand_slots);
  }
  m->sw_rowlen = m->qmax + 8; m->sw_threads = 16384;
  DA(m->sw_rows, (size_t)m->sw_threads * 2 * m->sw_rowlen);
  {
    m->wincap = 4 * m->qmax + 1024;
    m->strip_grid = 5120;                              // workgroups (one wave each) of the strip kernels: five per SIMD (96 registers); 2048: 5.2, 4096: 6.0, 5120: 6.2 TCUPS at 8 kbp
    if (const char *e = getenv("SMALTGPU_STRIP_GRID")) { const long v = atol(e); if (v >= 64 && v <= 16384) m->strip_grid = (uint32_t)v; }   // tuning hook
    if ((uint64_t)m->strip_grid * m->wincap * 18 > (8ull << 30)) m->strip_grid = (uint32_t)((8ull << 30) / ((uint64_t)m->wincap * 18));
    if (!rv) rv = dalloc((uint8_t **)&m->strip_bnd, (size_t)m->strip_grid * 2 * m->wincap * 8);
    DA(m->strip_win, (size_t)m->strip_grid * m->wincap * 2);      // code pairs of the packed strip kernel
    m->dircap = (uint64_t)(m->qmax + 64) * (m->wincap + 8);
    m->rescap_slot = 512; m->dstrcap_slot = 512 * (m->qmax / 4 + 48);
    // second pass: a few slots that hold the alignments of any read (up to MAXIMUM_DEPTH candidates, each of which may split)
    m->rescap_slot2 = 2 * MAXIMUM_DEPTH; m->dstrcap_slot2 = m->rescap_slot2 * (m->qmax / 4 + 48);
    m->dircap2 = m->dircap;
    if (m->dircap > (4ull << 20)) {
      // Long reads: a full direction matrix (read x window) is 10-300 MB, a band rarely needs a tenth of it.  The many
      // slots of the first pass hold a sixteenth; a read whose band does not fit is deferred to a second pass over a
      // few full-size slots.
      m->dircap2 = m->dircap;
      m->dircap = m->dircap / 16 > (4ull << 20) ? m->dircap / 16 : (4ull << 20);
      if (const char *e = getenv("SMALTGPU_ALIGN_DIRCAP")) { const long v = atol(e); if (v >= 4096 && (uint64_t)v < m->dircap2) m->dircap = (uint64_t)v; }   // test hook
    }
    m->align_bytes2 = align_scratch_bytes(m->qmax, m->wincap, m->dircap2, m->rescap_slot2, m->dstrcap_slot2);
    m->align_slots2 = max_batch_reads < 16 ? max_batch_reads : 16;
    DA(m->align_scr2, m->align_bytes2 * m->align_slots2);
    m->align_bytes = align_scratch_bytes(m->qmax, m->wincap, m->dircap, m->rescap_slot, m->dstrcap_slot);
    uint64_t budget = 8ull << 30;
    if (m->align_bytes * 512 > budget) budget = m->align_bytes * 512;      // long reads: direction matrices of 10-300 MB per slot
    if (budget > (48ull << 30)) budget = 48ull << 30;
    if (opt_budget && ((uint64_t)opt_budget << 30) < budget) budget = (uint64_t)opt_budget << 30;
    if (const char *e = getenv("SMALTGPU_SLOT_BUDGET_GB")) { const uint64_t g = (uint64_t)atol(e) << 30; if (g > 0 && g < budget) budget = g; }
    uint64_t slots = budget / m->align_bytes;
    if (slots > 8192) slots = 8192;
    if (slots < 16) slots = 16;
    if (slots > max_batch_reads) slots = max_batch_reads;
    m->align_slots = (uint32_t)slots;
    DA(m->align_scr, m->align_bytes * m->align_slots);
  }
#undef DA
  if (rv) { smaltgpu_mapper_free(m); return rv; }
  *out = m;
  return SMALTGPU_OK;
}

extern "C" void smaltgpu_mapper_free(smaltgpu_mapper *m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  void *ps[] = {m->d_bases, m->d_quals, m->d_codes, m->d_codes_rc, m->d_off, m->d_ids, m->b.hi, m->b.seeds, m->b.qmask, m->b.ch, m->b.ctl,
                m->b.stat, m->b.align_retry, m->b.cands_retry, m->b.hitpool, m->b_hitrun, m->b.rcpool, m->b.long_list, m->b.strip_list, m->strip_bnd, m->strip_win, m->b.respool, m->b.dstrpool, m->d_counters, m->seed_scr, m->cand_scr, m->cand_scr2, m->cand_scr_dbg,
                m->sw_rows, m->align_scr, m->align_scr2};
  for (void *p : ps) if (p) (void)hipFree(p);
  m->h_stat.release(); m->h_res.release(); m->h_dstr.release();
  m->cx_ivoff.release(); m->cx_iv.release(); m->cx_minsw.release(); m->cx_prevmax.release(); m->cx_fineidx.release(); m->cx_finepos.release(); m->cx_fineoff.release(); m->cx_alloclen.release(); m->cx_seedrange.release();
  if (m->ev_fetch) (void)hipEventDestroy(m->ev_fetch);
  for (int i = 0; i <= T_NUM; i++) if (m->ev[i]) (void)hipEventDestroy(m->ev[i]);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
}

extern "C" int smaltgpu_set_debug(smaltgpu_mapper *m, int level) {
  if (!m) return fail(SMALTGPU_EARG, "null mapper");
  m->debug = level;
  return SMALTGPU_OK;
}

static MapPar to_par(const smaltgpu_params *p) {
  MapPar q;
  q.cov_frac = p->min_cover_frac > 0.0 ? p->min_cover_frac : 0.0;
  q.ncut = p->ktuple_maxhit; q.min_cover = p->min_cover; q.min_swatscor = p->min_swatscor; q.below_max = p->min_swatscor_below_max;
  q.min_basq = p->min_basqval; q.target_depth = p->target_depth; q.max_depth = p->max_depth; q.flags = p->rmapflg;
  q.match = p->match; q.mismatch = p->mismatch; q.gap_init = p->gap_init; q.gap_ext = p->gap_ext;
  return q;
}

static int check_par(const smaltgpu_mapper *m, const smaltgpu_params *p) {
  if (p->match < 1 || p->mismatch >= 0 || p->gap_init >= 0 || p->gap_ext >= 0 || p->match > 15 || p->mismatch < -100)
    return fail(SMALTGPU_EARG, "unsupported penalties");
  if (p->min_swatscor < 1) return fail(SMALTGPU_EARG, "min_swatscor must be >= 1 (alignment.c:1569)");
  if ((p->rmapflg & SMALTGPU_FLG_SEQBYSEQ) && m->ix->d.nseq >= 512) return fail(SMALTGPU_EARG, "sequence-by-sequence mode needs < 512 sequences (smalt.c:599)");
  if (p->ktuple_maxhit < 1) return fail(SMALTGPU_EARG, "ktuple_maxhit < 1 is not supported");
  return 0;
}

// the device pipeline over reads already in HBM
// cx: the per-read context of one of rmapPair's rounds, already on the device (upload_ctx), or null
struct CtxDev { const uint32_t *iv_off; const IvRec *iv; const int32_t *min_sw, *prevmax; uint32_t *fine_idx, *fine_pos; const uint32_t *fine_off, *alloc_len, *seed_range; uint32_t raw; };

static int run_pipeline(smaltgpu_mapper *m, const uint8_t *d_bases, const uint8_t *d_quals, const uint64_t *d_off, uint32_t n,
                        const smaltgpu_params *par, const CtxDev *cx = nullptr, bool seed_only = false) {
  Batch &b = m->b;
  const DevIndex &d = m->ix->d;
  MapPar p = to_par(par);
  b.iv_off = cx ? cx->iv_off : nullptr; b.iv = cx ? cx->iv : nullptr; b.min_sw = cx ? cx->min_sw : nullptr; b.prevmax = cx ? cx->prevmax : nullptr;
  b.fine_idx = cx ? cx->fine_idx : nullptr; b.fine_pos = cx ? cx->fine_pos : nullptr; b.fine_off = cx ? cx->fine_off : nullptr;
  b.alloc_len = cx ? cx->alloc_len : nullptr;
  b.seed_range = cx ? cx->seed_range : nullptr;
  b.totals_only = seed_only ? 1u : 0u;
  b.raw_results = cx ? cx->raw : 0u;
  if (b.fine_idx) p.flags |= FLG_NOSHRTINFO;            // initRMAPINFO, not the short form (rmap.c:2024)
  m->last_fine = b.fine_idx != nullptr;
  m->last_par = p; m->last_n = n;
  b.nreads = n; b.codes = m->d_codes; b.codes_rc = m->d_codes_rc; b.qual = d_quals; b.read_off = d_off;
  hipStream_t s = m->stream;
  HIPCHK(hipMemsetAsync(m->d_counters, 0, smaltgpu_mapper::CT_BYTES, s));
  int rv = 0;
  const bool seqbyseq = (p.flags & FLG_SEQBYSEQ) != 0;
  const uint32_t ngrp = seqbyseq ? (uint32_t)d.nseq : 1u;
  uint8_t *cscr = m->cand_scr;
  uint32_t cslots = m->cand_slots;
  int slot_per_read = 0;
  CandGeom cgeom = m->cg;
  const bool two_pass = m->cand_scr2 != nullptr && !m->debug;
  const size_t cbytes_dbg = m->cand_bytes;      // debug batches (dumps read the slot of every read): one pass over first-pass-sized slots
  if (m->debug) {
    if (n > 2048) return fail(SMALTGPU_EARG, "debug batches are limited to 2048 reads");
    if (m->cand_dbg_reads < n) {
      if (m->cand_scr_dbg) (void)hipFree(m->cand_scr_dbg);
      m->cand_scr_dbg = nullptr;
      HIPCHK(hipMalloc((void **)&m->cand_scr_dbg, cbytes_dbg * n));
      m->cand_dbg_reads = n;
    }
    cscr = m->cand_scr_dbg; cslots = n; slot_per_read = 1;
  }
  HIPCHK(hipEventRecord(m->ev[T_ENCODE], s));
  rv = launch_encode(s, d_bases, d_off, n, m->d_codes, m->d_codes_rc);
  if (!rv && b.fine_idx) rv = launch_fine_index(s, b, d);
  HIPCHK(hipEventRecord(m->ev[T_SEED], s));
  if (!rv) rv = launch_seed(s, b, d, p, m->seed_scr, m->seed_bytes, m->seed_slots);
  HIPCHK(hipEventRecord(m->ev[T_HITS], s));
  // S3 ahead of the candidate stage (plain calls of short-read mappers; restricted calls keep it inside k_cands)
  const bool split_hits = m->hits_W && m->b_hitrun && !b.iv_off && !seed_only;
  b.hitrun = split_hits ? m->b_hitrun : nullptr;
  if (!rv && split_hits) rv = launch_hits(s, b, d, p, m->hits_W, m->cg.tab, m->hits_wg);
  HIPCHK(hipEventRecord(m->ev[T_CANDS], s));
  if (seed_only) {
    for (int i = T_CANDS + 1; i <= T_NUM; i++) HIPCHK(hipEventRecord(m->ev[i], s));
    if (rv) return fail(SMALTGPU_ENODEV, "kernel launch failed: %s", hipGetErrorString((hipError_t)rv));
    return SMALTGPU_OK;
  }
  { CandGeom g = cgeom; g.ngrp = ngrp; g.debug = slot_per_read; if (!two_pass) g.pass = 0; if (!rv) rv = launch_cands(s, b, d, p, cscr, cslots, g); }
  if (two_pass) { CandGeom g = m->cg2; g.ngrp = ngrp; g.debug = 0; g.pass = 2; if (!rv) rv = launch_cands(s, b, d, p, m->cand_scr2, m->cand_slots2, g); }
  HIPCHK(hipEventRecord(m->ev[T_SW_FULL], s));
  if (!rv) rv = launch_sw_full(s, b, d, p, m->max_len, b.rccap, 8192);
  HIPCHK(hipEventRecord(m->ev[T_SW_SCALAR], s));
  if (!rv) rv = launch_sw_strip(s, b, d, p, m->strip_bnd, m->strip_win, m->wincap, m->strip_grid);
  if (!rv) rv = launch_sw_scalar(s, b, d, p, m->sw_rows, m->sw_rowlen, m->sw_threads, m->max_len);
  HIPCHK(hipEventRecord(m->ev[T_REPLAY], s));
  if (!rv) rv = launch_replay(s, b, d, p);
  HIPCHK(hipEventRecord(m->ev[T_ALIGN], s));
  if (!rv) rv = launch_align(s, b, d, p, m->align_scr, m->align_bytes, m->align_slots, m->wincap, m->dircap, m->rescap_slot, m->dstrcap_slot, 1);
  if (!rv) rv = launch_align(s, b, d, p, m->align_scr2, m->align_bytes2, m->align_slots2, m->wincap, m->dircap2, m->rescap_slot2, m->dstrcap_slot2, 2);
  HIPCHK(hipEventRecord(m->ev[T_NUM], s));
  if (rv) return fail(SMALTGPU_ENODEV, "kernel launch failed: %s", hipGetErrorString((hipError_t)rv));
  return SMALTGPU_OK;
}

extern "C" int smaltgpu_synchronize(smaltgpu_mapper *m) {
  if (!m) return fail(SMALTGPU_EARG, "null mapper");
  HIPCHK(hipStreamSynchronize(m->stream));
  return SMALTGPU_OK;
}

extern "C" int smaltgpu_map_batch_device(smaltgpu_mapper *m, const uint8_t *d_bases, const uint8_t *d_quals, const uint64_t *d_read_off,
                                          uint32_t nreads, uint64_t total_bases, const smaltgpu_params *par) {
  if (!m || !d_bases || !d_read_off || !par) return fail(SMALTGPU_EARG, "null argument");
  if (nreads > m->max_reads || total_bases > m->max_bases) return fail(SMALTGPU_EARG, "batch exceeds the mapper's capacity");
  int rv = check_par(m, par);
  if (rv) return rv;
  HIPCHK(hipSetDevice(m->device));
  m->have_host_off = false;
  return run_pipeline(m, d_bases, d_quals, d_read_off, nreads, par);
}

// Results in two steps so that a caller can keep the device busy: _begin waits for the batch, reads the pool counters and
// enqueues the copies to pinned host memory; the next batch may be launched right behind them (same stream: the copies
// finish first); _end waits for the copies and puts the results in read order.
extern "C" int smaltgpu_fetch_begin(smaltgpu_mapper *m) {
  if (!m) return fail(SMALTGPU_EARG, "null argument");
  HIPCHK(hipSetDevice(m->device));
  const uint32_t n = m->last_n;
  uint8_t ctr[smaltgpu_mapper::CT_BYTES];
  if (!m->ev_fetch) HIPCHK(hipEventCreate(&m->ev_fetch));
  if (m->h_stat.ensure(n ? n : 1)) return fail(SMALTGPU_ENOMEM, "pinned host memory");
  HIPCHK(hipMemcpyAsync(ctr, m->d_counters, smaltgpu_mapper::CT_BYTES, hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  const uint32_t rc_count = *(uint32_t *)(ctr + smaltgpu_mapper::CT_RC);
  const uint64_t nres = *(unsigned long long *)(ctr + smaltgpu_mapper::CT_RES), ndstr = *(unsigned long long *)(ctr + smaltgpu_mapper::CT_DSTR);
  memcpy(m->work, ctr + smaltgpu_mapper::CT_WORK, sizeof(m->work));
  for (int i = 0; i < T_NUM; i++) { float f = 0; (void)hipEventElapsedTime(&f, m->ev[i], m->ev[i + 1]); m->ms[i] = f; }
  m->fetch_open = false;
  // A pool that overflowed leaves SMALTGPU_ECAP in the stat of every read that did not fit (their results are dropped on the
  // device); all other reads are complete.  smaltgpu_fetch_end reports the code, smaltgpu_map_batch re-maps those reads.
  m->pool_overflow = rc_count > m->b.rccap || nres > m->b.rescap || ndstr > m->b.dstrcap;
  const uint64_t nres_c = nres < m->b.rescap ? nres : m->b.rescap, ndstr_c = ndstr < m->b.dstrcap ? ndstr : m->b.dstrcap;
  if (m->h_res.ensure(nres_c ? nres_c : 1) || m->h_dstr.ensure(ndstr_c ? ndstr_c : 1)) return fail(SMALTGPU_ENOMEM, "pinned host memory");
  if (n) HIPCHK(hipMemcpyAsync(m->h_stat.data(), m->b.stat, (size_t)n * sizeof(ReadStat), hipMemcpyDeviceToHost, m->stream));
  if (nres_c) HIPCHK(hipMemcpyAsync(m->h_res.data(), m->b.respool, nres_c * sizeof(Result), hipMemcpyDeviceToHost, m->stream));
  if (ndstr_c) HIPCHK(hipMemcpyAsync(m->h_dstr.data(), m->b.dstrpool, ndstr_c, hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipEventRecord(m->ev_fetch, m->stream));
  m->fetch_n = n; m->fetch_nres = nres_c; m->fetch_open = true;
  return SMALTGPU_OK;
}

extern "C" int smaltgpu_fetch_end(smaltgpu_mapper *m, smaltgpu_batch_out *out) {
  if (!m || !out) return fail(SMALTGPU_EARG, "null argument");
  if (!m->fetch_open) return fail(SMALTGPU_EARG, "smaltgpu_fetch_begin must precede smaltgpu_fetch_end");
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipEventSynchronize(m->ev_fetch));
  m->fetch_open = false;
  const uint32_t n = m->fetch_n;
  const uint64_t nres = m->fetch_nres;
  // per-read order: results of read i are contiguous in the pool but reads finish in any order;
  // re-pack in read order so that res_off is monotone
  m->h_res_off.assign((size_t)n + 1, 0);
  m->o_res.resize(nres ? nres : 1);
  m->o_stat.resize(n ? n : 1);
  uint64_t w = 0;
  int first_err = 0;
  uint32_t first_err_read = 0, nerr = 0;
  for (uint32_t i = 0; i < n; i++) {                       // where each read's results go (reads finish in any order on the device)
    const ReadStat &st = m->h_stat[i];
    m->h_res_off[i] = w;
    w += st.nres;
    if (st.err) { if (!first_err) { first_err = st.err; first_err_read = i; } nerr++; }
  }
  m->h_res_off[n] = w;
  auto copy_range = [m](uint32_t lo, uint32_t hi) {
    for (uint32_t i = lo; i < hi; i++) {
      const ReadStat &st = m->h_stat[i];
      smaltgpu_readstat &os = m->o_stat[i];
      os.swatscor_max = st.swmax; os.swatscor_2ndmax = st.sw2nd; os.n_ali_done = st.nseg; os.n_ali_tot = st.nseg_tot;
      os.n_hits_used = st.nhit; os.n_hits_tot = st.nhit_tot; os.errcode = st.err; os.nres = st.nres; os.max1scor = st.max1; os.errsite = st.err_site;
      uint64_t at = m->h_res_off[i];
      for (uint32_t j = 0; j < st.nres; j++) {
        const Result &r = m->h_res[st.res_off + j];
        smaltgpu_result &o = m->o_res[at++];
        o.swatscor = r.swatscor; o.q_start = r.q_start; o.q_end = r.q_end; o.s_start = r.s_start; o.s_end = r.s_end;
        o.sidx = r.sidx; o.reverse = (r.reverse ? SMALTGPU_RES_REVERSE : 0u) | (r.pad ? SMALTGPU_RES_CANDFIRST : 0u);
        o.stroffs = (uint32_t)(st.dstr_off + r.stroffs); o.strlen = r.strlen;
      }
    }
  };
  int nt = m->host_threads;
  if ((uint32_t)nt > n / 8192 + 1) nt = (int)(n / 8192 + 1);
  if (nt <= 1) copy_range(0, n);
  else {
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(copy_range, (uint32_t)((uint64_t)n * t / nt), (uint32_t)((uint64_t)n * (t + 1) / nt));
    copy_range(0, (uint32_t)((uint64_t)n / nt));
    for (std::thread &x : th) x.join();
  }
  out->nreads = n; out->res_off = m->h_res_off.data(); out->res = m->o_res.data(); out->diffstr = m->h_dstr.data(); out->stat = m->o_stat.data();
  if (first_err) return fail(first_err, "%u of %u reads hit a device-side limit (-5%s), an assertion (-6) or the reference's own score check (-8); first: read %u code %d (see stat[].errcode)", nerr, n,
                             m->pool_overflow ? ": a batch-wide work pool overflowed" : "", first_err_read, first_err);
  return SMALTGPU_OK;
}

extern "C" int smaltgpu_fetch_results(smaltgpu_mapper *m, smaltgpu_batch_out *out) {
  if (!m || !out) return fail(SMALTGPU_EARG, "null argument");
  const int rv = smaltgpu_fetch_begin(m);
  if (rv) return rv;
  return smaltgpu_fetch_end(m, out);
}

// the context of a round to the device; fills cd
static int upload_ctx(smaltgpu_mapper *m, const smaltgpu_callctx *ctx, uint32_t n, CtxDev *cd) {
  memset(cd, 0, sizeof(*cd));
  hipStream_t s = m->stream;
  const DevIndex &d = m->ix->d;
  if (ctx->fine_index && !ctx->iv_off) return fail(SMALTGPU_EARG, "fine_index needs intervals");
  if (ctx->iv_off) {
    if (!ctx->iv && ctx->iv_off[n] > ctx->iv_off[0]) return fail(SMALTGPU_EARG, "null interval array");
    const uint64_t i0 = ctx->iv_off[0], niv = ctx->iv_off[n] - i0;
    if (niv > 0xFFFFFFF0ull) return fail(SMALTGPU_EARG, "too many intervals");
    m->h_ivoff.resize((size_t)n + 1);
    m->h_fineoff.assign((size_t)n + 1, 0);
    for (uint32_t i = 0; i <= n; i++) m->h_ivoff[i] = (uint32_t)(ctx->iv_off[i] - i0);
    for (uint32_t i = 0; i < n; i++) {
      uint64_t npos = 0;
      // (a read with more than IV_MAX = 2048 intervals is not an error of the batch: the candidate stage gives that read
      //  SMALTGPU_ECAP in its stat and every other read is mapped)
      for (uint64_t v = ctx->iv_off[i]; v < ctx->iv_off[i + 1]; v++) {
        const smaltgpu_interval &iv = ctx->iv[v];
        if (iv.sidx < 0 || iv.sidx >= d.nseq || iv.hi < iv.lo || m->ix->sop[(size_t)iv.sidx] + iv.hi >= m->ix->sop[(size_t)iv.sidx + 1])
          return fail(SMALTGPU_EARG, "interval %llu of read %u lies outside its sequence", (unsigned long long)(v - ctx->iv_off[i]), i);
        if (iv.hi - iv.lo + 1 >= (uint32_t)FINE_K) npos += iv.hi - iv.lo + 1 - FINE_K + 1;
      }
      if (m->h_fineoff[i] + npos > 0xFFFFFFF0ull) return fail(SMALTGPU_EARG, "interval windows too large");
      m->h_fineoff[i + 1] = (uint32_t)(m->h_fineoff[i] + npos);
    }
    if (m->cx_ivoff.ensure(((size_t)n + 1) * 4) || m->cx_iv.ensure((size_t)(niv ? niv : 1) * sizeof(IvRec))) return fail(SMALTGPU_ENOMEM, "device memory");
    HIPCHK(hipMemcpyAsync(m->cx_ivoff.p, m->h_ivoff.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice, s));
    static_assert(sizeof(IvRec) == sizeof(smaltgpu_interval), "interval layout");
    if (niv) HIPCHK(hipMemcpyAsync(m->cx_iv.p, ctx->iv + i0, (size_t)niv * sizeof(IvRec), hipMemcpyHostToDevice, s));
    cd->iv_off = (const uint32_t *)m->cx_ivoff.p; cd->iv = (const IvRec *)m->cx_iv.p;
    if (ctx->fine_index) {
      if (d.totlen + 1 > 0xFFFFFFFFull) return fail(SMALTGPU_EARG, "on-the-fly index: reference longer than 2^32 - 1 bases (rmap.c:1537-1541 would raise the stride)");
      if (m->cx_fineidx.ensure((size_t)n * FINE_IDX_STRIDE * 4) || m->cx_finepos.ensure(((size_t)m->h_fineoff[n] + 1) * 4) || m->cx_fineoff.ensure(((size_t)n + 1) * 4))
        return fail(SMALTGPU_ENOMEM, "device memory");
      HIPCHK(hipMemcpyAsync(m->cx_fineoff.p, m->h_fineoff.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice, s));
      cd->fine_idx = (uint32_t *)m->cx_fineidx.p; cd->fine_pos = (uint32_t *)m->cx_finepos.p; cd->fine_off = (const uint32_t *)m->cx_fineoff.p;
    }
  }
  if (ctx->min_swatscor) {
    // 0 is legal here: rmapPair passes the first mate's second-best score, which is 0 when there is none (rmap.c:2003-2031);
    // the threshold block of mapSingleRead then raises it to the best first-pass score (rmap.c:1385-1393)
    for (uint32_t i = 0; i < n; i++) if (ctx->min_swatscor[i] < 0) return fail(SMALTGPU_EARG, "negative min_swatscor");
    if (m->cx_minsw.ensure((size_t)(n ? n : 1) * 4)) return fail(SMALTGPU_ENOMEM, "device memory");
    if (n) HIPCHK(hipMemcpyAsync(m->cx_minsw.p, ctx->min_swatscor, (size_t)n * 4, hipMemcpyHostToDevice, s));
    cd->min_sw = (const int32_t *)m->cx_minsw.p;
  }
  if (ctx->prev_max) {
    if (m->cx_prevmax.ensure((size_t)(n ? n : 1) * 8)) return fail(SMALTGPU_ENOMEM, "device memory");
    if (n) HIPCHK(hipMemcpyAsync(m->cx_prevmax.p, ctx->prev_max, (size_t)n * 8, hipMemcpyHostToDevice, s));
    cd->prevmax = (const int32_t *)m->cx_prevmax.p;
  }
  cd->raw = ctx->raw_alignments ? 1u : 0u;
  if (ctx->hitlist_len) {
    for (uint32_t i = 0; i < n; i++) if (ctx->hitlist_len[i] > m->max_len) return fail(SMALTGPU_EARG, "hitlist_len of read %u exceeds the mapper's max_read_len", i);
    if (m->cx_alloclen.ensure((size_t)(n ? n : 1) * 4)) return fail(SMALTGPU_ENOMEM, "device memory");
    if (n) HIPCHK(hipMemcpyAsync(m->cx_alloclen.p, ctx->hitlist_len, (size_t)n * 4, hipMemcpyHostToDevice, s));
    cd->alloc_len = (const uint32_t *)m->cx_alloclen.p;
  }
  if (ctx->seed_range) {
    if (m->cx_seedrange.ensure((size_t)(n ? n : 1) * 8)) return fail(SMALTGPU_ENOMEM, "device memory");
    if (n) HIPCHK(hipMemcpyAsync(m->cx_seedrange.p, ctx->seed_range, (size_t)n * 8, hipMemcpyHostToDevice, s));
    cd->seed_range = (const uint32_t *)m->cx_seedrange.p;
  }
  return SMALTGPU_OK;
}

// one device batch over reads in host memory: bases/quals + read_off[0..n]; ctx (may be null) is indexed like the reads
static int map_range(smaltgpu_mapper *m, const uint8_t *bases, const uint8_t *quals, const uint64_t *read_off, uint32_t nreads,
                     const smaltgpu_params *par, smaltgpu_batch_out *out, const smaltgpu_callctx *ctx = nullptr, bool seed_only = false) {
  const uint64_t total = read_off[nreads] - read_off[0];
  m->h_off.resize((size_t)nreads + 1);
  for (uint32_t i = 0; i <= nreads; i++) {
    m->h_off[i] = read_off[i] - read_off[0];
    if (i && m->h_off[i] - m->h_off[i - 1] > m->max_len) return fail(SMALTGPU_EARG, "read %u is longer than the mapper's max_read_len", i - 1);
  }
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipMemcpyAsync(m->d_bases, bases + read_off[0], total, hipMemcpyHostToDevice, m->stream));
  if (quals) HIPCHK(hipMemcpyAsync(m->d_quals, quals + read_off[0], total, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->d_off, m->h_off.data(), ((size_t)nreads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, m->stream));
  CtxDev cd;
  if (ctx) { const int cr = upload_ctx(m, ctx, nreads, &cd); if (cr) return cr; }
  const int rv = run_pipeline(m, m->d_bases, quals ? m->d_quals : nullptr, m->d_off, nreads, par, ctx ? &cd : nullptr, seed_only);
  m->have_host_off = true;
  if (rv || seed_only) return rv;
  return smaltgpu_fetch_results(m, out);
}

// The work pools of a mapper (ranked candidates, results, DiffStr bytes) are sized for the average read of a large batch;
// the reference's buffers grow without bound instead (array.c).  When a batch overflows a pool, the reads that did not
// fit (stat[].errcode == SMALTGPU_ECAP; every other read is complete) are mapped again in smaller batches -- all of them
// together, in halves when none of them fitted -- down to a single read, for which the pools hold the reference's
// maximum of candidates.  Results are per read, so the outcome equals that of one unlimited batch.
static int remap_overflowed(smaltgpu_mapper *m, const uint8_t *bases, const uint8_t *quals, const uint64_t *read_off, uint32_t nreads,
                            const smaltgpu_params *par, smaltgpu_batch_out *out, const smaltgpu_callctx *ctx = nullptr) {
  struct Part { std::vector<smaltgpu_result> res; std::vector<uint8_t> dstr; };
  std::vector<smaltgpu_readstat> stat(out->stat, out->stat + nreads);
  std::vector<uint64_t> off0(out->res_off, out->res_off + nreads + 1);
  std::vector<smaltgpu_result> res0(out->res, out->res + off0[nreads]);
  size_t nd0 = 0;
  for (const smaltgpu_result &r : res0) if ((size_t)r.stroffs + r.strlen > nd0) nd0 = (size_t)r.stroffs + r.strlen;
  std::vector<uint8_t> dstr0(out->diffstr, out->diffstr + nd0);
  std::vector<Part> part(nreads);
  std::vector<std::vector<uint32_t>> todo(1);
  for (uint32_t i = 0; i < nreads; i++) if (stat[i].errcode == SMALTGPU_ECAP) todo[0].push_back(i);
  std::vector<uint8_t> sb, sq;
  std::vector<uint64_t> so, civo;
  std::vector<smaltgpu_interval> civ;
  std::vector<int32_t> cms, cpm;
  std::vector<uint32_t> chl, csr;
  uint32_t npermanent = 0, nbatches = 0;
  while (!todo.empty()) {
    std::vector<uint32_t> L;
    L.swap(todo.back());
    todo.pop_back();
    if (L.empty()) continue;
    sb.clear(); sq.clear(); so.assign(1, 0);
    for (uint32_t i : L) {
      sb.insert(sb.end(), bases + read_off[i], bases + read_off[i + 1]);
      if (quals) sq.insert(sq.end(), quals + read_off[i], quals + read_off[i + 1]);
      so.push_back(sb.size());
    }
    if (sb.empty()) sb.push_back(0);
    smaltgpu_callctx sub;
    if (ctx) {                                          // the same reads' slice of the round's context
      sub = *ctx;
      civo.assign(1, 0); civ.clear(); cms.clear(); cpm.clear(); chl.clear(); csr.clear();
      for (uint32_t i : L) {
        if (ctx->iv_off) { civ.insert(civ.end(), ctx->iv + ctx->iv_off[i], ctx->iv + ctx->iv_off[i + 1]); civo.push_back(civ.size()); }
        if (ctx->min_swatscor) cms.push_back(ctx->min_swatscor[i]);
        if (ctx->prev_max) { cpm.push_back(ctx->prev_max[2 * (size_t)i]); cpm.push_back(ctx->prev_max[2 * (size_t)i + 1]); }
        if (ctx->hitlist_len) chl.push_back(ctx->hitlist_len[i]);
        if (ctx->seed_range) { csr.push_back(ctx->seed_range[2 * (size_t)i]); csr.push_back(ctx->seed_range[2 * (size_t)i + 1]); }
      }
      if (civ.empty()) civ.resize(1);
      if (ctx->iv_off) { sub.iv_off = civo.data(); sub.iv = civ.data(); }
      if (ctx->min_swatscor) sub.min_swatscor = cms.data();
      if (ctx->prev_max) sub.prev_max = cpm.data();
      if (ctx->hitlist_len) sub.hitlist_len = chl.data();
      if (ctx->seed_range) sub.seed_range = csr.data();
    }
    smaltgpu_batch_out o;
    const int rv = map_range(m, sb.data(), quals ? sq.data() : nullptr, so.data(), (uint32_t)L.size(), par, &o, ctx ? &sub : nullptr);
    nbatches++;
    if (rv && !(SMALTGPU_IS_READ_ERROR(rv) && o.nreads == L.size())) return rv;
    std::vector<uint32_t> failed;
    for (size_t t = 0; t < L.size(); t++) {
      const uint32_t i = L[t];
      if (o.stat[t].errcode == SMALTGPU_ECAP) { failed.push_back(i); continue; }
      stat[i] = o.stat[t];
      Part &p = part[i];
      p.res.assign(o.res + o.res_off[t], o.res + o.res_off[t + 1]);
      for (smaltgpu_result &r : p.res) {
        const uint32_t at = (uint32_t)p.dstr.size();
        p.dstr.insert(p.dstr.end(), o.diffstr + r.stroffs, o.diffstr + r.stroffs + r.strlen);
        r.stroffs = at;
      }
    }
    if (failed.empty()) continue;
    if (failed.size() < L.size()) todo.push_back(failed);
    else if (L.size() == 1) npermanent++;                       // alone and still over a limit: the read keeps its error code
    else {
      todo.emplace_back(L.begin() + (ptrdiff_t)(L.size() / 2), L.end());
      todo.emplace_back(L.begin(), L.begin() + (ptrdiff_t)(L.size() / 2));
    }
  }
  // assemble in read order
  m->fin_res.clear(); m->fin_dstr.clear(); m->fin_off.assign((size_t)nreads + 1, 0); m->fin_stat = stat;
  int first_err = 0;
  for (uint32_t i = 0; i < nreads; i++) {
    m->fin_off[i] = m->fin_res.size();
    const smaltgpu_result *rp = part[i].res.empty() ? res0.data() + off0[i] : part[i].res.data();
    const size_t nr = part[i].res.empty() ? (size_t)(off0[i + 1] - off0[i]) : part[i].res.size();
    const uint8_t *dp = part[i].res.empty() ? dstr0.data() : part[i].dstr.data();
    if (stat[i].errcode && !first_err) first_err = stat[i].errcode;
    for (size_t j = 0; j < nr; j++) {
      smaltgpu_result r = rp[j];
      if (m->fin_dstr.size() + r.strlen > 0xFFFFFFF0ull) return fail(SMALTGPU_ECAP, "alignment strings of the batch exceed 4 GB: use smaller batches");
      const uint32_t at = (uint32_t)m->fin_dstr.size();
      m->fin_dstr.insert(m->fin_dstr.end(), dp + r.stroffs, dp + r.stroffs + r.strlen);
      r.stroffs = at;
      m->fin_res.push_back(r);
    }
  }
  m->fin_off[nreads] = m->fin_res.size();
  if (m->fin_res.empty()) m->fin_res.resize(1);
  if (m->fin_dstr.empty()) m->fin_dstr.resize(1);
  out->nreads = nreads; out->res_off = m->fin_off.data(); out->res = m->fin_res.data(); out->diffstr = m->fin_dstr.data(); out->stat = m->fin_stat.data();
  m->remap_batches += nbatches;
  if (first_err) return fail(first_err, "%u of %u reads exceed a device-side limit even alone or failed an assertion (see stat[].errcode)", npermanent, nreads);
  return SMALTGPU_OK;
}

extern "C" int smaltgpu_map_batch(smaltgpu_mapper *m, const uint8_t *bases, const uint8_t *quals, const uint64_t *read_off, uint32_t nreads,
                                   const smaltgpu_params *par, smaltgpu_batch_out *out) {
  return smaltgpu_map_batch_ctx(m, bases, quals, read_off, nreads, par, nullptr, out);
}

extern "C" const smaltgpu_index *smaltgpu_mapper_index(const smaltgpu_mapper *m) { return m ? m->ix : nullptr; }

extern "C" int smaltgpu_mapper_capacity(const smaltgpu_mapper *m, uint32_t *max_batch_reads, uint32_t *max_read_len, uint64_t *max_bases) {
  if (!m) return fail(SMALTGPU_EARG, "null mapper");
  if (max_batch_reads) *max_batch_reads = m->max_reads;
  if (max_read_len) *max_read_len = m->max_len;
  if (max_bases) *max_bases = m->max_bases;
  return SMALTGPU_OK;
}

extern "C" int smaltgpu_hit_totals(smaltgpu_mapper *m, const uint8_t *bases, const uint8_t *quals, const uint64_t *read_off, uint32_t nreads,
                                    const smaltgpu_params *par, uint32_t *nhits) {
  if (!m || !bases || !read_off || !par || !nhits) return fail(SMALTGPU_EARG, "null argument");
  if (nreads > m->max_reads) return fail(SMALTGPU_EARG, "batch of %u reads exceeds the mapper's capacity %u", nreads, m->max_reads);
  if (read_off[nreads] - read_off[0] > m->max_bases) return fail(SMALTGPU_EARG, "batch exceeds the mapper's base capacity");
  int rv = check_par(m, par);
  if (rv) return rv;
  rv = map_range(m, bases, quals, read_off, nreads, par, nullptr, nullptr, true);       // encode + seeding (S1, S2) only
  if (rv) return rv;
  m->h_hi.resize(2 * (size_t)nreads + 1);
  if (nreads) HIPCHK(hipMemcpyAsync(m->h_hi.data(), m->b.hi, 2 * (size_t)nreads * sizeof(HitInfoHdr), hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  for (int i = 0; i < T_NUM; i++) { float f = 0; (void)hipEventElapsedTime(&f, m->ev[i], m->ev[i + 1]); m->ms[i] = f; }      // seeding only: the later stages read 0
  memset(m->work, 0, sizeof(m->work));
  for (uint32_t i = 0; i < nreads; i++) nhits[i] = m->h_hi[2 * (size_t)i].nhit_cut + m->h_hi[2 * (size_t)i + 1].nhit_cut;
  return SMALTGPU_OK;
}

extern "C" int smaltgpu_map_batch_ctx(smaltgpu_mapper *m, const uint8_t *bases, const uint8_t *quals, const uint64_t *read_off, uint32_t nreads,
                                       const smaltgpu_params *par, const smaltgpu_callctx *ctx, smaltgpu_batch_out *out) {
  if (!m || !bases || !read_off || !par || !out) return fail(SMALTGPU_EARG, "null argument");
  if (nreads > m->max_reads) return fail(SMALTGPU_EARG, "batch of %u reads exceeds the mapper's capacity %u", nreads, m->max_reads);
  const uint64_t total = read_off[nreads] - read_off[0];
  if (total > m->max_bases) return fail(SMALTGPU_EARG, "batch exceeds the mapper's base capacity");
  int rv = check_par(m, par);
  if (rv) return rv;
  out->nreads = 0;
  smaltgpu_callctx serial;
  if (m->history && !(ctx && ctx->hitlist_len)) {          // serial-order mode: the capacity of the one hit list of a serial run
    if (ctx) serial = *ctx; else memset(&serial, 0, sizeof(serial));
    m->h_alloclen.resize(nreads ? nreads : 1);
    const uint64_t k = (uint64_t)m->ix->d.k;
    for (uint32_t i = 0; i < nreads; i++) {
      const uint64_t len = read_off[i + 1] - read_off[i];
      if (len >= k && len > m->hist_longest) m->hist_longest = (uint32_t)len;
      m->h_alloclen[i] = m->hist_longest;
    }
    serial.hitlist_len = m->h_alloclen.data();
    ctx = &serial;
  }
  rv = map_range(m, bases, quals, read_off, nreads, par, out, ctx);
  if (rv == SMALTGPU_ECAP && out->nreads == nreads && nreads > 0 && !m->debug) return remap_overflowed(m, bases, quals, read_off, nreads, par, out, ctx);
  return rv;
}

extern "C" int smaltgpu_mapper_set_host_threads(smaltgpu_mapper *m, int nthreads) {
  if (!m) return fail(SMALTGPU_EARG, "null mapper");
  m->host_threads = nthreads < 1 ? 1 : (nthreads > 64 ? 64 : nthreads);
  return SMALTGPU_OK;
}

extern "C" int smaltgpu_mapper_set_history(smaltgpu_mapper *m, int on) {
  if (!m) return fail(SMALTGPU_EARG, "null mapper");
  m->history = on != 0;
  m->hist_longest = 0;
  return SMALTGPU_OK;
}

// ---- rounds over batches that are resident in HBM (reads and mates of a block of pairs): the round's reads are gathered on the
// device; only the offsets (lengths) are needed on the host ----
static int gather_round(smaltgpu_mapper *m, const smaltgpu_resident_reads *src, const uint32_t *ids, uint32_t nreads, bool *with_quals) {
  m->h_off.resize((size_t)nreads + 1);
  uint64_t tot = 0;
  for (uint32_t i = 0; i < nreads; i++) {
    const uint32_t w = ids[i] & 1u, r = ids[i] >> 1;
    if (r >= src->nreads[w]) return fail(SMALTGPU_EARG, "read id %u is outside its batch", ids[i]);
    const uint64_t len = src->read_off[w][r + 1] - src->read_off[w][r];
    if (len > m->max_len) return fail(SMALTGPU_EARG, "read %u is longer than the mapper's max_read_len", i);
    m->h_off[i] = tot;
    tot += len;
  }
  m->h_off[nreads] = tot;
  if (tot > m->max_bases) return fail(SMALTGPU_EARG, "batch exceeds the mapper's base capacity");
  *with_quals = src->d_quals[0] && src->d_quals[1];
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipMemcpyAsync(m->d_off, m->h_off.data(), ((size_t)nreads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->d_ids, ids, (size_t)nreads * sizeof(uint32_t), hipMemcpyHostToDevice, m->stream));
  const int rv = launch_gather_reads(m->stream, m->d_bases, *with_quals ? m->d_quals : nullptr, m->d_off, nreads, m->d_ids, src->d_bases, src->d_quals, src->d_read_off);
  if (rv) return fail(SMALTGPU_ENODEV, "kernel launch failed: %s", hipGetErrorString((hipError_t)rv));
  return SMALTGPU_OK;
}

extern "C" int smaltgpu_map_batch_ctx_resident(smaltgpu_mapper *m, const smaltgpu_resident_reads *src, const uint32_t *ids, uint32_t nreads,
                                                const smaltgpu_params *par, const smaltgpu_callctx *ctx, smaltgpu_batch_out *out) {
  if (!m || !src || !ids || !par || !out || !src->d_bases[0] || !src->d_bases[1] || !src->d_read_off[0] || !src->d_read_off[1] || !src->read_off[0] || !src->read_off[1])
    return fail(SMALTGPU_EARG, "null argument");
  if (nreads > m->max_reads) return fail(SMALTGPU_EARG, "batch of %u reads exceeds the mapper's capacity %u", nreads, m->max_reads);
  int rv = check_par(m, par);
  if (rv) return rv;
  out->nreads = 0;
  bool wq = false;
  if ((rv = gather_round(m, src, ids, nreads, &wq))) return rv;
  CtxDev cd;
  if (ctx) { const int cr = upload_ctx(m, ctx, nreads, &cd); if (cr) return cr; }
  rv = run_pipeline(m, m->d_bases, wq ? m->d_quals : nullptr, m->d_off, nreads, par, ctx ? &cd : nullptr, false);
  m->have_host_off = true;
  if (!rv) rv = smaltgpu_fetch_results(m, out);
  if (rv == SMALTGPU_ECAP && out->nreads == nreads && nreads > 0 && !m->debug) {
    // a pool overflowed: the reads that did not fit are mapped again in smaller batches, from a host copy of the gathered round
    const uint64_t tot = m->h_off[nreads];
    std::vector<uint8_t> hb(tot + 1), hq(wq ? tot + 1 : 0);
    std::vector<uint64_t> ho(m->h_off.begin(), m->h_off.begin() + nreads + 1);
    HIPCHK(hipMemcpy(hb.data(), m->d_bases, tot, hipMemcpyDeviceToHost));
    if (wq) HIPCHK(hipMemcpy(hq.data(), m->d_quals, tot, hipMemcpyDeviceToHost));
    return remap_overflowed(m, hb.data(), wq ? hq.data() : nullptr, ho.data(), nreads, par, out, ctx);
  }
  return rv;
}

extern "C" int smaltgpu_hit_totals_resident(smaltgpu_mapper *m, const smaltgpu_resident_reads *src, const uint32_t *ids, uint32_t nreads,
                                             const smaltgpu_params *par, uint32_t *nhits) {
  if (!m || !src || !ids || !par || !nhits) return fail(SMALTGPU_EARG, "null argument");
  if (nreads > m->max_reads) return fail(SMALTGPU_EARG, "batch of %u reads exceeds the mapper's capacity %u", nreads, m->max_reads);
  int rv = check_par(m, par);
  if (rv) return rv;
  bool wq = false;
  if ((rv = gather_round(m, src, ids, nreads, &wq))) return rv;
  rv = run_pipeline(m, m->d_bases, wq ? m->d_quals : nullptr, m->d_off, nreads, par, nullptr, true);
  if (rv) return rv;
  m->h_hi.resize(2 * (size_t)nreads + 1);
  if (nreads) HIPCHK(hipMemcpyAsync(m->h_hi.data(), m->b.hi, 2 * (size_t)nreads * sizeof(HitInfoHdr), hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  for (int i = 0; i < T_NUM; i++) { float f = 0; (void)hipEventElapsedTime(&f, m->ev[i], m->ev[i + 1]); m->ms[i] = f; }      // seeding only: the later stages read 0
  memset(m->work, 0, sizeof(m->work));
  for (uint32_t i = 0; i < nreads; i++) nhits[i] = m->h_hi[2 * (size_t)i].nhit_cut + m->h_hi[2 * (size_t)i + 1].nhit_cut;
  return SMALTGPU_OK;
}

extern "C" int smaltgpu_timers(const smaltgpu_mapper *m, double *ms, uint64_t *work, int n) {
  if (!m) return fail(SMALTGPU_EARG, "null mapper");
  for (int i = 0; i < n && i < T_NUM; i++) if (ms) ms[i] = m->ms[i];
  for (int i = 0; i < n && i < WK_NWORK; i++) if (work) work[i] = m->work[i];
  return T_NUM;
}

// ------------------------------------------------------------------------------------------
template <class T>
static int fetch(std::vector<T> &v, const T *dev, size_t n) {
  v.resize(n ? n : 1);
  if (n) HIPCHK(hipMemcpy(v.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost));
  return 0;
}

extern "C" long smaltgpu_dump_read(smaltgpu_mapper *m, uint32_t i, const char *name, char *buf, size_t bufsiz) {
  if (!m) return fail(SMALTGPU_EARG, "null mapper");
  if (!m->debug || !m->cand_scr_dbg) return fail(SMALTGPU_EARG, "smaltgpu_set_debug(m, 1) must precede the batch");
  if (i >= m->last_n || !m->have_host_off) return fail(SMALTGPU_EARG, "read index out of range");
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipStreamSynchronize(m->stream));
  const Batch &b = m->b;
  const DevIndex &d = m->ix->d;
  DumpView v;
  std::vector<HitInfoHdr> hi; std::vector<SeedRec> seeds; std::vector<uint8_t> qmask;
  std::vector<CandHdr> ch; std::vector<ReadCtl> ctl; std::vector<ReadStat> st;
  std::vector<uint8_t> slot;
  if (fetch(hi, b.hi + 2 * (size_t)i, 2) || fetch(seeds, b.seeds + 2 * (size_t)i * m->qmax, 2 * (size_t)m->qmax) ||
      fetch(qmask, b.qmask + 2 * (size_t)i * m->qmax, 2 * (size_t)m->qmax) || fetch(ch, b.ch + i, 1) || fetch(ctl, b.ctl + i, 1) ||
      fetch(st, b.stat + i, 1) || fetch(slot, m->cand_scr_dbg + m->cand_bytes * i, m->cand_bytes)) return SMALTGPU_ENODEV;
  const uint32_t ngrp = (m->last_par.flags & FLG_SEQBYSEQ) ? (uint32_t)d.nseq : 1u;
  const int dk = m->last_fine ? (int)FINE_K : d.k, dsx = m->last_fine ? (int)FINE_S : d.s;
  const bool v2 = cands_v2_applicable(m->last_par, dk, dsx, (uint32_t)(m->h_off[i + 1] - m->h_off[i]), b.iv_off != nullptr);
  CandScratch cx = cand_scratch_carve(slot.data(), m->qmax, dsx, m->cg.hcap, ngrp, m->cg.segcap, m->cg.candcap);
  CandsV2Scratch c2 = cands_v2_carve(nullptr, 0, slot.data(), m->qmax, dsx, m->cg.hcap_strand, ngrp, m->cg.candcap, true);
  std::vector<RCand> rc; std::vector<Result> res; std::vector<uint8_t> dstr;
  if (fetch(rc, b.rcpool + ch[0].rc_off, ch[0].n_sort) || fetch(res, b.respool + st[0].res_off, st[0].nres)) return SMALTGPU_ENODEV;
  size_t nd = 0;
  for (uint32_t j = 0; j < st[0].nres; j++) { size_t e = (size_t)res[j].stroffs + res[j].strlen; if (e > nd) nd = e; }
  if (fetch(dstr, b.dstrpool + st[0].dstr_off, nd)) return SMALTGPU_ENODEV;
  v.qlen = (uint32_t)(m->h_off[i + 1] - m->h_off[i]); v.qmax = m->qmax; v.k = dk;
  for (int s2 = 0; s2 < 2; s2++) { v.hi[s2] = hi[s2]; v.seeds[s2] = seeds.data() + (size_t)s2 * m->qmax; v.qmask[s2] = qmask.data() + (size_t)s2 * m->qmax; }
  v.ch = ch[0]; v.rc = rc.data(); v.ctl = ctl[0]; v.st = st[0]; v.res = res.data(); v.dstr = dstr.data(); v.ngrp = ngrp;
  std::vector<SegCand> crec;
  if (v2) { cands_v2_records(crec, c2, ch[0].ncand <= m->cg.candcap ? ch[0].ncand : 0, m->qmax > 255); v.cand = crec.data(); v.sort_idx = c2.sort_idx; v.sort_keys = c2.sort_keys; v.hitwords = c2.dbg_words; v.grp_first = c2.dbg_first; v.grp_cnt = c2.dbg_cnt; }
  else { v.cand = cx.cand; v.sort_idx = cx.sort_idx; v.sort_keys = cx.sort_keys; v.hitwords = cx.keys; v.grp_first = cx.grp_first; v.grp_cnt = cx.grp_cnt; }
  std::string o;
  dump_read(o, v, i, name, m->debug >= 2);
  if (buf && bufsiz) { size_t c = o.size() < bufsiz - 1 ? o.size() : bufsiz - 1; memcpy(buf, o.data(), c); buf[c] = 0; }
  return (long)o.size();
}

extern "C" int smaltgpu_sw_full_batch(smaltgpu_mapper *m, const uint8_t *qcodes, const uint32_t *q_off, const uint8_t *rcodes,
                                       const uint32_t *r_off, uint32_t ntask, const smaltgpu_params *par, int32_t *scores, int packed16) {
  if (!m || !qcodes || !q_off || !rcodes || !r_off || !par || !scores) return fail(SMALTGPU_EARG, "null argument");
  HIPCHK(hipSetDevice(m->device));
  uint8_t *dq = nullptr, *dr = nullptr; uint32_t *dqo = nullptr, *dro = nullptr; int32_t *dsc = nullptr;
  uint32_t qmaxlen = 0;
  for (uint32_t t = 0; t < ntask; t++) { uint32_t l = q_off[t + 1] - q_off[t]; if (l > qmaxlen) qmaxlen = l; }
  int rv = 0;
  if (dalloc(&dq, q_off[ntask] + 16) || dalloc(&dr, r_off[ntask] + 16) || dalloc(&dqo, (size_t)ntask + 1) || dalloc(&dro, (size_t)ntask + 1) || dalloc(&dsc, ntask)) rv = SMALTGPU_ENOMEM;
  if (!rv) {
    (void)hipMemcpy(dq, qcodes, q_off[ntask], hipMemcpyHostToDevice);
    (void)hipMemcpy(dr, rcodes, r_off[ntask], hipMemcpyHostToDevice);
    (void)hipMemcpy(dqo, q_off, ((size_t)ntask + 1) * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dro, r_off, ((size_t)ntask + 1) * 4, hipMemcpyHostToDevice);
    uint32_t rmaxlen = 0;
    for (uint32_t t = 0; t < ntask; t++) { uint32_t l = r_off[t + 1] - r_off[t]; if (l > rmaxlen) rmaxlen = l; }
    int lr;
    if (!packed16 && (qmaxlen > 512 || rmaxlen > (uint32_t)SW_FULL_WMAX))     // beyond the register tiling: the strip kernel (windows up to the mapper's capacity)
      lr = launch_sw_strip_raw(m->stream, dq, dqo, dr, dro, ntask, to_par(par), dsc, m->strip_bnd, m->strip_win, m->wincap, m->strip_grid);
    else lr = launch_sw_full_raw(m->stream, dq, dqo, dr, dro, ntask, to_par(par), dsc, qmaxlen, packed16);
    if (lr) rv = fail(SMALTGPU_EARG, lr == -2 ? "scores do not fit the packed 16-bit kernel" : "query longer than the register-tiled kernel supports");
    else if (hipStreamSynchronize(m->stream) != hipSuccess) rv = fail(SMALTGPU_ENODEV, "kernel failed");
    else (void)hipMemcpy(scores, dsc, (size_t)ntask * 4, hipMemcpyDeviceToHost);
  }
  (void)hipFree(dq); (void)hipFree(dr); (void)hipFree(dqo); (void)hipFree(dro); (void)hipFree(dsc);
  return rv;
}

extern "C" int smaltgpu_rank_sort_batch(smaltgpu_mapper *m, const uint32_t *keys, const uint32_t *off, uint32_t narr, int nneed, int in_lds,
                                         uint32_t *out_keys, uint32_t *out_idx) {
  if (!m || !keys || !off || !out_keys || !out_idx) return fail(SMALTGPU_EARG, "null argument");
  HIPCHK(hipSetDevice(m->device));
  const size_t tot = off[narr];
  for (uint32_t t = 0; t < narr; t++) if (off[t + 1] < off[t] || off[t + 1] - off[t] > (1u << 22)) return fail(SMALTGPU_EARG, "bad array offsets");
  for (size_t i = 0; i < tot; i++) if (keys[i] >= 1024u) return fail(SMALTGPU_EARG, "key out of range");
  uint32_t *dk = nullptr, *doff = nullptr, *dkv = nullptr, *dok = nullptr, *doi = nullptr;
  int rv = 0;
  if (dalloc(&dk, tot + 1) || dalloc(&doff, (size_t)narr + 1) || dalloc(&dkv, tot + 1) || dalloc(&dok, tot + 1) || dalloc(&doi, tot + 1)) rv = SMALTGPU_ENOMEM;
  if (!rv) {
    (void)hipMemcpy(dk, keys, tot * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(doff, off, ((size_t)narr + 1) * 4, hipMemcpyHostToDevice);
    launch_rank_sort_raw(m->stream, dk, doff, narr, nneed, in_lds, dkv, dok, doi);
    if (hipStreamSynchronize(m->stream) != hipSuccess) rv = fail(SMALTGPU_ENODEV, "kernel failed");
    else { (void)hipMemcpy(out_keys, dok, tot * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(out_idx, doi, tot * 4, hipMemcpyDeviceToHost); }
  }
  (void)hipFree(dk); (void)hipFree(doff); (void)hipFree(dkv); (void)hipFree(dok); (void)hipFree(doi);
  return rv;
}
