// tests/hostemu/emu_main.cpp -- TEST-ONLY host build of the per-read stage logic that the HIP
// kernels run (smalt_amd/csrc/smg_stages.hpp compiled with SMG_NLANES = 1).  It lets the CPU
// test-suite check that logic against the reference's stage dumps on a machine without a GPU.
// It is NOT part of libsmaltgpu.so and nothing in the product links or calls it.
// usage: emu [-m minscor] [-d scordiff] [-c mincover] [-q minbasq] [-H ncut] [-x] [-n] <index_prefix> <reads.fq>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <string>
#include <vector>
#include "../../smalt_amd/csrc/smg_cands.hpp"
#include "../../smalt_amd/csrc/smg_dump.hpp"
#include "../../smalt_amd/csrc/smg_indexfile.hpp"

using namespace smg;

static uint8_t code_of(unsigned char c) {       // sequence.c:287-322
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 5;
  }
}

int main(int argc, char **argv) {
  int c, minscor = -1, scordiff = 0, ncut = 10000, minbasq = 0, xflag = 0, with_hl = 1;
  double mincover = 0.0;
  while ((c = getopt(argc, argv, "m:d:c:q:H:xn")) != -1) {
    switch (c) {
      case 'm': minscor = atoi(optarg); break;
      case 'd': scordiff = atoi(optarg); break;
      case 'c': mincover = atof(optarg); break;
      case 'q': minbasq = atoi(optarg); break;
      case 'H': ncut = atoi(optarg); break;
      case 'x': xflag = 1; break;
      case 'n': with_hl = 0; break;
      default: return 2;
    }
  }
  if (argc - optind < 2) return 2;
  HostIndex hix;
  std::string err;
  if (!read_index_files(argv[optind], hix, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
  std::vector<uint32_t> seqlo((size_t)hix.nseq + 1);
  for (int64_t i = 0; i <= hix.nseq; i++) seqlo[(size_t)i] = (uint32_t)(hix.sop[(size_t)i] / (uint64_t)hix.s);
  DevIndex ix;
  ix.k = hix.k; ix.s = hix.s; ix.typ = hix.typ; ix.nbits_key = hix.nbits_key; ix.nbits_lo = hix.nbits_lo;
  ix.nkeys = hix.nkeys; ix.npos = hix.npos; ix.nwords = hix.nwords;
  ix.idx = hix.idx.data(); ix.pos = hix.pos.data(); ix.wordidx = hix.wordidx.data(); ix.posidx = hix.posidx.data();
  ix.packed = hix.packed.data(); ix.sop = hix.sop.data(); ix.seqlo = seqlo.data(); ix.nseq = (int32_t)hix.nseq; ix.totlen = hix.totlen;

  MapPar p; p.cov_frac = 0.0;
  p.ncut = ncut; p.min_cover = 0; p.min_swatscor = minscor >= 0 ? minscor : ix.k + ix.s - 1; p.below_max = scordiff;
  p.min_basq = minbasq; p.target_depth = 512; p.max_depth = 2048;
  p.flags = (scordiff ? 0 : FLG_BEST) | (ix.nseq < 512 ? FLG_SEQBYSEQ : 0) | (xflag ? (FLG_NOSHRTINFO | FLG_SENSITIVE) : 0);
  p.match = 1; p.mismatch = -2; p.gap_init = -4; p.gap_ext = -3;

  // ---- read the FASTQ into one batch ----
  std::vector<std::string> names;
  std::vector<uint8_t> codes, codes_rc, qual;
  std::vector<uint64_t> off(1, 0);
  FILE *fp = fopen(argv[optind + 1], "r");
  if (!fp) return 1;
  char *l1 = nullptr, *l2 = nullptr, *l3 = nullptr, *l4 = nullptr;
  size_t c1 = 0, c2 = 0, c3 = 0, c4 = 0;
  uint32_t qmaxlen = 0;
  while (getline(&l1, &c1, fp) > 0 && getline(&l2, &c2, fp) > 0 && getline(&l3, &c3, fp) > 0 && getline(&l4, &c4, fp) > 0) {
    uint32_t len = (uint32_t)strcspn(l2, "\r\n");
    l1[strcspn(l1, " \t\r\n")] = 0;
    names.emplace_back(l1 + 1);
    for (uint32_t i = 0; i < len; i++) { codes.push_back(code_of((unsigned char)l2[i])); qual.push_back((uint8_t)l4[i]); }
    for (uint32_t i = 0; i < len; i++) { uint8_t cc = codes[off.back() + len - 1 - i]; codes_rc.push_back((uint8_t)((cc & 4) ? cc : 3 - cc)); }
    off.push_back(off.back() + len);
    if (len > qmaxlen) qmaxlen = len;
  }
  fclose(fp);
  const uint32_t n = (uint32_t)names.size();
  const uint32_t qmax = qmaxlen + 8;
  const bool seqbyseq = (p.flags & FLG_SEQBYSEQ) != 0;
  const uint32_t ngrp = seqbyseq ? (uint32_t)ix.nseq : 1u;

  std::vector<HitInfoHdr> hi(2 * (size_t)n);
  std::vector<SeedRec> seeds(2 * (size_t)n * qmax);
  std::vector<uint8_t> qmask(2 * (size_t)n * qmax, 0);
  std::vector<CandHdr> ch(n);
  std::vector<ReadCtl> ctl(n);
  std::vector<ReadStat> stat(n);
  const uint32_t rccap = n * 2048u + 16;
  std::vector<RCand> rcpool(rccap);
  std::vector<Result> respool((size_t)n * 4096 + 16);
  std::vector<uint8_t> dstrpool((size_t)n * 4096 * 64 + 16);
  uint32_t rc_count = 0; unsigned long long res_count = 0, dstr_count = 0; int32_t err_flag = 0;
  Batch b;
  memset((void *)&b, 0, sizeof(b));        // no retry list, no pair context
  b.nreads = n; b.qmax = qmax; b.codes = codes.data(); b.codes_rc = codes_rc.data(); b.qual = qual.data(); b.read_off = off.data();
  b.hi = hi.data(); b.seeds = seeds.data(); b.qmask = qmask.data(); b.ch = ch.data(); b.rcpool = rcpool.data(); b.rccap = rccap;
  b.rc_count = &rc_count; b.ctl = ctl.data(); b.stat = stat.data(); b.respool = respool.data(); b.rescap = respool.size();
  b.res_count = &res_count; b.dstrpool = dstrpool.data(); b.dstrcap = dstrpool.size(); b.dstr_count = &dstr_count; b.err_flag = &err_flag;
  // EMU_HISTORY=1: the file is one serial run -- the hit list's capacity follows the longest read (>= k bases) so far
  std::vector<uint32_t> alloc_len(n ? n : 1);
  if (getenv("EMU_HISTORY") && atoi(getenv("EMU_HISTORY")) != 0) {
    uint32_t longest = 0;
    for (uint32_t r = 0; r < n; r++) { const uint32_t len = (uint32_t)(off[r + 1] - off[r]); if (len >= (uint32_t)ix.k && len > longest) longest = len; alloc_len[r] = longest; }
    b.alloc_len = alloc_len.data();
  }
  unsigned long long workctr[WK_NWORK] = {0};
  b.work = workctr; b.long_list = nullptr; b.long_cap = 0; b.strip_list = nullptr; b.strip_cap = 0; b.tile_qmax = 512;

  std::vector<uint8_t> sscr(seed_scratch_bytes(qmax, ix.s));
  const uint32_t hcap = 1u << 16, segcap = 1u << 15, candcap = 1u << 16;
  const uint32_t hcap_strand = hcap / 2;
  std::vector<uint8_t> ldsmem(((strand_work_bytes<uint16_t>(CANDS_LDS_HITS) + 15) & ~(size_t)15) + CANDS_TAB_BYTES);   // stands for the workgroup's LDS block
  const uint32_t window = getenv("EMU_WINDOW") ? (uint32_t)atoi(getenv("EMU_WINDOW")) : 0;   // hits per window of the candidate stage
  const char *force = getenv("EMU_CANDS");       // "v1": sequential restatement only; default: as the kernel chooses
  size_t cbytes = cand_scratch_bytes(qmax, ix.s, hcap, ngrp, segcap, candcap);
  { size_t b2 = cands_v2_hbm_bytes(qmax, ix.s, hcap_strand, ngrp, candcap, true); if (b2 > cbytes) cbytes = b2; }
  uint8_t *cscr_mem = (uint8_t *)malloc(cbytes * (size_t)n);      // one slot per read: state kept for the dump
  struct { uint8_t *p; uint8_t *data() { return p; } } cscr = {cscr_mem};
  const uint32_t wincap = 4 * qmax + 4096; const uint64_t dircap = (uint64_t)wincap * (qmax + 64);
  std::vector<uint8_t> ascr(align_scratch_bytes(qmax, wincap, dircap, 4096, 4096 * (qmax / 4 + 48)));
  std::vector<int> Hrow(qmax + 2), Erow(qmax + 2);
  int8_t M[64];
  score_matrix(M, p.match, p.mismatch);

  // S3 as a stage of its own (k_hits on the device): EMU_SPLIT=1 runs stage_hits ahead of the candidate stage, which then
  // streams the sorted keys from the pool (HITRUN_SORTED); EMU_HITS_WINDOW sets the keys per window of stage_hits
  const bool split = getenv("EMU_SPLIT") && atoi(getenv("EMU_SPLIT")) != 0;
  const uint32_t hitsW = getenv("EMU_HITS_WINDOW") ? (uint32_t)atoi(getenv("EMU_HITS_WINDOW")) : 512u;
  std::vector<uint64_t> hitpool(split ? (size_t)n * 2 * hcap_strand / 8 + (1u << 20) : 1);
  std::vector<HitRun> hitrun((size_t)2 * n + 1);
  unsigned long long hit_count = 0;
  std::vector<uint8_t> hitlds(hits_lds_bytes(hitsW, CANDS_TAB) + 64);
  if (split) { b.hitpool = hitpool.data(); b.hitpool_cap = hitpool.size(); b.hitrun = hitrun.data(); b.hit_count = &hit_count; }
  unsigned long long nwin_strands = 0, nhbm_strands = 0, nsplit_strands = 0;
  for (uint32_t r = 0; r < n; r++) {
    const uint32_t len = (uint32_t)(off[r + 1] - off[r]);
    if (mincover < 1.01) { p.min_cover = (uint32_t)(mincover * len); if (p.min_cover > len) p.min_cover = len; }
    else p.min_cover = (uint32_t)mincover;
    SeedScratch sx = seed_scratch_carve(sscr.data(), qmax, ix.s);
    stage_seed(b, ix, p, r, 0, sx);
    stage_seed(b, ix, p, r, 1, sx);
    if (cands_v2_applicable(p, ix.k, ix.s, len, false) && !(force && !strcmp(force, "v1"))) {
      CandsV2Scratch c2 = cands_v2_carve(ldsmem.data(), ldsmem.size(), cscr.data() + cbytes * r, qmax, ix.s, hcap_strand, ngrp, candcap, true);
      c2.window = window;
      unsigned long long ph[16] = {0};
      if (split && len <= 255) {
        HitsScratch hx;
        hx.lds = hitlds.data(); hx.lds_bytes = hitlds.size(); hx.W = hitsW; hx.tab = CANDS_TAB;
        unsigned long long hph[4] = {0, 0, 0, 0};
        stage_hits(b, ix, p, r, 0, hx, hph);
        stage_hits(b, ix, p, r, 1, hx, hph);
        nsplit_strands += (hitrun[2 * r].mode == HITRUN_SORTED) + (hitrun[2 * r + 1].mode == HITRUN_SORTED);
        stage_cands_v2<false, true>(b, ix, p, r, c2, ph);
      } else if (len > 255) stage_cands_v2<true>(b, ix, p, r, c2, ph);
      else stage_cands_v2<false>(b, ix, p, r, c2, ph);
      nwin_strands += ph[11]; nhbm_strands += ph[14];
    } else {
      CandScratch cx = cand_scratch_carve(cscr.data() + cbytes * r, qmax, ix.s, hcap, ngrp, segcap, candcap);
      stage_cands(b, ix, p, r, cx);
    }
    // score pass: K2a (un-banded) or K2b (banded) by the predicate set in cand_offsets
    for (uint32_t i = 0; i < ch[r].n_sort; i++) {
      RCand &rc = rcpool[ch[r].rc_off + i];
      if (rc.flags & RCF_ERR) continue;
      const uint8_t *q = ((rc.flags & RCF_REVERSE) ? codes_rc.data() : codes.data()) + off[r];
      const uint64_t gbase = (rc.sqidx < 0 ? 0 : hix.sop[(size_t)rc.sqidx]) + rc.rs;
      const uint32_t wlen = (uint32_t)(rc.re - rc.rs + 1);
      bool banded = (rc.flags & RCF_BANDED) != 0;
      if (!banded) { rc.swscor = sw_full_scalar(q, len, ix.packed, gbase, wlen, M, -p.gap_init, -p.gap_ext, Hrow.data(), Erow.data()); if (rc.swscor >= 65535) banded = true; }
      if (banded) {
        Band bd;
        if (band_init(bd, rc.band_l, rc.band_r, (int)rc.qs, (int)rc.qe, (int)len, 0, (int)wlen - 1, (int)wlen)) { rc.flags |= RCF_ERR; continue; }
        rc.swscor = band_fast_scalar(bd, q, ix.packed, gbase, M, -p.gap_init, -p.gap_ext, Hrow.data(), Erow.data());
      }
      rc.flags |= RCF_SCORED;
    }
    stage_replay(b, ix, p, r);
    AlignScratch ax = align_scratch_carve(ascr.data(), qmax, wincap, dircap, 4096, 4096 * (qmax / 4 + 48));
    stage_align(b, ix, p, r, ax);
    align_tally_flush(b, ax);
  }
  std::string out;
  std::vector<SegCand> crec;
  for (uint32_t r = 0; r < n; r++) {
    CandScratch cx = cand_scratch_carve(cscr.data() + cbytes * r, qmax, ix.s, hcap, ngrp, segcap, candcap);
    DumpView v;
    v.qlen = (uint32_t)(off[r + 1] - off[r]); v.qmax = qmax; v.k = ix.k;
    if (mincover < 1.01) { p.min_cover = (uint32_t)(mincover * v.qlen); if (p.min_cover > v.qlen) p.min_cover = v.qlen; }   // as when the read was mapped
    else p.min_cover = (uint32_t)mincover;
    for (int st = 0; st < 2; st++) { v.hi[st] = hi[2 * r + st]; v.seeds[st] = seeds.data() + (size_t)(2 * r + st) * qmax; v.qmask[st] = qmask.data() + (size_t)(2 * r + st) * qmax; }
    v.ch = ch[r]; v.rc = rcpool.data() + ch[r].rc_off;
    v.ctl = ctl[r]; v.st = stat[r]; v.res = respool.data() + stat[r].res_off; v.dstr = dstrpool.data() + stat[r].dstr_off; v.ngrp = ngrp;
    if (cands_v2_applicable(p, ix.k, ix.s, v.qlen, false) && !(force && !strcmp(force, "v1"))) {
      CandsV2Scratch c2 = cands_v2_carve(nullptr, 0, cscr.data() + cbytes * r, qmax, ix.s, hcap_strand, ngrp, candcap, true);
      cands_v2_records(crec, c2, v.ch.ncand <= candcap ? v.ch.ncand : 0, v.qlen > 255); v.cand = crec.data(); v.sort_idx = c2.sort_idx; v.sort_keys = c2.sort_keys; v.hitwords = c2.dbg_words; v.grp_first = c2.dbg_first; v.grp_cnt = c2.dbg_cnt;
    } else { v.cand = cx.cand; v.sort_idx = cx.sort_idx; v.sort_keys = cx.sort_keys; v.hitwords = cx.keys; v.grp_first = cx.grp_first; v.grp_cnt = cx.grp_cnt; }
    out.clear();
    dump_read(out, v, r, names[r].c_str(), with_hl != 0);
    fwrite(out.data(), 1, out.size(), stdout);
  }
  if (err_flag) {
    fprintf(stderr, "emu: %d reads with errors\n", err_flag);
    for (uint32_t r = 0; r < n; r++) if (stat[r].err || ch[r].err) fprintf(stderr, "emu: read %u: candidate stage %d, alignment stage %d\n", r, ch[r].err, stat[r].err);
  }
  if (getenv("EMU_STATS")) fprintf(stderr, "windowed strands %llu, HBM strands %llu, strands sorted ahead %llu\n", nwin_strands, nhbm_strands, nsplit_strands);
  return 0;
}
