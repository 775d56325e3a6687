// smg_dump.hpp -- host-side rendering of the per-stage state of one read in the line format of
// oracle/DUMPFORMAT.md (the format oracle/refdump prints for the unmodified reference), from
// host copies of the device buffers.  Diagnostic code: used by smaltgpu_dump_read().
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "smg_stages.hpp"
#include "smg_cands.hpp"

namespace smg {

struct DumpView {               // host-visible copies for ONE read
  uint32_t qlen, qmax;
  HitInfoHdr hi[2];
  const SeedRec *seeds[2];
  const uint8_t *qmask[2];
  CandHdr ch;
  const SegCand *cand;          // ch.ncand
  const uint32_t *sort_idx, *sort_keys;
  const RCand *rc;              // ch.n_sort ranked candidates
  ReadCtl ctl;
  ReadStat st;
  const Result *res;
  const uint8_t *dstr;
  const uint64_t *hitwords;     // packed hit words grouped by (strand, seq)
  const uint32_t *grp_first, *grp_cnt;
  uint32_t ngrp;
  int k;
};

inline void appendf(std::string &o, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
inline void appendf(std::string &o, const char *fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  int n = vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (n > 0) o.append(buf, (size_t)(n < (int)sizeof(buf) ? n : (int)sizeof(buf) - 1));
}

// The candidate records of a wave-parallel slot as SegCand: reads below 256 bases (the lean kernel instance) hold
// 16-byte records, with the cover in the byte array and, in debug slots, segment count and sequence number beside them.
inline void cands_v2_records(std::vector<SegCand> &out, const CandsV2Scratch &x, uint32_t ncand, bool long_instance) {
  out.resize(ncand ? ncand : 1);
  for (uint32_t i = 0; i < ncand; i++) {
    if (long_instance) { out[i] = x.cand[i]; continue; }
    segcand_unpack(out[i], ((const SegCandP *)x.cand)[i], (uint32_t)x.cover8[i]);
    if (x.lw.dbg_nseg) { out[i].nseg = x.lw.dbg_nseg[i]; out[i].seqidx = x.lw.dbg_seq[i]; }
  }
}

inline void dump_read(std::string &o, const DumpView &v, unsigned long long readno, const char *name, bool with_hitlists) {
  appendf(o, "READ %llu %s len=%u err=%d\n", readno, name ? name : "-", v.qlen, 0);
  const bool mapped = v.qlen >= (uint32_t)v.k;
  if (mapped) {
    for (int st = 0; st < 2; st++) {
      const char sc = st ? 'R' : 'F';
      appendf(o, "HI %c nseeds=%u rank=%u status=%u\n", sc, v.hi[st].n_seeds, v.hi[st].seed_rank, v.hi[st].status);
      appendf(o, "QM %c ", sc);
      for (uint32_t i = 0; i < v.qlen; i++) o.push_back((char)('0' + v.qmask[st][i]));
      o.push_back('\n');
      for (uint32_t i = 0; i < v.hi[st].n_seeds; i++)
        appendf(o, "SD %c %u %u %u %u\n", sc, i, v.seeds[st][i].qoffs, v.seeds[st][i].nhits, v.seeds[st][i].posidx);
    }
    for (uint32_t i = 0; i < v.ch.ncand; i++) {
      const SegCand &c = v.cand[i];
      appendf(o, "CA %u %u %u %u %u %d %d %d %u %u %d %d\n", i, c.qs, c.qe, c.rs, c.re, (int)c.shiftoffs, (int)c.srange,
              (int)c.shift2mm, c.cover, (unsigned)c.flag, c.nseg, c.seqidx);
    }
    appendf(o, "ST %u %u %u %u %u %u %u\n", v.ch.max_cover, v.ch.max2nd_cover, v.ch.ncand, v.ch.n_mincover, v.ch.n_sort,
            v.ch.cover_deficit[0], v.ch.cover_deficit[1]);
    for (uint32_t i = 0; i < v.ch.n_sort; i++) appendf(o, "SI %u %u %u\n", i, v.sort_idx[i], v.sort_keys[i]);
    for (int i = 0; i < v.ctl.n_scored; i++) {
      const RCand &c = v.rc[i];
      appendf(o, "RC %d %u %u %u %llu %llu %d %d %lld %d\n", i, (c.flags & RCF_REVERSE) | RCF_SCORED, c.qs, c.qe,
              (unsigned long long)c.rs, (unsigned long long)c.re, c.band_l, c.band_r, (long long)c.sqidx, c.swscor);
    }
  }
  for (uint32_t i = 0; i < v.st.nres; i++) {
    const Result &r = v.res[i];
    appendf(o, "RS %u %c %d %u %u %llu %llu %lld ", i, r.reverse ? 'R' : 'F', r.swatscor, r.q_start, r.q_end,
            (unsigned long long)r.s_start, (unsigned long long)r.s_end, (long long)r.sidx);
    for (uint32_t j = 0; j < r.strlen; j++) appendf(o, "%02x", (unsigned)v.dstr[r.stroffs + j]);
    o.push_back('\n');
  }
  appendf(o, "RX %u %d %d %d %d %u %u\n", v.st.nres, v.st.swmax, v.st.sw2nd, mapped ? v.st.nseg : 0, mapped ? v.st.nseg_tot : 0,
          mapped ? v.st.nhit : 0u, mapped ? v.st.nhit_tot : 0u);
  if (with_hitlists && mapped) {
    for (uint32_t st = 0; st < 2; st++)
      for (uint32_t g = 0; g < v.ngrp; g++) {
        const uint32_t cnt = v.grp_cnt[st * v.ngrp + g], first = v.grp_first[st * v.ngrp + g];
        appendf(o, "HL %c %u %u", st ? 'R' : 'F', g, cnt);
        for (uint32_t i = 0; i < cnt; i++) appendf(o, " %llx", (unsigned long long)v.hitwords[first + i]);
        o.push_back('\n');
      }
  }
}

}  // namespace smg
