// smg_results.cpp -- result post-processing of the mapped reads of a batch: what the reference's
// resultSetSortAndAssignSequence (results.c:2022) does to the raw alignments of a read --
//   assignSequenceIndex            results.c:1695   concatenated mode: which sequence, offsets relative to it
//   sortAndPrune                   results.c:759    duplicates / contained alignments out, order for output, score ranks
//   labelComplementarySegments     results.c:707    groups of alignments that cover the same part of the read
//   calcPhredScaledMappingQuality  results.c:1143   mapping quality per group (+ propagateMapQualAsProb :1343)
// -- for all reads of a batch, on the host (worker threads over the reads).  Host code on purpose: the mapping quality is
// double arithmetic through libm's log/exp and the orders are libc qsort's on comparators that are not total orders
// (cmpRes compares a query length with a subject length, results.c:466-470), so bit-identical results need the very same
// libm / libc the reference runs on.  Compiled with g++ and -ffp-contract=off (the reference is plain gcc -O2).
//
// Not done here: alignments that span several reference sequences (splitMultiSpan, results.c:1472) -- such a read is
// flagged `needs_reference` and left to the caller (the reference's own code in the bound program).
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include "../../include/smaltgpu.h"

namespace {

enum {                                   // results.c:49-91
  MAPSCOR_MAX = 60, MAPSCOR_DUMMY_COUNT = 3, MAPSCOR_MAX_RANDOM = 3, MAPSCOR_MIN_UNIQ = MAPSCOR_MAX_RANDOM + 1,
  MAPSCOR_EXPFAC = 10,                   // results_mapscor_exp is defined for the build (results.h / resultpairs.h:40)
  QUALSCOR_SCAL = 10, RSLTX_INITVAL = -1, QSEGX_INITVAL = -1, N100PERCENT = 100, MIN_QSEGOVERLAP_PERCENT = 80,
  SEQCOD_QVAL_OFFS = 33                  // sequence.h: phred + 33
};
enum { F_SELECT = 0x01, F_REVERSE = 0x04, F_NOSEQID = 0x08, F_SINGLE = 0x100 };                        // results.h:67-78
enum { S_SEQX = 0x01, S_SERIALNO = 0x02, S_SWSORT = 0x04, S_SEGIDX = 0x08, S_MAPQ = 0x10 };            // results.c:93-100
const double MINLOGARG = 1E-7;           // results.c:102
const float QUALSCOR_LOGBASE = 2.30259;  // results.c:103 (a float in the reference: the arithmetic below depends on it)

struct Res {                             // struct _RESULT, results.c:121-160
  short serialno;
  uint32_t status;
  int swatscor, mapscor;
  double prob;
  uint32_t q_start, q_end;
  uint64_t s_start, s_end;
  int64_t sidx;
  uint32_t stroffs, strlen;
  short rsltx, qsegx, swrank;
};

int cmpRes(const void *p1, const void *p2) {           // results.c:450-476, including its mixed comparison of lengths
  const Res *ap = *(Res *const *)p1, *bp = *(Res *const *)p2;
  if (ap->sidx < bp->sidx) return -1;
  if (ap->sidx > bp->sidx) return 1;
  if ((ap->status & F_REVERSE) < (bp->status & F_REVERSE)) return -1;
  if ((ap->status & F_REVERSE) > (bp->status & F_REVERSE)) return 1;
  if (ap->s_start < bp->s_start) return -1;
  if (ap->s_start > bp->s_start) return 1;
  const uint32_t da = ap->q_end - ap->q_start, db = (uint32_t)(bp->s_end - bp->s_start);
  if (da > db) return -1;
  if (da < db) return 1;
  return 0;
}
int cmpResOutput(const void *p1, const void *p2) {     // results.c:478-507
  const Res *ap = *(Res *const *)p1, *bp = *(Res *const *)p2;
  if (ap->swatscor > bp->swatscor) return -1;
  if (ap->swatscor < bp->swatscor) return 1;
  if ((ap->status & F_REVERSE) < (bp->status & F_REVERSE)) return -1;
  if ((ap->status & F_REVERSE) > (bp->status & F_REVERSE)) return 1;
  if (ap->sidx < bp->sidx) return -1;
  if (ap->sidx > bp->sidx) return 1;
  if (ap->s_start < bp->s_start) return -1;
  if (ap->s_start > bp->s_start) return 1;
  const uint32_t da = ap->q_end - ap->q_start, db = bp->q_end - bp->q_start;
  if (da > db) return -1;
  if (da < db) return 1;
  return 0;
}
int cmpResSegSW(const void *p1, const void *p2) {      // results.c:509-523
  const Res *ap = *(Res *const *)p1, *bp = *(Res *const *)p2;
  if (ap->qsegx < bp->qsegx) return -1;
  if (ap->qsegx > bp->qsegx) return 1;
  if (ap->swatscor > bp->swatscor) return -1;
  if (ap->swatscor < bp->swatscor) return 1;
  return 0;
}
int cmpResSegLen(const void *p1, const void *p2) {     // results.c:525-554
  const Res *ap = *(Res *const *)p1, *bp = *(Res *const *)p2;
  if (ap->swatscor > bp->swatscor) return -1;
  if (ap->swatscor < bp->swatscor) return 1;
  const uint32_t da = ap->q_end - ap->q_start, db = bp->q_end - bp->q_start;
  if (da > db) return -1;
  if (da < db) return 1;
  if ((ap->status & F_REVERSE) < (bp->status & F_REVERSE)) return -1;
  if ((ap->status & F_REVERSE) > (bp->status & F_REVERSE)) return 1;
  if (ap->sidx < bp->sidx) return -1;
  if (ap->sidx > bp->sidx) return 1;
  if (ap->s_start < bp->s_start) return -1;
  if (ap->s_start > bp->s_start) return 1;
  return 0;
}

// sumQualOverMisMatch (results.c:232-285) with with_nonali == 0: base qualities summed over the substitutions of an alignment
int sum_qual_mismatch(int *sum, const uint8_t *qual, uint32_t slen, uint32_t pos_start, uint32_t pos_end, const uint8_t *dstr) {
  uint32_t qs = 0, spos;
  if (pos_end < pos_start) return -1;
  *sum = 0;
  spos = pos_start > 0 ? pos_start - 1 : 0;
  for (const uint8_t *dp = dstr; *dp; dp++) {
    const uint32_t gap = *dp & 0x3F, typ = *dp >> 6;                 // diffstr.h:90-107
    spos += gap;
    if (typ == 1) continue;                                          // DIFFCOD_D
    if (typ == 3) {                                                  // DIFFCOD_S
      if (!dp[1]) continue;
      if (spos < 1 || spos >= slen) return -1;
      const uint8_t q = qual[spos];
      if (q < SEQCOD_QVAL_OFFS) return -1;
      qs += (uint32_t)q - SEQCOD_QVAL_OFFS;
      if (qs > (uint32_t)INT_MAX) return -1;
    }
    spos++;
  }
  if (spos != pos_end) return -1;
  *sum = (int)qs;
  return 0;
}

struct Set {                             // the parts of struct _ResultSet this path touches
  std::vector<Res> resr;
  std::vector<Res *> sortr, segsrtr;
  std::vector<int> segnor;
  uint32_t status = 0;
  int qsegno = 0;
  int n_ali_done = 0, n_ali_tot = 0;
  uint32_t n_hits_used = 0, n_hits_tot = 0;
};

// assignSequenceIndex (results.c:1695-1781); returns 1 when an alignment spans several sequences (splitMultiSpan)
int assign_sequence_index(Set &rs, const uint64_t *ofp, int64_t nseq) {
  std::vector<uint32_t> idx;
  for (size_t i = 0; i < rs.resr.size(); i++) if ((rs.resr[i].status & F_SELECT) && rs.resr[i].sidx < 0) idx.push_back((uint32_t)i);
  // ascending s_start; the order of equal keys is immaterial here (each alignment is placed on its own)
  for (size_t a = 1; a < idx.size(); a++) { const uint32_t v = idx[a]; size_t b = a; while (b > 0 && rs.resr[idx[b - 1]].s_start > rs.resr[v].s_start) { idx[b] = idx[b - 1]; b--; } idx[b] = v; }
  int64_t s = 0, e;
  for (size_t i = 0; i < idx.size() && s < nseq; i++) {
    Res *rp = &rs.resr[idx[i]];
    if (rp->status & (F_NOSEQID | F_SELECT)) {
      for (; s < nseq && rp->s_start > ofp[s + 1]; s++);
      if (s >= nseq) return -1;
      for (e = s + 1; e < nseq && rp->s_end > ofp[e]; e++);
      if (rp->s_end > ofp[e]) return -1;
      if (e > s + 1) return 1;
      rp->sidx = s;
      rp->s_start -= ofp[s];
      rp->s_end -= ofp[s];
      rp->status &= ~(uint32_t)F_NOSEQID;
    }
  }
  rs.status &= ~(uint32_t)S_SWSORT;
  rs.status |= S_SEQX;
  return 0;
}

// sortAndPrune (results.c:759-837)
int sort_and_prune(Set &rs) {
  rs.sortr.clear();
  for (size_t i = 0; i < rs.resr.size(); i++) {
    Res *rp = &rs.resr[i];
    rp->serialno = (short)i;
    rp->swrank = 0;
    if (rp->status & F_SELECT) rs.sortr.push_back(rp);
  }
  rs.status |= S_SERIALNO;
  size_t nres = rs.sortr.size();
  if (nres < 2) { rs.status |= S_SWSORT; return 0; }
  qsort(rs.sortr.data(), nres, sizeof(Res *), cmpRes);
  Res **prevpp = rs.sortr.data(), **endpp = rs.sortr.data() + nres;
  nres = 1;
  for (Res **dpp = rs.sortr.data() + 1; dpp < endpp; dpp++) {
    if ((*dpp)->s_end > (*prevpp)->s_end || (*dpp)->swatscor > (*prevpp)->swatscor || (*dpp)->q_start < (*prevpp)->q_start ||
        (*dpp)->q_end > (*prevpp)->q_end || (*dpp)->sidx != (*prevpp)->sidx || (((*dpp)->status) & F_REVERSE) != (((*prevpp)->status) & F_REVERSE)) {
      if (nres == (size_t)SHRT_MAX) return -1;
      nres++;
      if ((++prevpp) < dpp) *prevpp = *dpp;
    } else (*dpp)->status &= ~(uint32_t)F_SELECT;
  }
  qsort(rs.sortr.data(), nres, sizeof(Res *), cmpResOutput);
  rs.sortr.resize(nres);
  rs.sortr[0]->swrank = 0;
  for (size_t i = 1; i < nres; i++) {
    if (rs.sortr[i]->swatscor > rs.sortr[i - 1]->swatscor) return -1;
    rs.sortr[i]->swrank = (short)(rs.sortr[i]->swatscor < rs.sortr[i - 1]->swatscor ? rs.sortr[i - 1]->swrank + 1 : rs.sortr[i - 1]->swrank);
  }
  rs.status |= S_SWSORT;
  return 0;
}

// labelComplementarySegments (results.c:707-757) + sortBySegmentAndSWscor (:669-705)
int label_segments(Set &rs) {
  const short n = (short)rs.sortr.size();
  const double min_overlap_frac = ((double)MIN_QSEGOVERLAP_PERCENT) / N100PERCENT;
  if (n < 1) return 0;
  if (n > 1 && !(rs.status & S_SWSORT)) return -1;
  for (short i = 0; i < n; i++) rs.sortr[i]->qsegx = QSEGX_INITVAL;
  short i_start = 0;
  rs.qsegno = 0;
  do {
    Res *r1p = rs.sortr[i_start];
    const uint32_t l1 = r1p->q_end - r1p->q_start;
    r1p->qsegx = (short)rs.qsegno;
    short i = (short)(i_start + 1);
    i_start = 0;
    for (; i < n; i++) {
      Res *r2p = rs.sortr[i];
      if (r2p->qsegx < 0) {
        const uint32_t l2 = r2p->q_end - r2p->q_start;
        const uint32_t min_overlap = (uint32_t)(((l1 < l2) ? l1 : l2) * min_overlap_frac);
        if (r1p->q_start + min_overlap < r2p->q_end && r2p->q_start + min_overlap < r1p->q_end) r2p->qsegx = (short)rs.qsegno;   // TEST_RESULT_OVERLAP
        else if (i_start == 0) i_start = i;
      }
    }
    if (rs.qsegno == SHRT_MAX) return -1;
    rs.qsegno++;
  } while (i_start != 0);
  rs.segsrtr = rs.sortr;
  if (n > 1) qsort(rs.segsrtr.data(), (size_t)n, sizeof(Res *), cmpResSegSW);
  rs.segnor.clear();
  rs.segnor.push_back(0);
  for (short i = 1; i < n; i++) {
    if (rs.segsrtr[i]->qsegx < rs.segsrtr[i - 1]->qsegx) return -1;
    if (rs.segsrtr[i]->qsegx > rs.segsrtr[i - 1]->qsegx) rs.segnor.push_back(i);
  }
  rs.segnor.push_back(n);
  if ((int)rs.segnor.size() != rs.qsegno + 1) return -1;
  rs.status |= S_SEGIDX;
  return 0;
}

// calcPhredScaledMappingQuality (results.c:1143-1341; the build defines results_mapscor_exp, not results_loscor_capped)
int mapping_quality(Set &rs, short qsegx, const uint8_t *qual, uint32_t qlen, const uint8_t *dstr) {
  if (!(rs.status & S_SEGIDX) || qsegx < 0 || qsegx >= rs.qsegno) return -1;
  Res **rspp = rs.segsrtr.data() + rs.segnor[(size_t)qsegx];
  const short n = (short)(rs.segnor[(size_t)qsegx + 1] - rs.segnor[(size_t)qsegx]);
  short i, i_min, n_swatscor_2nd = 0;
  int qn, swatscor_2nd, mapscor, maxmapscor, qvalsum_1st = 0, qvalsum_2nd = 0, qvalsum_ali;
  double fs, fa;
  if (n < 1) return 0;
  const int swatscor_1st = rspp[0]->swatscor;
  if (swatscor_1st < 1) { rspp[0]->mapscor = 0; return 0; }
  fs = ((double)rs.n_hits_used) / (rs.n_hits_tot + MAPSCOR_DUMMY_COUNT);
  fa = ((double)rs.n_ali_done) / (rs.n_ali_tot + MAPSCOR_DUMMY_COUNT);
  if (fs > fa) fs = fa;
  fs = (fs > MINLOGARG) ? -QUALSCOR_SCAL * log(fs) / QUALSCOR_LOGBASE : MAPSCOR_MAX;
  maxmapscor = (fs < MAPSCOR_MAX) ? MAPSCOR_MAX - (int)fs : 0;
  if (n > 1) {
    swatscor_2nd = rspp[1]->swatscor;
    for (i = 2; i < n && rspp[i]->swatscor == swatscor_2nd; i++);
    n_swatscor_2nd = (short)(i - 1);
    qn = (int)(QUALSCOR_SCAL * log((double)(n_swatscor_2nd)) / QUALSCOR_LOGBASE);
  } else { swatscor_2nd = 0; n_swatscor_2nd = 0; qn = 0; }
  if (swatscor_2nd == swatscor_1st && n > 1) {
    qsort(rspp, (size_t)n_swatscor_2nd + 1, sizeof(Res *), cmpResSegLen);
    const uint32_t seglen_1st = rspp[0]->q_end - rspp[0]->q_start;
    uint32_t seglen = rspp[1]->q_end - rspp[1]->q_start;
    if (seglen_1st == seglen) {
      if (qual) {
        if (sum_qual_mismatch(&qvalsum_1st, qual, qlen, rspp[0]->q_start, rspp[0]->q_end, dstr + rspp[0]->stroffs)) return -1;
        if (sum_qual_mismatch(&qvalsum_2nd, qual, qlen, rspp[1]->q_start, rspp[1]->q_end, dstr + rspp[1]->stroffs)) return -1;
        i_min = 1;
        for (i = 2; i < n && rspp[i]->swatscor == swatscor_1st; i++) {
          seglen = rspp[i]->q_end - rspp[i]->q_start;
          if (seglen < seglen_1st) break;
          if (sum_qual_mismatch(&qvalsum_ali, qual, qlen, rspp[i]->q_start, rspp[i]->q_end, dstr + rspp[i]->stroffs)) return -1;
          if (qvalsum_ali < qvalsum_2nd) { qvalsum_2nd = qvalsum_ali; i_min = i; }
        }
        if (qvalsum_1st > qvalsum_2nd) { Res *t = rspp[i_min]; rspp[i_min] = rspp[0]; rspp[0] = t; mapscor = MAPSCOR_MIN_UNIQ; }
        else mapscor = (qvalsum_1st == qvalsum_2nd) ? 0 : MAPSCOR_MIN_UNIQ;
      } else mapscor = 0;
    } else mapscor = MAPSCOR_MIN_UNIQ;
    if (mapscor < 1) qsort(rspp, (size_t)n_swatscor_2nd + 1, sizeof(Res *), cmpResOutput);
  } else {
    // results.c:1299-1303: exponential scaling of the score difference (double * int / uint32_t, as written there)
    mapscor = (int)(MAPSCOR_MAX * (1 - exp(((double)(swatscor_2nd - swatscor_1st)) * MAPSCOR_EXPFAC / qlen)) - qn);
    if (mapscor >= 0) mapscor += MAPSCOR_MIN_UNIQ;
    if (mapscor > maxmapscor) mapscor = maxmapscor;
  }
  if (mapscor > MAPSCOR_MAX) mapscor = MAPSCOR_MAX;
  else if (mapscor < 0) mapscor = 0;
  rspp[0]->mapscor = mapscor;
  for (i = 1; i < n; i++) rspp[i]->mapscor = 0;
  return 0;
}

// propagateMapQualAsProb (results.c:1343-1398)
int propagate_prob(Set &rs, short qsegx) {
  Res **rspp = rs.segsrtr.data() + rs.segnor[(size_t)qsegx];
  const short nn = (short)(rs.segnor[(size_t)qsegx + 1] - rs.segnor[(size_t)qsegx]);
  short i, ns, n1 = 0, n2 = 0;
  double p1 = 0.0, p2 = 0.0;
  if (nn < 1) return 0;
  for (i = 1; i < nn && rspp[i]->swatscor == rspp[0]->swatscor; i++);
  n1 = i;
  if (i < nn) { for (++i; i < nn && rspp[i]->swatscor == rspp[n1]->swatscor; i++); n2 = (short)(i - n1); }
  if (n1 == 1) {
    int isc = rspp[0]->mapscor;
    if (isc < 0) isc = 0;
    p2 = exp(((double)(-QUALSCOR_LOGBASE * isc)) / QUALSCOR_SCAL);       // float * int, then double: as written in the reference
    p1 = 1.0 - p2;
    if (n2 > 1) p2 /= n2;
  } else if (n1 > 1) { p1 = 1.0 / n1; p2 = p1; }
  for (i = 0; i < n1; i++) rspp[i]->prob = p1;
  if (n1 + n2 > SHRT_MAX) return -1;
  ns = (short)(n1 + n2);
  for (; i < ns; i++) rspp[i]->prob = p2;
  for (; i < nn; i++) rspp[i]->prob = 0.0;
  if (1 == n1 && 0 == n2) rspp[0]->status |= F_SINGLE;
  return 0;
}

// resultSetSortAndAssignSequence (results.c:2022-2064) without the split-read search (search_split == 0 on this path)
int post_one(Set &rs, const uint64_t *sop, int64_t nseq, const uint8_t *qual, uint32_t qlen, const uint8_t *dstr) {
  int rv = assign_sequence_index(rs, sop, nseq);
  if (rv) return rv;
  if (sort_and_prune(rs)) return -1;
  rs.qsegno = 0;
  if (!rs.sortr.empty()) {
    if (label_segments(rs)) return -1;
    for (short q = 0; q < rs.qsegno; q++) { if (mapping_quality(rs, q, qual, qlen, dstr)) return -1; if (propagate_prob(rs, q)) return -1; }
    rs.status |= S_MAPQ;
  }
  return 0;
}

}  // namespace

struct smaltgpu_post {
  std::vector<uint64_t> res_off, sort_off, seg_off;
  std::vector<smaltgpu_post_result> res;
  std::vector<int32_t> sortr, segsrtr, segnor, qsegno, needs_reference;
  std::vector<uint32_t> setstatus;
};

extern "C" smaltgpu_post *smaltgpu_post_create(void) { return new smaltgpu_post(); }
extern "C" void smaltgpu_post_free(smaltgpu_post *p) { delete p; }

extern "C" int smaltgpu_postprocess(smaltgpu_post *pp, const uint64_t *sop, int64_t nseq, const smaltgpu_batch_out *raw, const uint8_t *quals,
                                    const uint64_t *read_off, int nthreads, smaltgpu_post_out *out) {
  if (!pp || !sop || !raw || !read_off || !out || nseq < 1) return SMALTGPU_EARG;
  const uint32_t n = raw->nreads;
  struct PerRead { std::vector<smaltgpu_post_result> res; std::vector<int32_t> sortr, segsrtr, segnor; int qsegno = 0, needs = 0, err = 0; uint32_t status = 0; };
  std::vector<PerRead> per(n);
  auto work = [&](uint32_t lo, uint32_t hi) {
    Set rs;
    for (uint32_t r = lo; r < hi; r++) {
      PerRead &pr = per[r];
      const uint64_t a = raw->res_off[r], b = raw->res_off[r + 1];
      const smaltgpu_readstat &st = raw->stat[r];
      rs.resr.clear(); rs.sortr.clear(); rs.segsrtr.clear(); rs.segnor.clear(); rs.status = 0; rs.qsegno = 0;
      rs.n_ali_done = st.n_ali_done; rs.n_ali_tot = st.n_ali_tot; rs.n_hits_used = st.n_hits_used; rs.n_hits_tot = st.n_hits_tot;
      for (uint64_t j = a; j < b; j++) {                                   // as resultSetAddFromAli leaves them (results.c:1885-1922)
        const smaltgpu_result &x = raw->res[j];
        Res q;
        memset(&q, 0, sizeof(q));
        q.swatscor = x.swatscor; q.q_start = x.q_start; q.q_end = x.q_end; q.s_start = x.s_start; q.s_end = x.s_end; q.sidx = x.sidx;
        q.status = F_SELECT | ((x.reverse & SMALTGPU_RES_REVERSE) ? F_REVERSE : 0u) | (x.sidx < 0 ? F_NOSEQID : 0u);
        q.stroffs = x.stroffs; q.strlen = x.strlen; q.rsltx = RSLTX_INITVAL; q.qsegx = QSEGX_INITVAL;
        rs.resr.push_back(q);
      }
      const uint32_t qlen = (uint32_t)(read_off[r + 1] - read_off[r]);
      int rv = 0;
      // mapSingleRead sorts only when the score pass found something (rmap.c:1376); an unmapped read keeps a blank set
      if (st.max1scor >= 1 && !st.errcode) rv = post_one(rs, sop, nseq, quals ? quals + read_off[r] : nullptr, qlen, raw->diffstr);
      if (rv > 0) pr.needs = 1; else if (rv < 0) pr.err = 1;
      pr.status = rs.status; pr.qsegno = rs.qsegno;
      for (const Res &q : rs.resr) {
        smaltgpu_post_result o;
        memset(&o, 0, sizeof(o));
        o.swatscor = q.swatscor; o.q_start = q.q_start; o.q_end = q.q_end; o.s_start = q.s_start; o.s_end = q.s_end; o.sidx = (int32_t)q.sidx;
        o.status = q.status; o.mapscor = q.mapscor; o.prob = q.prob; o.rsltx = q.rsltx; o.qsegx = q.qsegx; o.swrank = q.swrank; o.stroffs = q.stroffs; o.strlen = q.strlen;
        pr.res.push_back(o);
      }
      if (!pr.needs && !pr.err) {
        for (Res *p : rs.sortr) pr.sortr.push_back((int32_t)(p - rs.resr.data()));
        if (rs.status & S_SEGIDX) { for (Res *p : rs.segsrtr) pr.segsrtr.push_back((int32_t)(p - rs.resr.data())); for (int v : rs.segnor) pr.segnor.push_back(v); }
      }
    }
  };
  if (nthreads < 1) nthreads = 1;
  if ((uint32_t)nthreads > n / 256 + 1) nthreads = (int)(n / 256 + 1);
  if (nthreads == 1) work(0, n);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(work, (uint32_t)((uint64_t)n * t / nthreads), (uint32_t)((uint64_t)n * (t + 1) / nthreads));
    for (std::thread &t : th) t.join();
  }
  smaltgpu_post &P = *pp;
  P.res_off.assign((size_t)n + 1, 0); P.sort_off.assign((size_t)n + 1, 0); P.seg_off.assign((size_t)n + 1, 0);
  P.res.clear(); P.sortr.clear(); P.segsrtr.clear(); P.segnor.clear(); P.qsegno.assign(n ? n : 1, 0); P.needs_reference.assign(n ? n : 1, 0); P.setstatus.assign(n ? n : 1, 0);
  int nerr = 0;
  for (uint32_t r = 0; r < n; r++) {
    const PerRead &pr = per[r];
    P.res_off[r] = P.res.size(); P.sort_off[r] = P.sortr.size(); P.seg_off[r] = P.segnor.size();
    P.res.insert(P.res.end(), pr.res.begin(), pr.res.end());
    P.sortr.insert(P.sortr.end(), pr.sortr.begin(), pr.sortr.end());
    P.segsrtr.insert(P.segsrtr.end(), pr.segsrtr.begin(), pr.segsrtr.end());
    P.segsrtr.resize(P.sortr.size(), -1);                                  // parallel to sortr (empty when the set has no segment index)
    P.segnor.insert(P.segnor.end(), pr.segnor.begin(), pr.segnor.end());
    P.qsegno[r] = pr.qsegno; P.needs_reference[r] = pr.needs; P.setstatus[r] = pr.status;
    nerr += pr.err;
  }
  P.res_off[n] = P.res.size(); P.sort_off[n] = P.sortr.size(); P.seg_off[n] = P.segnor.size();
  if (P.res.empty()) P.res.resize(1);
  if (P.sortr.empty()) { P.sortr.resize(1); P.segsrtr.resize(1); }
  if (P.segnor.empty()) P.segnor.resize(1);
  out->nreads = n; out->res_off = P.res_off.data(); out->res = P.res.data(); out->diffstr = raw->diffstr; out->sort_off = P.sort_off.data();
  out->sortr = P.sortr.data(); out->segsrtr = P.segsrtr.data(); out->seg_off = P.seg_off.data(); out->segnor = P.segnor.data();
  out->qsegno = P.qsegno.data(); out->setstatus = P.setstatus.data(); out->needs_reference = P.needs_reference.data();
  return nerr ? SMALTGPU_EINTERNAL : SMALTGPU_OK;
}
