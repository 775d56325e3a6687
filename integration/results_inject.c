/* results_inject.c -- OUR translation unit around the reference's results.c (its text is taken from the reference
 * tree at build time by oracle/Makefile, target ref_gpu; nothing is copied): adds one function that appends raw
 * alignments, in the state resultSetAddFromAli (results.c:1852) leaves them in, to a ResultSet.  Everything else in
 * the object is the reference's unmodified results.c. */
#include "results.c"
#include "smaltgpu.h"

int resultSetInjectRaw(ResultSet *rsp, unsigned n, const smaltgpu_result *res, const unsigned char *dstr,
                       int swatscor_max, int swatscor_2ndmax)
{
  unsigned i;
  int errcode;
  if (n < 1) return ERRCODE_SUCCESS;
  rsp->status = 0;
  for (i = 0; i < n; i++) {
    Result *rp;
    DiffStr view;
    ARRNEXTP(rp, rsp->resr);
    if (!rp) return ERRCODE_NOMEM;
    BLANK_RESULT(rp);
    rp->swatscor = res[i].swatscor;
    rp->q_start = res[i].q_start; rp->q_end = res[i].q_end;
    rp->s_start = (SEQLEN_t)res[i].s_start; rp->s_end = (SEQLEN_t)res[i].s_end;
    rp->sidx = res[i].sidx;
    rp->swrank = 0;
    rp->status = RSLTFLAG_SELECT;
    if (res[i].sidx == RESULTSET_UNKNOWN_SEQIDX) rp->status |= RSLTFLAG_NOSEQID;
    if (res[i].reverse & SMALTGPU_RES_REVERSE) rp->status |= RSLTFLAG_REVERSE;
    rp->stroffs = DIFFSTR_LENGTH(rsp->diffstrp);
    rp->strlen = (int)res[i].strlen;
    memset(&view, 0, sizeof(view));
    view.dstrp = (DIFFSTR_T *)(dstr + res[i].stroffs);
    view.len = (int)res[i].strlen;
    if ((errcode = diffStrAppend(rsp->diffstrp, &view))) return errcode;
    rp->mapscor = 0;
    rp->rsltx = RSLTX_INITVAL;
    rp->qsegx = QSEGX_INITVAL;
  }
  rsp->swatscor_max = swatscor_max;          /* include what UPDATE_SWATSCORMAX saw of results that were popped again */
  rsp->swatscor_2ndmax = swatscor_2ndmax;
  return ERRCODE_SUCCESS;
}

/* A mapSingleRead call of rmapPair that APPENDS to a ResultSet (rmap.c:1976-1989, :2019-2039).  `res` are the alignments
 * of the call as the library returns them for a fresh set, `swatscor_max/2ndmax` the set's running maxima after the call
 * (the library was given the maxima before it, smaltgpu_callctx.prev_max).  resultSetAddFromAli (results.c:1852-1942)
 * compares the first alignment of a candidate with the set's last one: if it repeats it, it is popped, and what the same
 * call writes after a popped slot is lost (the array length is not advanced again) -- SMALTGPU_RES_CANDFIRST marks where the
 * next candidate's alignments begin.  Within the call itself the library has applied the same rule already. */
int resultSetAppendRaw(ResultSet *rsp, unsigned n, const smaltgpu_result *res, const unsigned char *dstr,
                       int swatscor_max, int swatscor_2ndmax)
{
  unsigned first = 0;
  const size_t nold = ARRLEN(rsp->resr);
  if (n > 0 && nold > 0) {
    const Result *lp = rsp->resr + nold - 1;
    if ((SEQLEN_t)res[0].s_start == lp->s_start && (SEQLEN_t)res[0].s_end == lp->s_end && res[0].q_start == lp->q_start &&
        res[0].q_end == lp->q_end && res[0].swatscor == lp->swatscor && res[0].sidx == lp->sidx)        /* isIdenticalResult, results.c:556 */
      for (first = 1; first < n && !(res[first].reverse & SMALTGPU_RES_CANDFIRST); first++);
  }
  if (first < n) {
    const int errcode = resultSetInjectRaw(rsp, n - first, res + first, dstr, swatscor_max, swatscor_2ndmax);
    if (errcode) return errcode;
  } else if (n > 0) rsp->status = 0;
  rsp->swatscor_max = swatscor_max;          /* also what alignments that were popped again raised (results.c:1918) */
  rsp->swatscor_2ndmax = swatscor_2ndmax;
  return ERRCODE_SUCCESS;
}

/* The state resultSetSortAndAssignSequence (results.c:2022) leaves in a ResultSet, taken from the library's post-processing
 * (smaltgpu_postprocess) instead of being computed here: `pr` are the set's n alignments in array order (the
 * raw alignments were injected before with resultSetInjectRaw), sortr/segsrtr index into them. */
int resultSetInjectPost(ResultSet *rsp, unsigned n, const smaltgpu_post_result *pr, const unsigned char *pdstr, unsigned nsort, const int32_t *sortr,
                        const int32_t *segsrtr, unsigned nsegnor, const int32_t *segnor, int qsegno, unsigned setstatus)
{
  unsigned i;
  const unsigned nraw = (unsigned) ARRLEN(rsp->resr);
  if (n < nraw) return ERRCODE_ASSERT;
  for (i = nraw; i < n; i++) {             /* fragments of alignments that were cut at sequence junctions (splitMultiSpan, results.c:1570-1600) */
    Result *hp;
    DiffStr view;
    int errcode;
    ARRNEXTP(hp, rsp->resr);
    if (!hp) return ERRCODE_NOMEM;
    BLANK_RESULT(hp);
    hp->swatscor = pr[i].swatscor;
    hp->q_start = pr[i].q_start; hp->q_end = pr[i].q_end;
    hp->stroffs = DIFFSTR_LENGTH(rsp->diffstrp);
    hp->strlen = (int) pr[i].strlen;
    memset(&view, 0, sizeof(view));
    view.dstrp = (DIFFSTR_T *) (pdstr + pr[i].stroffs);
    view.len = (int) pr[i].strlen;
    if ((errcode = diffStrAppend(rsp->diffstrp, &view))) return errcode;
  }
  for (i = 0; i < n; i++) {
    Result *rp = rsp->resr + i;
    rp->serialno = (short) i;
    rp->status = (RSLTFLG_t) pr[i].status;
    rp->mapscor = pr[i].mapscor;
    rp->prob = pr[i].prob;
    rp->s_start = (SETSIZ_t) pr[i].s_start; rp->s_end = (SETSIZ_t) pr[i].s_end;
    rp->sidx = pr[i].sidx;
    rp->rsltx = pr[i].rsltx; rp->qsegx = pr[i].qsegx; rp->swrank = pr[i].swrank;
  }
  if (nsort > ARRNALLOC(rsp->sortr)) { void *hp = ARREALLOC(rsp->sortr, nsort); if (!hp) return ERRCODE_NOMEM; rsp->sortr = hp; }
  for (i = 0; i < nsort; i++) rsp->sortr[i] = rsp->resr + sortr[i];
  ARRLEN(rsp->sortr) = nsort;
  rsp->qsegno = (short) qsegno;
  if (nsegnor > 0) {
    if (nsort > ARRNALLOC(rsp->segsrtr)) { void *hp = ARREALLOC(rsp->segsrtr, nsort); if (!hp) return ERRCODE_NOMEM; rsp->segsrtr = hp; }
    if (nsegnor > ARRNALLOC(rsp->segnor)) { void *hp = ARREALLOC(rsp->segnor, nsegnor); if (!hp) return ERRCODE_NOMEM; rsp->segnor = hp; }
    for (i = 0; i < nsort; i++) rsp->segsrtr[i] = rsp->resr + segsrtr[i];
    ARRLEN(rsp->segsrtr) = nsort;
    for (i = 0; i < nsegnor; i++) rsp->segnor[i] = (short) segnor[i];
    ARRLEN(rsp->segnor) = nsegnor;
  }
  rsp->status = (uint8_t) setstatus;
  return ERRCODE_SUCCESS;
}
