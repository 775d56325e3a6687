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

/* A mapSingleRead call of rmapPair that APPENDS to a ResultSet (rmap.c:1976-1989, :2019-2039).  `res` are ALL alignments of
 * the call, candidate by candidate (smaltgpu_callctx.raw_alignments; SMALTGPU_RES_CANDFIRST marks the first of each),
 * `swatscor_max/2ndmax` the set's running maxima after the call (the library was given the maxima before it,
 * smaltgpu_callctx.prev_max).  They go into the set the way resultSetAddFromAli (results.c:1852-1942) puts a candidate's
 * alignments there, on the set's own array, so that its slots -- also the ones outside the array's length -- hold what they
 * hold in the reference: a candidate opens the next slot; an alignment equal to the slot before it gives the slot back and
 * stays open, anything else is completed and the next alignment opens the slot behind the array's end (after a repeat: the
 * same slot again, so the alignment behind a repeat is lost). */
int resultSetAppendRaw(ResultSet *rsp, unsigned n, const smaltgpu_result *res, const unsigned char *dstr,
                       int swatscor_max, int swatscor_2ndmax)
{
  unsigned i = 0;
  while (i < n) {
    unsigned t, j = i + 1;
    int stays = 0, errcode;
    Result *slot;
    while (j < n && !(res[j].reverse & SMALTGPU_RES_CANDFIRST)) j++;
    ARRNEXTP(slot, rsp->resr);
    if (!slot) return ERRCODE_NOMEM;
    BLANK_RESULT(slot);
    slot->status = 0;
    rsp->status = 0;
    for (t = i; t < j; t++) {
      const smaltgpu_result *a = res + t;
      if (stays) {
        ARRNEXTP(slot, rsp->resr);
        if (!slot) return ERRCODE_NOMEM;
        slot->status = 0;
      }
      slot->swatscor = a->swatscor;
      slot->q_start = a->q_start; slot->q_end = a->q_end;
      slot->s_start = (SEQLEN_t)a->s_start; slot->s_end = (SEQLEN_t)a->s_end;
      slot->sidx = a->sidx;
      slot->swrank = 0;
      if (a->sidx == RESULTSET_UNKNOWN_SEQIDX) slot->status |= RSLTFLAG_NOSEQID;
      stays = ARRLEN(rsp->resr) < 2 || !isIdenticalResult(slot, slot - 1);
      if (stays) {
        DiffStr view;
        slot->stroffs = DIFFSTR_LENGTH(rsp->diffstrp);
        slot->strlen = (int)a->strlen;
        memset(&view, 0, sizeof(view));
        view.dstrp = (DIFFSTR_T *)(dstr + a->stroffs);
        view.len = (int)a->strlen;
        if ((errcode = diffStrAppend(rsp->diffstrp, &view))) return errcode;
        slot->status |= RSLTFLAG_SELECT;
        if (a->reverse & SMALTGPU_RES_REVERSE) slot->status |= RSLTFLAG_REVERSE;
        slot->mapscor = 0;
        slot->rsltx = RSLTX_INITVAL;
        slot->qsegx = QSEGX_INITVAL;
      } else if (--ARRLEN(rsp->resr) == 0) rsp->status = 0;
    }
    i = j;
  }
  rsp->swatscor_max = swatscor_max;          /* also what alignments that were lost again raised (results.c:1918) */
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
