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
