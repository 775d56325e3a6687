/* oracle/refdump/rd_results.c -- TEST INFRASTRUCTURE (golden-vector generator).
 * #includes the reference's results.c text at build time to read the raw Result array. */
#define resultSetSortAndAssignSequence resultSetSortAndAssignSequence_impl      /* intercepted below (-p) */
#include "results.c"
#undef resultSetSortAndAssignSequence
#include <stdio.h>

/* -p: the alignments as they are BEFORE the post-processing (in concatenated mode assignSequenceIndex rewrites coordinates
 * and appends the fragments of alignments that span sequences), printed from a wrapper around the reference's routine */
int rdPostDump = 0;
int resultSetSortAndAssignSequence(ResultSet *rsp, SeqFastq *sbufp, BOOL search_split, const SeqFastq *sqp,
				   const ScoreProfile *scpp, const ScoreProfile *scpRCp, const SeqSet *ssp, const SeqCodec *codecp)
{
  if (rdPostDump) {
    size_t i, n = ARRLEN(rsp->resr);
    int j;
    for (i=0; i<n; i++) {
      const Result *rp = rsp->resr + i;
      printf("RW %u %c %d %u %u %llu %llu %lld ", (unsigned) i, (rp->status & RSLTFLAG_REVERSE)? 'R':'F', rp->swatscor, rp->q_start, rp->q_end,
	     (unsigned long long) rp->s_start, (unsigned long long) rp->s_end, (long long) rp->sidx);
      for (j=0; j<rp->strlen; j++) printf("%02x", (unsigned) rsp->diffstrp->dstrp[rp->stroffs + j]);
      putchar('\n');
    }
  }
  return resultSetSortAndAssignSequence_impl(rsp, sbufp, search_split, sqp, scpp, scpRCp, ssp, codecp);
}

void rdDumpResults(FILE *fp, const ResultSet *rsp)
{
  size_t i, n = ARRLEN(rsp->resr);
  int j;
  for (i=0; i<n; i++) {
    const Result *rp = rsp->resr + i;
    fprintf(fp, "RS %u %c %d %u %u %llu %llu %lld ", (unsigned) i,
	    (rp->status & RSLTFLAG_REVERSE)? 'R':'F', rp->swatscor, rp->q_start, rp->q_end,
	    (unsigned long long) rp->s_start, (unsigned long long) rp->s_end, (long long) rp->sidx);
    for (j=0; j<rp->strlen; j++)
      fprintf(fp, "%02x", (unsigned) rsp->diffstrp->dstrp[rp->stroffs + j]);
    fputc('\n', fp);
  }
  fprintf(fp, "RX %u %d %d %d %d %u %u\n", (unsigned) n, rsp->swatscor_max, rsp->swatscor_2ndmax,
	  rsp->n_ali_done, rsp->n_ali_tot, rsp->n_hits_used, rsp->n_hits_tot);
}

/* paired mode (rd_rmap.c): the results one mapSingleRead call added, and the running score maxima */
unsigned rdResultNum(const ResultSet *rsp) { return (unsigned) ARRLEN(rsp->resr); }
void rdScoreMaxGet(const ResultSet *rsp, int *mx, int *mx2) { *mx = rsp->swatscor_max; *mx2 = rsp->swatscor_2ndmax; }
void rdScoreMaxSet(ResultSet *rsp, int mx, int mx2) { rsp->swatscor_max = mx; rsp->swatscor_2ndmax = mx2; }
void rdScoreMaxUpdate(ResultSet *rsp, int scor) { UPDATE_SWATSCORMAX(rsp, scor); }

static void printResult(FILE *fp, const char *tag, unsigned i, const Result *rp, const ResultSet *rsp)
{
  int j;
  fprintf(fp, "%s %u %c %d %u %u %llu %llu %lld ", tag, i, (rp->status & RSLTFLAG_REVERSE)? 'R':'F', rp->swatscor, rp->q_start, rp->q_end,
	  (unsigned long long) rp->s_start, (unsigned long long) rp->s_end, (long long) rp->sidx);
  for (j=0; j<rp->strlen; j++) fprintf(fp, "%02x", (unsigned) rsp->diffstrp->dstrp[rp->stroffs + j]);
  fputc('\n', fp);
}

void rdDumpLastResult(FILE *fp, const ResultSet *rsp)
{
  const size_t n = ARRLEN(rsp->resr);
  if (n > 0) printResult(fp, "PL", (unsigned) (n - 1), rsp->resr + n - 1, rsp);
}

void rdDumpResultsFrom(FILE *fp, const ResultSet *rsp, unsigned first, int swmax, int sw2nd)
{
  size_t i, n = ARRLEN(rsp->resr);
  for (i=first; i<n; i++) printResult(fp, "RS", (unsigned) (i - first), rsp->resr + i, rsp);
  fprintf(fp, "RX %u %d %d %d %d %u %u\n", (unsigned) ((n > first)? n - first: 0), swmax, sw2nd,
	  rsp->n_ali_done, rsp->n_ali_tot, rsp->n_hits_used, rsp->n_hits_tot);
}

/* -p: the state resultSetSortAndAssignSequence (results.c:2022: assignSequenceIndex :1695, sortAndPrune :759,
 * labelComplementarySegments :707, calcPhredScaledMappingQualityPerQuerySegment) leaves behind.
 * The output filters (resultSetFilterResults) and the report set further status bits afterwards; they are masked out. */
void rdDumpPost(FILE *fp, const ResultSet *rsp)
{
  size_t i, n = ARRLEN(rsp->resr), ns = ARRLEN(rsp->sortr);
  const unsigned mask = ~(unsigned) (RSLTFLAG_NOOUTPUT | RSLTFLAG_BELOWRELSW | RSLTFLAG_REPORTED);
  fprintf(fp, "PS %u %u %d %u\n", (unsigned) n, (unsigned) ns, (int) rsp->qsegno, (unsigned) (rsp->status));
  for (i=0; i<n; i++) {
    const Result *rp = rsp->resr + i;
    int j;
    fprintf(fp, "RF %u %u %d %d %.17g %u %u %llu %llu %lld %d %d %d ", (unsigned) i, (unsigned) (rp->status & mask), rp->swatscor, rp->mapscor, rp->prob,
	    rp->q_start, rp->q_end, (unsigned long long) rp->s_start, (unsigned long long) rp->s_end, (long long) rp->sidx,
	    (int) rp->rsltx, (int) rp->qsegx, (int) rp->swrank);
    for (j=0; j<rp->strlen; j++) fprintf(fp, "%02x", (unsigned) rsp->diffstrp->dstrp[rp->stroffs + j]);
    fputc('\n', fp);
  }
  fprintf(fp, "SO");
  for (i=0; i<ns; i++) fprintf(fp, " %d", (int) (rsp->sortr[i] - rsp->resr));
  fputc('\n', fp);
  if (ns > 0 && (rsp->status & RSLTSETFLG_SEGIDX)) {
    fprintf(fp, "SS");
    for (i=0; i<ARRLEN(rsp->segsrtr); i++) fprintf(fp, " %d", (int) (rsp->segsrtr[i] - rsp->resr));
    fputc('\n', fp);
    fprintf(fp, "SG");
    for (i=0; i<ARRLEN(rsp->segnor); i++) fprintf(fp, " %d", (int) rsp->segnor[i]);
    fputc('\n', fp);
  }
}
