/* oracle/refdump/rd_results.c -- TEST INFRASTRUCTURE (golden-vector generator).
 * #includes the reference's results.c text at build time to read the raw Result array. */
#include "results.c"
#include <stdio.h>

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
