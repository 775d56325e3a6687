/* oracle/refdump/rd_segment.c -- TEST INFRASTRUCTURE (golden-vector generator).
 * #includes the reference's segment.c text at build time to read SegAliCands. */
#include "segment.c"
#include <stdio.h>

void rdDumpCands(FILE *fp, const SegAliCands *sacp)
{
  uint32_t i, n = ARRLEN(sacp->candr);
  for (i=0; i<n; i++) {
    const SEGCAND *c = sacp->candr + i;
    fprintf(fp, "CA %u %u %u %u %u %d %d %d %u %u %d %d\n", i, c->qs, c->qe, c->rs, c->re,
	    (int) c->shiftoffs, (int) c->srange, (int) c->shift2mm, c->cover, (unsigned) c->flag,
	    c->nseg, c->seqidx);
  }
  fprintf(fp, "ST %u %u %u %u %u %u %u\n", sacp->max_cover, sacp->max2nd_cover, n,
	  sacp->n_mincover, sacp->n_sort, sacp->cover_deficit[0], sacp->cover_deficit[1]);
  for (i=0; i<sacp->n_sort; i++)
    fprintf(fp, "SI %u %u %u\n", i, sacp->sort_idx[i], sacp->sort_keys[i]);
}
