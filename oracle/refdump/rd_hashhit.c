/* oracle/refdump/rd_hashhit.c -- TEST INFRASTRUCTURE (golden-vector generator).
 * Our own dump routine; it #includes the reference's hashhit.c text at build time
 * (found via -I$(REF)/src, never copied) to read its private HashHitInfo struct. */
#include "hashhit.c"
#include <stdio.h>

void rdDumpHitInfo(FILE *fp, char strand, const HashHitInfo *hip)
{
  uint32_t i;
  fprintf(fp, "HI %c nseeds=%u rank=%u status=%u\n", strand, hip->n_seeds, hip->seed_rank,
	  (unsigned) (hip->status));
  fprintf(fp, "QM %c ", strand);
  for (i=0; i<hip->qlen; i++) fputc('0' + hip->qmaskp[i], fp);
  fputc('\n', fp);
  for (i=0; i<hip->n_seeds; i++) {
    const SEED *sp = hip->seedp + hip->sidxp[i];
    fprintf(fp, "SD %c %u %u %u %u\n", strand, i, sp->qoffs, sp->nhits, sp->posidx);
  }
}
