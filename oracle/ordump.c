/* oracle/ordump.c -- TEST INFRASTRUCTURE: command-line twin of oracle/refdump for the CPU
 * restatement; prints the same per-stage line format for a 4-line-record FASTQ file.
 * usage: ordump [-m minscor] [-d scordiff] [-c mincover] [-q minbasq] [-H ncut] [-x] [-n] <index_prefix> <reads.fq> */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "smalt_oracle.h"

int main(int argc, char **argv)
{
  OrIndex *ix;
  OrMap *m;
  OrParams p;
  int c, minscor = -1, scordiff = 0, ncut = 10000, minbasq = 0, xflag = 0, with_hl = 1;
  int pm = 1, pmm = -2, pgo = -4, pge = -3;
  double mincover = 0.0;
  char *name = NULL, *seq = NULL, *plus = NULL, *qual = NULL;
  size_t cn = 0, cs = 0, cp = 0, cq = 0;
  unsigned long long readno = 0;
  FILE *fp;
  while ((c = getopt(argc, argv, "m:d:c:q:H:S:xn")) != -1) {
    switch (c) {
    case 'm': minscor = atoi(optarg); break;
    case 'd': scordiff = atoi(optarg); break;
    case 'c': mincover = atof(optarg); break;
    case 'q': minbasq = atoi(optarg); break;
    case 'H': ncut = atoi(optarg); break;
    case 'S': sscanf(optarg, "%d,%d,%d,%d", &pm, &pmm, &pgo, &pge); break;
    case 'x': xflag = 1; break;
    case 'n': with_hl = 0; break;
    default: return 2;
    }
  }
  if (argc - optind < 2) { fprintf(stderr, "usage: ordump [opts] index reads.fq\n"); return 2; }
  if (!(ix = or_index_read(argv[optind]))) { fprintf(stderr, "cannot read index\n"); return 1; }
  if (!(fp = fopen(argv[optind+1], "r"))) { fprintf(stderr, "cannot open reads\n"); return 1; }
  m = or_map_create(ix);
  or_params_default(&p, ix);
  if (minscor >= 0) p.min_swatscor = minscor;
  p.min_swatscor_below_max = scordiff;
  if (scordiff) p.flags &= ~OR_FLG_BEST;
  if (xflag) p.flags |= OR_FLG_NOSHRTINFO | OR_FLG_SENSITIVE;
  p.ncut = ncut; p.min_basq = minbasq;
  p.match = pm; p.mismatch = pmm; p.gap_init = pgo; p.gap_ext = pge;
  while (getline(&name, &cn, fp) > 0 && getline(&seq, &cs, fp) > 0 && getline(&plus, &cp, fp) > 0 && getline(&qual, &cq, fp) > 0) {
    uint32_t len = (uint32_t) strcspn(seq, "\r\n");
    name[strcspn(name, " \t\r\n")] = 0;
    if (mincover < 1.01) { p.min_cover = (uint32_t) (mincover*len); if (p.min_cover > len) p.min_cover = len; }
    else p.min_cover = (uint32_t) mincover;
    or_map_single(m, seq, qual, len, &p);
    or_map_dump(m, stdout, readno++, name + 1, with_hl);
  }
  fclose(fp);
  or_map_free(m);
  or_index_free(ix);
  return 0;
}
