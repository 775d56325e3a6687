/* oracle/refdump/rd_rmap.c -- TEST INFRASTRUCTURE (golden-vector generator).
 *
 * Driver of OUR OWN writing that runs the unmodified reference mapper read by read and
 * prints per-stage state in the line format documented in oracle/DUMPFORMAT.md.  It
 * #includes the reference's rmap.c text at build time (from $(REF)/src, never copied)
 * so that the private RMap buffers can be read after rmapSingle() returns.
 *
 * usage: refdump [-m minscor] [-d scordiff] [-c mincover] [-q minbasq] [-H ncut]
 *                [-S match,mismatch,gapopen,gapext] [-x] <index_prefix> <reads.fq>
 */
#include "rmap.c"
#include <stdio.h>
#include <string.h>
#include <getopt.h>

extern void rdDumpHitInfo(FILE *fp, char strand, const HashHitInfo *hip);
extern void rdDumpCands(FILE *fp, const SegAliCands *sacp);
extern void rdDumpResults(FILE *fp, const ResultSet *rsp);

static void dumpHitLists(FILE *fp, RMap *rmp, int ncut, BOOL with_seqidx,
			 const HashTable *htp, const SeqSet *ssp)
{
  int st;
  const SETSIZ_t *soffsp;
  const SEQNUM_t nseq = seqSetGetOffsets(ssp, &soffsp);
  for (st=0; st<2; st++) {
    const HashHitInfo *hip = (st)? rmp->mrp->hhiRp: rmp->mrp->hhiFp;
    SEQNUM_t s, ns = (with_seqidx)? nseq: 1;
    for (s=0; s<ns; s++) {
      int i, nhits;
      const uint64_t *dat;
      hashBlankHitList(rmp->bfp->hhlp);
      if (with_seqidx)
	hashCollectHitsForSegment(rmp->bfp->hhlp, soffsp[s], soffsp[s+1], (HASHNUM_t) ncut, 1, hip, htp, NULL);
      else
	hashCollectHitsUsingCutoff(rmp->bfp->hhlp, (HASHNUM_t) ncut, htp, hip);
      dat = hashGetHitListData(&nhits, NULL, NULL, NULL, NULL, NULL, rmp->bfp->hhlp);
      fprintf(fp, "HL %c %lld %d", (st)? 'R':'F', (long long) s, nhits);
      for (i=0; i<nhits; i++) fprintf(fp, " %llx", (unsigned long long) dat[i]);
      fputc('\n', fp);
    }
  }
}

int main(int argc, char *argv[])
{
  int errcode = 0, c;
  int minscor = -1, scordiff = 0, ncut = 10000, minbasq = 0;
  int with_hitlists = 1;
  double mincover = 0.0;
  int pm = 1, pmm = -2, pgo = -4, pge = -3, have_pen = 0;
  RMAPFLG_t rmapflg = 0;
  SeqCodec *codecp;
  SeqSet *ssp;
  HashTable *htp;
  ScorePenalties *penp;
  ScoreMatrix *smp;
  ResultFilter *rfp;
  SeqIO *sfp;
  SeqFastq *readp;
  RMap *rmp;
  ErrMsg *errmsgp = 0;
  UCHAR ktup, nskip;
  unsigned long long readno = 0;

  while ((c = getopt(argc, argv, "m:d:c:q:H:S:xn")) != -1) {
    switch (c) {
    case 'm': minscor = atoi(optarg); break;
    case 'd': scordiff = atoi(optarg); break;
    case 'c': mincover = atof(optarg); break;
    case 'q': minbasq = atoi(optarg); break;
    case 'H': ncut = atoi(optarg); break;
    case 'S': sscanf(optarg, "%d,%d,%d,%d", &pm, &pmm, &pgo, &pge); have_pen = 1; break;
    case 'x': rmapflg |= RMAPFLG_NOSHRTINFO | RMAPFLG_SENSITIVE; break;
    case 'n': with_hitlists = 0; break;
    default: return 2;
    }
  }
  if (argc - optind < 2) { fprintf(stderr, "usage: refdump [opts] index reads.fq\n"); return 2; }
  ERRMSG_CREATE(errmsgp);
  codecp = seqCodecCreate();
  ssp = seqSetReadBinFil(&errcode, argv[optind]);
  if (errcode) { fprintf(stderr, "cannot read .sma (%d)\n", errcode); return 1; }
  htp = hashTableRead(&errcode, argv[optind]);
  if (errcode) { fprintf(stderr, "cannot read .smi (%d)\n", errcode); return 1; }
  ktup = hashTableGetKtupLen(htp, &nskip);
  if (minscor < 0) minscor = ktup + nskip - 1;
  if (seqSetGetOffsets(ssp, NULL) < 512) rmapflg |= RMAPFLG_SEQBYSEQ;
  if (!scordiff) rmapflg |= RMAPFLG_BEST;
  penp = scorePenaltiesCreate();
  if (have_pen) {
    scoreSetPenalty(penp, SCORPNLTYP_MATCH, (short) pm);
    scoreSetPenalty(penp, SCORPNLTYP_MISMATCH, (short) pmm);
    scoreSetPenalty(penp, SCORPNLTYP_GAPOPEN, (short) pgo);
    scoreSetPenalty(penp, SCORPNLTYP_GAPEXT, (short) pge);
  }
  smp = scoreCreateMatrix(codecp, penp);
  rfp = resultSetCreateFilter();
  resultSetFilterData(rfp, minscor, scordiff, 0.0);
  rmp = rmapCreate(htp, codecp, ssp, smp, rmapflg);
  readp = seqFastqCreate(0, SEQTYP_FASTQ);
  sfp = seqIOopen(&errcode, argv[optind+1], SEQIO_READ, 0);
  if (errcode) { fprintf(stderr, "cannot open reads (%d)\n", errcode); return 1; }

  while (!seqIOstatus(sfp)) {
    uint32_t readlen, covermin, i, n;
    if ((errcode = seqFastqRead(readp, sfp))) break;
    seqFastqEncode(readp, codecp);
    seqFastqGetConstSequence(readp, &readlen, NULL);
    if (mincover < 1.01) {
      covermin = (uint32_t) (mincover*readlen);
      if (covermin > readlen) covermin = readlen;
    } else covermin = (uint32_t) mincover;
    errcode = rmapSingle(errmsgp, rmp, readp, ncut, covermin, minscor, scordiff,
			 (UCHAR) minbasq, 512, 2048, rmapflg, smp, rfp, htp, ssp, codecp);
    printf("READ %llu %s len=%u err=%d\n", readno++, seqFastqGetSeqName(readp), readlen, errcode);
    if (readlen >= ktup) {
      rdDumpHitInfo(stdout, 'F', rmp->mrp->hhiFp);
      rdDumpHitInfo(stdout, 'R', rmp->mrp->hhiRp);
      rdDumpCands(stdout, rmp->bfp->sacp);
      n = ARRLEN(rmp->bfp->candr);
      for (i=0; i<n; i++) {
	const RMAPCAND *cp = rmp->bfp->candr + i;
	printf("RC %u %u %u %u %llu %llu %d %d %lld %d\n", i, (unsigned) cp->flags, cp->qs, cp->qe,
	       (unsigned long long) cp->rs, (unsigned long long) cp->re, cp->band_l, cp->band_r,
	       (long long) cp->sqidx, cp->swscor);
      }
    }
    rdDumpResults(stdout, rmp->rsrp);
    if (with_hitlists && readlen >= ktup)
      dumpHitLists(stdout, rmp, ncut, (BOOL) ((rmapflg & RMAPFLG_SEQBYSEQ) != 0), htp, ssp);
  }
  seqIOclose(sfp);
  return 0;
}
