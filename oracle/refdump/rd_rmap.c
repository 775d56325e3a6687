/* oracle/refdump/rd_rmap.c -- TEST INFRASTRUCTURE (golden-vector generator).
 *
 * Driver of OUR OWN writing that runs the unmodified reference mapper read by read and
 * prints per-stage state in the line format documented in oracle/DUMPFORMAT.md.  It
 * #includes the reference's rmap.c text at build time (from $(REF)/src, never copied)
 * so that the private RMap buffers can be read after rmapSingle() returns.
 *
 * usage: refdump [-m minscor] [-d scordiff] [-c mincover] [-q minbasq] [-H ncut]
 *                [-S match,mismatch,gapopen,gapext] [-x] <index_prefix> <reads.fq>
 *        refdump -P <mates.fq> [-i maxins] [-j minins] [-l pe|mp|pp] ... <index_prefix> <reads.fq>
 *
 * Paired mode (-P) runs the reference's rmapPair (rmap.c:1744) pair by pair and prints one block per mapSingleRead call
 * it makes (rare mate, interval-restricted mate, unrestricted re-map, restricted re-map over the on-the-fly k=5 index):
 * the call's arguments (`MS`, `IV` lines; DUMPFORMAT.md) followed by the stage dump of that call.  The calls are
 * intercepted without touching the reference's text: every call site in rmap.c passes `errmsgp` first while the
 * definition's first parameter is `ErrMsg *errmsgp`, so a function-like macro that pastes its first token sends the
 * definition to mapSingleRead_impl and the calls to our hook.
 */
#include <stdio.h>
#include <string.h>
#include <getopt.h>
#include "rmap.h"
#include "interval.h"
typedef struct RMAPBUFF_ RMAPBUFF;
typedef struct RMAPINFO_ RMAPINFO;
typedef struct RMAPPROF_ RMAPPROF;
static int rd_msr_hook(ErrMsg *errmsgp, RMAPBUFF *bufp, ResultSet *rssp, const RMAPINFO *rmrp, const RMAPPROF *rprofp,
		       SeqFastq *readp, int ktuple_maxhit, uint32_t min_cover, int min_swatscor, int min_swatscor_below_max,
		       short target_depth, short max_depth, RMAPFLG_t rmapflg, const HashTable *htp, const SeqSet *ssp,
		       const SeqCodec *codecp, const InterVal *ivr);
#define mapSingleRead(first, ...) MSR_##first, __VA_ARGS__)
#define MSR_ErrMsg mapSingleRead_impl(ErrMsg
#define MSR_errmsgp rd_msr_hook(errmsgp
#include "rmap.c"
#undef mapSingleRead

extern void rdDumpHitInfo(FILE *fp, char strand, const HashHitInfo *hip);
extern void rdDumpCands(FILE *fp, const SegAliCands *sacp);
extern void rdDumpResults(FILE *fp, const ResultSet *rsp);
extern void rdDumpPost(FILE *fp, const ResultSet *rsp);
extern int rdPostDump;
static int g_with_post = 0;
extern void rdDumpResultsFrom(FILE *fp, const ResultSet *rsp, unsigned first, int swmax, int sw2nd);
extern void rdDumpLastResult(FILE *fp, const ResultSet *rsp);
extern unsigned rdResultNum(const ResultSet *rsp);
extern void rdScoreMaxGet(const ResultSet *rsp, int *mx, int *mx2);
extern void rdScoreMaxSet(ResultSet *rsp, int mx, int mx2);
extern void rdScoreMaxUpdate(ResultSet *rsp, int scor);

static void dumpStages(FILE *fp, const RMAPINFO *rmrp, const RMAPBUFF *bufp, uint32_t readlen, UCHAR ktup)
{
  uint32_t i, n;
  if (readlen < ktup) return;
  rdDumpHitInfo(fp, 'F', rmrp->hhiFp);
  rdDumpHitInfo(fp, 'R', rmrp->hhiRp);
  rdDumpCands(fp, bufp->sacp);
  n = ARRLEN(bufp->candr);
  for (i=0; i<n; i++) {
    const RMAPCAND *cp = bufp->candr + i;
    fprintf(fp, "RC %u %u %u %u %llu %llu %d %d %lld %d\n", i, (unsigned) cp->flags, cp->qs, cp->qe,
	    (unsigned long long) cp->rs, (unsigned long long) cp->re, cp->band_l, cp->band_r,
	    (long long) cp->sqidx, cp->swscor);
  }
}

/* ---- paired mode: one block per mapSingleRead call of rmapPair ---- */
static struct { int on; unsigned long long pairno; int callno; const SeqFastq *readp, *matep; const HashTable *htp; } g_pm;

static int rd_msr_hook(ErrMsg *errmsgp, RMAPBUFF *bufp, ResultSet *rssp, const RMAPINFO *rmrp, const RMAPPROF *rprofp,
		       SeqFastq *readp, int ktuple_maxhit, uint32_t min_cover, int min_swatscor, int min_swatscor_below_max,
		       short target_depth, short max_depth, RMAPFLG_t rmapflg, const HashTable *htp, const SeqSet *ssp,
		       const SeqCodec *codecp, const InterVal *ivr)
{
  int rv, mx, mx2;
  unsigned nbefore;
  uint32_t readlen;
  if (!g_pm.on)
    return mapSingleRead_impl(errmsgp, bufp, rssp, rmrp, rprofp, readp, ktuple_maxhit, min_cover, min_swatscor,
			      min_swatscor_below_max, target_depth, max_depth, rmapflg, htp, ssp, codecp, ivr);
  seqFastqGetConstSequence(readp, &readlen, NULL);
  rdScoreMaxGet(rssp, &mx, &mx2);         /* the set's running score maxima before the call (UPDATE_SWATSCORMAX, results.c:1013) */
  printf("MS %d mate=%d niv=%d fine=%d minscor=%d mincov=%u belowmax=%d flags=%u prevmax=%d,%d\n", g_pm.callno++, (readp == g_pm.matep)? 1: 0,
	 (ivr)? interValNum(ivr): -1, (htp != g_pm.htp)? 1: 0, min_swatscor, min_cover, min_swatscor_below_max, (unsigned) rmapflg, mx, mx2);
  if (ivr) {
    int i, n = interValNum(ivr);
    for (i=0; i<n; i++) {
      SEQLEN_t lo, hi; SEQNUM_t sx;
      interValGet(&lo, &hi, &sx, NULL, i, ivr);
      printf("IV %lld %u %u\n", (long long) sx, (unsigned) lo, (unsigned) hi);
    }
  }
  nbefore = rdResultNum(rssp);
  if (nbefore > 0) rdDumpLastResult(stdout, rssp);       /* PL: a first new result equal to it is dropped (results.c:1906) */
  rv = mapSingleRead_impl(errmsgp, bufp, rssp, rmrp, rprofp, readp, ktuple_maxhit, min_cover, min_swatscor,
			  min_swatscor_below_max, target_depth, max_depth, rmapflg, htp, ssp, codecp, ivr);
  rdScoreMaxGet(rssp, &mx, &mx2);
  printf("READ %llu %s len=%u err=%d\n", g_pm.pairno, seqFastqGetSeqName(readp), readlen, rv);
  dumpStages(stdout, rmrp, bufp, readlen, hashTableGetKtupLen(htp, NULL));
  rdDumpResultsFrom(stdout, rssp, nbefore, mx, mx2);      /* RX: new results, the set's running maxima after the call */
  if (g_with_post) rdDumpPost(stdout, rssp);
  return rv;
}

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
  const char *matefil = NULL;
  int ins_min = 0, ins_max = 500;
  RSLTPAIRLIB_t pairlib = RSLTPAIRLIB_PAIREDEND;
  SeqIO *mfp = NULL;
  SeqFastq *matep = NULL;

  int split = 0;
  while ((c = getopt(argc, argv, "m:d:c:q:H:S:xnpsP:i:j:l:")) != -1) {
    switch (c) {
    case 's': split = 1; rmapflg |= RMAPFLG_SPLIT | RMAPFLG_NOSHRTINFO | RMAPFLG_SENSITIVE; break;      /* smalt map -p (smalt.c:507-511) */
    case 'p': g_with_post = 1; rdPostDump = 1; break;
    case 'P': matefil = optarg; break;
    case 'i': ins_max = atoi(optarg); break;
    case 'j': ins_min = atoi(optarg); break;
    case 'l': pairlib = (!strcmp(optarg, "mp"))? RSLTPAIRLIB_MATEPAIR: ((!strcmp(optarg, "pp"))? RSLTPAIRLIB_SAMESTRAND: RSLTPAIRLIB_PAIREDEND); break;
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
  if (matefil) rmapflg |= RMAPFLG_PAIRED;
  rmp = rmapCreate(htp, codecp, ssp, smp, rmapflg);
  readp = seqFastqCreate(0, SEQTYP_FASTQ);
  if (matefil) {
    matep = seqFastqCreate(0, SEQTYP_FASTQ);
    mfp = seqIOopen(&errcode, (char *) matefil, SEQIO_READ, 0);
    if (errcode) { fprintf(stderr, "cannot open mates (%d)\n", errcode); return 1; }
  }
  sfp = seqIOopen(&errcode, argv[optind+1], SEQIO_READ, 0);
  if (errcode) { fprintf(stderr, "cannot open reads (%d)\n", errcode); return 1; }

  while (!seqIOstatus(sfp)) {
    uint32_t readlen, covermin;
    if ((errcode = seqFastqRead(readp, sfp))) break;
    seqFastqEncode(readp, codecp);
    seqFastqGetConstSequence(readp, &readlen, NULL);
    if (mincover < 1.01) {
      covermin = (uint32_t) (mincover*readlen);
      if (covermin > readlen) covermin = readlen;
    } else covermin = (uint32_t) mincover;
    if (matefil) {                      /* as processMapArgs, smalt.c:1131-1165 */
      uint32_t matelen, covermin_mate = covermin;
      RSLTPAIRFLG_t pairflg = 0;
      if ((errcode = seqFastqRead(matep, mfp))) break;
      seqFastqEncode(matep, codecp);
      seqFastqGetConstSequence(matep, &matelen, NULL);
      if (mincover < 1.01) { covermin_mate = (uint32_t) (mincover*matelen); if (covermin_mate > matelen) covermin_mate = matelen; }
      printf("PAIR %llu %s %s len=%u,%u\n", readno, seqFastqGetSeqName(readp), seqFastqGetSeqName(matep), readlen, matelen);
      g_pm.on = 1; g_pm.pairno = readno; g_pm.callno = 0; g_pm.readp = readp; g_pm.matep = matep; g_pm.htp = htp;
      errcode = rmapPair(errmsgp, rmp, readp, matep, &pairflg, ins_min, ins_max, pairlib, ncut, covermin, covermin_mate,
			 minscor, (UCHAR) minbasq, 512, 2048, (RMAPFLG_t) (rmapflg | RMAPFLG_PAIRED), smp, rfp, htp, ssp, codecp);
      g_pm.on = 0;
      if (getenv("REFDUMP_PAIRPOST")) {      /* debugging aid: the two sets as rmapPair leaves them (after pairing and the filters) */
	printf("PP %llu read\n", readno); rdDumpPost(stdout, rmp->rsrp);
	printf("PP %llu mate\n", readno); rdDumpPost(stdout, rmp->rsmp);
      }
      printf("PE %llu err=%d pairflg=%u ncalls=%d\n", readno, errcode, (unsigned) pairflg, g_pm.callno);
      readno++;
      continue;
    }
    if (split) {         /* split reads: one block per mapSingleRead call (the read's own, then mapSecondary's, rmap.c:1435), as in paired mode */
      printf("PAIR %llu %s - len=%u,0\n", readno, seqFastqGetSeqName(readp), readlen);
      g_pm.on = 1; g_pm.pairno = readno; g_pm.callno = 0; g_pm.readp = readp; g_pm.matep = NULL; g_pm.htp = htp;
      errcode = rmapSingle(errmsgp, rmp, readp, ncut, covermin, minscor, scordiff,
			   (UCHAR) minbasq, 512, 2048, rmapflg, smp, rfp, htp, ssp, codecp);
      g_pm.on = 0;
      printf("PE %llu err=%d pairflg=0 ncalls=%d\n", readno, errcode, g_pm.callno);
      if (g_with_post) rdDumpPost(stdout, rmp->rsrp);      /* the set as rmapSingle leaves it */
      readno++;
      continue;
    }
    errcode = rmapSingle(errmsgp, rmp, readp, ncut, covermin, minscor, scordiff,
			 (UCHAR) minbasq, 512, 2048, rmapflg, smp, rfp, htp, ssp, codecp);
    printf("READ %llu %s len=%u err=%d\n", readno++, seqFastqGetSeqName(readp), readlen, errcode);
    dumpStages(stdout, rmp->mrp, rmp->bfp, readlen, ktup);
    rdDumpResults(stdout, rmp->rsrp);
    if (g_with_post) rdDumpPost(stdout, rmp->rsrp);
    if (with_hitlists && readlen >= ktup)
      dumpHitLists(stdout, rmp, ncut, (BOOL) ((rmapflg & RMAPFLG_SEQBYSEQ) != 0), htp, ssp);
  }
  seqIOclose(sfp);
  return 0;
}
