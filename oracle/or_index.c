/* oracle/or_index.c -- TEST INFRASTRUCTURE: restatement of SMALT's reference storage (I2)
 * and k-mer hash index (I1): build, file formats and lookup.  See smalt_oracle.h. */
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include "smalt_oracle.h"

enum { FILIO_NHEAD = 12, FILIO_SIG = 0x73212173, FILIO_ENDIAN = 0x6E378A19 }; /* filio.c:33-38 */
enum { FILTYP_SEQSET = 1, FILTYP_HASHTAB = 2 };                              /* filio.h:44-50 */
enum { SMA_VERSION = 4, SMA_NHEAD = 8, SMI_VERSION = 3, SMI_NHEAD = 8 };       /* sequence.c:79-80, hashidx.c:44-47 */
enum { SEQSET_COMPRESSED = 2 };                                              /* sequence.h:90 */
enum { BASES_PER_WORD = 10 };

/* sequence.c:287-322: upper-case; U->T; A C G T -> 0..3; everything else (incl. X) -> 5 */
uint8_t or_code_of(unsigned char c)
{
  int u = toupper(c);
  if (u == 'U') u = 'T';
  switch (u) {
  case 'A': return 0;
  case 'C': return 1;
  case 'G': return 2;
  case 'T': return 3;
  default:  return 5;
  }
}

/* hashidx.c:163-172 */
static uint32_t hash32mix(uint32_t a)
{
  a = (a+0x7ed55d16) + (a<<12);
  a = (a^0xc761c23c) ^ (a>>19);
  a = (a+0x165667b1) + (a<<5);
  a = (a+0xd3a2646c) ^ (a<<9);
  a = (a+0xfd7046c5) + (a<<3);
  a = (a^0xb55a4f09) ^ (a>>16);
  return a;
}

/* hashidx.c:155-158 */
static uint32_t make_key(const OrIndex *ix, uint64_t word, uint32_t *word_hi)
{
  const uint64_t wordmask = (((uint64_t) 1) << (2*ix->k)) - 1;
  const uint64_t mask_lo = (((uint64_t) 1) << ix->nbits_lo) - 1;
  const uint32_t keymod = ((uint32_t) 1) << (ix->nbits_key - ix->nbits_lo);
  uint32_t hi = (uint32_t) ((word & wordmask & ~mask_lo) >> ix->nbits_lo);
  uint32_t key_hi = hash32mix(hi) % keymod;
  *word_hi = hi;
  return (key_hi << ix->nbits_lo) + (uint32_t) (word & mask_lo);
}

/* smalt.c:268-332 */
static int select_hash_type(int *typ, int *nbits_key, int *nbits_perf, int k, int s, uint64_t totlen)
{
  const int nbk = 2*k;
  uint64_t ntup, nkey;
  *typ = OR_IDX_PERFECT; *nbits_key = 0; *nbits_perf = 0;
  if (nbk > 63) return OR_ERR;
  if (s < 1) s = 1;
  ntup = totlen/s;
  nkey = ((uint64_t) 1) << nbk;
  if (ntup > UINT32_MAX) return OR_ERR;
  if (nkey > 2*ntup) {
    int last_b, i;
    uint32_t t;
    *typ = OR_IDX_HASH32MIX;
    last_b = (ntup & 1)? 1: 0;
    for (t = (uint32_t) ntup, i = 0; i < 32; i++) {
      t >>= 1;
      if (t & 1) last_b = i;
    }
    *nbits_key = (last_b & 1)? last_b + 1: last_b;
    if (nbk > 32) {
      *nbits_perf = nbk - 32;
      if (*nbits_perf > 10) return OR_ERR;
    }
    if (*nbits_key + *nbits_perf > 26) *nbits_key = 26 - *nbits_perf;
    if (*nbits_key < *nbits_perf + 1) *nbits_key = *nbits_perf + 1;
    if (*nbits_key > 26) *nbits_key = 26;
  }
  return OR_OK;
}

typedef struct { uint32_t key, word_hi, pos; } KmerRec;
static int cmp_rec(const void *a, const void *b)
{
  const KmerRec *x = a, *y = b;
  if (x->key != y->key) return (x->key < y->key)? -1: 1;
  if (x->word_hi != y->word_hi) return (x->word_hi < y->word_hi)? -1: 1;
  return (x->pos < y->pos)? -1: (x->pos > y->pos);
}

/* hashidx.c:465-531 (doWordsInSeq) restated: visit every sampled k-mer of one sequence.
 * `tuplectr` is the k-mer serial number (global base offset / s), `offs` the offset of the
 * first sampled k-mer within the sequence. */
typedef void (*kmer_fn)(void *ctx, uint64_t word, uint32_t serial);
static void scan_kmers(const uint8_t *codes, uint32_t len, int k, int s, uint32_t *tuplectr,
                       int *offs, kmer_fn fn, void *ctx)
{
  uint64_t word = 0;
  int countdown = k + *offs, bad = 0, o;
  uint32_t i;
  for (i = 0; i < len; i++) {
    if (codes[i] & 4) bad = k; else if (bad) bad--;
    word = (word << 2) + (codes[i] & 3);
    if (--countdown > 0) continue;
    if (!bad) fn(ctx, word, *tuplectr);
    (*tuplectr)++;
    countdown = s;
  }
  o = (k - countdown) % s;
  if (o) o = s - o;
  *offs = o;
  *tuplectr += (uint32_t) ((k - countdown + o)/s);
}

typedef struct { const OrIndex *ix; KmerRec *rec; size_t n; uint32_t *cnt; int pass; } BuildCtx;
static void build_visit(void *vp, uint64_t word, uint32_t serial)
{
  BuildCtx *c = vp;
  if (c->pass == 0) { c->n++; return; }
  if (c->ix->typ == OR_IDX_PERFECT) {
    c->rec[c->n].key = (uint32_t) (word & ((((uint64_t) 1) << (2*c->ix->k)) - 1));
    c->rec[c->n].word_hi = 0;
  } else {
    c->rec[c->n].key = make_key(c->ix, word, &c->rec[c->n].word_hi);
  }
  c->rec[c->n].pos = serial;
  c->n++;
}

OrIndex *or_index_build(int nseq, const char *const *seqs, const uint32_t *lens,
                        const char *const *names, int k, int s)
{
  OrIndex *ix = calloc(1, sizeof(OrIndex));
  uint64_t tot = 0, namsiz = 0, o;
  uint8_t *codes;
  int i, offs;
  uint32_t tuplectr, j;
  BuildCtx ctx;
  size_t n, a;

  ix->k = k; ix->s = s; ix->nseq = nseq;
  ix->sop = calloc(nseq + 1, sizeof(uint64_t));
  for (i = 0; i < nseq; i++) {
    if (lens[i] < (uint32_t) k) { or_index_free(ix); return NULL; } /* hashidx.c:499 */
    ix->sop[i+1] = ix->sop[i] + lens[i];
    namsiz += strlen(names[i]) + 1;
  }
  tot = ix->sop[nseq];
  ix->totlen = tot;
  ix->namsiz = namsiz;
  ix->names = malloc(namsiz);
  for (i = 0, o = 0; i < nseq; i++) { strcpy(ix->names + o, names[i]); o += strlen(names[i]) + 1; }

  /* sequence.c:1360-1424 (compressSeq): 10 codes/word, first base highest, TERM=7 after last */
  codes = malloc(tot + 1);
  for (i = 0; i < nseq; i++)
    for (j = 0; j < lens[i]; j++) codes[ix->sop[i] + j] = or_code_of((unsigned char) seqs[i][j]);
  ix->packed = calloc(tot/BASES_PER_WORD + 1, sizeof(uint32_t));
  for (o = 0; o <= tot; o++) {
    uint32_t c = (o < tot)? codes[o]: 7;
    ix->packed[o/BASES_PER_WORD] += c << (3*(BASES_PER_WORD - 1 - (int) (o % BASES_PER_WORD)));
  }

  if (select_hash_type(&ix->typ, &ix->nbits_key, &ix->nbits_lo, k, s, tot)) { free(codes); or_index_free(ix); return NULL; }
  if (ix->typ == OR_IDX_PERFECT) { ix->nbits_key = 2*k; ix->nbits_lo = 0; }
  ix->nkeys = ((uint32_t) 1) << ix->nbits_key;

  memset(&ctx, 0, sizeof(ctx));
  ctx.ix = ix;
  for (ctx.pass = 0; ctx.pass < 2; ctx.pass++) {
    if (ctx.pass == 1) { ctx.rec = malloc((ctx.n + 1)*sizeof(KmerRec)); ctx.n = 0; }
    tuplectr = 0; offs = 0;
    for (i = 0; i < nseq; i++)
      scan_kmers(codes + ix->sop[i], lens[i], k, s, &tuplectr, &offs, build_visit, &ctx);
  }
  free(codes);
  n = ctx.n;
  ix->npos = (uint32_t) n;
  ix->maxpos = (tuplectr > 0)? tuplectr - 1: 0;   /* hashidx.c:992 */
  qsort(ctx.rec, n, sizeof(KmerRec), cmp_rec);
  ix->pos = malloc((n + 1)*sizeof(uint32_t));
  for (a = 0; a < n; a++) ix->pos[a] = ctx.rec[a].pos;
  ix->idx = calloc((size_t) ix->nkeys + 2, sizeof(uint32_t));
  if (ix->typ == OR_IDX_PERFECT) {
    for (a = 0; a < n; a++) ix->idx[ctx.rec[a].key + 1]++;
    for (j = 0; j < ix->nkeys; j++) ix->idx[j+1] += ix->idx[j];
  } else {
    /* idx counts distinct words per key; wordidx lists them; posidx are offsets into pos */
    uint32_t nw = 0;
    for (a = 0; a < n; a++)
      if (a == 0 || ctx.rec[a].key != ctx.rec[a-1].key || ctx.rec[a].word_hi != ctx.rec[a-1].word_hi) nw++;
    ix->nwords = nw;
    ix->wordidx = calloc((size_t) nw + 2, sizeof(uint32_t));
    ix->posidx = calloc((size_t) nw + 2, sizeof(uint32_t));
    nw = 0;
    for (a = 0; a < n; a++) {
      if (a == 0 || ctx.rec[a].key != ctx.rec[a-1].key || ctx.rec[a].word_hi != ctx.rec[a-1].word_hi) {
        ix->wordidx[nw] = ctx.rec[a].word_hi;
        ix->posidx[nw] = (uint32_t) a;
        ix->idx[ctx.rec[a].key + 1]++;
        nw++;
      }
    }
    ix->posidx[nw] = (uint32_t) n;
    for (j = 0; j < ix->nkeys; j++) ix->idx[j+1] += ix->idx[j];
  }
  free(ctx.rec);
  return ix;
}

/* The on-the-fly index of rmapPair's rescue round (rmap.c:495-517 setupFineHashTable -> hashTableSetUp with an interval
 * set, hashidx.c:549-575 doAllWordsInSeqSet): perfect type, k = 5, s = 1 (rmap.c:91-92; hashTableCreate rmap.c:1543),
 * over the windows [lo, hi] of sequences sx only.  Serials are global (calcKtupOffs, hashidx.c:325: (sop[sx] + lo) / s) and
 * ascending per key because the intervals are pruned (sorted, disjoint).  The packed reference, offsets and names are
 * SHARED with `main` (the returned index must be freed with or_index_free_fine before `main`). */
OrIndex *or_index_build_fine(const OrIndex *main, int niv, const int64_t *sx, const uint32_t *lo, const uint32_t *hi, int k, int s)
{
  OrIndex *ix = calloc(1, sizeof(OrIndex));
  BuildCtx ctx;
  uint8_t *codes = NULL;
  size_t n, a, cap = 0;
  uint32_t j, tuplectr = 0;
  int i;
  *ix = *main;
  ix->k = k; ix->s = s; ix->typ = OR_IDX_PERFECT; ix->nbits_key = 2*k; ix->nbits_lo = 0;
  ix->nkeys = ((uint32_t) 1) << ix->nbits_key;
  ix->nwords = 0; ix->wordidx = ix->posidx = NULL; ix->idx = ix->pos = NULL;
  memset(&ctx, 0, sizeof(ctx));
  ctx.ix = ix;
  for (ctx.pass = 0; ctx.pass < 2; ctx.pass++) {
    if (ctx.pass == 1) { ctx.rec = malloc((ctx.n + 1)*sizeof(KmerRec)); ctx.n = 0; }
    for (i = 0; i < niv; i++) {
      const uint64_t g = main->sop[sx[i]] + lo[i];
      const uint32_t sl = hi[i] - lo[i] + 1;
      int offs;
      tuplectr = (uint32_t) (g/(uint64_t) s);
      offs = (int) (g - (uint64_t) tuplectr*(uint64_t) s);
      if (sl < (uint32_t) k) continue;                                  /* hashidx.c:561-563 */
      if (sl + 1 > cap) { cap = sl + 1024; codes = realloc(codes, cap); }
      or_index_fetch(main, g, sl, codes);
      scan_kmers(codes, sl, k, s, &tuplectr, &offs, build_visit, &ctx);
    }
  }
  free(codes);
  n = ctx.n;
  ix->npos = (uint32_t) n;
  ix->maxpos = (tuplectr > 0)? tuplectr - 1: 0;
  qsort(ctx.rec, n, sizeof(KmerRec), cmp_rec);
  ix->pos = malloc((n + 1)*sizeof(uint32_t));
  for (a = 0; a < n; a++) ix->pos[a] = ctx.rec[a].pos;
  ix->idx = calloc((size_t) ix->nkeys + 2, sizeof(uint32_t));
  for (a = 0; a < n; a++) ix->idx[ctx.rec[a].key + 1]++;
  for (j = 0; j < ix->nkeys; j++) ix->idx[j+1] += ix->idx[j];
  free(ctx.rec);
  return ix;
}

void or_index_free_fine(OrIndex *ix)
{
  if (!ix) return;
  free(ix->idx); free(ix->pos); free(ix);
}

void or_index_free(OrIndex *ix)
{
  if (!ix) return;
  free(ix->idx); free(ix->pos); free(ix->wordidx); free(ix->posidx);
  free(ix->sop); free(ix->packed); free(ix->names); free(ix);
}

static FILE *open_ext(const char *prefix, const char *ext, const char *mode)
{
  char *fn = malloc(strlen(prefix) + strlen(ext) + 2);
  FILE *fp;
  sprintf(fn, "%s.%s", prefix, ext);
  fp = fopen(fn, mode);
  free(fn);
  return fp;
}

/* filio.c:54-77 */
static void write_filio_header(FILE *fp, uint32_t siz, uint32_t typ, uint32_t version, uint32_t headsiz)
{
  uint32_t h[FILIO_NHEAD];
  memset(h, 0, sizeof(h));
  h[0] = FILIO_SIG; h[1] = FILIO_ENDIAN; h[2] = siz + FILIO_NHEAD; h[3] = typ; h[4] = version; h[5] = headsiz;
  fwrite(h, sizeof(uint32_t), FILIO_NHEAD, fp);
}

int or_index_write(const OrIndex *ix, const char *prefix)
{
  FILE *fp;
  uint32_t h[8], *seqlen, i;
  uint64_t seqnamsiz, seqsiz, totsiz;

  /* .sma: sequence.c:2448-2519 */
  if (!(fp = open_ext(prefix, "sma", "wb"))) return OR_ERR_FILE;
  h[0] = (uint32_t) ix->nseq; h[1] = (uint32_t) (((uint64_t) ix->nseq) >> 32);
  h[2] = (uint32_t) ix->namsiz; h[3] = (uint32_t) (ix->namsiz >> 32);
  h[4] = (uint32_t) ix->totlen; h[5] = (uint32_t) (ix->totlen >> 32);
  h[6] = SEQSET_COMPRESSED; h[7] = 0;
  seqnamsiz = (ix->namsiz - 1)/4 + 1;
  seqsiz = ix->totlen/BASES_PER_WORD + 1;
  totsiz = SMA_NHEAD + seqsiz + ix->nseq + seqnamsiz;
  write_filio_header(fp, (uint32_t) totsiz, FILTYP_SEQSET, SMA_VERSION, SMA_NHEAD);
  fwrite(h, sizeof(uint32_t), SMA_NHEAD, fp);
  fwrite(ix->names, 1, ix->namsiz, fp);
  seqlen = malloc(ix->nseq*sizeof(uint32_t));
  for (i = 0; i < ix->nseq; i++) seqlen[i] = (uint32_t) (ix->sop[i+1] - ix->sop[i]);
  fwrite(seqlen, sizeof(uint32_t), ix->nseq, fp);
  free(seqlen);
  fwrite(ix->packed, sizeof(uint32_t), seqsiz, fp);
  fclose(fp);

  /* .smi: hashidx.c:1214-1255 */
  if (!(fp = open_ext(prefix, "smi", "wb"))) return OR_ERR_FILE;
  h[0] = ix->k; h[1] = ix->s; h[2] = ix->npos; h[3] = ix->maxpos; h[4] = ix->typ;
  h[5] = ix->nbits_key; h[6] = ix->nbits_lo; h[7] = ix->nwords;
  totsiz = (uint64_t) ix->npos + ix->nkeys + 1;
  if (ix->typ != OR_IDX_PERFECT) totsiz += ((uint64_t) ix->nwords + 1)*2;
  write_filio_header(fp, (uint32_t) totsiz, FILTYP_HASHTAB, SMI_VERSION, SMI_NHEAD);
  fwrite(h, sizeof(uint32_t), SMI_NHEAD, fp);
  fwrite(ix->idx, sizeof(uint32_t), (size_t) ix->nkeys + 1, fp);
  fwrite(ix->pos, sizeof(uint32_t), ix->npos, fp);
  if (ix->typ != OR_IDX_PERFECT) {
    fwrite(ix->wordidx, sizeof(uint32_t), (size_t) ix->nwords + 1, fp);
    fwrite(ix->posidx, sizeof(uint32_t), (size_t) ix->nwords + 1, fp);
  }
  fclose(fp);
  return OR_OK;
}

static int read_filio_header(FILE *fp, uint32_t typ_want, uint32_t *version, uint32_t *h, uint32_t nh)
{
  uint32_t f[FILIO_NHEAD];
  if (fread(f, sizeof(uint32_t), FILIO_NHEAD, fp) != FILIO_NHEAD) return OR_ERR_FILE;
  if (f[0] != FILIO_SIG || f[1] != FILIO_ENDIAN) return OR_ERR_FILE; /* same-endian files only */
  if ((f[3] & 0xff) != typ_want || f[5] > nh) return OR_ERR_FILE;
  *version = f[4];
  if (fread(h, sizeof(uint32_t), f[5], fp) != f[5]) return OR_ERR_FILE;
  return OR_OK;
}

OrIndex *or_index_read(const char *prefix)
{
  OrIndex *ix = calloc(1, sizeof(OrIndex));
  FILE *fp;
  uint32_t h[8], ver, *seqlen, i;
  uint64_t seqsiz, o;
  int64_t s;

  /* .sma (format version 4 only): sequence.c:2521-2686 */
  if (!(fp = open_ext(prefix, "sma", "rb"))) goto fail;
  if (read_filio_header(fp, FILTYP_SEQSET, &ver, h, SMA_NHEAD) || ver != SMA_VERSION) { fclose(fp); goto fail; }
  ix->nseq = (int64_t) ((((uint64_t) h[1]) << 32) + h[0]);
  ix->namsiz = (((uint64_t) h[3]) << 32) + h[2];
  ix->totlen = (((uint64_t) h[5]) << 32) + h[4];
  ix->names = malloc(ix->namsiz + 1);
  if (fread(ix->names, 1, ix->namsiz, fp) != ix->namsiz) { fclose(fp); goto fail; }
  seqlen = malloc(ix->nseq*sizeof(uint32_t));
  if (fread(seqlen, sizeof(uint32_t), ix->nseq, fp) != (size_t) ix->nseq) { free(seqlen); fclose(fp); goto fail; }
  ix->sop = calloc(ix->nseq + 1, sizeof(uint64_t));
  for (s = 0; s < ix->nseq; s++) ix->sop[s+1] = ix->sop[s] + seqlen[s];
  free(seqlen);
  seqsiz = ix->totlen/BASES_PER_WORD + 1;
  ix->packed = malloc(seqsiz*sizeof(uint32_t));
  if (fread(ix->packed, sizeof(uint32_t), seqsiz, fp) != seqsiz) { fclose(fp); goto fail; }
  fclose(fp);

  /* .smi: hashidx.c:1257-1366.  NB the reference reads only 2*nwords+1 of the 2*(nwords+1)
   * collision-table words (:1334), so posidx[nwords] stays 0 from calloc: the last word of
   * the table then reports (0 - posidx[nwords-1]) hits.  Reproduced here on purpose. */
  if (!(fp = open_ext(prefix, "smi", "rb"))) goto fail;
  if (read_filio_header(fp, FILTYP_HASHTAB, &ver, h, SMI_NHEAD) || ver != SMI_VERSION) { fclose(fp); goto fail; }
  ix->k = h[0]; ix->s = h[1]; ix->npos = h[2]; ix->maxpos = h[3]; ix->typ = h[4];
  ix->nbits_key = h[5]; ix->nbits_lo = h[6]; ix->nwords = h[7];
  if (ix->typ == OR_IDX_PERFECT) { ix->nbits_key = 2*ix->k; ix->nbits_lo = 0; }
  ix->nkeys = ((uint32_t) 1) << ix->nbits_key;
  ix->idx = calloc((size_t) ix->nkeys + 2, sizeof(uint32_t));
  ix->pos = calloc((size_t) ix->npos + 1, sizeof(uint32_t));
  if (fread(ix->idx, sizeof(uint32_t), (size_t) ix->nkeys + 1, fp) != (size_t) ix->nkeys + 1 ||
      fread(ix->pos, sizeof(uint32_t), ix->npos, fp) != ix->npos) { fclose(fp); goto fail; }
  if (ix->typ != OR_IDX_PERFECT) {
    uint32_t *w = calloc(((size_t) ix->nwords + 1)*2, sizeof(uint32_t));
    size_t nr = 2*(size_t) ix->nwords + 1;
    if (fread(w, sizeof(uint32_t), nr, fp) != nr) { free(w); fclose(fp); goto fail; }
    ix->wordidx = calloc((size_t) ix->nwords + 2, sizeof(uint32_t));
    ix->posidx = calloc((size_t) ix->nwords + 2, sizeof(uint32_t));
    for (i = 0; i <= ix->nwords; i++) { ix->wordidx[i] = w[i]; ix->posidx[i] = w[ix->nwords + 1 + i]; }
    free(w);
  }
  fclose(fp);
  (void) o;
  return ix;
fail:
  or_index_free(ix);
  return NULL;
}

/* hashidx.c:1146-1191 */
uint32_t or_index_lookup(const OrIndex *ix, uint64_t word, uint32_t *posidx)
{
  uint32_t nhits = 0;
  if (ix->typ == OR_IDX_PERFECT) {
    uint32_t key = (uint32_t) (word & ((((uint64_t) 1) << (2*ix->k)) - 1));
    *posidx = key;
    nhits = ix->idx[key+1] - ix->idx[key];
  } else {
    uint32_t word_hi, a, b, pivot;
    uint32_t key = make_key(ix, word, &word_hi);
    b = ix->idx[key+1];
    if (b < 1) return 0;
    a = ix->idx[key];
    b--;
    while (a < b) {
      pivot = (a + b) >> 1;
      if (ix->wordidx[pivot] < word_hi) a = pivot + 1; else b = pivot;
    }
    if (a == b && ix->wordidx[b] == word_hi) {
      nhits = ix->posidx[b+1] - ix->posidx[b];
      *posidx = b;
    }
  }
  return nhits;
}

/* hashidx.c:1193-1212 */
uint32_t or_index_positions(const OrIndex *ix, uint32_t posidx, const uint32_t **posp)
{
  *posp = NULL;
  if (ix->typ == OR_IDX_PERFECT) {
    if (posidx < ix->nkeys) { *posp = ix->pos + ix->idx[posidx]; return ix->idx[posidx+1] - ix->idx[posidx]; }
  } else if (posidx < ix->npos) {
    *posp = ix->pos + ix->posidx[posidx];
    return ix->posidx[posidx+1] - ix->posidx[posidx];
  }
  return 0;
}

/* sequence.c:1499-1550 (uncompressSeq) followed by seqFastqEncode: 3-bit codes of a window.
 * Codes 6 decode to 'N' and re-encode as 5; 7 (terminator) becomes a 0 byte (code 0). */
void or_index_fetch(const OrIndex *ix, uint64_t start, uint32_t len, uint8_t *codes)
{
  uint32_t i;
  for (i = 0; i < len; i++) {
    uint64_t o = start + i;
    uint32_t c = (ix->packed[o/BASES_PER_WORD] >> (3*(BASES_PER_WORD - 1 - (int) (o % BASES_PER_WORD)))) & 7;
    codes[i] = (uint8_t) ((c == 7)? 0: (c == 6 || c == 4)? 5: c); /* 'X' re-encodes as N */
  }
}

void or_params_default(OrParams *p, const OrIndex *ix)
{
  memset(p, 0, sizeof(*p));
  p->ncut = 10000;
  p->min_cover = 0;
  p->min_swatscor = ix->k + ix->s - 1;     /* smalt.c:608-615 */
  p->min_swatscor_below_max = 0;
  p->min_basq = 0;
  p->target_depth = 512; p->max_depth = 2048; /* smalt.c:60-61 */
  p->flags = OR_FLG_BEST | ((ix->nseq < 512)? OR_FLG_SEQBYSEQ: 0); /* smalt.c:495-497, 599 */
  p->match = 1; p->mismatch = -2; p->gap_init = -4; p->gap_ext = -3; /* score.c:41-47 */
}
