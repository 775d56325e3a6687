/* oracle/or_align.c -- TEST INFRASTRUCTURE: restatement of SMALT's three Smith-Waterman
 * kernels: K2a (un-banded striped score pass, swsimd.c:868 -- textbook Gotoh maximum),
 * K2b (banded score-only scalar pass, alignment.c:1029) and K3 (banded pass with direction
 * matrix, traceback to a DiffStr and recursive split, alignment.c:788, 628, 1300). */
#include <stdlib.h>
#include <string.h>
#include "or_internal.h"

enum { DIR_COL = 1, DIR_ROW = 2, DIR_DIA = 3 };                 /* alignment.c:53-59 */
enum { DIFF_M = 0, DIFF_D = 1, DIFF_I = 2, DIFF_S = 3, DIFF_MAXMISMATCH = 61, DIFF_TYPSHIFT = 6 }; /* diffstr.h:90-107 */
enum { ALILEN_MIN = 5 };

/* score.c:138-173; alphabet "ACGTXN" (sequence.c:101): N rows/cols 0, X = mismatch-match */
void or_score_matrix(int8_t M[8][8], int match, int mismatch)
{
  static const char alphabet[] = "ACGTXN";
  int i, j;
  for (i = 0; i < 8; i++)
    for (j = 0; j < 8; j++) {
      if (i >= 6 || j >= 6 || alphabet[i] == 'N' || alphabet[j] == 'N') M[i][j] = 0;
      else if (alphabet[i] == 'X' || alphabet[j] == 'X') M[i][j] = (int8_t) (mismatch - match);
      else M[i][j] = (int8_t) ((i == j)? match: mismatch);
    }
}

/* K2a.  The SSE2 kernels (swsimd.c:656 unsigned bytes with bias, :443 shorts) compute, with
 * saturation standing in for max(0,.), exactly
 *    H = max(0, Hdiag + W, E, F);  E' = max(E - ext, H - init);  F' = max(F - ext, H - init)
 * and return max H (8-bit pass first, 16-bit on overflow); int arithmetic reproduces the
 * result for scores < 65535.  gap_init/gap_ext are the negative penalties (-4/-3). */
int or_sw_full(const uint8_t *q, uint32_t qlen, const uint8_t *r, uint32_t rlen,
               const int8_t M[8][8], int gap_init, int gap_ext)
{
  const int gi = -gap_init, ge = -gap_ext;
  int *H = calloc(qlen + 1, sizeof(int)), *E = calloc(qlen + 1, sizeof(int));
  int best = 0;
  uint32_t i, j;
  for (i = 0; i < rlen; i++) {
    const int8_t *w = M[r[i] & 7];
    int diag = 0, F = 0, h, t;
    for (j = 0; j < qlen; j++) {
      h = diag + w[q[j] & 7];
      if (h < 0) h = 0;
      if (h > best) best = h;
      if (E[j] > h) h = E[j];
      if (F > h) h = F;
      diag = H[j];
      H[j] = h;
      t = h - gi;
      E[j] -= ge; if (E[j] < t) E[j] = t; if (E[j] < 0) E[j] = 0;
      F -= ge; if (F < t) F = t; if (F < 0) F = 0;
    }
  }
  free(H); free(E);
  return best;
}

/* alignment.c:88-109, 310-396 (initALIBAND) */
typedef struct {
  int band_width, l_edge_orig, r_edge_orig, l_edge, r_edge;
  int s_left_orig, s_left, s_len, s_totlen, q_left_orig, q_left, q_len, q_totlen;
} Band;

static int band_init(Band *b, int l_edge, int r_edge, int q_left, int q_right, int q_len,
                     int s_left, int s_right, int s_len)
{
  b->s_len = (s_right < 0 || s_right >= s_len)? s_len: s_right + 1;
  b->q_len = (q_right < 0 || q_right >= q_len)? q_len: q_right + 1;
  b->s_totlen = s_len;
  b->q_totlen = q_len;
  b->s_left = b->s_left_orig = (s_left > 0 && s_left < b->s_len)? s_left: 0;
  b->q_left = b->q_left_orig = (q_left > 0 && q_left < b->q_len)? q_left: 0;
  b->l_edge_orig = b->l_edge = l_edge;
  b->r_edge_orig = b->r_edge = r_edge;
  b->band_width = r_edge - l_edge + 1;
  if (b->band_width <= 0) {
    b->band_width = 0;
    b->l_edge = b->q_left;
    b->r_edge = b->q_len - 1;
  } else {
    if (b->l_edge_orig + b->s_len > b->q_len) b->s_len = b->q_len - b->l_edge_orig;
    b->l_edge += b->s_left;
    if (b->l_edge >= b->q_len || b->r_edge_orig + b->s_len <= b->q_left) return OR_ERR_BAND;
    b->r_edge += b->s_left;
    if (b->r_edge < b->q_left) {
      b->s_left += b->q_left - b->r_edge;
      b->l_edge += b->q_left - b->r_edge;
      b->r_edge = b->q_left;
    }
    if (b->r_edge > b->q_len - 1) b->r_edge = b->q_len - 1;
  }
  b->band_width = b->r_edge - b->l_edge + 1;
  return (b->band_width >= 0)? OR_OK: OR_ERR_BAND;
}

/* The cell update shared by K2b and K3 (alignment.c:884-983 / :1109-1197): a restricted,
 * branch-ordered recurrence -- E and F are only re-seeded from H when H arrived by the
 * diagonal move and exceeds gap_init, and only such cells can raise the maximum.
 * Returns the direction code; *is_max_cand tells whether the cell may update the maximum. */
static inline int cell_update(int *Hj, int *Ej, int *Fp, int H, int gi, int ge, int *cand)
{
  int F = *Fp, E = *Ej, dir, tmp;
  *cand = 0;
  if (F > 0) {
    if (E > 0) {
      if (H > E) {
        if (H > F) {
          *Hj = H; F -= ge; E -= ge; dir = DIR_DIA;
          if (H > gi) { *cand = 1; tmp = H - gi; if (F < tmp) F = tmp; if (E < tmp) E = tmp; }
        } else { *Hj = F; F -= ge; E -= ge; dir = DIR_ROW; }
      } else {
        if (E >= F) { *Hj = E; dir = DIR_COL; } else { *Hj = F; dir = DIR_ROW; }
        E -= ge; F -= ge;
      }
    } else {
      if (H > F) {
        *Hj = H; F -= ge; dir = DIR_DIA;
        if (H > gi) { *cand = 1; E = H - gi; if (F < E) F = E; }
      } else { *Hj = F; F -= ge; dir = DIR_ROW; }
    }
  } else if (E > 0) {
    if (H > E) {
      *Hj = H; E -= ge; dir = DIR_DIA;
      if (H > gi) { *cand = 1; F = H - gi; if (E < F) E = F; }
    } else { *Hj = E; E -= ge; dir = DIR_COL; }
  } else {
    if (H > 0) {
      *Hj = H; dir = DIR_DIA;
      if (H > gi) { *cand = 1; F = E = H - gi; }
    } else { *Hj = 0; dir = 0; }
  }
  *Fp = F; *Ej = E;
  return dir;
}

/* K2b -- alignSmiWatBandFast, alignment.c:1029-1233.  NB unlike K3 the left band edge never
 * advances once it starts clipped at q_left (delta_band_start is not decremented, :1218). */
static int band_fast(const Band *bp, const uint8_t *q, const uint8_t *r, const int8_t M[8][8], int gi, int ge)
{
  int *Hp = calloc(bp->q_totlen + 2, sizeof(int)), *Ep = calloc(bp->q_totlen + 2, sizeof(int));
  int delta_start, j_start, j_len, i, j, H, currH = 0, F, best = 0, cand;
  if (bp->q_left > bp->l_edge) { delta_start = bp->q_left - bp->l_edge; j_start = bp->q_left; }
  else { delta_start = 0; j_start = bp->l_edge; }
  j_len = bp->r_edge + 1;
  for (i = bp->s_left; i < bp->s_len; i++) {
    const int8_t *w = M[r[i] & 7];
    F = 0;
    for (j = j_start; j < j_len; j++) {
      H = currH + w[q[j] & 7];
      currH = Hp[j];
      cell_update(&Hp[j], &Ep[j], &F, H, gi, ge, &cand);
      if (cand && H > best) best = H;
    }
    if (delta_start > 0) currH = 0;
    else { currH = Hp[j_start]; j_start++; }
    if (j_len < bp->q_len) j_len++;
  }
  free(Hp); free(Ep);
  return best;
}

int or_sw_band_fast(int *score, const uint8_t *q, uint32_t qlen, const uint8_t *r, int rlen,
                    const int8_t M[8][8], int gap_init, int gap_ext,
                    int l_edge, int r_edge, int q_left, int q_right, int s_left, int s_right)
{
  Band b;
  *score = 0;
  if (band_init(&b, l_edge, r_edge, q_left, q_right, (int) qlen, s_left, s_right, rlen)) return OR_ERR_BAND;
  *score = band_fast(&b, q, r, M, -gap_init, -gap_ext);
  return OR_OK;
}

/* K3 DP -- alignSmiWatBand, alignment.c:788-1027 */
typedef struct { int max_i, max_j, max_scor; uint8_t *dir; size_t ndir; } Track;

static void band_track(Track *t, const Band *bp, const uint8_t *q, const uint8_t *r, const int8_t M[8][8], int gi, int ge)
{
  int *Hp = calloc(bp->q_totlen + 2, sizeof(int)), *Ep = calloc(bp->q_totlen + 2, sizeof(int));
  int delta_start, delta_end = 0, j_start, j_len, i, j, H, currH = 0, F, cand;
  size_t need = (size_t) bp->band_width*(bp->s_len - bp->s_left) + bp->band_width + 8;
  uint8_t *dirp;
  if (need > t->ndir) { t->dir = realloc(t->dir, need); t->ndir = need; }
  t->max_i = t->max_j = t->max_scor = 0;
  if (bp->q_left > bp->l_edge) { delta_start = bp->q_left - bp->l_edge; j_start = bp->q_left; }
  else { delta_start = 0; j_start = bp->l_edge; }
  j_len = bp->r_edge + 1;
  dirp = t->dir + delta_start;
  for (i = bp->s_left; i < bp->s_len; i++) {
    const int8_t *w = M[r[i] & 7];
    F = 0;
    for (j = j_start; j < j_len; j++, dirp++) {
      H = currH + w[q[j] & 7];
      currH = Hp[j];
      *dirp = (uint8_t) cell_update(&Hp[j], &Ep[j], &F, H, gi, ge, &cand);
      if (cand && H > t->max_scor) { t->max_i = i; t->max_j = j; t->max_scor = H; }
    }
    if (delta_start > 0) { currH = 0; dirp += --delta_start; }
    else { currH = Hp[j_start]; j_start++; }
    if (j_len < bp->q_len) j_len++;
    else dirp += delta_end++;
  }
  free(Hp); free(Ep);
}

typedef struct { uint8_t *d; int n, cap; } DStr;
static void dstr_put(DStr *s, int count, int typ)
{
  if (s->n >= s->cap) { s->cap = s->cap? 2*s->cap: 64; s->d = realloc(s->d, s->cap); }
  s->d[s->n++] = (uint8_t) (count + (typ << DIFF_TYPSHIFT));
}

/* makeMetaFromTrack, alignment.c:628-781: traceback into a REVERSED DiffStr */
static int traceback(DStr *ds, int *qs, int *rs, const Track *t, const Band *bp, const uint8_t *q,
                     const uint8_t *r, const int8_t M[8][8], int gi, int ge)
{
  const uint8_t *dp;
  int i, j, s, checksum = 0, gap_open = 0;
  uint8_t nmatch = 0;
  ds->n = 0;
  dp = t->dir + (size_t) (t->max_i - bp->s_left)*(bp->band_width - 1) + (t->max_j - bp->l_edge);
  for (i = t->max_i, j = t->max_j; i >= bp->s_left && j >= bp->q_left && *dp;) {
    if (*dp == DIR_DIA) {
      s = M[r[i] & 7][q[j] & 7];
      if (s > 0) {
        if (nmatch > DIFF_MAXMISMATCH) { dstr_put(ds, DIFF_MAXMISMATCH, DIFF_M); nmatch -= DIFF_MAXMISMATCH; }
        else nmatch++;
      } else { dstr_put(ds, nmatch, DIFF_S); nmatch = 0; }
      checksum += s;
      gap_open = 0;
      dp -= bp->band_width; i--; j--;
      continue;
    }
    if (gap_open) checksum -= ge; else { checksum -= gi; gap_open = 1; }
    if (*dp & DIR_COL) { dstr_put(ds, nmatch, DIFF_D); nmatch = 0; dp -= bp->band_width - 1; i--; continue; }
    if (!(*dp & DIR_ROW)) return OR_ERR;
    dstr_put(ds, nmatch, DIFF_I); nmatch = 0; dp--; j--;
  }
  dstr_put(ds, nmatch, DIFF_S);
  dstr_put(ds, 0, DIFF_M);
  *rs = i + 1;
  *qs = j + 1;
  return (checksum != t->max_scor)? OR_ERR_SWATSCOR: OR_OK;
}

/* diffStrReverse, diffstr.c:850-896.  Output length includes the terminating M:0. */
static int dstr_reverse(uint8_t **out, const uint8_t *in)
{
  int l, u = 0;
  uint8_t count, count_prev, typ, *o;
  for (l = 0; in[l]; l++);
  o = malloc((size_t) l + 4);
  l--;
  count_prev = in[l] & 0x3f; typ = in[l] >> DIFF_TYPSHIFT;
  if (typ != DIFF_S) { free(o); *out = NULL; return -1; }
  for (l--; l >= 0; l--) {
    count = in[l] & 0x3f; typ = in[l] >> DIFF_TYPSHIFT;
    if (typ == DIFF_M) {
      count_prev = (uint8_t) (count_prev + count + 1);
      if (count_prev > DIFF_MAXMISMATCH) { o[u++] = (uint8_t) (DIFF_MAXMISMATCH + (DIFF_M << DIFF_TYPSHIFT)); count_prev -= DIFF_MAXMISMATCH + 1; }
    } else {
      o[u++] = (uint8_t) (count_prev + (typ << DIFF_TYPSHIFT));
      count_prev = count;
    }
  }
  o[u++] = (uint8_t) (count_prev + (DIFF_S << DIFF_TYPSHIFT));
  o[u++] = 0;
  *out = o;
  return u;
}

/* alignSmiWatBandRecursive, alignment.c:1300-1434: node first, then the reference range to
 * the left of the alignment, then the range to its right. */
typedef struct { OrAli *a; int n, cap; Track trk; DStr ds; } AliSet;

static int band_recursive(AliSet *as, const uint8_t *q, int q_len, const uint8_t *r, int s_len,
                          const int8_t M[8][8], int gi, int ge, int l_edge, int r_edge, int q_left, int q_right,
                          int s_left, int s_right, int minscore, int minscorlen)
{
  Band band;
  int qs, rs, qe, re, s_start, s_end, rv;
  if (minscorlen < 2) return OR_ERR;
  if (band_init(&band, l_edge, r_edge, q_left, q_right, q_len, s_left, s_right, s_len)) return OR_OK;
  if (band.s_left >= band.s_len || band.band_width < 0) return OR_ERR;   /* setMemALITRACK, :459 */
  band_track(&as->trk, &band, q, r, M, gi, ge);
  if (as->trk.max_scor < minscore) return OR_OK;
  if ((rv = traceback(&as->ds, &qs, &rs, &as->trk, &band, q, r, M, gi, ge))) return rv;
  qe = as->trk.max_j; re = as->trk.max_i;
  if (qs + minscorlen > qe + 1) return OR_OK;
  s_start = rs; s_end = re;
  if (as->trk.max_scor >= minscore) {
    OrAli *a;
    if (as->n >= as->cap) { as->cap = as->cap? 2*as->cap: 16; as->a = realloc(as->a, as->cap*sizeof(OrAli)); }
    a = as->a + as->n++;
    a->score = as->trk.max_scor; a->qs = qs; a->qe = qe; a->rs = rs; a->re = re;
    a->dlen = dstr_reverse(&a->diffstr, as->ds.d);
    if (a->dlen < 0) return OR_ERR;
  }
  if (s_left + minscorlen < s_start) {
    rv = band_recursive(as, q, q_len, r, s_len, M, gi, ge, l_edge, r_edge, q_left, q_right, s_left, s_start - 1, minscore, minscorlen);
    if (rv) return rv;
  }
  if (s_right > s_end + minscorlen) {
    rv = band_recursive(as, q, q_len, r, s_len, M, gi, ge, l_edge, r_edge, q_left, q_right, s_end + 1, s_right, minscore, minscorlen);
    if (rv) return rv;
  }
  return OR_OK;
}

/* aliSmiWatInBand, alignment.c:1548-1601 */
int or_sw_band_full(OrAli **out, int *nout, const uint8_t *q, uint32_t qlen, const uint8_t *r, int rlen,
                    const int8_t M[8][8], int gap_init, int gap_ext, int match_avg,
                    int l_edge, int r_edge, int q_left, int q_right, int s_left, int s_right,
                    int minscore, int minscorlen)
{
  AliSet as;
  int rv;
  *out = NULL; *nout = 0;
  if (minscore < 1 || match_avg <= 0) return OR_ERR;
  if (minscorlen*match_avg < minscore) minscorlen = minscore/match_avg;
  if (minscorlen < ALILEN_MIN) return OR_ERR;
  memset(&as, 0, sizeof(as));
  rv = band_recursive(&as, q, (int) qlen, r, rlen, M, -gap_init, -gap_ext, l_edge, r_edge, q_left, q_right,
                      s_left, s_right, minscore, minscorlen);
  free(as.trk.dir); free(as.ds.d);
  *out = as.a; *nout = as.n;
  return rv;
}

void or_ali_free(OrAli *a, int n)
{
  int i;
  for (i = 0; i < n; i++) free(a[i].diffstr);
  free(a);
}
