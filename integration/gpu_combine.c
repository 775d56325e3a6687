/* gpu_combine.c -- OUR host glue of the batched binding (integration/rmap_gpu.c): worker threads of `smalt map -n T`
 * each hand over a block of nthreads x 32 reads at a time; one GPU batch per block is too small to fill the device
 * (60 k reads/s at 512 reads per batch against 400 k+ at 32 k).  The threads therefore meet here: whoever arrives while
 * no batch is being assembled becomes the leader, waits a moment for the others, maps everything that is pending as ONE
 * batch on one shared mapper (smaltgpu_map_batch), and hands every thread its slice of the results.  Plain C on top
 * of the C ABI; nothing of the reference is needed here. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "smaltgpu.h"
#include "gpu_combine.h"

enum { COMB_MAXDEV = 16, COMB_MAXREQ = 512,
       COMB_MAXREADS = 8192 };   /* reads per combined batch: with more worker blocks than this pending, the rest forms the next batch
                                  * and is mapped while the first cohort's threads post-process (SMALTGPU_COMBINE_READS overrides) */

typedef struct {
  const char *bases, *quals; const uint64_t *off; uint32_t n;
  const smaltgpu_params *par;
  GpuCombOut *out;
  const GpuCombCtx *ctx;        /* round of rmapPair this request belongs to (NULL: plain reads); only equal kinds are combined */
  int done, rv;
  char err[256];                /* smaltgpu_last_error() is per thread: the leader copies its message to every request */
} CombReq;

/* tuning switches, read from the environment ONCE (pthread_once) before any batch: worker threads never call setenv, and
 * getenv only happens here */
static struct { uint32_t combine_reads; int nslot; } g_cfg = { COMB_MAXREADS, 2 };
static pthread_once_t g_cfg_once = PTHREAD_ONCE_INIT;
static void read_cfg(void)
{
  const char *e;
  if ((e = getenv("SMALTGPU_COMBINE_READS")) && atoi(e) > 0) g_cfg.combine_reads = (uint32_t)atoi(e);
  if ((e = getenv("SMALTGPU_COMBINE_SLOTS")) && atoi(e) >= 1 && atoi(e) <= 4) g_cfg.nslot = atoi(e);
}

/* one mapper (HIP stream) with its staging buffers; a device has COMB_NSLOT of them so that one batch is packed,
 * copied and unpacked while another one runs */
enum { COMB_NSLOT = 4 };                          /* capacity; SMALTGPU_COMBINE_SLOTS (default 2) of them are used */
#define g_nslot (g_cfg.nslot)
struct CombSlot {
  int busy;
  smaltgpu_mapper *mp; uint32_t cap_reads, cap_len;
  char *bases, *quals; uint64_t *off; size_t basecap;
  /* merged per-read context of a paired round */
  uint64_t *iv_off; smaltgpu_interval *iv; int32_t *minsw, *prevmax; uint32_t *tot, *seedrange;
  size_t cap_ivoff, cap_iv, cap_minsw, cap_prevmax, cap_tot, cap_seedrange;
};
/* ONE queue of pending worker blocks for all devices -- the reference's workers pull blocks from one FIFO (threads.c:548);
 * here whichever device has a free mapper slot takes the next cohort, so a device that got repeat-rich reads does not
 * hold the others up (no static assignment of threads to devices). */
static struct CombQueue {
  pthread_mutex_t mu; pthread_cond_t cv; int init;
  CombReq *pending[COMB_MAXREQ]; int npending; int assembling;      /* assembling: a leader is collecting requests */
  int ndev; const smaltgpu_index *ix[COMB_MAXDEV];
  struct CombSlot slot[COMB_MAXDEV][COMB_NSLOT];
  unsigned long batches[COMB_MAXDEV];                                /* combined batches run per device (diagnostic, SMALTGPU_COMBINE_STATS) */
} g_q;
static pthread_mutex_t g_init = PTHREAD_MUTEX_INITIALIZER;
static int g_closing = 0;                                            /* a worker failed: no new batch is launched (gpuCombineClose) */

static int grow(void **p, size_t *cap, size_t need, size_t elem)
{
  if (*cap >= need) return 0;
  size_t c = *cap ? *cap : 64;
  while (c < need) c *= 2;
  void *q = realloc(*p, c * elem);
  if (!q) return -1;
  *p = q; *cap = c;
  return 0;
}

/* one combined batch: reqs[0..nreq) on mapper slot d */
static void run_batch(struct CombSlot *d, const smaltgpu_index *ix, CombReq **reqs, int nreq)
{
  uint32_t ntot = 0, maxlen = 1, r0;
  size_t nb = 0;
  int i, has_qual = 1, rv = 0;
  smaltgpu_batch_out o;
  for (i = 0; i < nreq; i++) {
    uint32_t j;
    ntot += reqs[i]->n; nb += (size_t)reqs[i]->off[reqs[i]->n];
    if (!reqs[i]->quals) has_qual = 0;
    for (j = 0; j < reqs[i]->n; j++) { const uint32_t l = (uint32_t)(reqs[i]->off[j + 1] - reqs[i]->off[j]); if (l > maxlen) maxlen = l; }
  }
  if (!d->mp || d->cap_reads < ntot || d->cap_len < maxlen) {
    uint32_t cr = d->cap_reads > 16384 ? d->cap_reads : 16384, cl = d->cap_len > 64 ? d->cap_len : 64;
    while (cr < ntot) cr *= 2;
    if (cl < maxlen) cl = (maxlen + 31u) & ~31u;
    smaltgpu_mapper_opts opts = { 1024, 0 };             /* ranked candidates per read of the shared pools (the default sizing is for large batches) */
    pthread_mutex_lock(&g_init);                         /* one mapper at a time */
    if (d->mp) smaltgpu_mapper_free(d->mp);
    d->mp = NULL;
    free(d->off);
    d->off = malloc(((size_t)cr + 1) * sizeof(uint64_t));
    if (!d->off || smaltgpu_mapper_create_ex(&d->mp, ix, cr, cl, &opts)) rv = SMALTGPU_ENOMEM;
    else { d->cap_reads = cr; d->cap_len = cl; }
    pthread_mutex_unlock(&g_init);
  }
  if (!rv && d->basecap < nb + 1) {
    free(d->bases); free(d->quals);
    d->basecap = 2 * nb + 4096;
    d->bases = malloc(d->basecap); d->quals = malloc(d->basecap);
    if (!d->bases || !d->quals) rv = SMALTGPU_ENOMEM;
  }
  if (!rv) {
    size_t pos = 0;
    uint32_t k = 0;
    for (i = 0; i < nreq; i++) {
      uint32_t j;
      const size_t len = (size_t)reqs[i]->off[reqs[i]->n];
      memcpy(d->bases + pos, reqs[i]->bases, len);
      if (has_qual) memcpy(d->quals + pos, reqs[i]->quals, len);
      for (j = 0; j < reqs[i]->n; j++) d->off[k++] = pos + reqs[i]->off[j];
      pos += len;
    }
    d->off[k] = pos;
    if (!reqs[0]->ctx) rv = smaltgpu_map_batch(d->mp, (const uint8_t *)d->bases, has_qual ? (const uint8_t *)d->quals : NULL, d->off, ntot, reqs[0]->par, &o);
    else {                                              /* a round of rmapPair: the requests' contexts side by side */
      const GpuCombCtx *c0 = reqs[0]->ctx;
      smaltgpu_callctx ctx;
      size_t niv = 0;
      memset(&ctx, 0, sizeof(ctx));
      if (c0->iv_off) for (i = 0; i < nreq; i++) niv += (size_t)(reqs[i]->ctx->iv_off[reqs[i]->n] - reqs[i]->ctx->iv_off[0]);
      if ((c0->iv_off && (grow((void **)&d->iv_off, &d->cap_ivoff, (size_t)ntot + 1, sizeof(uint64_t)) || grow((void **)&d->iv, &d->cap_iv, niv + 1, sizeof(smaltgpu_interval)))) ||
          (c0->minsw && grow((void **)&d->minsw, &d->cap_minsw, (size_t)ntot + 1, sizeof(int32_t))) ||
          (c0->prevmax && grow((void **)&d->prevmax, &d->cap_prevmax, 2 * (size_t)ntot + 2, sizeof(int32_t))) ||
          (c0->seedrange && grow((void **)&d->seedrange, &d->cap_seedrange, 2 * (size_t)ntot + 2, sizeof(uint32_t))) ||
          (c0->kind == GPUCOMB_TOTALS && grow((void **)&d->tot, &d->cap_tot, (size_t)ntot + 1, sizeof(uint32_t)))) rv = SMALTGPU_ENOMEM;
      else {
        for (i = 0, k = 0, niv = 0; i < nreq; i++) {
          const GpuCombCtx *c = reqs[i]->ctx;
          uint32_t j;
          for (j = 0; j < reqs[i]->n; j++, k++) {
            if (c->iv_off) {
              uint64_t v;
              d->iv_off[k] = niv;
              for (v = c->iv_off[j]; v < c->iv_off[j + 1]; v++) d->iv[niv++] = c->iv[v];
            }
            if (c->minsw) d->minsw[k] = c->minsw[j];
            if (c->prevmax) { d->prevmax[2 * k] = c->prevmax[2 * j]; d->prevmax[2 * k + 1] = c->prevmax[2 * j + 1]; }
            if (c->seedrange) { d->seedrange[2 * k] = c->seedrange[2 * j]; d->seedrange[2 * k + 1] = c->seedrange[2 * j + 1]; }
          }
        }
        if (c0->iv_off) { d->iv_off[k] = niv; ctx.iv_off = d->iv_off; ctx.iv = d->iv; }
        if (c0->minsw) ctx.min_swatscor = d->minsw;
        if (c0->prevmax) ctx.prev_max = d->prevmax;
        if (c0->seedrange) ctx.seed_range = d->seedrange;
        ctx.fine_index = c0->kind == GPUCOMB_FINE;
        ctx.raw_alignments = c0->kind == GPUCOMB_APPEND || c0->kind == GPUCOMB_FINE || c0->kind == GPUCOMB_SPLIT;      /* these append to sets that hold alignments: resultSetAppendRaw compares */
        if (c0->kind == GPUCOMB_TOTALS) {
          rv = smaltgpu_hit_totals(d->mp, (const uint8_t *)d->bases, has_qual ? (const uint8_t *)d->quals : NULL, d->off, ntot, reqs[0]->par, d->tot);
          for (i = 0, k = 0; i < nreq && !rv; i++) { memcpy(reqs[i]->ctx->tot_out, d->tot + k, (size_t)reqs[i]->n * sizeof(uint32_t)); k += reqs[i]->n; }
          for (i = 0; i < nreq; i++) { reqs[i]->rv = rv; if (rv) { strncpy(reqs[i]->err, smaltgpu_last_error(), sizeof(reqs[i]->err) - 1); reqs[i]->err[sizeof(reqs[i]->err) - 1] = 0; } }
          return;
        }
        rv = smaltgpu_map_batch_ctx(d->mp, (const uint8_t *)d->bases, has_qual ? (const uint8_t *)d->quals : NULL, d->off, ntot, reqs[0]->par, &ctx, &o);
      }
    }
    /* pool overflows are recovered inside the library; what can remain is a read that fails on its own (stat[].errcode):
     * the batch is complete for every other read, so hand the slices out and let the owner of that read report it */
    if (SMALTGPU_IS_READ_ERROR(rv) && o.nreads == ntot) rv = 0;
  }
  if (rv) for (i = 0; i < nreq; i++) { strncpy(reqs[i]->err, smaltgpu_last_error(), sizeof(reqs[i]->err) - 1); reqs[i]->err[sizeof(reqs[i]->err) - 1] = 0; }
  for (i = 0, r0 = 0; i < nreq; i++) {                    /* every request gets its own copy of its slice */
    CombReq *q = reqs[i];
    GpuCombOut *w = q->out;
    q->rv = rv;
    if (!rv) {
      const uint64_t a = o.res_off[r0], b = o.res_off[r0 + q->n];
      uint64_t j;
      size_t nd = 0;
      uint32_t t;
      for (j = a; j < b; j++) nd += o.res[j].strlen;
      if (grow((void **)&w->res_off, &w->cap_off, (size_t)q->n + 1, sizeof(uint64_t)) || grow((void **)&w->stat, &w->cap_stat, q->n, sizeof(smaltgpu_readstat)) ||
          grow((void **)&w->res, &w->cap_res, (size_t)(b - a) + 1, sizeof(smaltgpu_result)) || grow((void **)&w->dstr, &w->cap_dstr, nd + 1, 1)) q->rv = SMALTGPU_ENOMEM;
      else {
        for (t = 0; t <= q->n; t++) w->res_off[t] = o.res_off[r0 + t] - a;
        memcpy(w->stat, o.stat + r0, (size_t)q->n * sizeof(smaltgpu_readstat));
        for (j = a, nd = 0; j < b; j++) {
          w->res[j - a] = o.res[j];
          memcpy(w->dstr + nd, o.diffstr + o.res[j].stroffs, o.res[j].strlen);
          w->res[j - a].stroffs = (uint32_t)nd;
          nd += o.res[j].strlen;
        }
        w->n = q->n;
      }
    }
    r0 += q->n;
  }
}

/* A worker hit an error it is going to end the program with (the reference's ERRMSGNO exits): stop launching, let the batches
 * in flight finish, so that no kernel is running and no copy is under way when the process is torn down. */
void gpuCombineClose(void)
{
  struct CombQueue *d = &g_q;
  int busy;
  pthread_mutex_lock(&g_init);
  if (!d->init) { g_closing = 1; pthread_mutex_unlock(&g_init); return; }
  pthread_mutex_unlock(&g_init);
  pthread_mutex_lock(&d->mu);
  g_closing = 1;
  pthread_cond_broadcast(&d->cv);
  do {
    int u, v;
    busy = d->assembling;
    for (v = 0; v < d->ndev; v++) for (u = 0; u < COMB_NSLOT; u++) busy |= d->slot[v][u].busy;
    if (busy) pthread_cond_wait(&d->cv, &d->mu);
  } while (busy);
  pthread_mutex_unlock(&d->mu);
}

int gpuCombineSubmit(int ndev, const smaltgpu_index *const *ixs, const char *bases, const char *quals, const uint64_t *off, uint32_t n,
                     const smaltgpu_params *par, GpuCombOut *out, char *errbuf, size_t errcap)
{
  return gpuCombineSubmitCtx(ndev, ixs, bases, quals, off, n, par, NULL, out, errbuf, errcap);
}

int gpuCombineSubmitCtx(int ndev, const smaltgpu_index *const *ixs, const char *bases, const char *quals, const uint64_t *off, uint32_t n,
                        const smaltgpu_params *par, const GpuCombCtx *ctx, GpuCombOut *out, char *errbuf, size_t errcap)
{
  struct CombQueue *d = &g_q;
  CombReq req;
  if (ndev < 1 || ndev > COMB_MAXDEV || !ixs || !n) return SMALTGPU_EARG;
  pthread_once(&g_cfg_once, read_cfg);
  pthread_mutex_lock(&g_init);
  if (!d->init) {
    int i;
    pthread_mutex_init(&d->mu, NULL); pthread_cond_init(&d->cv, NULL);
    d->ndev = ndev;
    for (i = 0; i < ndev; i++) d->ix[i] = ixs[i];
    d->init = 1;
  }
  pthread_mutex_unlock(&g_init);
  req.bases = bases; req.quals = quals; req.off = off; req.n = n; req.par = par; req.out = out; req.ctx = ctx; req.done = 0; req.rv = 0; req.err[0] = 0;
  pthread_mutex_lock(&d->mu);
  while (d->npending >= COMB_MAXREQ && !g_closing) pthread_cond_wait(&d->cv, &d->mu);
  if (g_closing) { pthread_mutex_unlock(&d->mu); if (errbuf && errcap) snprintf(errbuf, errcap, "shutting down after an error in another worker"); return GPUCOMB_CLOSING; }
  d->pending[d->npending++] = &req;
  pthread_cond_broadcast(&d->cv);
  while (!req.done) {
    int sl = -1, dv = -1, u, v;
    if (g_closing) {                                       /* not taken by a leader yet: withdraw */
      int i, keep = 0, mine = 0;
      for (i = 0; i < d->npending; i++) { if (d->pending[i] == &req) mine = 1; else d->pending[keep++] = d->pending[i]; }
      if (mine) {
        d->npending = keep;
        pthread_cond_broadcast(&d->cv);
        pthread_mutex_unlock(&d->mu);
        if (errbuf && errcap) snprintf(errbuf, errcap, "shutting down after an error in another worker");
        return GPUCOMB_CLOSING;
      }
      pthread_cond_wait(&d->cv, &d->mu);                   /* a leader holds the request: its batch is finishing */
      continue;
    }
    /* (a waiting thread may lead a batch of others)  first slot of every device before the second slot of any */
    if (!d->assembling && d->npending > 0 && !g_closing)
      for (u = 0; u < g_nslot && sl < 0; u++) for (v = 0; v < d->ndev; v++) if (!d->slot[v][u].busy) { sl = u; dv = v; break; }
    if (sl >= 0) {
      CombReq *take[COMB_MAXREQ];
      int ntake = 0, i, rounds;
      uint32_t reads = 0;
      d->assembling = 1; d->slot[dv][sl].busy = 1;
      for (rounds = 0; rounds < 8; rounds++) {              /* let the other workers arrive: up to 8 x 250 us while requests keep coming */
        struct timespec ts;
        const int before = d->npending;
        clock_gettime(CLOCK_REALTIME, &ts);
        ts.tv_nsec += 250000;
        if (ts.tv_nsec >= 1000000000L) { ts.tv_sec++; ts.tv_nsec -= 1000000000L; }
        (void)pthread_cond_timedwait(&d->cv, &d->mu, &ts);
        if (d->npending == before) break;
      }
      const uint32_t maxreads = g_cfg.combine_reads;
      {                                                    /* the oldest request decides what the batch is made of: requests of its round kind, with the
                                                            * same parameters and the same presence of base qualities (a batch has ONE parameter set, and
                                                            * qualities for all reads or for none); the others keep their place in the queue */
        const CombReq *first = d->pending[0];
        const int kind = first->ctx ? first->ctx->kind : GPUCOMB_PLAIN;
        int keep = 0;
        for (i = 0; i < d->npending; i++) {
          const CombReq *q = d->pending[i];
          const int ki = q->ctx ? q->ctx->kind : GPUCOMB_PLAIN;
          const int alike = ki == kind && (q->quals != NULL) == (first->quals != NULL) && !memcmp(q->par, first->par, sizeof(*q->par));
          if (alike && !(ntake && reads + q->n > maxreads) && ntake < COMB_MAXREQ) { reads += q->n; take[ntake++] = d->pending[i]; }
          else d->pending[keep++] = d->pending[i];
        }
        d->npending = keep;
      }
      d->assembling = 0;                                     /* the next leader may collect while this batch runs */
      pthread_cond_broadcast(&d->cv);
      pthread_mutex_unlock(&d->mu);
      if (ntake) run_batch(&d->slot[dv][sl], d->ix[dv], take, ntake);
      pthread_mutex_lock(&d->mu);
      for (i = 0; i < ntake; i++) take[i]->done = 1;
      if (ntake) d->batches[dv]++;
      d->slot[dv][sl].busy = 0;
      pthread_cond_broadcast(&d->cv);
    } else pthread_cond_wait(&d->cv, &d->mu);
  }
  pthread_mutex_unlock(&d->mu);
  if (errbuf && errcap > 64 && !req.rv && getenv("SMALTGPU_COMBINE_STATS")) {    /* diagnostic: batches per device so far */
    int i, w = 0;
    for (i = 0; i < d->ndev && w + 24 < (int)errcap; i++) w += snprintf(errbuf + w, errcap - (size_t)w, "%sdev%d=%lu", i ? " " : "", i, d->batches[i]);
  }
  if (req.rv && errbuf && errcap) { strncpy(errbuf, req.err, errcap - 1); errbuf[errcap - 1] = 0; }
  return req.rv;
}
