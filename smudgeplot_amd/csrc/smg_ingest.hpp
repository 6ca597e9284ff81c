// smg_ingest.hpp -- table records from a `smg_table_source` straight into HBM.
//
// The reference reads a table through seekable streams, 1024 records per read(2) (More_Kmer_Stream,
// libfastk.c:759-784); round 1 read every part into host memory first and copied it with pageable hipMemcpy's
// (1.1 s of a 1.8 s run on a 7.2 GB table, host RSS = table size).  Here `nthreads` readers pull pieces of
// ING_CHUNK bytes through the source's read callback (pread for a table on disk) into a ring of pinned buffers;
// every piece is sent with hipMemcpyAsync as soon as it is read, so reading piece c+1 overlaps the DMA of piece c,
// and the host holds nthreads + 2 pieces at any time.  With a piece hook (ingest_decoded) the copy lands in a small
// ring on the DEVICE and the hook's kernel -- the decode of just that piece -- is queued behind it: decoding hides
// behind PCIe and the device never holds the raw records of the whole table (round 2 decoded the complete shard in one
// launch after the last copy: 36-77 ms per 1e9 entries in the open).  The hook's kernels have a stream of their own,
// tied to the copies by events: on ONE stream every copy -> kernel -> copy hand-over between the DMA engine and the
// compute queue cost ~25 us, 45 ms over the 860 pieces of a 7.2 GB table (33 GB/s instead of the 44 GB/s of bare copies).
// Included by smg_hetmers.hip.

#pragma once
#include <pthread.h>

#define ING_CHUNK  ((size_t) 8 << 20)
#define ING_MAXT   64

struct IngPiece { int part; int64_t first, nent; size_t dst; };

// called after the copy of a piece has been queued, with the stream for the hook's kernels (ordered behind that copy): d_piece = its records on the device,
// first = its first entry counted from the start of the ingested range, nent = its entries.  0 = success.
typedef int (*IngestHook)(void *ctx, hipStream_t stream, const uint8_t *d_piece, int64_t first, int64_t nent);      // (stream: the hook's own)

struct Ingest
{ const smg_table_source *src;
  int          pbyte, device, nslots;
  uint8_t     *d_rec;                // whole-range destination, or NULL: pieces go to d_ring and to the hook
  uint8_t     *d_ring[ING_MAXT + 2];
  IngestHook   hook; void *hook_ctx;
  hipStream_t  stream, kstream;      // copies; the hook's kernels
  hipStream_t  cstream[4]; int ncs;  // the copy streams (piece c goes to cstream[c % ncs]; stream == cstream[0])
  hipEvent_t   kev[ING_MAXT + 2];    // the hook's kernel on device slot s has finished
  IngPiece    *piece; long npiece, next;
  char        *sent;                 // piece c has been handed to the copy stream
  uint8_t     *ring[ING_MAXT + 2];
  hipEvent_t   ev[ING_MAXT + 2];
  pthread_mutex_t mu; pthread_cond_t cv;
  volatile int failed;               // 1 read error, 2 HIP error
};

static void *ingest_worker(void *arg)
{ Ingest *g = (Ingest *) arg;
  if (hipSetDevice(g->device) != hipSuccess) { g->failed = 2; return NULL; }
  for (;;)
    { pthread_mutex_lock(&g->mu);
      const long c = g->next++;
      pthread_mutex_unlock(&g->mu);
      if (c >= g->npiece) break;
      const int slot = (int) (c % g->nslots);
      int bad = g->failed;
      if (!bad && c >= g->nslots)
        { // the slot is free once the copy of the piece that used it last has finished
          pthread_mutex_lock(&g->mu);
          while (!g->sent[c - g->nslots]) pthread_cond_wait(&g->cv, &g->mu);
          pthread_mutex_unlock(&g->mu);
          if (hipEventSynchronize(g->ev[slot]) != hipSuccess) bad = 2;
        }
      const IngPiece &p = g->piece[c];
      if (!bad && g->src->read(g->src->ctx, p.part, p.first, p.nent, g->ring[slot]) != 0) bad = 1;
      uint8_t *dst = g->d_rec ? g->d_rec + p.dst : g->d_ring[slot];
      hipStream_t cs = g->cstream[c % g->ncs];
      // (a device slot is written again nslots pieces on: that copy waits for the kernel that read the slot last)
      if (!bad && g->hook && !g->d_rec && c >= g->nslots && hipStreamWaitEvent(cs, g->kev[slot], 0) != hipSuccess) bad = 2;
      if (!bad && (hipMemcpyAsync(dst, g->ring[slot], (size_t) p.nent * g->pbyte, hipMemcpyHostToDevice, cs) != hipSuccess
                   || hipEventRecord(g->ev[slot], cs) != hipSuccess))
        bad = 2;
      if (!bad && g->hook
          && (hipStreamWaitEvent(g->kstream, g->ev[slot], 0) != hipSuccess
              || g->hook(g->hook_ctx, g->kstream, dst, (int64_t) (p.dst / (size_t) g->pbyte), p.nent) != 0
              || hipEventRecord(g->kev[slot], g->kstream) != hipSuccess))
        bad = 2;
      pthread_mutex_lock(&g->mu);
      if (bad && !g->failed) g->failed = bad;
      g->sent[c] = 1;                              // (also on failure: nobody may wait for ever)
      pthread_cond_broadcast(&g->cv);
      pthread_mutex_unlock(&g->mu);
    }
  return NULL;
}

// records of the entries [lo, hi) of the table -> d_rec[0 ..), on `device`.  0, or a negative SMG_E* code.
static int ingest_records(const smg_table_source *src, int pbyte, int64_t lo, int64_t hi, uint8_t *d_rec, int device,
                          int nthreads, double *seconds, char *errbuf, size_t errlen, IngestHook hook = NULL, void *hook_ctx = NULL,
                          hipEvent_t after = NULL /* the copy stream waits for this event first (what the hook's kernels read) */)
{ Ingest g;
  memset(&g, 0, sizeof(g));
  if (nthreads < 1) nthreads = 4;
  if (nthreads > ING_MAXT) nthreads = ING_MAXT;
  g.src = src; g.pbyte = pbyte; g.device = device; g.d_rec = d_rec; g.hook = hook; g.hook_ctx = hook_ctx;
  if (!d_rec && !hook) return fail(errbuf, errlen, SMG_EINVAL, "ingest: no destination%s");
  // (whole decode tiles per piece: the pieces of a part start on multiples of 1024 entries of the part)
  const int64_t per = (int64_t) (ING_CHUNK / (size_t) pbyte) / 1024 * 1024;
  // pieces: part by part, `per` entries each
  long cap = 0;
  { int64_t base = 0;
    for (int p = 0; p < src->nparts; p++)
      { const int64_t pn = src->part_nels[p];
        const int64_t a = lo > base ? lo : base, b = hi < base + pn ? hi : base + pn;
        if (a < b) cap += (long) ((b - a + per - 1) / per);
        base += pn;
      }
  }
  if (cap == 0) { if (seconds) *seconds = 0; return SMG_OK; }
  g.piece = (IngPiece *) malloc(sizeof(IngPiece) * (size_t) cap);
  g.sent = (char *) calloc((size_t) cap, 1);
  if (!g.piece || !g.sent) { free(g.piece); free(g.sent); return fail(errbuf, errlen, SMG_ENOMEM, "out of host memory%s"); }
  { int64_t base = 0; size_t dst = 0;
    for (int p = 0; p < src->nparts; p++)
      { const int64_t pn = src->part_nels[p];
        const int64_t a = lo > base ? lo : base, b = hi < base + pn ? hi : base + pn;
        for (int64_t i = a; i < b; i += per)
          { IngPiece &q = g.piece[g.npiece++];
            q.part = p; q.first = i - base; q.nent = b - i < per ? b - i : per; q.dst = dst;
            dst += (size_t) q.nent * pbyte;
          }
        base += pn;
      }
  }
  if ((long) nthreads > g.npiece) nthreads = (int) g.npiece;
  g.nslots = nthreads + 2;
  int rc = SMG_OK, made = 0, evs = 0;
  hipEvent_t t0 = NULL, t1 = NULL;
  g.ncs = 1;
  { const char *v = tune_env("SMG_COPY_STREAMS"); if (v && atoi(v) >= 1 && atoi(v) <= 4) g.ncs = atoi(v); }      // (tuning)
  for (int i = 0; i < g.ncs && rc == SMG_OK; i++)
    if (hipStreamCreateWithFlags(&g.cstream[i], hipStreamNonBlocking) != hipSuccess)
      rc = fail(errbuf, errlen, SMG_ENODEV, "cannot create the copy stream%s");
  g.stream = g.cstream[0];
  if (rc == SMG_OK && hook && hipStreamCreateWithFlags(&g.kstream, hipStreamNonBlocking) != hipSuccess)
    rc = fail(errbuf, errlen, SMG_ENODEV, "cannot create the copy stream%s");
  for (; rc == SMG_OK && made < g.nslots; made++)
    if (hipHostMalloc((void **) &g.ring[made], ING_CHUNK) != hipSuccess)
      { rc = fail(errbuf, errlen, SMG_ENOMEM, "cannot allocate the pinned staging buffers%s"); break; }
  int dmade = 0;
  for (; rc == SMG_OK && !d_rec && dmade < g.nslots; dmade++)
    if (hipMalloc((void **) &g.d_ring[dmade], ING_CHUNK) != hipSuccess)
      { rc = fail(errbuf, errlen, SMG_ENOMEM, "cannot allocate the device staging ring%s"); break; }
  for (; rc == SMG_OK && evs < g.nslots; evs++)
    if (hipEventCreateWithFlags(&g.ev[evs], hipEventDisableTiming) != hipSuccess)
      { rc = fail(errbuf, errlen, SMG_ENODEV, "cannot create an event%s"); break; }
  int kevs = 0;
  for (; rc == SMG_OK && hook && kevs < g.nslots; kevs++)
    if (hipEventCreateWithFlags(&g.kev[kevs], hipEventDisableTiming) != hipSuccess)
      { rc = fail(errbuf, errlen, SMG_ENODEV, "cannot create an event%s"); break; }
  if (rc == SMG_OK && (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess))
    rc = fail(errbuf, errlen, SMG_ENODEV, "cannot create an event%s");
  if (rc == SMG_OK && after && hook && hipStreamWaitEvent(g.kstream, after, 0) != hipSuccess)
    rc = fail(errbuf, errlen, SMG_ENODEV, "cannot order the copy stream%s");
  for (int i = 0; i < g.ncs && rc == SMG_OK && after && !hook; i++)
    if (hipStreamWaitEvent(g.cstream[i], after, 0) != hipSuccess) rc = fail(errbuf, errlen, SMG_ENODEV, "cannot order the copy stream%s");
  if (rc == SMG_OK)
    { pthread_mutex_init(&g.mu, NULL); pthread_cond_init(&g.cv, NULL);
      struct timespec a, b;
      clock_gettime(CLOCK_MONOTONIC, &a);
      pthread_t th[ING_MAXT];
      int started = 0;
      for (int i = 1; i < nthreads; i++)
        if (pthread_create(&th[started], NULL, ingest_worker, &g) == 0) started++;
      ingest_worker(&g);
      for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
      for (int i = 0; i < g.ncs; i++)
        if (hipStreamSynchronize(g.cstream[i]) != hipSuccess && !g.failed) g.failed = 2;
      if (g.kstream && hipStreamSynchronize(g.kstream) != hipSuccess && !g.failed) g.failed = 2;
      clock_gettime(CLOCK_MONOTONIC, &b);
      if (seconds) *seconds = (double) (b.tv_sec - a.tv_sec) + 1e-9 * (double) (b.tv_nsec - a.tv_nsec);
      pthread_mutex_destroy(&g.mu); pthread_cond_destroy(&g.cv);
      if (g.failed == 1) rc = fail(errbuf, errlen, SMG_EFORMAT, "cannot read the table records (file truncated or unreadable)%s");
      else if (g.failed) rc = fail(errbuf, errlen, SMG_ENODEV, "host to device copy failed%s");
    }
  if (t0) hipEventDestroy(t0);
  if (t1) hipEventDestroy(t1);
  for (int i = 0; i < evs; i++) hipEventDestroy(g.ev[i]);
  for (int i = 0; i < kevs; i++) hipEventDestroy(g.kev[i]);
  if (g.kstream) hipStreamDestroy(g.kstream);
  for (int i = 0; i < made; i++) hipHostFree(g.ring[i]);
  for (int i = 0; i < dmade; i++) hipFree(g.d_ring[i]);
  for (int i = 0; i < 4; i++) if (g.cstream[i]) hipStreamDestroy(g.cstream[i]);
  free(g.piece); free(g.sent);
  return rc;
}

// a table view (everything in host memory) as a source
struct ViewCtx { const smg_table_view *tv; int pbyte; };
static int view_read(void *ctx, int part, int64_t first, int64_t nent, void *dst)
{ const ViewCtx *v = (const ViewCtx *) ctx;
  if (part < 0 || part >= v->tv->nparts || first < 0 || first + nent > v->tv->part_nels[part]) return -1;
  memcpy(dst, v->tv->part_data[part] + (size_t) first * v->pbyte, (size_t) nent * v->pbyte);
  return 0;
}
static void view_source(const smg_table_view *tv, ViewCtx *ctx, smg_table_source *src)
{ ctx->tv = tv; ctx->pbyte = ((tv->kmer + 3) >> 2) + 2 - tv->ibyte;
  src->kmer = tv->kmer; src->ibyte = tv->ibyte; src->nparts = tv->nparts; src->minval = tv->minval; src->nels = tv->nels;
  src->part_nels = tv->part_nels; src->prefix_index = tv->prefix_index;
  src->read = view_read; src->ctx = ctx; src->host_threads = 4;
}
