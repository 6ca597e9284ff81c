/* smg_ktab.c -- host loader for FastK tables (format F).  See smg_ktab.h. */

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include <pthread.h>
#include <stdint.h>

#include "smg_ktab.h"

static int read_full(int fd, void *buf, size_t n)
{ uint8_t *p = (uint8_t *) buf;
  while (n > 0)
    { ssize_t r = read(fd, p, n > (1u << 30) ? (1u << 30) : n);
      if (r <= 0) return -1;
      p += r; n -= (size_t) r;
    }
  return 0;
}

/* ---- parallel part reader: the -T threads of the command line pread 32 MB chunks (the reference streams
        the parts through per-thread Kmer_Stream clones, PloidyPlot.c:1476-1479; here the whole table goes to
        the GPU, so host threads are only useful for getting it off the disk / out of the page cache) ---- */
#define SMG_CHUNK ((size_t) 32 << 20)

typedef struct { int fd; off_t off; uint8_t *dst; size_t len; } smg_chunk;
typedef struct { smg_chunk *ch; long n; long next; int failed; pthread_mutex_t mu; } smg_chunkq;

static void *chunk_worker(void *arg)
{ smg_chunkq *q = (smg_chunkq *) arg;
  for (;;)
    { long i;
      pthread_mutex_lock(&q->mu);
      i = q->next++;
      pthread_mutex_unlock(&q->mu);
      if (i >= q->n) break;
      { smg_chunk *c = q->ch + i;
        size_t done = 0;
        while (done < c->len)
          { ssize_t r = pread(c->fd, c->dst + done, c->len - done, c->off + (off_t) done);
            if (r <= 0) { q->failed = 1; break; }
            done += (size_t) r;
          }
      }
    }
  return NULL;
}

int smg_ktab_load(const char *name, smg_ktab *t, char *what)
{ return smg_ktab_load_mt(name, t, what, 1); }

static int ktab_open(const char *name, smg_ktab *t, char *what, int load, int nthreads);

static int64_t *g_ixbuf = NULL;
static int64_t  g_ixcap = 0;
static int      g_ixfilled = 0;

void smg_ktab_set_index_memory(int64_t *buf, int64_t cap_words, int filled)
{ g_ixbuf = buf; g_ixcap = buf ? cap_words : 0; g_ixfilled = buf ? filled : 0; }

int smg_ktab_open(const char *name, smg_ktab *t, char *what)
{ return ktab_open(name, t, what, 0, 1); }

int smg_ktab_load_mt(const char *name, smg_ktab *t, char *what, int nthreads)
{ return ktab_open(name, t, what, 1, nthreads); }

int smg_ktab_read(const smg_ktab *t, int part, int64_t first, int64_t nent, void *dst)
{ size_t len, done = 0;
  off_t  off;
  if (part < 0 || part >= t->nparts || first < 0 || nent < 0 || nent > t->part_nels[part]
      || first > t->part_nels[part] - nent)
    return -1;
  len = (size_t) nent * (size_t) t->pbyte;
  off = (off_t) 12 + (off_t) first * (off_t) t->pbyte;
  if (t->part[part] != NULL)
    { memcpy(dst, t->part[part] + (size_t) first * (size_t) t->pbyte, len); return 0; }
  if (t->fd == NULL || t->fd[part] < 0) return -1;
  while (done < len)
    { ssize_t r = pread(t->fd[part], (uint8_t *) dst + done, len - done, off + (off_t) done);
      if (r <= 0) return -1;
      done += (size_t) r;
    }
  return 0;
}

static int ktab_open(const char *name, smg_ktab *t, char *what, int load, int nthreads)
{ const char *slash = strrchr(name, '/');
  char  *dir, *root, *path;
  size_t len;
  int    fd, p, rc = SMG_KTAB_OK;
  int32_t hdr[4];

  memset(t, 0, sizeof(*t));
  if (slash) { dir = strndup(name, (size_t) (slash - name)); root = strdup(slash + 1); }
  else       { dir = strdup(".");                              root = strdup(name); }
  len = strlen(root);                       /* Root(name,".ktab"), gene_core.c Root()          */
  if (len > 5 && strcasecmp(root + len - 5, ".ktab") == 0) root[len - 5] = 0;
  path = (char *) malloc(strlen(dir) + strlen(root) + 64);
  if (!dir || !root || !path) { rc = SMG_KTAB_NOMEM; goto out; }

  sprintf(path, "%s/%s.ktab", dir, root);
  if (what) snprintf(what, 4096, "%s", path);
  fd = open(path, O_RDONLY);
  if (fd < 0) { rc = SMG_KTAB_NOSTUB; goto out; }
  if (read_full(fd, hdr, sizeof(hdr))) { close(fd); rc = SMG_KTAB_SHORT; goto out; }
  t->kmer = hdr[0]; t->nparts = hdr[1]; t->minval = hdr[2]; t->ibyte = hdr[3];
  /* k is bounded by what the engine takes (SMG_KTAB_MAX_KMER = SMG_MAX_KMER): the one-record scratch buffers of
     smg_ktab_entry / smg_ktab_find are sized from it, and a stub that claims k = 300 must not reach them */
  if (t->kmer < 1 || t->kmer > SMG_KTAB_MAX_KMER || t->nparts < 0 || t->nparts > (1 << 20) || t->ibyte < 1 || t->ibyte > 3
      || ((t->kmer + 3) >> 2) <= t->ibyte)           /* no suffix bytes: FastK never writes such a table (hbyte >= 1) */
    { close(fd); rc = SMG_KTAB_SHORT; goto out; }
  t->kbyte = (t->kmer + 3) >> 2;
  t->tbyte = t->kbyte + 2;
  t->pbyte = t->tbyte - t->ibyte;
  t->hbyte = t->kbyte - t->ibyte;
  t->ixlen = 1ll << (8 * t->ibyte);
  if (g_ixbuf != NULL && t->ixlen <= g_ixcap) { t->index = g_ixbuf; t->index_borrowed = 1; }
  else t->index = (int64_t *) malloc(sizeof(int64_t) * (size_t) t->ixlen);
  t->part = (uint8_t **) calloc((size_t) (t->nparts > 0 ? t->nparts : 1), sizeof(uint8_t *));
  t->part_nels = (int64_t *) calloc((size_t) (t->nparts > 0 ? t->nparts : 1), sizeof(int64_t));
  t->part_end = (int64_t *) calloc((size_t) (t->nparts > 0 ? t->nparts : 1), sizeof(int64_t));
  t->fd = (int *) malloc(sizeof(int) * (size_t) (t->nparts > 0 ? t->nparts : 1));
  if (!t->index || !t->part || !t->part_nels || !t->part_end || !t->fd)
    { close(fd); rc = SMG_KTAB_NOMEM; goto out; }
  for (p = 0; p < t->nparts; p++) t->fd[p] = -1;
  if (!(t->index_borrowed && g_ixfilled) && read_full(fd, t->index, sizeof(int64_t) * (size_t) t->ixlen))
    { close(fd); rc = SMG_KTAB_SHORT; goto out; }
  close(fd);

  { smg_chunkq q;
    long cap = 0;
    memset(&q, 0, sizeof(q));
    for (p = 1; p <= t->nparts && rc == SMG_KTAB_OK; p++)
      { int32_t km; int64_t n;
        struct stat sb;
        sprintf(path, "%s/.%s.ktab.%d", dir, root, p);
        if (what) snprintf(what, 4096, "%s", path);
        fd = open(path, O_RDONLY);
        if (fd < 0) { rc = SMG_KTAB_NOPART; break; }
        t->fd[p - 1] = fd;
        if (read_full(fd, &km, 4) || read_full(fd, &n, 8)) { rc = SMG_KTAB_SHORT; break; }
        if (km != t->kmer) { rc = SMG_KTAB_KMISMATCH; break; }
        if (n < 0 || n > (INT64_MAX - 12) / (int64_t) t->pbyte || t->nels > INT64_MAX - n
            || fstat(fd, &sb) != 0 || (int64_t) sb.st_size < 12 + n * (int64_t) t->pbyte)
          { rc = SMG_KTAB_SHORT; break; }
        t->part_nels[p - 1] = n;
        t->nels += n;
        t->part_end[p - 1] = t->nels;
        if (!load) continue;
        t->part[p - 1] = (uint8_t *) malloc((size_t) (n > 0 ? n : 1) * (size_t) t->pbyte);
        if (!t->part[p - 1]) { rc = SMG_KTAB_NOMEM; break; }
        { size_t bytes = (size_t) n * (size_t) t->pbyte, o;
          for (o = 0; o < bytes; o += SMG_CHUNK)
            { if (q.n >= cap)
                { cap = cap ? 2 * cap : 256;
                  q.ch = (smg_chunk *) realloc(q.ch, sizeof(smg_chunk) * (size_t) cap);
                  if (!q.ch) { rc = SMG_KTAB_NOMEM; break; }
                }
              q.ch[q.n].fd = fd; q.ch[q.n].off = (off_t) (12 + o); q.ch[q.n].dst = t->part[p - 1] + o;
              q.ch[q.n].len = bytes - o < SMG_CHUNK ? bytes - o : SMG_CHUNK;
              q.n++;
            }
        }
      }
    if (rc == SMG_KTAB_OK && q.n > 0)
      { pthread_t th[64];
        int nt = nthreads < 1 ? 1 : (nthreads > 64 ? 64 : nthreads), i, started = 0;
        if ((long) nt > q.n) nt = (int) q.n;
        pthread_mutex_init(&q.mu, NULL);
        for (i = 1; i < nt; i++)
          if (pthread_create(&th[started], NULL, chunk_worker, &q) == 0) started++;
        chunk_worker(&q);
        for (i = 0; i < started; i++) pthread_join(th[i], NULL);
        pthread_mutex_destroy(&q.mu);
        if (q.failed) rc = SMG_KTAB_SHORT;
      }
    free(q.ch);
    if (rc == SMG_KTAB_OK)
      { /* the index must be the cumulative entry count of the parts (a hostile or damaged stub would otherwise send
           the readers out of bounds): non-decreasing, ending at nels                                            */
        int64_t i, prev = 0;
        if (t->index_borrowed && g_ixfilled) prev = -1;        /* (checked by whoever filled it) */
        for (i = 0; prev >= 0 && i < t->ixlen && rc == SMG_KTAB_OK; i++)
          { if (t->index[i] < prev || t->index[i] > t->nels) rc = SMG_KTAB_SHORT;
            prev = t->index[i];
          }
        if (rc == SMG_KTAB_OK && t->ixlen > 0 && t->index[t->ixlen - 1] != t->nels) rc = SMG_KTAB_SHORT;
        if (rc != SMG_KTAB_OK && what) { sprintf(path, "%s/%s.ktab", dir, root); snprintf(what, 4096, "%s", path); }
      }
    if (load)                                   /* everything is in memory: the descriptors are not needed any more */
      for (p = 0; p < t->nparts; p++) if (t->fd[p] >= 0) { close(t->fd[p]); t->fd[p] = -1; }
  }
out:
  free(dir); free(root); free(path);
  if (rc != SMG_KTAB_OK) smg_ktab_free(t);
  return rc;
}

void smg_ktab_free(smg_ktab *t)
{ int p;
  if (t->part)
    for (p = 0; p < t->nparts; p++) free(t->part[p]);
  if (t->fd)
    for (p = 0; p < t->nparts; p++) if (t->fd[p] >= 0) close(t->fd[p]);
  free(t->part); free(t->part_nels); free(t->part_end); if (!t->index_borrowed) free(t->index); free(t->fd);
  memset(t, 0, sizeof(*t));
}

/* record i: a pointer into the loaded part, or the record fetched into buf (>= pbyte bytes) from disk */
static const uint8_t *record_at(const smg_ktab *t, int64_t i, uint8_t *buf)
{ int p = 0;
  int64_t base = 0;
  while (p < t->nparts - 1 && i >= t->part_end[p]) p++;
  if (p > 0) base = t->part_end[p - 1];
  if (t->part[p] != NULL) return t->part[p] + (size_t) (i - base) * (size_t) t->pbyte;
  if (smg_ktab_read(t, p, i - base, 1, buf) != 0) memset(buf, 0, (size_t) t->pbyte);
  return buf;
}

static int64_t prefix_of(const smg_ktab *t, int64_t i)      /* smallest p with index[p] > i */
{ int64_t lo = 0, hi = t->ixlen - 1;
  while (lo < hi)
    { int64_t m = (lo + hi) >> 1;
      if (t->index[m] <= i) lo = m + 1; else hi = m;
    }
  return lo;
}

void smg_ktab_entry(const smg_ktab *t, int64_t i, uint8_t *kmer_out, int *count_out)
{ uint8_t rb[SMG_KTAB_MAX_PBYTE];
  const uint8_t *r = record_at(t, i, rb);
  int64_t pre = prefix_of(t, i);
  int j;
  for (j = 0; j < t->ibyte; j++)
    kmer_out[j] = (uint8_t) ((pre >> (8 * (t->ibyte - 1 - j))) & 0xFF);
  memcpy(kmer_out + t->ibyte, r, (size_t) t->hbyte);
  if (count_out) *count_out = r[t->hbyte] | (r[t->hbyte + 1] << 8);
}

int64_t smg_ktab_find(const smg_ktab *t, const uint8_t *kmer)
{ int64_t m = 0, lo, hi;
  uint8_t rb[SMG_KTAB_MAX_PBYTE];
  int j;
  for (j = 0; j < t->ibyte; j++) m = (m << 8) | kmer[j];
  lo = m == 0 ? 0 : t->index[m - 1];
  hi = t->index[m];
  if (hi > t->nels) hi = t->nels;
  while (lo < hi)
    { int64_t mid = (lo + hi) >> 1;
      if (memcmp(record_at(t, mid, rb), kmer + t->ibyte, (size_t) t->hbyte) < 0) lo = mid + 1; else hi = mid;
    }
  if (lo < t->index[m] && lo < t->nels
      && memcmp(record_at(t, lo, rb), kmer + t->ibyte, (size_t) t->hbyte) == 0)
    return lo;
  return -1;
}

static void revcomp_packed(const uint8_t *x, uint8_t *out, int k, int kbyte)
{ int i;
  memset(out, 0, (size_t) kbyte);
  for (i = 0; i < k; i++)
    { int b = (x[i >> 2] >> (6 - 2 * (i & 3))) & 3;
      int j = k - 1 - i;
      out[j >> 2] |= (uint8_t) ((3 - b) << (6 - 2 * (j & 3)));
    }
}

/* One worker of the trim probe: the smallest count >= 1 among the entries [a, b) of the table (0x8000: none).
   The reference builds a histogram of these counts and takes its first non-empty bin above 0 (PloidyPlot.c:1169-1197):
   the decision needs the minimum only.  A worker stops as soon as it has seen a count below the threshold -- the
   decision is made then, whatever the rest holds.                                                                    */
typedef struct { const smg_ktab *t; int64_t a, b; int ethresh, minc, bad; } ProbeJob;

static void *probe_worker(void *arg)
{ ProbeJob *j = (ProbeJob *) arg;
  const smg_ktab *t = j->t;
  const int64_t blk = 65536;
  uint8_t *buf = (uint8_t *) malloc((size_t) blk * (size_t) t->pbyte);
  int p, minc = 0x8000;
  j->minc = 0x8000; j->bad = 0;
  if (buf == NULL) { j->bad = 2; return NULL; }
  for (p = 0; p < t->nparts && !j->bad && minc >= j->ethresh; p++)
    { const int64_t pb = p ? t->part_end[p - 1] : 0, pe = t->part_end[p];
      const int64_t a = j->a > pb ? j->a : pb, b = j->b < pe ? j->b : pe;
      int64_t i;
      for (i = a; i < b && minc >= j->ethresh; i += blk)
        { const int64_t m = b - i < blk ? b - i : blk;
          int64_t q;
          const uint8_t *r = buf;
          if (t->part[p] != NULL) r = t->part[p] + (size_t) (i - pb) * (size_t) t->pbyte;
          else if (smg_ktab_read(t, p, i - pb, m, buf) != 0) { j->bad = 1; break; }     /* a partial look would call an
                                                                                            unreadable table "trimmed" */
          r += t->hbyte;
          for (q = 0; q < m; q++, r += t->pbyte)
            { const int c = r[0] | (r[1] << 8);        /* (the reference reads an int16: counts above 32767 are outside
                                                           what it and FastK support, they are ignored here)           */
              if (c >= 1 && c < minc) minc = c;
            }
        }
    }
  free(buf);
  j->minc = minc;
  return NULL;
}

int smg_ktab_examine(const smg_ktab *t, int ethresh, int *trim, int *symm)
{ return smg_ktab_examine_mt(t, ethresh, 1, trim, symm); }

int smg_ktab_examine_mt(const smg_ktab *t, int ethresh, int nthreads, int *trim, int *symm)
{ int64_t frst, last;
  int     bad = 0, nz = 0x8000, w, started = 0;
  ProbeJob job[16];
  pthread_t th[16];

  *trim = 0; *symm = 0;
  /* "Histogram of middle 100M counts and see if trimmed to ETHRESH", PloidyPlot.c:1169-1197 */
  if (t->nels + 3 < 100000000) { frst = 0; last = t->nels; }
  else { frst = t->nels / 2 - 50000000; last = t->nels / 2 + 50000000; }
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 16) nthreads = 16;
  if (last - frst < 1000000) nthreads = 1;
  for (w = 0; w < nthreads; w++)
    { job[w].t = t; job[w].ethresh = ethresh;
      job[w].a = frst + (last - frst) / nthreads * w;
      job[w].b = w == nthreads - 1 ? last : frst + (last - frst) / nthreads * (w + 1);
    }
  for (w = 1; w < nthreads; w++)
    { if (pthread_create(&th[started], NULL, probe_worker, &job[w]) == 0) started++;
      else { int v; for (v = w; v < nthreads; v++) probe_worker(&job[v]); break; }      /* (no thread: do it here) */
    }
  probe_worker(&job[0]);
  for (w = 0; w < started; w++) pthread_join(th[w], NULL);
  for (w = 0; w < nthreads; w++)
    { if (job[w].bad > bad) bad = job[w].bad;
      if (job[w].minc < nz) nz = job[w].minc;
    }
  if (bad) return bad == 2 ? SMG_KTAB_NOMEM : SMG_KTAB_SHORT;
  *trim = (nz >= ethresh);

  /* "Walk to a non-palindromic k-mer and see if its complement is in T", PloidyPlot.c:1199-1229.
     Net effect of that loop (including its quirk for a self-complementary entry #1, see
     oracle/hetmers_oracle.c): symm = the complement of entry #1 is in the table.            */
  *symm = 0;
  if (t->nels > 1)
    { uint8_t *x = (uint8_t *) malloc((size_t) t->kbyte * 2);
      if (x)
        { smg_ktab_entry(t, 1, x, NULL);
          revcomp_packed(x, x + t->kbyte, t->kmer, t->kbyte);
          *symm = smg_ktab_find(t, x + t->kbyte) >= 0;
          free(x);
        }
      else return SMG_KTAB_NOMEM;
    }
  return SMG_KTAB_OK;
}
