// smg_multi.hpp -- several GPUs of one node behind the C ABI (smg_opts.ngpus > 1): what the `hetmers`
// executable uses when SMUDGEPLOT_GPUS=N is set, so that the unchanged `smudgeplot hetmers` CLI scales
// over the node.  Included at the end of smg_hetmers.hip (uses the engine's internals).
//
// One host thread per GPU, each with its own engine.  The conditioned table is cut into contiguous PREFIX
// shards on window-block boundaries (every scanned position is shard local; the reference's analogue is
// the prefix-subtree task list of small_window, PloidyPlot.c:1040-1084).  Exchanges:
//   1. complement requests: grouped by destination on the device (smg_engine_route), then one
//      send/recv pair per peer inside ONE RCCL group call (an all-to-all over the xGMI links);
//   2. ONE ncclAllReduce(SUM, int64[1001*501]) of the per-GPU histograms.
// The symmetry proof (missing counts + fingerprint residues) is a few words per GPU and is reduced on the
// host: all ranks live in one process.
//
// RCCL is resolved with dlopen at the first multi-GPU call: the single-GPU path (and a process that
// already carries another RCCL, e.g. PyTorch's) never loads it.
//
// SMG_VIRTUAL_SHARDS=N (tests): N shards on ONE device, the exchanges done with device-to-device copies
// and a host-side sum -- exercises the cutting, the ranged decode, the routing and the proof on a 1-GPU box.
// The multi-process equivalent (one process per GPU, torch.distributed) is smudgeplot_amd/sharded.py.

#pragma once
#include <pthread.h>
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and enums only: the functions are looked up at run time

#define SMG_MAXGPU 16

struct RcclApi
{ void *lib;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*GroupStart)(void);
  ncclResult_t (*GroupEnd)(void);
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  const char  *(*GetErrorString)(ncclResult_t);
};

static bool rccl_load(RcclApi *a, char *errbuf, size_t errlen)
{ memset(a, 0, sizeof(*a));
  const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" };
  // an RCCL that the process has loaded already (a Python caller's torch brings its own) is the one to use: two copies of the
  // library in one process take each other's state down at exit ("double free or corruption" after an otherwise clean run)
  for (unsigned i = 0; i < 2 && !a->lib; i++) a->lib = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
  // (RTLD_LOCAL: this copy's symbols are reached through dlsym alone -- with RTLD_GLOBAL a torch that loads ITS librccl later
  //  binds part of it to this one)
  for (unsigned i = 0; i < sizeof(names) / sizeof(names[0]) && !a->lib; i++) a->lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!a->lib) { fail(errbuf, errlen, SMG_ENODEV, "cannot load RCCL (librccl.so) for the multi-GPU run%s"); return false; }
#define SYM(field, name) *(void **) (&a->field) = dlsym(a->lib, name); if (!a->field) { fail(errbuf, errlen, SMG_ENODEV, "RCCL symbol missing: %s", name); return false; }
  SYM(CommInitAll, "ncclCommInitAll") SYM(CommDestroy, "ncclCommDestroy") SYM(GroupStart, "ncclGroupStart")
  SYM(GroupEnd, "ncclGroupEnd") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(AllReduce, "ncclAllReduce")
  SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  return true;
}

struct MultiCtx
{ int  n;                                   // shards == threads
  bool virt;                                // all shards on one device, no RCCL
  int  devs[SMG_MAXGPU];
  const smg_table_source *tv;
  int  symcheck, condition, ethresh, W, pbyte, io_threads;
  int64_t cut[SMG_MAXGPU + 1];              // entry ranges of the shards in the input table
  pthread_barrier_t bar;
  volatile int failed;
  // published by rank r, read by the others after a barrier
  u64      first[SMG_MAXGPU][4];
  int64_t  nshard[SMG_MAXGPU];
  int64_t  counts[SMG_MAXGPU][SMG_MAXGPU];  // counts[src][dst], records
  uint64_t *send[SMG_MAXGPU];
  const uint32_t *bmap[SMG_MAXGPU];         // candidate block maps of the shards (request filter), bm_bits[r] = 0: none
  int      bm_bits[SMG_MAXGPU];
  int64_t  missing[SMG_MAXGPU];
  u64      fp[SMG_MAXGPU][4];
  int64_t *h_plot[SMG_MAXGPU];              // virtual mode: per-shard histograms on the host
  smg_stats st[SMG_MAXGPU];
  int      rc[SMG_MAXGPU];
  char     err[SMG_MAXGPU][256];
  int      rw;
  int64_t *symm_hist[SMG_MAXGPU];           // symmetrise: entries / complements per leading `symm_bits` bits, per shard (host)
  int      symm_bits;
  pthread_mutex_t big_mu;                   // virtual shards: the sort of the symmetrise step runs one shard at a time
  RcclApi  api;
  ncclComm_t comm[SMG_MAXGPU];
  int64_t *plot;                            // result (host), written by rank 0
  const uint16_t *labels;                   // extract leg (or NULL): labels of the annotated pixels (host, SMG_PLOT_CELLS)
  uint64_t *h_rec[SMG_MAXGPU];              //   records of shard r (malloc'ed), nrec[r] of them
  int64_t   nrec[SMG_MAXGPU];
  Tab      gtab[SMG_MAXGPU];                // general path over virtual shards: the shards' tables ...
  TabSet  *d_set;                           // ... collected on the device by rank 0
  int      general;                         // 1: the table failed the proof and the shards run the general path together
};

struct MultiArg { MultiCtx *c; int r; };

static int decode_at(smg_engine *e, int kmer, int ibyte, int64_t nels, int64_t ibase, const uint8_t *d_records,
                     const int64_t *d_prefix_index, char *errbuf, size_t errlen);

// Every rank takes part in a collective (or in the peer copies of the virtual mode) or none does: the failure flag is
// read between two barriers, where nobody writes it, so all ranks see the same value.  (A rank that failed after
// the previous barrier would otherwise skip its ncclSend/ncclRecv while its peers block in theirs for ever.)
static bool multi_agree(MultiCtx *c)
{ pthread_barrier_wait(&c->bar);
  const bool ok = !c->failed;
  pthread_barrier_wait(&c->bar);
  return ok;
}

#define MFAIL(code, msg) do { c->rc[r] = fail(c->err[r], sizeof(c->err[r]), code, msg "%s"); c->failed = 1; } while (0)
#define MOK (!c->failed)

__global__ void km_or_words(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, int64_t n)
{ for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x)
    dst[i] |= src[i];
}

// request filter across shards: OR the word ranges that the k-mer ranges of the shards cover (neighbours share their
// boundary word) into one map on this shard's device, then drop the requests whose target block holds no candidate
static int multi_filter(MultiCtx *c, int r, smg_engine *e, const u64 *splitters, char *eb, size_t el)
{ const int n = c->n, bits = c->bm_bits[r];
  for (int s = 0; s < n; s++) if (c->bm_bits[s] != bits) return SMG_OK;       // (cannot happen: a function of k and the proof)
  if (!bits) return SMG_OK;
  const int two = e->bm2;                            // 64-bit map words (two-bit map): twice the 32-bit words per id range
  const int64_t nwords = (((1ll << bits) + 31) >> 5) << two;
  int64_t wlo[SMG_MAXGPU], wlen[SMG_MAXGPU], width = 1;
  for (int s = 0; s < n; s++)
    { const u64 a = s == 0 ? 0 : splitters[(size_t) (s - 1) * c->W] >> (64 - bits);
      const u64 b = s == n - 1 ? (1ull << bits) - 1 : splitters[(size_t) s * c->W] >> (64 - bits);
      wlo[s] = (int64_t) (a >> 5) << two;
      wlen[s] = (((int64_t) (b >> 5) << two) - wlo[s]) + (1 << two);
      if (wlen[s] < 1) wlen[s] = 1;
      if (wlen[s] > width) width = wlen[s];
    }
  uint32_t *full = NULL, *tmp = NULL;
  int rc = SMG_OK;
  if (hipMalloc(&full, (size_t) nwords * 4) != hipSuccess || hipMalloc(&tmp, (size_t) width * 4) != hipSuccess)
    rc = fail(eb, el, SMG_ENOMEM, "out of device memory for the block map exchange%s");
  if (rc == SMG_OK && hipMemsetAsync(full, 0, (size_t) nwords * 4, e->stream) != hipSuccess) rc = fail(eb, el, SMG_ENODEV, "memset failed%s");
  for (int s = 0; s < n && rc == SMG_OK; s++)
    { hipError_t he = c->virt || c->devs[s] == c->devs[r]
                        ? hipMemcpyAsync(tmp, c->bmap[s] + wlo[s], (size_t) wlen[s] * 4, hipMemcpyDeviceToDevice, e->stream)
                        : hipMemcpyPeerAsync(tmp, c->devs[r], c->bmap[s] + wlo[s], c->devs[s], (size_t) wlen[s] * 4, e->stream);
      if (he != hipSuccess) { rc = fail(eb, el, SMG_ENODEV, "block map copy failed: %s", hipGetErrorString(he)); break; }
      unsigned nb = (unsigned) ((wlen[s] + 255) / 256);
      if (nb > 4096) nb = 4096;
      hipLaunchKernelGGL(km_or_words, dim3(nb), dim3(256), 0, e->stream, full + wlo[s], tmp, wlen[s]);
    }
  if (rc == SMG_OK) rc = smg_engine_filter(e, full, NULL, eb, el);
  hipStreamSynchronize(e->stream);
  hipFree(full); hipFree(tmp);
  return rc;
}

// records of `rw` words: what rank r grouped for rank p in c->send[r] (counts[r][p], destination-major) -> recv of rank p.
// Call between two multi_agree()/barriers: every rank takes part.  Sets c->failed / c->rc[r] on error.
static void multi_exchange(MultiCtx *c, int r, smg_engine *e, uint64_t *recv, int rw, const char *what)
{ const int n = c->n;
  char *eb = c->err[r]; const size_t el = sizeof(c->err[r]);
  if (c->virt)
    { int64_t roff = 0;
      for (int s = 0; s < n && !c->failed; s++)
        { int64_t soff = 0;
          for (int d = 0; d < r; d++) soff += c->counts[s][d];
          const int64_t cnt = c->counts[s][r];
          if (cnt && hipMemcpy(recv + roff * rw, c->send[s] + soff * rw, sizeof(uint64_t) * (size_t) cnt * rw,
                               hipMemcpyDeviceToDevice) != hipSuccess)
            { c->rc[r] = fail(eb, el, SMG_ENODEV, "device to device copy failed (%s)", what); c->failed = 1; }
          roff += cnt;
        }
      return;
    }
  ncclResult_t nr = c->api.GroupStart();
  int64_t soff = 0, roff = 0;
  for (int p = 0; p < n && nr == ncclSuccess; p++)
    { if (c->counts[r][p])
        nr = c->api.Send(c->send[r] + soff * rw, (size_t) c->counts[r][p] * rw, ncclUint64, p, c->comm[r], e->stream);
      if (nr == ncclSuccess && c->counts[p][r])
        nr = c->api.Recv(recv + roff * rw, (size_t) c->counts[p][r] * rw, ncclUint64, p, c->comm[r], e->stream);
      soff += c->counts[r][p]; roff += c->counts[p][r];
    }
  const ncclResult_t ne = c->api.GroupEnd();
  if (nr == ncclSuccess) nr = ne;
  if (nr != ncclSuccess || hipStreamSynchronize(e->stream) != hipSuccess)
    { c->rc[r] = fail(eb, el, SMG_ENODEV, "RCCL exchange failed: %s", nr != ncclSuccess ? c->api.GetErrorString(nr) : "stream error");
      c->failed = 1;
    }
}

static void *multi_worker(void *argp)
{ MultiArg *arg = (MultiArg *) argp;
  MultiCtx *c = arg->c;
  const int r = arg->r, n = c->n, W = c->W;
  const smg_table_source *tv = c->tv;
  smg_engine *e = NULL;
  int64_t *d_index = NULL, *d_plot = NULL;
  uint64_t *recv = NULL;
  u64 splitters[SMG_MAXGPU * 4];
  char *eb = c->err[r]; const size_t el = sizeof(c->err[r]);
  c->rc[r] = SMG_OK; c->err[r][0] = 0;
  c->send[r] = NULL; c->h_plot[r] = NULL; c->nshard[r] = 0; c->missing[r] = 0;
  memset(c->fp[r], 0, sizeof(c->fp[r])); memset(c->first[r], 0xFF, sizeof(c->first[r]));

  // ---- shard on the device -------------------------------------------------------------------------------
  if (hipSetDevice(c->devs[r]) != hipSuccess) MFAIL(SMG_ENODEV, "cannot select HIP device");
  if (MOK && !(e = smg_engine_create(c->devs[r], NULL, eb, el))) { c->rc[r] = SMG_ENODEV; c->failed = 1; }
  if (MOK && !c->virt && n >= 8) smg_engine_set_blockmap_bits(e, 29);     // (the exchanged map: see TorchEngine.pass1 in sharded.py)
  const int64_t lo = c->cut[r], hi = c->cut[r + 1], ns = hi - lo;
  const size_t ixbytes = sizeof(int64_t) << (8 * tv->ibyte);
  if (MOK)
    { if (hipMalloc(&d_index, ixbytes) != hipSuccess || hipMalloc(&d_plot, sizeof(int64_t) * SMG_PLOT_CELLS) != hipSuccess)
        MFAIL(SMG_ENOMEM, "out of device memory for the table shard");
    }
  if (MOK && hipMemcpy(d_index, tv->prefix_index, ixbytes, hipMemcpyHostToDevice) != hipSuccess)
    MFAIL(SMG_ENODEV, "host to device copy failed");
  if (MOK && (c->rc[r] = decode_begin(e, tv->kmer, tv->ibyte, ns, eb, el))) c->failed = 1;
  if (MOK)
    { // the shard's record range, piece by piece: copied, and decoded behind its copy (entry lo of the table = entry 0 here)
      DecodeHook hk; hk.e = e; hk.d_index = d_index; hk.ibyte = tv->ibyte; hk.ibase = lo;
      if ((c->rc[r] = ingest_records(tv, c->pbyte, lo, hi, NULL, c->devs[r], c->io_threads, NULL, eb, el, decode_hook, &hk))) c->failed = 1;
    }
  // a shard that is taken as it is keeps the table's prefix index as its look-up directory (bucket starts relative to entry lo)
  if (MOK && !c->condition && (c->rc[r] = smg_engine_set_prefix_index(e, d_index, tv->ibyte, lo, eb, el))) c->failed = 1;
  if (MOK && !c->condition && hipStreamSynchronize(e->stream) != hipSuccess) MFAIL(SMG_ENODEV, "index conversion failed");
  if (d_index) { hipFree(d_index); d_index = NULL; }
  if (MOK && (c->condition & SMG_COND_TRIM))
    { int64_t nn = 0;
      if ((c->rc[r] = smg_engine_condition(e, c->ethresh, 1, 0, &nn, eb, el))) c->failed = 1;
    }
  const bool symm = (c->condition & SMG_COND_SYMM) != 0;
  if (symm)
    { // ---- close the table under reverse complement ACROSS the shards (Symmex, PloidyPlot.c:1395-1414) ------------
      // S1: shape of the closed table -> balanced splitters on a boundary of `symm_bits` leading bits
      const int bits = c->symm_bits, nbin = 1 << bits;
      c->symm_hist[r] = (int64_t *) calloc((size_t) 2 * nbin, sizeof(int64_t));
      if (MOK && !c->symm_hist[r]) MFAIL(SMG_ENOMEM, "out of host memory");
      if (MOK && (c->rc[r] = smg_engine_symm_hist(e, bits, c->symm_hist[r], eb, el))) c->failed = 1;
      pthread_barrier_wait(&c->bar);                                                     // S1: histograms published
      if (MOK && r == 0)
        { int64_t total = 0;
          for (int s = 0; s < n; s++) for (int b = 0; b < 2 * nbin; b++) total += c->symm_hist[s][b];
          int64_t acc = 0; int next = 1;
          for (int s = 0; s < n; s++) memset(c->first[s], 0, sizeof(c->first[s]));
          for (int b = 0; b < nbin && next < n; b++)
            { // rank `next` starts at the first bin at which the closed table has reached next / n of its entries
              while (next < n && acc >= total / n * next) { c->first[next][0] = (u64) b << (64 - bits); next++; }
              for (int s = 0; s < n; s++) acc += c->symm_hist[s][b] + c->symm_hist[s][nbin + b];
            }
          for (; next < n; next++) memset(c->first[next], 0xFF, sizeof(c->first[next]));     // (nothing left for them)
        }
      pthread_barrier_wait(&c->bar);                                                     // S2: splitters published
      free(c->symm_hist[r]); c->symm_hist[r] = NULL;
      for (int s = 1; s < n; s++) memcpy(splitters + (size_t) (s - 1) * W, c->first[s], sizeof(u64) * W);
      // S3: two records per entry (itself, its complement), grouped by destination
      const int rw = W + 1;
      const int64_t n2 = MOK ? 2 * e->n : 0;
      if (MOK && hipMalloc(&c->send[r], sizeof(uint64_t) * (size_t) (n2 > 0 ? n2 : 1) * rw) != hipSuccess)
        MFAIL(SMG_ENOMEM, "out of device memory while symmetrising");
      if (MOK && (c->rc[r] = smg_engine_symm_route(e, (const uint64_t *) splitters, n, c->send[r], n2, c->counts[r], eb, el))) c->failed = 1;
      pthread_barrier_wait(&c->bar);                                                     // S3: counts known
      int64_t nrecv = 0;
      if (MOK)
        { for (int s = 0; s < n; s++) nrecv += c->counts[s][r];
          if (hipMalloc(&recv, sizeof(uint64_t) * (size_t) (nrecv > 0 ? nrecv : 1) * rw) != hipSuccess)
            MFAIL(SMG_ENOMEM, "out of device memory while symmetrising");
        }
      if (multi_agree(c)) multi_exchange(c, r, e, recv, rw, "symmetrise");            // S4
      pthread_barrier_wait(&c->bar);                                                     // S5: peers have copied
      if (c->send[r]) { hipFree(c->send[r]); c->send[r] = NULL; }
      if (MOK)
        { int64_t nn = 0;
          if (c->virt) pthread_mutex_lock(&c->big_mu);
          if ((c->rc[r] = smg_engine_symm_finish(e, recv, nrecv, &nn, eb, el))) c->failed = 1;
          if (c->virt) pthread_mutex_unlock(&c->big_mu);
        }
      if (recv) { hipFree(recv); recv = NULL; }
      if (MOK) c->nshard[r] = e->n;
      pthread_barrier_wait(&c->bar);                                                     // A: shards ready (splitters = first[])
    }
  else
    { if (MOK)
        { c->nshard[r] = e->n;
          if (e->n > 0 && hipMemcpy(c->first[r], e->keys, sizeof(u64) * W, hipMemcpyDeviceToHost) != hipSuccess)
            MFAIL(SMG_ENODEV, "device to host copy failed");
        }
      pthread_barrier_wait(&c->bar);                                                     // A: shards + first k-mers
    }

  // ---- pass 1, requests grouped by destination ---------------------------------------------------------
  if (MOK && !symm)
    { for (int s = n - 2; s >= 0; s--)            // an empty shard inherits its successor's first k-mer
        if (c->nshard[s] == 0 && r == 0) memcpy(c->first[s], c->first[s + 1], sizeof(c->first[s]));
    }
  pthread_barrier_wait(&c->bar);                                                         // A2: splitters final
  int64_t nreq = 0;
  if (MOK)
    { for (int s = 1; s < n; s++) memcpy(splitters + (size_t) (s - 1) * W, c->first[s], sizeof(u64) * W);
      if ((c->rc[r] = smg_engine_pass1(e, c->symcheck, eb, el))) c->failed = 1;
    }
  c->bmap[r] = NULL; c->bm_bits[r] = 0;
  if (MOK && e->bm_bits) { c->bmap[r] = e->bmap; c->bm_bits[r] = e->bm_bits; hipStreamSynchronize(e->stream); }
  pthread_barrier_wait(&c->bar);                                                         // A3: block maps complete
  if (MOK && (c->rc[r] = multi_filter(c, r, e, splitters, eb, el))) c->failed = 1;
  if (MOK)
    { nreq = smg_engine_nreq(e);
      c->rw = smg_engine_record_words(e);
      if (hipMalloc(&c->send[r], sizeof(uint64_t) * (size_t) (nreq > 0 ? nreq : 1) * c->rw) != hipSuccess)
        MFAIL(SMG_ENOMEM, "out of device memory for the request exchange");
    }
  if (MOK && (c->rc[r] = smg_engine_route(e, (const uint64_t *) splitters, n, c->send[r], nreq, c->counts[r], eb, el))) c->failed = 1;
  pthread_barrier_wait(&c->bar);                                                         // B: all counts known

  // ---- exchange -------------------------------------------------------------------------------------------
  int64_t nrecv = 0;
  if (MOK)
    { const int rw = c->rw;
      for (int s = 0; s < n; s++) nrecv += c->counts[s][r];
      if (hipMalloc(&recv, sizeof(uint64_t) * (size_t) (nrecv > 0 ? nrecv : 1) * rw) != hipSuccess)
        MFAIL(SMG_ENOMEM, "out of device memory for the request exchange");
    }
  const bool go_exchange = multi_agree(c);                                               // B2: all in, or all out
  if (go_exchange) multi_exchange(c, r, e, recv, c->rw, "requests");
  if (MOK && (c->rc[r] = smg_engine_apply(e, recv, nrecv, &c->missing[r], eb, el))) c->failed = 1;
  if (MOK) smg_engine_symhash(e, (uint64_t *) c->fp[r], eb, el);
  pthread_barrier_wait(&c->bar);                                                         // C: proof words published
  if (c->send[r]) { hipFree(c->send[r]); c->send[r] = NULL; }     // peers have copied what they needed
  if (recv) { hipFree(recv); recv = NULL; }

  // ---- symmetry proof (host reduction), pass 2, histogram reduction -------------------------------------
  if (MOK)
    { int64_t miss = 0; u64 f[4] = { 0, 0, 0, 0 };
      for (int s = 0; s < n; s++) { miss += c->missing[s]; for (int q = 0; q < 4; q++) f[q] ^= c->fp[s][q]; }
      bool symmetric = miss == 0;
      if (c->symcheck == SMG_SYM_HASH) symmetric = symmetric && f[0] == f[2] && f[1] == f[3];
      if (!symmetric && c->virt)
        c->general = 1;                       // (every rank computes the same verdict from the same published words)
      else if (!symmetric)
        MFAIL(SMG_ENOTSYM, "the table is not closed under reverse complement with equal counts: "
                           "a multi-GPU run needs a conditioned table (use one GPU for this one)");
    }
  const bool go_pass2 = multi_agree(c);                                                  // C1: verdict known to all
  if (go_pass2 && c->general)
    { // The reference answers for ANY sorted table (it only spot-checks the symmetry, PloidyPlot.c:1199-1229): so do
      // the shards of one device, together -- both passes over all k positions, a prefix-side partner looked up in
      // whichever shard holds it (TabSet).  Several real GPUs hand such a table to one GPU instead (host_run).
      if (MOK && (c->rc[r] = general_shard_prepare(e, &c->gtab[r], eb, el))) c->failed = 1;
      pthread_barrier_wait(&c->bar);                                                     // G1: directories built
      if (MOK && r == 0)
        { TabSet hs; memset(&hs, 0, sizeof(hs));
          hs.ns = n;
          for (int s = 0; s < n; s++)
            { hs.shard[s] = c->gtab[s];
              for (int w = 0; w < 4; w++) hs.first[s][w] = w < W ? c->first[s][w] : 0;
            }
          if (hipMalloc(&c->d_set, sizeof(TabSet)) != hipSuccess
              || hipMemcpy(c->d_set, &hs, sizeof(TabSet), hipMemcpyHostToDevice) != hipSuccess)
            MFAIL(SMG_ENOMEM, "out of device memory for the shard set");
        }
      pthread_barrier_wait(&c->bar);                                                     // G2: shard set published
      if (MOK && (c->rc[r] = general_shard_pass(e, c->d_set, 1, d_plot, eb, el))) c->failed = 1;
      pthread_barrier_wait(&c->bar);                                                     // G3: all degrees final
      if (MOK && (c->rc[r] = general_shard_pass(e, c->d_set, 2, d_plot, eb, el))) c->failed = 1;
      pthread_barrier_wait(&c->bar);                                                     // G4: nobody reads the set any more
      if (r == 0 && c->d_set) { hipFree(c->d_set); c->d_set = NULL; }
    }
  else if (go_pass2 && MOK && (c->rc[r] = smg_engine_pass2(e, d_plot, eb, el))) c->failed = 1;
  const bool go_reduce = multi_agree(c);                                                 // C2: all in, or all out
  if (go_reduce && c->virt)
    { c->h_plot[r] = (int64_t *) malloc(sizeof(int64_t) * SMG_PLOT_CELLS);
      if (!c->h_plot[r] || hipMemcpy(c->h_plot[r], d_plot, sizeof(int64_t) * SMG_PLOT_CELLS, hipMemcpyDeviceToHost) != hipSuccess)
        MFAIL(SMG_ENODEV, "device to host copy failed");
    }
  else if (go_reduce)
    { const ncclResult_t nr = c->api.AllReduce(d_plot, d_plot, SMG_PLOT_CELLS, ncclInt64, ncclSum, c->comm[r], e->stream);
      if (nr != ncclSuccess || hipStreamSynchronize(e->stream) != hipSuccess)
        { c->rc[r] = fail(eb, el, SMG_ENODEV, "RCCL all-reduce of the histograms failed: %s",
                          nr != ncclSuccess ? c->api.GetErrorString(nr) : "stream error");
          c->failed = 1;
        }
      if (MOK && r == 0 && hipMemcpy(c->plot, d_plot, sizeof(int64_t) * SMG_PLOT_CELLS, hipMemcpyDeviceToHost) != hipSuccess)
        MFAIL(SMG_ENODEV, "device to host copy failed");
    }
  if (e) c->st[r] = e->st;
  pthread_barrier_wait(&c->bar);                                                         // D: histograms ready
  if (MOK && c->virt && r == 0)
    { for (int cell = 0; cell < SMG_PLOT_CELLS; cell++)
        { int64_t v = 0;
          for (int s = 0; s < n; s++) v += c->h_plot[s][cell];
          c->plot[cell] = v;
        }
    }
  // ---- extract leg (PloidyList.c:1207-1583 has no size limit either): every shard lists the pairs behind the labelled
  // pixels among ITS entries -- the two members of a pair share a window block, hence a shard -- the host concatenates
  c->h_rec[r] = NULL; c->nrec[r] = 0;
  if (MOK && c->labels)
    { uint16_t *d_labels = NULL; uint64_t *d_out = NULL;
      int64_t cnt = 0, got = 0;
      const int rw = W + 1;
      if (c->general) MFAIL(SMG_EINVAL, "extract needs a trimmed, reverse-complement closed table");
      if (MOK && (hipMalloc(&d_labels, sizeof(uint16_t) * SMG_PLOT_CELLS) != hipSuccess
                  || hipMemcpy(d_labels, c->labels, sizeof(uint16_t) * SMG_PLOT_CELLS, hipMemcpyHostToDevice) != hipSuccess))
        MFAIL(SMG_ENOMEM, "out of device memory for the pair list");
      if (MOK && (c->rc[r] = smg_engine_extract(e, d_labels, NULL, 0, &cnt, eb, el))) c->failed = 1;       // count first
      if (MOK && cnt > 0)
        { c->h_rec[r] = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) cnt * rw);
          if (!c->h_rec[r] || hipMalloc(&d_out, sizeof(uint64_t) * (size_t) cnt * rw) != hipSuccess)
            MFAIL(SMG_ENOMEM, "out of memory for the pair list");
          if (MOK && (c->rc[r] = smg_engine_extract(e, d_labels, d_out, cnt, &got, eb, el))) c->failed = 1;
          if (MOK && got != cnt) MFAIL(SMG_ENODEV, "internal error: the pair list changed between two passes");
          if (MOK && hipMemcpy(c->h_rec[r], d_out, sizeof(uint64_t) * (size_t) cnt * rw, hipMemcpyDeviceToHost) != hipSuccess)
            MFAIL(SMG_ENODEV, "device to host copy failed");
          if (MOK) c->nrec[r] = cnt;
        }
      if (d_labels) hipFree(d_labels);
      if (d_out) hipFree(d_out);
    }
  pthread_barrier_wait(&c->bar);                                                         // E: done with h_plot
  free(c->h_plot[r]); c->h_plot[r] = NULL;
  if (d_plot) hipFree(d_plot);
  if (e) smg_engine_destroy(e);
  return NULL;
}
#undef MFAIL
#undef MOK

// cut points: about n*r/N, moved to a prefix-index boundary that is also a window-block boundary
static void multi_cuts(const smg_table_source *tv, int n, int64_t *cut)
{ const int64_t ixlen = 1ll << (8 * tv->ibyte);
  const int p0 = tv->kmer / 2, ib = 4 * tv->ibyte;          // bases in a window-block prefix / in an index bucket
  // an index bucket fixes the first `ib` bases; a window block the first `p0`: when ib > p0 only every
  // 4^(ib-p0)-th bucket boundary is also a block boundary
  const int64_t gran = ib > p0 ? 1ll << (2 * (ib - p0)) : 1;
  cut[0] = 0; cut[n] = tv->nels;
  for (int r = 1; r < n; r++)
    { const int64_t target = tv->nels / n * r;
      int64_t lo = 0, hi = ixlen - 1;                         // smallest bucket b with index[b] >= target
      while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (tv->prefix_index[m] < target) lo = m + 1; else hi = m; }
      int64_t b = lo + 1;                                      // the cut goes in front of bucket b
      b = (b + gran / 2) / gran * gran;
      if (b >= ixlen) b = ixlen;
      int64_t cpos = b == 0 ? 0 : tv->prefix_index[b - 1];
      if (cpos < cut[r - 1]) cpos = cut[r - 1];
      cut[r] = cpos;
    }
}

// ---- out of core: prefix shards ONE AFTER THE OTHER on one device ---------------------------------------------------------
// The reference streams a table of any size from disk (small_recursion loads a subtree into its cache when it fits and
// falls back to the streaming twins when it does not, PloidyPlot.c:931-1038); the virtual shards above must all be
// resident together.  Here a table whose shards do not fit together is slow instead of an error: what has to outlive a
// shard between its two passes is its code bytes (1 byte per entry) and its requests (8 bytes for ~17 % of the entries,
// grouped by destination shard) -- not its k-mers.
//   round 1, shard by shard: read + decode, pass 1 (no candidate map: the filter would need the maps of the shards that have
//            not been read yet), requests grouped by destination, code bytes and requests kept, k-mers dropped;
//   round 2, shard by shard: read + decode again (the reference reads its table ~2 (BLEVEL + 1) times), code bytes back, look-ups
//            of all requests that name this shard, pass 2 into the one histogram.
// Symmetry proof as everywhere: XOR of the shards' fingerprints, no look-up may miss.  A table that fails it is refused with a
// precise message (condition the table with smg_condition first).  Round 6: a table that still has to be conditioned is -- shard by
// shard, host_condition_sequential below -- and k > 85 takes the same two rounds with the counted kernels (counted_resume).
// The extract leg runs per shard in round 2 (the two members of a pair share a shard).
// A RAW table out of core (round 6): Logex 'A[e-]' + Symmex (PloidyPlot.c:1381-1414) shard by shard.  The conditioned table
// never exists on the device as a whole -- and not on disk either (the reference writes .trim / .symx tables into the working
// directory): its prefix shards are left in HOST memory (8 W + 2 bytes per entry), from where the two rounds of the run
// below take them instead of reading and decoding part files.
//   sweep 1  every piece of the input: read + decode, trim, histogram of the entries AND their complements per leading 12 bits
//            -> splitters that cut the CLOSED table into n shards of equal size (the rule of the in-core protocol, host_run_multi);
//   sweep 2  every piece again: trim, one record per entry and per complement grouped by destination shard (symm_route),
//            copied out to the destination's list on the host (2 (W + 1) words per kept entry, for the length of this sweep);
//   then     every destination: its records back to the device, sort + dedupe (symm_finish), the shard's k-mers and counts
//            out to the host.
// Trim only: one sweep, the pieces are the shards.
struct HostShard { std::vector<u64> keys; std::vector<uint16_t> cnt; int64_t n = 0; };

static int host_condition_sequential(const smg_table_source *tv, const smg_opts *opts, int n, smg_engine *e, const int64_t *d_index,
                                     std::vector<HostShard> &out, std::vector<u64> &splitters, int verbose, char *errbuf, size_t errlen)
{ const int W = (tv->kmer + 31) / 32, kbyte = (tv->kmer + 3) >> 2, pbyte = kbyte + 2 - tv->ibyte, rw = W + 1;
  const bool trim = (opts->condition & SMG_COND_TRIM) != 0, symm = (opts->condition & SMG_COND_SYMM) != 0;
  std::vector<int64_t> cut((size_t) n + 1);
  multi_cuts(tv, n, cut.data());
  int rc = SMG_OK;
  out.assign((size_t) n, HostShard());
  splitters.assign((size_t) (n > 1 ? n - 1 : 1) * W, 0);
  auto load = [&](int s) -> int            // piece s of the input in the engine, trimmed
  { const int64_t lo = cut[s], hi = cut[s + 1];
    int r = decode_begin(e, tv->kmer, tv->ibyte, hi - lo, errbuf, errlen);
    if (r) return r;
    DecodeHook hk; hk.e = e; hk.d_index = d_index; hk.ibyte = tv->ibyte; hk.ibase = lo;
    if ((r = ingest_records(tv, pbyte, lo, hi, NULL, opts->device, tv->host_threads, NULL, errbuf, errlen, decode_hook, &hk))) return r;
    if (hipStreamSynchronize(e->stream) != hipSuccess) return fail(errbuf, errlen, SMG_ENODEV, "decode failed%s");
    if (trim) { int64_t nn = 0; if ((r = smg_engine_condition(e, opts->ethresh, 1, 0, &nn, errbuf, errlen))) return r; }
    return SMG_OK;
  };
  auto take = [&](int d) -> int            // the engine's table -> shard d on the host
  { HostShard &h = out[(size_t) d];
    h.n = e->n;
    h.keys.resize((size_t) (e->n > 0 ? e->n : 0) * W); h.cnt.resize((size_t) (e->n > 0 ? e->n : 0));
    if (e->n > 0 && (hipMemcpy(h.keys.data(), e->keys, sizeof(u64) * (size_t) e->n * W, hipMemcpyDeviceToHost) != hipSuccess
                     || hipMemcpy(h.cnt.data(), e->cnt, sizeof(uint16_t) * (size_t) e->n, hipMemcpyDeviceToHost) != hipSuccess))
      return fail(errbuf, errlen, SMG_ENODEV, "device to host copy failed%s");
    return SMG_OK;
  };
  if (!symm)
    { for (int s = 0; s < n && rc == SMG_OK; s++) { if ((rc = load(s)) == SMG_OK) rc = take(s); }
      // (a splitter = the first k-mer a shard can hold: the prefix bucket its piece starts with, as for a conditioned table)
      const int64_t ixlen = 1ll << (8 * tv->ibyte);
      for (int sh = 1; sh < n; sh++)
        { int64_t lo = 0, hi = ixlen;
          while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (tv->prefix_index[m] > cut[sh]) hi = m; else lo = m + 1; }
          splitters[(size_t) (sh - 1) * W] = lo >= ixlen ? ~0ull : (u64) lo << (64 - 8 * tv->ibyte);
          if (lo >= ixlen) for (int w = 1; w < W; w++) splitters[(size_t) (sh - 1) * W + w] = ~0ull;
        }
      return rc;
    }
  const int bits = 2 * (tv->kmer / 2) < 12 ? (2 * (tv->kmer / 2) < 2 ? 2 : 2 * (tv->kmer / 2)) : 12, nbin = 1 << bits;
  std::vector<int64_t> hist((size_t) 2 * nbin, 0), one((size_t) 2 * nbin, 0);
  for (int s = 0; s < n && rc == SMG_OK; s++)                                  // sweep 1
    { if ((rc = load(s))) break;
      if ((rc = smg_engine_symm_hist(e, bits, one.data(), errbuf, errlen))) break;
      for (int b = 0; b < 2 * nbin; b++) hist[(size_t) b] += one[(size_t) b];
    }
  if (rc) return rc;
  { int64_t total = 0, acc = 0; int next = 1;
    for (int b = 0; b < 2 * nbin; b++) total += hist[(size_t) b];
    for (int b = 0; b < nbin && next < n; b++)
      { while (next < n && acc >= total / n * next) { splitters[(size_t) (next - 1) * W] = (u64) b << (64 - bits); next++; }
        acc += hist[(size_t) b] + hist[(size_t) nbin + b];
      }
    for (; next < n; next++) for (int w = 0; w < W; w++) splitters[(size_t) (next - 1) * W + w] = ~0ull;     // (nothing left for them)
    if (total / n >= 0xFFFFFFF0ll - 16)
      return fail(errbuf, errlen, SMG_EINVAL, "out of core: a shard of the closed table would hold more than 2^32 entries (more shards are needed)%s");
  }
  std::vector<std::vector<u64> > rec((size_t) n);
  std::vector<int64_t> counts((size_t) n);
  for (int s = 0; s < n && rc == SMG_OK; s++)                                  // sweep 2
    { if ((rc = load(s))) break;
      const int64_t n2 = 2 * e->n;
      uint64_t *send = NULL;
      if (hipMalloc(&send, sizeof(uint64_t) * (size_t) (n2 > 0 ? n2 : 1) * rw) != hipSuccess)
        return fail(errbuf, errlen, SMG_ENOMEM, "out of device memory while symmetrising%s");
      rc = smg_engine_symm_route(e, (const uint64_t *) splitters.data(), n, send, n2, counts.data(), errbuf, errlen);
      int64_t off = 0;
      for (int d = 0; d < n && rc == SMG_OK; d++)
        { const size_t at = rec[(size_t) d].size(), words = (size_t) counts[(size_t) d] * rw;
          rec[(size_t) d].resize(at + words);
          if (words && hipMemcpy(rec[(size_t) d].data() + at, send + (size_t) off * rw, sizeof(uint64_t) * words, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(errbuf, errlen, SMG_ENODEV, "device to host copy failed%s");
          off += counts[(size_t) d];
        }
      hipFree(send);
    }
  for (int d = 0; d < n && rc == SMG_OK; d++)                                  // every destination: sort + dedupe, out to the host
    { const int64_t nrecv = (int64_t) (rec[(size_t) d].size() / (size_t) rw);
      uint64_t *recv = NULL;
      if (hipMalloc(&recv, sizeof(uint64_t) * (size_t) (nrecv > 0 ? nrecv : 1) * rw) != hipSuccess)
        return fail(errbuf, errlen, SMG_ENOMEM, "out of device memory while symmetrising%s");
      if (nrecv && hipMemcpy(recv, rec[(size_t) d].data(), sizeof(uint64_t) * (size_t) nrecv * rw, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(errbuf, errlen, SMG_ENODEV, "host to device copy failed%s");
      std::vector<u64>().swap(rec[(size_t) d]);
      int64_t nn = 0;
      if (rc == SMG_OK) rc = smg_engine_symm_finish(e, recv, nrecv, &nn, errbuf, errlen);
      hipFree(recv);
      if (rc == SMG_OK) rc = take(d);
    }
  if (rc == SMG_OK && verbose)
    { int64_t tot = 0; for (int d = 0; d < n; d++) tot += out[(size_t) d].n;
      fprintf(stderr, "  [smg] conditioned out of core: %lld -> %lld k-mers in %d prefix shards kept in host memory (%s%s)\n", (long long) tv->nels,
              (long long) tot, n, trim ? "trimmed, " : "", "closed under reverse complement");
    }
  return rc;
}

static int host_run_sequential(const smg_table_source *tv, const smg_opts *opts, int nshards, int64_t *plot, smg_stats *stats,
                               char *errbuf, size_t errlen, const uint16_t *labels = NULL, uint64_t **records = NULL,
                               int64_t *nrec_out = NULL, int *rec_words = NULL)
{ const int W = (tv->kmer + 31) / 32, kbyte = (tv->kmer + 3) >> 2, pbyte = kbyte + 2 - tv->ibyte;
  const bool counted = tv->kmer > FAST_MAX_K;    // k > 85: the counted kernels in the same steps (what stays between the rounds is the degree byte)
  if (nshards < 2) nshards = 2;
  if (nshards > SMG_MAXGPU) nshards = SMG_MAXGPU;            // (the router groups by at most 16 destinations)
  const int symcheck = opts->symcheck == SMG_SYM_NONE ? SMG_SYM_HASH : opts->symcheck;
  const int n = nshards;
  const bool raw = opts->condition != 0;   // the shards come out of host_condition_sequential (host memory), not out of the part files
  std::vector<HostShard> hshard;
  std::vector<int64_t> cut((size_t) n + 1), counts((size_t) n * n, 0), nreq((size_t) n, 0);
  std::vector<uint8_t *> codes((size_t) n, (uint8_t *) NULL);
  std::vector<uint64_t *> send((size_t) n, (uint64_t *) NULL);
  std::vector<u64> splitters((size_t) (n > 1 ? n - 1 : 1) * W, 0);
  multi_cuts(tv, n, cut.data());
  for (int sh = 0; sh < n; sh++)          // (a forced shard count -- SMG_SEQUENTIAL_SHARDS -- may leave a shard too large to index)
    if (cut[sh + 1] - cut[sh] >= 0xFFFFFFF0ll - 16)
      return fail(errbuf, errlen, SMG_EINVAL, "out of core: a shard of more than 2^32 entries (more shards are needed)%s");
  if (!raw)
  { // a splitter = the prefix bucket a shard starts with (cuts are bucket boundaries): everything in front is smaller
    const int64_t ixlen = 1ll << (8 * tv->ibyte);
    for (int sh = 1; sh < n; sh++)
      { int64_t lo = 0, hi = ixlen;                            // smallest bucket b with index[b] > cut[sh]
        while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (tv->prefix_index[m] > cut[sh]) hi = m; else lo = m + 1; }
        splitters[(size_t) (sh - 1) * W] = lo >= ixlen ? ~0ull : (u64) lo << (64 - 8 * tv->ibyte);
        if (lo >= ixlen) for (int w = 1; w < W; w++) splitters[(size_t) (sh - 1) * W + w] = ~0ull;
      }
  }
  smg_engine *e = smg_engine_create(opts->device, NULL, errbuf, errlen);
  if (!e) return SMG_ENODEV;
  e->no_filter = true;                    // (k <= 85: no candidate map out of core -- the filter would need the maps of the shards not read yet)
  // extract leg (round 5): the two members of a pair share a window block, hence a shard -- every shard lists the pairs behind
  // the labelled pixels while it is resident for its second round; the lists are only handed out if the whole table proves closed
  std::vector<uint64_t> xrec;
  uint16_t *d_labels = NULL;
  int rc = SMG_OK, rw = W;
  int64_t *d_index = NULL, *d_plot = NULL, *h_plot = NULL;
  uint64_t *recv = NULL;
  int64_t missing = 0, nels = 0, nemit = 0;
  u64 fp[4] = { 0, 0, 0, 0 };
  float ms_p1 = 0, ms_look = 0, ms_p2 = 0;
  const size_t ixbytes = sizeof(int64_t) << (8 * tv->ibyte);
  struct timespec w0, w1;
  clock_gettime(CLOCK_MONOTONIC, &w0);
#define SBAIL(code, msg) { rc = fail(errbuf, errlen, code, msg "%s"); goto done; }
  if (hipMalloc(&d_index, ixbytes) != hipSuccess || hipMalloc(&d_plot, sizeof(int64_t) * SMG_PLOT_CELLS) != hipSuccess)
    SBAIL(SMG_ENOMEM, "out of device memory for the table index")
  if (hipMemcpy(d_index, tv->prefix_index, ixbytes, hipMemcpyHostToDevice) != hipSuccess) SBAIL(SMG_ENODEV, "host to device copy failed")
  h_plot = (int64_t *) malloc(sizeof(int64_t) * SMG_PLOT_CELLS);
  if (!h_plot) SBAIL(SMG_ENOMEM, "out of host memory")
  memset(plot, 0, sizeof(int64_t) * SMG_PLOT_CELLS);
  if (raw && (rc = host_condition_sequential(tv, opts, n, e, d_index, hshard, splitters, opts->verbose, errbuf, errlen))) goto done;
  for (int round = 1; round <= 2 && rc == SMG_OK; round++)
    for (int sh = 0; sh < n && rc == SMG_OK; sh++)
      { const int64_t lo = cut[sh], hi = cut[sh + 1], ns = raw ? hshard[(size_t) sh].n : hi - lo;
        if ((rc = decode_begin(e, tv->kmer, tv->ibyte, ns, errbuf, errlen))) break;
        if (raw)                           // a conditioned shard from host memory (it has no prefix index: pass 1 / k_directory build a directory)
          { const HostShard &h = hshard[(size_t) sh];
            if (ns > 0 && (hipMemcpy(e->own_keys, h.keys.data(), sizeof(u64) * (size_t) ns * W, hipMemcpyHostToDevice) != hipSuccess
                           || hipMemcpy(e->own_cnt, h.cnt.data(), sizeof(uint16_t) * (size_t) ns, hipMemcpyHostToDevice) != hipSuccess))
              SBAIL(SMG_ENODEV, "host to device copy failed")
          }
        else
          { DecodeHook hk; hk.e = e; hk.d_index = d_index; hk.ibyte = tv->ibyte; hk.ibase = lo;
            if ((rc = ingest_records(tv, pbyte, lo, hi, NULL, opts->device, tv->host_threads, NULL, errbuf, errlen, decode_hook, &hk))) break;
            if ((rc = smg_engine_set_prefix_index(e, d_index, tv->ibyte, lo, errbuf, errlen))) break;
          }
        if (hipStreamSynchronize(e->stream) != hipSuccess) SBAIL(SMG_ENODEV, "decode failed")
        if (round == 1)
          { e->bm_want = 32;
            if ((rc = smg_engine_pass1(e, symcheck, errbuf, errlen))) break;
            rw = smg_engine_record_words(e);
            nreq[sh] = smg_engine_nreq(e);
            if (hipMalloc(&send[sh], sizeof(uint64_t) * (size_t) (nreq[sh] > 0 ? nreq[sh] : 1) * rw) != hipSuccess
                || hipMalloc(&codes[sh], (size_t) (ns > 0 ? ns : 1)) != hipSuccess)
              SBAIL(SMG_ENOMEM, "out of device memory for what a shard leaves behind (1 byte per entry and its requests)")
            if ((rc = smg_engine_route(e, (const uint64_t *) splitters.data(), n, send[sh], nreq[sh], &counts[(size_t) sh * n], errbuf, errlen))) break;
            if (ns > 0 && hipMemcpy(codes[sh], e->deg, (size_t) ns, hipMemcpyDeviceToDevice) != hipSuccess) SBAIL(SMG_ENODEV, "device copy failed")
            for (int q = 0; q < 4; q++) fp[q] ^= e->fp[q];
            ms_p1 += e->st.ms_pass1; nels += ns; nemit += e->st.nemitted;
          }
        else
          { if ((rc = counted ? counted_resume(e, codes[sh], errbuf, errlen) : fast_resume(e, codes[sh], symcheck == SMG_SYM_EXACT, errbuf, errlen))) break;
            int64_t nrecv = 0;
            for (int t = 0; t < n; t++) nrecv += counts[(size_t) t * n + sh];
            if (hipMalloc(&recv, sizeof(uint64_t) * (size_t) (nrecv > 0 ? nrecv : 1) * rw) != hipSuccess)
              SBAIL(SMG_ENOMEM, "out of device memory for a shard's requests")
            int64_t roff = 0;
            for (int t = 0; t < n; t++)
              { int64_t soff = 0;
                for (int d = 0; d < sh; d++) soff += counts[(size_t) t * n + d];
                const int64_t c = counts[(size_t) t * n + sh];
                if (c && hipMemcpy(recv + roff * rw, send[t] + soff * rw, sizeof(uint64_t) * (size_t) c * rw, hipMemcpyDeviceToDevice) != hipSuccess)
                  SBAIL(SMG_ENODEV, "device copy failed")
                roff += c;
              }
            int64_t miss = 0;
            if ((rc = smg_engine_apply(e, recv, nrecv, &miss, errbuf, errlen))) break;
            missing += miss;
            hipFree(recv); recv = NULL;
            hipFree(codes[sh]); codes[sh] = NULL;
            if ((rc = smg_engine_pass2(e, d_plot, errbuf, errlen))) break;
            if (hipMemcpy(h_plot, d_plot, sizeof(int64_t) * SMG_PLOT_CELLS, hipMemcpyDeviceToHost) != hipSuccess) SBAIL(SMG_ENODEV, "device to host copy failed")
            for (int cell = 0; cell < SMG_PLOT_CELLS; cell++) plot[cell] += h_plot[cell];
            if (labels)
              { int64_t cnt = 0, got = 0;
                uint64_t *d_out = NULL;
                if (!d_labels && (hipMalloc(&d_labels, sizeof(uint16_t) * SMG_PLOT_CELLS) != hipSuccess
                                  || hipMemcpy(d_labels, labels, sizeof(uint16_t) * SMG_PLOT_CELLS, hipMemcpyHostToDevice) != hipSuccess))
                  SBAIL(SMG_ENOMEM, "out of device memory for the pair list")
                if ((rc = smg_engine_extract(e, d_labels, NULL, 0, &cnt, errbuf, errlen))) break;          // count first
                if (cnt > 0)
                  { const size_t at = xrec.size(), words = (size_t) cnt * (W + 1);
                    if (hipMalloc(&d_out, sizeof(uint64_t) * words) != hipSuccess) SBAIL(SMG_ENOMEM, "out of device memory for the pair list")
                    rc = smg_engine_extract(e, d_labels, d_out, cnt, &got, errbuf, errlen);
                    if (rc == SMG_OK && got != cnt) rc = fail(errbuf, errlen, SMG_ENODEV, "internal error: the pair list changed between two passes%s");
                    if (rc == SMG_OK)
                      { xrec.resize(at + words);
                        if (hipMemcpy(xrec.data() + at, d_out, sizeof(uint64_t) * words, hipMemcpyDeviceToHost) != hipSuccess)
                          rc = fail(errbuf, errlen, SMG_ENODEV, "device to host copy failed%s");
                      }
                    hipFree(d_out);
                    if (rc) break;
                  }
              }
            smg_stats st2; smg_engine_stats(e, &st2);
            ms_look += st2.ms_rclookup; ms_p2 += st2.ms_pass2 > 0 ? st2.ms_pass2 : 0;
          }
      }
  if (rc == SMG_OK)
    { bool symmetric = missing == 0;
      if (symcheck == SMG_SYM_HASH) symmetric = symmetric && fp[0] == fp[2] && fp[1] == fp[3];
      if (!symmetric)
        rc = fail(errbuf, errlen, SMG_ENOTSYM, "the table is not closed under reverse complement with equal counts, and it does not fit the "
                  "device in one piece: condition it first (smg_condition)%s");
    }
  if (rc == SMG_OK && labels)
    { // as everywhere: the number of records is the plot's weight on the labelled pixels
      int64_t want = 0;
      for (int cell = 0; cell < SMG_PLOT_CELLS; cell++) if (labels[cell]) want += plot[cell];
      const int64_t have = (int64_t) (xrec.size() / (size_t) (W + 1));
      if (have != want) rc = fail(errbuf, errlen, SMG_ENODEV, "internal error: pair list and plot disagree%s");
      else
        { uint64_t *all = (uint64_t *) malloc(sizeof(uint64_t) * (xrec.size() ? xrec.size() : 1));
          if (!all) rc = fail(errbuf, errlen, SMG_ENOMEM, "out of host memory for the pair list%s");
          else
            { if (!xrec.empty()) memcpy(all, xrec.data(), sizeof(uint64_t) * xrec.size());
              *records = all; *nrec_out = have; *rec_words = W + 1;
            }
        }
    }
  if (rc == SMG_OK)
    { clock_gettime(CLOCK_MONOTONIC, &w1);
      smg_stats st; memset(&st, 0, sizeof(st));
      st.path = 1; st.key_words = W; st.nels = nels; st.nemitted = nemit; st.nrequests = nemit;
      st.ms_pass1 = ms_p1; st.ms_rclookup = ms_look; st.ms_pass2 = ms_p2;
      st.ms_total = (float) (((double) (w1.tv_sec - w0.tv_sec) + 1e-9 * (double) (w1.tv_nsec - w0.tv_nsec)) * 1e3);
      for (int cell = 0; cell < SMG_PLOT_CELLS; cell++) st.npairs += plot[cell];
      if (stats) *stats = st;
      if (opts->verbose)
        fprintf(stderr, "  [smg] n=%lld k=%d out of core: %d prefix shards one after the other, the table read twice; pass1 %.2f ms, "
                "rc-lookup %.2f, pass2 %.2f (sums over the shards), wall incl. both reads %.2f ms\n",
                (long long) nels, tv->kmer, n, ms_p1, ms_look, ms_p2, st.ms_total);
    }
done:
#undef SBAIL
  for (int sh = 0; sh < n; sh++) { if (codes[sh]) hipFree(codes[sh]); if (send[sh]) hipFree(send[sh]); }
  if (recv) hipFree(recv);
  if (d_index) hipFree(d_index);
  if (d_plot) hipFree(d_plot);
  if (d_labels) hipFree(d_labels);
  free(h_plot);
  smg_engine_destroy(e);
  return rc;
}

static int host_run_multi(const smg_table_source *tv, const smg_opts *opts, int ngpus, bool force_virtual, int64_t *plot,
                          smg_stats *stats, char *errbuf, size_t errlen, const uint16_t *labels = NULL,
                          uint64_t **records = NULL, int64_t *nrec = NULL, int *rec_words = NULL)
{ if (ngpus > SMG_MAXGPU) ngpus = SMG_MAXGPU;
  MultiCtx *c = new (std::nothrow) MultiCtx();
  if (!c) return fail(errbuf, errlen, SMG_ENOMEM, "out of host memory%s");
  memset(c, 0, sizeof(*c));
  c->n = ngpus; c->tv = tv; c->symcheck = opts->symcheck == SMG_SYM_NONE ? SMG_SYM_HASH : opts->symcheck;
  c->condition = opts->condition; c->ethresh = opts->ethresh;
  c->W = (tv->kmer + 31) / 32;
  c->pbyte = ((tv->kmer + 3) >> 2) + 2 - tv->ibyte;
  c->plot = plot;
  c->labels = labels;
  { const char *v = getenv("SMG_VIRTUAL_SHARDS"); c->virt = force_virtual || (v && atoi(v) > 0); }
  c->io_threads = (tv->host_threads > 0 ? tv->host_threads : 4) / ngpus;
  if (c->io_threads < 2) c->io_threads = 2;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    { delete c; return fail(errbuf, errlen, SMG_ENODEV, "no HIP device available (this engine has no CPU fallback)%s"); }
  for (int r = 0; r < ngpus; r++) c->devs[r] = c->virt ? opts->device : opts->device + r;
  if (!c->virt && opts->device + ngpus > ndev)
    { delete c; return fail(errbuf, errlen, SMG_EINVAL, "fewer HIP devices than SMUDGEPLOT_GPUS asks for%s"); }
  if (!c->virt)
    { if (!rccl_load(&c->api, errbuf, errlen)) { delete c; return SMG_ENODEV; }
      const ncclResult_t nr = c->api.CommInitAll(c->comm, ngpus, c->devs);
      if (nr != ncclSuccess)
        { const int rc = fail(errbuf, errlen, SMG_ENODEV, "ncclCommInitAll failed: %s", c->api.GetErrorString(nr)); delete c; return rc; }
    }
  multi_cuts(tv, ngpus, c->cut);
  { const int p0 = tv->kmer / 2;               // splitters of the symmetrise step: a boundary of <= 12 leading bits is a
    c->symm_bits = 2 * p0 < SY_MAXBITS ? 2 * p0 : SY_MAXBITS;      // window-block boundary too (blocks share p0 bases)
    if (c->symm_bits < 2) c->symm_bits = 2;
  }
  pthread_mutex_init(&c->big_mu, NULL);
  pthread_barrier_init(&c->bar, NULL, (unsigned) ngpus);
  pthread_t th[SMG_MAXGPU];
  MultiArg args[SMG_MAXGPU];
  hipEvent_t t0, t1;
  hipSetDevice(c->devs[0]);
  hipEventCreate(&t0); hipEventCreate(&t1);
  hipEventRecord(t0, 0);
  for (int r = 0; r < ngpus; r++) { args[r].c = c; args[r].r = r; }
  int started = 0;
  for (int r = 1; r < ngpus; r++)
    if (pthread_create(&th[r], NULL, multi_worker, &args[r]) == 0) started++;
  int rc = SMG_OK;
  if (started != ngpus - 1)
    { // cannot run with a partial team: the barrier would never release.  Nothing else has started waiting
      // on it yet only if NO thread was created; otherwise this process cannot recover cleanly.
      rc = fail(errbuf, errlen, SMG_ENOMEM, "cannot create the per-GPU host threads%s");
      if (started) { fprintf(stderr, "smg_hetmers: fatal: partial thread team\n"); abort(); }
    }
  else
    { multi_worker(&args[0]);
      for (int r = 1; r < ngpus; r++) pthread_join(th[r], NULL);
      for (int r = 0; r < ngpus && rc == SMG_OK; r++)
        if (c->rc[r] != SMG_OK) { rc = c->rc[r]; if (errbuf && errlen) snprintf(errbuf, errlen, "GPU %d: %s", c->devs[r], c->err[r]); }
    }
  hipSetDevice(c->devs[0]);
  hipEventRecord(t1, 0); hipEventSynchronize(t1);
  float wall = 0; hipEventElapsedTime(&wall, t0, t1);
  hipEventDestroy(t0); hipEventDestroy(t1);
  if (rc == SMG_OK)
    { smg_stats st; memset(&st, 0, sizeof(st));
      st.path = c->general ? 2 : 1; st.key_words = c->W;
      for (int r = 0; r < ngpus; r++)
        { st.nels += c->st[r].nels; st.nrequests += c->st[r].nrequests; st.nemitted += c->st[r].nemitted;
#define MX(f) if (c->st[r].f > st.f) st.f = c->st[r].f
          MX(ms_decode); MX(ms_pass1); MX(ms_rclookup); MX(ms_pass2);
#undef MX
        }
      for (int cell = 0; cell < SMG_PLOT_CELLS; cell++) st.npairs += plot[cell];
      st.ms_total = wall;                  // H2D + decode + both passes + exchanges, all shards
      if (stats) *stats = st;
      if (opts->verbose)
        fprintf(stderr, "  [smg] n=%lld k=%d gpus=%d%s  decode %.2f ms, pass1 %.2f, exchange+rc-lookup %.2f, pass2 %.2f "
                "(slowest shard each), wall incl. H2D %.2f ms\n", (long long) st.nels, tv->kmer, ngpus,
                c->virt ? " (virtual shards on one device)" : "", st.ms_decode, st.ms_pass1, st.ms_rclookup, st.ms_pass2, wall);
    }
  if (rc == SMG_OK && labels)
    { // the shards' records, one after the other; their number must be the plot's weight on the labelled pixels
      int64_t want = 0, have = 0;
      for (int cell = 0; cell < SMG_PLOT_CELLS; cell++) if (labels[cell]) want += plot[cell];
      for (int r = 0; r < ngpus; r++) have += c->nrec[r];
      const int rw = c->W + 1;
      uint64_t *all = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) (have > 0 ? have : 1) * rw);
      if (have != want) rc = fail(errbuf, errlen, SMG_ENODEV, "internal error: pair list and plot disagree%s");
      else if (!all) rc = fail(errbuf, errlen, SMG_ENOMEM, "out of host memory for the pair list%s");
      else
        { int64_t o = 0;
          for (int r = 0; r < ngpus; r++)
            { if (c->nrec[r]) memcpy(all + o * rw, c->h_rec[r], sizeof(uint64_t) * (size_t) c->nrec[r] * rw);
              o += c->nrec[r];
            }
          *records = all; *nrec = have; *rec_words = rw; all = NULL;
        }
      free(all);
    }
  for (int r = 0; r < ngpus; r++) free(c->h_rec[r]);
  if (!c->virt) for (int r = 0; r < ngpus; r++) c->api.CommDestroy(c->comm[r]);
  pthread_barrier_destroy(&c->bar);
  pthread_mutex_destroy(&c->big_mu);
  delete c;
  return rc;
}
