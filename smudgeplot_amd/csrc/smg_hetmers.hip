// smg_hetmers.hip -- MI355X (gfx950) heterozygous k-mer pair engine: kernels + C ABI.
//
// What it computes (reference: src/lib/PloidyPlot.c, restated in SURVEY.md section 8a):
//   pairs   = {(x,y) in table : x,y differ at exactly one base, cnt_x+cnt_y <= 1000}
//   deg(x)  = number of pairs containing x   (uint8, wraps mod 256: PloidyPlot.c:163,535)
//   plot[cnt_x+cnt_y][min(cnt_x,cnt_y)] += 1 for pairs with deg(x) <= 1 and deg(y) <= 1
//
// How (NOT the reference's k-level 4-way merge recursion, PloidyPlot.c:454-700, 851-923):
//   * Window scan.  In the sorted table the partners of x at a position p >= p0 share the
//     first p0 bases with x, so they sit in x's "window block" -- a handful of neighbouring
//     entries.  One thread per entry walks its block and tests neighbours with
//     XOR / popcount; blocks longer than WIN_LIM entries switch to binary searches.
//   * Reverse-complement half-scan.  A conditioned table contains every k-mer together with
//     its reverse complement at the same count (Symmex; PloidyPlot.c:1395-1414), and the
//     complement maps a pair at position p onto a pair at position k-1-p.  So only the
//     suffix-side positions p >= ceil((k-1)/2) are scanned (tiny blocks), pairs at p > k-1-p
//     count twice, and   deg(x) = S_all(x) + S_hi(rc(x))   where S_all / S_hi are the
//     numbers of pairs of x at p >= p0 / at p > k-1-p.  The second term is delivered by one
//     "request" per entry that owns such a pair, looked up through a bucket directory.
//     The symmetry is PROVEN per run (exact look-up of every complement, or a 128-bit
//     multiset fingerprint); if it fails the general path below is taken.
//   * General path.  Positions p < p0 are resolved by directory look-ups of the 3 variants;
//     nothing is assumed about the table.  Slow, exact, used only for asymmetric input.
//
// All arithmetic is integer; results are bit-exact against the reference.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>

#include "smg_hetmers.h"
#include "smg_device.hpp"

#define WIN_LIM   32          // window blocks up to this many entries are walked linearly
#define TPB       256

// ------------------------------------------------------------------------------------------
//  Device-resident run state
// ------------------------------------------------------------------------------------------

struct Ctrl                    // small control block in device memory, zeroed per run
{ u64      nreq;               // requests emitted (may exceed capacity => rerun)
  u64      missing;            // complements absent / wrong count
  u64      unsorted;           // order violations found while building the directory
  u64      fp[4];              // fingerprints: T (2 seeds), rc(T) (2 seeds)
  u64      npairs;             // weighted pairs entered into the plot
  u64      route_cnt[16];      // per-destination counters for route
  u64      route_cur[16];
};

struct Tab                     // kernel argument block (passed by value)
{ const u64      *keys;
  const uint16_t *cnt;
  uint8_t        *deg;
  int64_t         n;
  Geo             g;
  Dir             dir;
};

// ------------------------------------------------------------------------------------------
//  Kernels
// ------------------------------------------------------------------------------------------

// format F records -> interleaved left-aligned words + counts
// (restates Current_Entry, libfastk.c:1230-1269: prefix from the index, suffix from the record)
__global__ void __launch_bounds__(TPB)
k_decode(const uint8_t *__restrict__ rec, const int64_t *__restrict__ index, int ixlen,
         int ibyte, int kbyte, int W, int64_t n, u64 *__restrict__ keys,
         uint16_t *__restrict__ cnt)
{ const int64_t i = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  // smallest p with index[p] > i
  int lo = 0, hi = ixlen - 1;
  while (lo < hi)
    { const int m = (lo + hi) >> 1;
      if (index[m] <= i) lo = m + 1; else hi = m;
    }
  const int hbyte = kbyte - ibyte;
  const uint8_t *r = rec + i * (int64_t) (hbyte + 2);
  u64 word = 0;
  int w = 0;
  for (int b = 0; b < 8 * W; b++)
    { unsigned v = 0;
      if (b < ibyte)      v = (lo >> (8 * (ibyte - 1 - b))) & 0xFF;
      else if (b < kbyte) v = r[b - ibyte];
      word = (word << 8) | v;
      if ((b & 7) == 7) { keys[i * W + w] = word; w++; word = 0; }
    }
  cnt[i] = (uint16_t) (r[hbyte] | (r[hbyte + 1] << 8));
}

// bucket directory + strict-order validation
template <int W> __global__ void __launch_bounds__(TPB)
k_directory(Tab t, uint32_t *__restrict__ bstart, Ctrl *__restrict__ ctrl)
{ const int64_t i = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i > t.n) return;
  int64_t bprev = -1, bcur = t.dir.nb;
  if (i > 0)
    bprev = (int64_t) ((t.keys[(i - 1) * W] - t.dir.base) >> t.dir.shift);
  if (i < t.n)
    { bcur = (int64_t) ((t.keys[i * W] - t.dir.base) >> t.dir.shift);
      if (i > 0 && !key_lt<W>(load_key<W>(t.keys, i - 1), load_key<W>(t.keys, i)))
        atomicAdd(&ctrl->unsorted, 1ull);
    }
  for (int64_t b = bprev + 1; b <= bcur; b++) bstart[b] = (uint32_t) i;
}

// ---- pass 1 -------------------------------------------------------------------------------
// One thread per entry.  SYM: scan positions >= p0 inside the window block, write
// deg = S_all, emit a request (rc(x), cnt, S_hi) when S_hi > 0 (or for every entry when
// emit_all), accumulate the fingerprints when want_fp.
// !SYM (general path): additionally resolve positions < p0 by directory look-ups; deg = all.

template <int W, bool SYM> __global__ void __launch_bounds__(TPB)
k_pass1(Tab t, int64_t lo, int64_t hi, int emit_all, int want_fp, u64 *__restrict__ req,
        int64_t req_cap, Ctrl *__restrict__ ctrl)
{ const int64_t i = lo + (int64_t) blockIdx.x * TPB + threadIdx.x;
  const bool live = i < hi;
  const Geo g = t.g;
  unsigned s_all = 0, s_hi = 0;
  Key<W> x;
  unsigned c = 0;
#pragma unroll
  for (int w = 0; w < W; w++) x.w[w] = 0;

  if (live)
    { x = load_key<W>(t.keys, i);
      c = t.cnt[i];
      bool big = false;
      if (i + WIN_LIM < t.n) big |= same_block<W>(x, load_key<W>(t.keys, i + WIN_LIM), g);
      if (i - WIN_LIM >= 0)  big |= same_block<W>(x, load_key<W>(t.keys, i - WIN_LIM), g);
      if (!big)
        { for (int64_t j = i + 1; j < t.n; j++)
            { const Key<W> y = load_key<W>(t.keys, j);
              if (!same_block<W>(x, y, g)) break;
              const int p = pair_pos<W>(x, y);
              if (p >= 0 && c + (unsigned) t.cnt[j] <= SMG_SMAX)
                { s_all++; s_hi += (p != g.k - 1 - p); }
            }
          for (int64_t j = i - 1; j >= 0; j--)
            { const Key<W> y = load_key<W>(t.keys, j);
              if (!same_block<W>(x, y, g)) break;
              const int p = pair_pos<W>(x, y);
              if (p >= 0 && c + (unsigned) t.cnt[j] <= SMG_SMAX)
                { s_all++; s_hi += (p != g.k - 1 - p); }
            }
        }
      else
        { // block bounds by bisection on the (monotone) same_block predicate
          int64_t a = 0, b = i;
          while (a < b)
            { const int64_t m = (a + b) >> 1;
              if (same_block<W>(x, load_key<W>(t.keys, m), g)) b = m; else a = m + 1;
            }
          const int64_t blo = a;
          a = i + 1; b = t.n;
          while (a < b)
            { const int64_t m = (a + b) >> 1;
              if (!same_block<W>(x, load_key<W>(t.keys, m), g)) b = m; else a = m + 1;
            }
          const int64_t bhi = a;
          for (int p = g.p0; p < g.k; p++)
            for (int d = 1; d <= 3; d++)
              { const Key<W> y = flip_base<W>(x, p, d);
                const int64_t j = lower_bound_key<W>(t.keys, blo, bhi, y);
                if (j < bhi && key_eq<W>(load_key<W>(t.keys, j), y)
                    && c + (unsigned) t.cnt[j] <= SMG_SMAX)
                  { s_all++; s_hi += (p != g.k - 1 - p); }
              }
        }
      if (!SYM)
        { for (int p = 0; p < g.p0; p++)
            for (int d = 1; d <= 3; d++)
              { const int64_t j = find_key<W>(t.keys, t.dir, flip_base<W>(x, p, d));
                if (j >= 0 && c + (unsigned) t.cnt[j] <= SMG_SMAX) s_all++;
              }
        }
      t.deg[i] = (uint8_t) s_all;
    }

  if (SYM)
    { const bool emit = live && (emit_all || s_hi > 0);
      const u64 mask = __ballot(emit);
      if (mask)
        { const int lane = threadIdx.x & 63;
          u64 base = 0;
          if (lane == __ffsll((long long) mask) - 1)
            base = atomicAdd(&ctrl->nreq, (u64) __popcll(mask));
          base = __shfl(base, __ffsll((long long) mask) - 1, 64);
          if (emit)
            { const u64 slot = base + __popcll(mask & ((1ull << lane) - 1));
              if ((int64_t) slot < req_cap)
                { const Key<W> r = revcomp<W>(x, g.k);
                  u64 *o = req + slot * (W + 1);
#pragma unroll
                  for (int w = 0; w < W; w++) o[w] = r.w[w];
                  o[W] = (u64) c | ((u64) (s_hi & 0xFF) << 16);
                }
            }
        }
      if (want_fp)
        { u64 f0 = 0, f1 = 0, f2 = 0, f3 = 0;
          if (live)
            { const Key<W> r = revcomp<W>(x, g.k);
              f0 = hash_entry<W>(x, c, 0x243f6a8885a308d3ull);
              f1 = hash_entry<W>(x, c, 0x13198a2e03707344ull);
              f2 = hash_entry<W>(r, c, 0x243f6a8885a308d3ull);
              f3 = hash_entry<W>(r, c, 0x13198a2e03707344ull);
            }
          f0 = wave_sum_u64(f0); f1 = wave_sum_u64(f1);
          f2 = wave_sum_u64(f2); f3 = wave_sum_u64(f3);
          if ((threadIdx.x & 63) == 0)
            { atomicAdd(&ctrl->fp[0], f0); atomicAdd(&ctrl->fp[1], f1);
              atomicAdd(&ctrl->fp[2], f2); atomicAdd(&ctrl->fp[3], f3);
            }
        }
    }
}

// requests -> degrees (the S_hi(rc(x)) term), and the per-request symmetry proof
template <int W> __global__ void __launch_bounds__(TPB)
k_apply(Tab t, const u64 *__restrict__ req, int64_t nreq, Ctrl *__restrict__ ctrl)
{ const int64_t r = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (r >= nreq) return;
  Key<W> y;
  const u64 *q = req + r * (W + 1);
#pragma unroll
  for (int w = 0; w < W; w++) y.w[w] = q[w];
  const unsigned c = (unsigned) (q[W] & 0xFFFF), v = (unsigned) ((q[W] >> 16) & 0xFF);
  const int64_t j = find_key<W>(t.keys, t.dir, y);
  if (j < 0 || t.cnt[j] != c) { atomicAdd(&ctrl->missing, 1ull); return; }
  if (v) deg_add(t.deg, j, v, t.g.wrap);
}

// exact symmetry proof for every entry of [lo,hi) against the local table (single GPU)
template <int W> __global__ void __launch_bounds__(TPB)
k_verify(Tab t, int64_t lo, int64_t hi, Ctrl *__restrict__ ctrl)
{ const int64_t i = lo + (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i >= hi) return;
  const Key<W> r = revcomp<W>(load_key<W>(t.keys, i), t.g.k);
  const int64_t j = find_key<W>(t.keys, t.dir, r);
  if (j < 0 || t.cnt[j] != t.cnt[i]) atomicAdd(&ctrl->missing, 1ull);
}

// ---- pass 2 -------------------------------------------------------------------------------
// Entry i with (wrapped) degree <= 1 looks for its partners j > i; the pair enters the plot
// when deg(j) <= 1 too.  SYM: pairs at p != k-1-p stand for themselves and their complement
// image (weight 2).  General: every position is visited, weight 1.

SMG_DEV void plot_add(u64 *__restrict__ plot, unsigned ci, unsigned cj, unsigned wgt)
{ const unsigned s = ci + cj, m = ci < cj ? ci : cj;
  atomicAdd(plot + (size_t) s * SMG_PLOT_COLS + m, (u64) wgt);
}

template <int W, bool SYM> __global__ void __launch_bounds__(TPB)
k_pass2(Tab t, int64_t lo, int64_t hi, u64 *__restrict__ plot, Ctrl *__restrict__ ctrl)
{ const int64_t i = lo + (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i >= hi) return;
  const Geo g = t.g;
  const unsigned di = t.deg[i];
  if (di > 1) return;
  if (di == 0 && !g.wrap) return;            // no wrap possible: degree 0 means no pair at all
  const Key<W> x = load_key<W>(t.keys, i);
  const unsigned c = t.cnt[i];
  unsigned found = 0;

  bool big = false;
  if (i + WIN_LIM < t.n) big = same_block<W>(x, load_key<W>(t.keys, i + WIN_LIM), g);
  if (!big)
    { for (int64_t j = i + 1; j < t.n; j++)
        { const Key<W> y = load_key<W>(t.keys, j);
          if (!same_block<W>(x, y, g)) break;
          const int p = pair_pos<W>(x, y);
          if (p >= 0)
            { const unsigned cj = t.cnt[j];
              if (c + cj <= SMG_SMAX && t.deg[j] <= 1)
                { const unsigned wgt = (SYM && p != g.k - 1 - p) ? 2 : 1;
                  plot_add(plot, c, cj, wgt); found += wgt;
                }
            }
        }
    }
  else
    { int64_t a = i + 1, b = t.n;
      while (a < b)
        { const int64_t m = (a + b) >> 1;
          if (!same_block<W>(x, load_key<W>(t.keys, m), g)) b = m; else a = m + 1;
        }
      const int64_t bhi = a;
      for (int p = g.p0; p < g.k; p++)
        for (int d = 1; d <= 3; d++)
          { const Key<W> y = flip_base<W>(x, p, d);
            if (!key_lt<W>(x, y)) continue;
            const int64_t j = lower_bound_key<W>(t.keys, i + 1, bhi, y);
            if (j < bhi && key_eq<W>(load_key<W>(t.keys, j), y))
              { const unsigned cj = t.cnt[j];
                if (c + cj <= SMG_SMAX && t.deg[j] <= 1)
                  { const unsigned wgt = (SYM && p != g.k - 1 - p) ? 2 : 1;
                    plot_add(plot, c, cj, wgt); found += wgt;
                  }
              }
          }
    }
  if (!SYM)
    { for (int p = 0; p < g.p0; p++)
        for (int d = 1; d <= 3; d++)
          { const Key<W> y = flip_base<W>(x, p, d);
            if (!key_lt<W>(x, y)) continue;
            const int64_t j = find_key<W>(t.keys, t.dir, y);
            if (j >= 0)
              { const unsigned cj = t.cnt[j];
                if (c + cj <= SMG_SMAX && t.deg[j] <= 1) { plot_add(plot, c, cj, 1); found++; }
              }
          }
    }
  if (found) atomicAdd(&ctrl->npairs, (u64) found);
}

// ---- routing of requests to the rank that owns the complement (sharded runs) ----------------

template <int W> SMG_DEV int dest_rank(const u64 *q, const u64 *__restrict__ split, int nranks)
{ Key<W> y;
#pragma unroll
  for (int w = 0; w < W; w++) y.w[w] = q[w];
  int r = 0;
  for (int s = 0; s < nranks - 1; s++)
    { Key<W> sp;
#pragma unroll
      for (int w = 0; w < W; w++) sp.w[w] = split[s * W + w];
      if (!key_lt<W>(y, sp)) r = s + 1;
    }
  return r;
}

template <int W> __global__ void __launch_bounds__(TPB)
k_route_count(const u64 *__restrict__ req, int64_t nreq, const u64 *__restrict__ split,
              int nranks, Ctrl *__restrict__ ctrl)
{ const int64_t r = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (r >= nreq) return;
  atomicAdd(&ctrl->route_cnt[dest_rank<W>(req + r * (W + 1), split, nranks)], 1ull);
}

template <int W> __global__ void __launch_bounds__(TPB)
k_route_scatter(const u64 *__restrict__ req, int64_t nreq, const u64 *__restrict__ split,
                int nranks, u64 *__restrict__ out, Ctrl *__restrict__ ctrl)
{ const int64_t r = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (r >= nreq) return;
  const u64 *q = req + r * (W + 1);
  const int d = dest_rank<W>(q, split, nranks);
  const u64 slot = atomicAdd(&ctrl->route_cur[d], 1ull);
  u64 *o = out + slot * (W + 1);
#pragma unroll
  for (int w = 0; w <= W; w++) o[w] = q[w];
}

// ------------------------------------------------------------------------------------------
//  Host side
// ------------------------------------------------------------------------------------------

struct smg_engine
{ int          device;
  hipStream_t  stream;
  int          kmer, W;
  int64_t      n;
  const u64   *keys;          // bound or owned
  const uint16_t *cnt;
  u64         *own_keys;      // owned copies (decode path)
  uint16_t    *own_cnt;
  uint8_t     *deg;  int64_t deg_cap;
  uint32_t    *bstart; int64_t bstart_cap;
  u64         *req;  int64_t req_cap;      // capacity in records
  u64         *d_split;
  Ctrl        *ctrl;
  Ctrl        *h_ctrl;        // pinned mirror
  Geo          geo;
  Dir          dir;
  bool         prepared;
  smg_stats    st;
  hipEvent_t   ev[8];
};

static int fail(char *errbuf, size_t errlen, int code, const char *fmt, const char *a = "")
{ if (errbuf && errlen) snprintf(errbuf, errlen, fmt, a);
  return code;
}

#define HIPCHK(call)                                                                         \
  do { hipError_t _e = (call);                                                               \
       if (_e != hipSuccess)                                                                 \
         return fail(errbuf, errlen, _e == hipErrorOutOfMemory ? SMG_ENOMEM : SMG_ENODEV,     \
                     "HIP error: %s (" #call ")", hipGetErrorString(_e)); } while (0)

extern "C" const char *smg_version(void) { return "smudgeplot_amd 0.1 (hetmers engine, gfx950)"; }

extern "C" int smg_device_count(void)
{ int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" smg_engine *smg_engine_create(int device, void *stream, char *errbuf, size_t errlen)
{ int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    { fail(errbuf, errlen, SMG_ENODEV, "no HIP device available (this engine has no CPU fallback)%s");
      return NULL;
    }
  if (device < 0 || device >= n)
    { fail(errbuf, errlen, SMG_EINVAL, "device ordinal out of range%s"); return NULL; }
  if (hipSetDevice(device) != hipSuccess)
    { fail(errbuf, errlen, SMG_ENODEV, "cannot select HIP device%s"); return NULL; }
  smg_engine *e = new (std::nothrow) smg_engine();
  if (!e) { fail(errbuf, errlen, SMG_ENOMEM, "out of host memory%s"); return NULL; }
  memset(e, 0, sizeof(*e));
  e->device = device;
  e->stream = (hipStream_t) stream;
  if (hipMalloc(&e->ctrl, sizeof(Ctrl)) != hipSuccess
      || hipHostMalloc(&e->h_ctrl, sizeof(Ctrl)) != hipSuccess
      || hipMalloc(&e->d_split, sizeof(u64) * 16 * 4) != hipSuccess)
    { fail(errbuf, errlen, SMG_ENOMEM, "cannot allocate the control block%s");
      delete e; return NULL;
    }
  for (int i = 0; i < 8; i++) hipEventCreate(&e->ev[i]);
  return e;
}

extern "C" void smg_engine_destroy(smg_engine *e)
{ if (!e) return;
  hipSetDevice(e->device);
  hipStreamSynchronize(e->stream);
  hipFree(e->own_keys); hipFree(e->own_cnt); hipFree(e->deg); hipFree(e->bstart);
  hipFree(e->req); hipFree(e->ctrl); hipFree(e->d_split); hipHostFree(e->h_ctrl);
  for (int i = 0; i < 8; i++) hipEventDestroy(e->ev[i]);
  delete e;
}

static int set_table(smg_engine *e, int kmer, int64_t nels, char *errbuf, size_t errlen)
{ if (kmer < 1 || kmer > SMG_MAX_KMER)
    return fail(errbuf, errlen, SMG_EINVAL, "k-mer length out of range (1..128)%s");
  if (nels < 0 || nels >= 0xFFFFFFF0ll)
    return fail(errbuf, errlen, SMG_EINVAL, "table shard too large (entries per GPU must be < 2^32-16)%s");
  e->kmer = kmer;
  e->W = (kmer + 31) / 32;
  e->n = nels;
  e->prepared = false;
  memset(&e->st, 0, sizeof(e->st));
  e->st.nels = nels;
  e->st.key_words = e->W;
  return SMG_OK;
}

static void set_geo(smg_engine *e, bool sym)
{ Geo &g = e->geo;
  g.k = e->kmer;
  // first scanned position p0 = ceil((k-1)/2) = k/2 (integer division) for both parities: on
  // the symmetric path positions below it are the mirror images of the scanned ones, on the
  // general path they are resolved by directory look-ups
  (void) sym;
  g.p0 = e->kmer / 2;
  g.pw = g.p0 >> 5;
  const int r = g.p0 & 31;
  g.pmask = r ? ~0ull << (64 - 2 * r) : 0ull;
  g.mid = (e->kmer & 1) ? (e->kmer - 1) / 2 : -1;
  g.wrap = e->kmer > 85;
}

static int grow(void **p, int64_t *cap, int64_t need_bytes, char *errbuf, size_t errlen)
{ if (*cap >= need_bytes) return SMG_OK;
  if (*p) hipFree(*p);
  *p = NULL; *cap = 0;
  HIPCHK(hipMalloc(p, (size_t) need_bytes));
  *cap = need_bytes;
  return SMG_OK;
}

extern "C" int smg_engine_bind(smg_engine *e, int kmer, int64_t nels, const uint64_t *d_keys,
                               const uint16_t *d_counts, char *errbuf, size_t errlen)
{ if (!e) return fail(errbuf, errlen, SMG_EINVAL, "null engine%s");
  if (nels > 0 && (!d_keys || !d_counts)) return fail(errbuf, errlen, SMG_EINVAL, "null table pointer%s");
  HIPCHK(hipSetDevice(e->device));
  int rc = set_table(e, kmer, nels, errbuf, errlen);
  if (rc) return rc;
  e->keys = (const u64 *) d_keys;
  e->cnt = d_counts;
  return SMG_OK;
}

extern "C" int smg_engine_decode(smg_engine *e, int kmer, int ibyte, int64_t nels,
                                 const uint8_t *d_records, const int64_t *d_prefix_index,
                                 char *errbuf, size_t errlen)
{ if (!e) return fail(errbuf, errlen, SMG_EINVAL, "null engine%s");
  if (ibyte < 1 || ibyte > 3) return fail(errbuf, errlen, SMG_EINVAL, "ibyte must be 1, 2 or 3%s");
  HIPCHK(hipSetDevice(e->device));
  int rc = set_table(e, kmer, nels, errbuf, errlen);
  if (rc) return rc;
  const int kbyte = (kmer + 3) >> 2;
  if (kbyte <= ibyte) return fail(errbuf, errlen, SMG_EINVAL, "k-mer shorter than the index prefix%s");
  hipFree(e->own_keys); hipFree(e->own_cnt);
  e->own_keys = NULL; e->own_cnt = NULL;
  HIPCHK(hipMalloc(&e->own_keys, sizeof(u64) * (size_t) (nels > 0 ? nels : 1) * e->W));
  HIPCHK(hipMalloc(&e->own_cnt, sizeof(uint16_t) * (size_t) (nels > 0 ? nels : 1)));
  e->keys = e->own_keys; e->cnt = e->own_cnt;
  hipEventRecord(e->ev[0], e->stream);
  if (nels > 0)
    { const unsigned nblk = (unsigned) ((nels + TPB - 1) / TPB);
      hipLaunchKernelGGL(k_decode, dim3(nblk), dim3(TPB), 0, e->stream, d_records,
                         d_prefix_index, 1 << (8 * ibyte), ibyte, kbyte, e->W, nels,
                         e->own_keys, e->own_cnt);
    }
  hipEventRecord(e->ev[1], e->stream);
  HIPCHK(hipStreamSynchronize(e->stream));
  float ms = 0; hipEventElapsedTime(&ms, e->ev[0], e->ev[1]);
  e->st.ms_decode = ms;
  return SMG_OK;
}

template <int W> static void launch_directory(smg_engine *e, Tab &t)
{ const unsigned nblk = (unsigned) ((e->n + 1 + TPB - 1) / TPB);
  hipLaunchKernelGGL(k_directory<W>, dim3(nblk), dim3(TPB), 0, e->stream, t, e->bstart, e->ctrl);
}

static Tab make_tab(smg_engine *e)
{ Tab t;
  t.keys = e->keys; t.cnt = e->cnt; t.deg = e->deg; t.n = e->n; t.g = e->geo; t.dir = e->dir;
  return t;
}

// allocate degrees, build the directory, validate the order; zero the control block
static int prepare(smg_engine *e, char *errbuf, size_t errlen)
{ HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemsetAsync(e->ctrl, 0, sizeof(Ctrl), e->stream));
  int64_t cap;
  cap = e->deg_cap;
  int rc = grow((void **) &e->deg, &cap, ((e->n + 3) & ~3ll) + 4, errbuf, errlen);
  e->deg_cap = cap;
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(e->deg, 0, (size_t) (((e->n + 3) & ~3ll) + 4), e->stream));

  // directory geometry: ~2-4 entries per bucket over the shard's first-word range
  u64 first = 0, last = 0;
  if (e->n > 0)
    { HIPCHK(hipMemcpyAsync(&first, e->keys, 8, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipMemcpyAsync(&last, e->keys + (size_t) (e->n - 1) * e->W, 8, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
    }
  if (last < first) return fail(errbuf, errlen, SMG_EFORMAT, "table is not sorted%s");
  int bits = 4;
  while (bits < 28 && (1ll << (bits + 1)) <= e->n / 2) bits++;
  const u64 span = last - first;
  int shift = 0;
  while (shift < 63 && (span >> shift) >= (1ull << bits)) shift++;
  if ((span >> shift) >= (1ull << bits)) shift = 64 - bits;   // unreachable guard
  e->dir.base = first;
  e->dir.shift = shift;
  e->dir.nb = (uint32_t) ((span >> shift) + 1);
  cap = e->bstart_cap;
  rc = grow((void **) &e->bstart, &cap, sizeof(uint32_t) * ((int64_t) e->dir.nb + 2), errbuf, errlen);
  e->bstart_cap = cap;
  if (rc) return rc;
  e->dir.bstart = e->bstart;

  Tab t = make_tab(e);
  switch (e->W)
  { case 1: launch_directory<1>(e, t); break;
    case 2: launch_directory<2>(e, t); break;
    case 3: launch_directory<3>(e, t); break;
    default: launch_directory<4>(e, t); break;
  }
  HIPCHK(hipGetLastError());
  e->prepared = true;
  return SMG_OK;
}

static int read_ctrl(smg_engine *e, char *errbuf, size_t errlen)
{ HIPCHK(hipMemcpyAsync(e->h_ctrl, e->ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return SMG_OK;
}

template <int W, bool SYM> static void launch_pass1(smg_engine *e, int emit_all, int want_fp)
{ Tab t = make_tab(e);
  if (e->n <= 0) return;
  const unsigned nblk = (unsigned) ((e->n + TPB - 1) / TPB);
  hipLaunchKernelGGL((k_pass1<W, SYM>), dim3(nblk), dim3(TPB), 0, e->stream, t, (int64_t) 0,
                     e->n, emit_all, want_fp, e->req, e->req_cap, e->ctrl);
}

template <int W, bool SYM> static void launch_pass2(smg_engine *e, int64_t *d_plot)
{ Tab t = make_tab(e);
  if (e->n <= 0) return;
  const unsigned nblk = (unsigned) ((e->n + TPB - 1) / TPB);
  hipLaunchKernelGGL((k_pass2<W, SYM>), dim3(nblk), dim3(TPB), 0, e->stream, t, (int64_t) 0,
                     e->n, (u64 *) d_plot, e->ctrl);
}

template <int W> static void launch_apply(smg_engine *e, const u64 *req, int64_t nreq)
{ Tab t = make_tab(e);
  if (nreq <= 0) return;
  const unsigned nblk = (unsigned) ((nreq + TPB - 1) / TPB);
  hipLaunchKernelGGL(k_apply<W>, dim3(nblk), dim3(TPB), 0, e->stream, t, req, nreq, e->ctrl);
}

template <int W> static void launch_verify(smg_engine *e)
{ Tab t = make_tab(e);
  if (e->n <= 0) return;
  const unsigned nblk = (unsigned) ((e->n + TPB - 1) / TPB);
  hipLaunchKernelGGL(k_verify<W>, dim3(nblk), dim3(TPB), 0, e->stream, t, (int64_t) 0, e->n, e->ctrl);
}

#define DISPATCH_W(e, CALL)                                                                   \
  switch ((e)->W) { case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break;  \
                    default: CALL(4); break; }

static int ensure_req(smg_engine *e, int64_t records, char *errbuf, size_t errlen)
{ if (records < 1024) records = 1024;
  if (e->req_cap >= records) return SMG_OK;
  if (e->req) hipFree(e->req);
  e->req = NULL; e->req_cap = 0;
  HIPCHK(hipMalloc(&e->req, sizeof(u64) * (size_t) records * (e->W + 1)));
  e->req_cap = records;
  return SMG_OK;
}

// pass 1 of the symmetric path, with the capacity retry
static int do_pass1_sym(smg_engine *e, int emit_all, int want_fp, char *errbuf, size_t errlen)
{ int rc;
  if (!e->prepared && (rc = prepare(e, errbuf, errlen))) return rc;
  set_geo(e, true);
  rc = ensure_req(e, emit_all ? e->n : e->n / 4 + 1024, errbuf, errlen);
  if (rc) return rc;
  for (int attempt = 0; attempt < 2; attempt++)
    { hipEventRecord(e->ev[2], e->stream);
#define CALL(WW) launch_pass1<WW, true>(e, emit_all, want_fp)
      DISPATCH_W(e, CALL)
#undef CALL
      hipEventRecord(e->ev[3], e->stream);
      HIPCHK(hipGetLastError());
      if ((rc = read_ctrl(e, errbuf, errlen))) return rc;
      if (e->h_ctrl->unsorted)
        return fail(errbuf, errlen, SMG_EFORMAT, "table entries are not strictly increasing%s");
      if ((int64_t) e->h_ctrl->nreq <= e->req_cap) break;
      // request list overflowed its first-guess capacity: size it exactly and redo the pass
      const int64_t need = (int64_t) e->h_ctrl->nreq;
      if ((rc = ensure_req(e, need, errbuf, errlen))) return rc;
      HIPCHK(hipMemsetAsync(&e->ctrl->nreq, 0, sizeof(u64), e->stream));
      HIPCHK(hipMemsetAsync(e->ctrl->fp, 0, sizeof(u64) * 4, e->stream));
    }
  float ms = 0; hipEventElapsedTime(&ms, e->ev[2], e->ev[3]);
  e->st.ms_pass1 = ms;
  e->st.nrequests = (int64_t) e->h_ctrl->nreq;
  return SMG_OK;
}

extern "C" int smg_engine_pass1(smg_engine *e, int symcheck, char *errbuf, size_t errlen)
{ if (!e) return fail(errbuf, errlen, SMG_EINVAL, "null engine%s");
  HIPCHK(hipSetDevice(e->device));
  e->prepared = false;
  return do_pass1_sym(e, symcheck == SMG_SYM_EXACT, symcheck == SMG_SYM_HASH, errbuf, errlen);
}

extern "C" int64_t smg_engine_nreq(smg_engine *e) { return e ? (int64_t) e->h_ctrl->nreq : 0; }
extern "C" int smg_engine_record_words(smg_engine *e) { return e ? e->W + 1 : 0; }

extern "C" int smg_engine_apply(smg_engine *e, const uint64_t *d_recv, int64_t nrecv,
                                int64_t *missing, char *errbuf, size_t errlen)
{ if (!e || !e->prepared) return fail(errbuf, errlen, SMG_EINVAL, "apply before pass1%s");
  HIPCHK(hipSetDevice(e->device));
  hipEventRecord(e->ev[4], e->stream);
#define CALL(WW) launch_apply<WW>(e, (const u64 *) d_recv, nrecv)
  DISPATCH_W(e, CALL)
#undef CALL
  hipEventRecord(e->ev[5], e->stream);
  HIPCHK(hipGetLastError());
  int rc = read_ctrl(e, errbuf, errlen);
  if (rc) return rc;
  float ms = 0; hipEventElapsedTime(&ms, e->ev[4], e->ev[5]);
  e->st.ms_rclookup += ms;
  if (missing) *missing = (int64_t) e->h_ctrl->missing;
  return SMG_OK;
}

extern "C" int smg_engine_apply_own(smg_engine *e, int64_t *missing, char *errbuf, size_t errlen)
{ if (!e) return fail(errbuf, errlen, SMG_EINVAL, "null engine%s");
  return smg_engine_apply(e, (const uint64_t *) e->req, (int64_t) e->h_ctrl->nreq, missing, errbuf, errlen);
}

extern "C" int smg_engine_symhash(smg_engine *e, uint64_t out[4], char *errbuf, size_t errlen)
{ if (!e || !out) return fail(errbuf, errlen, SMG_EINVAL, "null argument%s");
  for (int i = 0; i < 4; i++) out[i] = e->h_ctrl->fp[i];
  return SMG_OK;
}

extern "C" int smg_engine_route(smg_engine *e, const uint64_t *splitters, int nranks,
                                uint64_t *d_send, int64_t capacity, int64_t *counts,
                                char *errbuf, size_t errlen)
{ if (!e || !counts || nranks < 1 || nranks > 16)
    return fail(errbuf, errlen, SMG_EINVAL, "bad route arguments (1..16 ranks)%s");
  HIPCHK(hipSetDevice(e->device));
  const int64_t nreq = (int64_t) e->h_ctrl->nreq;
  if (nreq > capacity) return fail(errbuf, errlen, SMG_EINVAL, "send buffer too small%s");
  if (nranks > 1)
    HIPCHK(hipMemcpyAsync(e->d_split, splitters, sizeof(u64) * (nranks - 1) * e->W,
                          hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemsetAsync(e->ctrl->route_cnt, 0, sizeof(u64) * 32, e->stream));
  const unsigned nblk = (unsigned) ((nreq + TPB - 1) / TPB);
  if (nreq > 0)
    {
#define CALL(WW) hipLaunchKernelGGL(k_route_count<WW>, dim3(nblk), dim3(TPB), 0, e->stream, \
                                    e->req, nreq, e->d_split, nranks, e->ctrl)
      DISPATCH_W(e, CALL)
#undef CALL
    }
  int rc = read_ctrl(e, errbuf, errlen);
  if (rc) return rc;
  u64 cur[16], acc = 0;
  for (int r = 0; r < 16; r++)
    { cur[r] = acc;
      if (r < nranks) { counts[r] = (int64_t) e->h_ctrl->route_cnt[r]; acc += e->h_ctrl->route_cnt[r]; }
    }
  HIPCHK(hipMemcpyAsync(e->ctrl->route_cur, cur, sizeof(cur), hipMemcpyHostToDevice, e->stream));
  if (nreq > 0)
    {
#define CALL(WW) hipLaunchKernelGGL(k_route_scatter<WW>, dim3(nblk), dim3(TPB), 0, e->stream, \
                                    e->req, nreq, e->d_split, nranks, (u64 *) d_send, e->ctrl)
      DISPATCH_W(e, CALL)
#undef CALL
    }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->stream));
  return SMG_OK;
}

extern "C" int smg_engine_pass2(smg_engine *e, int64_t *d_plot, char *errbuf, size_t errlen)
{ if (!e || !e->prepared || !d_plot) return fail(errbuf, errlen, SMG_EINVAL, "pass2 before pass1%s");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemsetAsync(d_plot, 0, sizeof(int64_t) * SMG_PLOT_CELLS, e->stream));
  hipEventRecord(e->ev[6], e->stream);
#define CALL(WW) launch_pass2<WW, true>(e, d_plot)
  DISPATCH_W(e, CALL)
#undef CALL
  hipEventRecord(e->ev[7], e->stream);
  HIPCHK(hipGetLastError());
  int rc = read_ctrl(e, errbuf, errlen);
  if (rc) return rc;
  float ms = 0; hipEventElapsedTime(&ms, e->ev[6], e->ev[7]);
  e->st.ms_pass2 = ms;
  e->st.npairs = (int64_t) e->h_ctrl->npairs;
  e->st.path = 1;
  return SMG_OK;
}

extern "C" int smg_engine_stats(smg_engine *e, smg_stats *stats)
{ if (!e || !stats) return SMG_EINVAL;
  *stats = e->st;
  return SMG_OK;
}

// general (assumption-free) path: both passes over every position
static int run_general(smg_engine *e, int64_t *d_plot, char *errbuf, size_t errlen)
{ int rc;
  e->prepared = false;
  if ((rc = prepare(e, errbuf, errlen))) return rc;
  set_geo(e, false);
  hipEventRecord(e->ev[2], e->stream);
#define CALL(WW) launch_pass1<WW, false>(e, 0, 0)
  DISPATCH_W(e, CALL)
#undef CALL
  hipEventRecord(e->ev[3], e->stream);
  HIPCHK(hipMemsetAsync(d_plot, 0, sizeof(int64_t) * SMG_PLOT_CELLS, e->stream));
  hipEventRecord(e->ev[6], e->stream);
#define CALL(WW) launch_pass2<WW, false>(e, d_plot)
  DISPATCH_W(e, CALL)
#undef CALL
  hipEventRecord(e->ev[7], e->stream);
  HIPCHK(hipGetLastError());
  if ((rc = read_ctrl(e, errbuf, errlen))) return rc;
  if (e->h_ctrl->unsorted)
    return fail(errbuf, errlen, SMG_EFORMAT, "table entries are not strictly increasing%s");
  float ms = 0;
  hipEventElapsedTime(&ms, e->ev[2], e->ev[3]); e->st.ms_pass1 += ms;
  hipEventElapsedTime(&ms, e->ev[6], e->ev[7]); e->st.ms_pass2 = ms;
  e->st.npairs = (int64_t) e->h_ctrl->npairs;
  e->st.path = 2;
  return SMG_OK;
}

extern "C" int smg_engine_run(smg_engine *e, int symcheck, int64_t *d_plot, smg_stats *stats,
                              char *errbuf, size_t errlen)
{ if (!e || !d_plot) return fail(errbuf, errlen, SMG_EINVAL, "null argument%s");
  HIPCHK(hipSetDevice(e->device));
  int rc;
  hipEvent_t t0, t1;
  hipEventCreate(&t0); hipEventCreate(&t1);
  hipEventRecord(t0, e->stream);
  e->st.ms_pass1 = e->st.ms_rclookup = e->st.ms_pass2 = 0;
  bool symmetric = false;
  if (symcheck != SMG_SYM_NONE)
    { e->prepared = false;
      rc = do_pass1_sym(e, 0, symcheck == SMG_SYM_HASH, errbuf, errlen);
      if (rc) return rc;
      int64_t missing = 0;
      if ((rc = smg_engine_apply_own(e, &missing, errbuf, errlen))) return rc;
      symmetric = (missing == 0);
      if (symmetric && symcheck == SMG_SYM_HASH)
        symmetric = e->h_ctrl->fp[0] == e->h_ctrl->fp[2] && e->h_ctrl->fp[1] == e->h_ctrl->fp[3];
      if (symmetric && symcheck == SMG_SYM_EXACT)
        { hipEventRecord(e->ev[4], e->stream);
#define CALL(WW) launch_verify<WW>(e)
          DISPATCH_W(e, CALL)
#undef CALL
          hipEventRecord(e->ev[5], e->stream);
          if ((rc = read_ctrl(e, errbuf, errlen))) return rc;
          float ms = 0; hipEventElapsedTime(&ms, e->ev[4], e->ev[5]);
          e->st.ms_rclookup += ms;
          symmetric = (e->h_ctrl->missing == 0);
        }
      if (symmetric && (rc = smg_engine_pass2(e, d_plot, errbuf, errlen))) return rc;
    }
  if (!symmetric && (rc = run_general(e, d_plot, errbuf, errlen))) return rc;
  hipEventRecord(t1, e->stream);
  HIPCHK(hipStreamSynchronize(e->stream));
  float ms = 0; hipEventElapsedTime(&ms, t0, t1);
  e->st.ms_total = ms;
  hipEventDestroy(t0); hipEventDestroy(t1);
  if (stats) *stats = e->st;
  return SMG_OK;
}

// ---- one-shot host entry ----------------------------------------------------------------------

extern "C" int smg_hetmers_run(const smg_table_view *tv, const smg_opts *opts, int64_t *plot,
                               smg_stats *stats, char *errbuf, size_t errlen)
{ if (!tv || !plot) return fail(errbuf, errlen, SMG_EINVAL, "null argument%s");
  const int device = opts ? opts->device : 0;
  const int symcheck = opts ? opts->symcheck : SMG_SYM_EXACT;
  const int verbose = opts ? opts->verbose : 0;
  if (tv->ibyte < 1 || tv->ibyte > 3) return fail(errbuf, errlen, SMG_EINVAL, "ibyte must be 1, 2 or 3%s");
  const int kbyte = (tv->kmer + 3) >> 2, pbyte = kbyte + 2 - tv->ibyte;
  if (pbyte < 3) return fail(errbuf, errlen, SMG_EINVAL, "k-mer shorter than the index prefix%s");
  int64_t sum = 0;
  for (int p = 0; p < tv->nparts; p++) sum += tv->part_nels[p];
  if (sum != tv->nels) return fail(errbuf, errlen, SMG_EFORMAT, "part sizes do not add up to nels%s");

  smg_engine *e = smg_engine_create(device, NULL, errbuf, errlen);
  if (!e) return SMG_ENODEV;
  int rc = SMG_OK;
  uint8_t *d_rec = NULL; int64_t *d_index = NULL, *d_plot = NULL;
  const size_t ixbytes = sizeof(int64_t) << (8 * tv->ibyte);
  hipEvent_t h0, h1;
  hipEventCreate(&h0); hipEventCreate(&h1);
#define BAIL(code, msg) { rc = fail(errbuf, errlen, code, msg "%s"); goto done; }
  if (hipMalloc(&d_rec, (size_t) (tv->nels > 0 ? tv->nels : 1) * pbyte) != hipSuccess
      || hipMalloc(&d_index, ixbytes) != hipSuccess
      || hipMalloc(&d_plot, sizeof(int64_t) * SMG_PLOT_CELLS) != hipSuccess)
    BAIL(SMG_ENOMEM, "out of device memory for the table")
  hipEventRecord(h0, 0);
  { size_t off = 0;
    for (int p = 0; p < tv->nparts; p++)
      { const size_t b = (size_t) tv->part_nels[p] * pbyte;
        if (b && hipMemcpy(d_rec + off, tv->part_data[p], b, hipMemcpyHostToDevice) != hipSuccess)
          BAIL(SMG_ENODEV, "host to device copy failed")
        off += b;
      }
    if (hipMemcpy(d_index, tv->prefix_index, ixbytes, hipMemcpyHostToDevice) != hipSuccess)
      BAIL(SMG_ENODEV, "host to device copy failed")
  }
  hipEventRecord(h1, 0);
  hipEventSynchronize(h1);
  if ((rc = smg_engine_decode(e, tv->kmer, tv->ibyte, tv->nels, d_rec, d_index, errbuf, errlen))) goto done;
  hipFree(d_rec); d_rec = NULL;
  if ((rc = smg_engine_run(e, symcheck, d_plot, NULL, errbuf, errlen))) goto done;
  if (hipMemcpy(plot, d_plot, sizeof(int64_t) * SMG_PLOT_CELLS, hipMemcpyDeviceToHost) != hipSuccess)
    BAIL(SMG_ENODEV, "device to host copy failed")
  { float ms = 0; hipEventElapsedTime(&ms, h0, h1); e->st.ms_h2d = ms; }
  if (stats) *stats = e->st;
  if (verbose)
    fprintf(stderr, "  [smg] n=%lld k=%d path=%s  h2d %.2f ms, decode %.2f, pass1 %.2f, rc-lookup %.2f, "
            "pass2 %.2f, total(device) %.2f ms => %.3g k-mers/s\n",
            (long long) e->st.nels, tv->kmer, e->st.path == 1 ? "rc-half-scan" : "general",
            e->st.ms_h2d, e->st.ms_decode, e->st.ms_pass1, e->st.ms_rclookup, e->st.ms_pass2,
            e->st.ms_total, e->st.ms_total > 0 ? e->st.nels / (e->st.ms_total * 1e-3) : 0.0);
done:
#undef BAIL
  hipEventDestroy(h0); hipEventDestroy(h1);
  if (d_rec) hipFree(d_rec);
  if (d_index) hipFree(d_index);
  if (d_plot) hipFree(d_plot);
  smg_engine_destroy(e);
  return rc;
}
