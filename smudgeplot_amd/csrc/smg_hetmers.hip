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
#include <vector>
#include <rocprim/rocprim.hpp>

#include "smg_hetmers.h"
#include "smg_device.hpp"
#include "smg_fast.hpp"
#include "smg_pass1d.hpp"
#include "smg_lookup.hpp"

// The environment, in three classes (round 6: the product library no longer carries tuning switches):
//   documented modes    getenv(): SMG_VIRTUAL_SHARDS, SMG_SEQUENTIAL_SHARDS, SMG_SHARD_LIMIT, SMG_HBM_LIMIT, SMG_FORCE_MULTI (smg_hetmers.h);
//   test hooks          test_hook(): force, on a table of any size, a code path that the engine otherwise picks from the table's
//                       size, k or counts (the 30-bit exchanged map, the one-bit map of k < 24, kl_probe_x, the unfiltered chain,
//                       pass 1's own directory, ..) -- tests/test_gpu_parity.py drives every one of them against the oracle;
//   tuning switches     tune_env(): variants that no table selects (A/B material) -- compiled in with -DSMG_TUNING only
//                       (tools/build_r06.sh), NULL in the product library.
static inline const char *test_hook(const char *name) { return getenv(name); }
#ifdef SMG_TUNING
static inline const char *tune_env(const char *name) { return getenv(name); }
#else
static inline const char *tune_env(const char *) { return NULL; }
#endif

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
  u64      plot_sum;           // total weight in the plot (kf_plot_sum)
  FastCtl  fast;               // control words of the fast path
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
//
// One workgroup decodes DEC_TILE consecutive entries, four per thread:
//   * their record bytes (DEC_TILE * pbyte, contiguous) are staged in LDS with coalesced dword loads
//     (records are 3..34 bytes wide and unaligned: per-thread byte loads ran at ~4 G entries/s);
//   * a k-mer is put together 32 bits at a time: an unaligned little-endian dword of the staged bytes is one
//     v_alignbyte over two LDS dwords, its big-endian value one v_perm; the first dword takes the prefix bytes from
//     the index, the last one is masked behind the k-mer's last byte (the count bytes follow it in the record).
//     (Round 2 walked the 8 W bytes of an entry one at a time, runtime shifts and all: 36-77 ms per 1e9 entries,
//      6 % of the HBM roofline, the slowest kernel of the product path.)
//   * the prefix of an entry is the smallest p with index[p] > i.  Thread 0 bisects the index for the first and the
//     last entry of the tile, the (few) index values in between are staged in LDS; a thread bisects there for its
//     first entry and steps on for the other three; a tile that spans more than DEC_IX buckets (sparse table)
//     bisects the global index per entry.
#define DEC_TILE  1024
#define DEC_EPT   4
#define DEC_IX    512
#define DEC_MAXPB 34                      // pbyte <= ceil(128/4) + 2 - 1

SMG_DEV int dec_bisect(const int64_t *__restrict__ index, int lo, int hi, int64_t i)
{ while (lo < hi)                         // smallest p in [lo, hi] with index[p] > i  (index[hi] > i)
    { const int m = (lo + hi) >> 1;
      if (index[m] <= i) lo = m + 1; else hi = m;
    }
  return lo;
}

// the little-endian dword at byte offset o of the staged bytes
SMG_DEV unsigned dec_u32(const unsigned *sraw, int o)
{ return __builtin_amdgcn_alignbyte(sraw[(o >> 2) + 1], sraw[o >> 2], (unsigned) (o & 3)); }

static inline size_t dec_lds_bytes(int pbyte) { return sizeof(unsigned) * (size_t) ((DEC_TILE * pbyte + 3) / 4 + 4); }

template <int W> __global__ void __launch_bounds__(TPB)
k_decode(const uint8_t *__restrict__ rec, const int64_t *__restrict__ index, int ixlen,
         int ibyte, int kbyte, int64_t n, int64_t ibase, u64 *__restrict__ keys,
         uint16_t *__restrict__ cnt)
{ // rec / keys / cnt are indexed by the LOCAL entry number 0..n-1; the prefix index by ibase + local
  // (a piece of a table that is decoded as it arrives, or a shard of a table that was cut for several GPUs,
  //  starts at entry ibase of the whole table)
  extern __shared__ unsigned sraw[];         // dec_lds_bytes(pbyte): the staged record bytes of a tile (+ slack)
  __shared__ int64_t  six[DEC_IX];
  __shared__ int      s_lo, s_hi;
  const int t = threadIdx.x;
  const int hbyte = kbyte - ibyte, pbyte = hbyte + 2;
  const int64_t i0 = (int64_t) blockIdx.x * DEC_TILE;
  const int64_t i1 = i0 + DEC_TILE < n ? i0 + DEC_TILE : n;
  // record bytes of [i0, i1): aligned dword loads around the (unaligned) byte range
  const int64_t b0 = i0 * pbyte, b1 = i1 * pbyte;
  const uintptr_t base = (uintptr_t) rec + (uintptr_t) b0;
  const uintptr_t abase = base & ~(uintptr_t) 3;
  const int skew = (int) (base - abase);
  const int ndw = (int) ((b1 - b0 + skew + 3) >> 2);
  const uintptr_t rbeg = (uintptr_t) rec, rend = (uintptr_t) rec + (uintptr_t) (n * pbyte);
  for (int w = t; w < ndw + 2; w += TPB)      // (+2: dec_u32 reads one dword past the last byte it needs)
    { const uintptr_t a = abase + 4 * (uintptr_t) w;
      unsigned v = 0;
      if (a >= rbeg && a + 4 <= rend) v = *reinterpret_cast<const unsigned *>(a);
      else                                  // first / last dword of the table: never touch bytes outside it
        for (int q = 0; q < 4; q++)
          if (a + q >= rbeg && a + q < rend) v |= (unsigned) *reinterpret_cast<const uint8_t *>(a + q) << (8 * q);
      sraw[w] = v;
    }
  if (t == 0)
    { s_lo = dec_bisect(index, 0, ixlen - 1, ibase + i0);
      s_hi = dec_bisect(index, 0, ixlen - 1, ibase + i1 - 1);
    }
  __syncthreads();
  const int lo = s_lo, hi = s_hi;
  const bool staged = hi - lo < DEC_IX;
  if (staged)
    for (int p = lo + t; p <= hi; p += TPB) six[p - lo] = index[p];
  __syncthreads();

  const int64_t e0 = i0 + (int64_t) DEC_EPT * t;
  if (e0 >= i1) return;
  int pa = 0;                                // position of the first entry's prefix in the staged slice
  if (staged)
    { int a = 0, b = hi - lo;
      while (a < b)
        { const int m = (a + b) >> 1;
          if (six[m] <= ibase + e0) a = m + 1; else b = m;
        }
      pa = a;
    }
  const int psh = 32 - 8 * ibyte;            // the prefix sits on top of the first dword
  u64 kw[DEC_EPT][W]; unsigned cw[DEC_EPT];
#pragma unroll
  for (int r = 0; r < DEC_EPT; r++)
    { const int64_t i = e0 + r;
      if (i >= i1) { for (int w = 0; w < W; w++) kw[r][w] = 0; cw[r] = 0; continue; }
      int p;
      if (staged) { while (six[pa] <= ibase + i) pa++; p = lo + pa; }      // (index[hi] > i1 - 1: the walk ends)
      else p = dec_bisect(index, lo, hi, ibase + i);
      const int o = (int) (i - i0) * pbyte + skew;
      unsigned D[2 * W];
#pragma unroll
      for (int j = 0; j < 2 * W; j++)
        { unsigned v = 0;
          if (4 * j < kbyte)
            { if (j == 0)
                { const unsigned sv = __builtin_bswap32(dec_u32(sraw, o));
                  v = ((unsigned) p << psh) | (sv >> (8 * ibyte));
                }
              else v = __builtin_bswap32(dec_u32(sraw, o + 4 * j - ibyte));
              const int over = 4 * j + 4 - kbyte;                         // bytes of this dword behind the k-mer
              if (over > 0) v &= ~0u << (8 * over);
            }
          D[j] = v;
        }
#pragma unroll
      for (int w = 0; w < W; w++) kw[r][w] = ((u64) D[2 * w] << 32) | D[2 * w + 1];
      cw[r] = dec_u32(sraw, o + hbyte) & 0xFFFFu;
    }
  const bool full = e0 + DEC_EPT <= i1 && (((uintptr_t) (keys + e0 * W)) & 15) == 0 && (((uintptr_t) (cnt + e0)) & 7) == 0;
  if (full)
    { if constexpr (W == 1)
        { ulonglong2 a, b; a.x = kw[0][0]; a.y = kw[1][0]; b.x = kw[2][0]; b.y = kw[3][0];
          *reinterpret_cast<ulonglong2 *>(keys + e0) = a;
          *reinterpret_cast<ulonglong2 *>(keys + e0 + 2) = b;
        }
      else
        {
#pragma unroll
          for (int r = 0; r < DEC_EPT; r++)
#pragma unroll
            for (int w = 0; w < W; w += 2)
              { if (w + 1 < W)
                  { ulonglong2 a; a.x = kw[r][w]; a.y = kw[r][w + 1];
                    *reinterpret_cast<ulonglong2 *>(keys + (e0 + r) * W + w) = a;
                  }
                else keys[(e0 + r) * W + w] = kw[r][w];
              }
        }
      *reinterpret_cast<ushort4 *>(cnt + e0) = make_ushort4((unsigned short) cw[0], (unsigned short) cw[1],
                                                            (unsigned short) cw[2], (unsigned short) cw[3]);
    }
  else
    for (int r = 0; r < DEC_EPT; r++)
      if (e0 + r < i1)
        { for (int w = 0; w < W; w++) keys[(e0 + r) * W + w] = kw[r][w];
          cnt[e0 + r] = (uint16_t) cw[r];
        }
}

// bucket directory + strict-order validation
template <int W> __global__ void __launch_bounds__(TPB)
k_directory(Tab t, uint32_t *__restrict__ bstart, Ctrl *__restrict__ ctrl)
{ const int64_t i = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i > t.n) return;
  if (i < t.n)
    { const uint32_t bcur = dir_bucket(t.dir, t.keys[i * W]);
      bool first = i == 0;
      if (i > 0)
        { first = dir_bucket(t.dir, t.keys[(i - 1) * W]) != bcur;
          if (!key_lt<W>(load_key<W>(t.keys, i - 1), load_key<W>(t.keys, i))) ctrl->unsorted = 1;
        }
      if (first) bstart[bcur] = (uint32_t) i;
    }
  else bstart[t.dir.nb] = (uint32_t) t.n;
}

// ---- the general path over a table that is cut into prefix shards ON ONE DEVICE (more than 2^32 entries) -----------
// The prefix-side partner of an entry (the k-mer with one of its first p0 bases replaced) may live in another shard:
// the look-up first picks the shard whose k-mer range holds it (<= 16 comparisons with the shards' first k-mers), then
// searches that shard's directory.  Window blocks never straddle a cut, so everything else stays shard local.
#define SET_MAX 16
struct TabSet
{ int ns;
  Tab shard[SET_MAX];
  u64 first[SET_MAX][4];       // first k-mer of shard s (all ones for an empty shard behind the last one)
};

template <int W> SMG_DEV int set_find(const TabSet *set, const Key<W> &y, int64_t &j)
{ int s = 0;
  for (int q = 1; q < set->ns; q++)
    { Key<W> f;
#pragma unroll
      for (int w = 0; w < W; w++) f.w[w] = set->first[q][w];
      if (!key_lt<W>(y, f)) s = q;
    }
  const Tab &t = set->shard[s];
  j = t.n > 0 ? find_key<W>(t.keys, t.dir, y) : -1;
  return s;
}

// ---- pass 1 -------------------------------------------------------------------------------
// One thread per entry.  SYM: scan positions >= p0 inside the window block, write
// deg = S_all, emit a request (rc(x), cnt, S_hi) when S_hi > 0 (or for every entry when
// emit_all), accumulate the fingerprints when want_fp.
// !SYM (general path): additionally resolve positions < p0 by directory look-ups; deg = all.

template <int W, bool SYM> __global__ void __launch_bounds__(TPB)
k_pass1(Tab t, int64_t lo, int64_t hi, int emit_all, int want_fp, u64 *__restrict__ req,
        int64_t req_cap, Ctrl *__restrict__ ctrl, const TabSet *__restrict__ set = NULL)
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
              { const Key<W> y = flip_base<W>(x, p, d);
                if (set)                      // (the partner may live in another shard of the same device)
                  { int64_t j;
                    const int sh = set_find<W>(set, y, j);
                    if (j >= 0 && c + (unsigned) set->shard[sh].cnt[j] <= SMG_SMAX) s_all++;
                  }
                else
                  { const int64_t j = find_key<W>(t.keys, t.dir, y);
                    if (j >= 0 && c + (unsigned) t.cnt[j] <= SMG_SMAX) s_all++;
                  }
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
          // (XOR, as on the fast path: the entries are strictly increasing, nothing can cancel; shards XOR their residues)
          f0 = wave_xor_u64(f0); f1 = wave_xor_u64(f1);
          f2 = wave_xor_u64(f2); f3 = wave_xor_u64(f3);
          if ((threadIdx.x & 63) == 0)
            { atomicXor((unsigned long long *) &ctrl->fp[0], (unsigned long long) f0); atomicXor((unsigned long long *) &ctrl->fp[1], (unsigned long long) f1);
              atomicXor((unsigned long long *) &ctrl->fp[2], (unsigned long long) f2); atomicXor((unsigned long long *) &ctrl->fp[3], (unsigned long long) f3);
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
  if (j < 0 || t.cnt[j] != c) { if (ctrl->missing == 0) ctrl->missing = 1; return; }
  if (v) deg_add(t.deg, j, v, t.g.wrap);
}

// exact symmetry proof for every entry of [lo,hi) against the local table (single GPU)
template <int W> __global__ void __launch_bounds__(TPB)
k_verify(Tab t, int64_t lo, int64_t hi, Ctrl *__restrict__ ctrl)
{ const int64_t i = lo + (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i >= hi) return;
  const Key<W> r = revcomp<W>(load_key<W>(t.keys, i), t.g.k);
  const int64_t j = find_key<W>(t.keys, t.dir, r);
  if (j < 0 || t.cnt[j] != t.cnt[i]) { if (ctrl->missing == 0) ctrl->missing = 1; }
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
k_pass2(Tab t, int64_t lo, int64_t hi, u64 *__restrict__ plot, const TabSet *__restrict__ set = NULL)
{ const int64_t i = lo + (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i >= hi) return;
  const Geo g = t.g;
  const unsigned di = t.deg[i];
  if (di > 1) return;
  if (di == 0 && !g.wrap) return;            // no wrap possible: degree 0 means no pair at all
  const Key<W> x = load_key<W>(t.keys, i);
  const unsigned c = t.cnt[i];

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
                  plot_add(plot, c, cj, wgt);
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
                    plot_add(plot, c, cj, wgt);
                  }
              }
          }
    }
  if (!SYM)
    { for (int p = 0; p < g.p0; p++)
        for (int d = 1; d <= 3; d++)
          { const Key<W> y = flip_base<W>(x, p, d);
            if (!key_lt<W>(x, y)) continue;
            if (set)
              { int64_t j;
                const int sh = set_find<W>(set, y, j);
                if (j >= 0)
                  { const unsigned cj = set->shard[sh].cnt[j];
                    if (c + cj <= SMG_SMAX && set->shard[sh].deg[j] <= 1) plot_add(plot, c, cj, 1);
                  }
              }
            else
              { const int64_t j = find_key<W>(t.keys, t.dir, y);
                if (j >= 0)
                  { const unsigned cj = t.cnt[j];
                    if (c + cj <= SMG_SMAX && t.deg[j] <= 1) plot_add(plot, c, cj, 1);
                  }
              }
          }
    }
}

// ---- extract on the counted path (k > 85): the pairs pass 2 counts, as records ---------------------
// The same walk as k_pass2<W, true>; a pair at a labelled pixel is written out like kf_extract writes it (the pair,
// and for p != k-1-p its mirror image at k-1-p).  One global atomic per pair: this path is exact, not fast.
template <int W> __global__ void __launch_bounds__(TPB)
k_extract(Tab t, const uint16_t *__restrict__ labels, u64 *__restrict__ out, u64 capacity, u64 *__restrict__ total)
{ const int64_t i = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i >= t.n) return;
  const Geo g = t.g;
  const unsigned di = t.deg[i];
  if (di > 1) return;
  if (di == 0 && !g.wrap) return;
  const Key<W> x = load_key<W>(t.keys, i);
  const unsigned c = t.cnt[i];
  const int k = g.k;

  auto emit = [&](int64_t j, const Key<W> &y, int p)
  { const unsigned cj = t.cnt[j];
    if (c + cj > SMG_SMAX || t.deg[j] > 1) return;
    const unsigned sum = c + cj, mn = c < cj ? c : cj;
    const unsigned lab = labels[(size_t) sum * SMG_PLOT_COLS + mn];
    if (!lab) return;
    const bool w2 = p != k - 1 - p;
    const unsigned bi = base_at<W>(x, p), bj = base_at<W>(y, p);              // bi < bj (i < j)
    const u64 q = atomicAdd(total, (u64) (w2 ? 2 : 1));
    if (q < capacity)
      { const bool pj = c < cj;                                                 // cnt[a] < cnt[b]: print b, alt base of a
        const Key<W> &who = pj ? y : x;
        u64 *o = out + q * (W + 1);
#pragma unroll
        for (int w = 0; w < W; w++) o[w] = who.w[w];
        o[W] = (u64) p | ((u64) (pj ? bi : bj) << 8) | ((u64) lab << 16);
      }
    if (w2 && q + 1 < capacity)
      { const bool pb = cj < c;                                                 // mirror image: a = rc(y), b = rc(x)
        const Key<W> who = revcomp<W>(pb ? x : y, k);
        u64 *o = out + (q + 1) * (W + 1);
#pragma unroll
        for (int w = 0; w < W; w++) o[w] = who.w[w];
        o[W] = (u64) (k - 1 - p) | ((u64) (pb ? 3u - bj : 3u - bi) << 8) | ((u64) lab << 16);
      }
  };

  bool big = false;
  if (i + WIN_LIM < t.n) big = same_block<W>(x, load_key<W>(t.keys, i + WIN_LIM), g);
  if (!big)
    { for (int64_t j = i + 1; j < t.n; j++)
        { const Key<W> y = load_key<W>(t.keys, j);
          if (!same_block<W>(x, y, g)) break;
          const int p = pair_pos<W>(x, y);
          if (p >= 0) emit(j, y, p);
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
            if (j < bhi && key_eq<W>(load_key<W>(t.keys, j), y)) emit(j, y, p);
          }
    }
}

// ------------------------------------------------------------------------------------------
//  Host side
// ------------------------------------------------------------------------------------------

#define FAST_MAX_K   85        // above this a uint8 degree can wrap: counted path (v1 kernels)
#define P1_GRID      1280      // persistent workgroups of the generic kf_pass1<W> (5 per CU resident: LDS bound)
#define P1_MAXGRID   4096      // upper bound of any pass-1 grid (partial fingerprint sums)
#define BF_MAXGRID   512       // workgroups of kf_bigfix (1024 threads each: two per CU)
#define P2_GRID      512       // persistent workgroups of kf_pass2 (2 per CU: 43.7 KB plot tile + 32 KB queue each)

struct smg_engine
{ int          device;
  hipStream_t  stream;
  int          kmer, W;
  int64_t      n;
  const u64   *keys;          // bound or owned
  const uint16_t *cnt;
  u64         *own_keys;      // owned copies (decode path)
  uint16_t    *own_cnt;
  uint8_t     *deg;    int64_t deg_cap;      // degree bytes (counted path) / code bytes (fast path)
  uint16_t    *sig;    int64_t sig_cap;       // k <= 32: look-up signatures (2 bytes per entry)
  uint32_t    *bstart; int64_t bstart_cap;
  uint32_t    *ixdir;  int64_t ixdir_cap;    // the table's own FastK prefix index (2^24 + 1 bucket starts, relative to this shard) as directory
  bool         have_ixdir;                   //   ... handed over with the current table (smg_engine_set_prefix_index); gone when the table changes
  bool         dir_preset;                   //   the current run looks up through ixdir: pass 1 writes no directory
  bool         have_ends;  u64 end_first, end_last;     // leading words of the first and the last entry of the bound table (read once)
  bool         no_filter;    // pass 1 builds no candidate map and nothing is filtered (out-of-core shards: smg_multi.hpp, host_run_sequential)
  // replay of the PHASE calls (sharded runs, smg_engine_set_replay): a step on a table whose last step went through pass1 ->
  // filter (hash proof, look-up chain) is queued without a read-back -- the counts the host needs are last step's (functions
  // of the table and of the exchanged maps), the device compares them with this step's and reports through smg_engine_proof
  bool         rp_want, rp_have, rp_active;    // asked for / a record exists / the current step runs from the record
  int64_t      rp_nreq, rp_nbig, rp_nf_req;    //   pass 1: requests, deferred entries; filter: requests kept
  unsigned     rp_nf_chunks, rp_grid;          //   filter: chunks of the kept list (the list the routing kernels walk); pass-1 grid (rows of `partials`)
  unsigned     rp_seen_chunks;                 //   chunks the plain step's filter filled (a replayed step walks that many + slack)
  u64         *rp_totals;  int rp_nranks;      //   router: per-destination totals of the recorded step (device, 16 words) and of this step (16 more)
  bool         rp_routed;                      //   ... this step's totals are there to be compared
  int          rp_bm_bits, rp_sym;             //   map geometry and proof the record belongs to
  u64         *req;    int64_t req_cap;      // bytes
  u64         *req2;   int64_t req2_cap;     // radix sort output
  void        *sort_tmp; int64_t sort_tmp_cap;
  u64         *dense;  int64_t dense_cap;    // compacted requests
  uint32_t    *chunk_off; int64_t chunk_off_cap;
  uint32_t    *skey[2]; int64_t skey_cap[2];   // index sort of wide records: leading k-mer bits (in, out)
  uint32_t    *sidx[2]; int64_t sidx_cap[2];   //                                  record numbers (in, out)
  uint32_t    *dbits;  int64_t dbits_cap;      // deferred entries of kf_pass1_d: one bit per table entry (bytes); all zero between runs
  bool         dbits_dirty;                     //   ... unless a run was abandoned between pass 1 and kf_bigfix
  uint32_t    *biglist; int64_t biglist_cap;    // the marked entries, compacted (kf_collect), bytes
  u64         *p1times; int64_t p1times_cap;    // SMG_P1_TIMES
  unsigned    *p1tick;  int64_t p1tick_cap;     // tile tickets of kf_pass1_d
  unsigned    *xtick;   int64_t xtick_cap;      // (bucket, part) tickets of kl_probe_x, one counter per XCD
  uint32_t    *farp;    int64_t farp_cap;       // beside it: the partner of a listed entry whose code is CODE_FAR (kf_bigfix -> kf_pass2_far)
  int          rw;                           // 64-bit words per request record
  uint32_t    *chunk_fill; int64_t chunk_cap; // bytes
  unsigned     max_chunks;
  uint32_t    *bmap;   int64_t bmap_cap;     // candidate block map (request filter)
  int          bm_bits;                      //   leading k-mer bits per block id (0 = not built by the last pass 1)
  bool         filtered;                     //   the request list of the last pass 1 has been filtered
  int          presorted;                    //   0 not decided, 1 req2 holds the list bucketed on its leading 8 bits, 2 left as it is
  u64         *reqf;   int64_t reqf_cap;     // filtered request chunks (swapped with req)
  uint32_t    *chunk_fillf; int64_t chunk_capf;
  uint32_t    *route_cnt;  int64_t route_cnt_cap;
  u64         *route_off;  int64_t route_off_cap;
  u64         *partials;   // [P1_MAXGRID][4]
  unsigned    *ghist;      // look-up chain: requests per bucket [L_BK], bucket cursor `bnext` behind it
  unsigned    *whist;  int64_t whist_cap;     //   requests per owner and bucket [owners][L_BK] (owner = a workgroup of pass 1 / kf_bigfix)
  unsigned     p1grid, nown; //  workgroups of the last pass 1; owners of the request list (+ BF_MAXGRID for kf_bigfix):
                           //   owner w fills the chunk slots w, w + nown, w + 2 nown, ..
  u64         *boff;       //   bucket offsets [L_BK + 1] and scatter cursors [L_BK] behind them
  LookupGeo    lg;         //   geometry of the current run (lg.nb = 0: the round-1 chain is used)
  bool         far_listed;   // pass 1 left the list of its deferred entries in biglist[0 .. st.nbig): pass 2 finishes the far partners from it
  bool         counted_done; // the counted path (k > 85) has run on a closed table: deg[] holds the wrapped degrees
  bool         lookup_pending; // look-ups of received requests were queued without a host wait (their time is read later)
  bool         use_sig;    // pass 1 writes the 2-byte look-up signatures (not worth their 5 GB when the filter leaves 1 request in 115)
  int          bm2;        //   the map is a two-bit map (smg_fast.hpp): 64-bit words, private to this engine
  int          bm_cap;     //   id bits of the block map: 32 on one GPU, 30 when the maps of several shards are exchanged
  int          bm_want;    //   ... as asked for by smg_engine_set_blockmap_bits (0: default)
  P1Cold      *p1cold;     // rarely used arguments of kf_pass1_d (device copy)
  P1Cold      *h_p1cold;   // pinned staging
  u64         *d_split;
  Ctrl        *ctrl;
  Ctrl        *h_ctrl;        // pinned mirror
  u64         *h_partials;    // pinned
  Geo          geo;
  Dir          dir;
  bool         prepared;      // pass 1 of the current table has run
  bool         fast;          // fast path in use
  unsigned     p1_grid[2][3]; // resident workgroups of kf_pass1_d<W, RW, ..> by [W-1][RW-1] (0 = not asked yet)
  unsigned     n_chunks;
  u64          fp[4];
  smg_stats    st;
  hipEvent_t   ev[13];        // 0,1 decode  2,3 pass 1  4,5 look-ups  6,7 pass 2  8,9 whole run  10 between partition and probe  11,12 filter of a replayed step
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

#define DISPATCH_W(e, CALL)                                                                   \
  switch ((e)->W) { case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break;  \
                    default: CALL(4); break; }
#define DISPATCH_W3(e, CALL)                                                                  \
  switch ((e)->W) { case 1: CALL(1); break; case 2: CALL(2); break; default: CALL(3); break; }

extern "C" const char *smg_version(void) { return "smudgeplot_amd 0.4 (hetmers engine, gfx950)"; }

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
  e->bm_cap = 30;
  if (hipMalloc(&e->ctrl, sizeof(Ctrl)) != hipSuccess
      || hipHostMalloc(&e->h_ctrl, sizeof(Ctrl)) != hipSuccess
      || hipMalloc(&e->partials, sizeof(u64) * 4 * (P1_MAXGRID + 1)) != hipSuccess      // (+ a row for the proof words)
      || hipHostMalloc(&e->h_partials, sizeof(u64) * 4 * P1_MAXGRID) != hipSuccess
      || hipMalloc(&e->d_split, sizeof(u64) * 16 * 4) != hipSuccess
      || hipMalloc(&e->p1cold, sizeof(P1Cold)) != hipSuccess
      || hipMalloc(&e->ghist, sizeof(unsigned) * (L_BK + 4)) != hipSuccess

      || hipMalloc(&e->boff, sizeof(u64) * (2 * L_BK + 4)) != hipSuccess
      || hipHostMalloc(&e->h_p1cold, sizeof(P1Cold)) != hipSuccess)
    { fail(errbuf, errlen, SMG_ENOMEM, "cannot allocate the control block%s");
      delete e; return NULL;
    }
  for (int i = 0; i < 13; i++) hipEventCreate(&e->ev[i]);
  return e;
}

extern "C" void smg_engine_destroy(smg_engine *e)
{ if (!e) return;
  hipSetDevice(e->device);
  hipStreamSynchronize(e->stream);
  hipFree(e->own_keys); hipFree(e->own_cnt); hipFree(e->deg); hipFree(e->sig); hipFree(e->bstart); hipFree(e->ixdir);
  hipFree(e->req); hipFree(e->req2); hipFree(e->sort_tmp); hipFree(e->dense); hipFree(e->chunk_off); hipFree(e->skey[0]); hipFree(e->skey[1]); hipFree(e->sidx[0]); hipFree(e->sidx[1]); hipFree(e->dbits); hipFree(e->biglist); hipFree(e->farp); hipFree(e->p1times); hipFree(e->p1tick); hipFree(e->xtick); hipFree(e->chunk_fill); hipFree(e->bmap); hipFree(e->reqf); hipFree(e->chunk_fillf); hipFree(e->route_cnt); hipFree(e->route_off);
  hipFree(e->partials); hipFree(e->ctrl); hipFree(e->d_split); hipFree(e->p1cold); hipFree(e->ghist); hipFree(e->boff);
  hipFree(e->whist); hipFree(e->rp_totals);
  hipHostFree(e->h_ctrl); hipHostFree(e->h_partials); hipHostFree(e->h_p1cold);
  for (int i = 0; i < 13; i++) hipEventDestroy(e->ev[i]);
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
  e->prepared = false; e->counted_done = false; e->lookup_pending = false;
  e->have_ixdir = false; e->dir_preset = false; e->have_ends = false;        // (properties of the table that was bound before)
  e->rp_have = false; e->rp_active = false;
  memset(&e->st, 0, sizeof(e->st));
  e->st.nels = nels;
  e->st.key_words = e->W;
  return SMG_OK;
}

static void set_geo(smg_engine *e)
{ Geo &g = e->geo;
  g.k = e->kmer;
  // first scanned position p0 = ceil((k-1)/2) = k/2 (integer division) for both parities: on
  // the symmetric path positions below it are the mirror images of the scanned ones, on the
  // general path they are resolved by directory look-ups
  g.p0 = e->kmer / 2;
  g.pw = g.p0 >> 5;
  const int r = g.p0 & 31;
  g.pmask = r ? ~0ull << (64 - 2 * r) : 0ull;
  g.mid = (e->kmer & 1) ? (e->kmer - 1) / 2 : -1;
  g.wrap = e->kmer > FAST_MAX_K;
}

template <typename T> static int grow(T **p, int64_t *cap, int64_t need_bytes, char *errbuf, size_t errlen)
{ if (*cap >= need_bytes && *p) return SMG_OK;
  if (*p) hipFree(*p);
  *p = NULL; *cap = 0;
  HIPCHK(hipMalloc((void **) p, (size_t) need_bytes));
  *cap = need_bytes;
  return SMG_OK;
}

extern "C" int smg_engine_bind(smg_engine *e, int kmer, int64_t nels, const uint64_t *d_keys,
                               const uint16_t *d_counts, char *errbuf, size_t errlen)
{ if (!e) return fail(errbuf, errlen, SMG_EINVAL, "null engine%s");
  if (nels > 0 && (!d_keys || !d_counts)) return fail(errbuf, errlen, SMG_EINVAL, "null table pointer%s");
  if (((uintptr_t) d_keys & 15) || ((uintptr_t) d_counts & 7))
    return fail(errbuf, errlen, SMG_EINVAL, "table pointers must be aligned (k-mers 16 bytes, counts 8 bytes)%s");
  HIPCHK(hipSetDevice(e->device));
  int rc = set_table(e, kmer, nels, errbuf, errlen);
  if (rc) return rc;
  e->keys = (const u64 *) d_keys;
  e->cnt = d_counts;
  return SMG_OK;
}

// the engine's own table for `nels` entries of a format-F table (allocated, not filled yet)
static int decode_begin(smg_engine *e, int kmer, int ibyte, int64_t nels, char *errbuf, size_t errlen)
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
  HIPCHK(hipMalloc(&e->own_cnt, sizeof(uint16_t) * (size_t) (nels > 0 ? nels : 1) + 16));
  e->keys = e->own_keys; e->cnt = e->own_cnt;
  return SMG_OK;
}

// decode `nent` records at d_records into the entries [first, first + nent) of the engine's table; the first of them is
// entry ibase + first of the whole table (whose prefix index is d_prefix_index).  Queued on `stream`.
static int decode_piece(smg_engine *e, hipStream_t stream, int ibyte, const uint8_t *d_records, const int64_t *d_prefix_index,
                        int64_t ibase, int64_t first, int64_t nent)
{ if (nent <= 0) return 0;
  const int kbyte = (e->kmer + 3) >> 2;
  const unsigned nblk = (unsigned) ((nent + DEC_TILE - 1) / DEC_TILE);
#define CALL(WW) hipLaunchKernelGGL(k_decode<WW>, dim3(nblk), dim3(TPB), dec_lds_bytes(kbyte + 2 - ibyte), stream, d_records, \
                         d_prefix_index, 1 << (8 * ibyte), ibyte, kbyte, nent, ibase + first, e->own_keys + first * e->W, e->own_cnt + first)
  DISPATCH_W(e, CALL)
#undef CALL
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

static int decode_at(smg_engine *e, int kmer, int ibyte, int64_t nels, int64_t ibase, const uint8_t *d_records,
                     const int64_t *d_prefix_index, char *errbuf, size_t errlen)
{ int rc = decode_begin(e, kmer, ibyte, nels, errbuf, errlen);
  if (rc) return rc;
  hipEventRecord(e->ev[0], e->stream);
  if (decode_piece(e, e->stream, ibyte, d_records, d_prefix_index, ibase, 0, nels))
    return fail(errbuf, errlen, SMG_ENODEV, "decode launch failed%s");
  hipEventRecord(e->ev[1], e->stream);
  HIPCHK(hipStreamSynchronize(e->stream));
  float ms = 0; hipEventElapsedTime(&ms, e->ev[0], e->ev[1]);
  e->st.ms_decode = ms;
  return SMG_OK;
}

// ingest hook: decode a piece behind its copy (smg_ingest.hpp)
struct DecodeHook { smg_engine *e; const int64_t *d_index; int ibyte; int64_t ibase; };
static int decode_hook(void *ctx, hipStream_t stream, const uint8_t *d_piece, int64_t first, int64_t nent)
{ DecodeHook *h = (DecodeHook *) ctx;
  return decode_piece(h->e, stream, h->ibyte, d_piece, h->d_index, h->ibase, first, nent);
}

extern "C" int smg_engine_decode(smg_engine *e, int kmer, int ibyte, int64_t nels,
                                 const uint8_t *d_records, const int64_t *d_prefix_index,
                                 char *errbuf, size_t errlen)
{ return decode_at(e, kmer, ibyte, nels, 0, d_records, d_prefix_index, errbuf, errlen); }

// ---- the table's own prefix index as look-up directory -------------------------------------------------------------
// A FastK table carries the number of entries up to every 3-byte prefix (libfastk.c:841, `index[p]` = entries whose first
// ibyte bytes are <= p).  With ibyte = 3 that IS a bucket directory over the leading 24 k-mer bits (bucket = hi32 >> 8;
// ~150 entries per bucket at 2.5e9 entries): ixdir[b] = first entry of bucket b relative to this shard, ixdir[2^24] = n.
// Pass 1 then writes no directory at all (round 3: a shift, a compare and a rare store per entry, and 130 MB cleared per
// run); the few million look-ups that survive the request filter bisect one or two steps longer.
#define IXDIR_BITS 24
__global__ void __launch_bounds__(TPB)
k_index_dir(const int64_t *__restrict__ index, int64_t ibase, int64_t n, uint32_t *__restrict__ out)
{ const int64_t b = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (b > (1ll << IXDIR_BITS)) return;
  int64_t v = (b ? index[b - 1] : 0) - ibase;
  if (b == (1ll << IXDIR_BITS)) v = n;
  out[b] = (uint32_t) (v < 0 ? 0 : v > n ? n : v);
}

extern "C" int smg_engine_set_prefix_index(smg_engine *e, const int64_t *d_prefix_index, int ibyte, int64_t first_entry,
                                           char *errbuf, size_t errlen)
{ if (!e) return fail(errbuf, errlen, SMG_EINVAL, "null engine%s");
  if (ibyte < 1 || ibyte > 3 || !d_prefix_index) return fail(errbuf, errlen, SMG_EINVAL, "prefix index: ibyte must be 1, 2 or 3%s");
  e->have_ixdir = false;
  if (ibyte != 3 || e->kmer < 12 || test_hook("SMG_NO_INDEX_DIR")) return SMG_OK;     // (a coarser index is of no use as a directory)
  HIPCHK(hipSetDevice(e->device));
  int rc = grow(&e->ixdir, &e->ixdir_cap, (int64_t) sizeof(uint32_t) * ((1ll << IXDIR_BITS) + 2), errbuf, errlen);
  if (rc) return rc;
  hipLaunchKernelGGL(k_index_dir, dim3((unsigned) (((1ll << IXDIR_BITS) + 1 + TPB - 1) / TPB)), dim3(TPB), 0, e->stream,
                     d_prefix_index, first_entry, e->n, e->ixdir);
  HIPCHK(hipGetLastError());
  e->have_ixdir = true;
  return SMG_OK;
}

static int read_ctrl(smg_engine *e, char *errbuf, size_t errlen)
{ HIPCHK(hipMemcpyAsync(e->h_ctrl, e->ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return SMG_OK;
}

// directory geometry over the first key word: `per` .. 2 `per` entries per bucket.  8 (one 128-byte line of k-mers) where
// every entry is looked up (exact proof, counted and general paths); 64 for the hash proof, whose filter leaves a few
// million look-ups: their bisection still stays inside one or two lines of signatures, while pass 1 writes -- and every
// run clears -- an eighth of the directory (1 GB -> 128 MB at 2.5e9 entries: -0.5 ms per run).
static int dir_geometry(smg_engine *e, int per, char *errbuf, size_t errlen, bool index_ok = false)
{ e->dir_preset = false;
  if (index_ok && e->have_ixdir && e->n > 0)
    { e->dir.bstart = e->ixdir; e->dir.b0 = 0; e->dir.dsh = 32 - IXDIR_BITS; e->dir.nb = 1u << IXDIR_BITS;
      e->dir_preset = true;
      return SMG_OK;
    }
  u64 first = 0, last = 0;
  if (e->n > 0 && !e->have_ends)      // (read once per bound table: a host round trip)
    { HIPCHK(hipMemcpyAsync(&e->end_first, e->keys, 8, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipMemcpyAsync(&e->end_last, e->keys + (size_t) (e->n - 1) * e->W, 8, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      e->have_ends = true;
    }
  if (e->n > 0) { first = e->end_first; last = e->end_last; }
  if (last < first) return fail(errbuf, errlen, SMG_EFORMAT, "table entries are not strictly increasing%s");
  int bits = 4;
  { const char *v = test_hook("SMG_DIR_PER"); if (v && atoi(v) >= 1) per = atoi(v); }       // tuning override
  while (bits < 30 && (1ll << (bits + 1)) <= e->n / per) bits++;
  const uint32_t hf = (uint32_t) (first >> 32), hl = (uint32_t) (last >> 32);
  int dsh = 0;
  while (dsh < 31 && ((hl >> dsh) - (hf >> dsh)) >= (1u << bits)) dsh++;
  e->dir.b0 = hf >> dsh;
  e->dir.dsh = dsh;
  e->dir.nb = (hl >> dsh) - (hf >> dsh) + 1;
  int rc = grow(&e->bstart, &e->bstart_cap, (int64_t) sizeof(uint32_t) * ((int64_t) e->dir.nb + 2), errbuf, errlen);
  if (rc) return rc;
  e->dir.bstart = e->bstart;
  return SMG_OK;
}

// ---- counted path (k > 85, and the general all-positions fallback at any k) ---------------------

static Tab make_tab(smg_engine *e)
{ Tab t;
  t.keys = e->keys; t.cnt = e->cnt; t.deg = e->deg; t.n = e->n; t.g = e->geo; t.dir = e->dir;
  return t;
}

static int counted_prepare(smg_engine *e, char *errbuf, size_t errlen)
{ HIPCHK(hipMemsetAsync(e->ctrl, 0, sizeof(Ctrl), e->stream));
  set_geo(e);
  e->fast = false; e->counted_done = false;
  const int64_t dbytes = ((e->n + 3) & ~3ll) + 4;
  int rc = grow(&e->deg, &e->deg_cap, dbytes, errbuf, errlen);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(e->deg, 0, (size_t) dbytes, e->stream));
  if ((rc = dir_geometry(e, 8, errbuf, errlen))) return rc;
  HIPCHK(hipMemsetAsync(e->bstart, 0xFF, sizeof(uint32_t) * ((size_t) e->dir.nb + 2), e->stream));
  Tab t = make_tab(e);
  const unsigned nblk = (unsigned) ((e->n + 1 + TPB - 1) / TPB);
#define CALL(WW) hipLaunchKernelGGL(k_directory<WW>, dim3(nblk), dim3(TPB), 0, e->stream, t, e->bstart, e->ctrl)
  DISPATCH_W(e, CALL)
#undef CALL
  HIPCHK(hipGetLastError());
  return SMG_OK;
}

static int plot_sum(smg_engine *e, int64_t *d_plot, char *errbuf, size_t errlen)
{ HIPCHK(hipMemsetAsync(&e->ctrl->plot_sum, 0, sizeof(u64), e->stream));
  hipLaunchKernelGGL(kf_plot_sum, dim3(64), dim3(1024), 0, e->stream, (const u64 *) d_plot, &e->ctrl->plot_sum);
  int rc = read_ctrl(e, errbuf, errlen);
  if (rc) return rc;
  e->st.npairs = (int64_t) e->h_ctrl->plot_sum;
  return SMG_OK;
}

// symmetric half-scan with counted degrees (exact uint8 wrap emulation), k > 85
static int counted_symmetric(smg_engine *e, int symcheck, int64_t *d_plot, bool *symmetric,
                             char *errbuf, size_t errlen)
{ int rc;
  if ((rc = counted_prepare(e, errbuf, errlen))) return rc;
  const unsigned nblk = (unsigned) ((e->n + TPB - 1) / TPB);
  int64_t cap = e->n / 4 + 1024;
  for (int attempt = 0; attempt < 2; attempt++)
    { if ((rc = grow(&e->req, &e->req_cap, cap * (int64_t) sizeof(u64) * (e->W + 1), errbuf, errlen))) return rc;
      Tab t = make_tab(e);
      hipEventRecord(e->ev[2], e->stream);
      if (e->n > 0)
        {
#define CALL(WW) hipLaunchKernelGGL((k_pass1<WW, true>), dim3(nblk), dim3(TPB), 0, e->stream, t, \
                   (int64_t) 0, e->n, 0, symcheck == SMG_SYM_HASH, e->req, cap, e->ctrl)
          DISPATCH_W(e, CALL)
#undef CALL
        }
      hipEventRecord(e->ev[3], e->stream);
      if ((rc = read_ctrl(e, errbuf, errlen))) return rc;
      if (e->h_ctrl->unsorted)
        return fail(errbuf, errlen, SMG_EFORMAT, "table entries are not strictly increasing%s");
      if ((int64_t) e->h_ctrl->nreq <= cap) break;
      cap = (int64_t) e->h_ctrl->nreq;
      HIPCHK(hipMemsetAsync(&e->ctrl->nreq, 0, sizeof(u64), e->stream));
      HIPCHK(hipMemsetAsync(e->ctrl->fp, 0, sizeof(u64) * 4, e->stream));
    }
  float ms = 0; hipEventElapsedTime(&ms, e->ev[2], e->ev[3]);
  e->st.ms_pass1 = ms;
  const int64_t nreq = (int64_t) e->h_ctrl->nreq;
  e->st.nrequests = nreq;
  hipEventRecord(e->ev[4], e->stream);
  { Tab t = make_tab(e);
    if (nreq > 0)
      { const unsigned rb = (unsigned) ((nreq + TPB - 1) / TPB);
#define CALL(WW) hipLaunchKernelGGL(k_apply<WW>, dim3(rb), dim3(TPB), 0, e->stream, t, e->req, nreq, e->ctrl)
        DISPATCH_W(e, CALL)
#undef CALL
      }
    if (symcheck == SMG_SYM_EXACT && e->n > 0)
      {
#define CALL(WW) hipLaunchKernelGGL(k_verify<WW>, dim3(nblk), dim3(TPB), 0, e->stream, t, (int64_t) 0, e->n, e->ctrl)
        DISPATCH_W(e, CALL)
#undef CALL
      }
  }
  hipEventRecord(e->ev[5], e->stream);
  if ((rc = read_ctrl(e, errbuf, errlen))) return rc;
  hipEventElapsedTime(&ms, e->ev[4], e->ev[5]);
  e->st.ms_rclookup = ms;
  *symmetric = e->h_ctrl->missing == 0;
  if (*symmetric && symcheck == SMG_SYM_HASH)
    *symmetric = e->h_ctrl->fp[0] == e->h_ctrl->fp[2] && e->h_ctrl->fp[1] == e->h_ctrl->fp[3];
  if (!*symmetric) return SMG_OK;
  HIPCHK(hipMemsetAsync(d_plot, 0, sizeof(int64_t) * SMG_PLOT_CELLS, e->stream));
  hipEventRecord(e->ev[6], e->stream);
  if (e->n > 0)
    { Tab t = make_tab(e);
#define CALL(WW) hipLaunchKernelGGL((k_pass2<WW, true>), dim3(nblk), dim3(TPB), 0, e->stream, t, \
                   (int64_t) 0, e->n, (u64 *) d_plot)
      DISPATCH_W(e, CALL)
#undef CALL
    }
  hipEventRecord(e->ev[7], e->stream);
  if ((rc = plot_sum(e, d_plot, errbuf, errlen))) return rc;
  hipEventElapsedTime(&ms, e->ev[6], e->ev[7]);
  e->st.ms_pass2 = ms;
  e->st.path = 1;
  e->counted_done = true;
  return SMG_OK;
}

// general (assumption-free) path: both passes over every position
static int run_general(smg_engine *e, int64_t *d_plot, char *errbuf, size_t errlen)
{ int rc;
  if ((rc = counted_prepare(e, errbuf, errlen))) return rc;
  const unsigned nblk = (unsigned) ((e->n + TPB - 1) / TPB);
  Tab t = make_tab(e);
  hipEventRecord(e->ev[2], e->stream);
  if (e->n > 0)
    {
#define CALL(WW) hipLaunchKernelGGL((k_pass1<WW, false>), dim3(nblk), dim3(TPB), 0, e->stream, t, \
                   (int64_t) 0, e->n, 0, 0, (u64 *) NULL, (int64_t) 0, e->ctrl)
      DISPATCH_W(e, CALL)
#undef CALL
    }
  hipEventRecord(e->ev[3], e->stream);
  HIPCHK(hipMemsetAsync(d_plot, 0, sizeof(int64_t) * SMG_PLOT_CELLS, e->stream));
  hipEventRecord(e->ev[6], e->stream);
  if (e->n > 0)
    {
#define CALL(WW) hipLaunchKernelGGL((k_pass2<WW, false>), dim3(nblk), dim3(TPB), 0, e->stream, t, \
                   (int64_t) 0, e->n, (u64 *) d_plot)
      DISPATCH_W(e, CALL)
#undef CALL
    }
  hipEventRecord(e->ev[7], e->stream);
  HIPCHK(hipGetLastError());
  if ((rc = plot_sum(e, d_plot, errbuf, errlen))) return rc;
  if (e->h_ctrl->unsorted)
    return fail(errbuf, errlen, SMG_EFORMAT, "table entries are not strictly increasing%s");
  float ms = 0;
  hipEventElapsedTime(&ms, e->ev[2], e->ev[3]); e->st.ms_pass1 += ms;
  hipEventElapsedTime(&ms, e->ev[6], e->ev[7]); e->st.ms_pass2 = ms;
  e->st.path = 2;
  return SMG_OK;
}

// the general path of ONE shard of a table whose shards all live on this device (smg_multi.hpp: virtual shards): step 1
// builds the directory and the degree array (then the caller collects the Tabs of all shards into a TabSet on the
// device), step 2 = pass 1 (degrees, over all positions), step 3 = pass 2 -- a barrier between the steps is the caller's.
static int general_shard_prepare(smg_engine *e, Tab *out, char *errbuf, size_t errlen)
{ int rc = counted_prepare(e, errbuf, errlen);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(e->stream));
  *out = make_tab(e);
  return SMG_OK;
}

static int general_shard_pass(smg_engine *e, const TabSet *d_set, int pass, int64_t *d_plot, char *errbuf, size_t errlen)
{ const unsigned nblk = (unsigned) ((e->n + TPB - 1) / TPB);
  Tab t = make_tab(e);
  if (pass == 1)
    { hipEventRecord(e->ev[2], e->stream);
      if (e->n > 0)
        {
#define CALL(WW) hipLaunchKernelGGL((k_pass1<WW, false>), dim3(nblk), dim3(TPB), 0, e->stream, t, \
                   (int64_t) 0, e->n, 0, 0, (u64 *) NULL, (int64_t) 0, e->ctrl, d_set)
          DISPATCH_W(e, CALL)
#undef CALL
        }
      hipEventRecord(e->ev[3], e->stream);
    }
  else
    { HIPCHK(hipMemsetAsync(d_plot, 0, sizeof(int64_t) * SMG_PLOT_CELLS, e->stream));
      hipEventRecord(e->ev[6], e->stream);
      if (e->n > 0)
        {
#define CALL(WW) hipLaunchKernelGGL((k_pass2<WW, false>), dim3(nblk), dim3(TPB), 0, e->stream, t, \
                   (int64_t) 0, e->n, (u64 *) d_plot, d_set)
          DISPATCH_W(e, CALL)
#undef CALL
        }
      hipEventRecord(e->ev[7], e->stream);
    }
  HIPCHK(hipGetLastError());
  int rc = read_ctrl(e, errbuf, errlen);
  if (rc) return rc;
  if (e->h_ctrl->unsorted) return fail(errbuf, errlen, SMG_EFORMAT, "table entries are not strictly increasing%s");
  float ms = 0;
  if (pass == 1) { hipEventElapsedTime(&ms, e->ev[2], e->ev[3]); e->st.ms_pass1 += ms; }
  else { hipEventElapsedTime(&ms, e->ev[6], e->ev[7]); e->st.ms_pass2 = ms; e->st.path = 2; }
  return SMG_OK;
}

// ---- fast path (k <= 85) --------------------------------------------------------------------------

static FastArgs make_fast(smg_engine *e)
{ FastArgs a;
  a.keys = e->keys; a.cnt = e->cnt; a.n = e->n; a.g = e->geo; a.dir = e->dir;
  a.code = e->deg;
  a.sig = (e->W <= 2 && e->use_sig) ? e->sig : NULL;
  a.sigsh = 16 + e->dir.dsh;               // the 16 bits right below the directory's bucket bits
  a.bmap = e->bm_bits ? e->bmap : NULL;
  a.bmsh = 32 - e->bm_bits;
  a.bm2 = e->bm2;
  return a;
}

static int bm_id_bits(int kmer, int cap);
// (k = 1 has no prefix bases to name a block by)
static bool filter_ok(const smg_engine *e)
{ return e->kmer >= 2 && ((e->W == 1 && e->rw == 1) || (e->W == 2 && e->rw != 1) || (e->W == 3 && e->rw == 4)); }

// replay: a step of the phase API on the table of the step before (smg_engine_set_replay).  Everything is launched as ever, but
// nothing is read back here: the lists are as large as last time, the counts that later launches take from the host (requests,
// deferred entries) are the recorded step's -- they are functions of the table -- and the device checks all of it against the
// control words at the end of the step (k_proof_replay).
static int fast_pass1(smg_engine *e, int emit_all, int with_meta, int want_fp, char *errbuf, size_t errlen, bool replay = false)
{ int rc;
  // record = the complement k-mer (W words) [+ one word: count | has-hi-pair << 16]
  // (two-word k-mers send key-only records for the hash proof as one-word ones do: 16 instead of 24 bytes for 21.6 % of
  //  the entries at k = 51; three-word k-mers go through the generic kernel, whose records always carry the count word)
  e->rw = e->W + ((with_meta || e->W > 2) ? 1 : 0);
  HIPCHK(hipMemsetAsync(e->ctrl, 0, sizeof(Ctrl), e->stream));
  set_geo(e);
  e->fast = true; e->counted_done = false;
  if ((rc = grow(&e->deg, &e->deg_cap, ((e->n + 15) & ~15ll) + 32, errbuf, errlen))) return rc;
  const int64_t pbytes = ((e->n + 15) & ~15ll) + 32;
  e->use_sig = e->W <= 2;
  // (hash proof of one- and two-word k-mers: the table's own prefix index, when it came with one, is the directory)
  if ((rc = dir_geometry(e, (emit_all || e->W > 1) ? 8 : 64, errbuf, errlen, !emit_all && e->W <= 2))) return rc;
  // request filter: the candidate block map (an empty shard has one too -- all zero -- so that every rank of a
  // sharded run reports the same geometry and takes part in the exchange of the maps)
  e->bm_bits = 0; e->bm2 = 0;
  e->filtered = false; e->presorted = 0;
  e->lg.nb = 0;
  if (filter_ok(e) && !emit_all && !e->no_filter && !test_hook("SMG_NO_FILTER"))
    { const int nbits = bm_id_bits(e->kmer, e->bm_cap);
      // the look-up chain of smg_lookup.hpp: one-word k-mers, key-only records, a map of >= 12 id bits
      const bool chain = nbits >= 12 && e->W <= 2 && e->rw == e->W && !tune_env("SMG_OLD_LOOKUP");
      // ... with the two-bit map (smg_fast.hpp) when the k-mer has bits below the id to hash (a function of k and the
      // environment alone: every shard of a table decides the same)
      e->bm2 = (chain && e->kmer >= 24 && !test_hook("SMG_ONE_BIT_MAP")) ? 1 : 0;
      const int64_t bytes = (4 * (((1ll << nbits) + 31) >> 5) + 4 * D_BMW + 64) << e->bm2;
      if ((rc = grow(&e->bmap, &e->bmap_cap, bytes, errbuf, errlen))) return rc;
      HIPCHK(hipMemsetAsync(e->bmap, 0, (size_t) bytes, e->stream));
      e->bm_bits = nbits;
      if (chain) e->lg = lookup_geo(nbits);
      // Signatures (2 bytes per entry written by pass 1, so that a look-up bisects 2-byte instead of 8-byte words)
      // pay when most requests are looked up.  With the 32-bit two-bit map of a single-GPU run 1 request in 115
      // survives the filter: 3.9e6 look-ups at 2.5e9 entries, which can afford the k-mer lines of their bucket, while
      // the signatures cost pass 1 5 GB of stores and four instructions per entry.  SMG_SIG=0/1 overrides.
      if (chain && e->bm2 && nbits >= 32) e->use_sig = false;
    }
  { const char *v = test_hook("SMG_SIG"); if (v && e->W <= 2) e->use_sig = atoi(v) != 0; }
  if (e->use_sig && (rc = grow(&e->sig, &e->sig_cap, 2 * pbytes, errbuf, errlen))) return rc;
  if (e->n > 0 && !e->dir_preset)
    HIPCHK(hipMemsetAsync(e->bstart, 0xFF, sizeof(uint32_t) * ((size_t) e->dir.nb + 2), e->stream));
  if (e->n == 0)
    { HIPCHK(hipMemsetAsync(e->bstart, 0, sizeof(uint32_t) * ((size_t) e->dir.nb + 2), e->stream));
      e->n_chunks = 0; e->prepared = true;
      memset(e->fp, 0, sizeof(e->fp));
      return SMG_OK;
    }
  const bool narrow = e->W <= 2;                       // k <= 64: kf_pass1_d (blocked register scan)
  const int64_t ntiles = narrow ? (e->n + D_OWN - 1) / D_OWN : (e->n + F_TILE - 1) / F_TILE;
  GeoR gr;
  { const int p0 = e->kmer / 2, sbits = 2 * (e->kmer - p0);
    gr.k = e->kmer;
    if (e->W == 1)
      { gr.kshift = 64 - 2 * e->kmer;
        gr.pshift = 32 - 2 * p0;
      }
    else
      { gr.kshift = 128 - 2 * e->kmer;                 // 33 <= k <= 64: 0 .. 62
        gr.pshift = 64 - 2 * p0;                        // p0 = 16 .. 32
      }
    gr.smask = sbits >= 64 ? ~0ull : ((1ull << sbits) - 1ull);
    gr.mshift = sbits - 2;
  }
  const bool odd = (e->kmer & 1) != 0;
  unsigned grid = P1_GRID;
  if (narrow)
    { // persistent workgroups: exactly what is resident (a static tile stride must not have stragglers)
      unsigned &cached = e->p1_grid[e->W - 1][e->rw - 1];
      if (!cached)
        { int nb = 0, cus = 0;
          hipError_t he;
          // (the ODD / KF variants of one (W, RW) class use the same registers and LDS: ask for one of them)
          if (e->W == 1 && e->rw == 1) he = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kf_pass1_d<1, 1, true, true>, D_TPB, 0);
          else if (e->W == 1)          he = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kf_pass1_d<1, 2, true, true>, D_TPB, 0);
          else if (e->rw == 2)         he = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kf_pass1_d<2, 2, true, false>, D_TPB, 0);
          else                         he = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kf_pass1_d<2, 3, true, false>, D_TPB, 0);
          if (he != hipSuccess || nb < 1) nb = 3;
          if (nb > 8) nb = 8;
          if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess || cus < 1) cus = 256;
          { const char *g = tune_env("SMG_P1_WGS_PER_CU"); if (g && atoi(g) > 0) nb = atoi(g); }
          cached = (unsigned) (nb * cus);
          if (tune_env("SMG_DEBUG")) fprintf(stderr, "  [smg] kf_pass1_d<%d,%d>: %d workgroups per CU x %d CUs\n", e->W, e->rw, nb, cus);
        }
      grid = cached;
    }
  if (grid > P1_MAXGRID) grid = P1_MAXGRID;
  if ((int64_t) grid > ntiles) grid = (unsigned) ntiles;
  // a chunk is closed as soon as the next tile's batch (<= one tile of records) does not fit, so
  // chunks fill to >= 75 %: size the list for that, and never below what is already allocated
  // (an engine that is reused on the same table must not redo pass 1 every time)
  int64_t want_rec = (emit_all ? e->n + e->n / 24 : e->n / 4) + (int64_t) (grid + 16 + 256) * F_CH;
  const int64_t dwords = ((((e->n + 31) >> 5) + 2) + 3) & ~3ll;          // (kf_bigfix reads the map four words at a time)
  if (narrow)
    { if (e->dbits_cap < dwords * 4) e->dbits_dirty = true;                 // (fresh memory)
      if ((rc = grow(&e->dbits, &e->dbits_cap, dwords * 4, errbuf, errlen))) return rc;
    }
  int64_t want_big = e->n / 16 + 4096;      // deferred entries: 0.13 % of the diploid table, 0.4 % with 5 % repeats
  bool done = false;
  for (int attempt = 0; attempt < 5 && !done; attempt++)
    { unsigned maxc = (unsigned) ((want_rec + F_CH - 1) / F_CH);
      if (e->lg.nb)
        { // look-up chain: owner w fills the slots w, w + G, w + 2G, .. (G = workgroups of this launch + those of kf_bigfix),
          // so the list must hold G times the chunks of the busiest owner -- pass 1 deals its tiles round-robin, the owners
          // are balanced to a few per cent -- and the slots of the kf_bigfix owners stay empty on a table without long blocks
          const int64_t G = (int64_t) grid + BF_MAXGRID;
          const int64_t per = ((int64_t) maxc + grid - 1) / grid + 2;
          if (per * G > (int64_t) maxc && per * G < 0x7FFFFFFFll) maxc = (unsigned) (per * G);
        }
      { const int64_t have = e->req_cap / ((int64_t) F_CH * (int64_t) sizeof(u64) * e->rw);
        if (e->req && have > (int64_t) maxc && have < 0x7FFFFFFFll) maxc = (unsigned) have;
      }
      if ((rc = grow(&e->req, &e->req_cap, (int64_t) maxc * F_CH * (int64_t) sizeof(u64) * e->rw, errbuf, errlen))) return rc;
      if ((rc = grow(&e->chunk_fill, &e->chunk_cap, (int64_t) maxc * 4 + 4, errbuf, errlen))) return rc;
      e->max_chunks = maxc;
      if (narrow && e->dbits_dirty)
        { HIPCHK(hipMemsetAsync(e->dbits, 0, (size_t) e->dbits_cap, e->stream)); e->dbits_dirty = false; }
      FastArgs a = make_fast(e);
      if (e->lg.nb)
        { // owners of the request chunks: the workgroups of this launch, then those of kf_bigfix (rows zero unless it runs)
          if ((rc = grow((char **) &e->whist, &e->whist_cap, (int64_t) (grid + BF_MAXGRID) * L_BK * 4, errbuf, errlen))) return rc;
          HIPCHK(hipMemsetAsync(e->whist + (size_t) grid * L_BK, 0, sizeof(unsigned) * BF_MAXGRID * L_BK, e->stream));
          HIPCHK(hipMemsetAsync(e->chunk_fill, 0, (size_t) maxc * 4, e->stream));      // (a slot that stays empty has fill 0)
          e->p1grid = grid; e->nown = grid + BF_MAXGRID;
        }
      hipEventRecord(e->ev[2], e->stream);
      if (narrow)
        {
          P1Hot hot;
          hot.keys = a.keys; hot.cnt = a.cnt; hot.n = a.n; hot.code = a.code; hot.sig = a.sig; hot.bstart = e->dir_preset ? (uint32_t *) NULL : e->bstart;
          hot.bmap = a.bmap; hot.b0 = e->dir.b0; hot.nb = e->dir.nb; hot.shifts = (unsigned) e->dir.dsh | ((unsigned) a.sigsh << 6) | (((unsigned) a.bmsh & 31u) << 12)
                       | ((emit_all ? 1u : 0u) << 18) | ((want_fp ? 1u : 0u) << 19) | ((unsigned) e->lg.nb << 20) | ((unsigned) e->bm2 << 24);
          hot.G = gr; hot.ntiles = ntiles;
          e->h_p1cold->req = e->req; e->h_p1cold->chunk_fill = e->chunk_fill; e->h_p1cold->dbits = e->dbits;
          e->dbits_dirty = true;                                               // until kf_bigfix has cleared the bits again
          e->h_p1cold->partials = e->partials; e->h_p1cold->ctl = &e->ctrl->fast; e->h_p1cold->max_chunks = maxc;
          e->h_p1cold->whist = e->whist; e->h_p1cold->owners = grid + BF_MAXGRID;
          e->h_p1cold->times = NULL;
          if ((rc = grow(&e->p1tick, &e->p1tick_cap, (int64_t) D_NCLS * D_TICKW * 4, errbuf, errlen))) return rc;
          HIPCHK(hipMemsetAsync(e->p1tick, 0, (size_t) D_NCLS * D_TICKW * 4, e->stream));
          e->h_p1cold->tick = e->p1tick;
          if (tune_env("SMG_P1_TIMES"))                                          // tuning: when did every workgroup start and end?
            { if ((rc = grow(&e->p1times, &e->p1times_cap, (int64_t) grid * 24, errbuf, errlen))) return rc;
              HIPCHK(hipMemsetAsync(e->p1times, 0, (size_t) grid * 24, e->stream));
              e->h_p1cold->times = e->p1times;
            }
          HIPCHK(hipMemcpyAsync(e->p1cold, e->h_p1cold, sizeof(P1Cold), hipMemcpyHostToDevice, e->stream));
#define LAUNCH_R(W_, RW_, ODD_, KF_) do { bool hot_form = false; \
          if constexpr ((RW_) == (W_))            /* (the hot form exists for key-only records: no dead instantiations) */ \
            if (hot.bstart == NULL && hot.sig == NULL && e->bm2 && hot.bmap != NULL && want_fp && !emit_all && e->lg.nb) \
              { hot_form = true; \
                hipLaunchKernelGGL((kf_pass1_d<W_, W_, ODD_, KF_, 2>), dim3(grid), dim3(D_TPB), 0, e->stream, hot, (const P1Cold *) e->p1cold); } \
          if (!hot_form) hipLaunchKernelGGL((kf_pass1_d<W_, RW_, ODD_, KF_, 1>), dim3(grid), dim3(D_TPB), 0, e->stream, hot, (const P1Cold *) e->p1cold); } while (0)
#define LAUNCH_R2(RW_, ODD_) { if (kf) LAUNCH_R(1, RW_, ODD_, true); else LAUNCH_R(1, RW_, ODD_, false); }
          const bool kf = gr.pshift < 32 && gr.kshift < 32;          // 17 <= k <= 32
          if (e->W == 2 && e->rw == 2) { if (odd) LAUNCH_R(2, 2, true, false); else LAUNCH_R(2, 2, false, false); }
          else if (e->W == 2)  { if (odd) LAUNCH_R(2, 3, true, false); else LAUNCH_R(2, 3, false, false); }
          else if (e->rw == 1) { if (odd) LAUNCH_R2(1, true) else LAUNCH_R2(1, false) }
          else                 { if (odd) LAUNCH_R2(2, true) else LAUNCH_R2(2, false) }
#undef LAUNCH_R2
#undef LAUNCH_R
        }
      else
        {
#define CALL(WW) hipLaunchKernelGGL(kf_pass1<WW>, dim3(grid), dim3(F_TPB), 0, e->stream, a, e->bstart, \
                   e->req, e->chunk_fill, maxc, emit_all, want_fp, e->partials, &e->ctrl->fast, ntiles)
          DISPATCH_W3(e, CALL)
#undef CALL
        }
      hipEventRecord(e->ev[3], e->stream);
      HIPCHK(hipGetLastError());
      if (want_fp)
        HIPCHK(hipMemcpyAsync(e->h_partials, e->partials, sizeof(u64) * 4 * grid, hipMemcpyDeviceToHost, e->stream));
      unsigned bigcap = 0;
      if (narrow)
        { // exact redo of the deferred entries (a pair at distance 4..30, or a window block longer than the window):
          // kf_collect compacts the bit map pass 1 marked them in into a list (and clears it), kf_bigfix redoes the list.
          // Launched without waiting for pass 1's control words (a request list that overflowed is guarded on the
          // device; the run is redone below either way, and so it is when the list of deferred entries was too short).
          if (want_big < e->biglist_cap / 4) want_big = e->biglist_cap / 4;
          if (want_big > 0xFFFFFFF0ll) want_big = 0xFFFFFFF0ll;
          if ((rc = grow(&e->biglist, &e->biglist_cap, want_big * 4, errbuf, errlen))) return rc;
          if ((rc = grow(&e->farp, &e->farp_cap, e->biglist_cap, errbuf, errlen))) return rc;
          bigcap = (unsigned) (e->biglist_cap / 4 > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : e->biglist_cap / 4);
          unsigned cb = (unsigned) ((dwords / 4 + BF_TPB - 1) / BF_TPB);
          if (cb > BF_MAXGRID) cb = BF_MAXGRID;
          hipEventRecord(e->ev[0], e->stream);
          hipLaunchKernelGGL(kf_collect, dim3(cb), dim3(BF_TPB), 0, e->stream, e->dbits, dwords, e->biglist, bigcap, &e->ctrl->fast.nbig);
#define BIGFIX(W_, RW_) hipLaunchKernelGGL((kf_bigfix<W_, RW_>), dim3(BF_MAXGRID), dim3(BF_TPB), 0, e->stream, a, e->biglist, &e->ctrl->fast.nbig, bigcap, e->req, \
                              e->chunk_fill, maxc, &e->ctrl->fast, e->lg.nb ? e->whist + (size_t) grid * L_BK : (unsigned *) NULL, \
                              grid, grid + BF_MAXGRID, e->lg.nb, e->farp)
          if (e->W == 2 && e->rw == 2) BIGFIX(2, 2); else if (e->W == 2) BIGFIX(2, 3); else if (e->rw == 1) BIGFIX(1, 1); else BIGFIX(1, 2);
#undef BIGFIX
          hipEventRecord(e->ev[3], e->stream);
          HIPCHK(hipGetLastError());
        }
      if (replay) { done = true; break; }
      if ((rc = read_ctrl(e, errbuf, errlen))) return rc;
      if (narrow && e->h_p1cold->times)
        { std::vector<u64> tm((size_t) grid * 3);
          HIPCHK(hipMemcpy(tm.data(), e->p1times, (size_t) grid * 24, hipMemcpyDeviceToHost));
          u64 t0 = ~0ull, t1 = 0;
          for (unsigned b = 0; b < grid; b++) { if (tm[3 * b] < t0) t0 = tm[3 * b]; if (tm[3 * b + 1] > t1) t1 = tm[3 * b + 1]; }
          // per XCD (bits of HW_ID differ between generations: printed raw as well): latest start, earliest / mean / latest end
          double sum = 0; u64 smax = 0, emin = ~0ull;
          for (unsigned b = 0; b < grid; b++)
            { sum += (double) (tm[3 * b + 1] - t0); if (tm[3 * b] - t0 > smax) smax = tm[3 * b] - t0; if (tm[3 * b + 1] - t0 < emin) emin = tm[3 * b + 1] - t0; }
          fprintf(stderr, "  [smg] pass-1 workgroups (100 MHz ticks = 10 ns): span %llu, latest start %llu, earliest end %llu, mean end %.0f\n",
                  (unsigned long long) (t1 - t0), (unsigned long long) smax, (unsigned long long) emin, sum / grid);
          unsigned hist[20]; memset(hist, 0, sizeof(hist));
          for (unsigned b = 0; b < grid; b++) { unsigned q = (unsigned) ((tm[3 * b + 1] - t0) * 20 / (t1 - t0 + 1)); hist[q < 20 ? q : 19]++; }
          fprintf(stderr, "  [smg] ends per 5 %% of the span:");
          for (int q = 0; q < 20; q++) fprintf(stderr, " %u", hist[q]);
          fprintf(stderr, "\n  [smg] first 16 workgroups: start end hw_id:");
          for (unsigned b = 0; b < 16 && b < grid; b++) fprintf(stderr, " (%llu %llu %llx)", (unsigned long long) (tm[3 * b] - t0), (unsigned long long) (tm[3 * b + 1] - t0), (unsigned long long) tm[3 * b + 2]);
          fprintf(stderr, "\n");
        }
      e->dbits_dirty = false;
      if (e->h_ctrl->fast.unsorted)
        return fail(errbuf, errlen, SMG_EFORMAT, "table entries are not strictly increasing%s");
      if (narrow && e->h_ctrl->fast.nbig > bigcap)
        { // more deferred entries than the list holds (a repeat-dominated table): size it from the count and redo
          want_big = (int64_t) e->h_ctrl->fast.nbig + (int64_t) e->h_ctrl->fast.nbig / 8 + 4096;
          e->dbits_dirty = true;                  // (the bits of the entries that did not fit are still set)
          HIPCHK(hipMemsetAsync(&e->ctrl->fast, 0, sizeof(FastCtl), e->stream));
          continue;
        }
      if (e->h_ctrl->fast.n_chunks > maxc)
        { // the request list outgrew its first-guess capacity: size it from the count and redo
          want_rec = (int64_t) (e->h_ctrl->fast.n_chunks + 16 + 256) * F_CH;
          HIPCHK(hipMemsetAsync(&e->ctrl->fast, 0, sizeof(FastCtl), e->stream));
          continue;
        }
      done = true;
    }
  // (each redo sizes the lists from the counts the failed attempt reported, so the second attempt fits; a list that
  //  still overflows after that is a bug, and must not be read as a complete request list)
  if (replay)
    { e->rp_grid = grid;
      e->n_chunks = 1;                      // (> 0: "there are requests"; the real number is checked at the end of the step)
      e->st.nrequests = e->st.nemitted = e->rp_nreq;
      e->st.nbig = e->rp_nbig;
      e->st.ms_filter = 0; e->st.ms_bigfix = 0;
      e->far_listed = narrow;
      e->prepared = true;
      return SMG_OK;
    }
  if (!done || e->h_ctrl->fast.n_chunks > e->max_chunks)
    return fail(errbuf, errlen, SMG_ENODEV, "pass 1: the request list overflowed its capacity on every attempt%s");
  e->n_chunks = e->h_ctrl->fast.n_chunks;
  memset(e->fp, 0, sizeof(e->fp));
  if (want_fp)
    for (unsigned b = 0; b < grid; b++)
      for (int q = 0; q < 4; q++) e->fp[q] ^= e->h_partials[b * 4 + q];     // (XOR fingerprint: smg_device.hpp)
  float ms = 0; hipEventElapsedTime(&ms, e->ev[2], e->ev[3]);
  e->st.ms_pass1 = ms;
  e->st.nrequests = (int64_t) e->h_ctrl->fast.nreq;
  e->st.nemitted = e->st.nrequests;
  e->st.ms_filter = 0;
  e->st.nbig = narrow ? (int64_t) e->h_ctrl->fast.nbig : 0;
  e->far_listed = narrow;
  e->st.ms_bigfix = 0;
  if (narrow) { float mb = 0; hipEventElapsedTime(&mb, e->ev[0], e->ev[3]); e->st.ms_bigfix = mb; }
  e->prepared = true;
  return SMG_OK;
}

// key-only records of one-word k-mers: radix sort on the leading 32 bits, then look up in order.
// rocPRIM's mid-size (merge sort) variant returned garbage with a partial bit range on gfx950 /
// ROCm 7.2, so Onesweep is forced (MergeSortLimit = 0) and small batches are not sorted at all
// (their look-ups are too few to matter).  SMG_VERIFY_SORT=1 checks every sort (order + checksums).
#define SORT_MIN 4096
typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                   rocprim::default_config, 0> smg_sort_config;

// sort nsort keys and look them up in that order; holes != 0: the list contains sentinel words (kf_fill_holes)
static int apply_sorted(smg_engine *e, const u64 *keys_in, int64_t nsort, int holes, char *errbuf, size_t errlen)
{ int rc;
  if (nsort <= 0) return SMG_OK;
  FastArgs a = make_fast(e);
  int64_t nb = (nsort + F_TPB - 1) / F_TPB;
  if (nb > 16384) nb = 16384;
  const u64 *src = keys_in;
  // (a short list -- what one rank of eight receives -- is looked up as it comes: the sort is eight launches, ~90 us, to put
  //  a few hundred thousand look-ups in order, which take 70 us either way)
  int64_t sort_min = 1 << 21;
  { const char *v = tune_env("SMG_APPLY_SORT_MIN"); if (v && atoll(v) >= SORT_MIN) sort_min = atoll(v); }
  if (nsort >= sort_min)
    { if ((rc = grow(&e->req2, &e->req2_cap, nsort * (int64_t) sizeof(u64), errbuf, errlen))) return rc;
      size_t tmp = 0;
      unsigned lobit = 40;                   // tuning knob: sort on bits [lobit, 64) of the k-mer
      { const char *lb = tune_env("SMG_SORT_LOBIT"); if (lb) lobit = (unsigned) atoi(lb); if (lobit > 56) lobit = 56; }
      HIPCHK(rocprim::radix_sort_keys<smg_sort_config>(nullptr, tmp, (u64 *) keys_in, e->req2, (size_t) nsort,
                                                       lobit, 64u, e->stream));
      if ((rc = grow((char **) &e->sort_tmp, &e->sort_tmp_cap, (int64_t) tmp + 16, errbuf, errlen))) return rc;
      HIPCHK(rocprim::radix_sort_keys<smg_sort_config>(e->sort_tmp, tmp, (u64 *) keys_in, e->req2, (size_t) nsort,
                                                       lobit, 64u, e->stream));
      src = e->req2;
      if (test_hook("SMG_VERIFY_SORT"))
        { u64 *d_chk = NULL, h[4];
          HIPCHK(hipMalloc(&d_chk, 32));
          HIPCHK(hipMemsetAsync(d_chk, 0, 32, e->stream));
          hipLaunchKernelGGL(kf_check_sorted, dim3((unsigned) nb), dim3(F_TPB), 0, e->stream, keys_in, e->req2, nsort, (int) lobit, d_chk);
          HIPCHK(hipMemcpyAsync(h, d_chk, 32, hipMemcpyDeviceToHost, e->stream));
          HIPCHK(hipStreamSynchronize(e->stream));
          hipFree(d_chk);
          if (h[0] || h[1] != h[2] || h[3])
            return fail(errbuf, errlen, SMG_ENODEV, "request sort self-check failed (rocPRIM radix sort)%s");
        }
    }
  hipLaunchKernelGGL(kf_apply_sorted<1>, dim3((unsigned) nb), dim3(F_TPB), 0, e->stream, a, src, nsort, holes, &e->ctrl->fast);
  return SMG_OK;
}

// records with a count/flag word: sort (leading 32 k-mer bits, record number) pairs on the upper 24 bits, look up in
// that order (small batches are looked up as they come)
static int apply_indexed(smg_engine *e, const u64 *rec, int64_t n, int check_count, char *errbuf, size_t errlen)
{ int rc;
  if (n <= 0) return SMG_OK;
  FastArgs a = make_fast(e);
  int64_t nb = (n + F_TPB - 1) / F_TPB;
  if (nb > 16384) nb = 16384;
  if (n < SORT_MIN || n >= 0xFFFFFFF0ll)
    {
#define CALL(WW) hipLaunchKernelGGL(kf_apply<WW>, dim3((unsigned) (nb > 8192 ? 8192 : nb)), dim3(F_TPB), 0, e->stream, a, rec, \
                   (const uint32_t *) NULL, n, check_count, &e->ctrl->fast, e->rw)
      DISPATCH_W3(e, CALL)
#undef CALL
      return SMG_OK;
    }
  for (int q = 0; q < 2; q++)
    { if ((rc = grow(&e->skey[q], &e->skey_cap[q], n * 4 + 16, errbuf, errlen))) return rc;
      if ((rc = grow(&e->sidx[q], &e->sidx_cap[q], n * 4 + 16, errbuf, errlen))) return rc;
    }
  hipLaunchKernelGGL(kf_sortkey, dim3((unsigned) nb), dim3(F_TPB), 0, e->stream, rec, e->rw, n, e->skey[0], e->sidx[0]);
  size_t tmp = 0;
  HIPCHK(rocprim::radix_sort_pairs<smg_sort_config>(nullptr, tmp, e->skey[0], e->skey[1], e->sidx[0], e->sidx[1], (size_t) n,
                                                    8u, 32u, e->stream));
  if ((rc = grow((char **) &e->sort_tmp, &e->sort_tmp_cap, (int64_t) tmp + 16, errbuf, errlen))) return rc;
  HIPCHK(rocprim::radix_sort_pairs<smg_sort_config>(e->sort_tmp, tmp, e->skey[0], e->skey[1], e->sidx[0], e->sidx[1], (size_t) n,
                                                    8u, 32u, e->stream));
#define CALL(WW) hipLaunchKernelGGL(kf_apply_indexed<WW>, dim3((unsigned) nb), dim3(F_TPB), 0, e->stream, a, rec, e->sidx[1], n, \
                   check_count, &e->ctrl->fast, e->rw)
  DISPATCH_W3(e, CALL)
#undef CALL
  return SMG_OK;
}

// block ids of the request filter = the leading bm_id_bits(k) bits of a k-mer (a function of k alone, so that the
// maps of all the shards of a table line up): window blocks, coarsened to 2^30 ids (128 MB of bits) at most
// cap: 30 when the maps of several shards are exchanged (128 MB in total), 32 on one GPU, where the probes of the
// first 30 bits are LDS reads (smg_lookup.hpp) and a finer map only costs its memset; SMG_BM_BITS overrides (8..32).
// An id never runs past the k-mer (2k bits) nor past its leading 32 bits.
static int bm_id_bits(int kmer, int cap)
{ { const char *v = test_hook("SMG_BM_BITS"); if (v && atoi(v) >= 8 && atoi(v) <= 32) cap = atoi(v); }
  int nbits = cap > 30 ? 2 * kmer : 2 * (kmer / 2);
  if (nbits > cap) nbits = cap;
  return nbits;
}

// first half of the request filter, which does not need the map (a sharded run overlaps it with the exchange of the
// block maps): long key-only lists are bucketed on their leading 8 bits.  The probes of the 128 MB map are random
// 64-byte fetches otherwise (7 ms for the 4.4e8 requests of the 1 Gbp table); one radix pass (holes of the chunk
// array as sentinels, as for the look-ups) keeps the map words that the resident workgroups probe inside the L2s.
static int filter_presort(smg_engine *e, char *errbuf, size_t errlen)
{ int rc;
  hipEventRecord(e->ev[4], e->stream);                      // start of the filter's time
  if (e->lg.nb)
    { // smg_lookup.hpp: bucket offsets from pass 1's histogram, one-pass partition into the dense array req2
      const int64_t nreq = e->st.nrequests;
      if ((rc = grow(&e->req2, &e->req2_cap, (nreq > 0 ? nreq : 1) * (int64_t) sizeof(u64) * e->rw, errbuf, errlen))) return rc;
      const int nbk = 1 << e->lg.nb;
      // bucket sizes = column sums of the owners' histogram rows -> bucket offsets -> first slot of every owner in every bucket
      hipLaunchKernelGGL(kl_tot, dim3(L_BK / LW_BPW), dim3(LW_SL * LW_BPW), 0, e->stream, (const unsigned *) e->whist, e->nown, e->ghist);
      hipLaunchKernelGGL(kl_scan, dim3(1), dim3(L_BK), 0, e->stream, e->ghist, nbk, e->boff, e->boff + L_BK + 2, e->ghist + L_BK);
      hipLaunchKernelGGL(kl_woff, dim3(L_BK / LW_BPW), dim3(LW_SL * LW_BPW), 0, e->stream, e->whist, e->nown, (const u64 *) e->boff);
      if (e->n_chunks && e->rw == 1)
        hipLaunchKernelGGL(kl_part<1>, dim3(e->nown), dim3(PT_TPB), 0, e->stream, e->req, e->chunk_fill, e->nown,
                           (const unsigned *) e->whist, e->max_chunks, e->lg.nb, e->req2);
      else if (e->n_chunks)
        hipLaunchKernelGGL(kl_part<2>, dim3(e->nown), dim3(PT_TPB), 0, e->stream, e->req, e->chunk_fill, e->nown,
                           (const unsigned *) e->whist, e->max_chunks, e->lg.nb, e->req2);
      HIPCHK(hipGetLastError());
      if (tune_env("SMG_DEBUG"))               // (kl_part's time moves with where its lists lie: profiles/r06_kl_part_addresses.txt)
        fprintf(stderr, "  [smg] kl_part<%d>: req %p (%lld MB) -> req2 %p (%lld MB), whist %p, %u owners, %u chunks\n", e->rw, (void *) e->req,
                (long long) (e->req_cap >> 20), (void *) e->req2, (long long) (e->req2_cap >> 20), (void *) e->whist, e->nown, e->n_chunks);
      e->presorted = 3;
      return SMG_OK;
    }
  e->presorted = 2;
  const int64_t nslots = (int64_t) e->n_chunks * F_CH;
  int64_t sort_min = 1 << 22;               // below this the probes are too few to matter
  { const char *v = test_hook("SMG_FILTER_SORT_MIN"); if (v) sort_min = atoll(v); if (sort_min < SORT_MIN) sort_min = SORT_MIN; }
  if (!(e->rw == 1 && nslots >= sort_min && e->kmer < 32) || tune_env("SMG_FILTER_UNSORTED")) return SMG_OK;
  hipLaunchKernelGGL(kf_fill_holes, dim3(e->n_chunks), dim3(F_TPB), 0, e->stream, e->req, e->chunk_fill);
  if ((rc = grow(&e->req2, &e->req2_cap, nslots * (int64_t) sizeof(u64), errbuf, errlen))) return rc;
  size_t tmp = 0;
  HIPCHK(rocprim::radix_sort_keys<smg_sort_config>(nullptr, tmp, e->req, e->req2, (size_t) nslots, 56u, 64u, e->stream));
  if ((rc = grow((char **) &e->sort_tmp, &e->sort_tmp_cap, (int64_t) tmp + 16, errbuf, errlen))) return rc;
  HIPCHK(rocprim::radix_sort_keys<smg_sort_config>(e->sort_tmp, tmp, e->req, e->req2, (size_t) nslots, 56u, 64u, e->stream));
  e->presorted = 1;
  return SMG_OK;
}

// smg_lookup.hpp: filter the partitioned requests against `map`; list = false: look the survivors up at once
static int lookup_probe(smg_engine *e, const uint32_t *map, bool list, unsigned maxout, char *errbuf, size_t errlen)
{ const bool two = e->bm2 != 0;                             // (a map that came from outside was built by the same rule)
  unsigned grid = 256;
  { int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) == hipSuccess && cus > 0) grid = (unsigned) cus;
  }
  const unsigned nbk = 1u << e->lg.nb;
  FastArgs a = make_fast(e);
  // The fused probe + look-up of a single shard has two forms.  kl_probe (a bucket per workgroup, the bucket's map folded
  // into LDS) is the faster one on a table without long window blocks: 2.5 against 2.7 ms at 2.5e9 entries.  kl_probe_x
  // (XCD by XCD straight from the full-resolution map, small independent workgroups) does not care how long a look-up
  // takes: with 5 % repeats in the genome the survivors' look-ups bisect directory buckets of thousands of k-mers, the
  // sixteen waves of a kl_probe workgroup wait for each other at every bucket, and it takes 4.5 ms against 2.9.  The
  // share of deferred entries that pass 1 reported tells the two kinds of table apart (0.13 % / 0.39 % in the two
  // bench workloads); SMG_PROBE_X=0/1 overrides.
  { const char *px = test_hook("SMG_PROBE_X");
    // ... and so does the share of entries that sent a request: 17.5 % on the diploid tables, a third on the polyploid ones, where
    // one request in seven survives the filter (2.5e7 look-ups at 6.4e8 entries) and kl_probe_x is 7-8 % ahead as well
    const bool auto_x = (e->st.nbig > 0 && e->st.nbig * 400 > e->n) || e->st.nemitted * 100 > e->n * 28;
    if (!list && e->lg.nb >= 3 && (px ? atoi(px) != 0 : auto_x))
      { int rc2;
        if ((rc2 = grow(&e->xtick, &e->xtick_cap, (int64_t) PX_NXCD * PX_TICKW * 4, errbuf, errlen))) return rc2;
        HIPCHK(hipMemsetAsync(e->xtick, 0, (size_t) PX_NXCD * PX_TICKW * 4, e->stream));
        // tickets of 2048 requests where many requests survive (polyploid tables: one in seven, all real hits -- half as many
        // buckets in flight per XCD keep more of a bucket's k-mer lines in reach: -0.25 ms on the hexaploid table), 4096 where the
        // kernel is mostly streaming (a table with repeats: 1 request in 50 survives, and a ticket's fixed cost shows: +0.5 ms at 2048)
        unsigned xw = PX_WGS, part = e->st.nemitted * 100 > e->n * 28 ? PX_PART : 2 * PX_PART;   // (tuning: SMG_PX_WGS workgroups per CU, SMG_PX_PART requests per ticket)
        { const char *v = tune_env("SMG_PX_WGS"); if (v && atoi(v) > 0 && atoi(v) <= 8) xw = (unsigned) atoi(v);
          v = tune_env("SMG_PX_PART"); if (v && atoi(v) >= 512) part = (unsigned) atoi(v) & ~511u;
          if (test_hook("SMG_PX_ONE_XCC")) part |= 0x80000000u;                    // (tests: every workgroup claims XCD 0)
        }
        const unsigned xg = grid * xw;
#define PROBEX(TWO_, RW_) hipLaunchKernelGGL((kl_probe_x<TWO_, RW_>), dim3(xg), dim3(PX_TPB), 0, e->stream, a, (const u64 *) e->req2, \
                            (const u64 *) e->boff, map, e->lg, e->xtick, &e->ctrl->fast, part)
        if (two) { if (e->rw == 1) PROBEX(true, 1); else PROBEX(true, 2); }
        else     { if (e->rw == 1) PROBEX(false, 1); else PROBEX(false, 2); }
#undef PROBEX
        HIPCHK(hipGetLastError());
        return SMG_OK;
      }
  }
  if (grid > nbk) grid = nbk;
#define PROBE(LIST_, TWO_, RW_, OUT_, FILL_, MAX_) hipLaunchKernelGGL((kl_probe<LIST_, TWO_, RW_>), dim3(grid), dim3(PB_TPB), 0, e->stream, a, \
                       (const u64 *) e->req2, (const u64 *) e->boff, map, e->lg, e->ghist + L_BK, OUT_, FILL_, MAX_, &e->ctrl->fast)
#define PROBE_RW(LIST_, TWO_, OUT_, FILL_, MAX_) { if (e->rw == 1) PROBE(LIST_, TWO_, 1, OUT_, FILL_, MAX_); else PROBE(LIST_, TWO_, 2, OUT_, FILL_, MAX_); }
  if (list) { if (two) PROBE_RW(true, true, e->reqf, e->chunk_fillf, maxout) else PROBE_RW(true, false, e->reqf, e->chunk_fillf, maxout) }
  else      { if (two) PROBE_RW(false, true, (u64 *) NULL, (uint32_t *) NULL, 0u) else PROBE_RW(false, false, (u64 *) NULL, (uint32_t *) NULL, 0u) }
#undef PROBE_RW
#undef PROBE
  HIPCHK(hipGetLastError());
  return SMG_OK;
}

// drop the requests whose target window block holds no candidate (kf_filter); map = NULL: this engine's own map
static int fast_filter(smg_engine *e, const uint32_t *map, char *errbuf, size_t errlen)
{ int rc;
  if (!filter_ok(e) || e->n_chunks == 0) return SMG_OK;
  if (!map) map = e->bm_bits ? e->bmap : NULL;
  if (!map) return SMG_OK;
  const int nbits = bm_id_bits(e->kmer, e->bm_cap);
  // two workgroups per CU, chunks dealt round-robin: the chunks in flight then span one or two of the 256 sorted
  // buckets, i.e. <= 1 MB of the map (measured: 512 workgroups 3.0 ms, 256 / 1024 / 2048 workgroups 3.4-3.7 ms)
  unsigned grid = 512;
  { int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) == hipSuccess && cus > 0) grid = 2u * (unsigned) cus;
    const char *v = tune_env("SMG_FILTER_GRID"); if (v && atoi(v) > 0) grid = (unsigned) atoi(v);
  }
  if (grid > e->n_chunks) grid = e->n_chunks;
  unsigned maxout = e->n_chunks + grid + 16;
  if (e->lg.nb)
    { // kl_probe: every wave of every workgroup (one per CU, lookup_probe) fills chunks of its own, each to the brim
      // but for the < 64 records that did not fit: size the list from the launch and the request count, not from 256 CUs
      int cus = 256;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess || cus < 1) cus = 256;
      const int64_t need = (int64_t) cus * PB_WAVES + e->st.nrequests / (F_CH - 63) + 16;
      maxout = need < 0x7FFFFFFFll ? (unsigned) need : 0x7FFFFFFFu;
      if (maxout < e->n_chunks + 16) maxout = e->n_chunks + 16;
    }
  // (the two chunk lists swap roles after every filter: keep them the same size, or pass 1 would reallocate)
  { int64_t want = (int64_t) maxout * F_CH * (int64_t) sizeof(u64) * e->rw, wantf = (int64_t) maxout * 4 + 4;
    if (want < e->req_cap) want = e->req_cap;
    if (wantf < e->chunk_cap) wantf = e->chunk_cap;
    if ((rc = grow(&e->reqf, &e->reqf_cap, want, errbuf, errlen))) return rc;
    if ((rc = grow(&e->chunk_fillf, &e->chunk_capf, wantf, errbuf, errlen))) return rc;
  }
  if (e->rp_active)
    { // replayed step: WHICH chunks the waves of the probe kernel fill is decided by an atomic counter, and how many by the
      // way the buckets fall to the workgroups -- so the routing kernels walk the whole list, and a chunk nobody opened must
      // read as empty
      hipEventRecord(e->ev[11], e->stream);
      // (as many chunks as the recorded step's filter filled, and some: the device checks that this step stayed inside)
      if (e->rp_seen_chunks + 64u < maxout) maxout = e->rp_seen_chunks + 64u;
      HIPCHK(hipMemsetAsync(e->chunk_fillf, 0, (size_t) maxout * 4, e->stream));
    }
  if (!e->presorted && (rc = filter_presort(e, errbuf, errlen))) return rc;
  const int64_t nslots = (int64_t) e->n_chunks * F_CH;
  if (e->presorted == 3)
    { if ((rc = lookup_probe(e, map, true, maxout, errbuf, errlen))) return rc; }
  else if (e->presorted == 1)
    hipLaunchKernelGGL(kf_filter<1>, dim3(grid), dim3(F_TPB), 0, e->stream, e->req2, (const uint32_t *) NULL, e->n_chunks, nslots,
                       map, 64 - nbits, e->reqf, e->chunk_fillf, maxout, &e->ctrl->fast);
  else if (e->rw == 1)
    hipLaunchKernelGGL(kf_filter<1>, dim3(grid), dim3(F_TPB), 0, e->stream, e->req, e->chunk_fill, e->n_chunks, (int64_t) 0,
                       map, 64 - nbits, e->reqf, e->chunk_fillf, maxout, &e->ctrl->fast);
  else if (e->rw == 2)
    hipLaunchKernelGGL(kf_filter<2>, dim3(grid), dim3(F_TPB), 0, e->stream, e->req, e->chunk_fill, e->n_chunks, (int64_t) 0,
                       map, 64 - nbits, e->reqf, e->chunk_fillf, maxout, &e->ctrl->fast);
  else if (e->rw == 3)
    hipLaunchKernelGGL(kf_filter<3>, dim3(grid), dim3(F_TPB), 0, e->stream, e->req, e->chunk_fill, e->n_chunks, (int64_t) 0,
                       map, 64 - nbits, e->reqf, e->chunk_fillf, maxout, &e->ctrl->fast);
  else
    hipLaunchKernelGGL(kf_filter<4>, dim3(grid), dim3(F_TPB), 0, e->stream, e->req, e->chunk_fill, e->n_chunks, (int64_t) 0,
                       map, 64 - nbits, e->reqf, e->chunk_fillf, maxout, &e->ctrl->fast);
  hipEventRecord(e->ev[5], e->stream);
  HIPCHK(hipGetLastError());
  if (e->rp_active)
    { // replayed step: the kept list is as long as last time (the device checks it: smg_engine_proof) -- nothing is read
      hipEventRecord(e->ev[12], e->stream);
      e->rp_nf_chunks = maxout;             // (the device checks that the probe kernel stayed inside the list)
    }
  else
    { if ((rc = read_ctrl(e, errbuf, errlen))) return rc;
      if (e->h_ctrl->fast.nf_chunks > maxout) return fail(errbuf, errlen, SMG_ENODEV, "request filter overflowed its chunk list%s");
    }
  { u64 *t = e->req; e->req = e->reqf; e->reqf = t; }
  { int64_t t = e->req_cap; e->req_cap = e->reqf_cap; e->reqf_cap = t; }
  { uint32_t *t = e->chunk_fill; e->chunk_fill = e->chunk_fillf; e->chunk_fillf = t; }
  { int64_t t = e->chunk_cap; e->chunk_cap = e->chunk_capf; e->chunk_capf = t; }
  if (e->rp_active)
    { e->n_chunks = e->rp_nf_chunks; e->st.nrequests = e->rp_nf_req; e->filtered = true;
      return SMG_OK;
    }
  e->n_chunks = e->h_ctrl->fast.nf_chunks;
  e->st.nrequests = (int64_t) e->h_ctrl->fast.nf_req;
  e->filtered = true;
  float ms = 0; hipEventElapsedTime(&ms, e->ev[4], e->ev[5]);
  e->st.ms_rclookup += ms;
  e->st.ms_filter = ms;
  // what a replayed step on this table may take for granted (key-only records through the look-up chain, k <= 64)
  e->rp_have = e->rp_want && e->presorted == 3 && e->W <= 2 && e->rw == e->W && !e->h_p1cold->times;
  if (e->rp_have)
    { e->rp_nreq = e->st.nemitted; e->rp_nbig = e->st.nbig; e->rp_nf_req = e->st.nrequests; e->rp_bm_bits = e->bm_bits;
      e->rp_seen_chunks = e->n_chunks; e->rp_nranks = 0;
    }
  return SMG_OK;
}

// squeeze the request chunks (ragged fills) into the dense array e->dense: exclusive scan of the fills + one copy
static int compact_chunks(smg_engine *e, int64_t nreq, char *errbuf, size_t errlen)
{ int rc;
  if ((rc = grow(&e->dense, &e->dense_cap, nreq * (int64_t) sizeof(u64) * e->rw, errbuf, errlen))) return rc;
  if ((rc = grow(&e->chunk_off, &e->chunk_off_cap, (int64_t) e->n_chunks * 4 + 4, errbuf, errlen))) return rc;
  size_t tmp = 0;
  HIPCHK(rocprim::exclusive_scan(nullptr, tmp, e->chunk_fill, e->chunk_off, 0u, (size_t) e->n_chunks,
                                 rocprim::plus<uint32_t>(), e->stream));
  if ((rc = grow((char **) &e->sort_tmp, &e->sort_tmp_cap, (int64_t) tmp + 16, errbuf, errlen))) return rc;
  HIPCHK(rocprim::exclusive_scan(e->sort_tmp, tmp, e->chunk_fill, e->chunk_off, 0u, (size_t) e->n_chunks,
                                 rocprim::plus<uint32_t>(), e->stream));
  hipLaunchKernelGGL(kf_compact, dim3(e->n_chunks), dim3(F_TPB), 0, e->stream, e->req, e->chunk_fill, e->chunk_off, e->rw, e->dense);
  return SMG_OK;
}

// look-ups of this engine's own request chunks (flat = NULL; filtered first if a block map was built) or of a flat
// array of received records
static int fast_apply(smg_engine *e, const u64 *flat, int64_t nflat, int check_count, int64_t *missing,
                      char *errbuf, size_t errlen)
{ int rc = SMG_OK;
  if (!flat && e->lg.nb && e->bm_bits && !e->filtered && !tune_env("SMG_LOOKUP_SPLIT"))
    { // own requests, own map: partition, then filter and look-ups in one kernel (no survivor list, no sort)
      if (e->n_chunks == 0 || e->st.nrequests == 0) { if (missing) *missing = 0; return SMG_OK; }
      if (!e->presorted && (rc = filter_presort(e, errbuf, errlen))) return rc;
      hipEvent_t mid = e->ev[10];
      hipEventRecord(mid, e->stream);
      if ((rc = lookup_probe(e, e->bmap, false, 0u, errbuf, errlen))) return rc;
      hipEventRecord(e->ev[5], e->stream);
      if ((rc = read_ctrl(e, errbuf, errlen))) return rc;
      float ms = 0;
      hipEventElapsedTime(&ms, e->ev[4], mid); e->st.ms_filter = ms;       // scan + partition
      hipEventElapsedTime(&ms, e->ev[4], e->ev[5]); e->st.ms_rclookup += ms;
      e->st.nrequests = (int64_t) e->h_ctrl->fast.nf_req;
      e->filtered = true;
      e->n_chunks = 0;                                            // nothing left to route or to look up
      if (missing) *missing = e->h_ctrl->fast.missing;
      return SMG_OK;
    }
  if (!flat && e->bm_bits && !e->filtered && (rc = fast_filter(e, NULL, errbuf, errlen))) return rc;
  hipEventRecord(e->ev[4], e->stream);
  const bool keys_only = e->W == 1 && e->rw == 1;
  const int64_t nreq = e->st.nrequests;
  if (flat)
    { if (nflat > 0) rc = keys_only ? apply_sorted(e, flat, nflat, 0, errbuf, errlen) : apply_indexed(e, flat, nflat, check_count, errbuf, errlen); }
  else if (nreq > 0 && e->n_chunks > 0)
    { const int64_t nslots = (int64_t) e->n_chunks * F_CH;
      if (keys_only && nslots <= nreq + nreq / 8 && nslots >= SORT_MIN && e->kmer < 32)
        { // nearly every chunk is full (pass 1 and the filter split their batches): sort the chunk array as it is,
          // holes as sentinels behind the requests, instead of compacting it first
          hipLaunchKernelGGL(kf_fill_holes, dim3(e->n_chunks), dim3(F_TPB), 0, e->stream, e->req, e->chunk_fill);
          rc = apply_sorted(e, e->req, nslots, 1, errbuf, errlen);
        }
      else
        { // ragged chunks, k = 32 (the all-T k-mer equals the sentinel) or records with a count/flag word
          // (W > 1, exact proof): compact, then sorted (keys) or index-sorted (records) look-ups
          if ((rc = compact_chunks(e, nreq, errbuf, errlen))) return rc;
          rc = keys_only ? apply_sorted(e, e->dense, nreq, 0, errbuf, errlen) : apply_indexed(e, e->dense, nreq, check_count, errbuf, errlen);
        }
    }
  if (rc) return rc;
  hipEventRecord(e->ev[5], e->stream);
  HIPCHK(hipGetLastError());
  if (!missing && flat)                   // (sharded run: the count of missing complements stays on the device -- smg_engine_proof)
    { e->lookup_pending = true; return SMG_OK; }
  rc = read_ctrl(e, errbuf, errlen);
  if (rc) return rc;
  float ms = 0; hipEventElapsedTime(&ms, e->ev[4], e->ev[5]);
  e->st.ms_rclookup += ms;
  if (missing) *missing = e->h_ctrl->fast.missing;
  return SMG_OK;
}

static int fast_pass2(smg_engine *e, int64_t *d_plot, bool with_sum, char *errbuf, size_t errlen)
{ HIPCHK(hipMemsetAsync(d_plot, 0, sizeof(int64_t) * SMG_PLOT_CELLS, e->stream));
  FastArgs a = make_fast(e);
  hipEventRecord(e->ev[6], e->stream);
  if (e->n > 0)
    { int64_t nb = (e->n + P2_TPB - 1) / P2_TPB;
      if (nb > P2_GRID) nb = P2_GRID;
#define CALL(WW) hipLaunchKernelGGL(kf_pass2<WW>, dim3((unsigned) nb), dim3(P2_TPB), 0, e->stream, a, (u64 *) d_plot)
      DISPATCH_W3(e, CALL)
#undef CALL
      // entries whose unique partner is out of the code's reach: from the list of deferred entries that pass 1 of one-
      // and two-word k-mers leaves, else (three words: the generic pass 1) from a scan of the code bytes
      if (e->far_listed && e->W <= 2)
        { const unsigned nl = (unsigned) e->st.nbig;
          unsigned fb = (nl + 255) / 256;
          if (fb > 4096) fb = 4096;
          if (nl && e->W == 1) hipLaunchKernelGGL(kf_pass2_far<1>, dim3(fb), dim3(256), 0, e->stream, a, (const uint32_t *) e->biglist, (const uint32_t *) e->farp, nl, (u64 *) d_plot);
          else if (nl)         hipLaunchKernelGGL(kf_pass2_far<2>, dim3(fb), dim3(256), 0, e->stream, a, (const uint32_t *) e->biglist, (const uint32_t *) e->farp, nl, (u64 *) d_plot);
        }
      else
        { int64_t fb = ((e->n + 15) / 16 + 255) / 256;
          if (fb > 4096) fb = 4096;
#define CALL(WW) hipLaunchKernelGGL(kf_pass2_farscan<WW>, dim3((unsigned) fb), dim3(256), 0, e->stream, a, (u64 *) d_plot)
          DISPATCH_W3(e, CALL)
#undef CALL
        }
    }
  hipEventRecord(e->ev[7], e->stream);
  HIPCHK(hipGetLastError());
  int rc = SMG_OK;
  e->st.path = 1;
  if (!with_sum)                         // (phase API: no host wait here; smg_engine_stats reads the events once they have fired)
    { e->st.npairs = 0; e->st.ms_pass2 = -1.0; return SMG_OK; }
  rc = plot_sum(e, d_plot, errbuf, errlen);                         // (one more kernel and a host round trip)
  if (rc) return rc;
  float ms = 0; hipEventElapsedTime(&ms, e->ev[6], e->ev[7]);
  e->st.ms_pass2 = ms;
  return SMG_OK;
}

// ---- out-of-core shards: take a shard up again after its keys were dropped (smg_multi.hpp, host_run_sequential) ------------
// The shard's k-mers and counts are back in the engine (decoded again from the table), `d_codes` are the code bytes its
// pass 1 left.  What the look-ups of the received requests and pass 2 need besides is the directory: the table's prefix
// index if it was handed over, else one pass of k_directory.  No candidate map, no signatures, no list of deferred entries
// (pass 2 finds the far partners by scanning the code bytes, as it does for three-word k-mers).
static int fast_resume(smg_engine *e, const uint8_t *d_codes, int with_meta, char *errbuf, size_t errlen)
{ int rc;
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemsetAsync(e->ctrl, 0, sizeof(Ctrl), e->stream));
  set_geo(e);
  e->fast = true; e->counted_done = false; e->lookup_pending = false;
  if ((rc = grow(&e->deg, &e->deg_cap, ((e->n + 15) & ~15ll) + 32, errbuf, errlen))) return rc;
  if (e->n > 0) HIPCHK(hipMemcpyAsync(e->deg, d_codes, (size_t) e->n, hipMemcpyDeviceToDevice, e->stream));
  e->rw = e->W + ((with_meta || e->W > 2) ? 1 : 0);
  e->use_sig = false; e->bm_bits = 0; e->bm2 = 0; e->lg.nb = 0;
  e->filtered = false; e->presorted = 0; e->n_chunks = 0; e->far_listed = false;
  e->st.nrequests = 0; e->st.ms_rclookup = 0;
  memset(e->fp, 0, sizeof(e->fp));
  if ((rc = dir_geometry(e, 8, errbuf, errlen, true))) return rc;
  if (!e->dir_preset)
    { HIPCHK(hipMemsetAsync(e->bstart, e->n > 0 ? 0xFF : 0, sizeof(uint32_t) * ((size_t) e->dir.nb + 2), e->stream));
      if (e->n > 0)
        { Tab t = make_tab(e);
          const unsigned nblk = (unsigned) ((e->n + 1 + TPB - 1) / TPB);
#define CALL(WW) hipLaunchKernelGGL(k_directory<WW>, dim3(nblk), dim3(TPB), 0, e->stream, t, e->bstart, e->ctrl)
          DISPATCH_W(e, CALL)
#undef CALL
          HIPCHK(hipGetLastError());
        }
    }
  e->prepared = true;
  return SMG_OK;
}

// the same for k > 85 (round 6): the byte a shard left behind is its array of counted degrees (S_all of every entry, uint8 with the
// reference's wrap); the look-ups of the received requests add S_hi of the complements on top (k_apply) and pass 2 reads the sums
static int counted_prepare(smg_engine *e, char *errbuf, size_t errlen);
static int counted_resume(smg_engine *e, const uint8_t *d_degs, char *errbuf, size_t errlen)
{ HIPCHK(hipSetDevice(e->device));
  int rc = counted_prepare(e, errbuf, errlen);               // control words, geometry, zeroed degrees, directory
  if (rc) return rc;
  if (e->n > 0) HIPCHK(hipMemcpyAsync(e->deg, d_degs, (size_t) e->n, hipMemcpyDeviceToDevice, e->stream));
  e->rw = e->W + 1;
  e->lookup_pending = false; e->n_chunks = 0; e->bm_bits = 0; e->filtered = false; e->presorted = 0;
  e->st.nrequests = 0; e->st.ms_rclookup = 0; e->st.path = 1;
  memset(e->fp, 0, sizeof(e->fp));
  e->prepared = true;
  return SMG_OK;
}

// ---- public phase API (sharded runs) ------------------------------------------------------------
// k <= 85: the fast path.  k > 85: the counted path in the same steps (counted_phase_* below) -- pass 1 leaves ONE flat
// list of (rc(x), count | S_hi << 16) records, presented to the router as full chunks; no block map, nothing is filtered.

#define NEED_FAST(e)                                                                          \
  if (!(e)) return fail(errbuf, errlen, SMG_EINVAL, "null engine%s");                          \
  if ((e)->kmer > FAST_MAX_K)                                                                 \
    return fail(errbuf, errlen, SMG_EINVAL, "no block map and no request filter above k = 85%s");
#define NEED_ENGINE(e)  if (!(e)) return fail(errbuf, errlen, SMG_EINVAL, "null engine%s");

static int counted_phase_pass1(smg_engine *e, int symcheck, char *errbuf, size_t errlen);
static int counted_phase_apply(smg_engine *e, const u64 *rec, int64_t nrec, int64_t *missing, char *errbuf, size_t errlen);
static int counted_phase_pass2(smg_engine *e, int64_t *d_plot, char *errbuf, size_t errlen);

extern "C" int smg_engine_pass1(smg_engine *e, int symcheck, char *errbuf, size_t errlen)
{ NEED_ENGINE(e)
  HIPCHK(hipSetDevice(e->device));
  e->st.ms_rclookup = 0; e->lookup_pending = false;
  if (e->kmer > FAST_MAX_K) return counted_phase_pass1(e, symcheck, errbuf, errlen);
  e->bm_cap = e->bm_want ? e->bm_want : 30;  // default: the maps of the shards are exchanged, 128 MB in total
  e->rp_active = false;
  if (e->rp_want && e->rp_have && symcheck == SMG_SYM_HASH && e->rp_sym == symcheck && e->n > 0
      && e->rp_bm_bits == bm_id_bits(e->kmer, e->bm_cap))
    { // the step before this one, on this very table, went through the look-up chain: queue this one from its counts
      const int rc = fast_pass1(e, 0, 0, 1, errbuf, errlen, true);
      if (rc) return rc;
      e->rp_active = true; e->rp_routed = false;
      return SMG_OK;
    }
  e->rp_sym = symcheck;
  return fast_pass1(e, symcheck == SMG_SYM_EXACT, symcheck == SMG_SYM_EXACT, symcheck == SMG_SYM_HASH, errbuf, errlen);
}

extern "C" int smg_engine_set_replay(smg_engine *e, int on)
{ if (!e) return SMG_EINVAL;
  e->rp_want = on != 0;
  if (!on) e->rp_have = false;
  return SMG_OK;
}

// is the current step a replayed one (1), and did the LAST finished step leave a record (2)?
extern "C" int smg_engine_replay_state(smg_engine *e) { return e ? (e->rp_active ? 1 : 0) | (e->rp_have ? 2 : 0) : 0; }

// after the caller has read the proof words of a replayed step: ok != 0 = the device found every count as recorded (the
// run stands: the engine's bookkeeping is brought up to date from the control words, which are complete by now),
// ok == 0 = the record is dropped and the next step runs the plain way
extern "C" int smg_engine_replay_done(smg_engine *e, int ok, char *errbuf, size_t errlen)
{ if (!e) return fail(errbuf, errlen, SMG_EINVAL, "null engine%s");
  if (!e->rp_active) return SMG_OK;
  e->rp_active = false;
  HIPCHK(hipSetDevice(e->device));
  if (!ok) { e->rp_have = false; e->dbits_dirty = true; e->lookup_pending = false; return SMG_OK; }
  int rc = read_ctrl(e, errbuf, errlen);
  if (rc) return rc;
  e->dbits_dirty = false;
  memset(e->fp, 0, sizeof(e->fp));
  for (unsigned b = 0; b < e->rp_grid; b++)
    for (int q = 0; q < 4; q++) e->fp[q] ^= e->h_partials[b * 4 + q];
  float ms = 0;
  hipEventElapsedTime(&ms, e->ev[2], e->ev[3]); e->st.ms_pass1 = ms;
  hipEventElapsedTime(&ms, e->ev[0], e->ev[3]); e->st.ms_bigfix = ms;
  hipEventElapsedTime(&ms, e->ev[11], e->ev[12]); e->st.ms_filter = ms; e->st.ms_rclookup += ms;
  return SMG_OK;
}

extern "C" int64_t smg_engine_nreq(smg_engine *e) { return e ? e->st.nrequests : 0; }
extern "C" int smg_engine_record_words(smg_engine *e) { return e ? e->rw : 0; }

extern "C" int smg_engine_apply(smg_engine *e, const uint64_t *d_recv, int64_t nrecv,
                                int64_t *missing, char *errbuf, size_t errlen)
{ NEED_ENGINE(e)
  if (!e->prepared) return fail(errbuf, errlen, SMG_EINVAL, "apply before pass1%s");
  HIPCHK(hipSetDevice(e->device));
  if (!e->fast) return counted_phase_apply(e, (const u64 *) d_recv, nrecv, missing, errbuf, errlen);
  return fast_apply(e, (const u64 *) d_recv, nrecv, 1, missing, errbuf, errlen);
}

__global__ void k_proof_words(const Ctrl *__restrict__ ctrl, int fast, u64 f0, u64 f1, u64 *__restrict__ dst)
{ if (threadIdx.x == 0) { dst[0] = fast ? (u64) ctrl->fast.missing : ctrl->missing; dst[1] = f0; dst[2] = f1; dst[3] = 0; } }

// the same for a replayed step: the fingerprint residue is folded from the workgroups' partial words on the device, and the
// counts the step was queued with are compared with what its kernels reported -- any difference sets dst[3], on which the
// caller (all ranks of a sharded run: the word is summed with the proof) drops the record and runs the step again the plain way
struct ReplayExpect { u64 nreq, nf_req; unsigned nbig, nf_chunks, max_chunks, grid; const u64 *totals; int nranks; };
__global__ void __launch_bounds__(64)
k_proof_replay(const Ctrl *__restrict__ ctrl, const u64 *__restrict__ partials, ReplayExpect x, u64 *__restrict__ dst)
{ u64 f0 = 0, f1 = 0;
  for (unsigned b = threadIdx.x; b < x.grid; b += 64)
    { f0 ^= partials[(size_t) b * 4] ^ partials[(size_t) b * 4 + 2]; f1 ^= partials[(size_t) b * 4 + 1] ^ partials[(size_t) b * 4 + 3]; }
  f0 = wave_xor_u64(f0); f1 = wave_xor_u64(f1);
  // the router's per-destination totals: recorded step in totals[0..16), this step in totals[16..32)
  bool moved = x.totals == NULL;
  if (x.totals && (int) threadIdx.x < x.nranks) moved = x.totals[threadIdx.x] != x.totals[16 + threadIdx.x];
  const bool anymoved = __ballot(moved) != 0;
  if (threadIdx.x == 0)
    { const FastCtl &f = ctrl->fast;
      const bool bad = f.unsorted != 0 || f.n_chunks > x.max_chunks || f.nbig != x.nbig || f.nreq != x.nreq
                       || f.nf_chunks > x.nf_chunks || f.nf_req != x.nf_req || anymoved;
      dst[0] = (u64) f.missing; dst[1] = f0; dst[2] = f1; dst[3] = bad ? 1ull : 0ull;
    }
}

// the proof tail of a sharded step's reduction buffer in ONE launch: tail[0] = missing, tail[1 + 2 s .. 2 + 2 s] = this rank's
// residue in ITS slot s (zeros in the others'), tail[1 + 2 nslots] = a replayed step found other counts, tail[2 + 2 nslots] =
// this step was a replayed one; src = the four words of k_proof_words / k_proof_replay
__global__ void __launch_bounds__(64)
k_proof_tail(const u64 *__restrict__ src, int nslots, int slot, int replayed, u64 *__restrict__ tail)
{ const int n = 3 + 2 * nslots;
  for (int i = threadIdx.x; i < n; i += 64)
    { u64 v = 0;
      if (i == 0) v = src[0];
      else if (i == 1 + 2 * slot) v = src[1];
      else if (i == 2 + 2 * slot) v = src[2];
      else if (i == n - 2) v = src[3];
      else if (i == n - 1) v = replayed ? 1ull : 0ull;
      tail[i] = v;
    }
}

static int proof_words(smg_engine *e, u64 *d_dst, char *errbuf, size_t errlen)
{ if (e->rp_active)
    { ReplayExpect x;
      x.nreq = (u64) e->rp_nreq; x.nf_req = (u64) e->rp_nf_req; x.nbig = (unsigned) e->rp_nbig; x.nf_chunks = e->rp_nf_chunks;
      x.max_chunks = e->max_chunks; x.grid = e->rp_grid;
      x.totals = e->rp_routed ? e->rp_totals : (const u64 *) NULL; x.nranks = e->rp_nranks;     // (not routed: nothing to vouch for the split)
      hipLaunchKernelGGL(k_proof_replay, dim3(1), dim3(64), 0, e->stream, (const Ctrl *) e->ctrl, (const u64 *) e->partials, x, d_dst);
    }
  else
    hipLaunchKernelGGL(k_proof_words, dim3(1), dim3(64), 0, e->stream,
                       (const Ctrl *) e->ctrl, e->fast ? 1 : 0, e->fp[0] ^ e->fp[2], e->fp[1] ^ e->fp[3], d_dst);
  HIPCHK(hipGetLastError());
  return SMG_OK;
}

extern "C" int smg_engine_proof(smg_engine *e, uint64_t *d_dst, char *errbuf, size_t errlen)
{ NEED_ENGINE(e)
  if (!e->prepared || !d_dst) return fail(errbuf, errlen, SMG_EINVAL, "proof before pass1%s");
  HIPCHK(hipSetDevice(e->device));
  return proof_words(e, (u64 *) d_dst, errbuf, errlen);
}

extern "C" int smg_engine_proof_tail(smg_engine *e, uint64_t *d_tail, int nslots, int slot, char *errbuf, size_t errlen)
{ NEED_ENGINE(e)
  if (!e->prepared || !d_tail || nslots < 1 || nslots > 16 || slot < 0 || slot >= nslots)
    return fail(errbuf, errlen, SMG_EINVAL, "proof_tail: 1..16 slots, after pass1%s");
  HIPCHK(hipSetDevice(e->device));
  u64 *tmp = e->partials + (size_t) 4 * P1_MAXGRID;           // (the row behind the workgroups' partial sums)
  int rc = proof_words(e, tmp, errbuf, errlen);
  if (rc) return rc;
  hipLaunchKernelGGL(k_proof_tail, dim3(1), dim3(64), 0, e->stream, (const u64 *) tmp, nslots, slot, e->rp_active ? 1 : 0, (u64 *) d_tail);
  HIPCHK(hipGetLastError());
  return SMG_OK;
}

extern "C" int smg_engine_apply_own(smg_engine *e, int64_t *missing, char *errbuf, size_t errlen)
{ NEED_ENGINE(e)
  if (!e->prepared) return fail(errbuf, errlen, SMG_EINVAL, "apply before pass1%s");
  HIPCHK(hipSetDevice(e->device));
  if (!e->fast) return counted_phase_apply(e, e->req, e->st.nrequests, missing, errbuf, errlen);
  return fast_apply(e, NULL, 0, 1, missing, errbuf, errlen);
}

extern "C" int smg_engine_set_blockmap_bits(smg_engine *e, int id_bits)
{ if (!e || (id_bits != 0 && (id_bits < 8 || id_bits > 32))) return SMG_EINVAL;
  e->bm_want = id_bits;
  return SMG_OK;
}

extern "C" int smg_engine_blockmap(smg_engine *e, int *id_bits, int64_t *nwords)
{ if (!e || !id_bits || !nwords) return SMG_EINVAL;
  *id_bits = e->prepared ? e->bm_bits : 0;
  *nwords = *id_bits ? (((1ll << *id_bits) + 31) >> 5) << e->bm2 : 0;      // (two-bit map: 64 bits per 32 block ids)
  return SMG_OK;
}

extern "C" int smg_engine_blockmap_copy(smg_engine *e, int64_t word_lo, int64_t nw, uint32_t *d_dst,
                                        char *errbuf, size_t errlen)
{ NEED_FAST(e)
  if (!e->prepared || !e->bm_bits) return fail(errbuf, errlen, SMG_EINVAL, "no block map: pass 1 (hash proof, k <= 85) has not run%s");
  const int64_t nwords = (((1ll << e->bm_bits) + 31) >> 5) << e->bm2;
  if (word_lo < 0 || nw < 0 || word_lo + nw > nwords || (nw > 0 && !d_dst))
    return fail(errbuf, errlen, SMG_EINVAL, "block map range out of bounds%s");
  HIPCHK(hipSetDevice(e->device));
  if (nw > 0) HIPCHK(hipMemcpyAsync(d_dst, e->bmap + word_lo, (size_t) nw * 4, hipMemcpyDeviceToDevice, e->stream));
  return SMG_OK;
}

struct MapGeo { int64_t lo[16], ln[16]; };

__global__ void __launch_bounds__(256)
km_merge_maps(const uint32_t *__restrict__ parts, int64_t width, int nranks, MapGeo geo, uint32_t *__restrict__ full, int64_t nwords)
{ __shared__ int64_t lo[16], ln[16];
  if (threadIdx.x < 16) { lo[threadIdx.x] = geo.lo[threadIdx.x]; ln[threadIdx.x] = geo.ln[threadIdx.x]; }
  __syncthreads();
  for (int64_t w = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (int64_t) gridDim.x * blockDim.x)
    { uint32_t v = 0;
      for (int r = 0; r < nranks; r++)
        { const int64_t o = w - lo[r];
          if (o >= 0 && o < ln[r]) v |= parts[(size_t) r * width + o];
        }
      full[w] = v;
    }
}

extern "C" int smg_engine_merge_maps(smg_engine *e, const uint32_t *d_parts, int64_t width, int nranks,
                                     const int64_t *word_lo, const int64_t *nwords_of, uint32_t *d_full,
                                     char *errbuf, size_t errlen)
{ NEED_FAST(e)
  if (!e->prepared || !e->bm_bits) return fail(errbuf, errlen, SMG_EINVAL, "no block map: pass 1 (hash proof, k <= 85) has not run%s");
  if (!d_parts || !d_full || !word_lo || !nwords_of || nranks < 1 || nranks > 16 || width < 1)
    return fail(errbuf, errlen, SMG_EINVAL, "bad merge_maps arguments (1..16 ranks)%s");
  const int64_t nwords = (((1ll << e->bm_bits) + 31) >> 5) << e->bm2;
  MapGeo geo;
  for (int r = 0; r < 16; r++) { geo.lo[r] = r < nranks ? word_lo[r] : 0; geo.ln[r] = r < nranks ? nwords_of[r] : 0; }
  for (int r = 0; r < nranks; r++)
    if (geo.lo[r] < 0 || geo.ln[r] < 0 || geo.ln[r] > width || geo.lo[r] + geo.ln[r] > nwords)
      return fail(errbuf, errlen, SMG_EINVAL, "block map range out of bounds%s");
  HIPCHK(hipSetDevice(e->device));
  hipLaunchKernelGGL(km_merge_maps, dim3(2048), dim3(256), 0, e->stream, d_parts, width, nranks, geo, d_full, nwords);
  HIPCHK(hipGetLastError());
  return SMG_OK;
}

extern "C" int smg_engine_presort(smg_engine *e, char *errbuf, size_t errlen)
{ NEED_ENGINE(e)
  if (!e->prepared) return fail(errbuf, errlen, SMG_EINVAL, "presort before pass1%s");
  if (!e->fast || !filter_ok(e) || !e->bm_bits || e->filtered || e->presorted || e->n_chunks == 0) return SMG_OK;
  HIPCHK(hipSetDevice(e->device));
  return filter_presort(e, errbuf, errlen);
}

extern "C" int smg_engine_filter(smg_engine *e, const uint32_t *d_map, int64_t *kept, char *errbuf, size_t errlen)
{ NEED_FAST(e)
  if (!e->prepared) return fail(errbuf, errlen, SMG_EINVAL, "filter before pass1%s");
  if (!filter_ok(e)) return fail(errbuf, errlen, SMG_EINVAL, "the request filter covers the hash proof at k <= 85%s");
  if (!d_map && !e->bm_bits) return fail(errbuf, errlen, SMG_EINVAL, "no block map to filter with%s");
  HIPCHK(hipSetDevice(e->device));
  int rc = e->filtered ? SMG_OK : fast_filter(e, d_map, errbuf, errlen);
  if (kept) *kept = e->st.nrequests;
  return rc;
}

extern "C" int smg_engine_symhash(smg_engine *e, uint64_t out[4], char *errbuf, size_t errlen)
{ if (!e || !out) return fail(errbuf, errlen, SMG_EINVAL, "null argument%s");
  for (int i = 0; i < 4; i++) out[i] = e->fp[i];
  return SMG_OK;
}

// records (rw words each, the first W of them a k-mer) in chunks of F_CH -> d_send, grouped by the rank whose k-mer
// range holds the k-mer (splitters = first k-mer of ranks 1..nranks-1); counts[nranks] on the host
static int route_records(smg_engine *e, const u64 *req, const uint32_t *chunk_fill, unsigned nc, int rw,
                         const uint64_t *splitters, int nranks, uint64_t *d_send, int64_t capacity, int64_t *counts,
                         char *errbuf, size_t errlen, int64_t *d_counts = NULL /* device: the totals stay there, no host wait */)
{ if (counts) for (int r = 0; r < nranks; r++) counts[r] = 0;
  if (nc == 0)
    { if (d_counts) HIPCHK(hipMemsetAsync(d_counts, 0, sizeof(int64_t) * nranks, e->stream));
      return SMG_OK;
    }
  int rc;
  if (nranks > 1)
    HIPCHK(hipMemcpyAsync(e->d_split, splitters, sizeof(u64) * (nranks - 1) * e->W,
                          hipMemcpyHostToDevice, e->stream));
  if ((rc = grow(&e->route_cnt, &e->route_cnt_cap, (int64_t) nc * nranks * 4, errbuf, errlen))) return rc;
  if ((rc = grow(&e->route_off, &e->route_off_cap, ((int64_t) nc * nranks + 16) * 8, errbuf, errlen))) return rc;
  u64 *totals = e->route_off + (size_t) nc * nranks;           // [nranks], behind the offsets
#define CALL(WW) hipLaunchKernelGGL(kf_route_count<WW>, dim3(nc), dim3(F_TPB), 0, e->stream, req, \
                   chunk_fill, rw, e->d_split, nranks, e->route_cnt)
  DISPATCH_W(e, CALL)
#undef CALL
  // offsets and scatter follow on the device; the host only learns the totals (what the exchange needs): ONE round trip
  hipLaunchKernelGGL(kf_route_offsets, dim3(nranks), dim3(1024), 0, e->stream, (const uint32_t *) e->route_cnt, nc, nranks, e->route_off, totals);
#define CALL(WW) hipLaunchKernelGGL(kf_route_scatter<WW>, dim3(nc), dim3(F_TPB), 0, e->stream, req, \
                   chunk_fill, rw, e->d_split, nranks, (const u64 *) e->route_off, (const u64 *) totals, (u64 *) d_send, \
                   (u64) (capacity < 0 ? 0 : capacity))
  DISPATCH_W(e, CALL)
#undef CALL
  HIPCHK(hipGetLastError());
  if (d_counts)
    { HIPCHK(hipMemcpyAsync(d_counts, totals, sizeof(u64) * nranks, hipMemcpyDeviceToDevice, e->stream));
      return SMG_OK;
    }
  u64 ht[16];
  HIPCHK(hipMemcpyAsync(ht, totals, sizeof(u64) * nranks, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (int r = 0; r < nranks; r++) counts[r] = (int64_t) ht[r];
  return SMG_OK;
}

extern "C" int smg_engine_route(smg_engine *e, const uint64_t *splitters, int nranks,
                                uint64_t *d_send, int64_t capacity, int64_t *counts,
                                char *errbuf, size_t errlen)
{ NEED_ENGINE(e)
  if (!counts || nranks < 1 || nranks > 16)
    return fail(errbuf, errlen, SMG_EINVAL, "bad route arguments (1..16 ranks)%s");
  if (!e->prepared) return fail(errbuf, errlen, SMG_EINVAL, "route before pass1%s");
  HIPCHK(hipSetDevice(e->device));
  if (e->st.nrequests > capacity) return fail(errbuf, errlen, SMG_EINVAL, "send buffer too small%s");
  return route_records(e, e->req, e->chunk_fill, e->n_chunks, e->rw, splitters, nranks, d_send, capacity, counts, errbuf, errlen);
}

extern "C" int smg_engine_route_device(smg_engine *e, const uint64_t *splitters, int nranks, uint64_t *d_send, int64_t capacity,
                                       int64_t *d_counts, char *errbuf, size_t errlen)
{ NEED_ENGINE(e)
  if (!d_counts || nranks < 1 || nranks > 16)
    return fail(errbuf, errlen, SMG_EINVAL, "bad route arguments (1..16 ranks)%s");
  if (!e->prepared) return fail(errbuf, errlen, SMG_EINVAL, "route before pass1%s");
  HIPCHK(hipSetDevice(e->device));
  if (e->st.nrequests > capacity) return fail(errbuf, errlen, SMG_EINVAL, "send buffer too small%s");
  int rc = route_records(e, e->req, e->chunk_fill, e->n_chunks, e->rw, splitters, nranks, d_send, capacity, NULL, errbuf, errlen, d_counts);
  if (rc || !e->rp_want || !e->fast) return rc;
  // replay: the per-destination totals of a plain step are kept, those of a replayed step are put beside them -- the verdict
  // kernel (smg_engine_proof) compares the two: the caller splits its exchange by the RECORDED totals
  if (!e->rp_totals) HIPCHK(hipMalloc(&e->rp_totals, sizeof(u64) * 32));
  const bool rep = e->rp_active;
  if (rep && e->rp_nranks != nranks) { e->rp_routed = false; return SMG_OK; }
  HIPCHK(hipMemcpyAsync(e->rp_totals + (rep ? 16 : 0), d_counts, sizeof(u64) * nranks, hipMemcpyDeviceToDevice, e->stream));
  if (rep) e->rp_routed = true; else e->rp_nranks = nranks;
  return SMG_OK;
}

extern "C" int smg_engine_pass2(smg_engine *e, int64_t *d_plot, char *errbuf, size_t errlen)
{ NEED_ENGINE(e)
  if (!e->prepared || !d_plot) return fail(errbuf, errlen, SMG_EINVAL, "pass2 before pass1%s");
  HIPCHK(hipSetDevice(e->device));
  if (!e->fast) return counted_phase_pass2(e, d_plot, errbuf, errlen);
  return fast_pass2(e, d_plot, false, errbuf, errlen);
}

extern "C" int smg_engine_stats(smg_engine *e, smg_stats *stats)
{ if (!e || !stats) return SMG_EINVAL;
  if (e->lookup_pending)                 // look-ups of received requests were queued without a wait
    { float ms = 0;
      hipSetDevice(e->device);
      if (hipEventSynchronize(e->ev[5]) == hipSuccess && hipEventElapsedTime(&ms, e->ev[4], e->ev[5]) == hipSuccess) e->st.ms_rclookup += ms;
      e->lookup_pending = false;
    }
  if (e->st.ms_pass2 < 0)                // pass 2 of the phase API was queued without a wait
    { float ms = 0;
      hipSetDevice(e->device);
      if (hipEventSynchronize(e->ev[7]) == hipSuccess && hipEventElapsedTime(&ms, e->ev[6], e->ev[7]) == hipSuccess) e->st.ms_pass2 = ms;
      else e->st.ms_pass2 = 0;
    }
  *stats = e->st;
  return SMG_OK;
}

extern "C" int smg_engine_run(smg_engine *e, int symcheck, int64_t *d_plot, smg_stats *stats,
                              char *errbuf, size_t errlen)
{ if (!e || !d_plot) return fail(errbuf, errlen, SMG_EINVAL, "null argument%s");
  HIPCHK(hipSetDevice(e->device));
  int rc;
  hipEventRecord(e->ev[8], e->stream);
  e->st.ms_pass1 = e->st.ms_rclookup = e->st.ms_pass2 = 0;
  bool symmetric = false;
  if (symcheck != SMG_SYM_NONE && e->kmer <= FAST_MAX_K)
    { // hash : requests from the entries that own a hi-side pair; closure proven by the fingerprint
      // exact: EVERY entry sends (rc(kmer), count) and the look-up compares the count -- the same records the
      //        sharded protocol exchanges, index-sorted look-ups (kf_apply_indexed)
      const int exact = symcheck == SMG_SYM_EXACT;
      e->bm_cap = 32;
      // (Rounds 3-5 queued a re-run on the same table from the counts of the run before -- "run_speculative", one host wait
      //  instead of four.  Measured twice without a gain, 17.94 against 17.7-18.0 ms per step, profiles/r05_lookup_experiments.txt:
      //  taken out in round 6.  The one mechanism of that kind left is the replayed step of the phase API, for sharded runs.)
      rc = fast_pass1(e, exact, exact, symcheck == SMG_SYM_HASH, errbuf, errlen);
      if (rc) return rc;
      int64_t missing = 0;
      if ((rc = fast_apply(e, NULL, 0, exact, &missing, errbuf, errlen))) return rc;
      symmetric = (missing == 0);
      if (symmetric && symcheck == SMG_SYM_HASH)
        symmetric = e->fp[0] == e->fp[2] && e->fp[1] == e->fp[3];
      if (symmetric && (rc = fast_pass2(e, d_plot, true, errbuf, errlen))) return rc;
    }
  else if (symcheck != SMG_SYM_NONE)
    { if ((rc = counted_symmetric(e, symcheck, d_plot, &symmetric, errbuf, errlen))) return rc; }
  if (!symmetric && (rc = run_general(e, d_plot, errbuf, errlen))) return rc;
  hipEventRecord(e->ev[9], e->stream);
  HIPCHK(hipStreamSynchronize(e->stream));
  float ms = 0; hipEventElapsedTime(&ms, e->ev[8], e->ev[9]);
  e->st.ms_total = ms;
  if (stats) *stats = e->st;
  return SMG_OK;
}

// ---- table conditioning on device ----------------------------------------
// The reference shells out to FastK's Logex / Symmex (PloidyPlot.c:1381-1414), which are neither vendored
// nor pinned.  Restated semantics:
//   trim        : keep the entries with count >= ethresh                    (Logex 'A[e-]')
//   symmetrise  : add rc(x) with the count of x for every entry x, sort, keep one copy of a k-mer that
//                 occurs twice (a self-complementary k-mer, or -- only on input that is neither canonical
//                 nor symmetric -- a k-mer whose complement was already present: the original entry wins)
// Sorting is an LSD radix sort over the 64-bit words of the k-mer (stable, rocPRIM Onesweep).

typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                   rocprim::default_config, 0> smg_pair_sort_config;

template <int W> __global__ void __launch_bounds__(TPB)
kc_append_rc(const u64 *__restrict__ keys, const uint16_t *__restrict__ cnt, int64_t n, int k,
             u64 *__restrict__ okeys, uint16_t *__restrict__ ocnt)
{ const int64_t i = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const Key<W> x = load_key<W>(keys, i);
  const Key<W> r = revcomp<W>(x, k);
#pragma unroll
  for (int w = 0; w < W; w++) { okeys[i * W + w] = x.w[w]; okeys[(n + i) * W + w] = r.w[w]; }
  ocnt[i] = cnt[i]; ocnt[n + i] = cnt[i];
}

__global__ void __launch_bounds__(TPB) kc_iota(uint32_t *__restrict__ p, int64_t n)
{ const int64_t i = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i < n) p[i] = (uint32_t) i;
}

__global__ void __launch_bounds__(TPB)
kc_gather_word(const u64 *__restrict__ keys, const uint32_t *__restrict__ perm, int W, int w, int64_t n,
               u64 *__restrict__ out)
{ const int64_t i = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i < n) out[i] = keys[(int64_t) perm[i] * W + w];
}

// flag[i] = 1 when the i-th entry in sorted order survives (first of its k-mer)
template <int W> __global__ void __launch_bounds__(TPB)
kc_flag_first(const u64 *__restrict__ keys, const uint32_t *__restrict__ perm, int64_t n,
              uint32_t *__restrict__ flag)
{ const int64_t i = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  bool first = i == 0;
  if (i > 0) first = !key_eq<W>(load_key<W>(keys, perm[i]), load_key<W>(keys, perm[i - 1]));
  flag[i] = first;
}

__global__ void __launch_bounds__(TPB)
kc_flag_trim(const uint16_t *__restrict__ cnt, int64_t n, unsigned ethresh, uint32_t *__restrict__ flag)
{ const int64_t i = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i < n) flag[i] = cnt[i] >= ethresh;
}

// out[pos[i]] = in[perm ? perm[i] : i] for the flagged i
template <int W> __global__ void __launch_bounds__(TPB)
kc_compact(const u64 *__restrict__ keys, const uint16_t *__restrict__ cnt, const uint32_t *__restrict__ perm,
           const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos, int64_t n,
           u64 *__restrict__ okeys, uint16_t *__restrict__ ocnt)
{ const int64_t i = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i >= n || !flag[i]) return;
  const int64_t src = perm ? (int64_t) perm[i] : i;
  const int64_t dst = pos[i];
#pragma unroll
  for (int w = 0; w < W; w++) okeys[dst * W + w] = keys[src * W + w];
  ocnt[dst] = cnt[src];
}

// flag -> positions + survivor count (exclusive scan); tmp is engine scratch
static int cond_scan(smg_engine *e, uint32_t *flag, uint32_t *pos, int64_t n, int64_t *kept,
                     char *errbuf, size_t errlen)
{ size_t tmp = 0;
  int rc;
  HIPCHK(rocprim::exclusive_scan(nullptr, tmp, flag, pos, 0u, (size_t) n, rocprim::plus<uint32_t>(), e->stream));
  if ((rc = grow((char **) &e->sort_tmp, &e->sort_tmp_cap, (int64_t) tmp + 16, errbuf, errlen))) return rc;
  HIPCHK(rocprim::exclusive_scan(e->sort_tmp, tmp, flag, pos, 0u, (size_t) n, rocprim::plus<uint32_t>(), e->stream));
  uint32_t lastp = 0, lastf = 0;
  HIPCHK(hipMemcpyAsync(&lastp, pos + (n - 1), 4, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(&lastf, flag + (n - 1), 4, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  *kept = (int64_t) lastp + lastf;
  return SMG_OK;
}

static int cond_sort_dedupe(smg_engine *e, const u64 *k2, const uint16_t *c2, const uint8_t *copy, int64_t n2,
                            int64_t *kept_out, char *errbuf, size_t errlen);

extern "C" int smg_engine_condition(smg_engine *e, int ethresh, int do_trim, int do_symm,
                                    int64_t *new_nels, char *errbuf, size_t errlen)
{ if (!e) return fail(errbuf, errlen, SMG_EINVAL, "null engine%s");
  if (!e->keys && e->n > 0) return fail(errbuf, errlen, SMG_EINVAL, "no table bound%s");
  HIPCHK(hipSetDevice(e->device));
  const int W = e->W;
  int rc;
  int64_t n = e->n;
  hipEvent_t c0, c1;
  hipEventCreate(&c0); hipEventCreate(&c1);
  hipEventRecord(c0, e->stream);
  uint32_t *flag = NULL, *pos = NULL, *perm = NULL, *perm2 = NULL;
  u64 *k2 = NULL, *wk = NULL, *wk2 = NULL, *ko = NULL;
  uint16_t *c2 = NULL, *co = NULL;
#define CFREE() { hipFree(flag); hipFree(pos); hipFree(perm); hipFree(perm2); hipFree(k2); hipFree(wk); \
                  hipFree(wk2); hipFree(c2); hipFree(ko); hipFree(co); hipEventDestroy(c0); hipEventDestroy(c1); }
#define CCHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) { CFREE(); \
                     return fail(errbuf, errlen, _e == hipErrorOutOfMemory ? SMG_ENOMEM : SMG_ENODEV, \
                                 "HIP error while conditioning: %s", hipGetErrorString(_e)); } } while (0)
#define CRC(call) do { if ((rc = (call))) { CFREE(); return rc; } } while (0)

  if (do_trim && n > 0)
    { const unsigned nblk = (unsigned) ((n + TPB - 1) / TPB);
      CCHK(hipMalloc(&flag, sizeof(uint32_t) * (size_t) n));
      CCHK(hipMalloc(&pos, sizeof(uint32_t) * (size_t) n));
      hipLaunchKernelGGL(kc_flag_trim, dim3(nblk), dim3(TPB), 0, e->stream, e->cnt, n, (unsigned) ethresh, flag);
      int64_t kept = 0;
      CRC(cond_scan(e, flag, pos, n, &kept, errbuf, errlen));
      CCHK(hipMalloc(&ko, sizeof(u64) * (size_t) (kept > 0 ? kept : 1) * W));
      CCHK(hipMalloc(&co, sizeof(uint16_t) * (size_t) (kept > 0 ? kept : 1) + 16));
#define CALL(WW) hipLaunchKernelGGL(kc_compact<WW>, dim3(nblk), dim3(TPB), 0, e->stream, e->keys, e->cnt, \
                   (const uint32_t *) NULL, flag, pos, n, ko, co)
      DISPATCH_W(e, CALL)
#undef CALL
      CCHK(hipStreamSynchronize(e->stream));
      hipFree(e->own_keys); hipFree(e->own_cnt);
      e->own_keys = ko; e->own_cnt = co; ko = NULL; co = NULL;
      e->keys = e->own_keys; e->cnt = e->own_cnt;
      n = kept;
      hipFree(flag); hipFree(pos); flag = pos = NULL;
    }

  if (do_symm && n > 0)
    { const int64_t n2 = 2 * n;
      if (n2 >= 0xFFFFFFF0ll) { CFREE(); return fail(errbuf, errlen, SMG_EINVAL, "table too large to symmetrise in one shard%s"); }
      const unsigned nblk = (unsigned) ((n + TPB - 1) / TPB);
      CCHK(hipMalloc(&k2, sizeof(u64) * (size_t) n2 * W));
      CCHK(hipMalloc(&c2, sizeof(uint16_t) * (size_t) n2));
#define CALL(WW) hipLaunchKernelGGL(kc_append_rc<WW>, dim3(nblk), dim3(TPB), 0, e->stream, e->keys, e->cnt, n, e->kmer, k2, c2)
      DISPATCH_W(e, CALL)
#undef CALL
      int64_t kept = 0;
      // (entries first, complements behind them: the stable sort keeps the entry in front of an equal complement)
      CRC(cond_sort_dedupe(e, k2, c2, NULL, n2, &kept, errbuf, errlen));
      n = kept;
    }
  hipEventRecord(c1, e->stream);
  CCHK(hipStreamSynchronize(e->stream));
  float ms = 0; hipEventElapsedTime(&ms, c0, c1);
  CFREE();
#undef CFREE
#undef CCHK
#undef CRC
  e->n = n;
  e->prepared = false; e->counted_done = false;
  e->have_ixdir = false; e->dir_preset = false; e->have_ends = false;      // (another table now: its index and ends are gone)
  e->rp_have = false; e->rp_active = false;
  e->st.nels = n;
  e->st.ms_decode += ms;
  if (new_nels) *new_nels = n;
  return SMG_OK;
}

// ---- symmetrising a table that is cut into prefix shards (several GPUs, or one device and more than 2^32 entries) -----
// The reference's Symmex has no size limit (PloidyPlot.c:1395-1414 hands it any table).  Here every shard
//   1. counts its own entries and their reverse complements per leading `bits` k-mer bits (smg_engine_symm_hist): the
//      sum over the shards is the shape of the CLOSED table, from which the caller takes balanced splitters -- a table
//      of canonical k-mers crowds the low end of the k-mer space (7/8 of the k-mers that start with an a are canonical,
//      1/8 of those that start with a t), its closure does not;
//   2. writes one record per entry and one per complement, grouped by the shard whose range holds the k-mer
//      (smg_engine_symm_route; records of W + 1 words: the k-mer, then count | is-a-complement << 16);
//   3. after the exchange sorts what it received and keeps one entry per k-mer (smg_engine_symm_finish; a k-mer that
//      arrives as an entry AND as a complement keeps the entry's count -- self-complementary k-mers, or input that was
//      neither canonical nor closed: the same rule as the single-shard code above).

#define SY_MAXBITS 12

template <int W> __global__ void __launch_bounds__(TPB)
kc_symm_hist(const u64 *__restrict__ keys, int64_t n, int k, int bits, u64 *__restrict__ hist /* [2 << bits]: own, complements */)
{ __shared__ unsigned h[2 << SY_MAXBITS];
  const int nb = 1 << bits;
  for (int b = threadIdx.x; b < 2 * nb; b += TPB) h[b] = 0;
  __syncthreads();
  for (int64_t i = (int64_t) blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t) gridDim.x * TPB)
    { const Key<W> x = load_key<W>(keys, i);
      const Key<W> r = revcomp<W>(x, k);
      atomicAdd(&h[(unsigned) (x.w[0] >> (64 - bits))], 1u);
      atomicAdd(&h[nb + (unsigned) (r.w[0] >> (64 - bits))], 1u);
    }
  __syncthreads();
  for (int b = threadIdx.x; b < 2 * nb; b += TPB)
    if (h[b]) atomicAdd(&hist[b], (u64) h[b]);
}

// record i = (entry i, count), record n + i = (its reverse complement, count | 1 << 16)
template <int W> __global__ void __launch_bounds__(TPB)
kc_symm_records(const u64 *__restrict__ keys, const uint16_t *__restrict__ cnt, int64_t n, int k, u64 *__restrict__ rec)
{ const int64_t i = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const Key<W> x = load_key<W>(keys, i);
  const Key<W> r = revcomp<W>(x, k);
  u64 *a = rec + (size_t) i * (W + 1), *b = rec + (size_t) (n + i) * (W + 1);
#pragma unroll
  for (int w = 0; w < W; w++) { a[w] = x.w[w]; b[w] = r.w[w]; }
  a[W] = (u64) cnt[i];
  b[W] = (u64) cnt[i] | (1ull << 16);
}

__global__ void __launch_bounds__(TPB) kc_fill_u32(uint32_t *__restrict__ p, int64_t n, uint32_t v, uint32_t last)
{ const int64_t i = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i < n) p[i] = i == n - 1 ? last : v;
}

// ---- the counted path (k > 85) in the steps of the phase API ---------------------------------------------------
// Pass 1 = k_pass1<W, true>: S_all into deg[], records (rc(x), count | S_hi << 16) from the owners of a hi-side pair (from
// every entry for the exact proof) into one flat list, which the router reads as full chunks of F_CH records; a received
// record adds S_hi to the degree of its k-mer (the uint8 wrap of PloidyPlot.c:163 emulated by deg_add) and proves that
// the k-mer is there with the same count; pass 2 = k_pass2<W, true>.
static int counted_phase_pass1(smg_engine *e, int symcheck, char *errbuf, size_t errlen)
{ int rc;
  if ((rc = counted_prepare(e, errbuf, errlen))) return rc;
  const int emit_all = symcheck == SMG_SYM_EXACT;
  const unsigned nblk = (unsigned) ((e->n + TPB - 1) / TPB);
  int64_t cap = (emit_all ? e->n : e->n / 4) + 1024;
  e->rw = e->W + 1;
  for (int attempt = 0; attempt < 2; attempt++)
    { if ((rc = grow(&e->req, &e->req_cap, cap * (int64_t) sizeof(u64) * e->rw, errbuf, errlen))) return rc;
      Tab t = make_tab(e);
      hipEventRecord(e->ev[2], e->stream);
      if (e->n > 0)
        {
#define CALL(WW) hipLaunchKernelGGL((k_pass1<WW, true>), dim3(nblk), dim3(TPB), 0, e->stream, t, \
                   (int64_t) 0, e->n, emit_all, symcheck == SMG_SYM_HASH, e->req, cap, e->ctrl)
          DISPATCH_W(e, CALL)
#undef CALL
        }
      hipEventRecord(e->ev[3], e->stream);
      if ((rc = read_ctrl(e, errbuf, errlen))) return rc;
      if (e->h_ctrl->unsorted)
        return fail(errbuf, errlen, SMG_EFORMAT, "table entries are not strictly increasing%s");
      if ((int64_t) e->h_ctrl->nreq <= cap) break;
      cap = (int64_t) e->h_ctrl->nreq;
      HIPCHK(hipMemsetAsync(&e->ctrl->nreq, 0, sizeof(u64), e->stream));
      HIPCHK(hipMemsetAsync(e->ctrl->fp, 0, sizeof(u64) * 4, e->stream));
    }
  float ms = 0; hipEventElapsedTime(&ms, e->ev[2], e->ev[3]);
  e->st.ms_pass1 = ms;
  const int64_t nreq = (int64_t) e->h_ctrl->nreq;
  e->st.nrequests = nreq;
  for (int i = 0; i < 4; i++) e->fp[i] = e->h_ctrl->fp[i];
  const int64_t nc = (nreq + F_CH - 1) / F_CH;
  if (nc >= 0x7FFFFFFFll) return fail(errbuf, errlen, SMG_EINVAL, "shard too large%s");
  if ((rc = grow(&e->chunk_fill, &e->chunk_cap, nc * 4 + 4, errbuf, errlen))) return rc;
  if (nc > 0)
    hipLaunchKernelGGL(kc_fill_u32, dim3((unsigned) ((nc + TPB - 1) / TPB)), dim3(TPB), 0, e->stream, e->chunk_fill, nc, (uint32_t) F_CH,
                       (uint32_t) (nreq - (nc - 1) * F_CH));
  HIPCHK(hipGetLastError());
  e->n_chunks = (unsigned) nc; e->bm_bits = 0; e->filtered = false; e->presorted = false;
  e->st.path = 1; e->st.ms_filter = 0;
  e->prepared = true;                     // (with e->fast == false: the counted steps)
  return SMG_OK;
}

static int counted_phase_apply(smg_engine *e, const u64 *rec, int64_t nrec, int64_t *missing, char *errbuf, size_t errlen)
{ hipEventRecord(e->ev[4], e->stream);
  if (nrec > 0)
    { Tab t = make_tab(e);
      const unsigned rb = (unsigned) ((nrec + TPB - 1) / TPB);
#define CALL(WW) hipLaunchKernelGGL(k_apply<WW>, dim3(rb), dim3(TPB), 0, e->stream, t, rec, nrec, e->ctrl)
      DISPATCH_W(e, CALL)
#undef CALL
    }
  hipEventRecord(e->ev[5], e->stream);
  HIPCHK(hipGetLastError());
  if (!missing) { e->lookup_pending = true; return SMG_OK; }
  const int rc = read_ctrl(e, errbuf, errlen);
  if (rc) return rc;
  float ms = 0; hipEventElapsedTime(&ms, e->ev[4], e->ev[5]);
  e->st.ms_rclookup += ms;
  *missing = (int64_t) e->h_ctrl->missing;
  return SMG_OK;
}

static int counted_phase_pass2(smg_engine *e, int64_t *d_plot, char *errbuf, size_t errlen)
{ HIPCHK(hipMemsetAsync(d_plot, 0, sizeof(int64_t) * SMG_PLOT_CELLS, e->stream));
  hipEventRecord(e->ev[6], e->stream);
  if (e->n > 0)
    { Tab t = make_tab(e);
      const unsigned nblk = (unsigned) ((e->n + TPB - 1) / TPB);
#define CALL(WW) hipLaunchKernelGGL((k_pass2<WW, true>), dim3(nblk), dim3(TPB), 0, e->stream, t, \
                   (int64_t) 0, e->n, (u64 *) d_plot)
      DISPATCH_W(e, CALL)
#undef CALL
    }
  hipEventRecord(e->ev[7], e->stream);
  HIPCHK(hipGetLastError());
  e->st.path = 1; e->st.npairs = 0; e->st.ms_pass2 = -1.0;      // (no host wait: smg_engine_stats resolves the events)
  e->counted_done = true;
  return SMG_OK;
}

// records -> separate k-mer / count / is-a-complement arrays
template <int W> __global__ void __launch_bounds__(TPB)
kc_symm_unpack(const u64 *__restrict__ rec, int64_t n, u64 *__restrict__ keys, uint16_t *__restrict__ cnt, uint8_t *__restrict__ copy)
{ const int64_t i = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const u64 *q = rec + (size_t) i * (W + 1);
#pragma unroll
  for (int w = 0; w < W; w++) keys[i * W + w] = q[w];
  cnt[i] = (uint16_t) q[W];
  copy[i] = (uint8_t) ((q[W] >> 16) & 1u);
}

// like kc_compact, for a sorted permutation in which a k-mer occurs at most twice (once as an entry, once as a
// complement): the survivor is the first of the two, its count the ENTRY's
template <int W> __global__ void __launch_bounds__(TPB)
kc_compact_pref(const u64 *__restrict__ keys, const uint16_t *__restrict__ cnt, const uint8_t *__restrict__ copy,
                const uint32_t *__restrict__ perm, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos,
                int64_t n, u64 *__restrict__ okeys, uint16_t *__restrict__ ocnt)
{ const int64_t i = (int64_t) blockIdx.x * TPB + threadIdx.x;
  if (i >= n || !flag[i]) return;
  const int64_t src = perm[i];
  int64_t csrc = src;
  if (copy[src] && i + 1 < n && !flag[i + 1]) csrc = perm[i + 1];        // (the next one is the same k-mer: the entry)
  const int64_t dst = pos[i];
#pragma unroll
  for (int w = 0; w < W; w++) okeys[dst * W + w] = keys[src * W + w];
  ocnt[dst] = cnt[csrc];
}

extern "C" int smg_engine_table(smg_engine *e, int64_t *nels, const uint64_t **d_keys, const uint16_t **d_counts)
{ if (!e) return SMG_EINVAL;
  if (nels) *nels = e->n;
  if (d_keys) *d_keys = (const uint64_t *) e->keys;
  if (d_counts) *d_counts = e->cnt;
  return SMG_OK;
}

extern "C" int smg_engine_symm_hist(smg_engine *e, int bits, int64_t *hist, char *errbuf, size_t errlen)
{ if (!e || !hist) return fail(errbuf, errlen, SMG_EINVAL, "null argument%s");
  if (bits < 1 || bits > SY_MAXBITS || bits > 2 * e->kmer) return fail(errbuf, errlen, SMG_EINVAL, "symm_hist: 1..12 leading bits, at most 2k%s");
  if (!e->keys && e->n > 0) return fail(errbuf, errlen, SMG_EINVAL, "no table bound%s");
  HIPCHK(hipSetDevice(e->device));
  const size_t bytes = sizeof(u64) * ((size_t) 2 << bits);
  u64 *d = NULL;
  HIPCHK(hipMalloc(&d, bytes));
  hipError_t he = hipMemsetAsync(d, 0, bytes, e->stream);
  if (he == hipSuccess && e->n > 0)
    { int64_t nb = (e->n + TPB - 1) / TPB;
      if (nb > 2048) nb = 2048;
#define CALL(WW) hipLaunchKernelGGL(kc_symm_hist<WW>, dim3((unsigned) nb), dim3(TPB), 0, e->stream, e->keys, e->n, e->kmer, bits, d)
      DISPATCH_W(e, CALL)
#undef CALL
      he = hipGetLastError();
    }
  if (he == hipSuccess) he = hipMemcpyAsync(hist, d, bytes, hipMemcpyDeviceToHost, e->stream);
  if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
  hipFree(d);
  if (he != hipSuccess) return fail(errbuf, errlen, SMG_ENODEV, "symm_hist: %s", hipGetErrorString(he));
  return SMG_OK;
}

extern "C" int smg_engine_symm_route(smg_engine *e, const uint64_t *splitters, int nranks, uint64_t *d_send,
                                     int64_t capacity, int64_t *counts, char *errbuf, size_t errlen)
{ if (!e || !counts || nranks < 1 || nranks > 16) return fail(errbuf, errlen, SMG_EINVAL, "bad symm_route arguments (1..16 ranks)%s");
  if (!e->keys && e->n > 0) return fail(errbuf, errlen, SMG_EINVAL, "no table bound%s");
  const int64_t n2 = 2 * e->n;
  if (capacity < n2 || (n2 > 0 && !d_send)) return fail(errbuf, errlen, SMG_EINVAL, "symm_route: the send buffer must hold two records per entry%s");
  for (int r = 0; r < nranks; r++) counts[r] = 0;
  if (n2 == 0) return SMG_OK;
  HIPCHK(hipSetDevice(e->device));
  const int rw = e->W + 1;
  const int64_t nc = (n2 + F_CH - 1) / F_CH;
  if (nc >= 0x7FFFFFFFll) return fail(errbuf, errlen, SMG_EINVAL, "symm_route: shard too large%s");
  u64 *rec = NULL; uint32_t *fill = NULL;
  int rc = SMG_OK;
  if (hipMalloc(&rec, sizeof(u64) * (size_t) n2 * rw) != hipSuccess || hipMalloc(&fill, sizeof(uint32_t) * (size_t) nc) != hipSuccess)
    { hipFree(rec); hipFree(fill); return fail(errbuf, errlen, SMG_ENOMEM, "out of device memory while symmetrising%s"); }
  const unsigned nblk = (unsigned) ((e->n + TPB - 1) / TPB);
#define CALL(WW) hipLaunchKernelGGL(kc_symm_records<WW>, dim3(nblk), dim3(TPB), 0, e->stream, e->keys, e->cnt, e->n, e->kmer, rec)
  DISPATCH_W(e, CALL)
#undef CALL
  hipLaunchKernelGGL(kc_fill_u32, dim3((unsigned) ((nc + TPB - 1) / TPB)), dim3(TPB), 0, e->stream, fill, nc, (uint32_t) F_CH,
                     (uint32_t) (n2 - (nc - 1) * F_CH));
  if (hipGetLastError() != hipSuccess) rc = fail(errbuf, errlen, SMG_ENODEV, "symm_route: launch failed%s");
  if (rc == SMG_OK) rc = route_records(e, rec, fill, (unsigned) nc, rw, splitters, nranks, d_send, capacity, counts, errbuf, errlen);
  hipStreamSynchronize(e->stream);
  hipFree(rec); hipFree(fill);
  return rc;
}

// sorted, duplicate-free table from 2-copies-at-most material: keys[n2 * W], counts, copy flags (NULL: the first of two
// equal k-mers in INPUT order wins, as the stable sort leaves it) -> the engine's own table.  Frees nothing of the caller's.
static int cond_sort_dedupe(smg_engine *e, const u64 *k2, const uint16_t *c2, const uint8_t *copy, int64_t n2,
                            int64_t *kept_out, char *errbuf, size_t errlen)
{ const int W = e->W;
  int rc = SMG_OK;
  uint32_t *flag = NULL, *pos = NULL, *perm = NULL, *perm2 = NULL;
  u64 *wk = NULL, *wk2 = NULL, *ko = NULL;
  uint16_t *co = NULL;
  int64_t kept = 0;
  if (n2 >= 0xFFFFFFF0ll) return fail(errbuf, errlen, SMG_EINVAL, "shard too large to symmetrise (2^32 entries per shard)%s");
  const unsigned nblk2 = (unsigned) ((n2 + TPB - 1) / TPB);
#define SFREE() { hipFree(flag); hipFree(pos); hipFree(perm); hipFree(perm2); hipFree(wk); hipFree(wk2); hipFree(ko); hipFree(co); }
#define SCHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) { SFREE(); \
                     return fail(errbuf, errlen, _e == hipErrorOutOfMemory ? SMG_ENOMEM : SMG_ENODEV, \
                                 "HIP error while conditioning: %s", hipGetErrorString(_e)); } } while (0)
#define SRC(call) do { if ((rc = (call))) { SFREE(); return rc; } } while (0)
  SCHK(hipMalloc(&perm, sizeof(uint32_t) * (size_t) n2));
  SCHK(hipMalloc(&perm2, sizeof(uint32_t) * (size_t) n2));
  SCHK(hipMalloc(&wk, sizeof(u64) * (size_t) n2));
  SCHK(hipMalloc(&wk2, sizeof(u64) * (size_t) n2));
  hipLaunchKernelGGL(kc_iota, dim3(nblk2), dim3(TPB), 0, e->stream, perm, n2);
  for (int w = W - 1; w >= 0; w--)              // LSD over the words, stable
    { hipLaunchKernelGGL(kc_gather_word, dim3(nblk2), dim3(TPB), 0, e->stream, k2, perm, W, w, n2, wk);
      size_t tmp = 0;
      SCHK(rocprim::radix_sort_pairs<smg_pair_sort_config>(nullptr, tmp, wk, wk2, perm, perm2, (size_t) n2, 0u, 64u, e->stream));
      SRC(grow((char **) &e->sort_tmp, &e->sort_tmp_cap, (int64_t) tmp + 16, errbuf, errlen));
      SCHK(rocprim::radix_sort_pairs<smg_pair_sort_config>(e->sort_tmp, tmp, wk, wk2, perm, perm2, (size_t) n2, 0u, 64u, e->stream));
      uint32_t *sw = perm; perm = perm2; perm2 = sw;
    }
  hipFree(wk); hipFree(wk2); hipFree(perm2); wk = wk2 = NULL; perm2 = NULL;
  SCHK(hipMalloc(&flag, sizeof(uint32_t) * (size_t) n2));
  SCHK(hipMalloc(&pos, sizeof(uint32_t) * (size_t) n2));
#define CALL(WW) hipLaunchKernelGGL(kc_flag_first<WW>, dim3(nblk2), dim3(TPB), 0, e->stream, k2, perm, n2, flag)
  DISPATCH_W(e, CALL)
#undef CALL
  SRC(cond_scan(e, flag, pos, n2, &kept, errbuf, errlen));
  SCHK(hipMalloc(&ko, sizeof(u64) * (size_t) (kept > 0 ? kept : 1) * W));
  SCHK(hipMalloc(&co, sizeof(uint16_t) * (size_t) (kept > 0 ? kept : 1) + 16));
  if (copy)
    {
#define CALL(WW) hipLaunchKernelGGL(kc_compact_pref<WW>, dim3(nblk2), dim3(TPB), 0, e->stream, k2, c2, copy, perm, flag, pos, n2, ko, co)
      DISPATCH_W(e, CALL)
#undef CALL
    }
  else
    {
#define CALL(WW) hipLaunchKernelGGL(kc_compact<WW>, dim3(nblk2), dim3(TPB), 0, e->stream, k2, c2, perm, flag, pos, n2, ko, co)
      DISPATCH_W(e, CALL)
#undef CALL
    }
  SCHK(hipStreamSynchronize(e->stream));
  hipFree(e->own_keys); hipFree(e->own_cnt);
  e->own_keys = ko; e->own_cnt = co; ko = NULL; co = NULL;
  e->keys = e->own_keys; e->cnt = e->own_cnt;
  SFREE();
#undef SFREE
#undef SCHK
#undef SRC
  *kept_out = kept;
  return SMG_OK;
}

extern "C" int smg_engine_symm_finish(smg_engine *e, const uint64_t *d_recv, int64_t nrecv, int64_t *new_nels,
                                      char *errbuf, size_t errlen)
{ if (!e || nrecv < 0 || (nrecv > 0 && !d_recv)) return fail(errbuf, errlen, SMG_EINVAL, "bad symm_finish arguments%s");
  if (e->kmer < 1) return fail(errbuf, errlen, SMG_EINVAL, "no table bound%s");
  HIPCHK(hipSetDevice(e->device));
  int64_t kept = 0;
  if (nrecv == 0)
    { hipFree(e->own_keys); hipFree(e->own_cnt); e->own_keys = NULL; e->own_cnt = NULL;
      HIPCHK(hipMalloc(&e->own_keys, sizeof(u64) * e->W)); HIPCHK(hipMalloc(&e->own_cnt, 16));
      e->keys = e->own_keys; e->cnt = e->own_cnt;
    }
  else
    { if (nrecv >= 0xFFFFFFF0ll) return fail(errbuf, errlen, SMG_EINVAL, "shard too large to symmetrise (2^32 entries per shard)%s");
      u64 *k2 = NULL; uint16_t *c2 = NULL; uint8_t *cp = NULL;
      if (hipMalloc(&k2, sizeof(u64) * (size_t) nrecv * e->W) != hipSuccess || hipMalloc(&c2, sizeof(uint16_t) * (size_t) nrecv + 16) != hipSuccess
          || hipMalloc(&cp, (size_t) nrecv + 16) != hipSuccess)
        { hipFree(k2); hipFree(c2); hipFree(cp); return fail(errbuf, errlen, SMG_ENOMEM, "out of device memory while symmetrising%s"); }
      const unsigned nblk = (unsigned) ((nrecv + TPB - 1) / TPB);
#define CALL(WW) hipLaunchKernelGGL(kc_symm_unpack<WW>, dim3(nblk), dim3(TPB), 0, e->stream, (const u64 *) d_recv, nrecv, k2, c2, cp)
      DISPATCH_W(e, CALL)
#undef CALL
      const int rc = cond_sort_dedupe(e, k2, c2, cp, nrecv, &kept, errbuf, errlen);
      hipStreamSynchronize(e->stream);
      hipFree(k2); hipFree(c2); hipFree(cp);
      if (rc) return rc;
    }
  e->n = kept;
  e->prepared = false; e->counted_done = false;
  e->have_ixdir = false; e->dir_preset = false; e->have_ends = false;
  e->st.nels = kept;
  if (new_nels) *new_nels = kept;
  return SMG_OK;
}

// ---- extract: the unique pairs of the labelled pixels (next row of the scope table: extract_kmer_pairs) ------

extern "C" int smg_engine_extract(smg_engine *e, const uint16_t *d_labels, uint64_t *d_out, int64_t capacity,
                                  int64_t *nrec, char *errbuf, size_t errlen)
{ if (!e || !d_labels || !nrec) return fail(errbuf, errlen, SMG_EINVAL, "null argument%s");
  const bool counted = e->counted_done && !e->fast && e->st.path == 1;        // k > 85: degrees instead of code bytes
  if (!counted && (!e->prepared || !e->fast || e->st.path != 1))
    return fail(errbuf, errlen, SMG_EINVAL,
                "extract needs a completed run on a conditioned (reverse-complement closed) table%s");
  HIPCHK(hipSetDevice(e->device));
  u64 *d_total = &e->ctrl->plot_sum;
  HIPCHK(hipMemsetAsync(d_total, 0, sizeof(u64), e->stream));
  if (e->n > 0 && counted)
    { Tab t = make_tab(e);
      const unsigned nblk = (unsigned) ((e->n + TPB - 1) / TPB);
#define CALL(WW) hipLaunchKernelGGL(k_extract<WW>, dim3(nblk), dim3(TPB), 0, e->stream, t, d_labels, (u64 *) d_out, \
                   (u64) (d_out ? capacity : 0), d_total)
      DISPATCH_W(e, CALL)
#undef CALL
    }
  else if (e->n > 0)
    { FastArgs a = make_fast(e);
      int64_t nb = (e->n + F_TPB - 1) / F_TPB;
      if (nb > 2048) nb = 2048;
#define CALL(WW) hipLaunchKernelGGL(kf_extract<WW>, dim3((unsigned) nb), dim3(F_TPB), 0, e->stream, a, d_labels, \
                   (u64 *) d_out, (u64) (d_out ? capacity : 0), d_total)
      DISPATCH_W3(e, CALL)
#undef CALL
    }
  HIPCHK(hipGetLastError());
  int rc = read_ctrl(e, errbuf, errlen);
  if (rc) return rc;
  *nrec = (int64_t) e->h_ctrl->plot_sum;
  return SMG_OK;
}

#include "smg_ingest.hpp"
#include "smg_multi.hpp"

// ---- one-shot host entry ----------------------------------------------------------------------

// labels == NULL: hetmers.  labels != NULL: additionally the extract leg; *records receives a malloc'ed
// array of *nrec records of (words + 1) uint64 each.
static int host_run(const smg_table_source *tv, const smg_opts *opts, int64_t *plot, smg_stats *stats,
                    const uint16_t *labels, uint64_t **records, int64_t *nrec, int *rec_words,
                    char *errbuf, size_t errlen)
{ if (!tv || !plot || !tv->read) return fail(errbuf, errlen, SMG_EINVAL, "null argument%s");
  struct timespec w0, w1, w2;
  clock_gettime(CLOCK_MONOTONIC, &w0);       // (the first HIP call of the process -- the free-memory query below -- starts the runtime)
  const int device = opts ? opts->device : 0;
  const int symcheck = opts ? opts->symcheck : SMG_SYM_EXACT;
  const int verbose = opts ? opts->verbose : 0;
  if (tv->ibyte < 1 || tv->ibyte > 3) return fail(errbuf, errlen, SMG_EINVAL, "ibyte must be 1, 2 or 3%s");
  const int kbyte = (tv->kmer + 3) >> 2, pbyte = kbyte + 2 - tv->ibyte;
  if (pbyte < 3) return fail(errbuf, errlen, SMG_EINVAL, "k-mer shorter than the index prefix%s");
  int64_t sum = 0;
  for (int p = 0; p < tv->nparts; p++) sum += tv->part_nels[p];
  if (sum != tv->nels) return fail(errbuf, errlen, SMG_EFORMAT, "part sizes do not add up to nels%s");
  { // several GPUs of the node (smg_multi.hpp); SMG_VIRTUAL_SHARDS is the 1-GPU test hook of that path
    int ng = opts ? opts->ngpus : 0;
    bool virt = false;
    const char *v = getenv("SMG_VIRTUAL_SHARDS");
    if (v && atoi(v) > 1) { ng = atoi(v); virt = true; }
    // A shard addresses its entries with 32 bits.  A table beyond that is cut into prefix shards that all live on
    // this one device (288 GB of HBM hold ~1.5e10 k=31 entries with their side arrays) -- the same code path as
    // several GPUs, with device-to-device copies for the exchange: the reference streams a table of any size
    // (PloidyPlot.c:931-1038), this engine must not refuse one just because of an index width.
    // SMG_SHARD_LIMIT (tests) lowers the threshold.
    { int64_t limit = 0xFFFFFFF0ll - 16;
      const char *sl = getenv("SMG_SHARD_LIMIT");
      if (sl && atoll(sl) > 0) limit = atoll(sl);
      // (a table that still has to be symmetrised closes to at most twice its entries: shard it for that size)
      const bool want_symm = opts && (opts->condition & SMG_COND_SYMM);
      const int64_t closed = want_symm ? 2 * tv->nels : tv->nels;
      if (ng <= 1 && closed >= limit)
        { int64_t per = limit > 3000000000ll ? 3000000000ll : limit;     // ~3e9 entries per shard
          if (per < 1) per = 1;
          ng = (int) ((closed + per - 1) / per);
          if (ng < 2) ng = 2;
          if (ng > SMG_MAXGPU)
            return fail(errbuf, errlen, SMG_EINVAL, "table too large for one GPU (more than 16 shards of 3e9 entries): use SMUDGEPLOT_GPUS%s");
          virt = true;
          if (verbose) fprintf(stderr, "  [smg] %lld entries%s: %d prefix shards on one device\n", (long long) tv->nels,
                               want_symm ? " before the table is symmetrised" : "", ng);
        }
    }
    // Out of core (smg_multi.hpp, host_run_sequential): a table whose shards do not fit the device TOGETHER -- keys, counts and
    // the lists of a run take ~8 W + 12 bytes per entry -- is run shard after shard, the table read twice; what stays between
    // the two rounds is a code byte per entry and the requests.  SMG_HBM_LIMIT=<bytes> (tests) stands in for the free device
    // memory, SMG_SEQUENTIAL_SHARDS=<n> forces the mode.
    if (ng <= 1 || virt)
      { const int W = (tv->kmer + 31) / 32;
        int seq = 0;
        { const char *sv = getenv("SMG_SEQUENTIAL_SHARDS"); if (sv && atoi(sv) > 1) seq = atoi(sv); }
        if (!seq)
          { size_t fr = 0, tot = 0;
            double limit = 0;
            { const char *hl = getenv("SMG_HBM_LIMIT"); if (hl && atof(hl) > 0) limit = atof(hl); }
            if (limit <= 0 && hipSetDevice(device) == hipSuccess && hipMemGetInfo(&fr, &tot) == hipSuccess) limit = 0.9 * (double) fr;
            // what a run in core takes per entry: k-mers 8 W, count 2, code byte 1, deferred-entry map and lists 0.6, the request
            // list (a quarter of the entries + slack) and its partitioned copy ~2.6 W -- and the candidate map (1 GiB at most)
            // (a raw table: the size of the CLOSED table counts -- at most twice the entries -- and conditioning a shard takes
            //  more than running it: its records (2 (W + 1) words per entry of the piece), the sort's copies and scratch)
            const bool raw = opts && opts->condition, want_symm = opts && (opts->condition & SMG_COND_SYMM);
            const double ne = (double) tv->nels * (want_symm ? 2.0 : 1.0);
            const double per_run = 10.6 * W + 5.5, per_cond = raw ? (want_symm ? 24.0 * W + 30.0 : 8.0 * W + 12.0) : 0.0;
            const double all = ne * (per_run > per_cond ? per_run : per_cond) + 1.1e9;
            if (limit > 0 && all > limit)
              { // what a shard leaves behind: its code bytes and its requests -- every entry's (W + 1 words) under the exact
                // proof, those of the owners of a pair at p > k-1-p otherwise (17 % of a diploid table, 36 % of a polyploid one)
                const bool exact = symcheck == SMG_SYM_EXACT;
                const double keep = ne * (1.0 + (exact ? 8.0 * (W + 1) : 0.36 * 8.0 * (tv->kmer > FAST_MAX_K ? W + 1 : W)));     // (k > 85: records carry a count word)
                for (int q = 2; q <= SMG_MAXGPU && !seq; q++)
                  if (keep + ne / q * per_run <= limit && ne / q * per_cond <= limit && ne / q < 0xFFFFFFF0ll - 16) seq = q;     // (no map out of core)
                if (!seq)
                  return fail(errbuf, errlen, SMG_ENOMEM, "the table does not fit the device even shard by shard (16 shards, a code byte and "
                              "the requests of every entry resident): use SMUDGEPLOT_GPUS%s");
              }
          }
        if (seq)
          { if (verbose) fprintf(stderr, "  [smg] %lld entries do not fit the device together: %d prefix shards one after the other\n",
                                 (long long) tv->nels, seq);
            smg_opts o; memset(&o, 0, sizeof(o));
            if (opts) o = *opts; else o.symcheck = SMG_SYM_HASH;
            return host_run_sequential(tv, &o, seq, plot, stats, errbuf, errlen, labels, records, nrec, rec_words);
          }
      }
    // SMG_FORCE_MULTI=1 (tests): take the multi-GPU code path even with one GPU -- a one-rank RCCL
    // communicator, send/recv to self, all-reduce: checks the dlopen'ed RCCL entry points on a 1-GPU box
    if (ng <= 1 && getenv("SMG_FORCE_MULTI") && !v) ng = -1;
    if (ng > 1 || ng == -1)
      { { smg_opts o; memset(&o, 0, sizeof(o));
            if (opts) o = *opts; else o.symcheck = SMG_SYM_HASH;
            const int mrc = host_run_multi(tv, &o, ng == -1 ? 1 : ng, virt, plot, stats, errbuf, errlen, labels, records, nrec, rec_words);
            // a table that fails the symmetry proof: shards on ONE device run the general path together (smg_multi.hpp,
            // TabSet); several real GPUs hand the table to one of them, which can hold it as long as it has < 2^32 entries
            if (mrc != SMG_ENOTSYM) return mrc;
            if (tv->nels >= 0xFFFFFFF0ll - 16)
              return fail(errbuf, errlen, SMG_ENOTSYM, "the table is not closed under reverse complement and has more than 2^32 entries: "
                          "the general path for such a table runs on one device (unset SMUDGEPLOT_GPUS), or condition the table%s");
            if (verbose) fprintf(stderr, "  [smg] the table is not closed under reverse complement: general path on one GPU\n");
          }
      }
  }

  smg_engine *e = smg_engine_create(device, NULL, errbuf, errlen);
  if (!e) return SMG_ENODEV;
  clock_gettime(CLOCK_MONOTONIC, &w1);       // (the first HIP call of the process: runtime start-up + code object load)
  int rc = SMG_OK;
  int64_t *d_index = NULL, *d_plot = NULL;
  uint16_t *d_labels = NULL; uint64_t *d_out = NULL;
  const size_t ixbytes = sizeof(int64_t) << (8 * tv->ibyte);
  double h2d_s = 0, alloc_s = 0, cond_s = 0, run_s = 0, out_s = 0;
#define BAIL(code, msg) { rc = fail(errbuf, errlen, code, msg "%s"); goto done; }
#define SECS(a, b) ((double) ((b).tv_sec - (a).tv_sec) + 1e-9 * (double) ((b).tv_nsec - (a).tv_nsec))
  if (hipMalloc(&d_index, ixbytes) != hipSuccess || hipMalloc(&d_plot, sizeof(int64_t) * SMG_PLOT_CELLS) != hipSuccess)
    BAIL(SMG_ENOMEM, "out of device memory for the table")
  if ((rc = decode_begin(e, tv->kmer, tv->ibyte, tv->nels, errbuf, errlen))) goto done;
  { // the index goes first (asynchronously from pageable memory: staged by the runtime); the records stream behind it,
    // piece by piece through a pinned ring, and every piece is decoded on the device as soon as its copy has landed
    if (hipMemcpyAsync(d_index, tv->prefix_index, ixbytes, hipMemcpyHostToDevice, e->stream) != hipSuccess
        || hipEventRecord(e->ev[0], e->stream) != hipSuccess)
      BAIL(SMG_ENODEV, "host to device copy failed")
    clock_gettime(CLOCK_MONOTONIC, &w2);
    alloc_s = SECS(w1, w2);
    DecodeHook hk; hk.e = e; hk.d_index = d_index; hk.ibyte = tv->ibyte; hk.ibase = 0;
    if ((rc = ingest_records(tv, pbyte, 0, tv->nels, NULL, device, tv->host_threads, &h2d_s, errbuf, errlen, decode_hook, &hk, e->ev[0]))) goto done;
    if (hipStreamSynchronize(e->stream) != hipSuccess) BAIL(SMG_ENODEV, "host to device copy failed")
  }
  clock_gettime(CLOCK_MONOTONIC, &w1);
  if (opts && opts->condition)
    { if ((rc = smg_engine_condition(e, opts->ethresh, opts->condition & SMG_COND_TRIM, opts->condition & SMG_COND_SYMM,
                                     NULL, errbuf, errlen))) goto done;
    }
  else if ((rc = smg_engine_set_prefix_index(e, d_index, tv->ibyte, 0, errbuf, errlen))) goto done;   // the table's index = its directory
  clock_gettime(CLOCK_MONOTONIC, &w2); cond_s = SECS(w1, w2);
  if ((rc = smg_engine_run(e, symcheck, d_plot, NULL, errbuf, errlen))) goto done;
  clock_gettime(CLOCK_MONOTONIC, &w1); run_s = SECS(w2, w1);
  if (hipMemcpy(plot, d_plot, sizeof(int64_t) * SMG_PLOT_CELLS, hipMemcpyDeviceToHost) != hipSuccess)
    BAIL(SMG_ENODEV, "device to host copy failed")
  if (labels)
    { // exact record count = plot weight on the labelled pixels (the plot counts a mirrored pair twice)
      if (e->st.path != 1)
        BAIL(SMG_EINVAL, "extract needs a trimmed, reverse-complement closed table")
      int64_t want = 0, got = 0;
      for (int c = 0; c < SMG_PLOT_CELLS; c++) if (labels[c]) want += plot[c];
      const int rw = e->W + 1;
      if (hipMalloc(&d_labels, sizeof(uint16_t) * SMG_PLOT_CELLS) != hipSuccess
          || hipMalloc(&d_out, sizeof(uint64_t) * (size_t) (want > 0 ? want : 1) * rw) != hipSuccess)
        BAIL(SMG_ENOMEM, "out of device memory for the pair list")
      if (hipMemcpy(d_labels, labels, sizeof(uint16_t) * SMG_PLOT_CELLS, hipMemcpyHostToDevice) != hipSuccess)
        BAIL(SMG_ENODEV, "host to device copy failed")
      if ((rc = smg_engine_extract(e, d_labels, d_out, want, &got, errbuf, errlen))) goto done;
      if (got != want) BAIL(SMG_ENODEV, "internal error: pair list and plot disagree")
      uint64_t *h = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) (want > 0 ? want : 1) * rw);
      if (!h) BAIL(SMG_ENOMEM, "out of host memory for the pair list")
      if (want && hipMemcpy(h, d_out, sizeof(uint64_t) * (size_t) want * rw, hipMemcpyDeviceToHost) != hipSuccess)
        { free(h); BAIL(SMG_ENODEV, "device to host copy failed") }
      *records = h; *nrec = want; *rec_words = rw;
    }
  clock_gettime(CLOCK_MONOTONIC, &w2); out_s = SECS(w1, w2);
  e->st.ms_h2d = h2d_s * 1e3;
  if (stats) *stats = e->st;
  if (verbose)
    { fprintf(stderr, "  [smg] n=%lld k=%d path=%s  read+h2d+decode %.2f ms (%.1f GB/s, %d readers; the decode of a piece runs behind its copy), "
              "pass1 %.2f, rc-lookup %.2f, pass2 %.2f, total(device) %.2f ms => %.3g k-mers/s\n",
              (long long) e->st.nels, tv->kmer, e->st.path == 1 ? "rc-half-scan" : "general",
              e->st.ms_h2d, h2d_s > 0 ? (double) tv->nels * pbyte / h2d_s / 1e9 : 0.0, tv->host_threads > 0 ? tv->host_threads : 4,
              e->st.ms_pass1, e->st.ms_rclookup, e->st.ms_pass2,
              e->st.ms_total, e->st.ms_total > 0 ? e->st.nels / (e->st.ms_total * 1e-3) : 0.0);
      // where the wall time of this call went (the process adds its own start-up, the table probe and the .smu writer)
      fprintf(stderr, "  [smg] wall %.1f ms: runtime start-up + code object %.1f, allocations + index %.1f, read+h2d+decode %.1f, "
              "conditioning %.1f, passes + host round trips %.1f, results %.1f\n",
              SECS(w0, w2) * 1e3, (SECS(w0, w2) - alloc_s - h2d_s - cond_s - run_s - out_s) * 1e3, alloc_s * 1e3, h2d_s * 1e3,
              cond_s * 1e3, run_s * 1e3, out_s * 1e3);
    }
done:
#undef BAIL
#undef SECS
  if (d_index) hipFree(d_index);
  if (d_plot) hipFree(d_plot);
  if (d_labels) hipFree(d_labels);
  if (d_out) hipFree(d_out);
  smg_engine_destroy(e);
  return rc;
}

extern "C" int smg_hetmers_run(const smg_table_view *tv, const smg_opts *opts, int64_t *plot,
                               smg_stats *stats, char *errbuf, size_t errlen)
{ if (!tv) return fail(errbuf, errlen, SMG_EINVAL, "null argument%s");
  ViewCtx vc; smg_table_source src;
  view_source(tv, &vc, &src);
  return host_run(&src, opts, plot, stats, NULL, NULL, NULL, NULL, errbuf, errlen);
}

extern "C" int smg_hetmers_run_source(const smg_table_source *src, const smg_opts *opts, int64_t *plot,
                                      smg_stats *stats, char *errbuf, size_t errlen)
{ return host_run(src, opts, plot, stats, NULL, NULL, NULL, NULL, errbuf, errlen); }

extern "C" int smg_hetmers_extract(const smg_table_view *tv, const smg_opts *opts, const uint16_t *labels,
                                   int64_t *plot, uint64_t **records, int64_t *nrec, int *rec_words,
                                   smg_stats *stats, char *errbuf, size_t errlen)
{ if (!labels || !records || !nrec || !rec_words) return fail(errbuf, errlen, SMG_EINVAL, "null argument%s");
  *records = NULL; *nrec = 0; *rec_words = 0;
  if (!tv) return fail(errbuf, errlen, SMG_EINVAL, "null argument%s");
  ViewCtx vc; smg_table_source src;
  view_source(tv, &vc, &src);
  return host_run(&src, opts, plot, stats, labels, records, nrec, rec_words, errbuf, errlen);
}

extern "C" void smg_free(void *p) { free(p); }

// ---- stand-alone conditioning: host table in, conditioned host table out (third "next" row of the scope table) ----
extern "C" int smg_condition_table(const smg_table_view *tv, const smg_opts *opts, uint64_t **keys_out,
                                   uint16_t **counts_out, int64_t *nels_out, int *words_out,
                                   char *errbuf, size_t errlen)
{ if (!tv || !opts || !keys_out || !counts_out || !nels_out || !words_out)
    return fail(errbuf, errlen, SMG_EINVAL, "null argument%s");
  *keys_out = NULL; *counts_out = NULL; *nels_out = 0; *words_out = 0;
  if (tv->ibyte < 1 || tv->ibyte > 3) return fail(errbuf, errlen, SMG_EINVAL, "ibyte must be 1, 2 or 3%s");
  const int kbyte = (tv->kmer + 3) >> 2, pbyte = kbyte + 2 - tv->ibyte;
  if (pbyte < 3) return fail(errbuf, errlen, SMG_EINVAL, "k-mer shorter than the index prefix%s");
  smg_engine *e = smg_engine_create(opts->device, NULL, errbuf, errlen);
  if (!e) return SMG_ENODEV;
  int rc = SMG_OK;
  uint8_t *d_rec = NULL; int64_t *d_index = NULL;
  uint64_t *hk = NULL; uint16_t *hc = NULL;
  const size_t ixbytes = sizeof(int64_t) << (8 * tv->ibyte);
#define BAIL(code, msg) { rc = fail(errbuf, errlen, code, msg "%s"); goto done; }
  if (hipMalloc(&d_index, ixbytes) != hipSuccess) BAIL(SMG_ENOMEM, "out of device memory for the table")
  if ((rc = decode_begin(e, tv->kmer, tv->ibyte, tv->nels, errbuf, errlen))) goto done;
  { ViewCtx vc; smg_table_source src;
    view_source(tv, &vc, &src);
    if (hipMemcpy(d_index, tv->prefix_index, ixbytes, hipMemcpyHostToDevice) != hipSuccess)
      BAIL(SMG_ENODEV, "host to device copy failed")
    DecodeHook hk; hk.e = e; hk.d_index = d_index; hk.ibyte = tv->ibyte; hk.ibase = 0;
    if ((rc = ingest_records(&src, pbyte, 0, tv->nels, NULL, opts->device, 4, NULL, errbuf, errlen, decode_hook, &hk))) goto done;
  }
  if (opts->condition
      && (rc = smg_engine_condition(e, opts->ethresh, opts->condition & SMG_COND_TRIM, opts->condition & SMG_COND_SYMM,
                                    NULL, errbuf, errlen)))
    goto done;
  { const int64_t n = e->n;
    hk = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) (n > 0 ? n : 1) * e->W);
    hc = (uint16_t *) malloc(sizeof(uint16_t) * (size_t) (n > 0 ? n : 1));
    if (!hk || !hc) BAIL(SMG_ENOMEM, "out of host memory for the conditioned table")
    if (n > 0 && (hipMemcpy(hk, e->keys, sizeof(uint64_t) * (size_t) n * e->W, hipMemcpyDeviceToHost) != hipSuccess
                  || hipMemcpy(hc, e->cnt, sizeof(uint16_t) * (size_t) n, hipMemcpyDeviceToHost) != hipSuccess))
      BAIL(SMG_ENODEV, "device to host copy failed")
    *keys_out = hk; *counts_out = hc; *nels_out = n; *words_out = e->W;
    hk = NULL; hc = NULL;
  }
done:
#undef BAIL
  free(hk); free(hc);
  if (d_rec) hipFree(d_rec);
  if (d_index) hipFree(d_index);
  smg_engine_destroy(e);
  return rc;
}

