// smg_fast.hpp -- the MI355X fast path (k <= 85, reverse-complement closed table).
//
// Three persistent, tile-based kernels.  No per-wave / per-thread atomics on shared addresses:
// a device-scope atomic on ONE address costs ~12 ns on gfx950 (8 XCDs, memory-side atomics), so
// every counter is aggregated per workgroup first (measured: profiles/r01_v1_*).
//
//   kf_pass1  tile of 1024 entries (+32 halo each side) staged in LDS; one thread walks the
//             window block of its entry in LDS (suffix-side positions only) and emits
//               code[i]   : 1 byte  -- 0 none | 63 several pairs | 31+delta (the unique partner is
//                           entry i+delta, |delta| <= 30) | 62 unique but farther; bit 6 = the
//                           pair is not self-mirrored (weight 2)
//               bstart[]  : the bucket directory (built on the fly from the staged tile)
//               requests  : rc(kmer) of every entry that owns a pair at p > k-1-p, compacted
//                           through an LDS queue and written, coalesced, into per-workgroup
//                           chunks of F_CH records (one global atomic per chunk)
//               fingerprints of T and rc(T) (per-workgroup partial sums, no atomics)
//   kf_apply  one workgroup per chunk: directory look-up of the complement, plain store of the
//             "has a prefix-side pair" byte P[j]  (the S_hi(rc(x)) > 0 term of the degree)
//   kf_pass2  needs no k-mers at all: entry i with a unique local partner j = i+delta > i counts
//             iff neither i nor j has its P bit and j's code is "unique" too; the (sum,min) cell is
//             bumped in an LDS tile (sum < 256, triangular, u32), flushed once per workgroup.
//
// Why the code byte is enough (k <= 85, so a uint8 degree cannot wrap, PloidyPlot.c:163):
//   deg(x) = S_all(x) + P(x), P(x) = number of prefix-side pairs = S_hi(rc(x)).
//   A suffix-side pair (i,j) enters the plot iff deg(i) <= 1 and deg(j) <= 1
//   <=> S_all(i) = S_all(j) = 1 and P(i) = P(j) = 0   (the pair itself is the 1).

#pragma once
#include "smg_device.hpp"

#define F_TPB    256
#define F_ITEMS  4
#define F_TILE   (F_TPB * F_ITEMS)
#define F_HALO   32
#define F_SPAN   (F_TILE + 2 * F_HALO)
#define F_CH     4096                 // records per request chunk
#define F_NOCHUNK 0xFFFFFFFFu

#define CODE_NONE   0
#define CODE_MULTI  63
#define CODE_FAR    62
#define CODE_W2     64
#define CODE_DEFER  0xFF                  // placeholder until kf_bigfix has redone the entry
#define CODE_P      0x80                  // set by the look-ups: the entry has a prefix-side pair ("P flag")
// The flag is a BLIND byte store (round 5; rounds 1-4 read the byte, ORed the flag in and stored it: a dependent load that
// misses the L2 for every look-up, 7.3e7 of them on the hexaploid table).  An entry with a prefix-side pair can neither be
// counted as a candidate nor as the partner of one -- every reader of a code byte (kf_pass2 and its far twins, kf_extract)
// tests the flag before anything else of the byte matters -- so what its low bits said is of no further use: the byte
// becomes CODE_P | CODE_NONE.  Bytes of other entries are untouched by a byte store, and a second request that names the
// entry (there is at most one: the one from its reverse complement) would store the same value.
#ifndef L_BLIND_P
#define L_BLIND_P 1                       // 0: read - or - store (A/B builds)
#endif
#ifdef L_ABL_NOSETP                       // ablation build (timing only, WRONG results): the look-ups without their flag stores
#define SET_P(A, j) ((void) 0)
#elif L_BLIND_P
#define SET_P(A, j) ((A).code[j] = (uint8_t) CODE_P)
#else
#define SET_P(A, j) ((A).code[j] = (uint8_t) ((A).code[j] | CODE_P))
#endif

#define P2_TPB   1024
#define P2_SMAX  208                  // LDS plot tile covers sums < P2_SMAX (43.7 KB; with the 32 KB queue: 2 WGs/CU)
#define P2_CELLS ((P2_SMAX / 2) * (P2_SMAX / 2 + 1))     // 10920

struct FastArgs
{ const u64      *keys;
  const uint16_t *cnt;
  int64_t         n;
  Geo             g;
  Dir             dir;           // bstart written by pass 1, read by apply
  uint8_t        *code;
  uint16_t       *sig;           // k <= 64: the 16 k-mer bits below the directory bucket bits (look-up signatures)
  int             sigsh;         //          sig[i] = (uint16_t) (keys[i] >> sigsh)
  uint32_t       *bmap;          // candidate block map (or NULL): bit (hi32(kmer) >> bmsh) is set when a window block
  int             bmsh;          //   (coarsened to <= 32 leading bits) holds an entry with exactly one suffix-side pair
  int             bm2;           // two-bit map (below): 64-bit map words
};

// code byte (P flag masked off) -> "owns a pair at p > k-1-p": several pairs, or one that is not self-mirrored
SMG_DEV bool code_hi(unsigned c) { return c - 63u < 65u; }                    // 63 .. 127

// Two-bit candidate map (one GPU, 32 id bits, one-word k-mers of >= 24 bases).  The map word of 32 block ids is 64 bits
// wide: the low half holds the usual bit (id & 31), the high half a SECOND bit at a position hashed from the 32 k-mer
// bits below the id.  A request passes the filter only if both bits of its target are set: its target's block must hold
// a candidate AND some candidate of the same 32-block group must hash to the same position -- for a request that aims
// at no candidate that happens 11 times less often than the first bit alone (density d = 0.057 on the 1 Gbp table:
// d -> d * d + d / 32), and every survivor costs a look-up of four random memory lines.  Both bits sit in one 8-byte
// word, so marking is one atomic and probing one load.
SMG_DEV unsigned bm2_pos(uint32_t lo) { return (lo * 0x9E3779B1u) >> 27; }
SMG_DEV u64 bm2_bits(uint32_t id, uint32_t lo) { return (u64) (1u << (id & 31u)) | ((u64) (1u << bm2_pos(lo)) << 32); }

struct FastCtl                    // device control words of the fast path
{ unsigned n_chunks;             // chunks handed out (may exceed max_chunks => rerun)
  unsigned missing;              // a complement was absent or carried another count
  unsigned unsorted;             // order violation seen
  unsigned bf_next;              // kf_bigfix: the next slab of the deferred-entry list to be taken
  u64      nreq;                 // requests written
  unsigned nbig;                 // entries deferred to kf_bigfix (window block longer than the halo)
  unsigned nf_chunks;            // kf_filter: chunks of the filtered request list
  u64      nf_req;               // kf_filter: requests that survived
};

// ---- helpers -----------------------------------------------------------------------------------

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence:
// it also waits for every outstanding global store of the wave (vmcnt(0)), which is not needed
// where only LDS is shared and costs a full memory round trip per barrier.
SMG_DEV void lds_barrier()
{ asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int W> SMG_DEV Key<W> lds_key(const u64 *sk, int idx)
{ Key<W> x;
#pragma unroll
  for (int w = 0; w < W; w++) x.w[w] = sk[idx * W + w];
  return x;
}

// first base at which x and y differ (32 * W when they are equal)
template <int W> SMG_DEV int first_diff(const Key<W> &x, const Key<W> &y)
{ int r = 32 * W;
#pragma unroll
  for (int w = W - 1; w >= 0; w--)
    { const u64 d = x.w[w] ^ y.w[w];
      if (d) r = 32 * w + (__clzll((long long) d) >> 1);
    }
  return r;
}

// Exact walk of a window block that outgrows the linear ranges (repeats, low-complexity sequence): hundreds to
// thousands of entries that share their first p0 bases with entry i.
//   1. block bounds: gallop outwards from i in doubling steps, then bisect the last step;
//   2. NARROW by prefix: [lo, hi) holds the entries that share the first p bases with x.  The entries that also share
//      base p are a sub-range around i; a partner at position p lies outside it (same first p bases, another base at p,
//      the same bases behind).  The two bounds of the sub-range and the three flips of base p are FIVE binary searches
//      over the same range: they run in lockstep, five independent loads per step, so a level costs log2(hi - lo)
//      memory round trips; then p + 1 with the sub-range, which is typically a quarter of the range;
//   3. once at most BB_LIN entries are left they are read eight at a time: a single-base difference among entries that
//      share p bases is a pair at a position >= p.
// (The first version bisected the bounds over the whole table and then looked up all 3 (k - p0) flips in the whole
//  block one after the other: ~550 dependent loads per entry; this one needs ~40 for a block of 500 entries.)
#define BB_LIN  24

template <int W> __device__ __forceinline__ void
big_block_walk(const u64 *__restrict__ keys, const uint16_t *__restrict__ cnt, int64_t n,
               const Geo g, int64_t i, unsigned &s_all, unsigned &s_hi, int64_t &partner,
               unsigned &w2)
{ const Key<W> x = load_key<W>(keys, i);
  const unsigned c = cnt[i];
  int64_t lo, hi;
  { int64_t in = i, out, step = 64;
    for (;;)
      { out = in - step;
        if (out < 0) { out = -1; break; }
        if (!same_block<W>(x, load_key<W>(keys, out), g)) break;
        in = out; step <<= 1;
      }
    int64_t a = out + 1, b = in;                            // first entry of the block in (out, in]
    while (a < b)
      { const int64_t m = (a + b) >> 1;
        if (same_block<W>(x, load_key<W>(keys, m), g)) b = m; else a = m + 1;
      }
    lo = a;
    in = i; step = 64;
    for (;;)
      { out = in + step;
        if (out >= n) { out = n; break; }
        if (!same_block<W>(x, load_key<W>(keys, out), g)) break;
        in = out; step <<= 1;
      }
    a = in + 1; b = out;                                    // first entry behind the block in (in, out]
    while (a < b)
      { const int64_t m = (a + b) >> 1;
        if (!same_block<W>(x, load_key<W>(keys, m), g)) b = m; else a = m + 1;
      }
    hi = a;
  }
  s_all = 0; s_hi = 0; partner = -1; w2 = 0;
  int p = g.p0;
#pragma unroll 1
  for (; hi - lo > BB_LIN && p < g.k; p++)
    { Key<W> y[3];
#pragma unroll
      for (int d = 0; d < 3; d++) y[d] = flip_base<W>(x, p, d + 1);
      // search 0: first entry of [lo, i) that shares base p too; 1: first entry of (i, hi) that does not;
      // 2..4: lower bound of the flips in [lo, hi)
      int64_t sa[5] = { lo, i + 1, lo, lo, lo }, sb[5] = { i, hi, hi, hi, hi };
      for (;;)
        { bool any = false;
#pragma unroll
          for (int q = 0; q < 5; q++) any |= sa[q] < sb[q];
          if (!any) break;
          Key<W> z[5]; int64_t m[5];
#pragma unroll
          for (int q = 0; q < 5; q++)
            { m[q] = (sa[q] + sb[q]) >> 1;
              z[q] = load_key<W>(keys, sa[q] < sb[q] ? m[q] : i);
            }
#pragma unroll
          for (int q = 0; q < 5; q++)
            if (sa[q] < sb[q])
              { bool right;                                  // the answer is at m or in front of it
                if (q == 0)      right = first_diff<W>(x, z[q]) > p;
                else if (q == 1) right = first_diff<W>(x, z[q]) <= p;
                else             right = !key_lt<W>(z[q], y[q - 2]);
                if (right) sb[q] = m[q]; else sa[q] = m[q] + 1;
              }
        }
      { Key<W> z[3]; unsigned cz[3]; bool in[3];
#pragma unroll
        for (int d = 0; d < 3; d++)
          { in[d] = sa[2 + d] < hi;
            z[d] = load_key<W>(keys, in[d] ? sa[2 + d] : i);
            cz[d] = cnt[in[d] ? sa[2 + d] : i];
          }
#pragma unroll
        for (int d = 0; d < 3; d++)
          if (in[d] && key_eq<W>(z[d], y[d]) && c + cz[d] <= SMG_SMAX)
            { const unsigned h = (p != g.k - 1 - p);
              if (s_all == 0) { partner = sa[2 + d]; w2 = h; }
              s_all++; s_hi += h;
            }
      }
      lo = sa[0]; hi = sa[1];
    }
  if (p >= g.k) return;
  // what is left shares the first p bases with x
#pragma unroll 1
  for (int64_t base = lo; base < hi; base += 8)
    { Key<W> z[8]; unsigned cz[8]; bool in[8];
#pragma unroll
      for (int j = 0; j < 8; j++)
        { const int64_t q = base + j;
          in[j] = q < hi && q != i;
          z[j] = load_key<W>(keys, in[j] ? q : i);
          cz[j] = cnt[in[j] ? q : i];
        }
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (in[j])
          { const int pp = pair_pos<W>(x, z[j]);
            if (pp >= 0 && c + cz[j] <= SMG_SMAX)
              { const unsigned h = (pp != g.k - 1 - pp);
                if (s_all == 0) { partner = base + j; w2 = h; }
                s_all++; s_hi += h;
              }
          }
    }
}

// out of line for the kernels that meet a long block once in a while (kf_bigfix inlines the walk: its register budget
// is the walk's)
template <int W> __device__ __noinline__ void
big_block_scan(const u64 *__restrict__ keys, const uint16_t *__restrict__ cnt, int64_t n,
               const Geo g, int64_t i, unsigned &s_all, unsigned &s_hi, int64_t &partner,
               unsigned &w2)
{ big_block_walk<W>(keys, cnt, n, g, i, s_all, s_hi, partner, w2); }

SMG_DEV unsigned make_code(unsigned s_all, int64_t delta, unsigned w2)
{ if (s_all == 0) return CODE_NONE;
  if (s_all >= 2) return CODE_MULTI;
  unsigned c = (delta >= -30 && delta <= 30) ? (unsigned) (31 + delta) : CODE_FAR;
  return c | (w2 ? CODE_W2 : 0);
}

// ---- pass 1 ------------------------------------------------------------------------------------

template <int W> __global__ void __launch_bounds__(F_TPB)
kf_pass1(FastArgs A, uint32_t *__restrict__ bstart, u64 *__restrict__ req,
         uint32_t *__restrict__ chunk_fill, unsigned max_chunks, int emit_all, int want_fp,
         u64 *__restrict__ partials, FastCtl *__restrict__ ctl, int64_t ntiles)
{ __shared__ u64      sk[F_SPAN * W];
  __shared__ uint16_t sc[F_SPAN];
  __shared__ u64      sq[F_TILE * (W + 1)];
  __shared__ u64      sfp[F_TPB / 64][4];
  __shared__ unsigned s_qn, s_chunk, s_used;
  __shared__ u64      s_base, s_total;

  const int t = threadIdx.x;
  const Geo g = A.g;
  const int64_t n = A.n;
  u64 f0 = 0, f1 = 0;
  if (t == 0) { s_chunk = F_NOCHUNK; s_used = 0; s_total = 0; }

  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
    { const int64_t lo = tile * F_TILE;
      // stage keys + counts of [lo-HALO, lo+TILE+HALO) in LDS: 16-byte key loads, 8-byte count
      // loads (lo-HALO is a multiple of 32 entries, so the groups never straddle the tile)
      if constexpr (W == 1)
        { for (int p = t; p < F_SPAN / 2; p += F_TPB)
            { const int64_t gi = lo - F_HALO + 2 * p;
              if (gi >= 0 && gi + 1 < n)
                { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(A.keys + gi);
                  sk[2 * p] = v.x; sk[2 * p + 1] = v.y;
                }
              else if (gi >= 0 && gi < n)
                sk[2 * p] = A.keys[gi];
            }
        }
      else
        { for (int idx = t; idx < F_SPAN; idx += F_TPB)
            { const int64_t gi = lo - F_HALO + idx;
              if (gi >= 0 && gi < n)
                { const Key<W> x = load_key<W>(A.keys, gi);
#pragma unroll
                  for (int w = 0; w < W; w++) sk[idx * W + w] = x.w[w];
                }
            }
        }
      for (int p = t; p < F_SPAN / 4; p += F_TPB)
        { const int64_t gi = lo - F_HALO + 4 * p;
          if (gi >= 0 && gi + 3 < n)
            { const ushort4 v = *reinterpret_cast<const ushort4 *>(A.cnt + gi);
              sc[4 * p] = v.x; sc[4 * p + 1] = v.y; sc[4 * p + 2] = v.z; sc[4 * p + 3] = v.w;
            }
          else
            for (int q = 0; q < 4; q++)
              if (gi + q >= 0 && gi + q < n) sc[4 * p + q] = A.cnt[gi + q];
        }
      if (t == 0) s_qn = 0;
      __syncthreads();

#pragma unroll 1
      for (int r = 0; r < F_ITEMS; r++)
        { const int li = t + r * F_TPB;
          const int64_t i = lo + li;
          if (i >= n) continue;
          const int idx = li + F_HALO;
          const Key<W> x = lds_key<W>(sk, idx);
          const unsigned c = sc[idx];
          unsigned s_all = 0, s_hi = 0, w2 = 0;
          int64_t delta = 0;
          bool big = false;
          // forward neighbours
          for (int d = 1; d <= F_HALO; d++)
            { if (i + d >= n) break;
              const Key<W> y = lds_key<W>(sk, idx + d);
              if (!same_block<W>(x, y, g)) break;
              if (d == F_HALO) { big = true; break; }
              const int p = pair_pos<W>(x, y);
              if (p >= 0 && c + (unsigned) sc[idx + d] <= SMG_SMAX)
                { const unsigned hi = (p != g.k - 1 - p);
                  if (s_all == 0) { delta = d; w2 = hi; }
                  s_all++; s_hi += hi;
                }
            }
          // backward neighbours
          if (!big)
            for (int d = 1; d <= F_HALO; d++)
              { if (i - d < 0) break;
                const Key<W> y = lds_key<W>(sk, idx - d);
                if (!same_block<W>(x, y, g)) break;
                if (d == F_HALO) { big = true; break; }
                const int p = pair_pos<W>(x, y);
                if (p >= 0 && c + (unsigned) sc[idx - d] <= SMG_SMAX)
                  { const unsigned hi = (p != g.k - 1 - p);
                    if (s_all == 0) { delta = -d; w2 = hi; }
                    s_all++; s_hi += hi;
                  }
              }
          if (big)
            { int64_t partner;
              big_block_scan<W>(A.keys, A.cnt, n, g, i, s_all, s_hi, partner, w2);
              delta = partner - i;
            }
          A.code[i] = (uint8_t) make_code(s_all, delta, w2);
          if (A.bmap && s_all == 1)                       // a candidate: mark its block for the request filter
            { const uint32_t id = (uint32_t) (x.w[0] >> 32) >> A.bmsh;
              atomicOr(&A.bmap[id >> 5], 1u << (id & 31));
            }

          // bucket directory + strict order check (the predecessor is in the halo)
          { const uint32_t bcur = dir_bucket(A.dir, x.w[0]);
            bool first = i == 0;
            if (i > 0)
              { const Key<W> pv = lds_key<W>(sk, idx - 1);
                first = dir_bucket(A.dir, pv.w[0]) != bcur;
                if (!key_lt<W>(pv, x)) ctl->unsorted = 1;
              }
            if (first) bstart[bcur] = (uint32_t) i;
            if (i == n - 1) bstart[A.dir.nb] = (uint32_t) n;
          }

          const bool emit = emit_all || s_hi > 0;
          if (emit || want_fp)
            { const Key<W> rc = revcomp<W>(x, g.k);
              if (emit)
                { const unsigned q = atomicAdd(&s_qn, 1u);
#pragma unroll
                  for (int w = 0; w < W; w++) sq[q * (W + 1) + w] = rc.w[w];
                  sq[q * (W + 1) + W] = (u64) c | ((u64) (s_hi > 0) << 16);
                }
              if (want_fp) fp_accumulate<W>(x, rc, c, f0, f1);
            }
        }
      __syncthreads();

      // flush the LDS queue into this workgroup's current chunk (coalesced)
      const unsigned qn = s_qn;
      if (qn > 0)
        { if (t == 0)
            { if (s_chunk == F_NOCHUNK || s_used + qn > F_CH)
                { if (s_chunk != F_NOCHUNK && s_chunk < max_chunks) chunk_fill[s_chunk] = s_used;
                  s_chunk = atomicAdd(&ctl->n_chunks, 1u);
                  s_used = 0;
                }
              s_base = (u64) s_chunk * F_CH + s_used;
              s_used += qn;
              s_total += qn;
            }
          __syncthreads();
          if (s_chunk < max_chunks)
            { u64 *o = req + s_base * (W + 1);
              for (unsigned e = t; e < qn * (W + 1); e += F_TPB) o[e] = sq[e];
            }
        }
      __syncthreads();
    }

  if (t == 0)
    { if (s_chunk != F_NOCHUNK && s_chunk < max_chunks) chunk_fill[s_chunk] = s_used;
      if (s_total) atomicAdd(&ctl->nreq, s_total);
    }
  if (want_fp)
    { f0 = wave_xor_u64(f0); f1 = wave_xor_u64(f1);
      if ((t & 63) == 0) { sfp[t >> 6][0] = f0; sfp[t >> 6][1] = f1; }
      __syncthreads();
      if (t < 2)
        { u64 s = 0;
          for (int w = 0; w < F_TPB / 64; w++) s ^= sfp[w][t];
          partials[(size_t) blockIdx.x * 4 + t] = s;
          partials[(size_t) blockIdx.x * 4 + 2 + t] = 0;
        }
    }
}

// index of the entry whose k-mer is y, or -1: directory bucket, bisection on the bucket's signatures, k-mer
// comparison only among entries that share the signature.  A single matching signature is taken for the
// complement ONLY when the closure of the table is proven by other means (the fingerprint of the hash proof, see
// kf_apply_sorted); `verify` (the exact proof, where the look-up IS the proof) compares the k-mer of every hit.
// Without signatures (W = 3) the k-mers are bisected directly.
template <int W> SMG_DEV int64_t sig_find(const FastArgs &A, const Key<W> &y, bool verify)
{ if (A.sig == NULL) return L_GALLOP ? find_key_near<W>(A.keys, A.dir, y) : find_key<W>(A.keys, A.dir, y);
  const Dir d = A.dir;
  const uint32_t hb = (uint32_t) (y.w[0] >> 32) >> d.dsh;
  if (hb < d.b0 || hb - d.b0 >= d.nb) return -1;
  uint32_t b = hb - d.b0;
  int64_t lo = d.bstart[b];
  if (lo == (int64_t) DIR_UNSET) return -1;
  int64_t hi = d.bstart[++b];
  while (hi == (int64_t) DIR_UNSET) hi = d.bstart[++b];
  const unsigned sy = (unsigned) (y.w[0] >> A.sigsh) & 0xFFFFu;
  const int64_t bhi = hi;
  while (lo < hi)                                    // first entry of the bucket with signature >= sy
    { const int64_t m = (lo + hi) >> 1;
      if (A.sig[m] < sy) lo = m + 1; else hi = m;
    }
  int64_t j = lo;
  if (j >= bhi || A.sig[j] != sy) return -1;
  if (j + 1 < bhi && A.sig[j + 1] == sy)             // several entries share the signature: compare k-mers
    { int64_t e2 = j + 1, eh = bhi;                   // end of the run of equal signatures: bisection (repeat-rich
      while (e2 < eh)                                 // tables hold runs of 1e5 entries: a linear walk took seconds)
        { const int64_t m = (e2 + eh) >> 1;
          if (A.sig[m] <= sy) e2 = m + 1; else eh = m;
        }
      j = lower_bound_key<W>(A.keys, j, e2, y);
      if (j >= e2 || !key_eq<W>(load_key<W>(A.keys, j), y)) return -1;
    }
  else if (verify && !key_eq<W>(load_key<W>(A.keys, j), y)) return -1;
  return j;
}

// ---- apply: set the P bit of every requested complement ---------------------------------------
// chunk_fill != NULL : one workgroup per chunk of the local request list
// chunk_fill == NULL : flat list of nflat records (received from other ranks), grid-stride

// rw = words per record: W (the k-mer; its sender owns a pair at p > k-1-p, or it would not have sent) or W + 1
// (+ count | has-such-a-pair << 16; only these can have their count checked)
template <int W> __global__ void __launch_bounds__(F_TPB)
kf_apply(FastArgs A, const u64 *__restrict__ req, const uint32_t *__restrict__ chunk_fill,
         int64_t nflat, int check_count, FastCtl *__restrict__ ctl, int rw)
{ int64_t first, count, stride;
  if (chunk_fill)
    { first = (int64_t) blockIdx.x * F_CH + threadIdx.x; count = (int64_t) blockIdx.x * F_CH + chunk_fill[blockIdx.x];
      stride = F_TPB;
    }
  else
    { first = (int64_t) blockIdx.x * F_TPB + threadIdx.x; count = nflat; stride = (int64_t) gridDim.x * F_TPB; }
  for (int64_t r = first; r < count; r += stride)
    { const u64 *q = req + r * rw;
      Key<W> y;
#pragma unroll
      for (int w = 0; w < W; w++) y.w[w] = q[w];
      const u64 meta = rw > W ? q[W] : 1ull << 16;
      const bool cc = check_count && rw > W;
      const int64_t j = sig_find<W>(A, y, cc);
      bool bad = j < 0;
      if (!bad && cc) bad = A.cnt[j] != (unsigned) (meta & 0xFFFF);
      if (bad) { if (ctl->missing == 0) ctl->missing = 1; continue; }
      if (meta >> 16 & 1) SET_P(A, j);
    }
}

// chunk list -> dense array: dense[off[c] + r] = chunk c, record r   (off = exclusive scan of fills)
__global__ void __launch_bounds__(F_TPB)
kf_compact(const u64 *__restrict__ req, const uint32_t *__restrict__ chunk_fill,
           const uint32_t *__restrict__ chunk_off, int rw, u64 *__restrict__ dense)
{ const unsigned fill = chunk_fill[blockIdx.x] * rw;
  const u64 *src = req + (size_t) blockIdx.x * F_CH * rw;
  u64 *dst = dense + (size_t) chunk_off[blockIdx.x] * rw;
  for (unsigned e = threadIdx.x; e < fill; e += F_TPB) dst[e] = src[e];
}

// holes of the chunk list -> sentinel words the look-ups skip (all ones: for k < 32 the pad bits of a real k-mer
// are zero, so it cannot be a request; k = 32 uses the compaction path)
__global__ void __launch_bounds__(F_TPB)
kf_fill_holes(u64 *__restrict__ req, const uint32_t *__restrict__ chunk_fill)
{ const unsigned fill = chunk_fill[blockIdx.x];
  u64 *c = req + (size_t) blockIdx.x * F_CH;
  for (unsigned e = fill + threadIdx.x; e < F_CH; e += F_TPB) c[e] = ~0ull;
}

// ---- request filter -------------------------------------------------------------------------------------------
// A request only matters when the entry it names is a CANDIDATE (exactly one suffix-side pair): its P flag is read
// by pass 2 for candidates only.  Pass 1 records in a bitmap which window blocks (coarsened to <= 30 leading k-mer
// bits: 128 MB at most, cache resident) hold a candidate; a request whose target block holds none is dropped before
// it is sorted, looked up -- or sent over xGMI in a sharded run, where the per-rank bitmaps are exchanged first
// (16 MB per rank at 8 GPUs against ~8 bytes x 17 % of the entries in requests).  Conservative by construction:
// a set bit only means "maybe".  Records of RW words, the first one the leading word of the target k-mer.
// chunk_fill = NULL: `req` is a flat array of nflat words (read in F_CH slices) that may hold sentinel words ~0 --
// the host sorts long lists on their leading 8 bits first, so that the map words a workgroup probes stay in L2.
template <int RW> __global__ void __launch_bounds__(F_TPB)
kf_filter(const u64 *__restrict__ req, const uint32_t *__restrict__ chunk_fill, unsigned n_chunks, int64_t nflat,
          const uint32_t *__restrict__ bmap, int idshift, u64 *__restrict__ out, uint32_t *__restrict__ out_fill,
          unsigned max_out, FastCtl *__restrict__ ctl)
{ constexpr unsigned SL = RW == 1 ? F_CH : 4 * F_TPB;        // records staged between two flushes
  __shared__ u64      stage[SL * RW];
  __shared__ unsigned s_n, s_chunk, s_used;
  __shared__ u64      s_base, s_total;
  const int t = threadIdx.x;
  if (t == 0) { s_n = 0; s_chunk = F_NOCHUNK; s_used = 0; s_total = 0; }
  __syncthreads();
  for (unsigned c = blockIdx.x; c < n_chunks; c += gridDim.x)
    { unsigned fill;
      if (chunk_fill) fill = chunk_fill[c];
      else
        { const int64_t left = nflat - (int64_t) c * F_CH;
          fill = left < F_CH ? (unsigned) left : (unsigned) F_CH;
        }
      const u64 *src = req + (size_t) c * F_CH * RW;
      for (unsigned s0 = 0; s0 < fill; s0 += SL)
        { const unsigned send = fill < s0 + SL ? fill : s0 + SL;
          for (unsigned r0 = s0; r0 < send; r0 += 4 * F_TPB)     // four independent map reads in flight per thread
            { u64 y[4]; uint32_t wd[4]; bool keep[4];
#pragma unroll
              for (int j = 0; j < 4; j++)
                { const unsigned r = r0 + j * F_TPB + t;
                  y[j] = r < send ? src[(size_t) r * RW] : 0;
                }
#pragma unroll
              for (int j = 0; j < 4; j++) wd[j] = bmap[(uint32_t) (y[j] >> idshift) >> 5];
#pragma unroll
              for (int j = 0; j < 4; j++)
                { const unsigned r = r0 + j * F_TPB + t;
                  keep[j] = r < send && ((wd[j] >> ((uint32_t) (y[j] >> idshift) & 31)) & 1u) && (chunk_fill || y[j] != ~0ull);
                }
#pragma unroll
              for (int j = 0; j < 4; j++)
                { const u64 m = __ballot(keep[j]);
                  if (m)
                    { const int lane = t & 63, lead = __ffsll((long long) m) - 1;
                      unsigned qb = 0;
                      if (lane == lead) qb = atomicAdd(&s_n, (unsigned) __popcll(m));
                      qb = __shfl(qb, lead, 64);
                      if (keep[j])
                        { const unsigned q = qb + __popcll(m & ((1ull << lane) - 1));
                          stage[q * RW] = y[j];
#pragma unroll
                          for (int w = 1; w < RW; w++) stage[q * RW + w] = src[(size_t) (r0 + j * F_TPB + t) * RW + w];
                        }
                    }
                }
            }
          __syncthreads();
          // append the survivors to the workgroup's output chunk, split so that chunks fill to the brim
          const unsigned qn = s_n;
          const unsigned old_chunk = s_chunk, old_used = s_used;      // read by every thread BEFORE thread 0 moves them on
          __syncthreads();
          if (qn)
            { const unsigned room = old_chunk == F_NOCHUNK ? 0u : F_CH - old_used;
              const unsigned head = qn < room ? qn : room;
              if (t == 0)
                { s_base = (u64) old_chunk * F_CH + old_used;
                  if (qn > head)
                    { if (old_chunk != F_NOCHUNK && old_chunk < max_out) out_fill[old_chunk] = F_CH;
                      s_chunk = atomicAdd(&ctl->nf_chunks, 1u);
                      s_used = qn - head;
                    }
                  else s_used = old_used + qn;
                  s_total += qn;
                  s_n = 0;
                }
              __syncthreads();
              if (head && old_chunk < max_out)
                for (unsigned e = t; e < head * RW; e += F_TPB) out[s_base * RW + e] = stage[e];
              if (qn > head && s_chunk < max_out)
                for (unsigned e = t; e < (qn - head) * RW; e += F_TPB) out[(u64) s_chunk * F_CH * RW + e] = stage[head * RW + e];
              __syncthreads();
            }
        }
    }
  if (t == 0)
    { if (s_chunk != F_NOCHUNK && s_chunk < max_out) out_fill[s_chunk] = s_used;
      if (s_total) atomicAdd(&ctl->nf_req, s_total);
    }
}

// self-check of a sort (debug / tests): order on the leading 32 bits + order-free checksums
__global__ void __launch_bounds__(F_TPB)
kf_check_sorted(const u64 *__restrict__ before, const u64 *__restrict__ after, int64_t n, int lobit,
                u64 *__restrict__ out /* [0]=violations [1]=sum before [2]=sum after [3]=xor both */)
{ const int64_t stride = (int64_t) gridDim.x * F_TPB;
  u64 bad = 0, sb = 0, sa = 0, x = 0;
  for (int64_t i = (int64_t) blockIdx.x * F_TPB + threadIdx.x; i < n; i += stride)
    { const u64 a = after[i], b = before[i];
      if (i > 0 && (after[i - 1] >> lobit) > (a >> lobit)) bad++;
      sb += mix64(b); sa += mix64(a); x ^= a ^ b;
    }
  bad = wave_sum_u64(bad); sb = wave_sum_u64(sb); sa = wave_sum_u64(sa);
  for (int o = 32; o > 0; o >>= 1) x ^= __shfl_xor(x, o, 64);
  if ((threadIdx.x & 63) == 0)
    { if (bad) atomicAdd(out, bad);
      atomicAdd(out + 1, sb); atomicAdd(out + 2, sa); atomicXor(out + 3, x);
    }
}

// Requests sorted by k-mer (key-only records): neighbouring lanes look up neighbouring k-mers, so the
// directory words, the k-mer lines and the P bytes they touch are shared -- the look-ups stream the
// table once instead of fetching ~4 random 128-byte lines per request.
// The look-up itself reads SIGNATURES, not k-mers: sig[i] = the 16 k-mer bits below the bucket bits (2 bytes
// per entry, written by pass 1).  Inside its bucket a request is located by bisection on the signatures; a
// single matching signature IS the complement (the run only counts if the table is closed under reverse
// complement, which the fingerprint proves: a request whose complement is absent can then only mark a wrong
// entry in a run that is discarded anyway); several matching signatures are told apart by their k-mers.
// This cuts the look-up traffic from 20 GB of k-mer lines to 5 GB of signature lines (+ rare k-mer reads).
template <int W> __global__ void __launch_bounds__(F_TPB)
kf_apply_sorted(FastArgs A, const u64 *__restrict__ keys_sorted, int64_t nreq, int skip_sentinels,
                FastCtl *__restrict__ ctl)
{ const int64_t stride = (int64_t) gridDim.x * F_TPB;
  for (int64_t r = (int64_t) blockIdx.x * F_TPB + threadIdx.x; r < nreq; r += stride)
    { Key<W> y;
#pragma unroll
      for (int w = 0; w < W; w++) y.w[w] = keys_sorted[r * W + w];
      if (skip_sentinels && y.w[0] == ~0ull) continue;
      const int64_t j = sig_find<W>(A, y, false);
      if (j < 0) { if (ctl->missing == 0) ctl->missing = 1; continue; }
      SET_P(A, j);
    }
}
// (Running 4 independent look-ups per thread in lockstep was tried and lost 1.1 ms of 8.1: the kernel streams
// ~26 GB -- every k-mer line of the table is touched -- so it is bandwidth, not latency, that bounds it.)

// ---- look-ups of records with a count/flag word (W > 1, or the exact proof): ordered through an index -----
// The records are (W+1) words wide, too wide to be moved by every radix pass: a (leading 32 k-mer bits,
// record number) pair is sorted instead and the look-ups walk the records in that order.
__global__ void __launch_bounds__(F_TPB)
kf_sortkey(const u64 *__restrict__ rec, int rw, int64_t n, uint32_t *__restrict__ key, uint32_t *__restrict__ idx)
{ const int64_t stride = (int64_t) gridDim.x * F_TPB;
  for (int64_t r = (int64_t) blockIdx.x * F_TPB + threadIdx.x; r < n; r += stride)
    { key[r] = (uint32_t) (rec[r * rw] >> 32); idx[r] = (uint32_t) r; }
}

template <int W> __global__ void __launch_bounds__(F_TPB)
kf_apply_indexed(FastArgs A, const u64 *__restrict__ rec, const uint32_t *__restrict__ perm, int64_t n,
                 int check_count, FastCtl *__restrict__ ctl, int rw)
{ const int64_t stride = (int64_t) gridDim.x * F_TPB;
  for (int64_t r = (int64_t) blockIdx.x * F_TPB + threadIdx.x; r < n; r += stride)
    { const u64 *q = rec + (size_t) perm[r] * rw;
      Key<W> y;
#pragma unroll
      for (int w = 0; w < W; w++) y.w[w] = q[w];
      const u64 meta = rw > W ? q[W] : 1ull << 16;
      const bool cc = check_count && rw > W;
      const int64_t j = sig_find<W>(A, y, cc);
      bool bad = j < 0;
      if (!bad && cc) bad = A.cnt[j] != (unsigned) (meta & 0xFFFF);
      if (bad) { if (ctl->missing == 0) ctl->missing = 1; continue; }
      if (meta >> 16 & 1) SET_P(A, j);
    }
}

// ---- pass 2 ------------------------------------------------------------------------------------

SMG_DEV unsigned tri_index(unsigned s, unsigned m)       // cell of (sum, min) in the LDS tile
{ const unsigned a = s >> 1;
  return (a + 1) * (a + (s & 1)) + m;
}


// Cells beyond the LDS tile (sum >= P2_SMAX: repeats, organelles, high-coverage tables) go through a small
// direct-mapped cache of (cell, weight) pairs in LDS: the pairs of a repeat family hit a handful of cells, and a global
// atomic per pair on those few addresses cost 3.4 of the 5.1 ms pass 2 took on the table with 5 % repeats.  A slot
// that is taken by another cell sends the pair to the global plot directly, as before.
#define P2_FAR    512                   // slots (4 KB: tile + queue + cache = 80.5 KB, two workgroups per CU)
#define P2_EMPTY  0xFFFFFFFFu

SMG_DEV void plot_bump(unsigned *tile, unsigned *fkey, unsigned *fval, u64 *__restrict__ plot, unsigned ci, unsigned cj, unsigned wgt)
{ const unsigned s = ci + cj, m = ci < cj ? ci : cj;
  if (s < P2_SMAX) { atomicAdd(tile + tri_index(s, m), wgt); return; }
  const unsigned cell = s * SMG_PLOT_COLS + m;
  const unsigned h = (cell * 0x9E3779B1u) >> 23;                  // 9 bits
  const unsigned old = atomicCAS(&fkey[h], P2_EMPTY, cell);
  if (old == P2_EMPTY || old == cell) atomicAdd(&fval[h], wgt);
  else atomicAdd(plot + cell, (u64) wgt);
}

// rare: the unique partner is more than 30 entries away -- search it again
template <int W> __device__ __noinline__ void
far_partner(const FastArgs &A, int64_t i, int64_t &partner, unsigned &w2)
{ unsigned s_all, s_hi;
  big_block_scan<W>(A.keys, A.cnt, A.n, A.g, i, s_all, s_hi, partner, w2);
}

#define P2_VEC   16                       // consecutive entries per thread: one 16-byte code load
#define P2_TILE  (P2_TPB * P2_VEC)        // 16384 entries per workgroup iteration
#define P2_QCAP  (P2_TILE / 2)            // candidates beyond half of a tile are handled where they are found (rare)

// Phase A: every thread inspects 16 code bytes and pushes its "unique pair, partner above"
//          candidates into an LDS queue (wave-aggregated slot allocation).
// Phase B: the queue is drained densely, one candidate per lane: the five dependent-free loads
//          (partner code, two P flags, two counts) are issued together -- walking the candidates
//          inside the 16-entry loop serialised one memory round trip per entry (r01_v2 profile).
// An entry whose unique partner is more than 30 entries away (CODE_FAR) is skipped here and finished by kf_pass2_far:
// from the list of deferred entries that kf_bigfix worked through (every such entry is on it), or -- three-word
// k-mers, whose pass 1 keeps no list -- from a scan of the code bytes.  (The search for a far partner needs ~120 VGPRs;
// called from this kernel it halved the occupancy of all the other entries and put 160 bytes of scratch on each.)
template <int W> SMG_DEV void
p2_candidate(const FastArgs &A, unsigned *tile, unsigned *fkey, unsigned *fval, u64 *__restrict__ plot, int64_t i, unsigned ci)
{ const unsigned lo6 = ci & 63;
  if (lo6 == CODE_FAR) return;
  const unsigned w2 = (ci & CODE_W2) != 0;
  const int64_t j = i + (int) lo6 - 31;
  const unsigned cj = A.code[j], pi = ci & CODE_P, pj = cj & CODE_P;
  const unsigned ni = A.cnt[i], nj = A.cnt[j];
  const unsigned lj = cj & 63;
  if (lj == CODE_NONE || lj == CODE_MULTI) return;    // partner has several pairs
  if (pi | pj) return;                                // a prefix-side pair exists too
  plot_bump(tile, fkey, fval, plot, ni, nj, w2 ? 2u : 1u);
}

// (1024 threads = 4 waves per SIMD and workgroup: two workgroups per CU need <= 64 VGPRs)
template <int W> __global__ void __launch_bounds__(P2_TPB) __attribute__((amdgpu_waves_per_eu(8, 8)))
kf_pass2(FastArgs A, u64 *__restrict__ plot)
{ __shared__ unsigned tile[P2_CELLS];
  __shared__ unsigned queue[P2_QCAP];     // local index (14 bits) | code << 16
  __shared__ unsigned fkey[P2_FAR], fval[P2_FAR];      // cells beyond the tile: cell number, weight
  __shared__ unsigned s_qn[2];            // queue fill of the even / odd tiles of this workgroup
  const int t = threadIdx.x;
  for (int c = t; c < P2_CELLS; c += P2_TPB) tile[c] = 0;
  for (int c = t; c < P2_FAR; c += P2_TPB) { fkey[c] = P2_EMPTY; fval[c] = 0; }
  if (t == 0) { s_qn[0] = 0; s_qn[1] = 0; }
  lds_barrier();

  const int64_t step = (int64_t) gridDim.x * P2_TILE;
  int64_t c0 = (int64_t) blockIdx.x * P2_TILE;
  uint4 cv = make_uint4(0, 0, 0, 0);
  if (c0 + (int64_t) t * P2_VEC < A.n) cv = *reinterpret_cast<const uint4 *>(A.code + c0 + t * P2_VEC);
  for (unsigned it = 0; c0 < A.n; c0 += step, it ^= 1u)
    { // ---- phase A: the candidates among this thread's 16 code bytes, found on the packed words --------------
      // candidate <=> unique partner ahead (low six bits 32..62: bit 5 set, not all six) and no P flag (bit 7)
      const unsigned wv[4] = { cv.x, cv.y, cv.z, cv.w };
      unsigned cm = 0;
#pragma unroll
      for (int q = 0; q < 4; q++)
        { const unsigned w = wv[q];
          const unsigned all6 = ((w & 0x3F3F3F3Fu) + 0x01010101u) >> 6;           // bit 0 of a byte: low six bits == 63
          const unsigned c = (w >> 5) & ~(w >> 7) & ~all6 & 0x01010101u;
          cm |= ((c * 0x10204080u) >> 28) << (4 * q);                             // bit 0 of the four bytes -> a nibble
        }
      { const int64_t left = A.n - (c0 + (int64_t) t * P2_VEC);                  // ragged end of the table
        if (left < P2_VEC) cm = left <= 0 ? 0u : cm & ((1u << left) - 1u);
      }
      // queue slots: exclusive prefix sum of the counts (<= 16) over the wave, bit plane by bit plane, then one
      // LDS atomic per wave
      const unsigned k = (unsigned) __popc(cm);
      unsigned excl = 0, total = 0;
#pragma unroll
      for (int b = 0; b < 5; b++)
        { const u64 m = __ballot((k >> b) & 1u);
          excl += (unsigned) __builtin_amdgcn_mbcnt_hi((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m, 0u)) << b;
          total += (unsigned) __popcll(m) << b;
        }
      if (total)
        { unsigned qb = 0;
          if ((t & 63) == 0) qb = atomicAdd(&s_qn[it], total);
          unsigned slot = (unsigned) __builtin_amdgcn_readfirstlane((int) qb) + excl;
          const u64 lo = (u64) cv.x | ((u64) cv.y << 32), hi = (u64) cv.z | ((u64) cv.w << 32);
          for (unsigned m = cm; m; m &= m - 1, slot++)
            { const int j = __ffs(m) - 1;
              const unsigned ci = (unsigned) ((j < 8 ? lo : hi) >> (8 * (j & 7))) & 0xFFu;
              const int li = t * P2_VEC + j;
              if (slot < P2_QCAP) queue[slot] = (unsigned) li | (ci << 16);
              else p2_candidate<W>(A, tile, fkey, fval, plot, c0 + li, ci);       // queue full: rare, done in place
            }
        }
      // prefetch the next tile's codes while the queue is drained
      const int64_t nx = c0 + step + (int64_t) t * P2_VEC;
      cv = make_uint4(0, 0, 0, 0);
      if (nx < A.n) cv = *reinterpret_cast<const uint4 *>(A.code + nx);
      lds_barrier();
      // ---- phase B ----
      const unsigned qn = s_qn[it] < P2_QCAP ? s_qn[it] : P2_QCAP;
      if (t == 0) s_qn[it ^ 1u] = 0;             // the other counter is idle until the next tile's phase A
      for (unsigned q = t; q < qn; q += P2_TPB)
        { const unsigned e = queue[q];
          p2_candidate<W>(A, tile, fkey, fval, plot, c0 + (e & 0xFFFF), e >> 16);
        }
      lds_barrier();
    }

  for (int c = t; c < P2_FAR; c += P2_TPB)
    if (fval[c]) atomicAdd(plot + fkey[c], (u64) fval[c]);
  // flush the LDS tile: (sum,min) rows are laid out triangularly
  for (int c = t; c < P2_CELLS; c += P2_TPB)
    { const unsigned v = tile[c];
      if (!v) continue;
      unsigned a = (unsigned) ((sqrtf(4.0f * c + 1.0f) - 1.0f) * 0.5f);
      while ((a + 1) * (a + 2) <= (unsigned) c) a++;
      while (a * (a + 1) > (unsigned) c) a--;
      unsigned s, m;
      if ((unsigned) c >= (a + 1) * (a + 1)) { s = 2 * a + 1; m = c - (a + 1) * (a + 1); }
      else { s = 2 * a; m = c - a * (a + 1); }
      atomicAdd(plot + (size_t) s * SMG_PLOT_COLS + m, (u64) v);
    }
}

// the far partners of pass 2: the deferred entries of pass 1 (list[0 .. nlist)) whose code says "one pair, partner out of
// the code's reach" -- found again by the block walk, then the same tests as p2_candidate, straight into the plot
template <int W> SMG_DEV void p2_far_entry(const FastArgs &A, int64_t i, u64 *__restrict__ plot)
{ int64_t j; unsigned w2;
  far_partner<W>(A, i, j, w2);
  if (j <= i) return;
  const unsigned cj = A.code[j], lj = cj & 63;
  if (lj == CODE_NONE || lj == CODE_MULTI || (cj & CODE_P)) return;
  const unsigned ni = A.cnt[i], nj = A.cnt[j];
  const unsigned sm = ni + nj, mn = ni < nj ? ni : nj;
  atomicAdd(plot + (size_t) sm * SMG_PLOT_COLS + mn, (u64) (w2 ? 2u : 1u));
}

// (kf_bigfix left the partner of such an entry in farp[q]: nothing is searched again)
template <int W> __global__ void __launch_bounds__(256)
kf_pass2_far(FastArgs A, const uint32_t *__restrict__ list, const uint32_t *__restrict__ farp, unsigned nlist, u64 *__restrict__ plot)
{ for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < nlist; q += gridDim.x * blockDim.x)
    { const int64_t i = list[q];
      const unsigned ci = A.code[i];
      if ((ci & 63) != CODE_FAR || (ci & CODE_P)) continue;
      const int64_t j = farp[q];
      if (j <= i) continue;
      const unsigned cj = A.code[j], lj = cj & 63;
      if (lj == CODE_NONE || lj == CODE_MULTI || (cj & CODE_P)) continue;
      const unsigned ni = A.cnt[i], nj = A.cnt[j];
      const unsigned sm = ni + nj, mn = ni < nj ? ni : nj;
      atomicAdd(plot + (size_t) sm * SMG_PLOT_COLS + mn, (u64) ((ci & CODE_W2) ? 2u : 1u));
    }
}

// the same without a list: 16 code bytes per thread and step
template <int W> __global__ void __launch_bounds__(256)
kf_pass2_farscan(FastArgs A, u64 *__restrict__ plot)
{ const int64_t nv = (A.n + 15) >> 4;
  for (int64_t v = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (int64_t) gridDim.x * blockDim.x)
    { const uint4 cv = *reinterpret_cast<const uint4 *>(A.code + (v << 4));      // (the code array is padded to 16 bytes)
      const unsigned wv[4] = { cv.x, cv.y, cv.z, cv.w };
#pragma unroll 1
      for (int q = 0; q < 4; q++)
        { // bytes whose low six bits are 62 and whose bit 7 is clear
          const unsigned w = wv[q];
          const unsigned x6 = (w & 0x3F3F3F3Fu) ^ 0x3E3E3E3Eu;                    // zero in the low six bits <=> 62
          unsigned hit = ~(((x6 & 0x3F3F3F3Fu) + 0x3F3F3F3Fu) >> 6) & ~(w >> 7) & 0x01010101u;
          for (; hit; hit &= hit - 1)
            { const int64_t i = (v << 4) + 4 * q + ((__ffs(hit) - 1) >> 3);
              if (i < A.n) p2_far_entry<W>(A, i, plot);
            }
        }
    }
}

// ---- extract: pass 2 with another sink (reference: extract_kmer_pairs, src/lib/PloidyList.c) ------------
// Every pair that enters the plot at a LABELLED pixel (labels[sum*501+min] > 0, from the .sma file,
// PloidyList.c:1312-1346) yields one record per represented pair: the k-mer to print (the member with the
// larger count; ties: the one with the smaller base at the variant position, PloidyList.c:430-446),
// then  position | alt base << 8 | label << 16.  A pair found at p != k-1-p also stands for its mirror
// image (rc of both members at k-1-p), which the reference finds and prints separately.
// Records are staged in LDS and appended with one global atomic per flush.
#define EX_STAGE 1024                   // records per workgroup staging buffer

template <int W> SMG_DEV unsigned base_at(const Key<W> &x, int p)
{ unsigned b = 0;
#pragma unroll
  for (int w = 0; w < W; w++)
    if (w == (p >> 5)) b = (unsigned) (x.w[w] >> (62 - 2 * (p & 31))) & 3u;
  return b;
}

template <int W> __global__ void __launch_bounds__(F_TPB)
kf_extract(FastArgs A, const uint16_t *__restrict__ labels, u64 *__restrict__ out, u64 capacity,
           u64 *__restrict__ total)
{ __shared__ u64      stage[EX_STAGE * (W + 1)];
  __shared__ unsigned s_n;
  __shared__ u64      s_base;
  const int t = threadIdx.x;
  const int k = A.g.k;
  if (t == 0) s_n = 0;
  __syncthreads();
  const int64_t step = (int64_t) gridDim.x * F_TPB;
  const int64_t rounds = (A.n + step - 1) / step;
  for (int64_t rd = 0; rd < rounds; rd++)
    { const int64_t i = rd * step + (int64_t) blockIdx.x * F_TPB + t;
      if (i < A.n)
        { const unsigned ci = A.code[i], lo6 = ci & 63;
          if (lo6 >= 32 && lo6 != CODE_MULTI)                     // unique pair, partner above
            { int64_t j;
              unsigned w2 = (ci & CODE_W2) != 0;
              bool ok = true;
              if (lo6 == CODE_FAR) { far_partner<W>(A, i, j, w2); ok = j > i; }
              else j = i + (int) lo6 - 31;
              if (ok)
                { const unsigned cj = A.code[j], lj = cj & 63;
                  ok = lj != CODE_NONE && lj != CODE_MULTI && !((ci | cj) & CODE_P);
                }
              if (ok)
                { const unsigned ni = A.cnt[i], nj = A.cnt[j];
                  const unsigned s = ni + nj, m = ni < nj ? ni : nj;
                  const unsigned lab = labels[(size_t) s * SMG_PLOT_COLS + m];
                  if (lab)
                    { const Key<W> xi = load_key<W>(A.keys, i), xj = load_key<W>(A.keys, j);
                      const int p = pair_pos<W>(xi, xj);
                      const unsigned bi = base_at<W>(xi, p), bj = base_at<W>(xj, p);     // bi < bj (i < j)
                      const unsigned nrec = w2 ? 2u : 1u;
                      const unsigned q = atomicAdd(&s_n, nrec);
                      // the pair itself: a = i (smaller base); cnt[a] < cnt[b] prints b with alt base of a
                      { const bool pj = ni < nj;
                        const Key<W> &who = pj ? xj : xi;
                        u64 *o = stage + (size_t) q * (W + 1);
#pragma unroll
                        for (int w = 0; w < W; w++) o[w] = who.w[w];
                        o[W] = (u64) p | ((u64) (pj ? bi : bj) << 8) | ((u64) lab << 16);
                      }
                      if (w2)
                        { // mirror image at k-1-p: a = rc(xj) (base 3-bj is the smaller one), b = rc(xi)
                          const bool pb = nj < ni;                  // cnt[a] < cnt[b]: print b = rc(xi)
                          const Key<W> who = revcomp<W>(pb ? xi : xj, k);
                          u64 *o = stage + (size_t) (q + 1) * (W + 1);
#pragma unroll
                          for (int w = 0; w < W; w++) o[w] = who.w[w];
                          o[W] = (u64) (k - 1 - p) | ((u64) (pb ? 3u - bj : 3u - bi) << 8) | ((u64) lab << 16);
                        }
                    }
                }
            }
        }
      __syncthreads();
      // flush when the next round might not fit (a round adds at most 2 records per thread)
      const unsigned have = s_n;
      __syncthreads();                       // everybody has read s_n before thread 0 resets it
      if (have + 2 * F_TPB > EX_STAGE || rd + 1 == rounds)
        { if (have)
            { if (t == 0) { s_base = atomicAdd(total, (u64) have); s_n = 0; }
              __syncthreads();
              const u64 base = s_base;
              for (unsigned e = t; e < have * (W + 1); e += F_TPB)
                { const u64 r = base + e / (W + 1);
                  if (r < capacity) out[r * (W + 1) + e % (W + 1)] = stage[e];
                }
            }
          __syncthreads();
        }
    }
}

// total weight in the plot (stat only)
__global__ void __launch_bounds__(1024) kf_plot_sum(const u64 *__restrict__ plot, u64 *__restrict__ out)
{ __shared__ u64 part[16];
  u64 s = 0;
  for (int c = blockIdx.x * 1024 + threadIdx.x; c < SMG_PLOT_CELLS; c += gridDim.x * 1024) s += plot[c];
  s = wave_sum_u64(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0)
    { u64 tot = 0;
      for (int w = 0; w < 16; w++) tot += part[w];
      if (tot) atomicAdd(out, tot);                    // (*out is zeroed by the host; a few dozen workgroups)
    }
}

// ---- routing (sharded runs): requests -> per-destination contiguous send buffer ----------------

template <int W> SMG_DEV int rank_of(const u64 *q, const u64 *__restrict__ split, int nranks)
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

// counts[chunk][rank]
template <int W> __global__ void __launch_bounds__(F_TPB)
kf_route_count(const u64 *__restrict__ req, const uint32_t *__restrict__ chunk_fill, int rw,
               const u64 *__restrict__ split, int nranks, uint32_t *__restrict__ counts)
{ __shared__ unsigned cnt[16];
  if (threadIdx.x < 16) cnt[threadIdx.x] = 0;
  __syncthreads();
  const unsigned fill = chunk_fill[blockIdx.x];
  for (unsigned r = threadIdx.x; r < fill; r += F_TPB)
    atomicAdd(&cnt[rank_of<W>(req + ((size_t) blockIdx.x * F_CH + r) * rw, split, nranks)], 1u);
  __syncthreads();
  if ((int) threadIdx.x < nranks) counts[(size_t) blockIdx.x * nranks + threadIdx.x] = cnt[threadIdx.x];
}

// offsets[chunk][rank] = records of that rank in the chunks in front (exclusive scan down column `rank`); totals[rank].
// One workgroup per rank; the send buffer is destination-major, so group (chunk, rank) starts at
// sum(totals[< rank]) + offsets[chunk][rank].  (Round 2 brought the counts to the host, scanned them there and sent the
// offsets back: two host round trips per step of a sharded run.)
__global__ void __launch_bounds__(1024)
kf_route_offsets(const uint32_t *__restrict__ counts, unsigned nc, int nranks, u64 *__restrict__ offsets, u64 *__restrict__ totals)
{ __shared__ u64 wsum[16];
  __shared__ u64 s_carry;
  const int r = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) s_carry = 0;
  __syncthreads();
  for (unsigned c0 = 0; c0 < nc; c0 += 1024)
    { const unsigned c = c0 + t;
      const u64 v = c < nc ? (u64) counts[(size_t) c * nranks + r] : 0ull;
      u64 incl = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1)
        { const u64 u = __shfl_up(incl, o, 64);
          if (lane >= o) incl += u;
        }
      if (lane == 63) wsum[wv] = incl;
      __syncthreads();
      u64 before = s_carry;
      for (int w = 0; w < wv; w++) before += wsum[w];
      if (c < nc) offsets[(size_t) c * nranks + r] = before + incl - v;
      __syncthreads();
      if (t == 1023) s_carry = before + incl;
      __syncthreads();
    }
  if (t == 0) totals[r] = s_carry;
}

// the records of every chunk -> their groups in the send buffer
template <int W> __global__ void __launch_bounds__(F_TPB)
kf_route_scatter(const u64 *__restrict__ req, const uint32_t *__restrict__ chunk_fill, int rw,
                 const u64 *__restrict__ split, int nranks, const u64 *__restrict__ offsets, const u64 *__restrict__ totals,
                 u64 *__restrict__ out, u64 capacity /* records the send buffer holds */)
{ __shared__ unsigned cur[16];
  __shared__ u64 rbase[16];
  if (threadIdx.x < 16)
    { cur[threadIdx.x] = 0;
      u64 b = 0;
      for (int r = 0; r < (int) threadIdx.x && r < nranks; r++) b += totals[r];
      rbase[threadIdx.x] = b;
    }
  __syncthreads();
  const unsigned fill = chunk_fill[blockIdx.x];
  for (unsigned r = threadIdx.x; r < fill; r += F_TPB)
    { const u64 *q = req + ((size_t) blockIdx.x * F_CH + r) * rw;
      const int d = rank_of<W>(q, split, nranks);
      const u64 slot = rbase[d] + offsets[(size_t) blockIdx.x * nranks + d] + atomicAdd(&cur[d], 1u);
      // A plain step sized the buffer from this very list.  A REPLAYED step sized it from the recorded step's count: if the
      // table's counts moved so that more requests survive, the surplus is dropped here instead of written past the end --
      // the totals differ from the record, the verdict kernel says so and every rank runs the step again the plain way.
      if (slot >= capacity) continue;
      u64 *o = out + slot * rw;
      for (int w = 0; w < rw; w++) o[w] = q[w];
    }
}

// Deferred entries of kf_pass1_d: entries with a pair at distance 4..30 in the sorted table (0.8 % of the pairs of a
// diploid table) and entries whose window block is longer than the +-30 entry window (repeats, low-complexity k-mers).
// Pass 1 sets their bits in `dbits` (one bit per table entry).  Two kernels redo them exactly:
//   kf_collect  scans the bit map, clears it for the next run and compacts the marked entries into a list (one global
//               atomic per workgroup and round: marked entries come in clusters -- whole repeat regions -- so the list,
//               not the map, is what deals them out evenly);
//   kf_bigfix   one entry per thread and round from that list: final code byte, map bit of a candidate, a request for an
//               owner of a pair at p > k-1-p.  Short blocks are walked in batches of eight independent loads, blocks
//               that reach past +-BF_LIN entries by bisection (~800 dependent steps).
// (The first version of this round let every thread work through the bits of its own map words: 63 ms on the table
//  with 5 % repeats, where a few workgroups owned all the work; the list takes 9.4e6 entries through in ~7 ms.)
#define BF_TPB   1024
#define BF_LIN   48                     // blocks of up to ~2 * BF_LIN entries are walked linearly

template <int W> SMG_DEV void
block_scan(const u64 *__restrict__ keys, const uint16_t *__restrict__ cnt, int64_t n, const Geo g, int64_t i,
           unsigned &s_all, unsigned &s_hi, int64_t &partner, unsigned &w2)
{ // The marked entries lie anywhere in the table: every load is a cold line (and mostly a cold page).  The
  // neighbours are therefore fetched eight at a time, independent loads in flight together -- a walk of one
  // dependent load per step took 0.19 us per entry with every lane of the chip busy.
  const Key<W> x = load_key<W>(keys, i);
  // a block that reaches past the linear range on either side goes straight to the bisection
  { const int64_t lo = i - BF_LIN - 1, hi = i + BF_LIN + 1;
    const bool far_lo = lo >= 0 && same_block<W>(x, load_key<W>(keys, lo >= 0 ? lo : i), g);
    const bool far_hi = hi < n && same_block<W>(x, load_key<W>(keys, hi < n ? hi : i), g);
    if (far_lo || far_hi) { big_block_walk<W>(keys, cnt, n, g, i, s_all, s_hi, partner, w2); return; }
  }
  const unsigned c = cnt[i];
  s_all = 0; s_hi = 0; partner = -1; w2 = 0;
#pragma unroll 1
  for (int dir = -1; dir <= 1; dir += 2)
    { bool open = true;
#pragma unroll 1
      for (int base = 1; open && base <= BF_LIN; base += 8)
        { Key<W> y[8]; unsigned cy[8]; bool in[8];
#pragma unroll
          for (int j = 0; j < 8; j++)
            { const int64_t q = i + (int64_t) dir * (base + j);
              in[j] = q >= 0 && q < n;
              y[j] = load_key<W>(keys, in[j] ? q : i);
              cy[j] = cnt[in[j] ? q : i];
            }
#pragma unroll
          for (int j = 0; j < 8; j++)
            { open = open && in[j] && same_block<W>(x, y[j], g);
              if (open)
                { const int p = pair_pos<W>(x, y[j]);
                  if (p >= 0 && c + cy[j] <= SMG_SMAX)               // (same block: p >= p0)
                    { const unsigned h = (p != g.k - 1 - p);
                      if (s_all == 0) { partner = i + (int64_t) dir * (base + j); w2 = h; }
                      s_all++; s_hi += h;
                    }
                }
            }
        }
    }
}

// marked entries of the bit map -> list[0 .. *count) (entries beyond `cap` are counted, not stored, and keep their
// bits: the host grows the list and runs another round); a thread takes BC_Q x four map words (512 table entries) per
// round -- independent 16-byte loads in flight together: a round costs two barriers and one returning global atomic,
// and with one quad per thread the 38 rounds of the 2.5e9-entry table were 0.24 ms of latency for 317 MB
#define BC_Q 4
__global__ void __launch_bounds__(BF_TPB)
kf_collect(uint32_t *__restrict__ dbits, int64_t nwords, uint32_t *__restrict__ list, unsigned cap, unsigned *__restrict__ count)
{ __shared__ unsigned s_n[2], s_base[2];
  const int t = threadIdx.x;
  if (t < 2) s_n[t] = 0;
  __syncthreads();
  const int64_t nquads = nwords >> 2;
  const int64_t per_round = (int64_t) gridDim.x * BF_TPB * BC_Q;
  const int64_t rounds = (nquads + per_round - 1) / per_round;
  for (int64_t rd = 0; rd < rounds; rd++)
    { const int par = (int) (rd & 1);
      const int64_t w0 = (rd * gridDim.x + blockIdx.x) * (int64_t) (BF_TPB * BC_Q) + t;
      uint4 w4[BC_Q];
      unsigned c = 0;
#pragma unroll
      for (int q = 0; q < BC_Q; q++)
        { const int64_t wi = w0 + (int64_t) q * BF_TPB;
          w4[q] = make_uint4(0, 0, 0, 0);
          if (wi < nquads) w4[q] = reinterpret_cast<const uint4 *>(dbits)[wi];
        }
#pragma unroll
      for (int q = 0; q < BC_Q; q++) c += (unsigned) (__popc(w4[q].x) + __popc(w4[q].y) + __popc(w4[q].z) + __popc(w4[q].w));
      unsigned pos = 0;
      if (c) pos = atomicAdd(&s_n[par], c);
      __syncthreads();
      const unsigned tot = s_n[par];
      if (tot == 0) continue;                           // (nothing marked in these entries)
      if (t == 0) s_base[par] = atomicAdd(count, tot);
      __syncthreads();
      const unsigned base = s_base[par];
      if (t == 0) s_n[par] = 0;                         // (used again two rounds on, behind the barriers of the next round)
      if (c && base + pos + c <= cap)
        { unsigned o = base + pos;
#pragma unroll
          for (int q = 0; q < BC_Q; q++)
            { const int64_t wi = w0 + (int64_t) q * BF_TPB;
              const uint32_t w[4] = { w4[q].x, w4[q].y, w4[q].z, w4[q].w };
              if (w[0] | w[1] | w[2] | w[3])
                {
#pragma unroll
                  for (int j = 0; j < 4; j++)
                    for (uint32_t m = w[j]; m; m &= m - 1) list[o++] = (uint32_t) ((wi * 4 + j) * 32 + __ffs(m) - 1);
                  reinterpret_cast<uint4 *>(dbits)[wi] = make_uint4(0, 0, 0, 0);
                }
            }
        }
    }
}

// (Round 4 tried to solve the listed entries of one long window block together: the wave finds the block's bounds with four
//  rounds of probes, stages up to 512 k-mers in its corner of LDS and every member looks its 3 (k - p0) flips up there.
//  Correct -- 255 GPU tests -- and slower: 4.07 instead of 3.39 ms for the 9.4e6 deferred entries of the repeats table
//  -- the group search runs for every batch, most long blocks of that table outgrow the stage, and a solved group still pays
//  4300 vector instructions; the per-entry prefix narrowing below stays.)
// Work is dealt out twice: a workgroup takes a SLAB of BF_SLAB listed entries at a time (global counter: the entries of
// a repeat region sit next to each other in the list), its waves take 64 of them at a time (LDS counter) and run without
// a barrier until the slab is done -- a walk through a block of a thousand entries holds up its own wave, not the other
// fifteen (the first version synchronised the workgroup after every 1024 entries: it waited for the longest walk 9000
// times).  Entries that owe a request are only noted; after the slab's barrier the workgroup writes their requests out.
#define BF_SLAB  8192


template <int W, int RW> __global__ void __launch_bounds__(BF_TPB)
kf_bigfix(FastArgs A, const uint32_t *__restrict__ biglist, const unsigned *__restrict__ pcount, unsigned cap, u64 *__restrict__ req,
          uint32_t *__restrict__ chunk_fill, unsigned max_chunks, FastCtl *__restrict__ ctl,
          unsigned *__restrict__ whist /* this kernel's rows */, unsigned owner0, unsigned owners, int hbits,
          uint32_t *__restrict__ farp /* [cap]: the partner of a listed entry whose code says "out of reach" */)
{ constexpr int rw = RW;
  __shared__ uint32_t slist[BF_SLAB];   // entries of the slab that owe a request
  __shared__ unsigned hist[1024];       // requests of this workgroup per look-up bucket (a row of whist, like pass 1's)
  __shared__ unsigned s_nl, s_next, s_slab, s_chunk, s_used;
  __shared__ u64      s_base, s_total;
  const int t = threadIdx.x, lane = t & 63;
  if (t == 0) { s_chunk = F_NOCHUNK; s_used = 0; s_total = 0; }
  for (int b = t; b < 1024; b += BF_TPB) hist[b] = 0;
  const unsigned nbig = *pcount;
  if (nbig > cap) return;                // the list overflowed (it has holes at its end): the host grows it and redoes the run
  // slab size: at least four slabs per workgroup (a short list -- one rank's eighth of a table -- in 8192-entry slabs kept
  // 50 of 512 workgroups busy: 175 us instead of 35), whole waves, at most what the note list holds
  unsigned slab = ((nbig / (gridDim.x * 4u)) + 63u) & ~63u;
  slab = slab < 256u ? 256u : slab > (unsigned) BF_SLAB ? (unsigned) BF_SLAB : slab;
  for (;;)
    { __syncthreads();
      if (t == 0) { s_slab = atomicAdd(&ctl->bf_next, slab); s_next = 0; s_nl = 0; }
      __syncthreads();
      const unsigned slab0 = s_slab;
      if (slab0 >= nbig) break;
      const unsigned slab_n = nbig - slab0 < slab ? nbig - slab0 : slab;
      for (;;)
        { unsigned b = 0;
          if (lane == 0) b = atomicAdd(&s_next, 64u);
          b = (unsigned) __builtin_amdgcn_readfirstlane((int) b);
          if (b >= slab_n) break;
          if (b + lane < slab_n)
            { const unsigned r = slab0 + b + lane;
              const int64_t i = biglist[r];
              unsigned s_all, s_hi, w2;
              int64_t partner;
              block_scan<W>(A.keys, A.cnt, A.n, A.g, i, s_all, s_hi, partner, w2);
              const unsigned code = make_code(s_all, partner - i, w2);
              A.code[i] = (uint8_t) code;
              if ((code & 63) == CODE_FAR) farp[r] = (uint32_t) partner;
              if (W <= 2 && A.bmap && s_all == 1)             // a candidate: mark its block for the request filter
                { const uint32_t id = (uint32_t) (A.keys[i * W] >> 32) >> A.bmsh;
                  if (A.bm2) atomicOr(reinterpret_cast<u64 *>(A.bmap) + (id >> 5), bm2_bits(id, (uint32_t) A.keys[i * W]));
                  else atomicOr(&A.bmap[id >> 5], 1u << (id & 31));
                }
              if (s_hi > 0) slist[atomicAdd(&s_nl, 1u)] = (uint32_t) i;
            }
        }
      __syncthreads();
      // the slab's requests, BF_TPB at a time into the workgroup's chunks
      const unsigned nl = s_nl;
      for (unsigned q0 = 0; q0 < nl; q0 += BF_TPB)
        { const unsigned qn = nl - q0 < BF_TPB ? nl - q0 : BF_TPB;
          if (t == 0)
            { if (s_chunk == F_NOCHUNK || s_used + qn > F_CH)
                { if (s_chunk != F_NOCHUNK && s_chunk < max_chunks) chunk_fill[s_chunk] = s_used;
                  // (look-up chain: chunk slots owner, owner + owners, .. as in pass 1; this kernel's owners follow pass 1's)
                  if (whist && hbits) s_chunk = s_chunk == F_NOCHUNK ? owner0 + blockIdx.x : s_chunk + owners;
                  else s_chunk = atomicAdd(&ctl->n_chunks, 1u);
                  s_used = 0;
                }
              s_base = (u64) s_chunk * F_CH + s_used;
              s_used += qn; s_total += qn;
            }
          __syncthreads();
          if ((unsigned) t < qn)
            { const int64_t i = slist[q0 + t];
              const Key<W> rc = revcomp<W>(load_key<W>(A.keys, i), A.g.k);
              if (s_chunk < max_chunks)
                { u64 *o = req + (s_base + t) * rw;
#pragma unroll
                  for (int w = 0; w < W; w++) o[w] = rc.w[w];
                  if (rw > W) o[W] = (u64) A.cnt[i] | (1ull << 16);
                }
              if (whist && hbits) atomicAdd(&hist[(unsigned) (rc.w[0] >> 32) >> (32 - hbits)], 1u);
            }
          __syncthreads();
        }
    }
  if (t == 0)
    { if (s_chunk != F_NOCHUNK && s_chunk < max_chunks) chunk_fill[s_chunk] = s_used;
      if (whist && hbits && s_chunk != F_NOCHUNK) atomicMax(&ctl->n_chunks, s_chunk + 1u);
      if (s_total) atomicAdd(&ctl->nreq, s_total);
    }
  __syncthreads();
  if (whist && hbits)
    for (int b = t; b < 1024; b += BF_TPB) whist[(size_t) blockIdx.x * 1024 + b] = hist[b];
}
