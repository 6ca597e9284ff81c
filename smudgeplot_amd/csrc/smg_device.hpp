// smg_device.hpp -- device-side primitives of the hetmers engine (gfx950 / CDNA4, wave64).
//
// K-mer layout in HBM: W = ceil(k/32) 64-bit words per entry, interleaved (entry i occupies
// words [i*W, i*W+W)), left aligned: base b sits in bits (63-2(b%32), 62-2(b%32)) of word b/32,
// a=0 c=1 g=2 t=3, pad bits zero.  Unsigned word-wise comparison therefore equals the bytewise
// order of the FastK table (libfastk.c:614-636 packing, PloidyPlot.c:125-131 mycmp).

#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SMG_DEV __device__ __forceinline__

typedef unsigned long long u64;

template <int W> struct Key { u64 w[W]; };

template <int W> SMG_DEV Key<W> load_key(const u64 *__restrict__ keys, int64_t i)
{ Key<W> x;
  if constexpr (W == 1) { x.w[0] = keys[i]; }
  else if constexpr (W == 2)
    { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(keys + 2 * i);
      x.w[0] = v.x; x.w[1] = v.y;
    }
  else if constexpr (W == 4)
    { const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(keys + 4 * i);
      const ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(keys + 4 * i + 2);
      x.w[0] = a.x; x.w[1] = a.y; x.w[2] = b.x; x.w[3] = b.y;
    }
  else
    {
#pragma unroll
      for (int w = 0; w < W; w++) x.w[w] = keys[(int64_t) W * i + w];
    }
  return x;
}

template <int W> SMG_DEV bool key_eq(const Key<W> &a, const Key<W> &b)
{ bool e = true;
#pragma unroll
  for (int w = 0; w < W; w++) e &= (a.w[w] == b.w[w]);
  return e;
}

template <int W> SMG_DEV bool key_lt(const Key<W> &a, const Key<W> &b)
{
#pragma unroll
  for (int w = 0; w < W; w++)
    { if (a.w[w] != b.w[w]) return a.w[w] < b.w[w]; }
  return false;
}

// Geometry of one run, uniform over the grid (lives in SGPRs).
struct Geo
{ int  k;          // k-mer length
  int  p0;         // first position the window scan covers: ceil((k-1)/2) on the symmetric path
  int  pw;         // word that holds base p0
  u64  pmask;      // bits of word pw that belong to bases < p0 (0 when p0 % 32 == 0)
  int  mid;        // the self-mirrored position (k-1)/2 for odd k, -1 for even k
  int  wrap;       // k > 85: a uint8 degree can wrap (PloidyPlot.c:163), emulate exactly
};

// do x and y share their first p0 bases?  (same "window block")
template <int W> SMG_DEV bool same_block(const Key<W> &x, const Key<W> &y, const Geo &g)
{ bool e = true;
#pragma unroll
  for (int w = 0; w < W; w++)
    { if (w < g.pw) e &= (x.w[w] == y.w[w]); }
  u64 d = 0;
#pragma unroll
  for (int w = 0; w < W; w++)
    { if (w == g.pw) d = (x.w[w] ^ y.w[w]) & g.pmask; }
  return e && d == 0;
}

// position of the single differing base, or -1 when x and y differ at 0 or >= 2 bases
template <int W> SMG_DEV int pair_pos(const Key<W> &x, const Key<W> &y)
{ int n = 0, pos = -1;
#pragma unroll
  for (int w = 0; w < W; w++)
    { const u64 d = x.w[w] ^ y.w[w];
      const u64 t = (d | (d >> 1)) & 0x5555555555555555ull;
      n += __popcll(t);
      if (t) pos = w * 32 + (__clzll((long long) t) >> 1);
    }
  return n == 1 ? pos : -1;
}

// x with base p replaced by (base ^ d), d in 1..3
template <int W> SMG_DEV Key<W> flip_base(const Key<W> &x, int p, int d)
{ Key<W> y = x;
  const int w = p >> 5;
  const u64 m = (u64) d << (62 - 2 * (p & 31));
#pragma unroll
  for (int v = 0; v < W; v++)
    { if (v == w) y.w[v] ^= m; }
  return y;
}

SMG_DEV u64 rev2_word(u64 x)        // reverse the order of the 32 2-bit groups of a word
{ const u64 y = __brevll(x);        // v_bfrev_b32 x2: groups reversed, the 2 bits of a group swapped
  return ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
}

// rev2_word(~x): the complement of a word with its 32 bases in reverse order.  Bit reversal of the two halves, then
// the two bits of every base are swapped back AND complemented by one v_bitop3_b32 per half:
// (~(y >> 1) & 0x55..) | (~(y << 1) & 0xAA..), truth table 0x1B on (y >> 1, y << 1, 0x55555555)  [8 instructions]
SMG_DEV u64 rev2_comp_word(u64 x)
{ const unsigned zh = __builtin_bitreverse32((unsigned) x), zl = __builtin_bitreverse32((unsigned) (x >> 32));
  const unsigned rh = __builtin_amdgcn_bitop3_b32(zh >> 1, zh << 1, 0x55555555u, 0x1B);
  const unsigned rl = __builtin_amdgcn_bitop3_b32(zl >> 1, zl << 1, 0x55555555u, 0x1B);
  return ((u64) rh << 32) | rl;
}

// reverse complement of a left aligned k-mer (restates compress_comp, PloidyPlot.c:1143-1165)
template <int W> SMG_DEV Key<W> revcomp(const Key<W> &x, int k)
{ Key<W> r, o;
#pragma unroll
  for (int w = 0; w < W; w++) r.w[w] = rev2_comp_word(x.w[W - 1 - w]);
  // r holds the complement reversed over 32*W bases: the wanted k bases are the LAST k of it,
  // i.e. shift the whole W-word value left by 2*(32W-k) bits (always < 64)
  const int s = 2 * (32 * W - k);
#pragma unroll
  for (int w = 0; w < W; w++)
    { u64 v = r.w[w] << s;
      if (w + 1 < W && s) v |= r.w[w + 1] >> (64 - s);
      o.w[w] = v;
    }
  return o;
}

// first index in [lo,hi) whose key is >= t
template <int W> SMG_DEV int64_t lower_bound_key(const u64 *__restrict__ keys, int64_t lo,
                                                 int64_t hi, const Key<W> &t)
{ while (lo < hi)
    { const int64_t m = (lo + hi) >> 1;
      if (key_lt<W>(load_key<W>(keys, m), t)) lo = m + 1; else hi = m;
    }
  return lo;
}

// bucket directory over the leading 32 bits of the k-mer: bstart[b] = first entry whose bucket is >= b.
// bucket(x) = (hi32(x) >> dsh) - b0: two 32-bit VALU operations (a 64-bit subtract + shift of the whole
// word cost 65 instructions per entry in the first version of the pass-1 epilogue).  Tables whose k-mers
// all share their first 16 bases fall into one bucket and are searched by plain bisection: correct,
// only slower, and no real FastK table looks like that.
struct Dir
{ const uint32_t *bstart;
  uint32_t b0;        // bucket value of the first entry before the offset
  int      dsh;       // 0..31
  uint32_t nb;        // number of buckets; bstart has nb+1 entries
};

#define DIR_UNSET 0xFFFFFFFFu   // bstart of an empty bucket (the directory is memset to 0xFF before pass 1)

// clamped into [0, nb): on a sorted table the clamp never acts; on garbage it keeps the writers in bounds
SMG_DEV uint32_t dir_bucket(const Dir &d, u64 w0)
{ const uint32_t b = ((uint32_t) (w0 >> 32) >> d.dsh) - d.b0;
  return b < d.nb ? b : d.nb - 1;
}

// Writers store ONE word per non-empty bucket (bstart[bucket(i)] = i for the first entry i of the bucket)
// plus bstart[nb] = n; empty buckets stay DIR_UNSET and are skipped here.  (Filling the empty buckets from
// the writer side needs a loop per entry whose trip count is the length of the empty run: unbounded on
// skewed tables.)
template <int W> SMG_DEV int64_t find_key(const u64 *__restrict__ keys, const Dir &d,
                                          const Key<W> &t)
{ const uint32_t hb = (uint32_t) (t.w[0] >> 32) >> d.dsh;
  if (hb < d.b0) return -1;
  uint32_t b = hb - d.b0;
  if (b >= d.nb) return -1;
  int64_t lo = d.bstart[b];
  if (lo == (int64_t) DIR_UNSET) return -1;
  int64_t hi = d.bstart[++b];
  while (hi == (int64_t) DIR_UNSET) hi = d.bstart[++b];       // bstart[nb] is always set
  // (Round 4 tried to start the bisection from an interpolated position -- the k-mers of a bucket are spread evenly below
  //  its leading bits -- to keep it inside one or two lines: no gain, 4.74 vs 4.68 ms for the look-ups of the octoploid
  //  table, whose fused probe kernel is bound by the volume of its random map and directory loads, not by their depth.)
  lo = lower_bound_key<W>(keys, lo, hi, t);
  if (lo < hi && key_eq<W>(load_key<W>(keys, lo), t)) return lo;
  return -1;
}

// Round 5: where a requested k-mer IS, found from where it is EXPECTED (the look-ups of the fast path: sig_find without
// signatures, smg_fast.hpp).  On a polyploid table the survivors are real hits
// (one request in seven names a candidate that does have a prefix-side pair: 7.3e7 look-ups on the hexaploid k = 51
// stand-in, more than half of its step), and the fused probe kernels are bound by the NUMBER of lines those look-ups
// miss the L2 with, not by their depth (profiles/r05_lookup_experiments.txt: more waves per CU or four look-ups per lane in
// lockstep change nothing, fewer lines do).  The k-mers of a directory bucket are spread evenly below its leading bits, so
//    position = bucket start + bucket size x (the 32 k-mer bits below the bucket bits) / 2^32
// is off by a few entries (sigma <= sqrt(size) / 2: 4.5 for the 81 entries of a 24-bit bucket of that table).  From there
// the search gallops towards the k-mer -- probes 4, 12 and 28 entries on -- until it is bracketed and bisects the bracket:
// ~4 probes in ~2 lines of k-mers instead of 6.5 in 3.5.  After three gallop steps without a bracket (a skewed bucket:
// repeats, low-complexity sequence) it bisects what is left, so the worst case is four probes more than plain bisection.
// A probe that hits the k-mer ends the search, and a search that ends without such a probe has seen both neighbours of
// the place where the k-mer would be: no separate verifying load.  (Round 4 started the BISECTION at the expected position
// and measured nothing: its next probe is the middle of what is left, i.e. just as far away.)
#ifndef L_GALLOP
#define L_GALLOP 1                         // 0: plain bisection of the directory bucket (find_key)
#endif
#define L_STEP0  3                         // first gallop step: the probe lands 4 entries from the expected position
#define L_GSTEPS 3                         // gallop steps before the search falls back to bisection

template <int W> SMG_DEV int64_t find_key_near(const u64 *__restrict__ keys, const Dir &d, const Key<W> &t)
{ const uint32_t hb = (uint32_t) (t.w[0] >> 32) >> d.dsh;
  if (hb < d.b0) return -1;
  uint32_t bk = hb - d.b0;
  if (bk >= d.nb) return -1;
  uint32_t a = d.bstart[bk];
  if (a == DIR_UNSET) return -1;
  uint32_t b = d.bstart[++bk];
  while (b == DIR_UNSET) b = d.bstart[++bk];                   // (a directory that pass 1 built skips empty buckets; bstart[nb] is set)
  if (a >= b) return -1;
  // [a, b): where the k-mer can still be.  First probe: the expected position; then gallop up or down, then bisect.
  unsigned m = a + __umulhi(b - a, (unsigned) ((t.w[0] << (32 - d.dsh)) >> 32));
  unsigned step = L_STEP0;
  int mode = 0, left = L_GSTEPS;                                // 0 first probe, 1 galloping up, 2 galloping down, 3 bisection
  for (;;)
    { const Key<W> z = load_key<W>(keys, m);
      if (key_eq<W>(z, t)) return (int64_t) m;
      if (key_lt<W>(z, t))
        { a = m + 1u;
          if (mode == 0) mode = 1;
          else if (mode == 1) { step = 2u * step + 1u; if (--left == 0) mode = 3; }
          else mode = 3;
        }
      else
        { b = m;
          if (mode == 0) mode = 2;
          else if (mode == 2) { step = 2u * step + 1u; if (--left == 0) mode = 3; }
          else mode = 3;
        }
      if (a >= b) return -1;
      m = a + ((b - a) >> 1);
      if (mode == 1) { if (a + step < b) m = a + step; else mode = 3; }
      else if (mode == 2) { if (b - a > step) m = b - 1u - step; else mode = 3; }
    }
}

SMG_DEV u64 mix64(u64 z)
{ z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ull;
  z ^= z >> 27; z *= 0x94d049bb133111ebull;
  z ^= z >> 31;
  return z;
}

template <int W> SMG_DEV u64 hash_entry(const Key<W> &x, unsigned c, u64 seed)
{ u64 a = seed;
#pragma unroll
  for (int w = 0; w < W; w++) a = mix64(a ^ x.w[w]);
  return mix64(a ^ (u64) c ^ 0x9e3779b97f4a7c15ull);
}

// 128-bit mixing of (k-mer, count) from full-rate 32-bit add/xor/rotate only (64-bit integer
// multiplies are quarter rate on CDNA): ChaCha-style quarter rounds over a 4-word state.
SMG_DEV void arx_qr(unsigned &a, unsigned &b, unsigned &c, unsigned &d)
{ a += b; d ^= a; d = __builtin_rotateleft32(d, 16);
  c += d; b ^= c; b = __builtin_rotateleft32(b, 12);
  a += b; d ^= a; d = __builtin_rotateleft32(d, 8);
  c += d; b ^= c; b = __builtin_rotateleft32(b, 7);
}

template <int W> SMG_DEV void arx_hash(const Key<W> &x, unsigned cnt, u64 &ha, u64 &hb)
{ unsigned a = 0x61707865u ^ cnt, b = 0x3320646eu, c = 0x79622d32u, d = 0x6b206574u;
#pragma unroll
  for (int w = 0; w < W; w++)
    { a ^= (unsigned) x.w[w]; b ^= (unsigned) (x.w[w] >> 32);
      arx_qr(a, b, c, d);
      if (w + 1 < W) arx_qr(a, b, c, d);
    }
  arx_qr(a, b, c, d);
  arx_qr(a, b, c, d);
  ha = (u64) a | ((u64) b << 32);
  hb = (u64) c | ((u64) d << 32);
}

// Canonical fingerprint of one entry: g(min(x,rc),cnt), nothing for a self-complementary k-mer.  XORed over a
// (strictly sorted, hence duplicate-free) table that is closed under reverse complement with equal counts, the two
// members of every class {x, rc(x)} cancel exactly; any entry whose complement is absent or carries another count
// leaves its 128 bits in the residue.
template <int W> SMG_DEV void fp_accumulate(const Key<W> &x, const Key<W> &rc, unsigned cnt,
                                            u64 &fa, u64 &fb)
{ const bool lt = key_lt<W>(x, rc);
  u64 ha, hb;
  arx_hash<W>(lt ? x : rc, cnt, ha, hb);
  if (!key_eq<W>(x, rc)) { fa ^= ha; fb ^= hb; }
}

SMG_DEV u64 wave_xor_u64(u64 v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o, 64);
  return v;
}

SMG_DEV u64 wave_sum_u64(u64 v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// add v (< 256) to the byte deg[j]; the table is padded to a multiple of 4 bytes
SMG_DEV void deg_add(uint8_t *deg, int64_t j, unsigned v, int wrap)
{ unsigned *wp = reinterpret_cast<unsigned *>(deg + (j & ~(int64_t) 3));
  const unsigned sh = (unsigned) (j & 3) * 8;
  if (!wrap)
    atomicAdd(wp, v << sh);                 // deg <= 3k <= 255: no carry into the neighbour
  else
    { unsigned old = *wp, assumed;
      do
        { assumed = old;
          const unsigned b = ((assumed >> sh) + v) & 0xFFu;
          old = atomicCAS(wp, assumed, (assumed & ~(0xFFu << sh)) | (b << sh));
        }
      while (old != assumed);
    }
}
