// smg_pass1d.hpp -- kf_pass1_d: pass 1 for k <= 64 (W = 1 or 2 64-bit words per k-mer); W = 1 is the headline kernel.
//
// Successor of kf_pass1_r (round 1, 222 vector instructions per table entry at 78 % of the VALU issue rate:
// one wave64 integer VALU instruction per 4 cycles and SIMD, tools/valu_rates.hip).  The instruction count is
// the only lever, so this version is built around three observations:
//
//   * LANE MASKS INSTEAD OF COUNTERS.  The result of a one-away test is one bit per lane = an SGPR pair.
//     "entry e has a pair / has several pairs / has its pair on the self-mirrored position" are then scalar
//     and/or chains over those masks -- SALU instructions, which issue beside the VALU ones -- and the hand-over
//     of a result to the neighbouring lane (entry a of lane l pairs with entry b of lane l+1) is a scalar shift
//     of the mask by one bit.  No credit words, no LDS atomics (7 per thread and tile, 4-way bank conflicted, in
//     kf_pass1_r), no packed-counter arithmetic: a test costs 5 VALU instructions (was 12).
//   * DPP INSTEAD OF REDUNDANT LOADS.  A thread owns 4 CONSECUTIVE entries and needs the next 3 for the
//     distances 1..3: they are its right neighbour's registers (v_xor_b32_dpp wave_shl:1 -- folded into the
//     test's xor, no extra instruction).  Lane 0 and lane 63 of every wave are halo lanes (they repeat the last /
//     first scanned thread of the neighbouring wave), so nothing crosses a wave through memory: 62 x 4 entries
//     per wave, 992 per tile, of which the first 32 repeat the previous tile's tail (far pairs across the seam).
//   * THE EPILOGUE STAYS IN REGISTERS: the four entries of a thread are finished with straight-line code (code
//     byte from a select chain over the masks, signature, directory, fingerprint, request), the requests of a
//     wave get their queue slots from one LDS atomic per wave + mbcnt.
//
// Entries whose window block goes on past distance 3 (~6 % on the diploid table) are NOT finished inside their tile
// any more (round 2 did: a tail phase by one wave while three waited, hand-over words in LDS, a merge phase and two
// more workgroup barriers per tile: 2.9 of 15.6 ms).  Their owners queue their slots; after the tile one dense pass
// (a lane per queued entry, partners from the staged LDS copy) only DETECTS whether such an entry has a pair at
// distance 4..30 or a block beyond 30 -- rare: 0.8 % of the pairs of the diploid table -- and sets the bits of the
// entry and of its partner in a one-bit-per-entry map; kf_bigfix recomputes the marked entries exactly after the
// launch (code byte, map bit, request).  Until then their code bytes are the provisional ones of the register scan:
// any request or map bit those caused is a superset of what the final state needs.
// (Collecting the queued entries over several tiles and detecting 256 at a time from global memory was tried first:
//  the passes -- 18 dependent-latency loads per entry -- cost 3.2 ms.)
//
// The fingerprint mixer is two dependent 32x32->64 multiply-adds (v_mad_u64_u32 is full rate on gfx950) and half a
// ChaCha quarter round: the avalanche of the three quarter rounds of round 1 to within +-0.03 (tests/test_mixer.py), 15
// instead of 39 instructions.
//
// Semantics are those of kf_pass1_s / kf_pass1_r (smg_fast.hpp: code byte, request protocol), including the
// relaxation that an entry with >= 2 pairs always sends a request.

#pragma once
#include "smg_fast.hpp"

#define D_TPB    256
#define D_WL     62                        // scanned lanes per wave (lanes 1..62; 0 and 63 are halo lanes)
#define D_SCAN   (4 * 4 * D_WL)            // 992 scanned entries per tile
#define D_HALO   32                        // the first 32 of them are owned by the previous tile
#define D_OWN    (D_SCAN - D_HALO)         // 960
#define D_SLOTS  (D_SCAN + 8)              // staged entries: scanned + the two outer halo threads
#define D_LEAD   (D_HALO + 4)              // staged entries in front of the first owned one
#define D_WIN    30                        // partners are searched within +-30 entries
#ifndef D_BMF
#define D_BMF    4096                      // block ids per tile with a bit in LDS (request filter)
#endif
#define D_BMW    (D_BMF / 32)
#define D_HB     1024                      // bins of the request histogram (smg_lookup.hpp: L_BK)
#ifndef D_QCAP
#define D_QCAP   1792                      // LDS request queue (records); flushed when the next tile might not fit (every ~5th tile;
                                           //   1280 records: every 2nd or 3rd, +0.3 ms)
#endif
// Scheduling fences (nothing is scheduled across them).  They were put between the tests and between the phases when
// the machine scheduler hoisted every compare to the front and the kernel spilled lane masks (SGPR pairs) by the dozen;
// with today's kernel only the ones between the entries of the complement loop still pay (they keep its four entries
// from being interleaved: ~30 vector registers): without the other two classes pass 1 takes 15.7 instead of 16.2 ms
// (profiles/r02_pass1_ablation.txt, k).
#ifndef D_NOFENCE
#define D_NOFENCE 3                        // bit mask of fence classes LEFT OUT: 1 between tests, 2 between phases, 4 between the entries of the complement loop
#endif
#define D_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define D_FENCE_T() do { if (!(D_NOFENCE & 1)) D_SCHED_FENCE(); } while (0)
#define D_FENCE_P() do { if (!(D_NOFENCE & 2)) D_SCHED_FENCE(); } while (0)
#define D_FENCE_R() do { if (!(D_NOFENCE & 4)) D_SCHED_FENCE(); } while (0)
#ifndef D_RC_WIDE
#define D_RC_WIDE 1
#endif
#ifndef D_TAILB
#define D_TAILB  6                         // tail: the next D_TAILB partners of a queued entry (distances 4 .. D_TAILB + 3) are read in one batch, the
                                           //   (rare) rest one by one.  8 until round 6: 4 / 5 / 6 are 0.2-0.25 ms faster on the bench table (most queued
                                           //   blocks end within two or three entries), 2 and 12 slower (profiles/r06_pass1_experiments.txt)
#endif
#define D_RD     3                         // distances tested register-to-register; the deferred tail starts at D_RD + 1 (a fourth
                                           //   distance in registers: ten vector registers spill, 16.4 instead of 14.7 ms)
#ifndef D_ABL
#define D_ABL    0                         // ablation mask (timing experiments only; results are WRONG when non-zero):
#endif                                     //   1 fingerprint, 2 directory, 4 signatures, 8 block map, 16 requests, 32 tail, 64 scan,
                                           //   512 tail detection passes, 4096/8192 the two barriers of a tile

// (Round 6 swizzled the staged k-mers -- every other group of eight threads swapped the 16-byte halves of its 32 bytes, so that
//  the sixteen lanes of an LDS cycle cover all 64 banks: the two-way conflicts of the 32-byte lane stride were 41 % of the
//  LDS-active cycles in round 5.  Conflict-free, correct, and no faster: 11.14-11.17 against 11.08-11.10 ms,
//  profiles/r06_pass1_experiments.txt -- the kernel does not wait for its LDS.  Taken out again.)
template <int W> struct DWord;
template <> struct DWord<1> { typedef unsigned type; };
template <> struct DWord<2> { typedef u64 type; };

struct GeoR
{ int  k;
  int  pshift;       // W=1: pre = hi32 >> pshift (32 - 2*p0; 32 means "no prefix": k == 1);  W=2: pre = w0 >> pshift (64 - 2*p0)
  int  kshift;       // W=1: 64 - 2k;  W=2: 128 - 2k  (the k-mer is left aligned in W words)
  u64  smask;        // low 2*(k-p0) bits
  int  mshift;       // odd k: suffix >> mshift != 0  <=> the top suffix base (position p0) differs
};

SMG_DEV int d_popc(unsigned v) { return __popc(v); }
SMG_DEV int d_popc(u64 v) { return __popcll(v); }

// value of the right neighbour lane (lane 63 reads 0)
SMG_DEV unsigned d_next(unsigned v) { return (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x130, 0xf, 0xf, false); }   // wave_shl:1
SMG_DEV u64 d_next(u64 v) { return (u64) d_next((unsigned) v) | ((u64) d_next((unsigned) (v >> 32)) << 32); }

SMG_DEV bool d_lane(u64 mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }       // scalar mask -> per-lane predicate, free

// One lane of a wave adds to an LDS counter and the wave takes the old value.  Written out: for `if (lane == 0) atomicAdd(..)`
// the compiler's atomic optimiser cannot see that one lane is active and builds its general sequence around the atomic
// (two mbcnt, a compare, a saveexec, a scalar population count and multiply, a multiply-add to hand every lane its share:
// ~12 instructions for each of the two wave-aggregated atomics of a tile).  LDS operations return in order and the wait
// is for all of them, so the compiler's own counting stays valid.
SMG_DEV unsigned d_wave_add(unsigned *ctr, unsigned v, int lane)
{ unsigned old = 0;
  if (lane == 0)
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"((unsigned) (uintptr_t) ctr), "v"(v) : "memory");
  return (unsigned) __builtin_amdgcn_readfirstlane((int) old);
}

template <int W, bool KF> SMG_DEV void
d_unpack(const Key<W> &x, const GeoR &G, typename DWord<W>::type &pre, typename DWord<W>::type &suf)
{ if constexpr (W == 1)
    { const unsigned hi = (unsigned) (x.w[0] >> 32), lo = (unsigned) x.w[0];
      if (KF)                               // 17 <= k <= 32: the k-mer straddles both 32-bit halves, no selects
        { pre = hi >> G.pshift;
          suf = __builtin_amdgcn_alignbit(hi, lo, G.kshift) & (unsigned) G.smask;
        }
      else
        { pre = G.pshift < 32 ? hi >> G.pshift : 0u;
          suf = (G.kshift >= 32 ? hi >> (G.kshift - 32) : __builtin_amdgcn_alignbit(hi, lo, G.kshift)) & (unsigned) G.smask;
        }
    }
  else
    { pre = x.w[0] >> G.pshift;                                     // pshift = 64 - 2*p0 in 0..32
      suf = (G.kshift ? (x.w[1] >> G.kshift) | (x.w[0] << (64 - G.kshift)) : x.w[1]) & G.smask;
    }
}

// 128-bit mixing of (k-mer, count): two dependent 32x32+64 multiply-adds, then the first half of a ChaCha quarter round over
// (q, p ^ count).  Every input bit flips every output bit with probability 0.5 +- 0.03 (tests/test_mixer.py).
SMG_DEV u64 d_mad(unsigned a, unsigned b, u64 c) { return (u64) a * (u64) b + c; }       // v_mad_u64_u32

template <int W> SMG_DEV void mix_hash(const Key<W> &x, unsigned cnt, u64 &ha, u64 &hb)
{ const unsigned lo = (unsigned) x.w[0], hi = (unsigned) (x.w[0] >> 32);
  u64 p = d_mad(lo ^ 0x9E3779B9u, hi ^ 0x85EBCA6Bu, (u64) (cnt << 20));
  if constexpr (W == 2)                     // absorb the second word
    { const unsigned l1 = (unsigned) x.w[1], h1 = (unsigned) (x.w[1] >> 32);
      p = d_mad(l1 ^ (unsigned) p ^ 0x165667B1u, h1 ^ (unsigned) (p >> 32) ^ 0xD3A2646Cu, p);
    }
  const unsigned pl = (unsigned) p, ph = (unsigned) (p >> 32);
  const u64 q = d_mad(pl ^ hi, ph ^ lo ^ 0xC2B2AE35u, p);
  unsigned a = (unsigned) q, b = (unsigned) (q >> 32), c = pl ^ cnt, d = ph;
  // the first half of a ChaCha quarter round (round 6; rounds 2-5 ran the whole one): every input bit still flips every output
  // bit with probability 0.5 +- 0.03 (tests/test_mixer.py), six vector instructions less per entry: -0.3 ms on the bench table
  a += b; d ^= a; d = __builtin_rotateleft32(d, 16);
  c += d; b ^= c; b = __builtin_rotateleft32(b, 12);
  ha = (u64) a | ((u64) b << 32);
  hb = (u64) c | ((u64) d << 32);
}

// the 4 entries of a thread, loaded one tile ahead
template <int W> struct DPrefetch
{ Key<W> k[4]; ushort4 c; bool valid;
  uint32_t anchor;               // leading 32 bits of the tile's first owned entry (the base of the tile's block-map window)
  SMG_DEV void load(const u64 *__restrict__ keys, const uint16_t *__restrict__ cnt, int64_t i0, int64_t ianchor)
  { anchor = (uint32_t) (keys[ianchor * W] >> 32);
    if constexpr (W == 1)
      { const ulonglong2 v0 = *reinterpret_cast<const ulonglong2 *>(keys + i0);
        const ulonglong2 v1 = *reinterpret_cast<const ulonglong2 *>(keys + i0 + 2);
        k[0].w[0] = v0.x; k[1].w[0] = v0.y; k[2].w[0] = v1.x; k[3].w[0] = v1.y;
      }
    else
      {
#pragma unroll
        for (int e = 0; e < 4; e++) k[e] = load_key<W>(keys, i0 + e);
      }
    c = *reinterpret_cast<const ushort4 *>(cnt + i0);
  }
  // "the registers are needed HERE": makes the compiler place its wait for the loads at this point
  SMG_DEV void arrive()
  {
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
      for (int w = 0; w < W; w++) asm volatile("" : "+v"(k[e].w[w]));
    unsigned c01 = (unsigned) c.x | ((unsigned) c.y << 16), c23 = (unsigned) c.z | ((unsigned) c.w << 16);
    asm volatile("" : "+v"(c01), "+v"(c23), "+v"(anchor));
    c.x = (unsigned short) c01; c.y = (unsigned short) (c01 >> 16); c.z = (unsigned short) c23; c.w = (unsigned short) (c23 >> 16);
  }
};

struct P1Hot                              // kernel argument: what every tile touches (stays in SGPRs)
{ const u64      *keys;
  const uint16_t *cnt;
  int64_t         n;
  uint8_t        *code;
  uint16_t       *sig;           // sig[i] = (uint16_t) (keys[i] >> sigsh): the 16 k-mer bits below the directory's bucket bits
  uint32_t       *bstart;        // bucket directory: bucket(x) = (hi32(x) >> dsh) - b0; NULL: the table came with its prefix index, none is built
  uint32_t       *bmap;          // candidate block map (or NULL): bit (hi32(x) >> bmsh)
  uint32_t        b0, nb;
  unsigned        shifts;        // dsh | sigsh << 6 | bmsh << 12 | emit_all << 18 | want_fp << 19 | hbits << 20 | two << 24  (one register)
  GeoR            G;
  int64_t         ntiles;
  SMG_DEV int dsh() const { return (int) (shifts & 63u); }
  SMG_DEV int sigsh() const { return (int) ((shifts >> 6) & 63u); }
  SMG_DEV int bmsh() const { return (int) ((shifts >> 12) & 31u); }
  SMG_DEV bool emit_all() const { return (shifts >> 18 & 1u) != 0; }
  SMG_DEV bool want_fp() const { return (shifts >> 19 & 1u) != 0; }
  SMG_DEV int hbits() const { return (int) ((shifts >> 20) & 15u); }    // request histogram on the leading hbits bits (0: none)
  SMG_DEV bool two() const { return (shifts >> 24 & 1u) != 0; }         // two-bit block map (BM2_* in smg_fast.hpp)
};

struct P1Cold                             // in device memory: what only a flush touches (kept out of the register file)
{ u64      *req;
  uint32_t *chunk_fill;
  u64      *partials;
  FastCtl  *ctl;
  unsigned *whist;               // look-up chain: requests of workgroup w per bucket (leading hbits bits of the target)
                                 //   in row w: whist[w * D_HB + b]
  unsigned  owners;              //   chunk slots are dealt out without a counter: owner w fills w, w + owners, w + 2 owners, ..
  unsigned  max_chunks;
  uint32_t *dbits;               // deferred entries: one bit per table entry (kf_bigfix redoes them exactly and clears the bits)
  u64      *times;               // SMG_P1_TIMES (tuning): start / end of every workgroup on the constant 100 MHz clock, or NULL
  unsigned *tick;                // tile tickets: D_NCLS counters, 128 bytes apart (zeroed before the launch)
};

struct DShared                            // the workgroup's LDS arrays (pointers: the tile body is a function)
{ uint16_t *tailq;                       // deferred tail: slots of this tile's entries whose block goes on past distance 3
  u64 *ent; uint16_t *lcn; u64 *sq;
  unsigned *s_tn, *s_qn, *s_unsorted;
  unsigned *bm;                          // candidate-block bits of this tile: D_BMW words (2 * D_BMW: two-bit map)
  unsigned *hist;                        // requests of this workgroup per bucket (D_HB bins), for the look-up chain's partition
};

// The 12 one-away tests of a thread (distances 1..3; entries 4..6 are the right neighbour's 0..2), aggregated on the
// fly.  Straight-line code: every test is a handful of instructions whose result mask dies at once.
template <typename WT, bool ODD, bool CHECK> SMG_DEV void
d_tests(const WT (&sx)[4 + D_RD], const unsigned (&cn)[4], const u64 (&Sm)[4 + D_RD], const GeoR &G,
        unsigned (&code)[4], unsigned (&npair)[4], u64 (&midM)[4])
{ const WT AA = (WT) 0xAAAAAAAAAAAAAAAAull;
  unsigned cx[8] = { cn[0], cn[1], cn[2], cn[3], 0, 0, 0, 0 };
  if (CHECK)
    {
#pragma unroll
      for (int e = 0; e < D_RD; e++) cx[4 + e] = d_next(cn[e]);
    }
  // odd k: a pair sits on the self-mirrored position (the top suffix base) iff the one differing 2-bit group is the
  // top one: tt >= TOPB (one more compare per test; keeping "top bases of e and e+1 differ" masks instead costs
  // twelve more scalar registers than the kernel has)
  const WT TOPB = (WT) 2 << (ODD ? G.mshift : 0);
#pragma unroll
  for (int a = 3; a >= 0; a--)               // descending: the neighbour's masks (indices 4..6) die first
    {
#pragma unroll
      for (int d = 1; d <= D_RD; d++)
        { const int b = a + d, eb = b & 3;
          const WT dd = sx[a] ^ sx[b];
          const WT tt = ((dd << 1) | dd) & AA;
          u64 h = __ballot(d_popc(tt) == 1);
          h &= Sm[a];
          if (d >= 2) h &= Sm[a + 1];
          if (d >= 3) h &= Sm[a + 2];
          if (CHECK) h &= __ballot(cx[a] + cx[b] <= SMG_SMAX);
          const u64 hb = b < 4 ? h : h << 1;
          npair[a] += d_lane(h) ? 1u : 0u;
          npair[eb] += d_lane(hb) ? 1u : 0u;
          if (ODD)
            { const u64 hm = h & __ballot(tt >= TOPB);
              midM[a] |= hm;
              midM[eb] |= b < 4 ? hm : hm << 1;
            }
          const unsigned w2c = ODD ? 0u : (unsigned) CODE_W2;
          code[a] = d_lane(h) ? ((unsigned) (31 + d) | w2c) : code[a];
          code[eb] = d_lane(hb) ? ((unsigned) (31 - d) | w2c) : code[eb];
          D_FENCE_T();
        }
    }
}

// queue slots for the requests of a wave: E[e] = lanes whose entry e sends; one LDS atomic per wave
template <int W, int RW> SMG_DEV void
d_emit(const DShared &S, const u64 (&E)[4], const Key<W> (&rc)[4], const unsigned (&cn)[4], const u64 (&hiM)[4], int lane)
{ const unsigned cnt_w = (unsigned) (__popcll(E[0]) + __popcll(E[1]) + __popcll(E[2]) + __popcll(E[3]));
  if (cnt_w == 0) return;
  unsigned base = 0;
  if (lane == 0) base = atomicAdd(S.s_qn, cnt_w);
  base = (unsigned) __builtin_amdgcn_readfirstlane((int) base);
#pragma unroll
  for (int e = 0; e < 4; e++)
    { if (d_lane(E[e]))
        { const unsigned q = __builtin_amdgcn_mbcnt_hi((unsigned) (E[e] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) E[e], base));
#pragma unroll
          for (int w = 0; w < W; w++) S.sq[q * RW + w] = rc[e].w[w];
          if (RW > W) S.sq[q * RW + W] = (u64) cn[e] | (d_lane(hiM[e]) ? 1ull << 16 : 0ull);
        }
      base += (unsigned) __popcll(E[e]);
    }
}

// code byte -> "owns a pair at p > k-1-p" (several pairs, or one that is not self-mirrored) / "exactly one pair";
// CODE_DEFER (0xFF) is neither
SMG_DEV bool d_code_hi(unsigned c) { return c - 63u < 65u; }                    // 63 .. 127
SMG_DEV bool d_code_uq(unsigned c) { return ((c & 63u) - 1u) < 62u && c < 128u; }

// One tile.  INNER tiles lie completely inside the table: vector loads, no bounds checks, no table-end cases.
// RW = 64-bit words per request record: W (the complement k-mer) or W + 1 (+ count | has-hi-pair << 16).
template <int W, int RW, bool ODD, bool KF, bool INNER, int VAR> SMG_DEV void
d_tile(const P1Hot &A, const DShared &S, int64_t g0, int64_t g0_next, int t, u64 &fa, u64 &fb, DPrefetch<W> &pf)
{ typedef typename DWord<W>::type WT;
  constexpr bool D_BM = (W == 1 && RW == 1) || (W == 2 && RW != 1);     // variants that feed the request filter
  constexpr bool DIR = (VAR & 1) != 0;       // this launch writes a directory and / or signatures
  const GeoR &G = A.G;
  const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);       // the wave number is uniform: keep it scalar
  const int slot0 = (wv * D_WL + lane) * 4;
  const int64_t i0 = g0 + slot0;
  const int64_t n = A.n;
  const u64 scanM = 0x7FFFFFFFFFFFFFFEull;                   // lanes 1..62
  const u64 ownM = wv == 0 ? scanM & ~((1ull << (D_LEAD / 4)) - 1ull) : scanM;
  const bool owned = d_lane(ownM);

  // ---- loads ----------------------------------------------------------------------------------------------
  //@mark D_LOAD
  Key<W> kk[4]; unsigned cn[4];
  uint32_t pf_anchor = 0;
  unsigned vmask = 0xF;                    // entries i0 .. i0+3 inside the table?
  if (INNER)
    { // (the caller has loaded them if the tile before this one did not: the first tile of a workgroup, or the one after an
      //  edge tile)
#pragma unroll
      for (int e = 0; e < 4; e++) kk[e] = pf.k[e];
      cn[0] = pf.c.x; cn[1] = pf.c.y; cn[2] = pf.c.z; cn[3] = pf.c.w;
      pf_anchor = (uint32_t) __builtin_amdgcn_readfirstlane((int) pf.anchor);
    }
  else
    { vmask = 0;
#pragma unroll
      for (int e = 0; e < 4; e++)
        { const int64_t i = i0 + e;
          const bool ok = i >= 0 && i < n;
          vmask |= (unsigned) ok << e;
          const Key<W> kx = load_key<W>(A.keys, ok ? i : 0);
#pragma unroll
          for (int w = 0; w < W; w++) kk[e].w[w] = ok ? kx.w[w] : 0ull;
          cn[e] = ok ? (unsigned) A.cnt[i] : 0xFFFFu;
        }
    }
  // LDS copy: the complement loop re-reads the thread's own entries from it (cheaper than four live registers), the
  // dense block-map loop after the tile reads the candidates' k-mers
  if constexpr (W == 1)
    { ulonglong2 w0, w1;
      w0.x = kk[0].w[0]; w0.y = kk[1].w[0]; w1.x = kk[2].w[0]; w1.y = kk[3].w[0];
      *reinterpret_cast<ulonglong2 *>(&S.ent[slot0]) = w0;
      *reinterpret_cast<ulonglong2 *>(&S.ent[slot0 + 2]) = w1;
    }
  else
    {
#pragma unroll
      for (int r = 0; r < 4; r++)
        { ulonglong2 v; v.x = kk[r].w[0]; v.y = kk[r].w[1];
          *reinterpret_cast<ulonglong2 *>(&S.ent[(slot0 + r) * W]) = v;
        }
    }
  *reinterpret_cast<ushort4 *>(&S.lcn[slot0]) = make_ushort4((unsigned short) cn[0], (unsigned short) cn[1],
                                                           (unsigned short) cn[2], (unsigned short) cn[3]);

  //@mark D_UNPACK
  WT pre[4], sx[4 + D_RD];                 // sx[4..]: the suffixes of the right neighbour's entries 0..D_RD-1
#pragma unroll
  for (int e = 0; e < 4; e++) d_unpack<W, KF>(kk[e], G, pre[e], sx[e]);
  const WT npre0 = d_next(pre[0]);
#pragma unroll
  for (int e = 0; e < D_RD; e++) sx[4 + e] = d_next(sx[e]);

  // entry e exists (edge tiles only; V[4..6] are the neighbour's)
  u64 V[7];
  if (!INNER)
    {
#pragma unroll
      for (int e = 0; e < 4; e++) V[e] = __ballot((vmask >> e) & 1u);
      V[4] = V[0] >> 1; V[5] = V[1] >> 1; V[6] = V[2] >> 1;
    }

  // ---- order check + bucket directory (the first entry of every bucket stores its index) ------------------
  //@mark D_DIR
  D_FENCE_P();
  if (!(D_ABL & 2))
    { Key<W> nk0;
#pragma unroll
      for (int w = 0; w < W; w++) nk0.w[w] = d_next(kk[0].w[w]);
      const Key<W> nxt[4] = { kk[1], kk[2], kk[3], nk0 };
      u64 bad = 0;
#pragma unroll
      for (int e = 0; e < 4; e++)
        { u64 b = __ballot(!key_lt<W>(kk[e], nxt[e]));
          if (!INNER) b &= V[e] & V[e + 1];
          bad |= b;
        }
      if (bad & scanM) *S.s_unsorted = 1u;
      // raw bucket numbers; the offset and the bound are applied on the (rare) store path only
      // (no directory is written when the table came with its own: the FastK prefix index, smg_engine_set_prefix_index)
      if (DIR && A.bstart != nullptr) {
      uint32_t bq[5];
#pragma unroll
      for (int e = 0; e < 4; e++) bq[e] = (uint32_t) (kk[e].w[0] >> 32) >> A.dsh();
      bq[4] = (uint32_t) (nk0.w[0] >> 32) >> A.dsh();
      if (owned)
        { if (!INNER && i0 == 0) A.bstart[0] = 0u;             // (the first bucket of the shard is bucket b0)
#pragma unroll
          for (int e = 0; e < 4; e++)
            { bool st = bq[e + 1] != bq[e];
              if (!INNER) st = st && ((vmask >> e) & 1u) && i0 + e + 1 < n;
              if (st)
                { const uint32_t br = bq[e + 1] - A.b0;
                  if (br < A.nb) A.bstart[br] = (uint32_t) (i0 + e + 1);
                }
              if (!INNER && ((vmask >> e) & 1u) && i0 + e + 1 == n) A.bstart[A.nb] = (uint32_t) n;
            }
        }
      }
    }

  // ---- signatures ---------------------------------------------------------------------------------------------------
  //@mark D_SIG
  D_FENCE_P();
  if (DIR && W <= 2 && !(D_ABL & 4) && A.sig && owned)     // (no signatures: the look-ups bisect the k-mers themselves)
    { unsigned sg[4];
#pragma unroll
      for (int e = 0; e < 4; e++)
        sg[e] = (unsigned) (kk[e].w[0] >> A.sigsh()) & 0xFFFFu;            // v_lshrrev_b64 is full rate
      if (INNER || vmask == 0xF)
        *reinterpret_cast<uint2 *>(A.sig + i0) = make_uint2(sg[0] | (sg[1] << 16), sg[2] | (sg[3] << 16));
      else
        for (int e = 0; e < 4; e++)
          if (vmask >> e & 1) A.sig[i0 + e] = (uint16_t) sg[e];
    }
  const int bmsh = A.bmsh();             // (the k-mers themselves are not needed past this point)
  (void) bmsh;

  // ---- window-block structure as lane masks ---------------------------------------------------------------
  //@mark D_MASKS
  D_FENCE_P();
  // Sm[e]: entries e and e+1 share their first p0 bases (e = 4..6: the neighbour's 0..2)
  u64 Sm[4 + D_RD];
#pragma unroll
  for (int e = 0; e < 3; e++) Sm[e] = __ballot(pre[e] == pre[e + 1]);
  Sm[3] = __ballot(pre[3] == npre0);
  if (!INNER)
    {
#pragma unroll
      for (int e = 0; e < 4; e++) Sm[e] &= V[e] & V[e + 1];
    }
#pragma unroll
  for (int e = 0; e < D_RD; e++) Sm[4 + e] = Sm[e] >> 1;

  // entries whose block continues past distance 3: deferred (their owner queues their slot)
  if (!(D_ABL & 32))
    { u64 Al[4];
#pragma unroll
      for (int e = 0; e < 4; e++) Al[e] = Sm[e] & Sm[e + 1] & Sm[e + 2] & Sm[e + 3] & ownM;
      const unsigned na = (unsigned) (__popcll(Al[0]) + __popcll(Al[1]) + __popcll(Al[2]) + __popcll(Al[3]));
      if (na)
        { unsigned base = 0;
          base = d_wave_add(S.s_tn, na, lane);
#pragma unroll
          for (int e = 0; e < 4; e++)
            { if (d_lane(Al[e]))
                { const unsigned q = __builtin_amdgcn_mbcnt_hi((unsigned) (Al[e] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) Al[e], base));
                  S.tailq[q] = (uint16_t) (slot0 + e);
                }
              base += (unsigned) __popcll(Al[e]);
            }
        }
    }

  // ---- the 12 one-away tests of a thread (distances 1..3), aggregated on the fly -----------------------------
  //@mark D_TESTS
  D_FENCE_P();
  // per entry e of a lane: npair[e] pairs seen (a carry-in add per hit mask), code[e]: delta code of the last pair seen
  // (the only one if the entry is unique), midM "has a pair on the self-mirrored position".  The b side of a test
  // with b >= 4 is entry b - 4 of the right neighbour lane: the same mask, shifted up by one lane.
  unsigned code[4] = { CODE_NONE, CODE_NONE, CODE_NONE, CODE_NONE };
  unsigned npair[4] = { 0, 0, 0, 0 };
  u64 midM[4] = { 0, 0, 0, 0 };
  if (!(D_ABL & 64))
    { unsigned mx = cn[0] > cn[1] ? cn[0] : cn[1];
      { const unsigned m2 = cn[2] > cn[3] ? cn[2] : cn[3]; mx = mx > m2 ? mx : m2; }
      // count sums can only exceed 1000 next to a count > 500: one wave-uniform branch, two straight-line variants
      if (__ballot(mx > SMG_FMAX) == 0) d_tests<WT, ODD, false>(sx, cn, Sm, G, code, npair, midM);
      else                             d_tests<WT, ODD, true>(sx, cn, Sm, G, code, npair, midM);
    }
  u64 uniqM[4], hiM[4];
#pragma unroll
  for (int e = 0; e < 4; e++)
    { const u64 mulM = __ballot(npair[e] >= 2u);
      code[e] = d_lane(mulM) ? (unsigned) CODE_MULTI : code[e];
      uniqM[e] = __ballot(npair[e] == 1u);
      if (ODD)
        { const u64 w2 = uniqM[e] & ~midM[e];
          code[e] = d_lane(w2) ? (code[e] | (unsigned) CODE_W2) : code[e];
          hiM[e] = mulM | w2;                          // owns a pair at p > k-1-p
        }
      else hiM[e] = mulM | uniqM[e];
    }

  // ---- request filter: a CANDIDATE (exactly one suffix-side pair) sets the bit of its block id -------------------
  //@mark D_BMAP
  D_FENCE_P();
  // word 0 of the tile's LDS bit map = the map word of the tile's first owned entry (a uniform, scalar load)
  // (inner tiles: loaded one tile ahead with the entries -- as a load of its own it was waited for on the spot)
  const uint32_t bmbase = D_BM ? (((INNER ? pf_anchor : (uint32_t) (A.keys[(g0 + D_LEAD) * W] >> 32)) >> bmsh) & ~31u) : 0u;
  if (D_BM && A.bmap && !(D_ABL & 8))
    { // leading word of the thread's own entries, back from the staged copy (cheaper than four registers kept alive
      // across the tests)
      u64 kw[4];
      if constexpr (W == 1)
        { const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&S.ent[slot0]);
          const ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(&S.ent[slot0 + 2]);
          kw[0] = a.x; kw[1] = a.y; kw[2] = b.x; kw[3] = b.y;
        }
      else
        {
#pragma unroll
          for (int e = 0; e < 4; e++) kw[e] = S.ent[(slot0 + e) * W];
        }
      if ((VAR & 2) || A.two())
        { u64 *bm64 = reinterpret_cast<u64 *>(S.bm);
          u64 *gm64 = reinterpret_cast<u64 *>(A.bmap);
#pragma unroll
          for (int e = 0; e < 4; e++)
            { u64 cm = uniqM[e] & ownM;
              if (!INNER) cm &= V[e];
              if (cm)
                { const uint32_t id = (uint32_t) (kw[e] >> 32) >> bmsh;
                  const uint32_t rel = id - bmbase;
                  const u64 v = bm2_bits(id, (uint32_t) kw[e]);
                  const u64 nearM = __ballot(rel < D_BMF) & cm;
                  if (d_lane(nearM)) atomicOr(&bm64[rel >> 5], v);
                  const u64 farM = cm & ~nearM;
                  if (farM) { if (d_lane(farM)) atomicOr(&gm64[id >> 5], v); }
                }
            }
        }
      else
        {
#pragma unroll
          for (int e = 0; e < 4; e++)
            { u64 cm = uniqM[e] & ownM;
              if (!INNER) cm &= V[e];
              if (cm)
                { const uint32_t id = (uint32_t) (kw[e] >> 32) >> bmsh;
                  const uint32_t rel = id - bmbase;
                  const u64 nearM = __ballot(rel < D_BMF) & cm;
                  if (d_lane(nearM)) atomicOr(&S.bm[rel >> 5], 1u << (rel & 31));
                  const u64 farM = cm & ~nearM;                                      // sparse table: outside the tile's LDS window
                  if (farM) { if (d_lane(farM)) atomicOr(&A.bmap[id >> 5], 1u << (id & 31)); }
                }
            }
        }
    }

  // ---- complement, fingerprint, requests ---------------------------------------------------------------------------
  unsigned codes = code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24);

  // ---- complement, fingerprint, requests: one entry at a time from the thread's own LDS copy ----------------------
  //@mark D_RC
  D_FENCE_P();
  if (!(D_ABL & 16) || (VAR & 2) || A.want_fp())
    { // hash proof: rc(x) of every owned entry that owns a pair at p > k-1-p; exact proof: of every owned entry, with
      // that flag.  (A deferred entry that turns out to own more pairs than the register scan saw sends again from
      // kf_bigfix: the flag of a request is only ever ORed into its target.)
      u64 E0 = 0, E1 = 0, E2 = 0, E3 = 0;
      if (!(D_ABL & 16))
        { const u64 all = (!(VAR & 2) && A.emit_all()) ? ~0ull : 0ull;       // (VAR & 2: the hash proof -- owners of a hi-side pair only)
          E0 = (hiM[0] | all) & ownM; E1 = (hiM[1] | all) & ownM; E2 = (hiM[2] | all) & ownM; E3 = (hiM[3] | all) & ownM;
          if (!INNER) { E0 &= V[0]; E1 &= V[1]; E2 &= V[2]; E3 &= V[3]; }
        }
      const unsigned cnt_w = (unsigned) (__popcll(E0) + __popcll(E1) + __popcll(E2) + __popcll(E3));
      unsigned base = 0;
      if (cnt_w)
        base = d_wave_add(S.s_qn, cnt_w, lane);
      const bool fp = ((VAR & 2) || A.want_fp()) && !(D_ABL & 1);
      // Unrolled, the four entries kept apart by scheduling fences (interleaved they need ~30 more vector registers
      // than the kernel has).  The rolled loop -- the masks rotating through one register pair, a counter, two branches
      // per entry -- cost 0.7 ms more: scalar instructions and branches are not free next to a busy vector unit
      // (tools/issue_mix.hip).
      const u64 EM[4] = { E0, E1, E2, E3 };
#if D_RC_WIDE
      // the thread's four entries in two 16-byte LDS reads + one 8-byte read of the counts (four 8-byte reads at a lane
      // stride of 32 bytes hit the same banks from eight lanes at a time)
      Key<W> xs[4]; unsigned cs[4];
      if constexpr (W == 1)
        { const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&S.ent[slot0]);
          const ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(&S.ent[slot0 + 2]);
          xs[0].w[0] = a.x; xs[1].w[0] = a.y; xs[2].w[0] = b.x; xs[3].w[0] = b.y;
        }
      else
        {
#pragma unroll
          for (int e = 0; e < 4; e++) xs[e] = lds_key<W>(S.ent, slot0 + e);
        }
      { const ushort4 c4 = *reinterpret_cast<const ushort4 *>(&S.lcn[slot0]); cs[0] = c4.x; cs[1] = c4.y; cs[2] = c4.z; cs[3] = c4.w; }
#endif
#pragma unroll
      for (int e = 0; e < 4; e++)
        { E0 = EM[e];
          D_FENCE_R();
#if D_RC_WIDE
          const Key<W> x = xs[e];
          const unsigned c = cs[e];
#else
          const Key<W> x = lds_key<W>(S.ent, slot0 + e);
          const unsigned c = S.lcn[slot0 + e];
#endif
          const Key<W> rc = revcomp<W>(x, G.k);
          if (fp && owned)
            { // XOR of h(min(x, rc x), count) over the entries: the two members of a closed class cancel, an entry
              // without its complement (or with another count) leaves its 128 bits behind -- no signs, no carries
              const bool lt = key_lt<W>(x, rc);
              u64 ha, hb;
              mix_hash<W>(lt ? x : rc, c, ha, hb);
              if (ODD && INNER) { fa ^= ha; fb ^= hb; }              // odd k: no k-mer is its own complement
              else
                { u64 keep = ~0ull;
                  if (!ODD) keep = key_eq<W>(x, rc) ? 0ull : ~0ull;  // self-complementary: occurs once, no term
                  if (!INNER) keep = ((vmask >> e) & 1u) ? keep : 0ull;
                  fa ^= ha & keep; fb ^= hb & keep;
                }
            }
          //@mark D_EMIT
          if (E0)
            { if (d_lane(E0))
                { const unsigned q = __builtin_amdgcn_mbcnt_hi((unsigned) (E0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) E0, base));
#pragma unroll
                  for (int w = 0; w < W; w++) S.sq[q * RW + w] = rc.w[w];
                  if (RW > W) S.sq[q * RW + W] = (u64) c | (d_lane(hiM[e]) ? 1ull << 16 : 0ull);
                }
              base += (unsigned) __popcll(E0);
            }
        }
    }
  // the next tile's entries: issued here, used after the flush -- the latency of the loads (a few thousand cycles on a
  // busy chip) disappears behind the flush phase and the barriers instead of stalling the head of the next tile
  pf.valid = false;
  if (g0_next >= 0) { pf.load(A.keys, A.cnt, g0_next + slot0, g0_next + D_LEAD); pf.valid = true; }

  //@mark D_STORE
  if (owned)
    { if (INNER || vmask == 0xF) *reinterpret_cast<unsigned *>(A.code + i0) = codes;
      else
        for (int e = 0; e < 4; e++)
          if (vmask >> e & 1) A.code[i0 + e] = (uint8_t) (codes >> (8 * e));
    }
}

// ---- deferred tail: does the entry in slot sa own a pair at distance 4..30, or does its block go on past 30? -------
// hm: bit d - (D_RD + 1) set for a pair with the entry d slots on; big: the entry 31 slots on still shares the prefix (block
// beyond the window: both are redone by bisection, as every entry of such a block is in one of these two roles).
// Partners come from the staged copy (global memory past its end: the last few slots of a tile).
// The COUNTS are not looked at (round 6): a one-away partner whose count sum exceeds 1000 marks the two entries as well, and
// kf_bigfix, which redoes a marked entry exactly, finds that it owns no such pair -- a superset costs a few entries of a table
// with counts above 500 an exact redo, and every queued entry of every table seven LDS reads and a dozen instructions less.
template <int W, bool ODD, bool KF> SMG_DEV void
d_detect(const P1Hot &A, const u64 *ent, const uint16_t *lcn, int64_t g0, int sa, unsigned &hm, bool &big)
{ typedef typename DWord<W>::type WT;
  const GeoR &G = A.G;
  const int64_t n = A.n;
  hm = 0; big = false;
  // (slots of the tile that lie inside the table, as a 32-bit number: one compare per partner instead of a 64-bit add and compare)
  const int nrel = n - g0 < (int64_t) (1 << 30) ? (int) (n - g0) : (1 << 30);
  WT pa, sfa;
  d_unpack<W, KF>(lds_key<W>(ent, sa), G, pa, sfa);
  (void) lcn;
  constexpr int D0 = D_RD + 1;                      // first distance the register scan did not cover
  int d = D0;
  if (sa + D0 + D_TAILB <= D_SLOTS)                 // distances D0 .. D0 + D_TAILB - 1 in one batch of LDS reads
    { Key<W> kb[D_TAILB];
#pragma unroll
      for (int j = 0; j < D_TAILB; j++) kb[j] = lds_key<W>(ent, sa + D0 + j);
      bool same = true;
#pragma unroll
      for (int j = 0; j < D_TAILB; j++)
        { WT pb, sfb;
          d_unpack<W, KF>(kb[j], G, pb, sfb);
          same = same && sa + D0 + j < nrel && pb == pa;
          const WT dd = sfa ^ sfb;
          const WT tt = ((dd << 1) | dd) & (WT) 0xAAAAAAAAAAAAAAAAull;
          if (same && d_popc(tt) == 1) hm |= 1u << j;
        }
      if (!same) return;
      d = D0 + D_TAILB;
    }
  for (; d <= D_WIN + 1; d++)
    { const int sb = sa + d;
      if (sb >= nrel) break;
      WT pb, sfb;
      if (sb < D_SLOTS) d_unpack<W, KF>(lds_key<W>(ent, sb), G, pb, sfb);
      else              d_unpack<W, KF>(load_key<W>(A.keys, g0 + sb), G, pb, sfb);
      if (pb != pa) break;
      if (d > D_WIN) { big = true; break; }
      const WT dd = sfa ^ sfb;
      const WT tt = ((dd << 1) | dd) & (WT) 0xAAAAAAAAAAAAAAAAull;
      if (d_popc(tt) == 1) hm |= 1u << (d - D0);
    }
}

#ifndef D_WAVES_W2
#define D_WAVES_W2 3                       // two-word k-mers with a count word in the request (exact proof): waves per SIMD
#endif                                     //   (the 23 KB request queue of that variant allows three workgroups per CU)
#ifndef D_WAVES_W2K
#define D_WAVES_W2K 4                      // ... with key-only requests (hash proof): 15 KB less LDS, 108 vector registers
#endif
#ifndef D_WAVES_PER_EU
#define D_WAVES_PER_EU 5
#endif
#define D_WAVES(W_, RW_) ((W_) == 2 ? ((RW_) == 2 ? D_WAVES_W2K : D_WAVES_W2) : ((RW_) == 1 ? D_WAVES_PER_EU : 5))

// tile tickets: one counter per class of workgroups (blockIdx.x mod D_NCLS -- the workgroups of one XCD, as the dispatcher
// deals them out); class c draws the tiles grid + D_NCLS r + c.  ONE counter for all was the bottleneck of the kernel:
// 2.6e6 returning atomics on one address go through at ~11 ns each (31 ms, whatever else the kernel does).
#define D_NCLS   8
#define D_TICKW  32                       // words between two counters

// VAR (round 5): bit 0 = this launch writes a bucket directory and / or look-up signatures; bit 1 = THE HOT FORM: the hash proof
// through the look-up chain on a table that came with its prefix index -- no directory, no signatures, the two-bit candidate map,
// the fingerprint, requests from the owners of a hi-side pair only, the per-bucket request histogram: all of it known at compile
// time, none of those tests in the tile and none of their arguments in registers (33 -> 15 spilled SGPRs, 52 -> ~20 spill moves
// per thread and tile: -0.27 ms on the diploid table for the directory / signature half alone).  VAR = 1 is the general form.
template <int W, int RW, bool ODD, bool KF, int VAR = 1> __global__ void __launch_bounds__(D_TPB)
__attribute__((amdgpu_waves_per_eu(D_WAVES(W, RW), D_WAVES(W, RW))))
kf_pass1_d(P1Hot A, const P1Cold *__restrict__ cold)
{ __shared__ uint16_t tailq[D_OWN + 8];   // deferred tail: slots of this tile's queued entries
  __shared__ u64      ent[D_SLOTS * W];  // the staged k-mers
  __shared__ uint16_t lcn[D_SLOTS];
  __shared__ u64      sq[(RW == 1 ? D_QCAP : D_OWN) * RW];
  __shared__ u64      sfp[D_TPB / 64][2];
  constexpr bool D_BM = (W == 1 && RW == 1) || (W == 2 && RW != 1);
  __shared__ __attribute__((aligned(16))) unsigned bm[D_BM ? 2 * D_BMW : 1];
  __shared__ unsigned hist[(D_BM && RW == W) ? D_HB : 1];
  __shared__ unsigned s_tn[2], s_qn, s_unsorted, s_chunk, s_used;      // (s_tn: one counter per tile parity)
  __shared__ unsigned s_tk[2], s_tk0;      // the ticket thread 0 hands on at the head of a tile, read at its end (one word per
                                           // tile parity: thread 0 is at the next head while other waves still read); the first one
  __shared__ u64      s_base, s_total;

  const int t = threadIdx.x;
  const int lane = t & 63, wv = t >> 6;
  const int slot0 = (wv * D_WL + lane) * 4;
  const int64_t n = A.n;
  u64 fa = 0, fb = 0;                      // fingerprint: XOR of the entries' 128-bit terms
  DShared S;
  S.tailq = tailq; S.ent = ent; S.lcn = lcn; S.sq = sq;
  S.s_tn = &s_tn[0]; S.s_qn = &s_qn; S.s_unsorted = &s_unsorted;
  S.bm = bm; S.hist = hist;
  if (D_BM) for (int w = t; w < 2 * D_BMW; w += D_TPB) bm[w] = 0;
  if (D_BM && RW == W) for (int w = t; w < D_HB; w += D_TPB) hist[w] = 0;
  // Tiles are drawn from a counter, not dealt out by stride: the SIMDs favour their oldest waves, so of the five
  // workgroups that share a CU one is served first and finishes its (equal) share at 60 % of the kernel's time, the
  // next at 68, 78, 88 % -- and the last one runs alone, at a fraction of the CU's issue rate (profiles/
  // r03_pass1_workgroup_ends.txt).  With tickets every workgroup works until the table is done.  A workgroup starts
  // with tile blockIdx.x and knows its next TWO tiles (the next one is being prefetched while a tile is worked on); the
  // ticket for the third is drawn by thread 0 at the head of a tile and picked up at the head of the next one, where the
  // wave waits for its prefetched entries anyway -- a returning atomic and the loads behind it share one in-order counter
  // (vmcnt), so waiting for the ticket anywhere else means waiting for the prefetch (31 ms instead of 15).
  if (t == 0)
    { s_chunk = F_NOCHUNK; s_used = 0; s_total = 0; s_tn[0] = 0; s_tn[1] = 0; s_qn = 0; s_unsorted = 0;
      s_tk0 = atomicAdd(&cold->tick[(blockIdx.x % D_NCLS) * D_TICKW], 3u);
    }
  lds_barrier();
  (void) lane; (void) slot0; (void) n;
  if (cold->times && t == 0)
    { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      cold->times[3 * (size_t) blockIdx.x] = wall_clock64(); cold->times[3 * (size_t) blockIdx.x + 2] = hw;
    }

  DPrefetch<W> pf;
  pf.valid = false;
  int par = 0;
  const unsigned cls = blockIdx.x % D_NCLS;
  unsigned ticket = (unsigned) __builtin_amdgcn_readfirstlane((int) s_tk0);      // ticket r stands for tile grid + D_NCLS r + cls
  int64_t tnext = (int64_t) gridDim.x + (int64_t) ticket * D_NCLS + cls, tnext2 = tnext + D_NCLS;
  ticket += 2u;                                 // thread 0: the ticket on its way (the first one is known)
  for (int64_t tile = blockIdx.x; tile < A.ntiles; par ^= 1)
    { const int64_t g0 = tile * D_OWN - D_LEAD;
      S.s_tn = &s_tn[par];
      // (the wait for the prefetched entries belongs in front of the ticket: behind it, the compiler -- which sees one
      //  path with the atomic and one without -- waits for everything, the atomic included)
      const bool inner = g0 >= 0 && g0 + D_SLOTS + 32 <= n;
      if (inner && !pf.valid) { pf.load(A.keys, A.cnt, g0 + slot0, g0 + D_LEAD); pf.valid = true; }
      pf.arrive();
      if (t == 0)
        { // `ticket` was drawn one tile ago (for the tile after `tnext2`)
          s_tk[par] = ticket;
          // Nothing may use the result before the next head.  It has to be a GLOBAL atomic that the compiler knows: a
          // flat one counts as an LDS operation too (the tile's first LDS wait would wait for it), one written in inline
          // assembly makes the compiler's counted waits for its own loads unsafe, and a global atomic on a uniform
          // address is rebuilt from a broadcast by the compiler's atomic optimiser on the spot -- which is a wait.  The
          // address is therefore made to look divergent: a zero that only the hardware knows is added to it.
          unsigned zero;
          asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
          ticket = __hip_atomic_fetch_add((__attribute__((address_space(1))) unsigned *) &cold->tick[cls * D_TICKW] + zero, 1u,
                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      // the tile after this one, if it is an inner tile too (-1: none, or an edge tile, which loads for itself)
      int64_t g0n = tnext * D_OWN - D_LEAD;
      if (tnext >= A.ntiles || g0n + D_SLOTS + 32 > n) g0n = -1;
      if (inner)
        d_tile<W, RW, ODD, KF, true, VAR>(A, S, g0, g0n, t, fa, fb, pf);
      else
        d_tile<W, RW, ODD, KF, false, VAR>(A, S, g0, g0n, t, fa, fb, pf);
      if (!(D_ABL & 4096)) lds_barrier();          // the staged copy and the queues of this tile are complete
      //@mark D_FLUSH
      const bool last = tnext >= A.ntiles;
      const unsigned tn = s_tn[par];               // (zeroed again behind the barrier at the end of this iteration: the
      const unsigned qn = s_qn;                    //  next tile counts in the other one)
      if (D_BM && A.bmap)                           // candidate-block bits of this tile -> global map
        { const int two = ((VAR & 2) || A.two()) ? 1 : 0;          // (two-bit map: twice the words, at twice the word offset)
          for (int w = t; w < (D_BMW << two); w += D_TPB)
            { const unsigned v = bm[w];
              if (v)
                { atomicOr(&A.bmap[(((size_t) (((uint32_t) (ent[D_LEAD * W] >> 32) >> A.bmsh()) >> 5)) << two) + w], v);
                  bm[w] = 0;
                }
            }
        }

      // ---- deferred tail: one dense detection pass over this tile's queued entries (a lane each) ---------------------
      // A hit sets the bits of the entry and of its partners in the deferred-entry map (one bit per table entry, so an
      // entry named twice is redone once).
      for (unsigned q = t; q < tn && !(D_ABL & 512); q += D_TPB)
        { const int sa = (int) tailq[q];
          unsigned hm; bool big;
          d_detect<W, ODD, KF>(A, ent, lcn, g0, sa, hm, big);
          if (hm | (unsigned) big)                           // rare: mark the entry and its partners for kf_bigfix
            { uint32_t *db = cold->dbits;
              const int64_t it = g0 + sa;
              atomicOr(&db[it >> 5], 1u << (it & 31));
              for (unsigned m = hm; m; m &= m - 1)
                { const int64_t j = it + D_RD + __ffs(m);
                  atomicOr(&db[j >> 5], 1u << (j & 31));
                }
              if (big) { const int64_t j = it + D_WIN + 1; atomicOr(&db[j >> 5], 1u << (j & 31)); }
            }
        }

      // ---- flush the request queue into this workgroup's chunk -------------------------------------------------------
      // (RW == 1: only when the next tile might overflow the queue, or after this workgroup's last tile --
      //  the barriers and the chunk bookkeeping of a flush cost as much as the copy itself)
      const unsigned qcap = RW == 1 ? D_QCAP : D_OWN;
      if (qn > 0 && (qn + D_OWN > qcap || last))
        { // a batch that does not fit is SPLIT: its head fills the current chunk to the brim, the rest opens a new
          // one -- every chunk but a workgroup's last is full, so the host can sort the chunk array as it is
          // (holes filled with a sentinel) instead of compacting it first
          const unsigned max_chunks = cold->max_chunks;
          u64 *req = cold->req;
          const unsigned old_chunk = s_chunk, old_used = s_used;
          const unsigned room = old_chunk == F_NOCHUNK ? 0u : F_CH - old_used;
          const unsigned head = qn < room ? qn : room;
          lds_barrier();
          if (t == 0)
            { s_base = (u64) old_chunk * F_CH + old_used;          // only used when head > 0
              if (qn > head)
                { if (old_chunk != F_NOCHUNK && old_chunk < max_chunks) cold->chunk_fill[old_chunk] = F_CH;
                  // (look-up chain: the partition kernel finds the chunks of owner w at w + j * owners, no list needed)
                  if (D_BM && RW == W && ((VAR & 2) || A.hbits())) s_chunk = old_chunk == F_NOCHUNK ? blockIdx.x : old_chunk + cold->owners;
                  else s_chunk = atomicAdd(&cold->ctl->n_chunks, 1u);
                  s_used = qn - head;
                }
              else s_used = old_used + qn;
              s_total += qn;
              s_qn = 0;
            }
          lds_barrier();
          if (D_BM && RW == W && ((VAR & 2) || A.hbits()))            // requests per bucket, for the partition of the look-up chain
            { const int hsh = 32 - A.hbits();
              for (unsigned e = t; e < qn; e += D_TPB) atomicAdd(&hist[(unsigned) (sq[e * RW] >> 32) >> hsh], 1u);
            }
          if (head && old_chunk < max_chunks)
            { u64 *o = req + s_base * RW;
              for (unsigned e = t; e < head * RW; e += D_TPB) o[e] = sq[e];
            }
          if (qn > head && s_chunk < max_chunks)
            { u64 *o = req + (u64) s_chunk * F_CH * RW;
              for (unsigned e = t; e < (qn - head) * RW; e += D_TPB) o[e] = sq[head * RW + e];
            }
        }
      if (!(D_ABL & 8192)) lds_barrier();
      if (t == 0) s_tn[par] = 0;
      tile = tnext; tnext = tnext2;
      tnext2 = (int64_t) gridDim.x + (int64_t) (unsigned) __builtin_amdgcn_readfirstlane((int) s_tk[par]) * D_NCLS + cls;
    }

  if (cold->times && t == 0) cold->times[3 * (size_t) blockIdx.x + 1] = wall_clock64();
  if (D_BM && RW == W && ((VAR & 2) || A.hbits()))                   // this workgroup's row of the request histogram (kl_tot / kl_woff)
    for (int w = t; w < D_HB; w += D_TPB) cold->whist[(size_t) blockIdx.x * D_HB + w] = hist[w];
  if (t == 0)
    { if (s_chunk != F_NOCHUNK && s_chunk < cold->max_chunks) cold->chunk_fill[s_chunk] = s_used;
      if (D_BM && RW == W && ((VAR & 2) || A.hbits()) && s_chunk != F_NOCHUNK) atomicMax(&cold->ctl->n_chunks, s_chunk + 1u);   // slots in use
      if (s_total) atomicAdd(&cold->ctl->nreq, s_total);
      if (s_unsorted) cold->ctl->unsorted = 1;
    }
  if ((VAR & 2) || A.want_fp())
    { fa = wave_xor_u64(fa);
      fb = wave_xor_u64(fb);
      if ((t & 63) == 0) { sfp[t >> 6][0] = fa; sfp[t >> 6][1] = fb; }
      lds_barrier();
      if (t < 2)
        { u64 s = 0;
          for (int w2 = 0; w2 < D_TPB / 64; w2++) s ^= sfp[w2][t];
          cold->partials[(size_t) blockIdx.x * 4 + t] = s;
          cold->partials[(size_t) blockIdx.x * 4 + 2 + t] = 0;
        }
    }
}
