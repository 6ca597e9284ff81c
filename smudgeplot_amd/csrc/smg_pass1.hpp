// smg_pass1.hpp -- kf_pass1_r: pass 1 for k <= 64 (W = 1 or 2 64-bit words per k-mer); W = 1 is the headline kernel.
//
// The v3 kernel (kf_pass1_s, LDS-staged strided scan) turned out to be VALU-issue bound: 310 vector
// instructions per table entry, VALU pipes 83 % busy, HBM at 1.8 TB/s (profiles/r01_v3_pmc_*).  This
// version is organised around the instruction count instead:
//
//   * BLOCKED layout: a thread owns 4 CONSECUTIVE entries and also loads the next 4 (its right
//     neighbour's entries: L1 hits, the vector-memory pipe is idle anyway).  The window scan of the
//     first D = 3 distances is then 12 register-to-register tests per thread (9 VALU each), no LDS
//     reads, no loop, no wave-uniform trip count (v3 ran ~8.5 trips of the scan loop because the
//     LONGEST block among 256 entries set the trip count; the mean need is 1.2 tests per entry).
//   * Entries whose window block continues past distance 3 (~6 % on the diploid workload) are queued
//     in LDS and finished densely by a short tail loop; blocks longer than the +-30 window go to
//     kf_bigfix exactly as before.
//   * Results are packed counters  count | delta code<<8 | mid<<24 (with exactly one pair the delta
//     field IS the code byte): a hit costs two selects and two adds.  Every result is added to one LDS
//     word per entry (ds_add, no return value), so the epilogue can run one entry at a time from LDS
//     and the kernel fits 5 waves per SIMD.
//   * count sum <= 1000 (PloidyPlot.c:528-540) is only evaluated on waves that hold a count > 500.
//   * exact loads: a tile reads what it uses (v3's unconditional register prefetch fetched 1.5x); no
//     software prefetch at all -- resident workgroups hide the latency.
//   * the 4 code bytes of a thread leave as one dword; the directory costs one compare and (rarely) one
//     store per entry; the fingerprint accumulates (h ^ sign) and adds the number of negated terms at
//     the end.
//
// Semantics are those of kf_pass1_s (see smg_fast.hpp for the code byte and the request protocol),
// with one relaxation that needs the table symmetry the run proves anyway: an entry with >= 2 pairs
// always sends a request, even if all of its pairs sit on the self-mirrored position.  The receiver
// rc(x) then has >= 2 pairs itself (the mirror images), i.e. is excluded from the plot regardless of
// its P flag.

#pragma once
#include "smg_fast.hpp"

#define R_TPB   256
#define R_SCAN  1024                     // entries scanned per tile (4 per thread)
#define R_HALO  32                       // left halo: scanned, owned by the previous tile
#define R_OWN   (R_SCAN - R_HALO)        // 992 == S_OWN: same tiling as v3 (kf_bigfix, chunk sizing)
#define R_WIN   30                       // partners are searched within +-30 entries
#define R_CRED  (R_SCAN + 32)
#define R_BIG   0x80000000u
// (A bank-conflict-free SoA layout of cred/ent/lcn -- slot s at (s&3)*264 + (s>>2) -- was measured: the v5 PMC
//  shows more LDS conflict cycles than LDS issue cycles, but the extra index arithmetic and the narrower
//  LDS writes cost more VALU than the conflicts cost time: 21.9 ms vs 20.8 ms.  The kernel is VALU bound.)
#ifndef R_D
#define R_D     3                        // distances scanned in registers; farther partners: tail loop
#endif
// block ids of the request filter: the leading min(30, 2*p0) bits of the k-mer = hi32(word 0) >> r_bmsh
// (one-word k-mers: pshift = 32 - 2*p0; two-word k-mers have p0 >= 16)
#define R_BMF   4096                     // block ids per tile with a bit in LDS (request filter)
#define R_BMW   (R_BMF / 32)             //   = words of the global bit map they cover
#define R_QCAP  1280                     // LDS request queue (records); flushed when the next tile might not fit
                                         // (sized so that SIX workgroups fit the 160 KB of LDS of a CU)

// per-entry scan words: the first p0 bases ("pre") and the last k-p0 bases ("suf") of the k-mer, each in ONE
// machine word of this type: 32 bits for k <= 32 (p0 <= 16), 64 bits for 33 <= k <= 64 (p0 <= 32)
template <int W> struct RWord;
template <> struct RWord<1> { typedef unsigned type; };
template <> struct RWord<2> { typedef u64 type; };

struct GeoR
{ int  k;
  int  pshift;       // W=1: pre = hi32 >> pshift (32 - 2*p0; 32 means "no prefix": k == 1);  W=2: pre = w0 >> pshift (64 - 2*p0)
  int  kshift;       // W=1: 64 - 2k;  W=2: 128 - 2k  (the k-mer is left aligned in W words)
  u64  smask;        // low 2*(k-p0) bits
  int  mshift;       // odd k: suffix >> mshift != 0  <=> the top suffix base (position p0) differs
};

template <int W> SMG_DEV int r_bmsh(const GeoR &G) { return W == 1 ? (G.pshift > 2 ? G.pshift : 2) : 2; }

SMG_DEV int r_popc(unsigned v) { return __popc(v); }
SMG_DEV int r_popc(u64 v) { return __popcll(v); }

// contribution of one pair to its LOWER entry: count 1 | delta code (31 + d) << 8 | mid << 24; the UPPER
// entry gets the same with delta code 31 - d.  With exactly one pair the delta field IS the code byte.
template <bool ODD, typename WT> SMG_DEV unsigned r_val(WT dd, int d, const GeoR &G)
{ unsigned v = 1u | ((unsigned) (31 + d) << 8);
  if (ODD) v += (unsigned) (dd >> G.mshift) << 24;
  return v;
}
#define R_UP(v, d) ((v) - ((unsigned) (2 * (d)) << 8))

// KF (W=1 only): 17 <= k <= 32, the k-mer straddles both 32-bit halves (pshift and kshift below 32): no selects
template <int W, bool KF> SMG_DEV void
r_unpack(const Key<W> &x, const GeoR &G, typename RWord<W>::type &pre, typename RWord<W>::type &suf)
{ if constexpr (W == 1)
    { const unsigned hi = (unsigned) (x.w[0] >> 32), lo = (unsigned) x.w[0];
      if (KF)
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

// the 12 register tests of one thread: entries 0..3 are its own, 4..6 its right neighbour's
template <bool ODD, bool CHECK, typename WT> SMG_DEV void
r_slots(const WT (&pre)[8], const WT (&suf)[8], const unsigned (&cn)[8], const GeoR &G, unsigned (&acc)[4 + R_D])
{ const WT AA = (WT) 0xAAAAAAAAAAAAAAAAull;
#pragma unroll
  for (int a = 0; a < 4; a++)
    {
#pragma unroll
      for (int d = 1; d <= R_D; d++)
        { const int b = a + d;
          const WT dd = suf[a] ^ suf[b];
          const WT tt = ((dd << 1) | dd) & AA;
          bool hit = (pre[a] == pre[b]) && (r_popc(tt) == 1);
          if (CHECK) hit = hit && (cn[a] + cn[b] <= SMG_SMAX);
          const unsigned v = r_val<ODD, WT>(dd, d, G);
          acc[a] += hit ? v : 0u;
          acc[b] += hit ? R_UP(v, d) : 0u;
        }
    }
}

struct RShared                            // the workgroup's LDS arrays (pointers: the tile body is a function)
{ unsigned *cred; uint16_t *tailq; u64 *ent; uint16_t *lcn; u64 *sq;
  unsigned *s_tn, *s_qn, *s_nbig;
  unsigned *bm;                        // candidate-block bits of this tile: R_BMW words (request filter)
};

// One tile: phases 1-3.  INNER tiles lie completely inside the table (all but the first and the last one):
// vector loads, no bounds checks, no table-end cases in the directory code.
// RW = 64-bit words per request record: W (the complement k-mer) or W + 1 (+ count | has-hi-pair << 16).
template <int W, int RW, bool ODD, bool KF, bool INNER> SMG_DEV void
r_tile(const FastArgs &A, const GeoR &G, const RShared &S, uint32_t *__restrict__ bstart,
       FastCtl *__restrict__ ctl, int emit_all, int want_fp, int64_t g0, int t,
       u64 &fa, u64 &fb, unsigned &fneg, unsigned &bigmask)
{ typedef typename RWord<W>::type WT;
  bigmask = 0;                             // owned entries deferred to kf_bigfix (listed by the caller)
  const int slot0 = 4 * t;
  const int64_t i0 = g0 + slot0;
  const int64_t n = A.n;
  const u64 *__restrict__ keys = A.keys;
  const uint16_t *__restrict__ cnts = A.cnt;
  unsigned vmask = 0xFF;                   // entries i0 .. i0+7 inside the table?

  //@mark P1_LOAD
  // ---- phase 1: load 8 consecutive entries, scan the first three distances in registers ------------------
  // (no software prefetch: 5 resident workgroups per CU hide the load latency, and the registers a
  //  prefetch pins are worth more as occupancy)
  { Key<W> kk[8]; WT pre[8], suf[8]; unsigned cn[8];
    if (INNER)
      { const uint16_t *ct = cnts + i0;
        if constexpr (W == 1)
          { const u64 *kt = keys + i0;
#pragma unroll
            for (int q = 0; q < 4; q++)
              { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(kt + 2 * q);
                kk[2 * q].w[0] = v.x; kk[2 * q + 1].w[0] = v.y;
              }
          }
        else
          {
#pragma unroll
            for (int e = 0; e < 8; e++) kk[e] = load_key<W>(keys, i0 + e);
          }
#pragma unroll
        for (int q = 0; q < 2; q++)
          { const ushort4 v = *reinterpret_cast<const ushort4 *>(ct + 4 * q);
            cn[4 * q] = v.x; cn[4 * q + 1] = v.y; cn[4 * q + 2] = v.z; cn[4 * q + 3] = v.w;
          }
      }
    else                                   // first / last tile: guarded element loads
      { vmask = 0;
#pragma unroll
        for (int e = 0; e < 8; e++)
          { const int64_t i = i0 + e;
            const bool ok = i >= 0 && i < n;
            vmask |= (unsigned) ok << e;
            const Key<W> kx = load_key<W>(keys, ok ? i : 0);
#pragma unroll
            for (int w = 0; w < W; w++) kk[e].w[w] = ok ? kx.w[w] : 0ull;
            cn[e] = ok ? (unsigned) cnts[i] : 0xFFFFu;
          }
      }
    //@mark P1_UNPACK
#pragma unroll
    for (int e = 0; e < 8; e++) r_unpack<W, KF>(kk[e], G, pre[e], suf[e]);
    // LDS copy for the tail loop and the epilogue
    { if constexpr (W == 1)
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
      if (t == R_TPB - 1)
        {
#pragma unroll
          for (int w = 0; w < W; w++) S.ent[R_SCAN * W + w] = kk[4].w[w];
          S.lcn[R_SCAN] = (uint16_t) cn[4];
        }
    }
    //@mark P1_SLOTS
    unsigned acc[4 + R_D];
#pragma unroll
    for (int e = 0; e < 4 + R_D; e++) acc[e] = 0;
    { unsigned mx = cn[0];
#pragma unroll
      for (int e = 1; e < 4 + R_D; e++) mx = mx > cn[e] ? mx : cn[e];
      if (__all(mx <= SMG_FMAX)) r_slots<ODD, false, WT>(pre, suf, cn, G, acc);
      else                       r_slots<ODD, true, WT>(pre, suf, cn, G, acc);
    }
    //@mark P1_CREDIT
    // every result goes to the entry's credit word (own entries too: frees the registers)
#pragma unroll
    for (int e = 0; e < 4 + R_D; e++) atomicAdd(&S.cred[slot0 + e], acc[e]);     // unconditional: no VALU spent on tests
    // entries whose block continues past distance R_D
    unsigned alive = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) alive |= (unsigned) (pre[r] == pre[r + R_D + 1]) << r;
    alive &= vmask & (vmask >> (R_D + 1));
    if (alive) S.tailq[atomicAdd(S.s_tn, 1u)] = (uint16_t) (t | (alive << 8));
  }
  lds_barrier();

  //@mark P2_TAIL
  // ---- phase 2: tail, distances R_D+1..30 from the LDS copy (global memory past the tile edge), rare ------
  { const unsigned tn = *S.s_tn;
    // items are dealt to waves 0 and 1 only: the other two skip the whole phase (the kernel is VALU bound)
    for (unsigned q = ((unsigned) (t & 63) << 1) | (unsigned) (t >> 6); t < 128 && q < tn; q += 128)
      { const unsigned item = S.tailq[q];
        const int ts = 4 * (int) (item & 0xFF);
        for (unsigned m = item >> 8; m; m &= m - 1)
          { const int sa = ts + __ffs(m) - 1;
            WT pa, sfa, pb, sfb;
            r_unpack<W, KF>(lds_key<W>(S.ent, sa), G, pa, sfa);
            const unsigned ca = S.lcn[sa];
            for (int d = R_D + 1; d <= R_WIN + 1; d++)
              { const int sb = sa + d;
                unsigned cb;
                if (!INNER && g0 + sb >= n) break;
                if (sb < R_SCAN) { r_unpack<W, KF>(lds_key<W>(S.ent, sb), G, pb, sfb); cb = S.lcn[sb]; }
                else             { r_unpack<W, KF>(load_key<W>(keys, g0 + sb), G, pb, sfb); cb = cnts[g0 + sb]; }
                if (pb != pa) break;
                if (d > R_WIN) { atomicOr(&S.cred[sa], R_BIG); atomicOr(&S.cred[sb], R_BIG); break; }
                const WT dd = sfa ^ sfb;
                const WT tt = ((dd << 1) | dd) & (WT) 0xAAAAAAAAAAAAAAAAull;
                if (r_popc(tt) == 1 && ca + cb <= SMG_SMAX)
                  { const unsigned v = r_val<ODD, WT>(dd, d, G);
                    atomicAdd(&S.cred[sa], v);
                    atomicAdd(&S.cred[sb], R_UP(v, d));
                  }
              }
          }
      }
  }
  lds_barrier();

  //@mark P3_HEAD
  // ---- phase 3: owned entries, one at a time from LDS (keeps the register count low) ----------------------
  if (slot0 >= R_HALO)
    { unsigned codes = 0;
      uint32_t bcur = dir_bucket(A.dir, S.ent[slot0 * W]);
      constexpr bool R_BM = (W == 1 && RW == 1) || (W == 2 && RW == 3);     // variants that feed the request filter
      const int bmsh = r_bmsh<W>(G);
      const uint32_t bmbase = R_BM ? (((uint32_t) (S.ent[R_HALO * W] >> 32) >> bmsh) & ~31u) : 0u;
      unsigned farmask = 0;
#pragma unroll 1
      for (int r = 0; r < 4; r++)
        { const int64_t i = i0 + r;
          const bool ok = INNER || (vmask >> r & 1) != 0;
          const unsigned R = S.cred[slot0 + r];
          const Key<W> x = lds_key<W>(S.ent, slot0 + r);
          const unsigned c = S.lcn[slot0 + r];
          const unsigned count = R & 0xFF, c1 = (R >> 8) & 0x7F;
          const bool w2 = !ODD || ((R >> 24) & 0x7F) == 0;
          const bool big = (R & R_BIG) != 0;
          unsigned code = count ? CODE_MULTI : CODE_NONE;
          if (count == 1) code = c1 | (w2 ? (unsigned) CODE_W2 : 0u);
          const bool hi = count == 1 ? w2 : count >= 2;            // owns a pair at p > k-1-p
          if (big) code = CODE_DEFER;
          codes |= code << (8 * r);
          if (W <= 2)        // look-up signature: parked in the (already consumed) count slot, stored 4 at a time below
            { const unsigned xh = (unsigned) (x.w[0] >> 32), xl = (unsigned) x.w[0];
              S.lcn[slot0 + r] = (uint16_t) (A.sigsh >= 32 ? xh >> (A.sigsh - 32) : __builtin_amdgcn_alignbit(xh, xl, A.sigsh));
            }
          if (ok && big) { bigmask |= 1u << r; atomicAdd(S.s_nbig, 1u); }
          if (R_BM && count == 1 && A.bmap)
            { // request filter: a CANDIDATE (exactly one suffix-side pair; a deferred entry may be marked in vain, which
              // is harmless) sets the bit of its block id in the tile's LDS bit map -- word 0 is the map word of
              // the tile's first owned entry; the few ids beyond R_BMF (sparse tables) are marked after the loop
              const uint32_t rel = ((uint32_t) (x.w[0] >> 32) >> bmsh) - bmbase;
              if (rel < R_BMF) atomicOr(&S.bm[rel >> 5], 1u << (rel & 31));
              else farmask |= 1u << r;
            }
          //@mark P3_DIR
          // order check + bucket directory: the first entry of every bucket stores its index
          if (INNER)
            { const Key<W> xn = lds_key<W>(S.ent, slot0 + r + 1);
              const uint32_t bn = dir_bucket(A.dir, xn.w[0]);
              if (!key_lt<W>(x, xn)) ctl->unsorted = 1;
              if (bn != bcur) bstart[bn] = (uint32_t) (i + 1);
              bcur = bn;
            }
          else if (ok)
            { if (i == 0) bstart[bcur] = 0u;
              if (i + 1 < n)
                { const Key<W> xn = lds_key<W>(S.ent, slot0 + r + 1);
                  const uint32_t bn = dir_bucket(A.dir, xn.w[0]);
                  if (!key_lt<W>(x, xn)) ctl->unsorted = 1;
                  if (bn != bcur) bstart[bn] = (uint32_t) (i + 1);
                  bcur = bn;
                }
              else bstart[A.dir.nb] = (uint32_t) n;
            }
          //@mark P3_RC
          // complement: for the fingerprint of every owned entry, and for the request of the emitting ones
          const bool emit = ok && !big && (emit_all || hi);
          if (emit || (want_fp && ok))
            { const Key<W> rc = revcomp<W>(x, G.k);
              if (want_fp && ok)
                { const bool lt = key_lt<W>(x, rc);
                  const bool gt = ODD ? !lt : key_lt<W>(rc, x);   // odd k: no k-mer is its own complement
                  u64 ha, hb;
                  arx_hash<W>(lt ? x : rc, c, ha, hb);
                  const u64 sg = gt ? ~0ull : 0ull;               // -h == (h ^ ~0) + 1
                  if (ODD) { fa += ha ^ sg; fb += hb ^ sg; }
                  else
                    { const u64 keep = (lt || gt) ? ~0ull : 0ull;  // self-complementary: no term
                      fa += (ha ^ sg) & keep; fb += (hb ^ sg) & keep;
                    }
                  fneg += gt;
                }
              //@mark P3_EMIT
              if (emit)
                { const unsigned q = atomicAdd(S.s_qn, 1u);
#pragma unroll
                  for (int w = 0; w < W; w++) S.sq[q * RW + w] = rc.w[w];
                  if (RW > W) S.sq[q * RW + W] = (u64) c | ((u64) hi << 16);
                }
            }
        }
      if (R_BM && farmask)                         // sparse table: block ids outside the tile's LDS window
        for (unsigned m = farmask; m; m &= m - 1)
          { const uint32_t id = (uint32_t) (S.ent[(slot0 + __ffs(m) - 1) * W] >> 32) >> bmsh;
            atomicOr(&A.bmap[id >> 5], 1u << (id & 31));
          }
      //@mark P3_STORE
      { if (INNER || (vmask & 0xF) == 0xF)
            { *reinterpret_cast<unsigned *>(A.code + i0) = codes;
              if (W <= 2) *reinterpret_cast<u64 *>(A.sig + i0) = *reinterpret_cast<const u64 *>(&S.lcn[slot0]);
            }
          else
            for (int r = 0; r < 4; r++)
              if (vmask >> r & 1)
                { A.code[i0 + r] = (uint8_t) (codes >> (8 * r));
                  if (W <= 2) A.sig[i0 + r] = S.lcn[slot0 + r];
                }
        }
    }
}

#ifndef R_WAVES_PER_EU
#define R_WAVES_PER_EU 6
#endif

template <int W, int RW, bool ODD, bool KF> __global__ void __launch_bounds__(R_TPB)
__attribute__((amdgpu_waves_per_eu(W == 2 ? 3 : (RW == 1 ? R_WAVES_PER_EU : 5), W == 2 ? 3 : (RW == 1 ? R_WAVES_PER_EU : 5))))
kf_pass1_r(FastArgs A, GeoR G, uint32_t *__restrict__ bstart, u64 *__restrict__ req,
           uint32_t *__restrict__ chunk_fill, unsigned max_chunks, uint32_t *__restrict__ biglist,
           unsigned big_cap, int emit_all, int want_fp, u64 *__restrict__ partials,
           FastCtl *__restrict__ ctl, int64_t ntiles)
{ __shared__ unsigned cred[R_CRED];      // per entry: count | delta code << 8 | mid << 24 | BIG
  __shared__ uint16_t tailq[R_TPB];      // thread | alive mask << 8
  __shared__ u64      ent[(R_SCAN + 4) * W];   // the scanned k-mers (+ the first one of the next tile)
  __shared__ uint16_t lcn[R_SCAN + 4];
  __shared__ u64      sq[(RW == 1 ? R_QCAP : R_OWN) * RW];
  __shared__ u64      sfp[R_TPB / 64][2];
  constexpr bool R_BM = (W == 1 && RW == 1) || (W == 2 && RW == 3);
  __shared__ unsigned bm[R_BM ? R_BMW : 1];
  __shared__ unsigned s_tn, s_qn, s_nbig, s_chunk, s_used, s_bigbase, s_bigcur;
  __shared__ u64      s_base, s_total;

  const int t = threadIdx.x;
  const int slot0 = 4 * t;
  const int64_t n = A.n;
  u64 fa = 0, fb = 0;                      // fingerprint: sum of (h ^ sign); the -1's are added at the end
  unsigned fneg = 0;
  RShared S;
  S.cred = cred; S.tailq = tailq; S.ent = ent; S.lcn = lcn; S.sq = sq;
  S.s_tn = &s_tn; S.s_qn = &s_qn; S.s_nbig = &s_nbig;
  S.bm = bm;
  if (R_BM && t < R_BMW) bm[t] = 0;
  for (int s = t; s < R_CRED; s += R_TPB) cred[s] = 0;
  if (t == 0) { s_chunk = F_NOCHUNK; s_used = 0; s_total = 0; s_tn = 0; s_qn = 0; s_nbig = 0; }
  lds_barrier();

  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
    { const int64_t g0 = tile * R_OWN - R_HALO;
      unsigned bigmask;
      if (g0 >= 0 && g0 + R_SCAN + 4 <= n)
        r_tile<W, RW, ODD, KF, true>(A, G, S, bstart, ctl, emit_all, want_fp, g0, t, fa, fb, fneg, bigmask);
      else
        r_tile<W, RW, ODD, KF, false>(A, G, S, bstart, ctl, emit_all, want_fp, g0, t, fa, fb, fneg, bigmask);
      lds_barrier();
      //@mark P4_FLUSH
      if (R_BM && A.bmap && t < R_BMW)              // candidate-block bits of this tile -> global map
        { const unsigned v = bm[t];
          if (v)
            { const int bmsh = r_bmsh<W>(G);
              atomicOr(&A.bmap[(((uint32_t) (ent[R_HALO * W] >> 32) >> bmsh) >> 5) + t], v);
              bm[t] = 0;
            }
        }
      // zero the credit words for the next tile
      *reinterpret_cast<uint4 *>(&cred[slot0]) = make_uint4(0, 0, 0, 0);
      if (t < (R_CRED - R_SCAN) / 4) *reinterpret_cast<uint4 *>(&cred[R_SCAN + slot0]) = make_uint4(0, 0, 0, 0);

      // ---- flush the request queue into this workgroup's chunk; publish deferred entries -----------------
      // (RW == 1: only when the next tile might overflow the queue, or after this workgroup's last tile --
      //  the barriers and the chunk bookkeeping of a flush cost as much as the copy itself)
      const unsigned qn = s_qn;
      const unsigned qcap = RW == 1 ? R_QCAP : R_OWN;
      if (qn > 0 && (qn + R_OWN > qcap || tile + gridDim.x >= ntiles))
        { // a batch that does not fit is SPLIT: its head fills the current chunk to the brim, the rest opens a new
          // one -- every chunk but a workgroup's last is full, so the host can sort the chunk array as it is
          // (holes filled with a sentinel) instead of compacting it first
          const unsigned old_chunk = s_chunk, old_used = s_used;
          const unsigned room = old_chunk == F_NOCHUNK ? 0u : F_CH - old_used;
          const unsigned head = qn < room ? qn : room;
          lds_barrier();
          if (t == 0)
            { s_base = (u64) old_chunk * F_CH + old_used;          // only used when head > 0
              if (qn > head)
                { if (old_chunk != F_NOCHUNK && old_chunk < max_chunks) chunk_fill[old_chunk] = F_CH;
                  s_chunk = atomicAdd(&ctl->n_chunks, 1u);
                  s_used = qn - head;
                }
              else s_used = old_used + qn;
              s_total += qn;
              s_qn = 0;
            }
          lds_barrier();
          if (head && old_chunk < max_chunks)
            { u64 *o = req + s_base * RW;
              for (unsigned e = t; e < head * RW; e += R_TPB) o[e] = sq[e];
            }
          if (qn > head && s_chunk < max_chunks)
            { u64 *o = req + (u64) s_chunk * F_CH * RW;
              for (unsigned e = t; e < (qn - head) * RW; e += R_TPB) o[e] = sq[head * RW + e];
            }
        }
      const unsigned nb = s_nbig;
      if (nb > 0)                                   // rare: list the deferred entries (the k-mer copy is free now)
        { uint32_t *list = reinterpret_cast<uint32_t *>(ent);
          lds_barrier();
          if (t == 0) { s_bigbase = atomicAdd(&ctl->nbig, nb); s_nbig = 0; s_bigcur = 0; }
          lds_barrier();
          for (unsigned m = bigmask; m; m &= m - 1)
            list[atomicAdd(&s_bigcur, 1u)] = (uint32_t) (g0 + slot0 + __ffs(m) - 1);
          lds_barrier();
          for (unsigned e = t; e < nb; e += R_TPB)
            if (s_bigbase + e < big_cap) biglist[s_bigbase + e] = list[e];
        }
      if (t == 0) s_tn = 0;
      lds_barrier();
    }

  if (t == 0)
    { if (s_chunk != F_NOCHUNK && s_chunk < max_chunks) chunk_fill[s_chunk] = s_used;
      if (s_total) atomicAdd(&ctl->nreq, s_total);
    }
  if (want_fp)
    { fa = wave_sum_u64(fa + (u64) fneg);
      fb = wave_sum_u64(fb + (u64) fneg);
      if ((t & 63) == 0) { sfp[t >> 6][0] = fa; sfp[t >> 6][1] = fb; }
      lds_barrier();
      if (t < 2)
        { u64 s = 0;
          for (int w2 = 0; w2 < R_TPB / 64; w2++) s += sfp[w2][t];
          partials[(size_t) blockIdx.x * 4 + t] = s;
          partials[(size_t) blockIdx.x * 4 + 2 + t] = 0;
        }
    }
}
