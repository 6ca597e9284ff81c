// smg_lookup.hpp -- the look-up chain of the hash proof for k <= 64 (key-only request records of one or two words): requests -> P flags.
//
// Round 1: sentinel fill, rocPRIM Onesweep histogram + scatter on the leading 8 bits (3.5 ms on the 1 Gbp table),
// kf_filter probing the block map through the L2s (3.0), Onesweep on 24 bits of the survivors (1.1), in-order look-ups
// (1.75): 9.3 ms for 4.4e8 requests.  Now (5.0 ms):
//
//   kl_tot / kl_scan / kl_woff   Every workgroup of PASS 1 counts its requests per bucket (leading NB <= 10 bits of
//             rc(x): one LDS add per request there, instead of a 1.5 ms pass over the 3.5 GB request list) and fills
//             chunk slots that are its own by construction (w, w + owners, ..).  Column sums of the owners' rows ->
//             bucket offsets -> the first output slot of every owner in every bucket;
//   kl_part   one-pass partition into a dense, bucket-ordered array: workgroup w takes the chunks of owner w, 16384
//             records at a time, sorts them by bucket inside LDS and copies them out in runs of ~16 records.  The output
//             cursors are LDS words of the owner -- the first version took a slot range per (batch, bucket) from
//             global cursors: 5.5e7 atomics, 2.4 of its 3.4 ms;
//   kl_probe  one 1024-thread workgroup per CU takes whole buckets.  The part of the block map that a bucket can hit
//             is folded 4:1 into 128 KB of LDS (a "coarse" bit = four neighbouring block ids), so the first test of
//             every request is an LDS read (the 128 MB - 1 GB map itself is read exactly once, as a stream); the ~20 %
//             that pass probe the full-resolution map in global memory, and the survivors (1 in 115 with the two-bit
//             map of a single-GPU run, 1 in 5 with the 30-bit map of a sharded one) are queued in LDS and looked up
//             64 at a time with every lane busy (directory bucket, bisection on the signatures, P flag) -- inside
//             their bucket, i.e. inside 1/1024 of the table, without being sorted first.
//             In a sharded run the same kernel appends the survivors to a chunk list instead (they have to travel).
//
// The map resolution went from 30 to 32 id bits (5.7 % instead of 20 % of the requests survive) once the filter no
// longer had to keep the map words of the resident workgroups inside the L2s; the second bit per 32-block group
// (smg_fast.hpp, bm2_bits) took the survivors to 0.9 %.

#pragma once
#include "smg_fast.hpp"

#define L_NB_MAX   10                      // request buckets: leading NB bits of the target k-mer
#define L_BK       (1 << L_NB_MAX)
#define L_SLICE_LG 20                      // coarse map bits per bucket in LDS (128 KB)

struct LookupGeo
{ int fb;          // id bits of the block map: bit (key >> (64 - fb))
  int cb;          // coarse ids: fb - 2 bits
  int nb;          // bucket bits: 1 .. 10, cb - nb <= 20
};

static inline LookupGeo lookup_geo(int fb)
{ LookupGeo g;
  g.fb = fb; g.cb = fb - 2;
  g.nb = g.cb - L_SLICE_LG;
#ifdef L_NB_FORCE                          // tuning build: another number of request buckets (kl_probe_x only: kl_probe's LDS slice assumes cb - nb <= 20)
  g.nb = L_NB_FORCE;
#endif
  if (g.nb < 1) g.nb = 1;
  if (g.nb > L_NB_MAX) g.nb = L_NB_MAX;
  return g;
}

// One survivor of the filter: the record is rc(x) of an entry x that owns a pair at p > k-1-p, the entry it names gets its P
// flag; a record whose k-mer is not in the table refutes the symmetry of the table.
// (Tried in round 4: turning the question round on a single shard -- the CANDIDATES send, the owners of a pair at p > k-1-p
//  mark the map, the look-up reads the answer from the complement's code byte.  Correct, and no gain: on the bench table
//  18.7 % of the entries are candidates, 17.5 % own such a pair -- the two streams are the same size.)
template <int W> SMG_DEV void lookup_one(const FastArgs &A, const Key<W> &y, FastCtl *__restrict__ ctl)
{
#ifdef L_ABL_NOLOOKUP                      // ablation build (timing only, WRONG results): what the probe kernels cost without their look-ups
  return;
#endif
#ifdef L_ABL_EXTRAROW                      // ablation build (timing only): one more random k-mer line per look-up -- what a survivor of
  { const u64 h = (y.w[0] * 0x9E3779B97F4A7C15ull) >> 20;          // an 8-byte "requester's entry number" record would have to fetch
    const Key<W> z = load_key<W>(A.keys, (int64_t) (h % (u64) A.n));
    if (z.w[0] == 0x0123456789ABCDEFull) ctl->missing = 2;
  }
#endif
  const int64_t i = sig_find<W>(A, y, false);          // (no signatures bound: find_key_near, smg_device.hpp)
  if (i < 0) { if (ctl->missing == 0) ctl->missing = 1; return; }
  SET_P(A, i);
}

// ---- bucket offsets: exclusive scan of <= 1024 counts ----------------------------------------------------------
__global__ void __launch_bounds__(L_BK)
kl_scan(const unsigned *__restrict__ ghist, int nbk, u64 *__restrict__ boff /* [nbk + 1] */, u64 *__restrict__ bcur /* [nbk] */,
        unsigned *__restrict__ bnext)
{ __shared__ u64 s[L_BK];
  const int t = threadIdx.x;
  const u64 v = t < nbk ? (u64) ghist[t] : 0ull;
  s[t] = v;
  __syncthreads();
  for (int o = 1; o < L_BK; o <<= 1)
    { const u64 a = t >= o ? s[t - o] : 0ull;
      __syncthreads();
      s[t] += a;
      __syncthreads();
    }
  if (t < nbk) { boff[t] = s[t] - v; bcur[t] = s[t] - v; }
  if (t == nbk - 1) boff[nbk] = s[t];
  if (t == 0) *bnext = 0;
}

// ---- per-owner offsets ---------------------------------------------------------------------------------------------
// Every workgroup of pass 1 (and of kf_bigfix) is the OWNER of the chunks it filled and has counted its requests per
// bucket (row w of whist).  kl_tot sums the rows (-> the bucket sizes kl_scan turns into bucket offsets), kl_woff turns
// row w into the first output slot of owner w in every bucket.  A workgroup takes LW_BPW buckets (one 64-byte segment
// of every row); thread t = (slice of the owners t >> 4, bucket t & 15): the 16 lanes of a slice read one whole segment,
// a load instruction of a wave touches 4 lines instead of 64 (a wave per bucket with a lane per slice: 71 us for
// kl_woff, whatever the size of the table).
#define LW_BPW 16                          // buckets per workgroup (one 64-byte segment of a row)
#define LW_SL  64                          // slices of the owners per workgroup
#define LW_UNR 8                           // rows of a thread loaded together (independent loads in flight)

__global__ void __launch_bounds__(LW_SL * LW_BPW)
kl_tot(const unsigned *__restrict__ whist, unsigned nown, unsigned *__restrict__ tot /* [L_BK] */)
{ __shared__ unsigned red[LW_SL][LW_BPW];
  const int bb = threadIdx.x & (LW_BPW - 1), g = threadIdx.x / LW_BPW, b = blockIdx.x * LW_BPW + bb;
  const unsigned per = (nown + LW_SL - 1) / LW_SL, w0 = g * per, w1 = w0 + per < nown ? w0 + per : nown;
  unsigned s = 0;
  for (unsigned w = w0; w < w1; w += LW_UNR)
    { unsigned v[LW_UNR];
#pragma unroll
      for (int j = 0; j < LW_UNR; j++) v[j] = w + j < w1 ? whist[(size_t) (w + j) * L_BK + b] : 0u;
#pragma unroll
      for (int j = 0; j < LW_UNR; j++) s += v[j];
    }
  red[g][bb] = s;
  __syncthreads();
  if (g == 0)
    { unsigned a = 0;
      for (int q = 0; q < LW_SL; q++) a += red[q][bb];
      tot[b] = a;
    }
}

__global__ void __launch_bounds__(LW_SL * LW_BPW)
kl_woff(unsigned *__restrict__ whist /* in: counts, out: first slots */, unsigned nown, const u64 *__restrict__ boff)
{ __shared__ unsigned red[LW_SL][LW_BPW];
  const int bb = threadIdx.x & (LW_BPW - 1), g = threadIdx.x / LW_BPW, b = blockIdx.x * LW_BPW + bb;
  const unsigned per = (nown + LW_SL - 1) / LW_SL, w0 = g * per, w1 = w0 + per < nown ? w0 + per : nown;
  unsigned s = 0;
  for (unsigned w = w0; w < w1; w += LW_UNR)
    { unsigned v[LW_UNR];
#pragma unroll
      for (int j = 0; j < LW_UNR; j++) v[j] = w + j < w1 ? whist[(size_t) (w + j) * L_BK + b] : 0u;
#pragma unroll
      for (int j = 0; j < LW_UNR; j++) s += v[j];
    }
  red[g][bb] = s;
  __syncthreads();
  unsigned run = (unsigned) boff[b];                           // (slots are 32-bit: a shard holds < 2^32 requests)
  for (int q = 0; q < g; q++) run += red[q][bb];               // the slices in front of this one
  for (unsigned w = w0; w < w1; w += LW_UNR)
    { unsigned v[LW_UNR];
#pragma unroll
      for (int j = 0; j < LW_UNR; j++) v[j] = w + j < w1 ? whist[(size_t) (w + j) * L_BK + b] : 0u;
#pragma unroll
      for (int j = 0; j < LW_UNR; j++)
        if (w + j < w1) { whist[(size_t) (w + j) * L_BK + b] = run; run += v[j]; }
    }
}

// ---- partition ---------------------------------------------------------------------------------------------------
// Workgroup w takes the chunks of owner w (slots w, w + owners, ..), PT_CH at a time: count per bucket, sort by bucket INSIDE LDS,
// copy out in order -- neighbouring lanes then write neighbouring addresses (runs of ~16 records per bucket and batch;
// scattering the records straight from registers ran at a third of the speed).  The output cursors of the owner live
// in LDS: the first version took one slot range per (batch, bucket) from global cursors -- 5.5e7 atomics that cost
// 2.4 of its 3.4 ms.
#ifndef PT_TPB
#define PT_TPB    1024
#endif
#define PT_LDSREC 16384                    // records of one word per batch = 128 KB of LDS: one workgroup per CU
#define PT_BPT    (L_BK / PT_TPB)          // buckets per thread in the scan

// RW = 64-bit words per record (1: k <= 32, 2: 33 <= k <= 64; the bucket is taken from the first word).  A batch holds
// PT_LDSREC / RW records: four chunks of one-word records (two with 512 threads took 3.0 instead of 2.6 ms -- shorter
// runs), two chunks of two-word ones.
template <int RW> __global__ void __launch_bounds__(PT_TPB)
kl_part(const u64 *__restrict__ req, const uint32_t *__restrict__ chunk_fill, unsigned owners,
        const unsigned *__restrict__ woff, unsigned max_chunks, int nb, u64 *__restrict__ out)
{ constexpr int PT_BATCH = PT_LDSREC / RW;           // records per batch
  constexpr int PT_CH = PT_BATCH / F_CH;             // chunks per batch
  constexpr int PT_PER = PT_BATCH / PT_TPB;          // records per thread, held in registers between the phases
  __shared__ u64      sorted[PT_BATCH * RW];
  __shared__ unsigned cur[L_BK];           // phase A: counts; phase C: cursors
  __shared__ unsigned lbase[L_BK];         // first slot of the bucket in `sorted`
  __shared__ unsigned gbase[L_BK];         // first output slot of this batch's run
  __shared__ unsigned gcur[L_BK];          // the owner's next output slot per bucket
  __shared__ unsigned wsum[PT_TPB / 64];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int hsh = 32 - nb;
  for (int b = t; b < L_BK; b += PT_TPB) { cur[b] = 0; gcur[b] = woff[(size_t) blockIdx.x * L_BK + b]; }

  // batch q = the chunks blockIdx.x + (q * PT_CH + j) * owners, j < PT_CH, of this owner (fill 0: not used).
  // The PT_PER records of a thread; `ok` = which of them exist, bit 31 = the batch's first chunk is in use.
  auto load = [&](unsigned q, u64 (&y)[PT_PER][RW]) -> unsigned
  { unsigned ok = 0;
#pragma unroll
    for (int j = 0; j < PT_PER; j++)
      { const unsigned cj = (unsigned) j / (PT_PER / PT_CH), r = ((unsigned) j % (PT_PER / PT_CH)) * PT_TPB + t;
        const u64 ch = (u64) blockIdx.x + (u64) (q * PT_CH + cj) * owners;
        const unsigned fill = ch < max_chunks ? chunk_fill[ch] : 0u;
        const bool k = r < fill;
        ok |= (unsigned) k << j;
        if (j == 0 && fill) ok |= 1u << 31;
        if constexpr (RW == 2)
          { ulonglong2 v = make_ulonglong2(0ull, 0ull);
            if (k) v = *reinterpret_cast<const ulonglong2 *>(req + ((size_t) ch * F_CH + r) * 2);
            y[j][0] = v.x; y[j][1] = v.y;
          }
        else y[j][0] = k ? req[(size_t) ch * F_CH + r] : 0ull;
      }
    return ok;
  };

  // The barriers order LDS traffic only (lds_barrier): a __syncthreads() also drains the wave's outstanding stores,
  // which made every batch wait for its own copy-out.  The NEXT batch is loaded while this one is sorted.
  u64 y[PT_PER][RW], yn[PT_PER][RW];
  unsigned ok = load(0u, y), okn = 0;
  lds_barrier();
  for (unsigned q = 0; ok >> 31; q++)
    { // A: count per bucket
#pragma unroll
      for (int j = 0; j < PT_PER; j++)
        if (ok >> j & 1u) atomicAdd(&cur[(unsigned) (y[j][0] >> 32) >> hsh], 1u);
      okn = load(q + 1, yn);                                     // (in flight until the end of this batch)
      lds_barrier();
      // B: exclusive scan of the counts (PT_BPT buckets per thread); the owner's cursors move on
      { unsigned cbk[PT_BPT], incl = 0;
#pragma unroll
        for (int j = 0; j < PT_BPT; j++) { cbk[j] = cur[PT_BPT * t + j]; incl += cbk[j]; }
        const unsigned mine = incl;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1)
          { const unsigned v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
          }
        if (lane == 63) wsum[wv] = incl;
        lds_barrier();
        unsigned woffs = 0;
        for (int w = 0; w < wv; w++) woffs += wsum[w];
        unsigned ex = woffs + incl - mine;
#pragma unroll
        for (int j = 0; j < PT_BPT; j++)
          { const int b = PT_BPT * t + j;
            lbase[b] = ex; cur[b] = ex;
            const unsigned g = gcur[b];
            gbase[b] = g; gcur[b] = g + cbk[j];
            ex += cbk[j];
          }
      }
      lds_barrier();
      unsigned total = 0;
#pragma unroll
      for (int w = 0; w < PT_TPB / 64; w++) total += wsum[w];
      // C: sort inside LDS (the order inside a bucket is arbitrary)
#pragma unroll
      for (int j = 0; j < PT_PER; j++)
        if (ok >> j & 1u)
          { const unsigned slot = atomicAdd(&cur[(unsigned) (y[j][0] >> 32) >> hsh], 1u);
#pragma unroll
            for (int w = 0; w < RW; w++) sorted[slot * RW + w] = y[j][w];
          }
      lds_barrier();
#pragma unroll
      for (int j = 0; j < PT_BPT; j++) cur[PT_BPT * t + j] = 0;   // (for the next batch; D does not read them)
      // D: copy out in order
      for (unsigned i = t; i < total; i += PT_TPB)
        { const unsigned b = (unsigned) (sorted[i * RW] >> 32) >> hsh;
          const size_t o = (size_t) gbase[b] + (i - lbase[b]);
          if constexpr (RW == 2)
            *reinterpret_cast<ulonglong2 *>(out + o * 2) = make_ulonglong2(sorted[i * 2], sorted[i * 2 + 1]);
          else out[o] = sorted[i];
        }
#pragma unroll
      for (int j = 0; j < PT_PER; j++)
#pragma unroll
        for (int w = 0; w < RW; w++) y[j][w] = yn[j][w];
      ok = okn;
      lds_barrier();
    }
}

// ---- probe: block-map filter (LDS coarse slice, then the full map) + look-ups or survivor list ------------------------
// One 1024-thread workgroup per CU owns a bucket at a time and keeps the bucket's part of the map, folded 4:1, in LDS.
// Inside a bucket its 16 waves work on their own: a wave streams its share of the records, queues its survivors in its
// own corner of LDS and, whenever 64 have gathered, looks them up with every lane busy.  No workgroup barrier inside a
// bucket: the first version synchronised the workgroup five times per 8192 records and spent most of its 4.2 ms waiting.
#define PB_TPB    1024
#define PB_WAVES  (PB_TPB / 64)
#define PB_PER    8                        // records per lane and iteration
#define PB_WQ     192                      // queue slots per wave: < 64 left over + two wave-instructions
#define PB_SLICEW (1 << (L_SLICE_LG - 5))  // 32768 words

// 32 map bits -> 8 coarse bits (bit j = OR of the bits 4j .. 4j+3)
SMG_DEV unsigned pb_fold(unsigned f)
{ unsigned c = f | (f >> 1);
  c = (c | (c >> 2)) & 0x11111111u;
  c = (c | (c >> 3)) & 0x03030303u;
  c = (c | (c >> 6)) & 0x000F000Fu;
  return (c | (c >> 12)) & 0xFFu;
}

// LIST = false: look the survivors up and set their P flags.  LIST = true: append them to the chunk list `out`
// (every wave fills chunks of its own).
// RW = words per record: the filter reads the first one; the survivors' queue holds the k-mer itself (RW = 1) or the
// number of the record (RW = 2: two more queue words per slot would not fit next to the 128 KB map slice)
template <bool LIST, bool TWO, int RW> __global__ void __launch_bounds__(PB_TPB)
kl_probe(FastArgs A, const u64 *__restrict__ recs, const u64 *__restrict__ boff, const uint32_t *__restrict__ fmap,
         LookupGeo g, unsigned *__restrict__ bnext, u64 *__restrict__ out, uint32_t *__restrict__ out_fill,
         unsigned max_out, FastCtl *__restrict__ ctl)
{ __shared__ unsigned cmap[PB_SLICEW];
  __shared__ u64      wq[PB_WAVES][PB_WQ];
  __shared__ unsigned s_b;
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nbk = 1 << g.nb;
  const int slice_lg = g.cb - g.nb;                  // coarse bits per bucket: 9 .. 20
  const unsigned ncw = 1u << (slice_lg - 5);         // coarse words per bucket
  const unsigned smask = (1u << slice_lg) - 1u;
  u64 *q = wq[wv];
  unsigned qn = 0;                                   // this wave's queue fill (uniform)
  unsigned chunk = F_NOCHUNK, used = 0;              // LIST: this wave's output chunk
  u64 kept = 0;

  // the last `take` (<= 64) survivors of this wave's queue
  auto drain = [&](unsigned take)
  { const u64 yv = q[qn - take + ((unsigned) lane < take ? lane : 0)];
    Key<RW> y;
    if constexpr (RW == 1) y.w[0] = yv;
    else
      { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(recs + (size_t) yv * 2);     // (read a moment ago: a cache hit)
        y.w[0] = v.x; y.w[1] = v.y;
      }
    if (!LIST)
      { if ((unsigned) lane < take) lookup_one<RW>(A, y, ctl);
      }
    else
      { if (chunk == F_NOCHUNK || used + take > F_CH)
          { if (chunk != F_NOCHUNK && chunk < max_out && lane == 0) out_fill[chunk] = used;
            unsigned c = 0;
            if (lane == 0) c = atomicAdd(&ctl->nf_chunks, 1u);
            chunk = (unsigned) __builtin_amdgcn_readfirstlane((int) c);
            used = 0;
          }
        if ((unsigned) lane < take && chunk < max_out)
          {
#pragma unroll
            for (int w = 0; w < RW; w++) out[((size_t) chunk * F_CH + used + lane) * RW + w] = y.w[w];
          }
        used += take;
      }
    qn -= take; kept += take;
  };

  for (;;)
    { __syncthreads();
      if (t == 0) s_b = atomicAdd(bnext, 1u);
      __syncthreads();
      const unsigned b = s_b;
      if (b >= (unsigned) nbk) break;
      const u64 r0 = boff[b], r1 = boff[b + 1];
      if (r0 == r1) continue;
      // the bucket's part of the map, folded 4:1 into LDS  (fb >= 12: at least 2^9 coarse bits per bucket)
      { const uint32_t *fw = fmap + (((size_t) b << (g.fb - g.nb - 5)) << (TWO ? 1 : 0));
        for (unsigned cw = t; cw < ncw; cw += PB_TPB)
          { if (TWO)                                  // 64-bit map words: the first bits are the even 32-bit words
              { const uint4 f0 = *reinterpret_cast<const uint4 *>(fw + 8 * cw), f1 = *reinterpret_cast<const uint4 *>(fw + 8 * cw + 4);
                cmap[cw] = pb_fold(f0.x) | (pb_fold(f0.z) << 8) | (pb_fold(f1.x) << 16) | (pb_fold(f1.z) << 24);
              }
            else
              { const uint4 f = *reinterpret_cast<const uint4 *>(fw + 4 * cw);
                cmap[cw] = pb_fold(f.x) | (pb_fold(f.y) << 8) | (pb_fold(f.z) << 16) | (pb_fold(f.w) << 24);
              }
          }
      }
      __syncthreads();
      for (u64 i0 = r0 + (u64) wv * (64 * PB_PER); i0 < r1; i0 += (u64) PB_WAVES * 64 * PB_PER)
        { u64 y[PB_PER]; bool keep[PB_PER]; unsigned fwd[PB_PER], fw2[PB_PER];
#pragma unroll
          for (int j = 0; j < PB_PER; j++)
            { const u64 i = i0 + (u64) j * 64 + lane;
              keep[j] = i < r1;
              y[j] = keep[j] ? recs[i * RW] : 0ull;
            }
#pragma unroll
          for (int j = 0; j < PB_PER; j++)
            { const unsigned cid = (unsigned) (y[j] >> (64 - g.cb)) & smask;
              keep[j] = keep[j] && ((cmap[cid >> 5] >> (cid & 31)) & 1u);
            }
#pragma unroll
          for (int j = 0; j < PB_PER; j++)                     // the full-resolution map: global memory, ~20 % of the lanes
            { const unsigned fid = (unsigned) (y[j] >> (64 - g.fb));
              if (TWO)
                { const uint2 f = keep[j] ? reinterpret_cast<const uint2 *>(fmap)[fid >> 5] : make_uint2(0u, 0u);
                  fwd[j] = f.x; fw2[j] = f.y;
                }
              else { fwd[j] = keep[j] ? fmap[fid >> 5] : 0u; fw2[j] = 0u; }
            }
#pragma unroll
          for (int j = 0; j < PB_PER; j++)
            { const unsigned fid = (unsigned) (y[j] >> (64 - g.fb));
              keep[j] = keep[j] && ((fwd[j] >> (fid & 31)) & 1u);
              if (TWO) keep[j] = keep[j] && ((fw2[j] >> bm2_pos((uint32_t) y[j])) & 1u);
            }
#pragma unroll
          for (int j = 0; j < PB_PER; j += 2)                  // queue two wave-instructions' worth, drain to below 64
            { const u64 m0 = __ballot(keep[j]), m1 = __ballot(keep[j + 1]);
              const unsigned n0 = (unsigned) __popcll(m0), n1 = (unsigned) __popcll(m1);
              const u64 v0 = RW == 1 ? y[j] : i0 + (u64) j * 64 + lane, v1 = RW == 1 ? y[j + 1] : i0 + (u64) (j + 1) * 64 + lane;
              if (keep[j]) q[qn + __builtin_amdgcn_mbcnt_hi((unsigned) (m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m0, 0u))] = v0;
              if (keep[j + 1]) q[qn + n0 + __builtin_amdgcn_mbcnt_hi((unsigned) (m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m1, 0u))] = v1;
              qn += n0 + n1;
              while (qn >= 64) drain(64);
            }
        }
      if (qn) drain(qn);                                       // (the look-ups of a bucket stay inside its 1/2^nb of the table)
    }
  if (LIST && chunk != F_NOCHUNK && chunk < max_out && lane == 0) out_fill[chunk] = used;
  if (kept && lane == 0) atomicAdd(&ctl->nf_req, kept);
}

// ---- kl_probe_x: the same test without the LDS map, XCD by XCD -------------------------------------------------------
// kl_probe folds a bucket's slice of the map 4:1 into LDS because a bucket's requests hit its 1 MB of map at random:
// from 256 buckets in flight that is 256 MB of working set, every probe that passes the coarse test costs a 128-byte
// line of HBM for 8 bytes (12 of the kernel's 17 GB).  Here the workgroups of ONE XCD take the SAME bucket at the same
// time -- buckets b = xcd (mod 8) belong to XCD xcd, its workgroups draw (bucket, part) tickets from the XCD's own
// counter, PX_PART requests per ticket -- so the slice in use is 1-2 MB per XCD and stays in that XCD's 4 MB L2: every
// request probes the full-resolution map word directly, the map crosses HBM once.  The request stream is read with
// non-temporal loads (it is touched once and must not push the map out).  Survivors are looked up on the spot as in
// kl_probe.  Needs >= 8 buckets (nb >= 3); the XCD is the hardware's XCC id (HW_REG_XCC_ID: the dispatcher's round-robin is
// not taken for granted, nor is the number of XCDs -- see the stealing loop).
#define PX_TPB   256
#define PX_PART  2048                     // requests per ticket on a table with many survivors (~0.6 buckets in flight per XCD; lookup_probe doubles it elsewhere)
#define PX_WGS   4
#define PX_PER   8                        // requests per lane and step
#define PX_NXCD  8
#define PX_TICKW 32                       // words between two XCD counters

template <bool TWO, int RW> __global__ void __launch_bounds__(PX_TPB)
kl_probe_x(FastArgs A, const u64 *__restrict__ recs, const u64 *__restrict__ boff, const uint32_t *__restrict__ fmap,
           LookupGeo g, unsigned *__restrict__ xtick, FastCtl *__restrict__ ctl, unsigned part /* requests per ticket */)
{ __shared__ unsigned pstart[L_BK / PX_NXCD + 1];        // first ticket of the XCD's j-th bucket
  __shared__ u64      wq[PX_TPB / 64][PB_WQ];
  __shared__ unsigned s_item;
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= PX_NXCD - 1;
  if (part >> 31) { xcc = 0; part &= 0x7FFFFFFFu; }        // test hook (SMG_PX_ONE_XCC): a device that shows ONE XCC id
  const int nbk = 1 << g.nb, mine = nbk / PX_NXCD;        // (nb >= 3)
  u64 *q = wq[wv];
  unsigned qn = 0;
  u64 kept = 0;
  auto drain = [&](unsigned take)
  { const u64 yv = q[qn - take + ((unsigned) lane < take ? lane : 0)];
    Key<RW> y;
    if constexpr (RW == 1) y.w[0] = yv;
    else
      { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(recs + (size_t) yv * 2);
        y.w[0] = v.x; y.w[1] = v.y;
      }
    if ((unsigned) lane < take) lookup_one<RW>(A, y, ctl);
    qn -= take; kept += take;
  };
  // The buckets of this workgroup's own XCD first, then -- once those are handed out -- the other classes' in turn: the
  // tail is shared, and a device that shows fewer XCC ids than eight (another partition mode) still does every bucket.
  for (int st = 0; st < PX_NXCD; st++)
  { const unsigned cls = (xcc + (unsigned) st) & (PX_NXCD - 1);
    __syncthreads();
    if (t < 64)                                            // tickets per bucket of this class: an exclusive scan, by one wave
      { unsigned run = 0;
        for (int j0 = 0; j0 < mine; j0 += 64)
          { const int j = j0 + lane;
            unsigned np = 0;
            if (j < mine) { const int b = j * PX_NXCD + (int) cls; np = (unsigned) ((boff[b + 1] - boff[b] + part - 1) / part); }
            unsigned incl = np;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
            if (j < mine) pstart[j] = run + incl - np;
            run += __shfl(incl, 63, 64);
          }
        if (lane == 0) pstart[mine] = run;
      }
    __syncthreads();
    const unsigned nitem = pstart[mine];
  for (;;)
    { __syncthreads();
      if (t == 0) s_item = atomicAdd(&xtick[cls * PX_TICKW], 1u);
      __syncthreads();
      const unsigned item = s_item;
      if (item >= nitem) break;
      int lo = 0, hi = mine;                               // the bucket of this ticket: last j with pstart[j] <= item
      while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (pstart[m] <= item) lo = m; else hi = m; }
      const int b = lo * PX_NXCD + (int) cls;
      const u64 r0 = boff[b] + (u64) (item - pstart[lo]) * part;
      const u64 r1 = r0 + part < boff[b + 1] ? r0 + part : boff[b + 1];
      for (u64 i0 = r0 + (u64) wv * (64 * PX_PER); i0 < r1; i0 += (u64) (PX_TPB / 64) * 64 * PX_PER)
        { u64 y[PX_PER]; bool keep[PX_PER]; unsigned fwd[PX_PER], fw2[PX_PER];
#pragma unroll
          for (int j = 0; j < PX_PER; j++)
            { const u64 i = i0 + (u64) j * 64 + lane;
              keep[j] = i < r1;
              y[j] = keep[j] ? __builtin_nontemporal_load(recs + i * RW) : 0ull;
            }
#pragma unroll
          for (int j = 0; j < PX_PER; j++)
            { const unsigned fid = (unsigned) (y[j] >> (64 - g.fb));
              if (TWO)
                { const uint2 f = keep[j] ? reinterpret_cast<const uint2 *>(fmap)[fid >> 5] : make_uint2(0u, 0u);
                  fwd[j] = f.x; fw2[j] = f.y;
                }
              else { fwd[j] = keep[j] ? fmap[fid >> 5] : 0u; fw2[j] = 0u; }
            }
#pragma unroll
          for (int j = 0; j < PX_PER; j++)
            { const unsigned fid = (unsigned) (y[j] >> (64 - g.fb));
              keep[j] = keep[j] && ((fwd[j] >> (fid & 31)) & 1u);
              if (TWO) keep[j] = keep[j] && ((fw2[j] >> bm2_pos((uint32_t) y[j])) & 1u);
            }
#pragma unroll
          for (int j = 0; j < PX_PER; j += 2)
            { const u64 m0 = __ballot(keep[j]), m1 = __ballot(keep[j + 1]);
              const unsigned n0 = (unsigned) __popcll(m0), n1 = (unsigned) __popcll(m1);
              const u64 v0 = RW == 1 ? y[j] : i0 + (u64) j * 64 + lane, v1 = RW == 1 ? y[j + 1] : i0 + (u64) (j + 1) * 64 + lane;
              if (keep[j]) q[qn + __builtin_amdgcn_mbcnt_hi((unsigned) (m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m0, 0u))] = v0;
              if (keep[j + 1]) q[qn + n0 + __builtin_amdgcn_mbcnt_hi((unsigned) (m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m1, 0u))] = v1;
              qn += n0 + n1;
              while (qn >= 64) drain(64);
            }
        }
      if (qn) drain(qn);
    }
  }
  if (kept && lane == 0) atomicAdd(&ctl->nf_req, kept);
}

