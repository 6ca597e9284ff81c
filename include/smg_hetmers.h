/*******************************************************************************************
 *
 *  smg_hetmers.h -- C ABI of the MI355X-native heterozygous k-mer pair engine.
 *
 *  The reference (KamilSJaron/smudgeplot) has no in-process plugin API for this path: its
 *  boundary is the `hetmers` PROCESS (src/smudgeplot/cli.py:57-72 run_binary) plus two file
 *  formats (FastK .ktab in, .smu out).  The drop-in executable lives in
 *  smudgeplot_amd/csrc/hetmers_main.c; it is plain C and reaches the GPU only through the
 *  entry points declared here, which are what a cgo/ctypes/JNI binding of this path would bind.
 *  Each entry point names the reference code it stands in for.
 *
 *  Conventions (all entry points):
 *    - plain pointers and sizes only; no C++/torch types; `void *stream` is a hipStream_t
 *      (NULL = the device's default stream);
 *    - return 0 on success, a negative SMG_E* code on failure with a message in errbuf
 *      (never calls exit(), never prints unless opts->verbose asks for timing lines on stderr);
 *    - NO CPU fallback: without a usable gfx950 device every compute call fails with
 *      SMG_ENODEV.  (The CPU restatement lives under oracle/ and is test infrastructure.)
 *    - one engine per device per thread of control; an engine is not re-entrant.
 *
 ********************************************************************************************/

#ifndef SMG_HETMERS_H
#define SMG_HETMERS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMG_SMAX        1000                     /* PloidyPlot.c:48  max covA+covB            */
#define SMG_FMAX         500                     /* PloidyPlot.c:49  max min(covA,covB)       */
#define SMG_PLOT_ROWS   (SMG_SMAX + 1)
#define SMG_PLOT_COLS   (SMG_FMAX + 1)
#define SMG_PLOT_CELLS  (SMG_PLOT_ROWS * SMG_PLOT_COLS)   /* int64 plot[sum][min], row major   */
#define SMG_MAX_KMER     128                     /* keys are 1..4 64-bit words                */

#define SMG_OK        0
#define SMG_ENODEV   -1      /* no usable HIP device / HIP runtime error                     */
#define SMG_EINVAL   -2      /* bad argument (k out of range, n too large, null pointer ...)  */
#define SMG_ENOMEM   -3      /* device or host allocation failed                             */
#define SMG_EFORMAT  -4      /* table violates format F (unsorted, duplicate k-mers ...)      */
#define SMG_ENOTSYM  -5      /* sharded run on a table that is not reverse-complement closed */

/* How the engine proves that the table is closed under reverse complement with equal counts
   (the property `Symmex` establishes; the reference only spot-checks entry #1,
   PloidyPlot.c:1199-1229).  If the proof fails the engine silently switches to the general
   all-positions path, which assumes nothing, so the answer is the reference's either way. */
#define SMG_SYM_EXACT  0     /* look up the complement of EVERY entry (exact, ~4x slower)      */
#define SMG_SYM_HASH   1     /* 128-bit XOR multiset fingerprint of canonical (k-mer, count) + exact */
                             /* look-ups for the entries that own a pair (what the `hetmers`  */
                             /* executable and bench.py use unless told otherwise)            */
#define SMG_SYM_NONE   2     /* never use the symmetry identity: general path only            */

/* A FastK table in host memory, exactly as it sits on disk minus the file headers.
   Replaces: Kmer_Stream / Open_Kmer_Stream, src/lib/libfastk.c:717-745, 786-908.            */
typedef struct smg_table_view
{ int32_t               kmer;          /* k                                                    */
  int32_t               ibyte;         /* prefix bytes folded into the index: 1, 2 or 3        */
  int32_t               nparts;        /* number of part files                                 */
  int32_t               minval;        /* stub header field, informational                     */
  int64_t               nels;          /* total entries                                        */
  const uint8_t *const *part_data;     /* [nparts] raw records, (kbyte+2-ibyte) bytes each     */
  const int64_t        *part_nels;     /* [nparts] entries per part                            */
  const int64_t        *prefix_index;  /* [2^(8*ibyte)] cumulative END offsets                 */
} smg_table_view;

/* A FastK table that is still on disk (or anywhere else): the engine PULLS the records, in pieces of a few tens of
   megabytes, from up to `host_threads` threads at once, into pinned buffers from which they go to the device while
   the next pieces are being read -- the host never holds the table.
   Replaces: More_Kmer_Stream / GoTo_Kmer_Index, src/lib/libfastk.c:759-784, 1273-1307 (1024-record read(2) blocks).  */
typedef struct smg_table_source
{ int32_t        kmer, ibyte, nparts, minval;
  int64_t        nels;
  const int64_t *part_nels;      /* [nparts]                                                               */
  const int64_t *prefix_index;   /* [2^(8*ibyte)] cumulative END offsets (host memory)                    */
  /* copy `nent` records of part `part`, starting at its entry `first`, to dst; 0 = success.  Thread safe. */
  int          (*read)(void *ctx, int part, int64_t first, int64_t nent, void *dst);
  void          *ctx;
  int32_t        host_threads;   /* reader threads (the -T of the command line); <= 0: 4                   */
} smg_table_source;

/* Table conditioning the reference delegates to FastK's Logex / Symmex
   (PloidyPlot.c:1381-1414), done on the device instead.                                     */
#define SMG_COND_TRIM  1     /* drop entries with count < ethresh          (Logex 'A[e-]')    */
#define SMG_COND_SYMM  2     /* add the reverse complement of every entry  (Symmex)           */

typedef struct smg_opts
{ int32_t device;        /* HIP device ordinal                                               */
  int32_t symcheck;      /* SMG_SYM_*                                                        */
  int32_t verbose;       /* >0: per-phase timing lines on stderr                             */
  int32_t condition;     /* SMG_COND_* bits: condition the table before the scan             */
  int32_t ethresh;       /* -e threshold for SMG_COND_TRIM                                   */
  int32_t ngpus;         /* > 1: smg_hetmers_run shards the table by k-mer prefix over the HIP    */
                         /* devices device .. device+ngpus-1 (one host thread each, RCCL request  */
                         /* exchange + histogram all-reduce); 0 / 1: one GPU                      */
} smg_opts;

typedef struct smg_stats
{ int64_t nels;          /* entries scanned                                                  */
  int64_t npairs;        /* one-away pairs that entered the histogram (weighted)             */
  int64_t nrequests;     /* complement look-ups issued (after the request filter)            */
  int32_t path;          /* 1 = symmetric half-scan, 2 = general all-positions path          */
  int32_t key_words;     /* 64-bit words per k-mer                                           */
  double  ms_h2d;        /* host -> device copies                                            */
  double  ms_decode;     /* format F records -> device table                                 */
  double  ms_pass1;      /* window scan + degree                                             */
  double  ms_rclookup;   /* complement look-ups / degree exchange                            */
  double  ms_pass2;      /* unique-pair histogram                                            */
  double  ms_total;      /* decode .. histogram on device (no H2D)                           */
  int64_t nemitted;      /* complement requests pass 1 emitted (before the request filter)   */
  double  ms_filter;     /* request filter (also counted in ms_rclookup)                     */
  int64_t nbig;          /* entries whose window block outgrew the +-30 entry window (redone   */
                         /* exactly by kf_bigfix: low-complexity / repeat k-mers)              */
  double  ms_bigfix;     /* ... and the time of that (also counted in ms_pass1)                */
} smg_stats;

/* ---- one-shot entry: host FastK table -> plot ------------------------------------------
   Replaces the compute section of main(), PloidyPlot.c:1433-1575 (both passes over the
   conditioned table and the reduction of the per-thread plots).
   plot: caller-allocated int64[SMG_PLOT_CELLS], overwritten.  stats may be NULL.            */
int smg_hetmers_run(const smg_table_view *table, const smg_opts *opts, int64_t *plot,
                    smg_stats *stats, char *errbuf, size_t errlen);

/* the same from a table source (what the `hetmers` executable uses: smg_ktab_open + smg_ktab_read)      */
int smg_hetmers_run_source(const smg_table_source *source, const smg_opts *opts, int64_t *plot,
                           smg_stats *stats, char *errbuf, size_t errlen);

/* ---- extract: the pairs behind the labelled pixels ----------------------------------------
   Replaces the compute section of extract_kmer_pairs, src/lib/PloidyList.c:1207-1583 (same two
   passes as hetmers; the pass-2 sink prints the pair instead of counting it, PloidyList.c:424-448).
   labels: host uint16[SMG_PLOT_CELLS], labels[sum*501+min] = 1-based smudge number of the pixel, 0 =
   not annotated (PloidyList.c:1312-1346 PLOT[i+j][i] = s+1).
   *records: malloc'ed by the library (release with smg_free), *nrec records of *rec_words uint64:
     words 0 .. rec_words-2 : the k-mer to print, left aligned (base 0 in bits 63..62 of word 0)
     last word              : variant position | alt base (0..3 = acgt) << 8 | label << 16
   One record per printed line of the reference (print_het, PloidyList.c:128-165); the order of the
   records is unspecified (so is the order of the reference's lines: mutex-guarded fprintf from threads).
   plot receives the same histogram smg_hetmers_run computes.                                    */
int smg_hetmers_extract(const smg_table_view *table, const smg_opts *opts, const uint16_t *labels,
                        int64_t *plot, uint64_t **records, int64_t *nrec, int *rec_words,
                        smg_stats *stats, char *errbuf, size_t errlen);
void smg_free(void *p);

/* ---- conditioning as a service: trim / symmetrise a host table, get the conditioned table back -------
   Stands in for running FastK's `Logex '<out>=A[e-]'` and `Symmex` by hand (the reference's own use of
   them: PloidyPlot.c:1381-1414).  opts->condition / opts->ethresh select the steps.
   *keys_out: malloc'ed (smg_free) *nels_out * *words_out uint64, k-mers left aligned, sorted;
   *counts_out: malloc'ed uint16[*nels_out].                                                          */
int smg_condition_table(const smg_table_view *table, const smg_opts *opts, uint64_t **keys_out,
                        uint16_t **counts_out, int64_t *nels_out, int *words_out,
                        char *errbuf, size_t errlen);

/* number of usable HIP devices (0 when there is none or the runtime is missing)             */
int smg_device_count(void);


/* ---- engine object: device-resident table, phase-level calls ---------------------------
   Used by the one-shot entry, by bench.py and by the one-process-per-GPU sharded driver
   (smudgeplot_amd/sharded.py), which interleaves the phases with RCCL collectives issued
   through torch.distributed.  Device pointers handed in must belong to `device`.            */
typedef struct smg_engine smg_engine;

smg_engine *smg_engine_create(int device, void *stream, char *errbuf, size_t errlen);
void        smg_engine_destroy(smg_engine *e);

/* Decode `nels` format-F records already in device memory (parts concatenated, file headers
   stripped) plus the stub's prefix index into the engine's own table.
   Replaces Current_Entry / Next_Kmer_Entry expansion, libfastk.c:1159-1176, 1230-1269, and
   the cache fill of small_recursion, PloidyPlot.c:954-961.                                   */
int smg_engine_decode(smg_engine *e, int kmer, int ibyte, int64_t nels,
                      const uint8_t *d_records, const int64_t *d_prefix_index,
                      char *errbuf, size_t errlen);

/* Bind an already decoded device table (not copied; must outlive the engine's use of it):
   d_keys  = nels * words 64-bit words, k-mer left aligned (base 0 in bits 63..62 of word 0),
             words = ceil(k/32), entries strictly increasing;
   d_counts= nels uint16.
   While a table is bound its K-MERS must not change (the engine keeps what it has learnt about them between runs: the
   first and the last k-mer, the directory made of the table's prefix index); its COUNTS may -- every run reads them
   again, and a step that was queued from the counts of the step before (replayed phase steps, below) notices the
   difference and is repeated the plain way.  Bind again after changing k-mers.                                     */
int smg_engine_bind(smg_engine *e, int kmer, int64_t nels, const uint64_t *d_keys,
                    const uint16_t *d_counts, char *errbuf, size_t errlen);

/* Hand over the prefix index that a FastK table carries in its stub (`index[p]` = number of entries whose first ibyte
   bytes are <= p, 2^(8 ibyte) int64 in DEVICE memory; libfastk.c:841, the table GoTo_Kmer_Entry bisects on,
   libfastk.c:1360-1386) for the table that is bound / decoded right now; first_entry = number of this shard's first
   entry in the whole table (0 for a whole table).  With ibyte = 3 and k >= 12 the engine takes it as its look-up
   directory (a bucket per 24 leading k-mer bits) and pass 1 builds none of its own; a coarser index is accepted and not
   used.  The index is read during this call's stream work only (it may be freed after the stream has drained).
   Binding, decoding or conditioning another table forgets it.                                                       */
int smg_engine_set_prefix_index(smg_engine *e, const int64_t *d_prefix_index, int ibyte, int64_t first_entry,
                                char *errbuf, size_t errlen);

/* Condition the engine's table in place (decoded or bound; the result is engine-owned):
   trim to count >= ethresh and / or close it under reverse complement.  *new_nels (may be NULL)
   receives the new number of entries.
   Replaces the Logex / Symmex / Fastrm shell-outs, PloidyPlot.c:1381-1414, 1584-1592.        */
int smg_engine_condition(smg_engine *e, int ethresh, int do_trim, int do_symm,
                         int64_t *new_nels, char *errbuf, size_t errlen);

/* ---- symmetrising a table that is cut into prefix shards (several GPUs, or more than 2^32 entries) --------------
   The reference hands a table of any size to Symmex (PloidyPlot.c:1395-1414).  Sharded, the same closure takes one
   exchange: every shard sends each of its entries AND the reverse complement of each to the shard whose k-mer range
   holds it, the receivers sort and keep one entry per k-mer.
   symm_hist  : hist[0 .. 2^bits) = this shard's entries, hist[2^bits .. 2^(bits+1)) = their reverse complements, per
                leading `bits` (1..12, <= 2k) k-mer bits (host array).  Summed over the shards it is the shape of the
                closed table: the caller takes balanced splitters from it (a canonical table crowds the low end of
                the k-mer space, its closure does not).  Splitters on such a boundary are window-block boundaries
                as long as bits <= 2 * (k / 2).
   symm_route : 2 * nels records of (words + 1) uint64 -- the k-mer, then count | is-a-complement << 16 -- grouped by
                destination rank into d_send (capacity in records >= 2 * nels); counts[nranks] on the host.
                splitters as for smg_engine_route.
   symm_finish: the records this shard received (its own share included) -> the engine's table: sorted, one entry
                per k-mer (a k-mer that arrives both as an entry and as a complement keeps the entry's count).
                *new_nels (may be NULL) = entries of the shard.  The engine owns the table afterwards.            */
int smg_engine_symm_hist(smg_engine *e, int bits, int64_t *hist, char *errbuf, size_t errlen);
int smg_engine_symm_route(smg_engine *e, const uint64_t *splitters, int nranks, uint64_t *d_send,
                          int64_t capacity, int64_t *counts, char *errbuf, size_t errlen);
int smg_engine_symm_finish(smg_engine *e, const uint64_t *d_recv, int64_t nrecv, int64_t *new_nels,
                           char *errbuf, size_t errlen);
/* the engine's current table (bound, decoded or conditioned): entries, device pointers (any may be NULL)         */
int smg_engine_table(smg_engine *e, int64_t *nels, const uint64_t **d_keys, const uint16_t **d_counts);

/* Whole single-GPU computation on the bound/decoded table; d_plot = int64[SMG_PLOT_CELLS]
   in device memory, overwritten.  Asynchronous on the engine's stream except for the small
   control read-backs between phases.                                                         */
int smg_engine_run(smg_engine *e, int symcheck, int64_t *d_plot, smg_stats *stats,
                   char *errbuf, size_t errlen);

/* (smg_hetmers_run / _run_source on a table that does not fit the device run it OUT OF CORE, prefix shards one after the other
   with the table read twice: there symcheck = SMG_SYM_NONE is taken as SMG_SYM_HASH -- the shards cannot help each other on a
   table that is not closed, which is refused with SMG_ENOTSYM and the advice to condition it first.  A table that opts->condition
   asks to be trimmed / symmetrised is conditioned out of core as well: its closed prefix shards are parked in host memory.)  */

/* ---- sharded (one process per GPU) phase calls ------------------------------------------
   The table is split by k-mer PREFIX: rank r owns the entries in [splitter[r-1], splitter[r]).
   Every position the half-scan visits is shard local; the only cross-shard traffic is one
   record per entry that owns a suffix-side pair, sent to the rank that owns its complement.

   pass1   : window scan, fills degrees, builds the request list, accumulates the fingerprints.
   nreq    : number of pending requests.
   route   : writes the requests grouped by destination rank into d_send (capacity in records)
             and the per-rank counts into counts[nranks] (host).  A record is
             (words+1) uint64: the complement k-mer, then count | value<<16.
             splitters = (nranks-1)*words host uint64, the first k-mer of ranks 1..nranks-1.
             Bound device tables must be 16-byte (k-mers) / 8-byte (counts) aligned.
   apply   : adds received (or own) requests to the local degrees; *missing counts requests
             whose k-mer is absent or carries another count (=> table not symmetric).
   symhash : out[0..1] = 128-bit canonical XOR-fingerprint residue of the shard (XOR over the ranks
             : zero iff T == rc(T) with equal counts, up to a 2^-128 collision),
             out[2..3] = 0 (the value to compare with).
   pass2   : histogram of unique pairs into d_plot (device int64[SMG_PLOT_CELLS], overwritten).*/
int     smg_engine_pass1(smg_engine *e, int symcheck, char *errbuf, size_t errlen);
int64_t smg_engine_nreq(smg_engine *e);
int     smg_engine_record_words(smg_engine *e);
int     smg_engine_route(smg_engine *e, const uint64_t *splitters, int nranks, uint64_t *d_send,
                         int64_t capacity, int64_t *counts, char *errbuf, size_t errlen);
/* the same with the per-rank counts left on the device (d_counts[nranks], int64, written in stream order): a driver that
   exchanges the counts with a collective hands them over without a host round trip and reads both sides' counts once */
int     smg_engine_route_device(smg_engine *e, const uint64_t *splitters, int nranks, uint64_t *d_send,
                                int64_t capacity, int64_t *d_counts, char *errbuf, size_t errlen);
int     smg_engine_apply(smg_engine *e, const uint64_t *d_recv, int64_t nrecv, int64_t *missing,
                         char *errbuf, size_t errlen);
int     smg_engine_apply_own(smg_engine *e, int64_t *missing, char *errbuf, size_t errlen);
/* (smg_engine_apply with missing == NULL does not wait for the device: the count stays there.)
   proof: d_dst[0..3] (device) = { complements that were missing or carried another count, fingerprint residue words 0, 1,
   1 if a replayed step (below) did not find the counts it was queued with, else 0 } of this shard, written in stream order -- a sharded run appends them to the histogram buffer of its final all_reduce
   without a host round trip (the residues of the ranks combine by XOR: give each rank its own two words).            */
int     smg_engine_proof(smg_engine *e, uint64_t *d_dst, char *errbuf, size_t errlen);
/* ... or the whole proof tail of a sharded step's reduction buffer in one launch: d_tail[0] = missing, d_tail[1 + 2 slot],
   d_tail[2 + 2 slot] = this shard's residue words (the other ranks' slots are zeroed: a SUM over the ranks then hands every
   rank all residues), d_tail[1 + 2 nslots] = a replayed step found other counts, d_tail[2 + 2 nslots] = 1 if this step was a
   replayed one: 3 + 2 nslots words.                                                                                      */
int     smg_engine_proof_tail(smg_engine *e, uint64_t *d_tail, int nslots, int slot, char *errbuf, size_t errlen);
/* Replay of the phase calls (round 5) -- the ONE mechanism that queues a step without reading anything back in between
   (smg_engine_run's own variant of it, "run_speculative", measured no gain in two rounds and was removed in round 6).  With
   set_replay(e, 1) a step pass1 -> [presort] -> filter -> route_device -> apply(missing = NULL) -> pass2 -> proof on a table
   whose PREVIOUS step went the same way (hash proof, k <= 64, look-up chain) is queued without a single read-back: the
   counts the host needs between the calls (requests emitted, deferred entries, requests kept) are last step's -- functions of
   the table and of the exchanged maps -- and the device compares them with this step's (and the per-destination totals of
   route_device with the recorded ones: the caller splits its exchange by those); a difference, an overflow or an order
   violation is reported through smg_engine_proof as d_dst[3] != 0 (route_device never writes past the capacity it was given: a
   replayed step whose list outgrew the record drops the surplus and fails this check).  A group of ranks must replay a step
   TOGETHER or not at all (sharded.py votes in the step's final all_reduce).  The caller reads the proof words
   (its one host wait of the step), tells the engine with replay_done(e, ok) -- ok = 0 drops the record -- and on a failure
   runs the step again, which then takes the plain path.  replay_state: bit 0 = the current step is a replayed one, bit 1 =
   a record exists.  Binding or conditioning a table drops the record.  No counterpart in the reference.                 */
int     smg_engine_set_replay(smg_engine *e, int on);
int     smg_engine_replay_state(smg_engine *e);
int     smg_engine_replay_done(smg_engine *e, int ok, char *errbuf, size_t errlen);
/* Request filter (hash proof, k <= 85).  A request only matters when its target is a candidate of
   pass 2 (exactly one suffix-side pair).  Pass 1 records in a bit map which block ids -- the leading
   id_bits = min(30, 2*(k/2)) bits of a k-mer -- hold a candidate; smg_engine_filter drops every request
   whose target id has a clear bit, before the requests are routed, sorted or looked up.  Single shard:
   apply_own filters with the engine's own map.  Sharded: every rank copies out the words its k-mer
   range covers (blockmap_copy), the ranks exchange them, OR them into one map of `nwords` uint32 on
   the device and pass that to filter, then route().  id_bits = 0: no map (exact proof or k > 85).
   One-word k-mers of >= 24 bases use a TWO-BIT map: 64 bits per 32 block ids (the id bit in the low half, a
   second bit hashed from the k-mer bits below the id in the high half; a request must find both), so
   nwords = 2 * 2^id_bits / 32 and the words of ids [a, b] are [2 * (a >> 5), 2 * (b >> 5) + 2).
   No counterpart in the reference (it has no complement look-ups at all: PloidyPlot.c scans every
   position of every k-mer). */
/* id bits of the block map that the NEXT smg_engine_pass1 builds (8..32; 0 = the default: 30, what a sharded run
   exchanges -- 128 MB, or 256 MB as a two-bit map).  A caller that will not exchange maps (one rank) asks for 32:
   several times fewer requests survive the filter, at the price of a 0.5 - 1 GB map that only this GPU ever reads. */
int     smg_engine_set_blockmap_bits(smg_engine *e, int id_bits);
int     smg_engine_blockmap(smg_engine *e, int *id_bits, int64_t *nwords);
int     smg_engine_blockmap_copy(smg_engine *e, int64_t word_lo, int64_t nw, uint32_t *d_dst,
                                 char *errbuf, size_t errlen);
/* the gathered word ranges of `nranks` shards -> one map: d_parts holds nranks slices of `width` words, slice r
   covers the map words [word_lo[r], word_lo[r] + nwords_of[r]) (neighbours share their boundary word: OR, not
   copy); d_full (the whole map, smg_engine_blockmap's nwords) is overwritten.  One kernel launch.           */
int     smg_engine_merge_maps(smg_engine *e, const uint32_t *d_parts, int64_t width, int nranks,
                              const int64_t *word_lo, const int64_t *nwords_of, uint32_t *d_full,
                              char *errbuf, size_t errlen);
int     smg_engine_filter(smg_engine *e, const uint32_t *d_map, int64_t *kept, char *errbuf, size_t errlen);
/* optional, between pass1 and filter: start the part of the filter that does not need the map (long request
   lists are bucketed on their leading 8 bits so that the map probes stay in L2) -- a sharded run calls it
   while the block maps are in flight.  A no-op when there is nothing to do. */
int     smg_engine_presort(smg_engine *e, char *errbuf, size_t errlen);
int     smg_engine_symhash(smg_engine *e, uint64_t out[4], char *errbuf, size_t errlen);
/* (smg_stats.npairs is left 0 by this entry: the caller sums the plot it reduces over the ranks anyway) */
int     smg_engine_pass2(smg_engine *e, int64_t *d_plot, char *errbuf, size_t errlen);
int     smg_engine_stats(smg_engine *e, smg_stats *stats);

/* extract leg on an engine whose run has completed on the symmetric path: d_labels = device
   uint16[SMG_PLOT_CELLS]; d_out = device buffer of `capacity` records (may be NULL with capacity 0
   to count only); *nrec = records produced (compare with capacity).                            */
int     smg_engine_extract(smg_engine *e, const uint16_t *d_labels, uint64_t *d_out, int64_t capacity,
                           int64_t *nrec, char *errbuf, size_t errlen);

/* library / build identification, e.g. "smudgeplot_amd 0.1 gfx950" */
const char *smg_version(void);

#ifdef __cplusplus
}
#endif
#endif
