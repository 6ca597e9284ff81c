/* smg_ktab.h -- host-side (plain C) loader for FastK k-mer tables, "format F".
 *
 * Stands in for the Kmer_Stream part of the reference's libfastk
 * (/root/reference/src/lib/libfastk.c:717-1409).  Two ways to use it:
 *   smg_ktab_open     stub, prefix index and part headers only; the records stay on disk and are fetched with
 *                     smg_ktab_read (pread, thread safe) -- what `hetmers` does: the engine pulls the parts
 *                     through a small pinned ring straight into HBM, the host never holds the table
 *                     (the reference streams 1024-record blocks, libfastk.c:759-784);
 *   smg_ktab_load_mt  additionally reads every part into host memory (parts kept as they are on disk,
 *                     headers stripped), for callers that hand a `smg_table_view` to the engine.
 */
#ifndef SMG_KTAB_H
#define SMG_KTAB_H

#include <stddef.h>
#include <stdint.h>

typedef struct smg_ktab
{ int       kmer, nparts, minval, ibyte;
  int       kbyte, tbyte, hbyte, pbyte;     /* as in libfastk.c:823-827                        */
  int64_t   nels;
  int64_t   ixlen;
  int64_t  *index;                          /* [ixlen] cumulative end offsets                  */
  uint8_t **part;                           /* [nparts] raw records                            */
  int64_t  *part_nels;                      /* [nparts]                                        */
  int64_t  *part_end;                       /* [nparts] cumulative (neps, libfastk.c:857)      */
  int      *fd;                             /* [nparts] open part files (-1 when closed)       */
  int       index_borrowed;                 /* `index` is the caller's memory (smg_ktab_set_index_memory): not freed here */
} smg_ktab;

#define SMG_KTAB_MAX_KMER   128   /* = SMG_MAX_KMER of the engine: larger k is refused at open time            */
#define SMG_KTAB_MAX_PBYTE  ((SMG_KTAB_MAX_KMER + 3) / 4 + 2)   /* widest record (ibyte >= 1 makes it one less)  */

#define SMG_KTAB_OK        0
#define SMG_KTAB_NOSTUB    1     /* stub cannot be opened     (Open_Kmer_Stream returns NULL)   */
#define SMG_KTAB_NOPART    2     /* "Table part %s is missing ?"            libfastk.c:850-853 */
#define SMG_KTAB_KMISMATCH 3     /* "... does not have k-mer length matching stub ?"  858-862  */
#define SMG_KTAB_NOMEM     4
#define SMG_KTAB_SHORT     5     /* file shorter than its header promises                       */

/* The prefix index (8 << 8 ibyte bytes: 134 MB at ibyte = 3) of the NEXT table that is opened goes into -- filled = 0 -- or is
   taken as it stands from -- filled = 1: nothing is read, nothing is checked again -- `buf` (room for cap_words words; a table
   whose index does not fit is opened the ordinary way).  `hetmers` puts it into a mapping shared by the process that opens and
   probes the table and the one that drives the GPU.  buf = NULL: back to malloc.  Not thread safe (set, open, unset). */
void smg_ktab_set_index_memory(int64_t *buf, int64_t cap_words, int filled);

/* name: "<path>[.ktab]".  On failure `what` (>= 4096 bytes) receives the offending file name. */
int  smg_ktab_open(const char *name, smg_ktab *t, char *what);      /* part[p] stay NULL: use smg_ktab_read */
/* `nent` records of part `part` starting at its entry `first` -> dst; 0 on success.  Thread safe.        */
int  smg_ktab_read(const smg_ktab *t, int part, int64_t first, int64_t nent, void *dst);
int  smg_ktab_load(const char *name, smg_ktab *t, char *what);
/* same, the part files read by `nthreads` threads (the -T of the command line, <= 64)           */
int  smg_ktab_load_mt(const char *name, smg_ktab *t, char *what, int nthreads);
void smg_ktab_free(smg_ktab *t);

/* expand entry i into kbyte packed bytes (Current_Entry, libfastk.c:1230-1269) + its count     */
void smg_ktab_entry(const smg_ktab *t, int64_t i, uint8_t *kmer_out, int *count_out);

/* index of the entry equal to the packed k-mer, or -1 (GoTo_Kmer_Entry, libfastk.c:1320-1409)  */
int64_t smg_ktab_find(const smg_ktab *t, const uint8_t *kmer);

/* the reference's conditioning probe, PloidyPlot.c:1167-1230.  Returns SMG_KTAB_OK, or SMG_KTAB_SHORT when the
   records of a table left on disk cannot be read (the decisions are then meaningless), SMG_KTAB_NOMEM          */
int smg_ktab_examine(const smg_ktab *t, int ethresh, int *trim, int *symm);
/* the same with the count scan spread over up to 16 threads (it reads 1e8 records: ~90 ms on one core) */
int smg_ktab_examine_mt(const smg_ktab *t, int ethresh, int nthreads, int *trim, int *symm);

#endif
