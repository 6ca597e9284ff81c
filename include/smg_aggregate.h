/*******************************************************************************************
 *
 *  smg_aggregate.h -- C ABI of the step AFTER the hot path:
 *                     the greedy local aggregation of the .smu pixels into smudges.
 *
 *  Host code only (plain C, no GPU): the algorithm is a sequential greedy walk over at most
 *  ~2.5e5 rows -- every decision depends on all earlier ones -- so there is nothing for a
 *  GPU to do; the reference spends seconds in Python dictionaries per table, this spends
 *  milliseconds, which matters for batch runs over hundreds of genomes (tests/README.md:38-63
 *  of the reference).
 *
 *  Replaces: Coverages.local_aggregation, src/smudgeplot/smudgeplot.py:29-69
 *            (called from cli.py:407,411 on load_hetmers' table: rows sorted by freq, descending).
 *
 ********************************************************************************************/
#ifndef SMG_AGGREGATE_H
#define SMG_AGGREGATE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* rows i = 0 .. n-1 in the caller's order (the reference: freq descending; the order of rows with equal
   freq is the caller's business -- it decides the result exactly as it does in the reference):
     covB[i] <= covA[i]   the pixel (minor, major coverage), freq[i] its pair count
   distance      Manhattan radius of the neighbourhood         (cli.py -d)
   noise_filter  the walk stops at the first row with freq < noise_filter (that row and all later ones get 0)
   mask_errors   != 0: rows with covB < min(covB) + distance are the error line, label -1
   peak[i]       out: the smudge label of row i (1 .. *npeaks in order of creation, -1, or 0 = not reached)
   returns 0, or -1 on bad arguments / out of memory                                                    */
int smg_local_aggregation(const int32_t *covB, const int32_t *covA, const int64_t *freq, int64_t n,
                          int32_t distance, int64_t noise_filter, int32_t mask_errors,
                          int32_t *peak, int32_t *npeaks);

#ifdef __cplusplus
}
#endif
#endif
