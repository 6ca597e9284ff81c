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
 *            (called from cli.py:407,411 on load_hetmers' table: rows sorted by freq, descending),
 *            and the cell statistics of the 1n-coverage grid search (get_smudge_container + get_centrality).
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
   returns 0, -1 on bad arguments, -2 when a coverage exceeds 65535 (two dense grids of (max coverage + 2 distance + 3)^2
   cells are allocated: 13 MB for hetmers output, whose coverages end at 1000), -3 when out of memory        */
int smg_local_aggregation(const int32_t *covB, const int32_t *covA, const int64_t *freq, int64_t n,
                          int32_t distance, int64_t noise_filter, int32_t mask_errors,
                          int32_t *peak, int32_t *npeaks);

/* Fishnet centrality of a 1n-coverage candidate (the inner loop of the reference's coverage grid search).
   Replaces: Smudges.get_smudge_container(cov, smudge_filter, "fishnet") + get_centrality(container, cov),
             src/smudgeplot/smudgeplot.py:150-176, 307-352, 799-802, as called from get_best_coverage (:137-148).
   rows i = 0 .. n-1 in the caller's order (the reference: sorted by covA, covB after peak_aggregation, :75); rows with
   smudge[i] == -1 (error line) are ignored.  For every candidate cov the pixels are cut into the cells
   (As, Bs), Bs = 1..8, As = Bs..16-Bs, by the OPEN intervals cov*(X-0.5) < c < cov*(X+0.5) (X = 1: 0 < c); a cell
   counts if its pairs / total_genomic_kmers > smudge_filter; its centre is its first row of maximal freq; the
   result is the freq-weighted mean of |cA - cov*As| / cov + |cB - cov*Bs| / cov over the cells in (Bs, As) order,
   computed with exactly rounded sums like Python's statistics.fmean -- so that the doubles, and with them the
   argmin over a coverage grid, are the reference's bit for bit.  No cell passes: 1.0.
   returns 0, or -1 on bad arguments / out of memory                                                      */
int smg_fishnet_centralities(const int32_t *covB, const int32_t *covA, const int64_t *freq, const int32_t *smudge,
                             int64_t n, int64_t total_genomic_kmers, double smudge_filter,
                             const double *cov, int64_t ncov, double *centrality);

#ifdef __cplusplus
}
#endif
#endif
