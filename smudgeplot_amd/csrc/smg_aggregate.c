/*******************************************************************************************
 *
 *  smg_aggregate.c -- greedy local aggregation of .smu pixels (include/smg_aggregate.h).
 *
 *  Restates Coverages.local_aggregation, /root/reference/src/smudgeplot/smudgeplot.py:29-69, on two dense
 *  grids (frequency and label per pixel) instead of two dictionaries.  Two properties of the reference
 *  that a cleaner formulation would lose, kept on purpose because they decide labels:
 *    - the neighbourhood walk orders a probed coordinate pair as (larger, smaller) by REASSIGNING the
 *      outer loop variable, so the larger value carries over to the remaining probes of the same outer
 *      step (smudgeplot.py:52-58);
 *    - a pixel of the error line (label -1) counts as an assigned neighbour, so a pixel next to it can
 *      join the error line (smudgeplot.py:58, "if cov2peak[...]" is true for -1).
 *
 ********************************************************************************************/
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "smg_aggregate.h"

int smg_local_aggregation(const int32_t *covB, const int32_t *covA, const int64_t *freq, int64_t n,
                          int32_t distance, int64_t noise_filter, int32_t mask_errors,
                          int32_t *peak, int32_t *npeaks)
{ int64_t i, side;
  int32_t maxc = 0, minB = 0, next_peak = 1, off;
  int64_t *F;
  int32_t *P;

  if (n < 0 || distance < 0 || (n > 0 && (!covB || !covA || !freq || !peak))) return -1;
  if (npeaks) *npeaks = 0;
  if (n == 0) return 0;
  for (i = 0; i < n; i++)
    { if (covA[i] < 0 || covB[i] < 0) return -1;
      if (covA[i] > maxc) maxc = covA[i];
      if (covB[i] > maxc) maxc = covB[i];
      if (i == 0 || covB[i] < minB) minB = covB[i];      /* L = min(covB), smudgeplot.py:36 */
      peak[i] = 0;
    }
  /* hetmers writes coverages <= 1000 (a 13 MB pair of grids); a .smu of 16-bit coverages from another tool still works
     (up to 51 GB of grids -- the calloc may fail, reported as such), anything beyond that is not a coverage table */
  if (maxc > 65535 || distance > 65535) return -2;
  off = distance + 1;                                     /* probes reach coordinates -distance .. maxc + distance */
  side = (int64_t) maxc + 2 * (int64_t) distance + 3;
  F = (int64_t *) calloc((size_t) (side * side), sizeof(int64_t));
  P = (int32_t *) calloc((size_t) (side * side), sizeof(int32_t));
  if (!F || !P) { free(F); free(P); return -3; }
#define AT(a, b) (((int64_t) (a) + off) * side + ((int64_t) (b) + off))

  for (i = 0; i < n; i++)
    { const int32_t a = covA[i], b = covB[i];
      int64_t best_f = 0, best_at = -1;
      int32_t xa0;
      F[AT(a, b)] = freq[i];                              /* the frequency grid is filled on the fly, :42 */
      if (freq[i] < noise_filter) break;                  /* :45 */
      if (mask_errors && b < minB + distance)             /* :47-49 */
        { P[AT(a, b)] = -1; peak[i] = -1; continue; }
      for (xa0 = a - distance; xa0 <= a + distance; xa0++)
        { int32_t xA = xa0, xb0;                           /* xA: the loop variable the reference reassigns */
          const int32_t dA = a > xa0 ? a - xa0 : xa0 - a, dB = distance - dA;
          for (xb0 = b - dB; xb0 <= b + dB; xb0++)
            { const int32_t lo = xA < xb0 ? xA : xb0, hi = xA < xb0 ? xb0 : xA;
              int64_t at;
              xA = hi;                                     /* "xB, xA = sorted([xA, xB])", :55 */
              at = AT(hi, lo);
              if (P[at] != 0 && F[at] > best_f) { best_f = F[at]; best_at = at; }
            }
        }
      if (best_f > 0) P[AT(a, b)] = P[best_at];            /* joins its strongest assigned neighbour, :61-62 */
      else            P[AT(a, b)] = next_peak++;           /* or founds a smudge, :63-67 */
      peak[i] = P[AT(a, b)];
    }
#undef AT
  if (npeaks) *npeaks = next_peak - 1;
  free(F); free(P);
  return 0;
}

/* ---- fishnet centrality ------------------------------------------------------------------------------------ */

/* exactly rounded sum of doubles (Shewchuk's algorithm, the one behind Python's math.fsum, which statistics.fmean
   uses for numerator and denominator): partials hold a non-overlapping expansion of the running sum */
typedef struct { double p[40]; int n; } fsum_t;

static void fsum_add(fsum_t *f, double x)
{ int i, j = 0;
  for (i = 0; i < f->n; i++)
    { double y = f->p[i], hi, lo;
      if (fabs(x) < fabs(y)) { double t = x; x = y; y = t; }
      hi = x + y;
      lo = y - (hi - x);
      if (lo != 0.0) f->p[j++] = lo;
      x = hi;
    }
  f->p[j++] = x;
  f->n = j;
}

static double fsum_result(const fsum_t *f)
{ /* msum's final step: sum the partials from the top, then fix the round-half-even case */
  int n = f->n;
  double hi = 0.0, lo = 0.0;
  if (n == 0) return 0.0;
  hi = f->p[--n];
  while (n > 0)
    { const double x = hi, y = f->p[--n];
      hi = x + y;
      lo = y - (hi - x);
      if (lo != 0.0) break;
    }
  if (n > 0 && ((lo < 0.0 && f->p[n - 1] < 0.0) || (lo > 0.0 && f->p[n - 1] > 0.0)))
    { const double y = lo * 2.0, x = hi + y, yr = x - hi;
      if (y == yr) hi = x;
    }
  return hi;
}

/* the cell index X (1-based) of coverage c on the grid of candidate cov: cov*(X-0.5) < c < cov*(X+0.5), X = 1: 0 < c;
   0 = on a cell border, below the grid or above cell `maxX` */
static int cell_of(int32_t c, double cov, int maxX)
{ int X = (int) ((double) c / cov + 0.5), d;
  for (d = -1; d <= 1; d++)                                 /* (the rounding of the quotient may be off by one: test) */
    { const int x = X + d;
      if (x >= 1 && x <= maxX)
        { const double lo = x == 1 ? 0.0 : cov * ((double) x - 0.5), hi = cov * ((double) x + 0.5);
          if ((double) c > lo && (double) c < hi) return x;
        }
    }
  return 0;
}

int smg_fishnet_centralities(const int32_t *covB, const int32_t *covA, const int64_t *freq, const int32_t *smudge,
                             int64_t n, int64_t total_genomic_kmers, double smudge_filter,
                             const double *cov, int64_t ncov, double *centrality)
{ int64_t q, i;
  if (n < 0 || ncov < 0 || (n > 0 && (!covB || !covA || !freq || !smudge)) || (ncov > 0 && (!cov || !centrality))) return -1;
  for (q = 0; q < ncov; q++)
    { /* cells [Bs 1..8][As 1..15] */
      int64_t sum[9][16], best_f[9][16];
      int32_t cA[9][16], cB[9][16];
      int Bs, As, any = 0;
      fsum_t num, den;
      const double c = cov[q];
      memset(sum, 0, sizeof(sum)); memset(cA, 0, sizeof(cA)); memset(cB, 0, sizeof(cB));
      for (Bs = 0; Bs < 9; Bs++) for (As = 0; As < 16; As++) best_f[Bs][As] = -1;
      if (!(c > 0.0)) { centrality[q] = 1.0; continue; }
      for (i = 0; i < n; i++)
        { int b, a;
          if (smudge[i] == -1) continue;
          b = cell_of(covB[i], c, 8);
          if (!b) continue;
          a = cell_of(covA[i], c, 16 - b);                                  /* As in Bs .. 16-Bs: range(Bs, 17-Bs), :163 */
          if (a < b) continue;
          sum[b][a] += freq[i];
          if (freq[i] > best_f[b][a]) { best_f[b][a] = freq[i]; cA[b][a] = covA[i]; cB[b][a] = covB[i]; }    /* idxmax: first maximum */
        }
      num.n = den.n = 0;
      for (Bs = 1; Bs <= 8; Bs++)
        for (As = Bs; As < 17 - Bs; As++)
          { if (best_f[Bs][As] < 0) continue;                                   /* (an empty cell: 0 / total > filter is false for filter >= 0;
                                                                                   for a negative filter the reference's idxmax would raise) */
            if ((double) sum[Bs][As] / (double) total_genomic_kmers > smudge_filter)
              { const double dA = fabs(((double) cA[Bs][As] - c * (double) As) / c);
                const double dB = fabs(((double) cB[Bs][As] - c * (double) Bs) / c);
                const double cen = dA + dB;
                fsum_add(&num, cen * (double) sum[Bs][As]);
                fsum_add(&den, (double) sum[Bs][As]);
                any = 1;
              }
          }
      centrality[q] = any ? fsum_result(&num) / fsum_result(&den) : 1.0;
    }
  return 0;
}
