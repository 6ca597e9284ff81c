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
  off = distance + 1;                                     /* probes reach coordinates -distance .. maxc + distance */
  side = (int64_t) maxc + 2 * (int64_t) distance + 3;
  F = (int64_t *) calloc((size_t) (side * side), sizeof(int64_t));
  P = (int32_t *) calloc((size_t) (side * side), sizeof(int32_t));
  if (!F || !P) { free(F); free(P); return -1; }
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
