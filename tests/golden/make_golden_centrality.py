#!/usr/bin/env python3
"""Regenerate tests/golden/centrality.json: the REFERENCE's coverage grid search (Smudges.get_centrality_df,
/root/reference/src/smudgeplot/smudgeplot.py:105-148, imported here) on synthetic smudge landscapes: input table in the
order the reference hands it over (after local_aggregation + count_kmers: sorted by covA, covB, with the smudge
column), parameters, and its outputs -- every tested coverage with its centrality (float64, exact) and the winner.

    python tests/golden/make_golden_centrality.py
"""
import contextlib
import io
import json
import os
import sys

sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from smudgeplot import smudgeplot as ref                      # noqa: E402
from make_golden_aggregation import synthetic_smu            # noqa: E402  (the same landscapes)

OUT = os.path.dirname(os.path.abspath(__file__))


def fmean311(data, weights=None):
    """statistics.fmean of Python 3.11 (the reference calls fmean(..., weights=...), which the 3.10 of this container
    does not have): every product rounded to double, numerator and denominator summed exactly (math.fsum).
    (3.12+ uses math.sumprod for the numerator: unrounded products -- its result can differ in the last bit.)"""
    from math import fsum
    from operator import mul
    assert weights is not None and len(data) == len(weights)
    return fsum(map(mul, data, weights)) / fsum(weights)


if sys.version_info < (3, 11):
    ref.fmean = fmean311

if __name__ == "__main__":
    cases = []
    specs = [("diploid_cov30", 1, 30, [((1, 1), 0.8), ((2, 2), 0.1), ((2, 1), 0.05)], 400000, 0.05, 10, 60, 0),
             ("tetraploid_cov18", 2, 18, [((3, 1), 0.45), ((2, 2), 0.3), ((1, 1), 0.1), ((4, 2), 0.05)], 600000, 0.1, 10, 60, 0),
             ("triploid_cov55", 3, 55, [((2, 1), 0.7), ((1, 1), 0.15), ((4, 2), 0.05)], 300000, 0.02, 10, 60, 0),
             ("hexaploid_cov12", 4, 12, [((5, 1), 0.3), ((4, 2), 0.25), ((3, 3), 0.2), ((2, 1), 0.1), ((1, 1), 0.05)], 500000, 0.1, 5, 40, 0.01),
             ("diploid_cov30_cutoff", 1, 30, [((1, 1), 0.8), ((2, 2), 0.1), ((2, 1), 0.05)], 400000, 0.05, 20, 45, 0.02)]
    for name, seed, cov, pairs, npairs, err, min_c, max_c, cutoff in specs:
        smu = synthetic_smu(seed, cov, pairs, npairs, err)
        c = ref.Coverages(ref.load_hetmers(io.StringIO(smu)))
        c.local_aggregation(distance=5, noise_filter=50, mask_errors=True)
        c.count_kmers()
        s = ref.Smudges(c.cov_tab, c.total_genomic_kmers)
        with contextlib.redirect_stderr(io.StringIO()):
            s.get_centrality_df(min_c, max_c, cutoff)
        rows = [(int(b), int(a), int(f), int(p)) for _, b, a, f, p in c.cov_tab.itertuples()]
        cases.append(dict(name=name, min_c=min_c, max_c=max_c, cutoff=cutoff, total_genomic_kmers=int(c.total_genomic_kmers),
                          rows=rows, coverage=[float(x) for x in s.centrality_df["coverage"]],
                          centrality=[float(x) for x in s.centrality_df["centrality"]], best=float(s.cov)))
        print(name, len(rows), "rows, true cov", cov, "-> best", s.cov, len(s.centrality_df), "candidates")
    with open(os.path.join(OUT, "centrality.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    print(os.path.getsize(os.path.join(OUT, "centrality.json")), "bytes")
