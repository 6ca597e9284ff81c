#!/usr/bin/env python3
"""Regenerate tests/golden/aggregation.json: inputs and outputs of the REFERENCE Coverages.local_aggregation
(/root/reference/src/smudgeplot/smudgeplot.py:29-69, imported here -- it is Python) on .smu tables: the golden
`.smu` files of this directory and synthetic smudge landscapes (several coverage peaks + an error line + noise).
The rows are stored in the order the reference processed them (its load_hetmers sorts by freq with an unstable sort:
the order of ties is part of the input).  Run in the build container:

    python tests/golden/make_golden_aggregation.py
"""
import io
import json
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/src")
from smudgeplot import smudgeplot as ref          # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def synthetic_smu(seed, cov, ploidy_pairs, npairs, err):
    """pixels (covB, covA, freq) of a made-up landscape: Poisson clouds around (B*cov, A*cov) + low-coverage errors"""
    rng = np.random.default_rng(seed)
    plot = {}
    for (A, B), w in ploidy_pairs:
        m = int(npairs * w)
        a = rng.poisson(A * cov, m); b = rng.poisson(B * cov, m)
        for x, y in zip(np.maximum(a, b), np.minimum(a, b)):
            if y >= 3 and x + y <= 1000 and y < 500:
                plot[(int(y), int(x))] = plot.get((int(y), int(x)), 0) + 1
    m = int(npairs * err)
    a = rng.poisson(cov, m); b = 3 + rng.geometric(0.5, m)
    for x, y in zip(np.maximum(a, b), np.minimum(a, b)):
        plot[(int(y), int(x))] = plot.get((int(y), int(x)), 0) + 1
    rows = sorted((a + b, b, a, f) for (b, a), f in plot.items())           # .smu order: sum, then min
    return "".join(f"{b}\t{a}\t{f}\n" for _, b, a, f in rows)


if __name__ == "__main__":
    cases = []
    inputs = []
    for name in ("k31_i1", "k21_i2_p2", "k51_i1_p3"):
        inputs.append((name, open(os.path.join(OUT, name + ".smu")).read()))
    inputs.append(("diploid_cov30", synthetic_smu(1, 30, [((1, 1), 0.8), ((2, 2), 0.1), ((2, 1), 0.05)], 400000, 0.05)))
    inputs.append(("tetraploid_cov18", synthetic_smu(2, 18, [((3, 1), 0.45), ((2, 2), 0.3), ((1, 1), 0.1), ((4, 2), 0.05)], 600000, 0.1)))
    inputs.append(("triploid_cov55", synthetic_smu(3, 55, [((2, 1), 0.7), ((1, 1), 0.15), ((4, 2), 0.05)], 300000, 0.02)))

    for name, smu in inputs:
        for distance, noise_filter, mask in ((5, 50, True), (5, 1000, True), (3, 10, False), (1, 1, True), (8, 20, False), (0, 1, True)):
            tab = ref.load_hetmers(io.StringIO(smu))
            if len(tab) == 0:
                continue
            cov = ref.Coverages(tab)
            cov.local_aggregation(distance=distance, noise_filter=noise_filter, mask_errors=mask)
            rows = [(int(b), int(a), int(f)) for _, b, a, f in tab.itertuples()]
            peaks = [int(cov.cov2peak[(a, b)]) for b, a, f in rows]
            if len(set(peaks)) < 2 and noise_filter > 1:
                continue                                           # (everything below the noise filter: not a test)
            cases.append(dict(name=name, distance=distance, noise_filter=noise_filter, mask_errors=mask,
                              rows=rows, peaks=peaks))
            print(name, distance, noise_filter, mask, len(rows), "rows", max(peaks), "smudges", peaks.count(-1), "error pixels")

    with open(os.path.join(OUT, "aggregation.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    print(len(cases), "cases,", os.path.getsize(os.path.join(OUT, "aggregation.json")), "bytes")
