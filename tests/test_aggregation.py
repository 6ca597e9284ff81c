"""local aggregation of the .smu pixels (smudgeplot_amd/aggregation.py over csrc/smg_aggregate.c)
against the REFERENCE's Coverages.local_aggregation: tests/golden/aggregation.json holds the reference's inputs (rows in
the order it processed them) and labels, made by tests/golden/make_golden_aggregation.py, which imports the reference."""
import ctypes as C
import io
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from smudgeplot_amd import aggregation


def cases():
    with open(os.path.join(GOLDEN, "aggregation.json")) as f:
        return json.load(f)


CASES = cases()


@pytest.mark.parametrize("i", range(len(CASES)))
def test_labels_equal_the_reference(i):
    c = CASES[i]
    rows = np.array(c["rows"], dtype=np.int64)
    peak, npk = aggregation.local_aggregation(rows[:, 0], rows[:, 1], rows[:, 2], c["distance"], c["noise_filter"], c["mask_errors"])
    assert peak.tolist() == c["peaks"], (c["name"], c["distance"], c["noise_filter"], c["mask_errors"])
    assert npk == max(max(c["peaks"]), 0)


def test_golden_set_exercises_every_branch():
    assert len(CASES) >= 20
    labels = [set(c["peaks"]) for c in CASES]
    assert any(-1 in s for s in labels) and any(0 in s for s in labels)          # error line, rows below the noise filter
    assert any(c["mask_errors"] is False for c in CASES) and any(c["distance"] == 0 for c in CASES)
    # a pixel that JOINED the error line (label -1 although it lies above it): the reference treats -1 as assigned
    joined = 0
    for c in CASES:
        if not c["mask_errors"]:
            continue
        L = min(r[0] for r in c["rows"])
        joined += sum(1 for r, p in zip(c["rows"], c["peaks"]) if p == -1 and r[0] >= L + c["distance"])
    assert joined > 0


def test_class_mirror_and_file_round_trip(tmp_path):
    c = next(c for c in CASES if c["name"] == "tetraploid_cov18" and c["distance"] == 5 and c["noise_filter"] == 50)
    rows = np.array(c["rows"], dtype=np.int64)
    cov = aggregation.Coverages((rows[:, 0], rows[:, 1], rows[:, 2]))
    cov.local_aggregation(distance=5, noise_filter=50, mask_errors=True)
    assert [cov.cov2peak[(a, b)] for b, a, _ in c["rows"]] == c["peaks"]
    out = io.StringIO()
    cov.write_peaks(out)
    lines = out.getvalue().splitlines()
    assert len(lines) == len(rows)
    keys = [(int(l.split("\t")[1]), int(l.split("\t")[0])) for l in lines]
    assert keys == sorted(keys)                                                  # (covA, covB) ascending, smudgeplot.py:75
    cov.count_kmers()
    f = rows[:, 2]; p = np.array(c["peaks"])
    assert cov.total_kmers == f.sum() and cov.total_error_kmers == f[p == -1].sum()
    assert cov.total_genomic_kmers_in_smudges == f[p > 0].sum()
    # load_hetmers: .smu text -> columns by freq descending
    smu = tmp_path / "x.smu"
    smu.write_text("".join(f"{b}\t{a}\t{fr}\n" for b, a, fr in sorted(c["rows"], key=lambda r: (r[0] + r[1], r[0]))))
    b2, a2, f2 = aggregation.load_hetmers(str(smu))
    assert len(f2) == len(rows) and np.all(np.diff(f2) <= 0) and sorted(zip(b2, a2, f2)) == sorted(map(tuple, c["rows"]))


def test_edge_cases():
    e = np.zeros(0, dtype=np.int64)
    peak, npk = aggregation.local_aggregation(e, e, e, 5, 1, True)
    assert len(peak) == 0 and npk == 0
    peak, npk = aggregation.local_aggregation([7], [30], [12], 5, 1, False)
    assert peak.tolist() == [1] and npk == 1
    peak, npk = aggregation.local_aggregation([7], [30], [12], 5, 1, True)       # the only row IS the error line
    assert peak.tolist() == [-1] and npk == 0
    peak, npk = aggregation.local_aggregation([7], [30], [12], 5, 100, True)     # below the noise filter
    assert peak.tolist() == [0] and npk == 0
    with pytest.raises(RuntimeError):
        aggregation.local_aggregation([-1], [30], [12], 5, 1, True)
    with pytest.raises(ValueError):
        aggregation.local_aggregation([1, 2], [30], [12], 5, 1, True)
    # coordinates at the corner of the plot, a radius larger than the plot
    peak, _ = aggregation.local_aggregation([499, 0], [501, 1000], [9, 8], 2000, 1, False)
    assert peak.tolist() == [1, 1]


def test_library_exports_what_the_header_declares():
    hdr = open(os.path.join(ROOT, "include", "smg_aggregate.h")).read()
    names = re.findall(r"\b(smg_[a-z_0-9]+)\s*\(", hdr)
    assert names == ["smg_local_aggregation", "smg_fishnet_centralities"]
    lib = C.CDLL(os.path.join(ROOT, "smudgeplot_amd", "libsmg_aggregate.so"))
    for n in names:
        assert hasattr(lib, n)


def test_speed_against_python_dictionaries():
    """not a benchmark, a guard: the largest golden case (7668 rows) in well under the reference's time
    (0.35 s in this container for distance 8)"""
    import time
    c = max(CASES, key=lambda c: len(c["rows"]) * (c["distance"] + 1) ** 2)
    rows = np.array(c["rows"], dtype=np.int64)
    t0 = time.perf_counter()
    for _ in range(20):
        aggregation.local_aggregation(rows[:, 0], rows[:, 1], rows[:, 2], c["distance"], c["noise_filter"], c["mask_errors"])
    dt = (time.perf_counter() - t0) / 20
    assert dt < 0.05, dt


# ---- the coverage grid search (second half of the row) ---------------------------------------------------------------

def centrality_cases():
    with open(os.path.join(GOLDEN, "centrality.json")) as f:
        return json.load(f)


CCASES = centrality_cases()


@pytest.mark.parametrize("i", range(len(CCASES)))
def test_grid_search_equals_the_reference(i):
    """every tested coverage, its centrality (the same double, bit for bit) and the winner of Smudges.get_centrality_df"""
    c = CCASES[i]
    rows = np.array(c["rows"], dtype=np.int64)
    s = aggregation.Smudges((rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3]), c["total_genomic_kmers"])
    s.get_centrality_df(c["min_c"], c["max_c"], c["cutoff"])
    assert s.centrality_df["coverage"].tolist() == c["coverage"]
    got, want = s.centrality_df["centrality"], np.array(c["centrality"])
    assert got.tobytes() == want.tobytes(), np.abs(got - want).max()
    assert float(s.cov) == c["best"]


def test_centrality_cells_by_hand():
    """two pixels, cov = 10: (covB 10, covA 10) is cell AB at its centre, (covB 10, covA 21) is cell AAB one off"""
    b = np.array([10, 10]); a = np.array([10, 21]); f = np.array([300, 100]); sm = np.array([1, 2])
    out = aggregation.fishnet_centralities(b, a, f, sm, 400, np.array([10.0]))
    assert out[0] == (0.0 * 300 + 0.1 * 100) / 400
    # error-line pixels do not count; nothing left: 1.0
    out = aggregation.fishnet_centralities(b, a, f, np.array([-1, -1]), 400, np.array([10.0, 7.5]))
    assert out.tolist() == [1.0, 1.0]
    # a pixel exactly on a cell border (covB = 15 = 10 * 1.5) belongs to no cell
    out = aggregation.fishnet_centralities([15], [15], [5], [1], 5, np.array([10.0]))
    assert out[0] == 1.0
    # the size cut-off is a strict "greater than"
    out = aggregation.fishnet_centralities(b, a, f, sm, 400, np.array([10.0]), smudge_filter=0.25)
    assert out[0] == 0.0                                    # only the 300-pair cell (0.75 > 0.25; 0.25 > 0.25 is false)


def test_coverage_table_with_ties_goes_through_load_hetmers(tmp_path):
    """rows of equal freq keep the order of the file (documented: the reference leaves it to pandas' unstable sort);
    the labels then are those of local_aggregation on exactly that order"""
    rows = [(5, 30, 700), (6, 30, 700), (5, 31, 700), (20, 20, 900), (21, 20, 900), (2, 40, 100), (7, 33, 700)]
    smu = tmp_path / "t.smu"
    smu.write_text("".join(f"{b}\t{a}\t{f}\n" for b, a, f in rows))
    b, a, f = aggregation.load_hetmers(str(smu))
    assert f.tolist() == [900, 900, 700, 700, 700, 700, 100]
    assert list(zip(b.tolist(), a.tolist()))[:2] == [(20, 20), (21, 20)]           # ties: file order
    assert list(zip(b.tolist(), a.tolist()))[2:6] == [(5, 30), (6, 30), (5, 31), (7, 33)]
    cov = aggregation.Coverages((b, a, f))
    cov.local_aggregation(distance=2, noise_filter=200, mask_errors=False)
    want, _ = aggregation.local_aggregation(b, a, f, 2, 200, False)
    assert cov.smudge.tolist() == want.tolist() and want[-1] == 0 and want[0] == want[1] == 1


def test_coverages_that_are_not_a_hetmers_table_are_refused_with_a_reason():
    with pytest.raises(RuntimeError, match="65535"):
        aggregation.local_aggregation([1], [70000], [5], 2, 1, False)


@pytest.mark.gpu
def test_aggregation_goldens_on_the_gpu_box():
    """the same goldens once more under the `gpu` marker, so that the driver's GPU-side run (which loads the in-tree
    libraries on that box) shows this row too; nothing here needs the GPU"""
    for c in CASES:
        rows = np.array(c["rows"], dtype=np.int64)
        peak, _ = aggregation.local_aggregation(rows[:, 0], rows[:, 1], rows[:, 2], c["distance"], c["noise_filter"], c["mask_errors"])
        assert peak.tolist() == c["peaks"], c["name"]
    for c in CCASES:
        rows = np.array(c["rows"], dtype=np.int64)
        s = aggregation.Smudges((rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3]), c["total_genomic_kmers"])
        s.get_centrality_df(c["min_c"], c["max_c"], c["cutoff"])
        assert s.centrality_df["centrality"].tobytes() == np.array(c["centrality"]).tobytes() and float(s.cov) == c["best"]
