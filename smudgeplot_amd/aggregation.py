"""The step after the hot path: local aggregation of the `.smu` pixels into smudges.

`Coverages` mirrors the reference class of the same name (src/smudgeplot/smudgeplot.py:20-93): the same attributes and
method names with the same meaning, so `cli.py:402-411` reads the same with either --

    coverages = Coverages(load_hetmers(path))
    coverages.local_aggregation(distance=5, noise_filter=1000, mask_errors=True)
    coverages.write_peaks()

-- but the greedy walk runs in C (include/smg_aggregate.h, smudgeplot_amd/csrc/smg_aggregate.c) on dense grids
instead of Python dictionaries.  No pandas needed: the table is three integer columns.
"""
import ctypes as C
import os
import sys

import numpy as np

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("SMG_AGG_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsmg_aggregate.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: run `make -C smudgeplot_amd/csrc` (or __graft_entry__.build())")
        _LIB = C.CDLL(path)
        _LIB.smg_local_aggregation.restype = C.c_int
        _LIB.smg_local_aggregation.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64,
                                               C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]
    return _LIB


def load_hetmers(path):
    """(covB, covA, freq) int arrays of a .smu file, rows by freq descending (smudgeplot.py:789-791).  Rows of EQUAL freq
    keep the file's order here; the reference sorts with pandas' default (unstable) quicksort, which leaves their order
    to the library version -- and the greedy aggregation depends on the order of its rows, so pixels of equal freq at or
    above the noise filter can be labelled differently by the two; the golden tests feed the rows in
    the order the reference processed them."""
    tab = np.loadtxt(path, dtype=np.int64, delimiter="\t", ndmin=2)
    if tab.size == 0:
        tab = np.zeros((0, 3), dtype=np.int64)
    order = np.argsort(-tab[:, 2], kind="stable")
    tab = tab[order]
    return tab[:, 0].copy(), tab[:, 1].copy(), tab[:, 2].copy()


def local_aggregation(covB, covA, freq, distance, noise_filter, mask_errors):
    """labels per row (int32: 1.. = smudge, -1 = error line, 0 = below the noise filter) and the number of smudges"""
    covB = np.ascontiguousarray(covB, dtype=np.int32)
    covA = np.ascontiguousarray(covA, dtype=np.int32)
    freq = np.ascontiguousarray(freq, dtype=np.int64)
    n = len(freq)
    if not (len(covB) == len(covA) == n):
        raise ValueError("columns of different lengths")
    peak = np.zeros(n, dtype=np.int32)
    npk = C.c_int32(0)
    rc = _lib().smg_local_aggregation(covB.ctypes.data, covA.ctypes.data, freq.ctypes.data, n, int(distance),
                                      int(noise_filter), int(bool(mask_errors)), peak.ctypes.data, C.byref(npk))
    if rc != 0:
        raise RuntimeError({-2: "smg_local_aggregation: coverage above 65535 (not a hetmers table)",
                            -3: "smg_local_aggregation: out of memory"}.get(rc, "smg_local_aggregation failed (negative coverage or bad distance)"))
    return peak, int(npk.value)


class Coverages:
    """reference: smudgeplot.py:20-93.  cov_tab = (covB, covA, freq) arrays in processing order."""

    def __init__(self, cov_tab):
        self.covB, self.covA, self.freq = (np.asarray(c) for c in cov_tab)
        self.cov2peak = {}
        self.smudge = None
        self.total_kmers = None
        self.total_genomic_kmers = None
        self.total_genomic_kmers_in_smudges = None
        self.total_error_kmers = None
        self.error_fraction = None

    def local_aggregation(self, distance, noise_filter, mask_errors):
        self.smudge, self.npeaks = local_aggregation(self.covB, self.covA, self.freq, distance, noise_filter, mask_errors)
        self.cov2peak = {(int(a), int(b)): int(p) for a, b, p in zip(self.covA, self.covB, self.smudge)}

    def peak_aggregation(self):
        """rows sorted by (covA, covB) with their labels, like the reference's cov_tab after peak_aggregation"""
        order = np.lexsort((self.covB, self.covA))
        return self.covB[order], self.covA[order], self.freq[order], self.smudge[order]

    def write_peaks(self, out=None):
        out = out or sys.stdout
        for b, a, f, p in zip(*self.peak_aggregation()):
            out.write(f"{b}\t{a}\t{f}\t{p}\n")
        out.flush()

    def count_kmers(self):
        f, s = self.freq, self.smudge
        self.total_kmers = int(f.sum())
        self.total_genomic_kmers = int(f[s != -1].sum())
        self.total_genomic_kmers_in_smudges = int(f[s > 0].sum())
        self.total_error_kmers = int(f[s == -1].sum())
        self.error_fraction = self.total_error_kmers / self.total_kmers


def fishnet_centralities(covB, covA, freq, smudge, total_genomic_kmers, covs, smudge_filter=0.0):
    """centrality of every 1n-coverage candidate in `covs` (float64 array): smudgeplot.py:150-176 + 307-352, in C"""
    covB = np.ascontiguousarray(covB, dtype=np.int32)
    covA = np.ascontiguousarray(covA, dtype=np.int32)
    freq = np.ascontiguousarray(freq, dtype=np.int64)
    smudge = np.ascontiguousarray(smudge, dtype=np.int32)
    covs = np.ascontiguousarray(covs, dtype=np.float64)
    out = np.zeros(len(covs), dtype=np.float64)
    lib = _lib()
    lib.smg_fishnet_centralities.restype = C.c_int
    lib.smg_fishnet_centralities.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                             C.c_double, C.c_void_p, C.c_int64, C.c_void_p]
    rc = lib.smg_fishnet_centralities(covB.ctypes.data, covA.ctypes.data, freq.ctypes.data, smudge.ctypes.data, len(freq),
                                      int(total_genomic_kmers), float(smudge_filter), covs.ctypes.data, len(covs),
                                      out.ctypes.data)
    if rc != 0:
        raise RuntimeError("smg_fishnet_centralities failed")
    return out


class Smudges:
    """The 1n-coverage search of the reference's class of the same name (smudgeplot.py:96-148), behind the same two
    entry points (`get_centrality_df`, then `cov` / `centrality_df`).  What has to agree with the reference -- and is
    pinned by tests/golden/centrality.json -- is the LIST of candidate coverages (three nested numpy.arange grids whose
    end points and steps are below, then half of the winner), their centralities and the winner; how the search is
    written down is this module's own.  `centrality_df` is a dict of two float64 arrays instead of a DataFrame.
    cov_tab = (covB, covA, freq, smudge) in the reference's order after peak_aggregation (by covA, covB)."""

    # numpy.arange(int(lo) + a, int(hi) + b, step) around the previous winner (lo = hi = winner after the first grid)
    REFINEMENTS = ((0.05, 0.05, 2), (-1.9, 1.9, 0.2), (-0.19, 0.19, 0.01))

    def __init__(self, cov_tab, total_genomic_kmers):
        self.covB, self.covA, self.freq, self.smudge = (np.asarray(c) for c in cov_tab)
        self.total_genomic_kmers = int(total_genomic_kmers)
        self.cov = None
        self.centrality_df = None

    def _score(self, candidates, cutoff):
        return fishnet_centralities(self.covB, self.covA, self.freq, self.smudge, self.total_genomic_kmers, candidates, cutoff)

    def get_centrality_df(self, min_c, max_c, smudge_size_cutoff=0.02, log=None):
        tried, scores = [], []
        lo, hi, winner = min_c, max_c, None
        for depth, (a, b, step) in enumerate(self.REFINEMENTS):
            grid = np.arange(int(lo) + a, int(hi) + b, step)
            cen = self._score(grid, smudge_size_cutoff)
            winner = grid[int(np.argmin(cen))]
            tried.append(grid); scores.append(cen)
            lo = hi = winner
            if depth and log:
                log.write(f"Best coverage to precision of 1/{10 ** depth}: {winner:.2f}\n")
        # the reference's last look: half of the winner is appended to the finest grid and numpy.argmin picks again
        # (the first of equal minima wins -- the grid stands in front -- and the first NaN, if there is one: the same
        # formulation as smudgeplot.py:136-148, so that a NaN centrality selects what it selects there)
        half = winner / 2
        cen_half = float(self._score(np.array([half]), smudge_size_cutoff)[0])
        tried[-1] = np.append(tried[-1], half)
        scores[-1] = np.append(scores[-1], cen_half)
        winner = tried[-1][int(np.argmin(scores[-1]))]
        if log:
            log.write(f"Best coverage to precision of 1/{10 ** (len(self.REFINEMENTS) - 1)} (just to be sure): {winner:.2f}\n")
        self.cov = winner
        self.centrality_df = {"coverage": np.concatenate(tried), "centrality": np.concatenate(scores).astype(np.float64)}
