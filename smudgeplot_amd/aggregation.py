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
    """(covB, covA, freq) int arrays of a .smu file, rows by freq descending (smudgeplot.py:789-791; ties keep the
    file's order here -- the reference leaves their order to pandas' unstable sort)"""
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
        raise RuntimeError("smg_local_aggregation failed (negative coverage, bad distance, or out of memory)")
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
    """The 1n-coverage grid search of the reference's class of the same name (smudgeplot.py:96-148): the same three
    grids (step 2, then 0.2 around the best, then 0.01 around that, then best/2 "just to be sure"), the same
    arrays in `centrality_df` (a dict of two float64 arrays instead of a DataFrame), the same `cov`.
    cov_tab = (covB, covA, freq, smudge) in the reference's order after peak_aggregation (by covA, covB)."""

    def __init__(self, cov_tab, total_genomic_kmers):
        self.covB, self.covA, self.freq, self.smudge = (np.asarray(c) for c in cov_tab)
        self.total_genomic_kmers = int(total_genomic_kmers)
        self.cov = None
        self.centrality_df = None

    def get_best_coverage(self, cov_list, smudge_size_cutoff=0.02, centralities=None, last_check=False):
        if centralities is None:
            centralities = []
        to_test = cov_list[-1:] if last_check else cov_list
        centralities = list(centralities) + fishnet_centralities(self.covB, self.covA, self.freq, self.smudge,
                                                                 self.total_genomic_kmers, to_test, smudge_size_cutoff).tolist()
        return cov_list[int(np.argmin(centralities))], centralities

    def get_centrality_df(self, min_c, max_c, smudge_size_cutoff=0.02, log=None):
        grid_params = [(0.05, 0.05, 2), (-1.9, 1.9, 0.2), (-0.19, 0.19, 0.01)]
        results = []
        for i, params in enumerate(grid_params):
            cov_list = np.arange(int(min_c) + params[0], int(max_c) + params[1], params[2])
            best_cov, centralities = self.get_best_coverage(cov_list, smudge_size_cutoff)
            results.append({"covs": cov_list, "centralities": centralities, "best_cov": best_cov})
            min_c, max_c = best_cov, best_cov
            if i > 0 and log:
                log.write(f"Best coverage to precision of 1/{10**i}: {best_cov:.2f}\n")
        results[-1]["covs"] = np.append(results[-1]["covs"], results[-1]["best_cov"] / 2)
        best_cov, centralities = self.get_best_coverage(cov_list=results[-1]["covs"], smudge_size_cutoff=smudge_size_cutoff,
                                                        centralities=results[-1]["centralities"], last_check=True)
        results[-1]["centralities"] = centralities
        if log:
            log.write(f"Best coverage to precision of 1/{10**i} (just to be sure): {best_cov:.2f}\n")
        self.cov = best_cov
        self.centrality_df = {"coverage": np.concatenate([r["covs"] for r in results]),
                              "centrality": np.concatenate([np.asarray(r["centralities"], dtype=np.float64) for r in results])}
