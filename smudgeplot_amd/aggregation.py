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
