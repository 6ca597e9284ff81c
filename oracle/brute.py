"""brute.py -- TEST INFRASTRUCTURE ONLY (numpy restatement, small tables).

Second, independent CPU restatement of the hetmers semantics (SURVEY.md section 8a "Restated
semantics"; reference rules at src/lib/PloidyPlot.c:528-540 pass 1, 657-671 pass 2, 1603-1617
writer).  For every position p the k-mers are grouped by the packed bytes with base p zeroed
(np.unique on a void view); pairs, wrapping uint8 degrees and the (sum, min) histogram follow.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

SMAX, FMAX = 1000, 500


def hetmers_plot(packed: np.ndarray, counts: np.ndarray, k: int) -> np.ndarray:
    """-> plot[1001][501] int64, plot[sum][min]."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    n, kb = packed.shape
    cnt = counts.astype(np.int64)
    deg = np.zeros(n, dtype=np.int64)
    pa, pb = [], []
    for p in range(k):
        m = packed.copy()
        m[:, p >> 2] &= np.uint8(~(3 << (6 - 2 * (p & 3))) & 0xFF)
        v = m.view(np.dtype((np.void, kb))).ravel()
        order = np.argsort(v, kind="stable")
        vs = v[order]
        new = np.ones(n, dtype=bool)
        new[1:] = vs[1:] != vs[:-1]
        gid = np.cumsum(new) - 1
        for d in (1, 2, 3):                      # groups have at most 4 members
            same = gid[d:] == gid[:-d]
            a, b = order[:-d][same], order[d:][same]
            ok = cnt[a] + cnt[b] <= SMAX
            pa.append(a[ok]); pb.append(b[ok])
    pa = np.concatenate(pa) if pa else np.zeros(0, np.int64)
    pb = np.concatenate(pb) if pb else np.zeros(0, np.int64)
    np.add.at(deg, pa, 1)
    np.add.at(deg, pb, 1)
    deg8 = deg & 0xFF                            # uint8 wrap, PloidyPlot.c:163
    keep = (deg8[pa] <= 1) & (deg8[pb] <= 1)
    s = cnt[pa[keep]] + cnt[pb[keep]]
    mn = np.minimum(cnt[pa[keep]], cnt[pb[keep]])
    plot = np.zeros((SMAX + 1, FMAX + 1), dtype=np.int64)
    np.add.at(plot, (s, mn), 1)
    return plot


def smu_text(plot: np.ndarray) -> str:
    """The reference writer, PloidyPlot.c:1612-1615 (min == 500 is never printed)."""
    out = []
    for s in range(SMAX + 1):
        row = plot[s]
        for m in np.nonzero(row[:FMAX])[0]:
            out.append(f"{m}\t{s - m}\t{row[m]}\n")
    return "".join(out)
