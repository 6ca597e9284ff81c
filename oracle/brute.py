"""brute.py -- TEST INFRASTRUCTURE ONLY (numpy restatement, small tables).

Second, independent CPU restatement of the hetmers semantics (SURVEY.md section 8a "Restated
semantics"; reference rules at src/lib/PloidyPlot.c:528-540 pass 1, 657-671 pass 2, 1603-1617
writer).  For every position p the k-mers are grouped by the packed bytes with base p zeroed
(np.unique on a void view); pairs, wrapping uint8 degrees and the (sum, min) histogram follow.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

SMAX, FMAX = 1000, 500


def unique_pairs(packed: np.ndarray, counts: np.ndarray, k: int):
    """-> (a, b, pos) index arrays of the pairs that enter the plot (a < b in table order, i.e. a carries
    the smaller base at `pos`), after the deg <= 1 filter."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    n, kb = packed.shape
    cnt = counts.astype(np.int64)
    deg = np.zeros(n, dtype=np.int64)
    pa, pb, pp = [], [], []
    for p in range(k):
        m = packed.copy()
        m[:, p >> 2] &= np.uint8(~(3 << (6 - 2 * (p & 3))) & 0xFF)
        v = m.view(np.dtype((np.void, kb))).ravel()
        order = np.argsort(v, kind="stable")
        vs = v[order]
        new = np.ones(n, dtype=bool)
        new[1:] = vs[1:] != vs[:-1]
        gid = np.cumsum(new) - 1
        for d in (1, 2, 3):
            same = gid[d:] == gid[:-d]
            a, b = order[:-d][same], order[d:][same]
            ok = cnt[a] + cnt[b] <= SMAX
            a, b = a[ok], b[ok]
            lo, hi = np.minimum(a, b), np.maximum(a, b)
            pa.append(lo); pb.append(hi); pp.append(np.full(len(lo), p, np.int64))
    pa = np.concatenate(pa) if pa else np.zeros(0, np.int64)
    pb = np.concatenate(pb) if pb else np.zeros(0, np.int64)
    pp = np.concatenate(pp) if pp else np.zeros(0, np.int64)
    np.add.at(deg, pa, 1)
    np.add.at(deg, pb, 1)
    deg8 = deg & 0xFF
    keep = (deg8[pa] <= 1) & (deg8[pb] <= 1)
    return pa[keep], pb[keep], pp[keep]


def extract_lines(packed: np.ndarray, counts: np.ndarray, k: int, labels: dict) -> dict:
    """Restates `extract_kmer_pairs` (src/lib/PloidyList.c:424-448 sink, 128-165 print_het, 1312-1346 .sma
    labels): labels maps a pixel (covB, covA) to a smudge name "<a>A<b>B"; every pair that enters the plot
    at a labelled pixel prints ONE line into that smudge's file -- the sequence of the member with the
    larger count (ties: the member with the smaller base at the variant position) with `(x/y)` at the
    variant position, y = the other member's base.  -> {smudge name: sorted list of lines}.
    The reference writes the lines of a file in thread-schedule order; only the multiset is defined."""
    a, b, pos = unique_pairs(packed, counts, k)
    cnt = counts.astype(np.int64)
    out = {name: [] for name in set(labels.values())}
    dna = "acgt"
    for ia, ib, p in zip(a.tolist(), b.tolist(), pos.tolist()):
        ca, cb = int(cnt[ia]), int(cnt[ib])
        name = labels.get((min(ca, cb), max(ca, cb)))
        if name is None:
            continue
        who, other = (ib, ia) if ca < cb else (ia, ib)
        row = packed[who]
        bases = [(int(row[q >> 2]) >> (6 - 2 * (q & 3))) & 3 for q in range(k)]
        alt = (int(packed[other][p >> 2]) >> (6 - 2 * (p & 3))) & 3
        out[name].append("".join(dna[x] for x in bases[:p]) + f"({dna[bases[p]]}/{dna[alt]})"
                         + "".join(dna[x] for x in bases[p + 1:]) + "\n")
    return {name: sorted(v) for name, v in out.items()}


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
