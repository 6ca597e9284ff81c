"""Synthetic, already-conditioned (trimmed + symmetric) FastK tables.

The reference ships no data and FastK cannot run offline, so every test
and bench input is generated here and written in format F by `ktab.write_ktab`.

Two generators:
  * `adversarial_table`  -- small, any k: random k-mers with 1..3 single-base variants at random
    positions (unique pairs AND non-unique groups of 3-4), counts drawn from values that sit on
    the SMAX=1000 / FMAX=500 edges, low-complexity (poly-A like) seeds that create dense local
    neighbourhoods, optional palindromes for even k.
  * `diploid_table_u64`  -- large, k <= 32, vectorised on uint64: a "diploid genome" stand-in where
    a fraction of k-mers has exactly one one-away partner; Poisson coverage.
"""

from __future__ import annotations

import numpy as np

from . import ktab

# counts above 32767 are undefined behaviour in the reference: examine_table indexes a 0x8000-entry
# stack histogram with the count read as int16 (PloidyPlot.c:1171,1189), so a count of 65535
# writes hist[-1].  FastK itself saturates at 32767.  Generators therefore stay <= 32767.
EDGE_COUNTS = np.array([20, 21, 40, 60, 499, 500, 501, 600, 999, 1000, 32767], dtype=np.int64)


def adversarial_table(k: int, m: int, L: int, seed: int, low_complexity: int = 0,
                      dense: int = 0):
    """Return (packed [N,kbyte] uint8 sorted, counts [N] uint16) of a symmetric trimmed table."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 4, size=(m, k), dtype=np.uint8)
    rows = [base]
    # single-base variants: p(1)=.5, p(2)=.15, p(3)=.05 (survey section 8c)
    nvar = rng.choice([0, 1, 2, 3], size=m, p=[0.30, 0.50, 0.15, 0.05])
    for nv in (1, 2, 3):
        src = base[nvar >= nv].copy()
        if len(src):
            pos = rng.integers(0, k, size=len(src))
            delta = rng.integers(1, 4, size=len(src)).astype(np.uint8)
            src[np.arange(len(src)), pos] = (src[np.arange(len(src)), pos] + delta) & 3
            rows.append(src)
    if low_complexity:
        # long shared prefixes: big local windows (poly-A like), variants near both ends
        lc = np.zeros((low_complexity, k), dtype=np.uint8)
        tail = max(2, min(6, k // 4))
        lc[:, k - tail:] = rng.integers(0, 4, size=(low_complexity, tail), dtype=np.uint8)
        lc[:, :1] = rng.integers(0, 2, size=(low_complexity, 1), dtype=np.uint8)
        rows.append(lc)
    if dense:
        # dense neighbourhood: all 4^t completions of a fixed (k-t)-prefix => every entry has
        # 3t partners (degree far above 1; exercises the "non unique" rule and window overflow)
        t = 3
        pre = rng.integers(0, 4, size=(dense, 1, k - t), dtype=np.uint8)
        comb = np.array(np.meshgrid(*[np.arange(4)] * t, indexing="ij"), dtype=np.uint8)
        comb = comb.reshape(t, -1).T[None, :, :]
        blk = np.concatenate([np.repeat(pre, comb.shape[1], axis=1),
                              np.repeat(comb, dense, axis=0)], axis=2)
        rows.append(blk.reshape(-1, k))
    bases = np.concatenate(rows)
    packed = ktab.pack_bases(bases)
    n = packed.shape[0]
    cnt = rng.integers(L, 81, size=n)
    edge = rng.random(n) < 0.25
    cnt[edge] = np.maximum(L, rng.choice(EDGE_COUNTS, size=int(edge.sum())))
    packed, cnt = ktab.sort_unique_packed(packed, cnt.astype(np.uint16))
    packed, cnt = ktab.symmetrize(packed, cnt, k)
    return packed, cnt.astype(np.uint16)


def diploid_table_u64(n0: int, k: int = 31, seed: int = 1, het_frac: float = 0.3,
                      cov: float = 50.0, L: int = 10):
    """Large symmetric table for k <= 32 as (keys uint64 sorted, counts uint16).

    n0 random k-mers (homozygous, Poisson(cov)); a fraction het_frac of them is turned into a
    heterozygous pair: both members get Poisson(cov/2) and differ at one random position.
    Reverse complements are added with equal counts, the table is sorted and de-duplicated.
    """
    assert k <= 32
    rng = np.random.default_rng(seed)
    keys = rng.integers(0, 1 << 62, size=n0, dtype=np.uint64) << np.uint64(2)
    keys |= rng.integers(0, 4, size=n0, dtype=np.uint64)
    keys &= ~np.uint64(0) << np.uint64(2 * (32 - k))
    cnt = rng.poisson(cov, size=n0)
    het = rng.random(n0) < het_frac
    nh = int(het.sum())
    pos = rng.integers(0, k, size=nh).astype(np.uint64)
    delta = rng.integers(1, 4, size=nh).astype(np.uint64)
    partner = keys[het] ^ (delta << (np.uint64(62) - np.uint64(2) * pos))
    cnt[het] = rng.poisson(cov / 2, size=nh)
    pcnt = rng.poisson(cov / 2, size=nh)
    keys = np.concatenate([keys, partner])
    cnt = np.clip(np.concatenate([cnt, pcnt]), L, 32767).astype(np.uint16)
    keys = np.concatenate([keys, ktab.revcomp_u64(keys, k)])
    cnt = np.concatenate([cnt, cnt])
    order = np.argsort(keys, kind="stable")
    keys, cnt = keys[order], cnt[order]
    keep = np.ones(len(keys), dtype=bool)
    keep[1:] = keys[1:] != keys[:-1]
    keys, cnt = keys[keep], cnt[keep]
    # dedupe kept the first of equal keys; a k-mer and an unrelated k-mer's complement can
    # collide with different counts -> re-impose symmetry of the counts
    rc = ktab.revcomp_u64(keys, k)
    j = np.searchsorted(keys, rc)
    cnt = np.minimum(cnt, cnt[j])
    return keys, cnt


def write_u64_table(path: str, keys: np.ndarray, cnt: np.ndarray, k: int, ibyte: int = 3,
                    nparts: int = 1) -> None:
    ktab.write_ktab(path, k, ktab.u64_to_packed(keys, k), cnt, ibyte=ibyte, nparts=nparts)
