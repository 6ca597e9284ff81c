"""FastK k-mer table ("format F") reader / writer in numpy.

Host-side tooling for tests, fixtures and the bench: it lets us write tables the reference
`hetmers` (and our engine) accept, without FastK, which is not available offline.

Format (reference: src/lib/libfastk.c:786-908 `Open_Kmer_Stream`, 1230-1269 `Current_Entry`):

  <dir>/<root>.ktab          stub : int32 kmer, int32 nparts, int32 minval, int32 ibyte,
                                    int64 index[2^(8*ibyte)]   (index[p] = number of entries whose
                                    first ibyte bytes are <= p, i.e. cumulative END offsets)
  <dir>/.<root>.ktab.<1..n>  parts: int32 kmer, int64 n, then n records of pbyte bytes =
                                    (kbyte-ibyte) suffix bytes + uint16 count (little endian)

k-mers are packed 2 bits/base (a=0 c=1 g=2 t=3), base 0 in bits 7..6 of byte 0, the last byte
zero padded (libfastk.c:614-636); entries are globally sorted by packed bytes; parts are
consecutive ranges that must break on prefix boundaries (GoTo_Kmer_Entry, libfastk.c:1360-1386).
"""

from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

__all__ = [
    "kbyte_of", "pack_bases", "unpack_bases", "revcomp_packed", "sort_unique_packed",
    "symmetrize", "write_ktab", "read_ktab", "KTable", "packed_to_u64", "u64_to_packed",
    "revcomp_u64",
]


def kbyte_of(k: int) -> int:
    return (k + 3) >> 2


def pack_bases(bases: np.ndarray) -> np.ndarray:
    """[N,k] uint8 in 0..3 -> [N,kbyte] uint8 packed, base 0 in the top bits of byte 0."""
    bases = np.asarray(bases, dtype=np.uint8)
    n, k = bases.shape
    kb = kbyte_of(k)
    pad = np.zeros((n, kb * 4), dtype=np.uint8)
    pad[:, :k] = bases
    q = pad.reshape(n, kb, 4)
    return ((q[:, :, 0] << 6) | (q[:, :, 1] << 4) | (q[:, :, 2] << 2) | q[:, :, 3]).astype(np.uint8)


def unpack_bases(packed: np.ndarray, k: int) -> np.ndarray:
    packed = np.asarray(packed, dtype=np.uint8)
    n, kb = packed.shape
    out = np.empty((n, kb, 4), dtype=np.uint8)
    out[:, :, 0] = packed >> 6
    out[:, :, 1] = (packed >> 4) & 3
    out[:, :, 2] = (packed >> 2) & 3
    out[:, :, 3] = packed & 3
    return out.reshape(n, kb * 4)[:, :k]


def revcomp_packed(packed: np.ndarray, k: int) -> np.ndarray:
    b = unpack_bases(packed, k)
    return pack_bases((3 - b[:, ::-1]).astype(np.uint8))


def _as_void(packed: np.ndarray) -> np.ndarray:
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    return packed.view(np.dtype((np.void, packed.shape[1]))).ravel()


def sort_unique_packed(packed: np.ndarray, counts: np.ndarray):
    """Sort rows bytewise (memcmp order) and drop duplicate k-mers (first occurrence wins)."""
    if packed.shape[0] == 0:
        return packed, counts
    v = _as_void(packed)
    order = np.argsort(v, kind="stable")
    packed = packed[order]
    counts = counts[order]
    v = v[order]
    keep = np.ones(len(v), dtype=bool)
    keep[1:] = v[1:] != v[:-1]
    return packed[keep], counts[keep]


def symmetrize(packed: np.ndarray, counts: np.ndarray, k: int):
    """Add the reverse complement of every k-mer with the same count (what Symmex does)."""
    rc = revcomp_packed(packed, k)
    packed, counts = sort_unique_packed(np.concatenate([packed, rc]),
                                        np.concatenate([counts, counts]))
    # x may also be the complement of an unrelated y that carried another count: the
    # de-duplication kept one of the two, so re-impose count(x) == count(rc(x))
    j = np.searchsorted(_as_void(packed), _as_void(revcomp_packed(packed, k)))
    return packed, np.minimum(counts, counts[j])


# ---- 64-bit views for k <= 32 (left aligned: base 0 in bits 63..62) ----

def packed_to_u64(packed: np.ndarray) -> np.ndarray:
    n, kb = packed.shape
    assert kb <= 8
    buf = np.zeros((n, 8), dtype=np.uint8)
    buf[:, :kb] = packed
    return buf.view(">u8").ravel().astype(np.uint64)


def u64_to_packed(keys: np.ndarray, k: int) -> np.ndarray:
    kb = kbyte_of(k)
    return np.ascontiguousarray(keys.astype(">u8").view(np.uint8).reshape(-1, 8)[:, :kb])


def revcomp_u64(keys: np.ndarray, k: int) -> np.ndarray:
    """Vectorised reverse complement of left-aligned 2-bit k-mers (k <= 32)."""
    x = ~keys.astype(np.uint64)                       # complement: 3-b == ~b on 2 bits
    m2, m4, m8, m16, m32 = (np.uint64(v) for v in (
        0x3333333333333333, 0x0F0F0F0F0F0F0F0F, 0x00FF00FF00FF00FF,
        0x0000FFFF0000FFFF, 0x00000000FFFFFFFF))
    x = ((x >> np.uint64(2)) & m2) | ((x & m2) << np.uint64(2))
    x = ((x >> np.uint64(4)) & m4) | ((x & m4) << np.uint64(4))
    x = ((x >> np.uint64(8)) & m8) | ((x & m8) << np.uint64(8))
    x = ((x >> np.uint64(16)) & m16) | ((x & m16) << np.uint64(16))
    x = ((x >> np.uint64(32)) & m32) | ((x & m32) << np.uint64(32))
    # now base k-1 sits where base 0 of a 32-mer would: shift out the (32-k) pad bases
    return x << np.uint64(2 * (32 - k))


@dataclass
class KTable:
    k: int
    ibyte: int
    nparts: int
    minval: int
    packed: np.ndarray      # [N,kbyte] uint8
    counts: np.ndarray      # [N] uint16
    index: np.ndarray       # [2^(8 ibyte)] int64
    part_nels: np.ndarray   # [nparts] int64

    @property
    def nels(self) -> int:
        return int(self.packed.shape[0])


def _paths(path: str):
    path = str(path)
    d, base = os.path.split(path)
    d = d or "."
    if base.lower().endswith(".ktab") and len(base) > 5:
        base = base[:-5]
    return d, base


def write_ktab(path: str, k: int, packed: np.ndarray, counts: np.ndarray, ibyte: int = 3,
               nparts: int = 1, minval: int = 1) -> None:
    """Write a sorted, duplicate-free table in format F.  `packed` must already be sorted."""
    assert ibyte in (1, 2, 3)
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    counts = np.asarray(counts).astype("<u2")
    n, kb = packed.shape
    assert kb == kbyte_of(k) and kb > ibyte, "kbyte must exceed ibyte"
    d, base = _paths(path)

    pre = np.zeros(n, dtype=np.int64)
    for j in range(ibyte):
        pre = (pre << 8) | packed[:, j].astype(np.int64)
    ixlen = 1 << (8 * ibyte)
    index = np.cumsum(np.bincount(pre, minlength=ixlen).astype(np.int64))

    # parts: consecutive ranges cut on prefix-bucket boundaries
    cuts = [0]
    for p in range(1, nparts):
        target = (n * p) // nparts
        b = int(np.searchsorted(index, target, side="left"))
        c = int(index[min(b, ixlen - 1)])
        cuts.append(max(c, cuts[-1]))
    cuts.append(n)

    with open(os.path.join(d, base + ".ktab"), "wb") as f:
        np.array([k, nparts, minval, ibyte], dtype="<i4").tofile(f)
        index.astype("<i8").tofile(f)

    rec = np.empty((n, kb - ibyte + 2), dtype=np.uint8)
    rec[:, : kb - ibyte] = packed[:, ibyte:]
    rec[:, kb - ibyte:] = counts.view(np.uint8).reshape(n, 2)
    for p in range(nparts):
        lo, hi = cuts[p], cuts[p + 1]
        with open(os.path.join(d, f".{base}.ktab.{p + 1}"), "wb") as f:
            np.array([k], dtype="<i4").tofile(f)
            np.array([hi - lo], dtype="<i8").tofile(f)
            rec[lo:hi].tofile(f)


def read_ktab(path: str) -> KTable:
    d, base = _paths(path)
    with open(os.path.join(d, base + ".ktab"), "rb") as f:
        k, nparts, minval, ibyte = np.fromfile(f, dtype="<i4", count=4)
        index = np.fromfile(f, dtype="<i8", count=1 << (8 * int(ibyte)))
    k, nparts, minval, ibyte = int(k), int(nparts), int(minval), int(ibyte)
    kb = kbyte_of(k)
    pb = kb + 2 - ibyte
    recs, part_nels = [], []
    for p in range(1, nparts + 1):
        with open(os.path.join(d, f".{base}.ktab.{p}"), "rb") as f:
            km = int(np.fromfile(f, dtype="<i4", count=1)[0])
            n = int(np.fromfile(f, dtype="<i8", count=1)[0])
            assert km == k
            recs.append(np.fromfile(f, dtype=np.uint8, count=n * pb).reshape(n, pb))
            part_nels.append(n)
    rec = np.concatenate(recs) if recs else np.zeros((0, pb), np.uint8)
    n = rec.shape[0]
    sizes = np.diff(np.concatenate([[0], index]))
    pre = np.repeat(np.arange(len(index), dtype=np.int64), sizes)[:n]
    packed = np.empty((n, kb), dtype=np.uint8)
    for j in range(ibyte):
        packed[:, j] = (pre >> (8 * (ibyte - 1 - j))) & 0xFF
    packed[:, ibyte:] = rec[:, : kb - ibyte]
    counts = np.ascontiguousarray(rec[:, kb - ibyte:]).view("<u2").ravel().astype(np.uint16)
    return KTable(k, ibyte, nparts, minval, packed, counts, index, np.array(part_nels, np.int64))


def remove_ktab(path: str) -> None:
    d, base = _paths(path)
    for name in os.listdir(d):
        if name == base + ".ktab" or name.startswith(f".{base}.ktab."):
            os.remove(os.path.join(d, name))
