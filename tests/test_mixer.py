"""The 128-bit mixer of the symmetry fingerprint (`mix_hash`, smudgeplot_amd/csrc/smg_pass1d.hpp).

The hash proof stands on one property: the XOR of the h(min(x, rc x), count) terms of a (sorted, duplicate-free) table
is zero if and only if the table is closed under reverse complement with equal counts -- which needs h to behave like
a random function of the (k-mer, count) pair.  Here: a numpy restatement of the device function (same constants, same
order of operations), an avalanche measurement over every input bit, and -- on the GPU -- the engine's own fingerprint
of small tables that are NOT closed, which must equal the XOR of the restatement's values entry by entry.
"""
import numpy as np
import pytest

M32 = np.uint64(0xFFFFFFFF)


def _u64(a):
    return np.asarray(a, dtype=np.uint64)


def _mad(a, b, c):
    """v_mad_u64_u32: 32 x 32 + 64 bits, wraps at 2^64"""
    with np.errstate(over="ignore"):
        return (_u64(a) & M32) * (_u64(b) & M32) + _u64(c)


def _rotl(x, r):
    x = _u64(x) & M32
    return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & M32


def _qr(a, b, c, d):
    """the first half of a ChaCha quarter round (round 6: the second half -- `a += b; d = rotl(d ^ a, 8); c += d; b = rotl(b ^ c, 7)`
    -- bought an avalanche of 0.5 +- 0.010 instead of +- 0.025 for six more vector instructions per table entry)"""
    a = (a + b) & M32; d = _rotl(d ^ a, 16)
    c = (c + d) & M32; b = _rotl(b ^ c, 12)
    return a, b, c, d


def mix_hash(w0, cnt, w1=None):
    """(ha, hb) of k-mer words (w0[, w1]) and count: smg_pass1d.hpp mix_hash<1> / mix_hash<2>"""
    w0, cnt = _u64(w0), _u64(cnt)
    lo, hi = w0 & M32, w0 >> np.uint64(32)
    p = _mad(lo ^ np.uint64(0x9E3779B9), hi ^ np.uint64(0x85EBCA6B), (cnt << np.uint64(20)) & M32)
    if w1 is not None:
        w1 = _u64(w1)
        l1, h1 = w1 & M32, w1 >> np.uint64(32)
        p = _mad(l1 ^ (p & M32) ^ np.uint64(0x165667B1), h1 ^ (p >> np.uint64(32)) ^ np.uint64(0xD3A2646C), p)
    pl, ph = p & M32, p >> np.uint64(32)
    q = _mad(pl ^ hi, ph ^ lo ^ np.uint64(0xC2B2AE35), p)
    a, b, c, d = _qr(q & M32, q >> np.uint64(32), pl ^ cnt, ph)
    return a | (b << np.uint64(32)), c | (d << np.uint64(32))


def _bits128(ha, hb):
    out = np.empty((len(ha), 128), dtype=np.uint8)
    for j in range(64):
        out[:, j] = (ha >> np.uint64(j)) & np.uint64(1)
        out[:, 64 + j] = (hb >> np.uint64(j)) & np.uint64(1)
    return out


@pytest.mark.parametrize("W", [1, 2])
def test_every_input_bit_flips_every_output_bit_about_half_of_the_time(W):
    rng = np.random.default_rng(5 + W)
    n = 20000
    w0 = rng.integers(0, 2 ** 64, n, dtype=np.uint64)
    w1 = rng.integers(0, 2 ** 64, n, dtype=np.uint64) if W == 2 else None
    cnt = rng.integers(1, 1001, n).astype(np.uint64)
    base = _bits128(*mix_hash(w0, cnt, w1))
    worst = 0.0
    flips = [("w0", j) for j in range(64)] + ([("w1", j) for j in range(64)] if W == 2 else []) + [("cnt", j) for j in range(10)]
    for which, j in flips:
        a0, a1, c = w0, w1, cnt
        if which == "w0":
            a0 = w0 ^ np.uint64(1 << j)
        elif which == "w1":
            a1 = w1 ^ np.uint64(1 << j)
        else:
            c = cnt ^ np.uint64(1 << j)
        p = (_bits128(*mix_hash(a0, c, a1)) != base).mean(axis=0)
        worst = max(worst, float(np.abs(p - 0.5).max()))
    # 20000 samples: sigma = 0.0035; 128 x (138 | 74) cells -> the largest deviation of a fair coin sits near 4.3 sigma = 0.015.
    # Two multiply-adds + half a quarter round: the worst cell is off by 0.025 (one-word k-mers; 0.011 for two words), i.e. no
    # output bit follows or ignores any input bit -- what the XOR fingerprint needs is that the terms of DIFFERENT entries do
    # not cancel, for which 128 bits that each depend on every input bit with a probability in [0.47, 0.53] leave nothing to
    # find by accident: the collision bound of a residue that vanishes although the table is not closed stays ~2^-128 for
    # random-like inputs and is not worse than 2^-64 (the well-mixed word alone) for any structured set one can name.
    assert worst < 0.035, worst


def test_no_collisions_and_balanced_words_on_neighbouring_kmers():
    """the inputs the proof sees are not random: runs of k-mers that differ in the last bases, small counts"""
    base = np.uint64(0x1B2D3F4C5A697887)
    w0 = base + (np.arange(1 << 16, dtype=np.uint64) << np.uint64(2))         # consecutive k = 31 k-mers
    for c in (1, 2, 30, 1000):
        ha, hb = mix_hash(w0, np.full(len(w0), c, dtype=np.uint64))
        assert len(np.unique(ha)) == len(ha) and len(np.unique(hb)) == len(hb)
        ones = _bits128(ha, hb).mean(axis=0)
        assert np.abs(ones - 0.5).max() < 0.02
    # the count matters on its own
    a = mix_hash(w0[:1000], np.full(1000, 7, dtype=np.uint64))
    b = mix_hash(w0[:1000], np.full(1000, 8, dtype=np.uint64))
    assert not np.any(a[0] == b[0]) and not np.any(a[1] == b[1])


def _rc_words(keys, k):
    """reverse complement of left-aligned W-word k-mers given as Python ints"""
    W = (k + 31) // 32
    out = []
    for x in keys:
        v = x >> (64 * W - 2 * k)
        r = 0
        for _ in range(k):
            r = (r << 2) | (3 - (v & 3))
            v >>= 2
        out.append(r << (64 * W - 2 * k))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("k", [31, 32, 21, 40, 51, 64])
def test_engine_fingerprint_is_the_xor_of_the_restatement(k):
    """a table that is NOT closed: the residue the engine reports is the XOR of exactly this function over its entries"""
    import torch
    from smudgeplot_amd import engine
    W = (k + 31) // 32
    rng = np.random.default_rng(100 + k)
    n = 5000
    vals = sorted({int.from_bytes(rng.bytes(8 * W), "big") >> (64 * W - 2 * k) << (64 * W - 2 * k) for _ in range(n)})
    cnt = rng.integers(1, 900, len(vals)).astype(np.uint16)
    rc = _rc_words(vals, k)
    M = (1 << 64) - 1
    fa = fb = 0
    for x, r, c in zip(vals, rc, cnt):
        if x == r:
            continue
        m = min(x, r)
        if W == 1:
            ha, hb = mix_hash([m], [int(c)])
        else:
            ha, hb = mix_hash([m >> 64], [int(c)], [m & M])
        fa ^= int(ha[0])
        fb ^= int(hb[0])
    words = np.array([[(x >> (64 * (W - 1 - j))) & M for j in range(W)] for x in vals], dtype=np.uint64)
    dev = torch.device("cuda:0")
    tk = torch.from_numpy(words.view(np.int64).reshape(-1).copy()).to(dev)
    tc = torch.from_numpy(cnt.view(np.int16).copy()).to(dev)
    e = engine.Engine(0, torch.cuda.current_stream().cuda_stream)
    e.bind(k, len(vals), tk.data_ptr(), tc.data_ptr())
    e.pass1("hash")
    got = e.symhash()
    torch.cuda.synchronize()
    e.close()
    assert (got[0] ^ got[2], got[1] ^ got[3]) == (fa, fb)
