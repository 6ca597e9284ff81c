"""On-device generation of the BASELINE "synthetic diploid" table (k <= 31) with torch.

torch is plumbing: the bench needs a ~2-3e9-entry conditioned FastK-like table resident in HBM
and there is neither network, FastK nor enough host time to build it on the CPU.

Model: a uniform random haploid genome of G bases, a second
haplotype that differs by SNPs at rate `het`, all k-mers of both haplotypes and of their reverse
complements; a k-mer seen in both haplotypes gets coverage ~Poisson(cov), one seen in a single
haplotype ~Poisson(cov/2) (normal approximation from a hash of the canonical k-mer, so a k-mer
and its complement always carry the same count), clipped to [L, 32767].  The result is sorted,
duplicate free, trimmed and reverse-complement closed: exactly what `hetmers` expects after
Logex/Symmex conditioning.
"""

from __future__ import annotations

import math

import torch


def _kmers_of(bases: torch.Tensor, k: int) -> torch.Tensor:
    """bases uint8 [G] in 0..3 -> int64 [G-k+1], k-mer right aligned in 2k bits."""
    g = bases.numel()
    out = torch.zeros(g - k + 1, dtype=torch.int64, device=bases.device)
    for j in range(k):
        out <<= 2
        out |= bases[j: g - k + 1 + j].to(torch.int64)
    return out


def _revcomp_right(x: torch.Tensor, k: int) -> torch.Tensor:
    """reverse complement of right-aligned 2k-bit k-mers (k <= 31)"""
    m = (1 << (2 * k)) - 1
    x = (~x) & m
    x = ((x >> 2) & 0x3333333333333333) | ((x & 0x3333333333333333) << 2)
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0F) | ((x & 0x0F0F0F0F0F0F0F0F) << 4)
    x = ((x >> 8) & 0x00FF00FF00FF00FF) | ((x & 0x00FF00FF00FF00FF) << 8)
    x = ((x >> 16) & 0x0000FFFF0000FFFF) | ((x & 0x0000FFFF0000FFFF) << 16)
    x = ((x >> 32) & 0x00000000FFFFFFFF) | ((x & 0x00000000FFFFFFFF) << 32)
    # logical shift right by 64-2k: clear the sign-extension afterwards
    return (x >> (64 - 2 * k)) & m


def _mix(z: torch.Tensor) -> torch.Tensor:
    z = (z ^ (z >> 30)) * -4658895280553007687          # 0xbf58476d1ce4e5b9
    z = (z ^ ((z >> 27) & 0x1FFFFFFFFF)) * -7723592293110705685   # 0x94d049bb133111eb
    return z ^ ((z >> 31) & 0x1FFFFFFFF)


def _plant_repeats(h1: torch.Tensor, frac: float, gen) -> None:
    """Overwrite about `frac` of the haploid genome with the three kinds of repeats real genomes have (SURVEY.md
    section 8d, value distributions 4 and 5): copies of one 3 kb unit with 2 % divergence (dispersed repeats), short
    motifs repeated in tandem, and homopolymer runs.  They give the engine what a uniform random genome never does:
    k-mers with many one-away neighbours, window blocks far beyond the +-30 entry window (kf_bigfix), counts in
    the repeat tail of the plot (global atomics instead of the LDS tile)."""
    G, dev = h1.numel(), h1.device
    if frac <= 0 or G < 200000:
        return
    ar = torch.arange(3000, device=dev)
    # dispersed: frac/2 of the genome
    unit = torch.randint(0, 4, (3000,), dtype=torch.uint8, device=dev, generator=gen)
    ncopy = min(max(int(G * frac / 2 / 3000), 2), G // 3000)
    # (starts on a grid of the segment length: two segments of one kind never overlap -- an indexed store with
    #  duplicate indices would leave the winner to the scheduler and the table would differ from run to run)
    starts = torch.randperm(G // 3000, device=dev, generator=gen)[:ncopy] * 3000
    idx = (starts[:, None] + ar[None, :]).reshape(-1)
    mut = torch.rand(idx.numel(), device=dev, generator=gen) < 0.02
    val = unit.repeat(ncopy)
    val = torch.where(mut, (val + torch.randint(1, 4, (idx.numel(),), dtype=torch.uint8, device=dev, generator=gen)) & 3, val)
    h1[idx] = val
    # tandem: frac/4, segments of 1000 bases, motif length 2..24
    nseg = max(int(G * frac / 4 / 1000), 1)
    starts = torch.randperm(G // 1000 - 1, device=dev, generator=gen)[:nseg] * 1000 + 37
    nseg = starts.numel()
    mlen = torch.randint(2, 25, (nseg,), device=dev, generator=gen)
    motif = torch.randint(0, 4, (nseg, 24), dtype=torch.uint8, device=dev, generator=gen)
    pos = ar[None, :1000] % mlen[:, None]
    h1[(starts[:, None] + ar[None, :1000]).reshape(-1)] = torch.gather(motif, 1, pos).reshape(-1)
    # homopolymers: frac/4, runs of 300
    nrun = max(int(G * frac / 4 / 300), 1)
    starts = torch.randperm(G // 300 - 1, device=dev, generator=gen)[:nrun] * 300 + 111
    nrun = starts.numel()
    base = torch.randint(0, 4, (nrun,), dtype=torch.uint8, device=dev, generator=gen)
    h1[(starts[:, None] + ar[None, :300]).reshape(-1)] = base[:, None].expand(nrun, 300).reshape(-1)


def diploid_table(G: int, k: int = 31, het: float = 0.01, cov: float = 50.0, L: int = 10,
                  seed: int = 1, device="cuda", chunks: int = 1, repeats: float = 0.0):
    """-> (keys int64 viewing LEFT-aligned uint64 k-mers, sorted as unsigned; counts int16
    viewing uint16).  k <= 31 so the right-aligned value is non-negative and sorts correctly.
    repeats > 0: that fraction of the genome is repeats (see _plant_repeats) and a k-mer's coverage scales with
    its copy number."""
    assert k <= 31
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    h1 = torch.randint(0, 4, (G,), dtype=torch.uint8, device=device, generator=gen)
    _plant_repeats(h1, repeats, gen)
    snp = torch.rand(G, device=device, generator=gen) < het
    delta = torch.randint(1, 4, (G,), dtype=torch.uint8, device=device, generator=gen)
    h2 = torch.where(snp, (h1 + delta) & 3, h1)
    del snp, delta
    km = [_kmers_of(h1, k), _kmers_of(h2, k)]
    del h1, h2
    # torch.unique (rocPRIM) takes < 2^31 elements: split the key space by leading bits
    total = 4 * km[0].numel()
    nchunk = 1
    while total / nchunk > 1.2e9:
        nchunk *= 2
    cb = nchunk.bit_length() - 1
    shift = 2 * k - cb
    key_parts, cnt_parts = [], []
    for c in range(nchunk):
        sel = []
        for a in km:
            sel.append(a[(a >> shift) == c] if cb else a)
            r = _revcomp_right(a, k)
            sel.append(r[(r >> shift) == c] if cb else r)
            del r
        allk = torch.cat(sel)
        del sel
        keys, mult = torch.unique(allk, sorted=True, return_counts=True)
        del allk
        # multiplicity 2+ => present in both haplotypes (or a genomic repeat): homozygous coverage
        canon = torch.minimum(keys, _revcomp_right(keys, k))
        u1 = (_mix(canon ^ 0x243F6A8885A308D3) >> 11).to(torch.float64) / float(1 << 53)
        u2 = (_mix(canon ^ 0x13198A2E03707344) >> 11).to(torch.float64) / float(1 << 53)
        del canon
        u1 = u1 - torch.floor(u1); u2 = u2 - torch.floor(u2)
        z = torch.sqrt(-2.0 * torch.log(u1.clamp_min(1e-300))) * torch.cos(2 * math.pi * u2)
        del u1, u2
        if repeats > 0:         # copy number: every occurrence in either haplotype (on either strand) adds cov / 2
            mean = (float(cov) / 2) * mult.clamp(max=1200).to(torch.float64)
        else:
            mean = torch.where(mult >= 2, torch.tensor(float(cov), device=device, dtype=torch.float64),
                               torch.tensor(float(cov) / 2, device=device, dtype=torch.float64))
        cnt = torch.round(mean + torch.sqrt(mean) * z).clamp_(L, 32767).to(torch.int16)
        del z, mean, mult
        key_parts.append(keys)
        cnt_parts.append(cnt)
    del km
    keys = torch.cat(key_parts) if nchunk > 1 else key_parts[0]
    cnt = torch.cat(cnt_parts) if nchunk > 1 else cnt_parts[0]
    del key_parts, cnt_parts
    keys <<= (64 - 2 * k)                      # left align: base 0 in bits 63..62
    return keys, cnt


def polyploid_table(G: int, ploidy: int = 4, div: float = 0.01, cov_hap: float = 25.0, k: int = 31,
                    L: int = 12, seed: int = 3265401, device="cuda"):
    """Stand-in for a real polyploid FastK table (BASELINE configs 1/2/4: the yeast SRR3265401 and
    strawberry data cannot be fetched offline): `ploidy` haplotypes, each the base genome with its own
    SNPs at rate `div`; a k-mer carried by d haplotypes gets coverage ~ Normal(d*cov_hap), trimmed at L
    (k-mers below L are DROPPED, like FastK -t / Logex would), both strands present.
    -> (keys int64 left aligned sorted, counts int16), conditioned (trimmed + symmetric)."""
    assert k <= 31 and 4 * ploidy * G < 1.2e9
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    base = torch.randint(0, 4, (G,), dtype=torch.uint8, device=device, generator=gen)
    parts = []
    for _ in range(ploidy):
        snp = torch.rand(G, device=device, generator=gen) < div
        delta = torch.randint(1, 4, (G,), dtype=torch.uint8, device=device, generator=gen)
        h = torch.where(snp, (base + delta) & 3, base)
        km = _kmers_of(h, k)
        parts += [km, _revcomp_right(km, k)]
        del snp, delta, h, km
    keys, mult = torch.unique(torch.cat(parts), sorted=True, return_counts=True)
    del parts
    canon = torch.minimum(keys, _revcomp_right(keys, k))
    u1 = (_mix(canon ^ 0x243F6A8885A308D3) >> 11).to(torch.float64) / float(1 << 53)
    u2 = (_mix(canon ^ 0x13198A2E03707344) >> 11).to(torch.float64) / float(1 << 53)
    u1 = u1 - torch.floor(u1); u2 = u2 - torch.floor(u2)
    z = torch.sqrt(-2.0 * torch.log(u1.clamp_min(1e-300))) * torch.cos(2 * math.pi * u2)
    mean = cov_hap * mult.clamp(max=ploidy).to(torch.float64)
    cnt = torch.round(mean + torch.sqrt(mean) * z).clamp_(0, 32767).to(torch.int16)
    keep = cnt >= L
    keys, cnt = keys[keep], cnt[keep]
    keys <<= (64 - 2 * k)
    return keys, cnt


def diploid_table_wide(G: int, k: int = 51, het: float = 0.01, cov: float = 50.0, L: int = 10, seed: int = 1,
                       device="cuda"):
    """The diploid model of `diploid_table` for 33 <= k <= 64 (two 64-bit words per k-mer): BASELINE
    configs[4] is a k=51 table.  -> (keys int64 [n, 2] viewing left-aligned uint64 words, sorted as
    unsigned pairs; counts int16), trimmed at L and closed under reverse complement with equal counts.

    The reverse complements are taken as the k-mers of the reverse-complemented haplotypes; every
    occurrence carries a per-position normal deviate, a table entry takes the MINIMUM over its
    occurrences (the occurrence sets of x and rc(x) mirror each other, so the two get the same count)."""
    assert 33 <= k <= 64
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    h1 = torch.randint(0, 4, (G,), dtype=torch.uint8, device=device, generator=gen)
    snp = torch.rand(G, device=device, generator=gen) < het
    delta = torch.randint(1, 4, (G,), dtype=torch.uint8, device=device, generator=gen)
    h2 = torch.where(snp, (h1 + delta) & 3, h1)
    z = torch.randn(G - k + 1, device=device, generator=gen, dtype=torch.float32)
    del snp, delta
    SIGN = -(1 << 63)

    def words(h):
        g = h.numel()
        m = g - k + 1
        w0 = torch.zeros(m, dtype=torch.int64, device=device)
        for j in range(32):
            w0 <<= 2
            w0 |= h[j: m + j].to(torch.int64)
        w1 = torch.zeros(m, dtype=torch.int64, device=device)
        for j in range(32, k):
            w1 <<= 2
            w1 |= h[j: m + j].to(torch.int64)
        w1 <<= 2 * (64 - k)
        return w0 ^ SIGN, w1 ^ SIGN            # biased: signed order == unsigned order

    W0, W1, Z = [], [], []
    for h in (h1, h2):
        a0, a1 = words(h)
        b0, b1 = words((3 - h).flip(0))        # occurrence i of h mirrors occurrence m-1-i of rc(h)
        W0 += [a0, b0]; W1 += [a1, b1]; Z += [z, z.flip(0)]
    del h1, h2
    w0 = torch.cat(W0); w1 = torch.cat(W1); zz = torch.cat(Z)
    del W0, W1, Z
    # order by (w0, w1): one sort on w0, then odd-even transposition inside the (tiny) runs of equal w0
    # (two stable sorts in a row came back unsorted for > 1e8 elements with this torch/ROCm build)
    w0, o = torch.sort(w0)
    w1, zz = w1[o], zz[o]
    del o
    for rnd in range(64):
        swapped = 0
        for parity in (0, 1):
            a = slice(parity, w0.numel() - 1, 2)
            b = slice(parity + 1, w0.numel(), 2)
            n2 = min(w0[a].numel(), w0[b].numel())
            ia = torch.arange(parity, parity + 2 * n2, 2, device=device)
            bad = (w0[ia] == w0[ia + 1]) & (w1[ia] > w1[ia + 1])
            idx = ia[bad]
            swapped += int(idx.numel())
            if idx.numel():
                t1, tz = w1[idx].clone(), zz[idx].clone()
                w1[idx], zz[idx] = w1[idx + 1], zz[idx + 1]
                w1[idx + 1], zz[idx + 1] = t1, tz
            del ia, bad, idx
        if swapped == 0:
            break
    assert bool(((w0[1:] > w0[:-1]) | ((w0[1:] == w0[:-1]) & (w1[1:] >= w1[:-1]))).all()), "generator: not sorted"
    first = torch.ones(w0.numel(), dtype=torch.bool, device=device)
    first[1:] = (w0[1:] != w0[:-1]) | (w1[1:] != w1[:-1])
    run = torch.cumsum(first, 0) - 1
    nrun = int(run[-1].item()) + 1
    zmin = torch.full((nrun,), float("inf"), device=device, dtype=torch.float32)
    zmin.scatter_reduce_(0, run, zz, reduce="amin")
    mult = torch.zeros(nrun, dtype=torch.int64, device=device)
    mult.scatter_add_(0, run, torch.ones_like(run))
    u0, u1 = w0[first] ^ SIGN, w1[first] ^ SIGN
    del w0, w1, zz, run, first
    mean = torch.where(mult >= 2, float(cov), float(cov) / 2)
    cnt = torch.round(mean + torch.sqrt(mean) * zmin).clamp_(0, 32767).to(torch.int16)
    keep = cnt >= L                            # x and rc(x) carry the same count: trimming keeps the closure
    u0, u1, cnt = u0[keep], u1[keep], cnt[keep].contiguous()
    keys = torch.empty((u0.numel(), 2), dtype=torch.int64, device=device)
    keys[:, 0] = u0                            # (torch.stack / 2-D boolean indexing came back scrambled
    keys[:, 1] = u1                            #  beyond ~1e8 rows with this torch/ROCm build)
    return keys, cnt
