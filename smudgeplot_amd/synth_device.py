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


def key_range_of(rank: int, world: int):
    """[lo, hi) of the LEADING 16 KEY BITS that rank `rank` of `world` owns when a table is generated shard by shard:
    equal slices of the key space (a uniform random genome fills it evenly), cut on 8-base boundaries -- which are
    window-block boundaries for every k >= 16, so the shards are the prefix shards of the sharded run and the cut values
    (lo << 48, left aligned) are its splitters.  The shards of all ranks, in rank order, ARE the table of world = 1."""
    return (rank * 65536) // world, ((rank + 1) * 65536) // world


def _in_range16(lead16: torch.Tensor, key_range) -> torch.Tensor:
    return (lead16 >= key_range[0]) & (lead16 < key_range[1])


def diploid_table(G: int, k: int = 31, het: float = 0.01, cov: float = 50.0, L: int = 10,
                  seed: int = 1, device="cuda", chunks: int = 1, repeats: float = 0.0, key_range=None):
    """-> (keys int64 viewing LEFT-aligned uint64 k-mers, sorted as unsigned; counts int16
    viewing uint16).  k <= 31 so the right-aligned value is non-negative and sorts correctly.
    repeats > 0: that fraction of the genome is repeats (see _plant_repeats) and a k-mer's coverage scales with
    its copy number.
    key_range = key_range_of(rank, world): only the entries whose leading 16 key bits lie in that range (one rank's
    shard of the table: the genome is the same on every rank, the k-mers outside the range are dropped before the sort)."""
    assert k <= 31
    assert key_range is None or k >= 16
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
    if key_range is not None:         # this rank's k-mers of both strands (a complement may fall into the range when its k-mer does not)
        own = []
        for a in km:
            own.append(a[_in_range16(a >> (2 * k - 16), key_range)])
            r = _revcomp_right(a, k)
            own.append(r[_in_range16(r >> (2 * k - 16), key_range)])
            del r
        del km
        return _diploid_finish(own, k, cov, L, repeats, device, both_strands=False)
    return _diploid_finish(km, k, cov, L, repeats, device, both_strands=True)


def _diploid_finish(km, k, cov, L, repeats, device, both_strands: bool):
    """occurrence lists (right-aligned k-mers; both_strands: their complements are added here) -> the sorted, counted table"""
    # torch.unique (rocPRIM) takes < 2^31 elements: split the key space by leading bits
    total = 4 * km[0].numel() if both_strands else sum(a.numel() for a in km)
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
            if both_strands:
                r = _revcomp_right(a, k)
                sel.append(r[(r >> shift) == c] if cb else r)
                del r
        allk = torch.cat(sel)
        del sel
        if allk.numel() == 0:
            continue
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
    if not key_parts:
        return torch.zeros(0, dtype=torch.int64, device=device), torch.zeros(0, dtype=torch.int16, device=device)
    keys = torch.cat(key_parts) if len(key_parts) > 1 else key_parts[0]
    cnt = torch.cat(cnt_parts) if len(cnt_parts) > 1 else cnt_parts[0]
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


# ---- polyploid stand-ins with GRADED divergences (BASELINE configs[3] octoploid k=31, configs[4] hexaploid k=51) ----------

def graded_haplotypes(G: int, ploidy: int, rates, gen, device):
    """`ploidy` haplotypes (uint8 [G], bases 0..3) of one uniform random genome.  rates[m-1] = SNP rate of the variant
    sets that are carried by exactly m haplotypes: every haplotype has a private set (m = 1), every pair (2j, 2j+1) a
    shared one (m = 2), and so on for groups of m consecutive haplotypes -- so the k-mer pairs of the table fall on the
    smudges A..AB (1 of P), A..ABB (2 of P), ... ("graded divergences so smudges
    AAAAAAAB ... AAAABBBB appear"), and k-mers that span two sites of different sets form the groups of 3-4 one-away
    neighbours that polyploid tables are made of."""
    base = torch.randint(0, 4, (G,), dtype=torch.uint8, device=device, generator=gen)
    haps = [base.clone() for _ in range(ploidy)]
    for m, rate in enumerate(rates, start=1):
        if rate <= 0 or m > ploidy:
            continue
        step = m
        if ploidy % m:
            step = 1
            while step < m:
                step *= 2
        for s in range(0, ploidy - m + 1, step):
            snp = torch.rand(G, device=device, generator=gen) < rate
            delta = torch.randint(1, 4, (G,), dtype=torch.uint8, device=device, generator=gen)
            for h in range(s, s + m):
                haps[h] = torch.where(snp, (haps[h] + delta) & 3, haps[h])
            del snp, delta
    del base
    return haps


def _normal_from_hash(canon: torch.Tensor) -> torch.Tensor:
    """a standard normal deviate per canonical k-mer (Box-Muller on two hashes): x and rc(x) get the same one"""
    u1 = (_mix(canon ^ 0x243F6A8885A308D3) >> 11).to(torch.float64) / float(1 << 53)
    u2 = (_mix(canon ^ 0x13198A2E03707344) >> 11).to(torch.float64) / float(1 << 53)
    u1 = u1 - torch.floor(u1); u2 = u2 - torch.floor(u2)
    return torch.sqrt(-2.0 * torch.log(u1.clamp_min(1e-300))) * torch.cos(2 * math.pi * u2)


def polyploid_table_graded(G: int, ploidy: int = 8, rates=(0.001, 0.0015, 0.001, 0.002), cov_hap: float = 14.0,
                           k: int = 31, L: int = 8, seed: int = 4, device="cuda", key_range=None):
    """Octoploid-like table at any size that fits (k <= 31): the haplotypes of `graded_haplotypes`, every k-mer of both
    strands, a k-mer carried by d haplotypes ~ Normal(d * cov_hap) (the deviate hashed from the canonical k-mer),
    entries below L dropped.  The key space is cut by leading bits so that no torch.unique sees more than 1.2e9 elements.
    -> (keys int64 left aligned, sorted as unsigned; counts int16): trimmed and closed under reverse complement.
    key_range: one rank's shard of that table (see diploid_table)."""
    assert k <= 31
    assert key_range is None or k >= 16
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    haps = graded_haplotypes(G, ploidy, rates, gen, device)
    km = [_kmers_of(h, k) for h in haps]
    del haps
    total = 2 * ploidy * km[0].numel()
    both_strands = key_range is None
    if key_range is not None:
        own = []
        for a in km:
            own.append(a[_in_range16(a >> (2 * k - 16), key_range)])
            r = _revcomp_right(a, k)
            own.append(r[_in_range16(r >> (2 * k - 16), key_range)])
            del r
        km = own
        del own
        total = sum(a.numel() for a in km)
    nchunk = 1
    while total / nchunk > 1.0e9:
        nchunk *= 2
    cb = nchunk.bit_length() - 1
    shift = 2 * k - cb
    key_parts, cnt_parts = [], []
    for c in range(nchunk):
        sel = []
        for a in km:
            sel.append(a[(a >> shift) == c] if cb else a)
            if both_strands:
                r = _revcomp_right(a, k)
                sel.append(r[(r >> shift) == c] if cb else r)
                del r
        allk = torch.cat(sel)
        del sel
        if allk.numel() == 0:
            continue
        keys, mult = torch.unique(allk, sorted=True, return_counts=True)
        del allk
        z = _normal_from_hash(torch.minimum(keys, _revcomp_right(keys, k)))
        mean = float(cov_hap) * mult.clamp(max=ploidy).to(torch.float64)
        cnt = torch.round(mean + torch.sqrt(mean) * z).clamp_(0, 32767).to(torch.int16)
        del z, mean, mult
        keep = cnt >= L
        key_parts.append(keys[keep]); cnt_parts.append(cnt[keep])
        del keys, cnt, keep
    del km
    if not key_parts:
        return torch.zeros(0, dtype=torch.int64, device=device), torch.zeros(0, dtype=torch.int16, device=device)
    keys = torch.cat(key_parts) if len(key_parts) > 1 else key_parts[0]
    cnt = torch.cat(cnt_parts) if len(cnt_parts) > 1 else cnt_parts[0]
    del key_parts, cnt_parts
    keys <<= (64 - 2 * k)
    return keys, cnt.contiguous()


def polyploid_table_wide(G: int, ploidy: int = 6, rates=(0.001, 0.0015, 0.002), cov_hap: float = 10.0, k: int = 51,
                         L: int = 5, seed: int = 5, device="cuda", max_chunk: float = 6.0e8, key_range=None):
    """Two-word k-mers (33 <= k <= 64) of `ploidy` graded haplotypes (ploidy = 2, rates = (het,) is the diploid model of
    diploid_table_wide; ploidy = 6 the hexaploid stand-in of BASELINE configs[4]), generated CHUNK BY CHUNK of the
    key space (the leading bases of a k-mer pick its chunk, a chunk is at most `max_chunk` occurrences), so that the
    peak is the finished table plus one chunk's scratch instead of four copies of everything.

    Every occurrence (haplotype, strand, position) carries the normal deviate of its position (mirrored for the
    reverse strand); an entry takes the minimum over its occurrences and Normal(d * cov_hap) with d = occurrences
    (<= ploidy): the occurrence sets of x and rc(x) mirror each other, so both get the same count.  Entries below L are
    dropped.  -> (keys int64 [n, 2] viewing left-aligned uint64 words, sorted as unsigned pairs; counts int16).
    key_range: one rank's shard of that table (see diploid_table): chunks outside the range are skipped, the chunks on
    its borders are cut, and the chunk size is taken from the rank's share -- a rank of eight generates its eighth of a
    table eight times the size with the scratch of one chunk."""
    assert 33 <= k <= 64
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    haps = graded_haplotypes(G, ploidy, rates, gen, device)
    m = G - k + 1
    z = torch.randn(m, device=device, generator=gen, dtype=torch.float32)
    SIGN = -(1 << 63)
    total = 2 * ploidy * m
    cbases = 0
    while total / (4 ** cbases) > max_chunk:
        cbases += 1
    cbases = min(cbases, 8)                     # (chunks are whole 16-bit ranges of the leading key bits)
    nchunk = 4 ** cbases
    strands = []
    for h in haps:
        strands.append((h, z))
        strands.append(((3 - h).flip(0), z.flip(0)))        # occurrence i of h mirrors occurrence m-1-i of rc(h)
    del haps

    def chunk_ids(h):
        cid = torch.zeros(m, dtype=torch.int32, device=device)
        for j in range(cbases):
            cid = (cid << 2) | h[j: m + j].to(torch.int32)
        return cid.to(torch.uint8) if cbases <= 4 else cid.to(torch.int16) if cbases <= 7 else cid

    def words_at(h, idx):
        w0 = torch.zeros(idx.numel(), dtype=torch.int64, device=device)
        for j in range(32):
            w0 <<= 2
            w0 |= h[idx + j].to(torch.int64)
        w1 = torch.zeros(idx.numel(), dtype=torch.int64, device=device)
        for j in range(32, k):
            w1 <<= 2
            w1 |= h[idx + j].to(torch.int64)
        w1 <<= 2 * (64 - k)
        return w0 ^ SIGN, w1 ^ SIGN            # biased: signed order == unsigned order

    cids = [chunk_ids(h) for h, _ in strands] if cbases else None
    K0, K1, CN = [], [], []
    for c in range(nchunk):
        if key_range is not None:               # chunk c holds the leading 16-bit values [c, c + 1) << (16 - 2 cbases)
            clo, chi = c << (16 - 2 * cbases), (c + 1) << (16 - 2 * cbases)
            if chi <= key_range[0] or clo >= key_range[1]:
                continue
            cut = clo < key_range[0] or chi > key_range[1]
        W0, W1, Z = [], [], []
        for s, (h, zs) in enumerate(strands):
            idx = torch.nonzero(cids[s] == c).reshape(-1) if cbases else torch.arange(m, device=device)
            a0, a1 = words_at(h, idx)
            zi = zs[idx]
            del idx
            if key_range is not None and cut:
                keep = _in_range16(((a0 ^ SIGN) >> 48) & 0xFFFF, key_range)
                a0, a1, zi = a0[keep], a1[keep], zi[keep]
                del keep
            W0.append(a0); W1.append(a1); Z.append(zi)
        w0 = torch.cat(W0); w1 = torch.cat(W1); zz = torch.cat(Z)
        del W0, W1, Z
        if w0.numel() == 0:
            continue
        # order by (w0, w1): one sort on w0, then odd-even transposition inside the (short) runs of equal w0
        w0, o = torch.sort(w0)
        w1, zz = w1[o], zz[o]
        del o
        n = w0.numel()
        for rnd in range(4 * ploidy + 64):
            swapped = 0
            for parity in (0, 1):
                n2 = (n - parity) // 2
                if n2 <= 0:
                    continue
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
        first = torch.ones(n, dtype=torch.bool, device=device)
        first[1:] = (w0[1:] != w0[:-1]) | (w1[1:] != w1[:-1])
        run = torch.cumsum(first, 0) - 1
        nrun = int(run[-1].item()) + 1
        zmin = torch.full((nrun,), float("inf"), device=device, dtype=torch.float32)
        zmin.scatter_reduce_(0, run, zz, reduce="amin")
        mult = torch.zeros(nrun, dtype=torch.int64, device=device)
        mult.scatter_add_(0, run, torch.ones_like(run))
        u0, u1 = w0[first] ^ SIGN, w1[first] ^ SIGN
        del w0, w1, zz, run, first
        mean = float(cov_hap) * mult.clamp(max=ploidy).to(torch.float32)
        cnt = torch.round(mean + torch.sqrt(mean) * zmin).clamp_(0, 32767).to(torch.int16)
        keep = cnt >= L                        # x and rc(x) carry the same count: trimming keeps the closure
        K0.append(u0[keep]); K1.append(u1[keep]); CN.append(cnt[keep])
        del u0, u1, cnt, keep, mean, mult, zmin
    del strands, cids, z
    ntot = sum(int(a.numel()) for a in K0)
    keys = torch.empty((ntot, 2), dtype=torch.int64, device=device)
    cnt = torch.empty(ntot, dtype=torch.int16, device=device)
    at = 0
    for a0, a1, cc in zip(K0, K1, CN):         # (column by column, piece by piece: see diploid_table_wide on torch.stack)
        e = at + a0.numel()
        keys[at:e, 0] = a0
        keys[at:e, 1] = a1
        cnt[at:e] = cc
        at = e
    return keys, cnt


def write_table_from_device(path: str, keys: torch.Tensor, cnt: torch.Tensor, k: int, nparts: int = 4) -> int:
    """Write a device-resident table (keys int64 [n] or [n, W] left aligned, counts int16) as a FastK table, format F
    with ibyte = 3 (libfastk.c:786-908): the records and the 2^24-entry prefix index are put together on the device,
    the host only copies and writes -- a 2.5e9-entry table takes its 18 GB of records and nothing else of host memory.
    Parts are cut on prefix boundaries.  Returns the bytes written."""
    import os
    import numpy as np
    kw = keys.reshape(cnt.numel(), -1)
    n, kb = cnt.numel(), (k + 3) // 4
    hb = kb - 3
    pre = (kw[:, 0] >> 40) & 0xFFFFFF
    index = torch.cumsum(torch.bincount(pre, minlength=1 << 24), 0)
    del pre
    cuts = [0]
    for p in range(1, nparts):
        b = int(torch.searchsorted(index, torch.tensor([(n * p) // nparts], device=index.device), right=False).item())
        cuts.append(max(int(index[min(b, (1 << 24) - 1)].item()), cuts[-1]))
    cuts.append(n)
    d, base = os.path.split(path)
    d = d or "."
    total = 0
    with open(os.path.join(d, base + ".ktab"), "wb") as f:
        np.array([k, nparts, 1, 3], dtype="<i4").tofile(f)
        index.cpu().numpy().astype("<i8").tofile(f)
        total += f.tell()
    del index
    c = cnt.to(torch.int32) & 0xFFFF
    for p in range(nparts):
        lo, hi = cuts[p], cuts[p + 1]
        with open(os.path.join(d, f".{base}.ktab.{p + 1}"), "wb") as f:
            np.array([k], dtype="<i4").tofile(f)
            np.array([hi - lo], dtype="<i8").tofile(f)
            step = 1 << 27                    # records of 2^27 entries at a time through the device
            for a in range(lo, hi, step):
                e = min(a + step, hi)
                rec = torch.empty((e - a, hb + 2), dtype=torch.uint8, device=keys.device)
                for j in range(hb):           # suffix bytes: bytes 3 .. kb-1 of the left-aligned k-mer
                    bj = 3 + j
                    rec[:, j] = ((kw[a:e, bj // 8] >> (8 * (7 - bj % 8))) & 0xFF).to(torch.uint8)
                rec[:, hb] = (c[a:e] & 0xFF).to(torch.uint8)
                rec[:, hb + 1] = (c[a:e] >> 8).to(torch.uint8)
                rec.cpu().numpy().tofile(f)
                del rec
            total += f.tell()
    return total


def table_hash(keys: torch.Tensor, cnt: torch.Tensor, first_entry: int = 0, piece: int = 1 << 26):
    """Order-sensitive 2 x 64-bit checksum of a table (or of the shard that starts at entry `first_entry` of it) as two
    Python ints mod 2^64: sums over the entries of a mixed k-mer word / count times an odd multiplier taken from the
    GLOBAL position, so the shards of a table add up (mod 2^64) to the hash of the whole and any change of a word, a
    count or the order shows.  bench.py prints it next to the golden .smu it compares with (tests/golden/bench_tables.json)."""
    n = cnt.numel()
    kf = keys.reshape(-1)
    words = kf.numel() // max(n, 1) if n else 1
    hk = torch.zeros((), dtype=torch.int64, device=cnt.device)
    hc = torch.zeros((), dtype=torch.int64, device=cnt.device)
    for a in range(0, kf.numel(), piece):
        v = kf[a: a + piece]
        pos = torch.arange(a, a + v.numel(), dtype=torch.int64, device=v.device) + first_entry * words
        hk += (_mix(v ^ 0x243F6A8885A308D3) * (2 * pos + 1)).sum()
        del v, pos
    for a in range(0, n, piece):
        c = cnt[a: a + piece].to(torch.int64) & 0xFFFF
        pos = torch.arange(a, a + c.numel(), dtype=torch.int64, device=c.device) + first_entry
        hc += (_mix(c + 0x13198A2E03707344) * (2 * pos + 1)).sum()
        del c, pos
    return int(hk.item()) & 0xFFFFFFFFFFFFFFFF, int(hc.item()) & 0xFFFFFFFFFFFFFFFF


def table_hash_text(n: int, hk: int, hc: int) -> str:
    return f"{n}:{hk:016x}:{hc:016x}"
