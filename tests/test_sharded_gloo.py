"""world_size-2 (and 3) runs of the prefix-sharded driver on CPU under `gloo`.

The collectives, splitter exchange and request routing of `sharded.hetmers_sharded` are the real
ones; only the per-shard compute is the numpy stand-in of tests/fake_engine.py (the HIP engine
needs a GPU and has no CPU fallback).  The summed plot must equal the oracle's on the whole table.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import brute
from smudgeplot_amd import ktab, sharded, synth

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, k, keys, cnt, cuts, symcheck, drop, q, fallback=True):
    sys.path.insert(0, HERE)
    from fake_engine import NumpyEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = cuts[rank], cuts[rank + 1]
        kk, cc = keys[lo:hi], cnt[lo:hi]
        if drop is not None and lo <= drop < hi:          # break the symmetry on one rank
            kk = np.delete(kk, drop - lo, axis=0); cc = np.delete(cc, drop - lo)
        tk = torch.from_numpy(np.ascontiguousarray(kk).view(np.int64).reshape(-1).copy())
        tc = torch.from_numpy(cc.view(np.int16).copy())
        try:
            plot, st = sharded.hetmers_sharded(k, tk, tc, symcheck=symcheck, engine_factory=NumpyEngine, fallback=fallback)
            q.put((rank, "ok" if st["path"] == 1 else "general", plot.numpy().copy(), st["sent"], st["received"]))
        except sharded.NotSymmetric:
            q.put((rank, "notsym", None, 0, 0))
    finally:
        dist.destroy_process_group()


def _run(world, k, keys, cnt, cuts, symcheck, drop=None, fallback=True):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, k, keys, cnt, cuts, symcheck, drop, q, fallback))
             for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(out, key=lambda t: t[0])


@pytest.mark.parametrize("world,symcheck,k", [(2, "hash", 31), (2, "exact", 31), (3, "hash", 24), (2, "hash", 12),
                                                (3, "hash", 13)])
def test_prefix_sharded_matches_oracle(world, symcheck, k):
    keys, cnt = synth.diploid_table_u64(4000, k=k, seed=40 + world, het_frac=0.4, cov=30, L=5)
    want = brute.hetmers_plot(ktab.u64_to_packed(keys, k), cnt, k)
    assert want.sum() > 0
    n = len(cnt)
    cuts = [sharded.fix_cut(keys, 1, k, c) for c in sharded.shard_bounds(n, world)]
    res = _run(world, k, keys, cnt, cuts, symcheck)
    for rank, status, plot, sent, received in res:
        assert status == "ok"
        assert np.array_equal(plot.reshape(1001, 501), want), f"rank {rank}"     # all_reduce: same on all
    assert sum(r[3] for r in res) == sum(r[4] for r in res)                      # every request delivered
    if symcheck == "exact" or k <= 16:
        # (hash proof: the request filter drops requests whose target block holds no candidate -- on a sparse
        #  k = 31 table that can be all of them; the short k-mers fill their 2^(2*(k/2)) blocks, so requests survive)
        assert sum(r[3] for r in res) > 0


def test_empty_shard_and_uneven_cuts():
    k = 31
    keys, cnt = synth.diploid_table_u64(1500, k=k, seed=77, het_frac=0.5, cov=30, L=5)
    want = brute.hetmers_plot(ktab.u64_to_packed(keys, k), cnt, k)
    n = len(cnt)
    cuts = [0, 0, sharded.fix_cut(keys, 1, k, n // 3), n]          # rank 0 owns nothing
    res = _run(3, k, keys, cnt, cuts, "hash")
    for _, status, plot, _, _ in res:
        assert status == "ok" and np.array_equal(plot.reshape(1001, 501), want)


def test_asymmetric_table_falls_back_to_the_general_path_on_rank_0():
    """a table that is not closed under reverse complement: every rank sees the failed proof (it is reduced over the
    ranks), rank 0 collects the shards and runs the general path, every rank gets the reference's answer"""
    k = 31
    keys, cnt = synth.diploid_table_u64(1500, k=k, seed=78, het_frac=0.5, cov=30, L=5)
    n = len(cnt)
    cuts = [sharded.fix_cut(keys, 1, k, c) for c in sharded.shard_bounds(n, 2)]
    # drop a k-mer that is not its own complement: its partner's request / fingerprint is orphaned
    rc = ktab.revcomp_u64(keys, k)
    drop = int(np.nonzero(rc != keys)[0][5])
    kept = np.ones(n, bool); kept[drop] = False
    want = brute.hetmers_plot(ktab.u64_to_packed(keys[kept], k), cnt[kept], k)
    for symcheck in ("hash", "exact"):
        res = _run(2, k, keys, cnt, cuts, symcheck, drop=drop)
        assert [r[1] for r in res] == ["general", "general"]
        for r in res:
            assert np.array_equal(r[2].reshape(1001, 501), want)
        res = _run(2, k, keys, cnt, cuts, symcheck, drop=drop, fallback=False)      # the strict mode still exists
        assert [r[1] for r in res] == ["notsym", "notsym"]


def _wide_table(k, m, seed):
    """(keys [n, W] uint64 left aligned, counts) of a closed adversarial table with k > 32"""
    packed, cnt = synth.adversarial_table(k, m, 4, seed, low_complexity=10, dense=1)
    W = (k + 31) // 32
    buf = np.zeros((len(cnt), 8 * W), dtype=np.uint8)
    buf[:, : packed.shape[1]] = packed
    return packed, np.ascontiguousarray(buf.view(">u8").astype(np.uint64)), cnt


@pytest.mark.parametrize("world,k,symcheck", [(2, 40, "hash"), (3, 51, "hash"), (2, 51, "exact"), (2, 70, "hash")])
def test_two_and_three_word_kmers_through_the_sharded_driver(world, k, symcheck):
    """BASELINE configs[4] is k = 51: records of W + 1 words, splitters of W words, block ids from word 0"""
    packed, keys, cnt = _wide_table(k, 700, 90 + k)
    W = keys.shape[1]
    want = brute.hetmers_plot(packed, cnt, k)
    assert want.sum() > 0
    n = len(cnt)
    cuts = [sharded.fix_cut(keys.reshape(-1), W, k, c) for c in sharded.shard_bounds(n, world)]
    res = _run(world, k, keys, cnt, cuts, symcheck)
    for rank, status, plot, sent, received in res:
        assert status == "ok"
        assert np.array_equal(plot.reshape(1001, 501), want), f"rank {rank}"
    assert sum(r[3] for r in res) == sum(r[4] for r in res)
