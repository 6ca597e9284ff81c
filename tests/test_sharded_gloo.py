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


def _worker(rank, world, port, k, keys, cnt, cuts, symcheck, drop, q, fallback=True, own_splitters=False):
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
        split = None
        if own_splitters:      # the caller knows the cut k-mers (first k-mer of ranks 1..world-1 of the UNDAMAGED table)
            split = np.concatenate([np.ascontiguousarray(keys[cuts[r]]).reshape(-1) for r in range(1, world)]).astype(np.uint64)
        try:
            plot, st = sharded.hetmers_sharded(k, tk, tc, symcheck=symcheck, engine_factory=NumpyEngine, fallback=fallback,
                                               splitters=split)
            q.put((rank, "ok" if st["path"] == 1 else "general", plot.numpy().copy(), st["sent"], st["received"]))
        except sharded.NotSymmetric:
            q.put((rank, "notsym", None, 0, 0))
    finally:
        dist.destroy_process_group()


def _run(world, k, keys, cnt, cuts, symcheck, drop=None, fallback=True, own_splitters=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, k, keys, cnt, cuts, symcheck, drop, q, fallback, own_splitters))
             for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(out, key=lambda t: t[0])


@pytest.mark.parametrize("world,symcheck,k", [(2, "hash", 31), (2, "exact", 31), (3, "hash", 24), (2, "hash", 12),
                                                (3, "hash", 13), (8, "hash", 31), (8, "hash", 16)])
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


def test_callers_splitters_with_a_table_that_fails_the_proof():
    """splitters handed in by the caller (no shard sizes with them) + a table that is not closed: the general path on
    rank 0 needs the sizes of the other shards -- it used to raise on rank 0 while rank 1 sat in its send"""
    k = 31
    keys, cnt = synth.diploid_table_u64(1500, k=k, seed=79, het_frac=0.5, cov=30, L=5)
    n = len(cnt)
    cuts = [sharded.fix_cut(keys, 1, k, c) for c in sharded.shard_bounds(n, 2)]
    rc = ktab.revcomp_u64(keys, k)
    cand = np.nonzero(rc != keys)[0]
    drop = int(cand[(cand != cuts[1])][7])                  # (not the cut k-mer itself: it is a splitter)
    kept = np.ones(n, bool); kept[drop] = False
    want = brute.hetmers_plot(ktab.u64_to_packed(keys[kept], k), cnt[kept], k)
    res = _run(2, k, keys, cnt, cuts, "hash", drop=drop, own_splitters=True)
    assert [r[1] for r in res] == ["general", "general"]
    for r in res:
        assert np.array_equal(r[2].reshape(1001, 501), want)
    good = brute.hetmers_plot(ktab.u64_to_packed(keys, k), cnt, k)
    res = _run(2, k, keys, cnt, cuts, "hash", own_splitters=True)          # ... and the closed table with them
    assert [r[1] for r in res] == ["ok", "ok"] and np.array_equal(res[0][2].reshape(1001, 501), good)


def _wide_table(k, m, seed):
    """(keys [n, W] uint64 left aligned, counts) of a closed adversarial table with k > 32"""
    packed, cnt = synth.adversarial_table(k, m, 4, seed, low_complexity=10, dense=1)
    W = (k + 31) // 32
    buf = np.zeros((len(cnt), 8 * W), dtype=np.uint8)
    buf[:, : packed.shape[1]] = packed
    return packed, np.ascontiguousarray(buf.view(">u8").astype(np.uint64)), cnt


@pytest.mark.parametrize("world,k,symcheck", [(2, 40, "hash"), (3, 51, "hash"), (2, 51, "exact"), (2, 70, "hash"),
                                                 (2, 100, "hash"), (3, 128, "exact")])
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


# ---- conditioning across shards: raw canonical table in, every rank ends up with its range of the closed table ----

def _cond_worker(rank, world, port, k, keys, cnt, cuts, L, q):
    sys.path.insert(0, HERE)
    from fake_engine import NumpyEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = cuts[rank], cuts[rank + 1]
        tk = torch.from_numpy(np.ascontiguousarray(keys[lo:hi]).view(np.int64).reshape(-1).copy())
        tc = torch.from_numpy(cnt[lo:hi].view(np.int16).copy())
        eng, split = sharded.condition_sharded(k, tk, tc, ethresh=L, trim=True, symm=True, engine_factory=NumpyEngine)
        shard = [int(x) for x in eng.keys]
        plot, st = sharded.hetmers_sharded(k, None, None, symcheck="hash", eng=eng, splitters=split)
        q.put((rank, "ok" if st["path"] == 1 else "general", plot.numpy().copy(), shard, [int(c) for c in eng.cnt]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k", [(2, 31), (3, 31), (2, 21), (2, 51)])
def test_raw_table_is_conditioned_across_the_ranks(world, k):
    """PloidyPlot.c:1381-1414 hands a raw table of any size to Logex + Symmex.  Sharded: every rank trims its piece of
    the RAW canonical table (cut anywhere), sends entries and complements to the rank that owns them, sorts what it
    receives.  The union of the shards must be the numpy-conditioned table, entry for entry, and the plot the oracle's
    on that table."""
    L = 5
    packed, cnt = synth.adversarial_table(k, 600, L, seed=300 + k + world, low_complexity=30, dense=1)
    rc = ktab.revcomp_packed(packed, k)
    canon = np.array([bytes(a) <= bytes(b) for a, b in zip(packed, rc)])
    rp, rcnt = packed[canon], cnt[canon].copy()
    rng = np.random.default_rng(k)
    low = rng.random(len(rcnt)) < 0.2
    rcnt[low] = rng.integers(1, L, size=int(low.sum()))
    keep = rcnt >= L
    cp, cc = ktab.symmetrize(rp[keep], rcnt[keep], k)
    want = brute.hetmers_plot(cp, cc, k)
    assert want.sum() > 0
    W = (k + 31) // 32
    buf = np.zeros((len(rcnt), 8 * W), dtype=np.uint8)
    buf[:, : rp.shape[1]] = rp
    keys = np.ascontiguousarray(buf.view(">u8").astype(np.uint64))
    n = len(rcnt)
    cuts = [(n * r) // world for r in range(world + 1)]                  # raw pieces: any cut will do
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cond_worker, args=(r, world, port, k, keys, rcnt, cuts, L, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, status, plot, _, _ in out:
        assert status == "ok"
        assert np.array_equal(plot.reshape(1001, 501), want), f"rank {rank}"
    # the shards, in rank order, ARE the conditioned table
    cbuf = np.zeros((len(cc), 8 * W), dtype=np.uint8)
    cbuf[:, : cp.shape[1]] = cp
    ck = np.ascontiguousarray(cbuf.view(">u8").astype(np.uint64))
    want_keys = [int.from_bytes(b"".join(int(w).to_bytes(8, "big") for w in row), "big") for row in ck]
    got_keys = [x for o in out for x in o[3]]
    got_cnt = [c for o in out for c in o[4]]
    assert got_keys == want_keys and got_cnt == [int(c) for c in cc]
    assert min(len(o[3]) for o in out) > 0.25 * len(cc) / world          # balanced splitters: nobody is left (nearly) empty


def test_symm_splitters_follow_the_closed_table():
    """a canonical table crowds the low end of the k-mer space: splitters taken from entries + complements do not"""
    bits = 6
    own = np.zeros(1 << bits, dtype=np.int64); own[:16] = 100          # every entry starts with an a
    rcs = np.zeros(1 << bits, dtype=np.int64); rcs[48:] = 100          # ... so every complement starts with a t
    sp = sharded.symm_splitters(np.concatenate([own, rcs]), bits, 4, 1)
    assert [int(v) >> (64 - bits) for v in sp] == [8, 16, 56]         # (bins 16..47 are empty: 16 and 48 cut the same place)
    assert list(sharded.symm_splitters(np.concatenate([own, rcs * 0]), bits, 2, 1)) == [np.uint64(8) << np.uint64(58)]


# ---- replayed steps (round 5): a step on the table of the step before is queued from recorded counts ------------------
def _replay_worker(rank, world, port, k, keys, cnt, cuts, edits, q):
    sys.path.insert(0, HERE)
    from fake_engine import NumpyEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SMG_REPLAY"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = cuts[rank], cuts[rank + 1]
        tk = torch.from_numpy(np.ascontiguousarray(keys[lo:hi]).view(np.int64).reshape(-1).copy())
        tc = torch.from_numpy(cnt[lo:hi].view(np.int16).copy())
        eng = NumpyEngine(torch.device("cpu"))
        eng.bind(k, tk, tc)
        out = []
        for step, edit in enumerate(edits):
            # edit = None | ("count", entry, value): the table changes IN PLACE between two steps (same tensors)
            if edit is not None:
                for e_, v in edit[1:]:
                    if lo <= e_ < hi:
                        tc[e_ - lo] = v
                eng.bind(k, tk, tc)                       # (the stand-in copies the table at bind; the engine reads it in place)
                eng._rp_want = True                       # ... and must keep its record, as an engine that was not re-bound would
            plot, st = sharded.hetmers_sharded(k, tk, tc, symcheck="hash", engine_factory=NumpyEngine, eng=eng, prebound=True)
            out.append((st["path"], bool(st.get("replayed")), plot.numpy().copy()))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k", [(2, 12), (3, 13)])
def test_replayed_steps_and_a_table_that_changes_under_them(world, k):
    """Step 1 runs the plain way and leaves records (the engine's counts, the driver's send / receive counts); step 2 on the
    same table is queued from them and must give the same plot; before step 3 counts change in place so that pairs drop out
    (another number of requests on one rank): the replayed step reports the mismatch, EVERY rank runs the step again the
    plain way, and the answer is the oracle's for the changed table; step 4 replays again; before step 5 one count of a
    (k-mer, complement) couple changes alone: the table is not closed any more, the replayed step is settled by the plain
    path and the general path on rank 0 answers."""
    keys, cnt = synth.diploid_table_u64(3000, k=k, seed=11 + world, het_frac=0.5, cov=30, L=5)
    n = len(cnt)
    cuts = [sharded.fix_cut(keys, 1, k, c) for c in sharded.shard_bounds(n, world)]
    rc = ktab.revcomp_u64(keys, k)
    pos = {int(x): i for i, x in enumerate(keys)}
    # couples (x, rc x), x != rc x: set both counts to 900 -> every pair of either exceeds the sum limit; every 7th couple of
    # the table, so that requests certainly drop out
    cnt3 = cnt.copy()
    changed = []
    for i in range(0, n, 7):
        if int(rc[i]) != int(keys[i]):
            j = pos[int(rc[i])]
            cnt3[i] = 900; cnt3[j] = 900
            changed += [(i, 900), (j, 900)]
    i = changed[0][0]
    cnt5 = cnt3.copy(); cnt5[i] = 901                                     # (not closed: the counts of the couple differ)
    want1 = brute.hetmers_plot(ktab.u64_to_packed(keys, k), cnt, k)
    want3 = brute.hetmers_plot(ktab.u64_to_packed(keys, k), cnt3, k)
    want5 = brute.hetmers_plot(ktab.u64_to_packed(keys, k), cnt5, k)
    edits = [None, None, ("count", *changed), None, ("count", (i, 901))]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replay_worker, args=(r, world, port, k, keys, cnt, cuts, edits, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        steps = res[rank]
        assert [s[0] for s in steps] == [1, 1, 1, 1, 2], rank
        for s, want in zip(steps, [want1, want1, want3, want3, want5]):
            assert np.array_equal(s[2].reshape(1001, 501), want), rank
        assert steps[0][1] is False and steps[1][1] is True and steps[3][1] is True
        assert steps[2][1] is False and steps[4][1] is False           # (settled by the plain path)


# ---- shards cut on key-space boundaries (what bench.py --gpus N generates): cut values as splitters, the gathered map ranges ARE the map
def _cut_worker(rank, world, port, k, keys, cnt, q):
    sys.path.insert(0, HERE)
    from fake_engine import NumpyEngine
    from smudgeplot_amd import synth_device
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lead = (keys >> np.uint64(48)).astype(np.int64)
        cuts = [int(np.searchsorted(lead, synth_device.key_range_of(r, world)[0], side="left")) for r in range(world)] + [len(cnt)]
        lo, hi = cuts[rank], cuts[rank + 1]
        tk = torch.from_numpy(np.ascontiguousarray(keys[lo:hi]).view(np.int64).copy())
        tc = torch.from_numpy(cnt[lo:hi].view(np.int16).copy())
        split = np.array([np.uint64(synth_device.key_range_of(r, world)[0]) << np.uint64(48) for r in range(1, world)], dtype=np.uint64)
        sizes = [cuts[r + 1] - cuts[r] for r in range(world)]
        plot, st = sharded.hetmers_sharded(k, tk, tc, symcheck="hash", engine_factory=NumpyEngine, splitters=split, sizes=sizes)
        q.put((rank, st["path"], plot.numpy().copy(), st["sent"], st["received"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k", [(2, 12), (4, 13), (3, 12)])
def test_key_space_cuts_with_cut_values_as_splitters(world, k):
    keys, cnt = synth.diploid_table_u64(3000, k=k, seed=5 + world, het_frac=0.5, cov=30, L=5)
    want = brute.hetmers_plot(ktab.u64_to_packed(keys, k), cnt, k)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cut_worker, args=(r, world, port, k, keys, cnt, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, path, plot, sent, received in res:
        assert path == 1 and np.array_equal(plot.reshape(1001, 501), want), rank
    assert sum(r[3] for r in res) == sum(r[4] for r in res) > 0


def test_blockmap_ranges_of_cut_values_and_of_first_kmers():
    from smudgeplot_amd import synth_device
    bits, scale = 29, 2
    nwords = (((1 << bits) + 31) >> 5) * scale
    for world in (2, 4, 8):
        sp = np.array([np.uint64(synth_device.key_range_of(r, world)[0]) << np.uint64(48) for r in range(1, world)], dtype=np.uint64)
        wlo, wlen = sharded.blockmap_ranges(sp, 1, world, bits, scale)
        assert sharded.ranges_tile_the_map(wlo, wlen, nwords)
    # first k-mers of the ranks (something below the id bits): neighbours share their boundary word group, as ever
    sp = np.array([0x4000000012345678, 0x8000000000000001], dtype=np.uint64)
    wlo, wlen = sharded.blockmap_ranges(sp, 1, 3, bits, scale)
    assert wlo[1] + wlen[1] == wlo[2] + scale and wlo[0] + wlen[0] == wlo[1] + scale
    assert not sharded.ranges_tile_the_map(wlo, wlen, nwords)
    # two-word k-mers: a second word that is not zero makes the cut an ordinary k-mer
    sp = np.array([0x4000000000000000, 5], dtype=np.uint64)
    wlo, wlen = sharded.blockmap_ranges(sp, 2, 2, bits, scale)
    assert wlo[0] + wlen[0] == wlo[1] + scale


# ---- bench.py --gpus N on CPU: make_shard (per-rank generation, summed prefix index, summed checksum, cut-value splitters) and the
#      sharded step on what it returns -- everything of the N > 1 bench path but the HIP engine
def _bench_shard_worker(rank, world, port, workload, G, k, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    from fake_engine import NumpyEngine
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        sh = bench.make_shard(workload, G, k, dev, rank, world)
        keys, cnt = sh["keys"].reshape(-1), sh["counts"]
        plot, st = sharded.hetmers_sharded(k, keys, cnt, symcheck="hash", engine_factory=NumpyEngine, splitters=sh["splitters"],
                                           sizes=sh["sizes"])
        par = bench.parity_against_golden(workload, G, k, sh["n_total"], sh["hk"], sh["hc"], plot)
        q.put((rank, sh["keys"].numpy().copy(), cnt.numpy().copy(), sh["sizes"], sh["first_entry"], sh["index"].numpy().copy(),
               sh["hk"], sh["hc"], st["path"], plot.numpy().copy(), par["ok"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,workload,G,k", [(2, "uniform", 3000, 16), (3, "octoploid", 700, 17), (2, "hexaploid", 500, 35)])
def test_bench_shards_and_the_sharded_step_on_them(world, workload, G, k):
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    from smudgeplot_amd import synth_device
    dev = torch.device("cpu")
    k0, c0, L, _ = bench.make_table(workload, G, k, dev)
    W = (k + 31) // 32
    k0 = k0.reshape(c0.numel(), W)
    keys_u = k0.numpy().view(np.uint64)
    packed = np.ascontiguousarray(keys_u.astype(">u8")).view(np.uint8).reshape(len(keys_u), 8 * W)[:, : (k + 3) // 4]
    want = brute.hetmers_plot(np.ascontiguousarray(packed), c0.numpy().view(np.uint16), k)
    assert want.sum() > 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_shard_worker, args=(r, world, port, workload, G, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole_index = torch.cumsum(torch.bincount((k0[:, 0] >> 40) & 0xFFFFFF, minlength=1 << 24), 0).numpy()
    hk, hc = synth_device.table_hash(k0, c0)
    first = 0
    for rank, kk, cc, sizes, fe, index, rhk, rhc, path, plot, par_ok in res:
        assert fe == first and sizes == [len(r[2]) for r in res]
        assert np.array_equal(kk.reshape(-1, W), k0.numpy()[first: first + len(cc)]) and np.array_equal(cc, c0.numpy()[first: first + len(cc)])
        assert np.array_equal(index, whole_index)                 # the WHOLE table's prefix index on every rank
        assert (rhk, rhc) == (hk, hc)                             # ... and its checksum
        assert path == 1 and np.array_equal(plot.reshape(1001, 501), want), rank
        assert par_ok is None                                     # (no golden at this size: the block says so, it does not fail)
        first += len(cc)
    assert first == c0.numel()
