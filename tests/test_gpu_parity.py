"""GPU parity tests (run with -m gpu on an MI355X): the HIP engine, called through the C ABI,
against the reference's golden vectors and the CPU oracle.  Integer work: the bar is bit-exact."""
import os
import subprocess

import numpy as np
import pytest

import brute
from conftest import HETMERS_BIN, ORACLE_BIN, golden_names, load_golden, make_table
from smudgeplot_amd import engine, ktab, synth

pytestmark = pytest.mark.gpu


def table_from(packed, cnt, k, ibyte=1, nparts=1):
    return make_table(dict(packed=packed, counts=cnt, k=k, ibyte=ibyte, nparts=nparts))


@pytest.mark.parametrize("mode", ["exact", "hash", "none"])
@pytest.mark.parametrize("name", golden_names())
def test_engine_matches_reference_golden(name, mode):
    g = load_golden(name)
    plot, st = engine.hetmers_run(make_table(g), symcheck=mode)
    assert engine.smu_text(plot) == g["smu"]
    assert st["path"] == (2 if mode == "none" else 1)
    assert st["nels"] == len(g["counts"])


@pytest.mark.parametrize("name", ["k31_i1", "k32_i1_p2", "k21_i2_p2", "k51_i1_p3", "k100_wrap"])
def test_hetmers_executable_is_byte_identical(name, tmp_path):
    """the drop-in binary: same argv as the CLI passes, .smu compared byte for byte"""
    g = load_golden(name)
    ktab.write_ktab(str(tmp_path / "t"), g["k"], g["packed"], g["counts"], ibyte=g["ibyte"],
                    nparts=g["nparts"])
    r = subprocess.run([HETMERS_BIN, "-okmerpairs", f"-e{g['L']}", "-T4", "-v", "t.ktab"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "  The input table is trimmed and symmetric\n" in r.stderr
    assert "  Count complete, outputting table\n" in r.stderr
    assert (tmp_path / "kmerpairs.smu").read_text() == g["smu"]
    # and the C oracle agrees on the same files
    subprocess.run([ORACLE_BIN, f"-e{g['L']}", f"-o{tmp_path}/orc", str(tmp_path / "t")], check=True)
    assert (tmp_path / "orc.smu").read_text() == g["smu"]


@pytest.mark.parametrize("k,seed", [(19, 1), (27, 2), (31, 3), (32, 4), (35, 5), (47, 6), (63, 7),
                                    (66, 8), (96, 9), (97, 10), (128, 11)])
def test_fresh_tables_vs_oracle(k, seed):
    packed, cnt = synth.adversarial_table(k, 2500, 4, seed, low_complexity=150, dense=2)
    want = brute.hetmers_plot(packed, cnt, k)
    for mode in ("exact", "hash", "none"):
        plot, _ = engine.hetmers_run(table_from(packed, cnt, k), symcheck=mode)
        assert np.array_equal(plot, want), (k, mode)


def test_big_window_blocks():
    """thousands of entries sharing their first k/2 bases: the binary-search branch"""
    k = 31
    rng = np.random.default_rng(5)
    lc = np.zeros((6000, k), np.uint8)
    lc[:, 20:] = rng.integers(0, 4, size=(6000, 11), dtype=np.uint8)
    lc[:3000, 0] = 1
    packed = ktab.pack_bases(lc)
    cnt = rng.integers(5, 60, size=len(packed)).astype(np.uint16)
    packed, cnt = ktab.sort_unique_packed(packed, cnt)
    packed, cnt = ktab.symmetrize(packed, cnt, k)
    want = brute.hetmers_plot(packed, cnt, k)
    assert want.sum() > 0
    for mode in ("exact", "hash", "none"):
        plot, _ = engine.hetmers_run(table_from(packed, cnt, k), symcheck=mode)
        assert np.array_equal(plot, want), mode


def _long_block_families(k, seed):
    """families of 120 .. 2600 k-mers that share their first k/2 + 1 bases (one window block each, in buckets of their
    own): dense ones (a k-mer, all its single mutants behind the shared part, double mutants) and sparse ones (random
    tails, a few hundred of them with exactly one partner -- often far away in the block), some counts beyond the
    sum limit, a random background"""
    rng = np.random.default_rng(seed)
    share = k // 2 + 1
    rows = []
    for f, size in enumerate([120, 400, 1000, 1800, 2600, 150, 700, 1500, 2000, 2300]):
        base = rng.integers(0, 4, k, dtype=np.uint8)
        base[0], base[1] = f & 3, f >> 2                       # (a leading 2-mer of its own: never two families in a bucket)
        fam = np.tile(base, (size, 1))
        if f < 5:                                              # dense
            j = 1
            for p in range(share, k):
                for d in (1, 2, 3):
                    if j < size:
                        fam[j, p] = (base[p] + d) & 3; j += 1
            while j < size:
                p, q = rng.integers(share, k, 2)
                fam[j, p] = (base[p] + rng.integers(1, 4)) & 3
                fam[j, q] = (base[q] + rng.integers(1, 4)) & 3
                j += 1
        else:                                                  # sparse
            fam[:, share:] = rng.integers(0, 4, (size, k - share), dtype=np.uint8)
            for j in range(0, min(size - 1, 600), 2):
                fam[j + 1] = fam[j]
                p = rng.integers(share, k)
                fam[j + 1, p] = (fam[j, p] + rng.integers(1, 4)) & 3
        rows.append(fam)
    rows.append(rng.integers(0, 4, (3000, k), dtype=np.uint8))
    packed = ktab.pack_bases(np.concatenate(rows))
    cnt = rng.integers(5, 60, size=len(packed)).astype(np.uint16)
    cnt[rng.random(len(cnt)) < 0.04] = 700                     # (pairs of two such counts exceed the sum limit)
    packed, cnt = ktab.sort_unique_packed(packed, cnt)
    return ktab.symmetrize(packed, cnt, k)


@pytest.mark.parametrize("k", [22, 24, 31, 32, 40, 51, 64])
def test_long_window_blocks_dense_and_sparse_families(k):
    """kf_bigfix on window blocks of 120 .. 2600 entries -- dense families (most members own several pairs) and sparse
    ones (unique partners hundreds of entries away: the far code and kf_pass2_far) -- with the directory pass 1 builds
    (coarse on a small table) and with the table's 24-bit prefix index as directory (ibyte = 3: one bucket per block at
    k = 24, finer than the blocks at k = 22), one- and two-word k-mers, counts beyond the sum limit -- against the
    numpy oracle.  (Written for the round-4 attempt to solve such blocks from an LDS copy of their directory bucket,
    profiles/r04_pass1_experiments.txt; it holds for any kf_bigfix.)"""
    packed, cnt = _long_block_families(k, 300 + k)
    want = brute.hetmers_plot(packed, cnt, k)
    assert want.sum() > 500
    for ibyte in (1, 3):
        for mode in ("hash", "exact"):
            plot, st = engine.hetmers_run(table_from(packed, cnt, k, ibyte=ibyte), symcheck=mode)
            assert st["path"] == 1 and st["nbig"] > 5000, st
            assert np.array_equal(plot, want), (k, ibyte, mode)


def test_asymmetric_tables_fall_back_to_the_general_path():
    """tables that are NOT reverse-complement closed (the reference only probes entry #1)"""
    k = 31
    packed, cnt = synth.adversarial_table(k, 3000, 4, seed=21, low_complexity=100, dense=2)
    rng = np.random.default_rng(1)
    # (a) drop 5% of the entries at random, keep entry #1 and its complement
    keep = rng.random(len(cnt)) > 0.05
    rc1 = ktab.revcomp_packed(packed[1:2], k)[0]
    keep[1] = True
    keep[(packed == rc1).all(axis=1)] = True
    pa, ca = packed[keep], cnt[keep]
    want = brute.hetmers_plot(pa, ca, k)
    for mode in ("exact", "hash"):
        plot, st = engine.hetmers_run(table_from(pa, ca, k), symcheck=mode)
        assert st["path"] == 2, "asymmetry must be detected"
        assert np.array_equal(plot, want), mode
    # (b) all k-mers present but ONE count differs between a k-mer and its complement
    cb = cnt.copy()
    j = int(np.nonzero((packed != ktab.revcomp_packed(packed, k)).any(axis=1))[0][7])
    cb[j] += 1
    want = brute.hetmers_plot(packed, cb, k)
    for mode in ("exact", "hash"):
        plot, st = engine.hetmers_run(table_from(packed, cb, k), symcheck=mode)
        assert st["path"] == 2
        assert np.array_equal(plot, want), mode


def test_canonical_only_table_general_path():
    """half of every complement pair missing (a raw, unsymmetrised FastK table)"""
    k = 25
    packed, cnt = synth.adversarial_table(k, 2000, 4, seed=33)
    rc = ktab.revcomp_packed(packed, k)
    canon = np.array([bytes(a) <= bytes(b) for a, b in zip(packed, rc)])
    pa, ca = packed[canon], cnt[canon]
    want = brute.hetmers_plot(pa, ca, k)
    plot, st = engine.hetmers_run(table_from(pa, ca, k), symcheck="exact")
    assert st["path"] == 2 and np.array_equal(plot, want)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 5])
def test_empty_and_tiny_tables(n):
    k = 31
    rng = np.random.default_rng(n)
    bases = rng.integers(0, 4, size=(n, k), dtype=np.uint8)
    packed = ktab.pack_bases(bases)
    cnt = np.full(len(packed), 7, np.uint16)
    packed, cnt = ktab.sort_unique_packed(packed, cnt)
    packed, cnt = ktab.symmetrize(packed, cnt, k) if n else (packed, cnt)
    plot, st = engine.hetmers_run(table_from(packed, cnt, k), symcheck="exact")
    assert np.array_equal(plot, brute.hetmers_plot(packed, cnt, k))


def test_unsorted_table_is_rejected():
    k = 31
    packed, cnt = synth.adversarial_table(k, 300, 4, seed=8)
    packed = packed.copy()
    packed[[10, 11]] = packed[[11, 10]]
    t = table_from(packed, cnt, k)
    with pytest.raises(engine.EngineError) as ei:
        engine.hetmers_run(t)
    assert ei.value.code == -4


def test_medium_table_vs_c_oracle(tmp_path):
    """4e5 entries, multi-part, ibyte 2: engine == C oracle == properties of the plot"""
    k = 31
    keys, cnt = synth.diploid_table_u64(150000, k=k, seed=5, het_frac=0.3, cov=40, L=8)
    packed = ktab.u64_to_packed(keys, k)
    ktab.write_ktab(str(tmp_path / "t"), k, packed, cnt, ibyte=2, nparts=3)
    subprocess.run([ORACLE_BIN, "-e8", f"-o{tmp_path}/orc", str(tmp_path / "t")], check=True)
    r = subprocess.run([HETMERS_BIN, "-e8", "-T4", "-ogpu", "t"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "gpu.smu").read_text() == (tmp_path / "orc.smu").read_text()
    env = dict(os.environ, SMUDGEPLOT_SYMCHECK="hash")
    r = subprocess.run([HETMERS_BIN, "-e8", "-ogpuh", "t"], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "gpuh.smu").read_text() == (tmp_path / "orc.smu").read_text()


def test_large_table_sorted_lookup_path_vs_reference_binary(tmp_path, monkeypatch):
    """~7.6e6 entries: the request list is long enough for the radix-sorted look-up path; compare the
    engine with the REFERENCE binary (oracle/_ref, prebuilt) on the same table, and let the engine
    self-check every sort (SMG_VERIFY_SORT)."""
    import torch
    from conftest import REF_BIN
    from smudgeplot_amd import synth_device
    if not os.path.exists(REF_BIN):
        pytest.skip("prebuilt reference binary not present")
    monkeypatch.setenv("SMG_VERIFY_SORT", "1")
    k = 31
    dev = torch.device("cuda:0")
    tk, tc = synth_device.diploid_table(3_000_000, k=k, het=0.01, cov=50.0, L=10, seed=3, device=dev)
    keys = tk.cpu().numpy().view(np.uint64)
    cnt = tc.cpu().numpy().view(np.uint16)
    plot = torch.zeros(engine.PLOT_CELLS, dtype=torch.int64, device=dev)
    e = engine.Engine(0)
    e.bind(k, len(cnt), tk.data_ptr(), tc.data_ptr())
    texts = {}
    for mode in ("hash", "exact", "hash-unfiltered"):
        if mode == "hash-unfiltered":            # (on a table this sparse the request filter leaves too few
            monkeypatch.setenv("SMG_NO_FILTER", "1")     #  requests for the sorted path: run it without as well)
        st = e.run(plot.data_ptr(), mode.split("-")[0])
        torch.cuda.synchronize()
        assert st["path"] == 1 and st["nemitted"] > 100000
        assert mode == "hash" or st["nrequests"] > 100000
        texts[mode] = engine.smu_text(plot.cpu().numpy().reshape(1001, 501))
    assert texts["hash"] == texts["exact"] == texts["hash-unfiltered"]
    synth.write_u64_table(str(tmp_path / "t"), keys, cnt, k, ibyte=2, nparts=3)
    r = subprocess.run([REF_BIN, "-e10", f"-T{min(16, os.cpu_count() or 1)}", "-oref", "t.ktab"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "ref.smu").read_text() == texts["hash"]


def test_repeat_rich_table_vs_reference_binary(tmp_path, monkeypatch):
    """5 % of the genome are dispersed copies, tandem arrays and homopolymer runs (bench.py --workload repeats): window
    blocks of hundreds to thousands of entries, so the deferred entries (kf_collect / kf_bigfix), the prefix-narrowing
    block walk and the far partners of pass 2 (kf_pass2_far) all carry real weight -- against the REFERENCE binary,
    on one shard and on three, and the table generator must give the same table twice"""
    import torch
    from conftest import REF_BIN
    from smudgeplot_amd import synth_device
    if not os.path.exists(REF_BIN):
        pytest.skip("prebuilt reference binary not present")
    k = 31
    dev = torch.device("cuda:0")
    tk, tc = synth_device.diploid_table(4_000_000, k=k, het=0.01, cov=50.0, L=10, seed=11, device=dev, repeats=0.05)
    tk2, tc2 = synth_device.diploid_table(4_000_000, k=k, het=0.01, cov=50.0, L=10, seed=11, device=dev, repeats=0.05)
    assert torch.equal(tk, tk2) and torch.equal(tc, tc2), "the repeats generator is not deterministic"
    keys = tk.cpu().numpy().view(np.uint64)
    cnt = tc.cpu().numpy().view(np.uint16)
    synth.write_u64_table(str(tmp_path / "t"), keys, cnt, k, ibyte=2, nparts=3)
    r = subprocess.run([REF_BIN, "-e10", f"-T{min(16, os.cpu_count() or 1)}", "-oref", "t.ktab"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = (tmp_path / "ref.smu").read_text()
    plot = torch.zeros(engine.PLOT_CELLS, dtype=torch.int64, device=dev)
    e = engine.Engine(0)
    e.bind(k, len(cnt), tk.data_ptr(), tc.data_ptr())
    for mode, px in (("hash", None), ("exact", None), ("hash", "0"), ("hash", "1"), ("hash", "one-xcc")):
        if px == "one-xcc":                              # ... and a device that shows one XCC id: the other classes are stolen
            monkeypatch.setenv("SMG_PX_ONE_XCC", "1")
        elif px is not None:
            monkeypatch.setenv("SMG_PROBE_X", px)        # both forms of the fused probe (the default picks by the deferred share)
        st = e.run(plot.data_ptr(), mode)
        torch.cuda.synchronize()
        assert st["path"] == 1 and st["nbig"] > 10000, st
        assert engine.smu_text(plot.cpu().numpy().reshape(1001, 501)) == want, (mode, px)
    monkeypatch.delenv("SMG_PROBE_X")
    monkeypatch.delenv("SMG_PX_ONE_XCC")
    for env in ({"SMG_VIRTUAL_SHARDS": "3"}, {}):
        r = subprocess.run([HETMERS_BIN, "-e10", "-T4", "-ogpu", "t.ktab"], cwd=tmp_path, capture_output=True, text=True,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        assert (tmp_path / "gpu.smu").read_text() == want, env


def test_engine_object_with_torch_tensors_and_manual_two_shard_exchange():
    """phase-level API on device tensors; then the same table split into two prefix shards on
    one GPU with the request exchange done by hand (what sharded.py does with all_to_all)"""
    import torch
    from smudgeplot_amd import sharded
    k = 31
    keys, cnt = synth.diploid_table_u64(60000, k=k, seed=9, het_frac=0.35, cov=30, L=6)
    want = brute.hetmers_plot(ktab.u64_to_packed(keys, k), cnt, k)
    dev = torch.device("cuda:0")
    tk = torch.from_numpy(keys.view(np.int64)).to(dev)
    tc = torch.from_numpy(cnt.view(np.int16)).to(dev)
    plot = torch.zeros(engine.PLOT_CELLS, dtype=torch.int64, device=dev)
    e = engine.Engine(0, torch.cuda.current_stream().cuda_stream)
    e.bind(k, len(cnt), tk.data_ptr(), tc.data_ptr())
    for mode in ("exact", "hash", "none"):
        st = e.run(plot.data_ptr(), mode)
        torch.cuda.synchronize()
        assert np.array_equal(plot.cpu().numpy().reshape(1001, 501), want), mode
        assert st["nels"] == len(cnt) and st["ms_total"] > 0

    # single-rank driver (world size 1, no process group)
    p1, st1 = sharded.hetmers_sharded(k, tk, tc, symcheck="hash")
    assert np.array_equal(p1.cpu().numpy().reshape(1001, 501), want)

    # two shards, manual exchange
    cut = sharded.fix_cut(keys, 1, k, len(cnt) // 2)
    shards = [(tk[:cut].clone(), tc[:cut].clone()), (tk[cut:].clone(), tc[cut:].clone())]
    engs = [sharded.TorchEngine(dev) for _ in range(2)]
    split = keys[cut:cut + 1].copy()
    sends, counts = [], []
    for (sk, sc), en in zip(shards, engs):
        en.bind(k, sk, sc)
        en.pass1("hash")
        buf = torch.empty(max(en.nreq(), 1) * en.record_words(), dtype=torch.int64, device=dev)
        counts.append(en.route(split, 2, buf))
        sends.append(buf)
    rw = engs[0].record_words()
    total = torch.zeros(engine.PLOT_CELLS, dtype=torch.int64, device=dev)
    fps = np.zeros(4, dtype=np.uint64)
    for dst, en in enumerate(engs):
        parts = []
        for src in range(2):
            off = sum(counts[src][:dst]) * rw
            parts.append(sends[src][off: off + counts[src][dst] * rw])
        recv = torch.cat(parts)
        assert en.apply(recv, recv.numel() // rw) == 0
        fps = fps ^ np.array(en.symhash(), dtype=np.uint64)          # the shards' residues combine by XOR
    assert fps[0] == fps[2] and fps[1] == fps[3]
    for en in engs:
        pl = torch.zeros(engine.PLOT_CELLS, dtype=torch.int64, device=dev)
        en.pass2(pl)
        total += pl
    torch.cuda.synchronize()
    assert np.array_equal(total.cpu().numpy().reshape(1001, 501), want)


@pytest.mark.parametrize("k,m,seed", [(31, 30000, 3), (21, 30000, 4), (12, 20000, 5), (32, 8000, 6), (33, 8000, 7),
                                      (40, 20000, 8), (51, 6000, 9), (64, 5000, 10), (65, 5000, 11), (70, 5000, 12),
                                      (85, 4000, 13)])
def test_request_filter_changes_nothing_but_the_request_count(k, m, seed, monkeypatch):
    """hash proof, k <= 85: requests whose target block holds no candidate are dropped before the look-ups"""
    packed, cnt = synth.adversarial_table(k, m, 4, seed, low_complexity=40, dense=1)
    want = brute.hetmers_plot(packed, cnt, k) if m * k <= 400000 else None
    tab = table_from(packed, cnt, k)
    plot_f, st_f = engine.hetmers_run(tab, symcheck="hash")
    monkeypatch.setenv("SMG_FILTER_SORT_MIN", "1")         # long lists are bucketed on their leading 8 bits first
    plot_s, st_s = engine.hetmers_run(tab, symcheck="hash")
    assert np.array_equal(plot_f, plot_s) and st_s["nrequests"] == st_f["nrequests"]
    monkeypatch.setenv("SMG_NO_FILTER", "1")
    plot_u, st_u = engine.hetmers_run(tab, symcheck="hash")
    assert np.array_equal(plot_f, plot_u)
    if want is not None:
        assert np.array_equal(plot_f, want)
    assert st_u["nrequests"] == st_u["nemitted"] == st_f["nemitted"] > 0
    assert st_f["nrequests"] <= st_f["nemitted"]
    if k >= 24:                                            # (sparse tables: nearly every target block is empty)
        assert st_f["nrequests"] < st_f["nemitted"]


@pytest.mark.parametrize("k,symcheck", [(31, "hash"), (12, "hash"), (31, "exact"), (40, "hash"), (97, "hash"), (128, "exact")])
def test_sharded_driver_on_the_real_backend_one_rank_group(k, symcheck, monkeypatch):
    """`sharded.hetmers_sharded` as bench.py runs it for N > 1 -- splitter all_gather, block-map all_gather, request
    filter, route, all_to_all_single of counts and requests, apply, all_reduce of histogram + proof -- on RCCL with the
    real engine; one rank is all this box has, SMG_FORCE_EXCHANGE makes it run every collective of the protocol"""
    import socket
    import torch
    import torch.distributed as dist
    from smudgeplot_amd import sharded
    packed, cnt = synth.adversarial_table(k, 6000, 4, 21, low_complexity=30, dense=1)
    want = brute.hetmers_plot(packed, cnt, k)
    words = (k + 31) // 32
    buf = np.zeros((len(cnt), 8 * words), dtype=np.uint8)              # left-aligned big-endian words
    buf[:, :packed.shape[1]] = packed
    keys = buf.view(">u8").astype(np.uint64)
    dev = torch.device("cuda:0")
    tk = torch.from_numpy(np.ascontiguousarray(keys).view(np.int64).reshape(-1)).to(dev)
    tc = torch.from_numpy(cnt.view(np.int16)).to(dev)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    monkeypatch.setenv("SMG_FORCE_EXCHANGE", "1")
    monkeypatch.setenv("SMG_REPLAY", "1")                # (replayed steps: off by default, see sharded.py)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        for _ in range(2):                               # second call: cached splitters, reused engine
            plot, st = sharded.hetmers_sharded(k, tk, tc, symcheck=symcheck, eng=st["engine"] if _ else None)
            assert np.array_equal(plot.cpu().numpy().reshape(1001, 501), want)
            assert st["sent"] == st["received"] and st["world"] == 1
            assert not st["replayed"]                    # (every call binds its table: a new table as far as the engine knows)
        # the 29-bit map that 8 ranks exchange (TorchEngine.pass1 picks it for world >= 8, which this box cannot be): forced
        # here, together with the table's prefix index as the shard's look-up directory (what bench.py --gpus N hands over)
        monkeypatch.setenv("SMG_BM_BITS", "29")
        eng = sharded.TorchEngine(dev)
        if k >= 12:
            kw0 = tk.reshape(len(cnt), -1)[:, 0]
            index = torch.cumsum(torch.bincount((kw0 >> 40) & 0xFFFFFF, minlength=1 << 24), 0)
            eng.bind(k, tk, tc, index=index)
        else:
            eng.bind(k, tk, tc)
        for _ in range(3):
            plot, st = sharded.hetmers_sharded(k, tk, tc, symcheck=symcheck, eng=eng, prebound=True)
            assert np.array_equal(plot.cpu().numpy().reshape(1001, 501), want)
            # a step on the table of the step before is queued from recorded counts (hash proof, look-up chain: 12 <= k <= 64)
            assert st["replayed"] == (_ > 0 and symcheck == "hash" and 12 <= k <= 64), (_, st["replayed"])
            assert st["ms_pass1"] > 0 and st["sent"] == st["received"]
            if st["replayed"]:
                assert st["nemitted"] >= st["nrequests"] > 0 and st["ms_rclookup"] > 0
        # the table changes IN PLACE under a prebound engine (counts only: the k-mers of a bound table must stay): every 5th
        # (k-mer, complement) couple gets counts that take its pairs over the sum limit -- other request counts, so the
        # replayed step reports that its record does not hold and the step is run again the plain way; the answer is the
        # oracle's for the changed table, and the step after that replays again
        if symcheck == "hash" and 12 <= k <= 64:
            assert st["replayed"]
            rc = ktab.revcomp_packed(packed, k)
            pos = {bytes(x): i for i, x in enumerate(packed)}
            cnt2 = cnt.copy()
            for i in range(0, len(cnt), 5):
                cnt2[i] = 900; cnt2[pos[bytes(rc[i])]] = 900
            tc.copy_(torch.from_numpy(cnt2.view(np.int16)).to(dev))
            want2 = brute.hetmers_plot(packed, cnt2, k)
            assert not np.array_equal(want2, want)
            plot, st = sharded.hetmers_sharded(k, tk, tc, symcheck=symcheck, eng=eng, prebound=True)
            assert np.array_equal(plot.cpu().numpy().reshape(1001, 501), want2) and not st["replayed"] and st["path"] == 1
            plot, st = sharded.hetmers_sharded(k, tk, tc, symcheck=symcheck, eng=eng, prebound=True)
            assert np.array_equal(plot.cpu().numpy().reshape(1001, 501), want2) and st["replayed"]
            # ... and without SMG_REPLAY=1 it is the round-4 step (every count read back)
            monkeypatch.delenv("SMG_REPLAY")
            plot, st = sharded.hetmers_sharded(k, tk, tc, symcheck=symcheck, eng=eng, prebound=True)
            assert np.array_equal(plot.cpu().numpy().reshape(1001, 501), want2) and not st["replayed"]
    finally:
        dist.destroy_process_group()


# ---- table conditioning on the device (row A0: what the reference delegates to Logex / Symmex) -------------

def _raw_table(k, seed, L):
    """a RAW FastK-like table: canonical k-mers only, some of them below the -e threshold, plus the
    conditioned (trimmed + symmetric) table a correct Logex + Symmex would make of it"""
    packed, cnt = synth.adversarial_table(k, 2500, L, seed, low_complexity=120, dense=2)
    rc = ktab.revcomp_packed(packed, k)
    canon = np.array([bytes(a) <= bytes(b) for a, b in zip(packed, rc)])
    rp, rcnt = packed[canon], cnt[canon].copy()
    rng = np.random.default_rng(seed)
    low = rng.random(len(rcnt)) < 0.2
    rcnt[low] = rng.integers(1, L, size=int(low.sum()))           # erroneous k-mers
    keep = rcnt >= L
    cp, cc = ktab.symmetrize(rp[keep], rcnt[keep], k)
    return (rp, rcnt), (cp, cc)


@pytest.mark.parametrize("k,seed", [(31, 1), (32, 2), (21, 3), (40, 4), (65, 5), (100, 6)])
def test_condition_on_device_matches_numpy_conditioning(k, seed):
    L = 5
    (rp, rcnt), (cp, cc) = _raw_table(k, seed, L)
    want = brute.hetmers_plot(cp, cc, k)
    assert want.sum() > 0
    for mode in ("hash", "exact"):
        plot, st = engine.hetmers_run(table_from(rp, rcnt, k), symcheck=mode,
                                      condition=engine.COND_TRIM | engine.COND_SYMM, ethresh=L)
        assert st["nels"] == len(cc)
        assert st["path"] == 1, "the conditioned table must pass the symmetry proof"
        assert np.array_equal(plot, want), (k, mode)
    # trim only / symmetrise only
    keep = rcnt >= L
    plot, st = engine.hetmers_run(table_from(rp[keep], rcnt[keep], k), condition=engine.COND_SYMM, ethresh=L)
    assert np.array_equal(plot, want) and st["nels"] == len(cc)
    sp, sc = ktab.symmetrize(rp, rcnt, k)
    plot, st = engine.hetmers_run(table_from(sp, sc, k), condition=engine.COND_TRIM, ethresh=L)
    keep2 = sc >= L
    assert np.array_equal(plot, brute.hetmers_plot(sp[keep2], sc[keep2], k)) and st["nels"] == int(keep2.sum())


@pytest.mark.parametrize("k", [31, 51])
def test_hetmers_on_raw_table_equals_reference_on_conditioned_table(k, tmp_path):
    """condition-then-reference == ours-on-raw: the drop-in binary reads the RAW
    canonical table; the reference binary (or, where it is absent, the C oracle) reads the table
    conditioned by numpy."""
    from conftest import REF_BIN
    L = 6
    (rp, rcnt), (cp, cc) = _raw_table(k, 40 + k, L)
    ktab.write_ktab(str(tmp_path / "raw"), k, rp, rcnt, ibyte=1, nparts=2)
    ktab.write_ktab(str(tmp_path / "cond"), k, cp, cc, ibyte=1, nparts=2)
    r = subprocess.run([HETMERS_BIN, f"-e{L}", "-T4", "-v", "-ogpu", "raw"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "  The input table is untrimmed and not symmetric\n" in r.stderr
    assert "  Trimming k-mers in table with count < 6\n" in r.stderr
    assert "  Making trimmed table symmetric\n" in r.stderr
    assert not (tmp_path / ".trim.ktab").exists() and not (tmp_path / ".symx.ktab").exists()
    if os.path.exists(REF_BIN):
        q = subprocess.run([REF_BIN, f"-e{L}", "-T4", "-oref", "cond"], cwd=tmp_path, capture_output=True, text=True)
        assert q.returncode == 0, q.stderr
    else:
        subprocess.run([ORACLE_BIN, f"-e{L}", f"-o{tmp_path}/ref", str(tmp_path / "cond")], check=True)
    assert (tmp_path / "gpu.smu").read_text() == (tmp_path / "ref.smu").read_text()
    assert (tmp_path / "gpu.smu").read_text() != ""


# ---- conditioning across shards (row A0 beyond one shard: several GPUs, or more than 2^32 entries) -----------------

def _golden_raw(name, L_extra=3):
    """the canonical half of a golden table, some entries pushed below the threshold: what FastK would have written;
    and the table a correct Logex + Symmex makes of it"""
    g = load_golden(name)
    k, L = g["k"], g["L"] + L_extra
    packed, cnt = g["packed"], g["counts"]
    rc = ktab.revcomp_packed(packed, k)
    canon = np.array([bytes(a) <= bytes(b) for a, b in zip(packed, rc)])
    rp, rcnt = packed[canon], cnt[canon].copy()
    rng = np.random.default_rng(len(cnt))
    low = rng.random(len(rcnt)) < 0.15
    rcnt[low] = rng.integers(1, L, size=int(low.sum()))
    keep = rcnt >= L
    cp, cc = ktab.symmetrize(rp[keep], rcnt[keep], k)
    return k, L, (rp, rcnt), (cp, cc)


@pytest.mark.parametrize("shards", [2, 3, 7])
@pytest.mark.parametrize("name", ["k31_i1", "k31_i3_p4", "k21_i2_p2", "k51_i1_p3", "k32_i1_p2", "k65_i1", "k17_i1"])
def test_raw_table_is_trimmed_and_symmetrised_across_virtual_shards(name, shards, monkeypatch):
    """PloidyPlot.c:1381-1414 conditions a table of any size.  Raw canonical table, cut into shards anywhere: every
    shard trims, the shards exchange entries + complements (balanced splitters from the shape of the closed table),
    sort, and run -- the plot must be the oracle's on the numpy-conditioned table."""
    k, L, (rp, rcnt), (cp, cc) = _golden_raw(name)
    want = brute.hetmers_plot(cp, cc, k)
    assert want.sum() > 0
    monkeypatch.setenv("SMG_VIRTUAL_SHARDS", str(shards))
    for mode in ("hash", "exact"):
        plot, st = engine.hetmers_run(table_from(rp, rcnt, k), symcheck=mode,
                                      condition=engine.COND_TRIM | engine.COND_SYMM, ethresh=L)
        assert st["path"] == 1, "a table closed by the engine must pass the symmetry proof"
        assert st["nels"] == len(cc), (name, shards, mode)
        assert np.array_equal(plot, want), (name, shards, mode)


@pytest.mark.parametrize("name", ["k31_i1", "k51_i1_p3"])
def test_raw_table_beyond_the_shard_limit_and_on_one_rank_rccl(name, monkeypatch, tmp_path):
    """the same through the automatic shards of a table whose CLOSED size exceeds the shard limit (the 2^32-entry case
    with the threshold lowered), through the RCCL calls with one rank, and through the executable with SMUDGEPLOT_GPUS
    (which used to fall back to one GPU for a raw table): byte-identical .smu against the reference binary (or the C
    oracle) on the conditioned table"""
    from conftest import REF_BIN
    k, L, (rp, rcnt), (cp, cc) = _golden_raw(name)
    want = brute.hetmers_plot(cp, cc, k)
    monkeypatch.setenv("SMG_SHARD_LIMIT", str(len(cc) // 3 + 1))          # closed size / 3: three or more shards
    plot, st = engine.hetmers_run(table_from(rp, rcnt, k), symcheck="hash", condition=engine.COND_TRIM | engine.COND_SYMM, ethresh=L)
    assert np.array_equal(plot, want) and st["nels"] == len(cc)
    monkeypatch.delenv("SMG_SHARD_LIMIT")
    monkeypatch.setenv("SMG_FORCE_MULTI", "1")
    plot, st = engine.hetmers_run(table_from(rp, rcnt, k), symcheck="hash", condition=engine.COND_TRIM | engine.COND_SYMM, ethresh=L)
    assert np.array_equal(plot, want) and st["nels"] == len(cc)
    monkeypatch.delenv("SMG_FORCE_MULTI")
    ktab.write_ktab(str(tmp_path / "raw"), k, rp, rcnt, ibyte=1, nparts=2)
    ktab.write_ktab(str(tmp_path / "cond"), k, cp, cc, ibyte=1, nparts=2)
    r = subprocess.run([HETMERS_BIN, f"-e{L}", "-T4", "-v", "-ogpu", "raw"], cwd=tmp_path, capture_output=True, text=True,
                       env=dict(os.environ, SMUDGEPLOT_GPUS="3", SMG_VIRTUAL_SHARDS="3"))
    assert r.returncode == 0, r.stderr
    assert "  Making trimmed table symmetric\n" in r.stderr and "gpus=3" in r.stderr
    if os.path.exists(REF_BIN):
        q = subprocess.run([REF_BIN, f"-e{L}", "-T4", "-oref", "cond"], cwd=tmp_path, capture_output=True, text=True)
        assert q.returncode == 0, q.stderr
    else:
        subprocess.run([ORACLE_BIN, f"-e{L}", f"-o{tmp_path}/ref", str(tmp_path / "cond")], check=True)
    assert (tmp_path / "gpu.smu").read_text() == (tmp_path / "ref.smu").read_text() != ""


@pytest.mark.parametrize("k,seed", [(31, 11), (40, 12), (70, 13)])
def test_condition_sharded_driver_one_rank_and_engine_primitives(k, seed):
    """sharded.condition_sharded on the real engine (one rank: symm_hist / symm_route / symm_finish without the
    collectives), then hetmers_sharded on the table the engine owns; and the histogram of symm_hist against numpy"""
    import torch
    from smudgeplot_amd import sharded
    L = 5
    (rp, rcnt), (cp, cc) = _raw_table(k, seed, L)
    want = brute.hetmers_plot(cp, cc, k)
    W = (k + 31) // 32
    buf = np.zeros((len(rcnt), 8 * W), dtype=np.uint8)
    buf[:, : rp.shape[1]] = rp
    keys = np.ascontiguousarray(buf.view(">u8").astype(np.uint64))
    dev = torch.device("cuda:0")
    tk = torch.from_numpy(keys.view(np.int64).reshape(-1).copy()).to(dev)
    tc = torch.from_numpy(rcnt.view(np.int16).copy()).to(dev)
    eng = sharded.TorchEngine(dev)
    eng.bind(k, tk, tc)
    bits = 8
    h = eng.symm_hist(bits)
    own = np.bincount((keys[:, 0] >> np.uint64(64 - bits)).astype(np.int64), minlength=1 << bits)
    rcp = ktab.revcomp_packed(rp, k)
    rcs = np.bincount(rcp[:, 0].astype(np.int64), minlength=1 << bits)
    assert np.array_equal(h[: 1 << bits], own) and np.array_equal(h[1 << bits:], rcs)
    eng2, split = sharded.condition_sharded(k, tk, tc, ethresh=L, trim=True, symm=True)
    assert eng2.nels() == len(cc) and len(split) == 0
    plot, st = sharded.hetmers_sharded(k, None, None, symcheck="hash", eng=eng2, splitters=split)
    torch.cuda.synchronize()
    assert st["path"] == 1
    assert np.array_equal(plot.cpu().numpy().reshape(1001, 501), want)


# ---- BASELINE config 3 at FULL size: size-independent properties (no CPU oracle finishes 2.5e9 entries) ----

def _manual_sharded(k, tk, tc, cuts, dev, symcheck="hash"):
    """the sharded protocol on ONE GPU: one engine per prefix shard, the request exchange done by hand.
    tk: the table's k-mer words (n * W int64, W = ceil(k / 32)), cuts in entries."""
    import torch
    from smudgeplot_amd import sharded
    world = len(cuts) - 1
    W = (k + 31) // 32
    engs, sends, counts = [], [], []
    ntab = tc.numel()                          # (an empty trailing shard starts at +infinity)
    firsts = [tk[c * W:(c + 1) * W].cpu().numpy().view(np.uint64) if c < ntab else np.full(W, ~np.uint64(0)) for c in cuts[1:-1]]
    split = np.concatenate(firsts) if firsts else np.zeros(0, np.uint64)
    for r in range(world):
        en = sharded.TorchEngine(dev)
        en.bind(k, tk[cuts[r] * W:cuts[r + 1] * W].clone(), tc[cuts[r]:cuts[r + 1]].clone())    # aligned copies
        en.pass1(symcheck)
        engs.append(en)
    bits, nwords = engs[0].blockmap()
    assert all(en.blockmap() == (bits, nwords) for en in engs)        # empty shards included
    emitted = sum(en.nreq() for en in engs)
    if bits:                                   # request filter: the exchange of the candidate block maps, by hand
        wlo, wlen = sharded.blockmap_ranges(split, W, world, bits, nwords // (((1 << bits) + 31) >> 5))
        full = torch.zeros(nwords, dtype=torch.int32, device=dev)
        for r, en in enumerate(engs):
            part = torch.zeros(wlen[r], dtype=torch.int32, device=dev)
            en.blockmap_copy(wlo[r], wlen[r], part)
            full[wlo[r]: wlo[r] + wlen[r]] |= part
        for en in engs:
            before = en.nreq()
            assert en.filter(full) == en.nreq() <= before
    for en in engs:
        buf = torch.empty(max(en.nreq(), 1) * en.record_words(), dtype=torch.int64, device=dev)
        counts.append(en.route(split, world, buf))
        sends.append(buf)
    rw = engs[0].record_words()
    fps = np.zeros(4, dtype=np.uint64)
    for dst, en in enumerate(engs):
        parts = []
        for src in range(world):
            off = sum(counts[src][:dst]) * rw
            parts.append(sends[src][off: off + counts[src][dst] * rw])
        recv = torch.cat(parts)
        assert en.apply(recv, recv.numel() // rw) == 0
        fps = fps ^ np.array(en.symhash(), dtype=np.uint64)          # the shards' residues combine by XOR
    assert fps[0] == fps[2] and fps[1] == fps[3]
    total = torch.zeros(engine.PLOT_CELLS, dtype=torch.int64, device=dev)
    for en in engs:
        pl = torch.zeros(engine.PLOT_CELLS, dtype=torch.int64, device=dev)
        en.pass2(pl)
        total += pl
    torch.cuda.synchronize()
    assert sum(sum(c) for c in counts) <= emitted
    return total, emitted


@pytest.mark.parametrize("k", [12, 31, 40])
def test_manual_shards_with_an_empty_shard(k):
    """a rank that owns nothing still reports the block-map geometry of the others and joins the exchange"""
    import torch
    packed, cnt = synth.adversarial_table(k, 5000, 4, 33, low_complexity=20, dense=1)
    want = brute.hetmers_plot(packed, cnt, k)
    W = (k + 31) // 32
    buf = np.zeros((len(cnt), 8 * W), dtype=np.uint8)
    buf[:, : packed.shape[1]] = packed
    keys = np.ascontiguousarray(buf.view(">u8").astype(np.uint64)).reshape(-1)       # n * W words, left aligned
    dev = torch.device("cuda:0")
    tk = torch.from_numpy(keys.view(np.int64)).to(dev)
    tc = torch.from_numpy(cnt.view(np.int16)).to(dev)
    n = len(cnt)
    from smudgeplot_amd import sharded
    cut = sharded.fix_cut(keys, W, k, n // 2)
    for cuts in ([0, 0, cut, n], [0, cut, cut, n], [0, cut, n, n]):
        tot, _ = _manual_sharded(k, tk, tc, cuts, dev)
        assert np.array_equal(tot.cpu().numpy().reshape(1001, 501), want), cuts


def test_full_size_config3_properties():
    """Synthetic diploid 1 Gbp, 50x, k=31 (BASELINE configs[2], 2.5e9 entries, the bench workload):
       idempotence, hash == exact symmetry proof, invariance under 3-way prefix sharding (uneven cuts),
       and the mirror-image identity: on a symmetric table every pair has a distinct mirror image in the
       same cell, except a pair {x, rc(x)} on the self-mirrored position, whose two counts are equal --
       so every cell OFF the diagonal sum == 2*min holds an even number."""
    import torch
    from smudgeplot_amd import sharded, synth_device
    k = 31
    dev = torch.device("cuda:0")
    free, _ = torch.cuda.mem_get_info()
    G = 1_000_000_000 if free > 150e9 else 100_000_000
    print(f"full-size property test: G={G}, free HBM {free / 1e9:.0f} GB")
    tk, tc = synth_device.diploid_table(G, k=k, het=0.01, cov=50.0, L=10, seed=1, device=dev)
    n = tc.numel()
    assert n > 2.0 * G
    torch.cuda.empty_cache()
    plot = torch.zeros(engine.PLOT_CELLS, dtype=torch.int64, device=dev)
    e = engine.Engine(0)
    e.bind(k, n, tk.data_ptr(), tc.data_ptr())
    st = e.run(plot.data_ptr(), "hash"); torch.cuda.synchronize()
    assert st["path"] == 1
    p_hash = plot.clone()
    st = e.run(plot.data_ptr(), "hash"); torch.cuda.synchronize()
    assert torch.equal(plot, p_hash), "idempotence"
    st = e.run(plot.data_ptr(), "exact"); torch.cuda.synchronize()
    assert st["path"] == 1 and torch.equal(plot, p_hash), "hash and exact proofs agree"
    total = int(p_hash.sum().item())
    assert total > n // 20
    assert int(p_hash.view(1001, 501)[:, 500].sum().item()) >= 0
    # sums above 1000 can never be counted; min <= sum / 2
    pm = p_hash.view(1001, 501)
    s_idx, m_idx = torch.nonzero(pm, as_tuple=True)
    assert bool((2 * m_idx <= s_idx).all())
    off_diag = pm[s_idx, m_idx][2 * m_idx != s_idx]
    assert off_diag.numel() > 100 and bool((off_diag % 2 == 0).all()), "mirror-image identity"
    del e
    # 3 uneven prefix shards on one GPU
    keys_first = tk
    cuts = [0, n // 5, n // 5 + n // 2, n]
    sh = 64 - 2 * (k // 2)
    for i in (1, 2):
        c = cuts[i]
        while int(keys_first[c] >> sh) == int(keys_first[c - 1] >> sh):
            c += 1
        cuts[i] = c
    tot, nsent = _manual_sharded(k, tk, tc, cuts, dev)
    assert nsent > 0 and torch.equal(tot, p_hash), "prefix sharding does not change the plot"


@pytest.mark.parametrize("k,m,seed", [(5, 25, 1), (5, 300, 11), (6, 80, 2), (6, 900, 12), (8, 600, 3), (8, 3000, 13), (9, 3000, 4), (12, 4000, 5),
                                      (15, 4000, 6), (16, 4000, 7), (17, 4000, 8), (24, 3000, 9), (30, 3000, 10)])
def test_small_and_even_k_vs_oracle(k, m, seed):
    """k <= 16 (the k-mer lives in the upper 32-bit half only), dense tables with window blocks far longer
    than the +-30 window (4^k possible k-mers), even k with self-complementary k-mers"""
    packed, cnt = synth.adversarial_table(k, m, 4, seed, low_complexity=0, dense=0)
    want = brute.hetmers_plot(packed, cnt, k)
    assert want.sum() > 0 or m >= 300          # the dense small-k tables have no unique pair at all
    for mode in ("hash", "exact", "none"):
        plot, st = engine.hetmers_run(table_from(packed, cnt, k), symcheck=mode)
        assert np.array_equal(plot, want), (k, mode)
        assert st["path"] == (2 if mode == "none" else 1)


@pytest.mark.parametrize("n_target", [959, 960, 961, 963, 964, 965, 991, 992, 993, 995, 996, 997, 1000, 1023, 1024, 1025, 1031, 1032,
                                      1033, 1919, 1920, 1921, 1924, 1956, 2880 + 36, 4 * 960 + 1, 4 * 960 + 37])
def test_table_sizes_around_the_pass1_tile(n_target):
    """entry counts on and next to the tile edges of kf_pass1_d (960 owned entries per tile, scanned with 36 staged
    entries in front and 4 behind; a tile is 'inner' when 1032 entries from its first staged one exist)"""
    k = 31
    keys, cnt = synth.diploid_table_u64(1500, k=k, seed=n_target, het_frac=0.5, cov=30, L=5)
    assert len(cnt) > n_target
    # cut to the wanted size while keeping the table closed under reverse complement
    rc = ktab.revcomp_u64(keys, k)
    canon = np.minimum(keys, rc)
    order = np.argsort(canon, kind="stable")
    keep = np.zeros(len(keys), bool)
    keep[order[: (n_target // 2) * 2]] = True              # whole {x, rc(x)} classes (odd k: always 2 members)
    kk, cc = keys[keep], cnt[keep]
    if n_target % 2:                                        # odd size: one extra entry breaks the symmetry
        extra = order[(n_target // 2) * 2]
        keep[extra] = True
        kk, cc = keys[keep], cnt[keep]
    assert len(cc) == n_target
    packed = ktab.u64_to_packed(kk, k)
    want = brute.hetmers_plot(packed, cc, k)
    plot, st = engine.hetmers_run(table_from(packed, cc, k), symcheck="hash")
    assert np.array_equal(plot, want)
    assert st["path"] == (2 if n_target % 2 else 1)


# ---- several GPUs behind the C ABI (smg_multi.hpp), exercised as VIRTUAL shards on the one GPU of the box --

@pytest.mark.parametrize("shards", [2, 3, 5])
@pytest.mark.parametrize("name", ["k31_i1", "k31_i3_p4", "k21_i2_p2", "k17_i1", "k51_i1_p3", "k65_i1", "k32_i1_p2"])
def test_multi_gpu_path_virtual_shards_golden(name, shards, monkeypatch):
    """prefix cuts from the FastK index, ranged decode, request routing between shards, host-side proof,
    histogram sum: everything of the multi-GPU path except the RCCL calls themselves"""
    g = load_golden(name)
    monkeypatch.setenv("SMG_VIRTUAL_SHARDS", str(shards))
    for mode in ("hash", "exact"):
        plot, st = engine.hetmers_run(make_table(g), symcheck=mode)
        assert engine.smu_text(plot) == g["smu"], (name, shards, mode)
        assert st["nels"] == len(g["counts"]) and st["path"] == 1


@pytest.mark.parametrize("shards", [2, 5])
@pytest.mark.parametrize("name", ["k100_i1", "k100_wrap"])
def test_multi_gpu_path_above_k_85_golden(name, shards, monkeypatch):
    """k > 85 (a uint8 degree can wrap, PloidyPlot.c:163): the counted kernels in the steps of the sharded protocol --
    flat record list through the router, degrees added on the owner shard, no block map; the reference has no k limit"""
    g = load_golden(name)
    monkeypatch.setenv("SMG_VIRTUAL_SHARDS", str(shards))
    for mode in ("hash", "exact"):
        plot, st = engine.hetmers_run(make_table(g), symcheck=mode)
        assert engine.smu_text(plot) == g["smu"], (name, shards, mode)
        assert st["nels"] == len(g["counts"]) and st["path"] == 1
    monkeypatch.delenv("SMG_VIRTUAL_SHARDS")
    monkeypatch.setenv("SMG_FORCE_MULTI", "1")             # a one-rank RCCL communicator: send/recv to self, all-reduce
    plot, st = engine.hetmers_run(make_table(g), symcheck="hash")
    assert engine.smu_text(plot) == g["smu"]


@pytest.mark.parametrize("k,seed,shards", [(86, 31, 3), (97, 32, 4), (128, 33, 2)])
def test_multi_gpu_path_above_k_85_fresh_tables_and_raw_input(k, seed, shards, monkeypatch):
    """fresh tables at three- and four-word k against the brute-force restatement, cut into shards; then the RAW table
    (canonical k-mers, some below the threshold) trimmed and closed ACROSS the shards first; then a table that is not
    closed: the shards take the general path together"""
    (rp, rcnt), (cp, cc) = _raw_table(k, seed, 4)
    want = brute.hetmers_plot(cp, cc, k)
    monkeypatch.setenv("SMG_VIRTUAL_SHARDS", str(shards))
    for mode in ("hash", "exact"):
        plot, st = engine.hetmers_run(table_from(cp, cc, k), symcheck=mode)
        assert np.array_equal(plot, want), (k, mode)
        assert st["path"] == 1
    plot, st = engine.hetmers_run(table_from(rp, rcnt, k), symcheck="hash",
                                  condition=engine.COND_TRIM | engine.COND_SYMM, ethresh=4)
    assert np.array_equal(plot, want)
    assert st["nels"] == len(cc)
    keep = np.ones(len(cc), bool)
    keep[len(cc) // 3] = False                                         # one complement missing
    want_g = brute.hetmers_plot(cp[keep], cc[keep], k)
    plot, st = engine.hetmers_run(table_from(cp[keep], cc[keep], k), symcheck="hash")
    assert np.array_equal(plot, want_g) and st["path"] == 2


def test_multi_gpu_path_small_k_wide_index(monkeypatch):
    """k=17 with a 3-byte index: an index bucket (12 bases) is FINER than a window block (8 bases), so only
    every 256th bucket boundary may carry a cut"""
    g = load_golden("k17_i1")
    g = dict(g, ibyte=3, nparts=3)
    monkeypatch.setenv("SMG_VIRTUAL_SHARDS", "4")
    plot, st = engine.hetmers_run(make_table(g), symcheck="hash")
    assert engine.smu_text(plot) == g["smu"]


def test_multi_gpu_path_medium_table_and_executable(tmp_path, monkeypatch):
    k = 31
    keys, cnt = synth.diploid_table_u64(200000, k=k, seed=15, het_frac=0.3, cov=40, L=8)
    packed = ktab.u64_to_packed(keys, k)
    ktab.write_ktab(str(tmp_path / "t"), k, packed, cnt, ibyte=2, nparts=3)
    subprocess.run([ORACLE_BIN, "-e8", f"-o{tmp_path}/orc", str(tmp_path / "t")], check=True)
    want = (tmp_path / "orc.smu").read_text()
    env = dict(os.environ, SMUDGEPLOT_GPUS="4", SMG_VIRTUAL_SHARDS="4")
    r = subprocess.run([HETMERS_BIN, "-e8", "-T4", "-v", "-ogpu4", "t"], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "gpus=4 (virtual shards on one device)" in r.stderr
    assert (tmp_path / "gpu4.smu").read_text() == want
    # untrimmed input: every shard trims its own entries
    cnt2 = cnt.copy()
    rng = np.random.default_rng(1)
    # lower BOTH members of some complement classes below the threshold (the table stays symmetric)
    rc = ktab.revcomp_u64(keys, k)
    j = np.searchsorted(keys, rc)
    pick = rng.random(len(keys)) < 0.1
    pick = pick | pick[j]
    cnt2[pick] = 3
    ktab.write_ktab(str(tmp_path / "u"), k, packed, cnt2, ibyte=2, nparts=2)
    keep = cnt2 >= 8
    ktab.write_ktab(str(tmp_path / "uc"), k, packed[keep], cnt2[keep], ibyte=2, nparts=2)
    subprocess.run([ORACLE_BIN, "-e8", f"-o{tmp_path}/orc2", str(tmp_path / "uc")], check=True)
    r = subprocess.run([HETMERS_BIN, "-e8", "-v", "-ogpu2", "u"], cwd=tmp_path, capture_output=True, text=True,
                       env=dict(os.environ, SMUDGEPLOT_GPUS="3", SMG_VIRTUAL_SHARDS="3"))
    assert r.returncode == 0, r.stderr
    assert "  The input table is untrimmed yet symmetric\n" in r.stderr and "gpus=3" in r.stderr
    assert (tmp_path / "gpu2.smu").read_text() == (tmp_path / "orc2.smu").read_text()
    # a table that is not closed: the shards notice (proof reduced over all of them), one GPU then takes the general
    # path on the whole table -- the reference gives an answer for such a table, so must every configuration
    bad = np.ones(len(cnt), bool); bad[len(cnt) // 3] = False
    ktab.write_ktab(str(tmp_path / "b"), k, packed[bad], cnt[bad], ibyte=2, nparts=2)
    subprocess.run([ORACLE_BIN, "-e8", f"-o{tmp_path}/orcb", str(tmp_path / "b")], check=True)
    monkeypatch.setenv("SMG_VIRTUAL_SHARDS", "2")
    plot, st = engine.hetmers_run(table_from(packed[bad], cnt[bad], k), symcheck="hash")
    assert st["path"] == 2
    assert engine.smu_text(plot) == (tmp_path / "orcb.smu").read_text()


def test_multi_gpu_path_one_rank_rccl(monkeypatch):
    """SMG_FORCE_MULTI: the multi-GPU path with ONE rank on the real RCCL calls (communicator, grouped
    send/recv to self, all-reduce) -- the most a 1-GPU box can check of them"""
    g = load_golden("k31_i3_p4")
    monkeypatch.setenv("SMG_FORCE_MULTI", "1")
    plot, st = engine.hetmers_run(make_table(g), symcheck="hash", verbose=1)
    assert engine.smu_text(plot) == g["smu"]


def test_config4_octoploid_standin_eight_shards_vs_reference(tmp_path):
    """BASELINE configs[3] (Fragaria x ananassa, octoploid, 8 GPUs) as a synthetic stand-in: 8 haplotypes with
    independent SNPs, k=31.  The drop-in executable runs it as EIGHT prefix shards (virtual shards on this
    box's one GPU: same cutting / routing / reduction as 8 GPUs, copies instead of RCCL) and must be byte
    identical to the reference binary on the same files; one GPU must agree too."""
    import torch
    from conftest import REF_BIN
    from smudgeplot_amd import synth_device
    if not os.path.exists(REF_BIN):
        pytest.skip("prebuilt reference binary not present")
    k, L = 31, 12
    tk, tc = synth_device.polyploid_table(600_000, ploidy=8, div=0.02, cov_hap=15.0, k=k, L=L, seed=4, device="cuda:0")
    keys = tk.cpu().numpy().view(np.uint64); cnt = tc.cpu().numpy().view(np.uint16)
    del tk, tc
    assert len(cnt) > 3_000_000
    synth.write_u64_table(str(tmp_path / "t"), keys, cnt, k, ibyte=3, nparts=8)
    r = subprocess.run([REF_BIN, f"-e{L}", f"-T{min(32, os.cpu_count() or 1)}", "-oref", "t.ktab"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = (tmp_path / "ref.smu").read_text()
    assert want.count("\n") > 1000
    for gpus, out in ((8, "g8"), (1, "g1")):
        env = dict(os.environ, SMUDGEPLOT_GPUS=str(gpus))
        if gpus > 1:
            env["SMG_VIRTUAL_SHARDS"] = str(gpus)
        q = subprocess.run([HETMERS_BIN, f"-e{L}", "-T8", "-v", f"-o{out}", "t.ktab"], cwd=tmp_path, capture_output=True,
                           text=True, env=env)
        assert q.returncode == 0, q.stderr
        assert (f"gpus={gpus}" in q.stderr) == (gpus > 1)
        assert (tmp_path / f"{out}.smu").read_text() == want, gpus


def test_randomised_soak_against_the_oracle():
    """tools/soak.py: random k / size / block structure / proof mode / virtual multi-GPU shards / conditioning from
    raw tables, every plot compared with the numpy oracle (3000 cases were run when this was written)"""
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "250", "11"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "250/250 cases agree" in r.stdout


def test_standalone_conditioning_tool_writes_the_numpy_conditioned_table(tmp_path):
    """smg_condition: raw canonical table in, FastK table out -- the k-mers and counts it writes must equal the
    numpy conditioning entry for entry, and the REFERENCE binary must accept the file as conditioned"""
    from conftest import REF_BIN, ROOT
    tool = os.path.join(ROOT, "smudgeplot_amd", "bin", "smg_condition")
    for k, ibyte, nparts in ((31, 2, 3), (51, 1, 2), (24, 1, 1)):
        L = 6
        (rp, rcnt), (cp, cc) = _raw_table(k, 70 + k, L)
        ktab.write_ktab(str(tmp_path / f"raw{k}"), k, rp, rcnt, ibyte=ibyte, nparts=nparts)
        r = subprocess.run([tool, f"-e{L}", "-v", f"raw{k}.ktab", f"cond{k}"], cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        t = ktab.read_ktab(str(tmp_path / f"cond{k}"))
        assert t.k == k and t.ibyte == ibyte and t.nparts == nparts
        assert np.array_equal(t.packed, cp) and np.array_equal(t.counts, cc)
        # trim only / symmetrise only
        r = subprocess.run([tool, f"-e{L}", "-t", f"raw{k}", f"trim{k}"], cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        tt = ktab.read_ktab(str(tmp_path / f"trim{k}"))
        keep = rcnt >= L
        assert np.array_equal(tt.packed, rp[keep]) and np.array_equal(tt.counts, rcnt[keep])
        if os.path.exists(REF_BIN):
            q = subprocess.run([REF_BIN, f"-e{L}", "-T2", "-v", "-oref", f"cond{k}"], cwd=tmp_path, capture_output=True, text=True)
            assert q.returncode == 0, q.stderr
            assert "  The input table is trimmed and symmetric\n" in q.stderr
            assert (tmp_path / "ref.smu").read_text() == brute.smu_text(brute.hetmers_plot(cp, cc, k))
            os.remove(tmp_path / "ref.smu")


# ---- tables beyond the 32-bit entry index of a shard: prefix shards on ONE device, automatically ------------------

@pytest.mark.parametrize("nshards", [2, 5, 13])
@pytest.mark.parametrize("name", ["k31_i1", "k31_i3_p4", "k21_i2_p2", "k51_i1_p3", "k32_i1_p2"])
def test_tables_beyond_the_shard_limit_are_cut_into_shards_on_one_device(name, nshards, monkeypatch, tmp_path):
    """the product path for tables of >= 2^32 entries (SMG_SHARD_LIMIT lowers the threshold so that the golden
    vectors take it): host_run cuts the table itself, nobody has to ask for it"""
    g = load_golden(name)
    limit = len(g["counts"]) // nshards + 1
    monkeypatch.setenv("SMG_SHARD_LIMIT", str(limit))
    plot, st = engine.hetmers_run(make_table(g), symcheck="hash")
    assert engine.smu_text(plot) == g["smu"]
    assert st["nels"] == len(g["counts"])
    ktab.write_ktab(str(tmp_path / "t"), g["k"], g["packed"], g["counts"], ibyte=g["ibyte"], nparts=g["nparts"])
    r = subprocess.run([HETMERS_BIN, "-oout", f"-e{g['L']}", "-T4", "-v", "t.ktab"], cwd=tmp_path, capture_output=True, text=True,
                       env=dict(os.environ, SMG_SHARD_LIMIT=str(limit)))
    assert r.returncode == 0, r.stderr
    assert f"{nshards} prefix shards on one device" in r.stderr
    assert (tmp_path / "out.smu").read_text() == g["smu"]


@pytest.mark.parametrize("k,nshards", [(31, 3), (40, 4), (21, 2), (70, 3)])
def test_unsymmetric_table_beyond_the_shard_limit_takes_the_general_path_across_the_shards(k, nshards, monkeypatch):
    """the reference answers for ANY sorted table (it only probes entry #1 for symmetry, PloidyPlot.c:1199-1229).  A
    table of more than 2^32 entries lives in several shards of one device; when it fails the proof the shards run the
    general path TOGETHER -- a prefix-side partner is looked up in whichever shard holds it (the threshold is
    lowered here so that small tables take that road)."""
    packed, cnt = synth.adversarial_table(k, 3000, 4, seed=60 + k, low_complexity=80, dense=2)
    rng = np.random.default_rng(k)
    keep = rng.random(len(cnt)) > 0.07                     # holes: the table is no longer closed
    pa, ca = packed[keep], cnt[keep]
    want = brute.hetmers_plot(pa, ca, k)
    assert want.sum() > 0
    monkeypatch.setenv("SMG_SHARD_LIMIT", str(len(ca) // nshards + 1))
    for mode in ("hash", "exact"):
        plot, st = engine.hetmers_run(table_from(pa, ca, k), symcheck=mode)
        assert st["path"] == 2 and st["nels"] == len(ca), (k, mode)
        assert np.array_equal(plot, want), (k, mode)


def _records_on_device(tk, tc, k):
    """format F records (ibyte = 3) + prefix index of a device-resident one-word table, built with torch"""
    import torch
    assert k <= 31
    n = tc.numel()
    kb = (k + 3) // 4
    hb = kb - 3
    rec = torch.empty((n, hb + 2), dtype=torch.uint8, device=tk.device)
    for j in range(hb):                                   # suffix bytes: bytes 3 .. kb-1 of the left-aligned k-mer
        rec[:, j] = ((tk >> (8 * (7 - (3 + j)))) & 0xFF).to(torch.uint8)
    c = tc.to(torch.int32) & 0xFFFF
    rec[:, hb] = (c & 0xFF).to(torch.uint8)
    rec[:, hb + 1] = (c >> 8).to(torch.uint8)
    pre = (tk >> 40) & 0xFFFFFF
    index = torch.cumsum(torch.bincount(pre, minlength=1 << 24), 0)
    return rec, index.cpu().numpy().astype(np.int64)


def _run_source_from_device(rec, index, k, n, limit=None):
    """smg_hetmers_run_source with a read callback that copies the wanted records from the DEVICE tensor `rec`:
    the table never exists in host memory, the engine's own ingestion path (pinned ring, H2D) is what runs"""
    import ctypes as C
    import torch
    lib = engine.load_library()
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    pb = rec.shape[1]
    base = rec.data_ptr()
    READ = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p)

    def read(ctx, part, first, nent, dst):
        return 0 if hip.hipMemcpy(dst, base + first * pb, nent * pb, 2) == 0 else -1      # 2 = device to host

    class Source(C.Structure):
        _fields_ = [("kmer", C.c_int32), ("ibyte", C.c_int32), ("nparts", C.c_int32), ("minval", C.c_int32),
                    ("nels", C.c_int64), ("part_nels", C.POINTER(C.c_int64)), ("prefix_index", C.POINTER(C.c_int64)),
                    ("read", READ), ("ctx", C.c_void_p), ("host_threads", C.c_int32)]
    pn = (C.c_int64 * 1)(n)
    cb = READ(read)
    src = Source(k, 3, 1, 1, n, pn, index.ctypes.data_as(C.POINTER(C.c_int64)), cb, None, 4)
    opts = engine.Opts(0, engine._SYM["hash"], 0, 0, 0, 0)
    plot = np.zeros(engine.PLOT_CELLS, dtype=np.int64)
    st = engine.Stats()
    buf = C.create_string_buffer(512)
    lib.smg_hetmers_run_source.argtypes = [C.c_void_p, C.POINTER(engine.Opts), C.c_void_p, C.POINTER(engine.Stats), C.c_char_p, C.c_size_t]
    old = os.environ.get("SMG_SHARD_LIMIT")
    if limit is not None:
        os.environ["SMG_SHARD_LIMIT"] = str(limit)
    try:
        rc = lib.smg_hetmers_run_source(C.byref(src), C.byref(opts), plot.ctypes.data, C.byref(st), buf, 512)
    finally:
        if limit is not None:
            if old is None:
                del os.environ["SMG_SHARD_LIMIT"]
            else:
                os.environ["SMG_SHARD_LIMIT"] = old
    assert rc == 0, buf.value
    return plot.reshape(1001, 501), st.asdict()


def test_table_source_with_the_reference_binary_as_judge(tmp_path):
    """the twin of the big test below at 1/50 of its size: device generated table -> smg_hetmers_run_source (records
    pulled from the device through the read callback) in 1, 2 and 3 shards, against the REFERENCE binary"""
    import torch
    from conftest import REF_BIN
    from smudgeplot_amd import synth_device
    if not os.path.exists(REF_BIN):
        pytest.skip("prebuilt reference binary not present")
    k = 31
    dev = torch.device("cuda:0")
    tk, tc = synth_device.diploid_table(35_000_000, k=k, het=0.01, cov=50.0, L=10, seed=5, device=dev)
    n = tc.numel()
    rec, index = _records_on_device(tk, tc, k)
    plots = [_run_source_from_device(rec, index, k, n, limit)[0] for limit in (None, n // 2 + 1, n // 3 + 1)]
    assert np.array_equal(plots[0], plots[1]) and np.array_equal(plots[0], plots[2])
    synth.write_u64_table(str(tmp_path / "t"), tk.cpu().numpy().view(np.uint64), tc.cpu().numpy().view(np.uint16), k, ibyte=3, nparts=4)
    r = subprocess.run([REF_BIN, "-e10", f"-T{min(64, os.cpu_count() or 1)}", "-oref", "t.ktab"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "ref.smu").read_text() == engine.smu_text(plots[0])
    # and the drop-in executable on the same files (parts streamed from disk by 8 readers)
    r = subprocess.run([HETMERS_BIN, "-e10", "-T4", "-v", "-ogpu", "t.ktab"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "gpu.smu").read_text() == (tmp_path / "ref.smu").read_text()


def test_a_table_of_more_than_2_to_the_32_entries():
    """4.4e9 entries (synthetic diploid 1.75 Gbp, k=31): more than a shard can index.  The records live in a device
    tensor and reach the engine through the table-source callback, so the host holds nothing.  Checked: the automatic
    2-shard run equals a 3-shard run (shard invariance) and the out-of-core run in four sequential shards; the pair count is 1.75x the 1 Gbp table's within 2 %; and the
    SHAPE of the plot is that of its 1/50 twin (the 35 Mbp table of the same generator, which the test above runs
    through the reference binary): every well filled cell holds 50x the twin's count within 5 sigma of the twin's
    counting noise."""
    import torch
    from smudgeplot_amd import synth_device
    dev = torch.device("cuda:0")
    free, total = torch.cuda.mem_get_info(dev)
    if total < 250e9:
        pytest.skip("needs a 288 GB device")
    k = 31
    tk, tc = synth_device.diploid_table(35_000_000, k=k, het=0.01, cov=50.0, L=10, seed=5, device=dev)
    rec, index = _records_on_device(tk, tc, k)
    twin, _ = _run_source_from_device(rec, index, k, tc.numel())
    del tk, tc, rec
    torch.cuda.empty_cache()
    tk, tc = synth_device.diploid_table(1_750_000_000, k=k, het=0.01, cov=50.0, L=10, seed=1, device=dev)
    n = tc.numel()
    assert n > (1 << 32)
    rec, index = _records_on_device(tk, tc, k)
    del tk, tc
    torch.cuda.empty_cache()
    p2, st2 = _run_source_from_device(rec, index, k, n)
    p3, st3 = _run_source_from_device(rec, index, k, n, limit=n // 3 + 1)
    assert st2["nels"] == n and st2["path"] == 1
    assert np.array_equal(p2, p3)
    # ... and equals the OUT-OF-CORE run of the same table: four shards one after the other, the records pulled twice, no
    # candidate map, every request looked up from a sorted list -- another protocol and other kernels for the second half
    # of the computation, the same histogram cell for cell
    os.environ["SMG_SEQUENTIAL_SHARDS"] = "4"
    try:
        p4, st4 = _run_source_from_device(rec, index, k, n)
    finally:
        del os.environ["SMG_SEQUENTIAL_SHARDS"]
    assert st4["nels"] == n and np.array_equal(p2, p4)
    pairs = int(p2.sum())
    assert 0.98 < pairs / (458466309 * 1.75) < 1.02, pairs
    full = twin >= 20000                               # (a cell counts a pair twice: sigma = sqrt(2 c))
    assert full.sum() > 50
    z = (p2[full] / 50.0 - twin[full]) / np.sqrt(2.0 * twin[full] * (1 + 1 / 50.0))
    assert np.abs(z).max() < 5, float(np.abs(z).max())
    assert 0.99 < p2.sum() / (50.0 * twin.sum()) < 1.01


def test_exact_proof_is_not_fooled_by_a_signature_twin():
    """A closed table in which ONE complement is missing, while a k-mer with the same directory bucket, the same 16-bit
    look-up signature and the same count stands where it would be (its SNP partner in the last base).  The look-ups of
    the exact proof must compare the k-mer itself, not just signature and count: the table is not closed, the general
    path has to run and give the reference's (= the brute force) answer."""
    k = 31
    keys, cnt = synth.diploid_table_u64(3000, k=k, seed=77, het_frac=0.3, cov=30, L=5)
    rc = ktab.revcomp_u64(keys, k)
    top = keys >> np.uint64(39)
    uniq_top = np.ones(len(keys), bool)
    uniq_top[1:] &= top[1:] != top[:-1]
    uniq_top[:-1] &= top[:-1] != top[1:]
    cand = np.flatnonzero((keys < rc) & uniq_top[np.searchsorted(keys, rc)])
    for i in cand:
        x, y = keys[i], rc[i]
        yp = y ^ np.uint64(1 << 2)                                  # last base changed: same leading 25 bits
        xp = ktab.revcomp_u64(np.array([yp], np.uint64), k)[0]
        if yp in keys or xp in keys or yp == xp:
            continue
        c = cnt[i]
        keep = keys != y
        nk = np.concatenate([keys[keep], np.array([yp, xp], np.uint64)])
        nc = np.concatenate([cnt[keep], np.array([c, c], np.uint16)])
        o = np.argsort(nk)
        nk, nc = nk[o], nc[o]
        break
    else:
        pytest.fail("no suitable entry in the synthetic table")
    packed = ktab.u64_to_packed(nk, k)
    want = brute.hetmers_plot(packed, nc, k)
    for mode in ("exact", "hash"):
        plot, st = engine.hetmers_run(table_from(packed, nc, k), symcheck=mode)
        assert st["path"] == 2, mode
        assert np.array_equal(plot, want), mode


_BRUTE = {}


def _brute_cached(k, m, seed, packed, cnt):
    """the oracle's plot of a synthetic table, computed once per table (eight switch settings share it)"""
    if (k, m, seed) not in _BRUTE:
        _BRUTE[(k, m, seed)] = brute.hetmers_plot(packed, cnt, k)
    return _BRUTE[(k, m, seed)]


@pytest.mark.parametrize("env", [{"SMG_ONE_BIT_MAP": "1"}, {"SMG_BM_BITS": "30"},
                                 {"SMG_SIG": "1"}, {"SMG_SIG": "0", "SMG_BM_BITS": "30"},
                                 {"SMG_BM_BITS": "24", "SMG_ONE_BIT_MAP": "1"}, {"SMG_DIR_PER": "8"}, {"SMG_DIR_PER": "200"},
                                 {"SMG_NO_FILTER": "1"}, {"SMG_PROBE_X": "1"}, {"SMG_PROBE_X": "1", "SMG_ONE_BIT_MAP": "1"},
                                 {"SMG_PROBE_X": "0"}, {"SMG_PROBE_X": "1", "SMG_PX_ONE_XCC": "1"}, {"SMG_NO_INDEX_DIR": "1"}])
@pytest.mark.parametrize("k,m,seed", [(31, 60000, 21), (27, 40000, 22), (24, 30000, 23)])
def test_every_variant_of_the_lookup_chain_gives_the_same_plot(k, m, seed, env, monkeypatch):
    """the test hooks of the look-up chain (smg_hetmers.hip, test_hook): every form the engine picks from a table's size, k or
    counts -- two-bit / one-bit map, the 30-bit map of an exchanging run, signatures, directory bucket size, pass 1's own
    directory, no filter at all, either probe kernel -- forced on one small table: one answer"""
    packed, cnt = synth.adversarial_table(k, m, 4, seed, low_complexity=60, dense=1)
    tab = table_from(packed, cnt, k)
    base, st0 = engine.hetmers_run(tab, symcheck="hash")
    assert np.array_equal(base, _brute_cached(k, m, seed, packed, cnt)), "the default chain against the numpy oracle"
    for name, val in env.items():
        monkeypatch.setenv(name, val)
    plot, st = engine.hetmers_run(tab, symcheck="hash")
    assert np.array_equal(plot, base), env
    assert st["path"] == st0["path"] == 1 and st["nemitted"] == st0["nemitted"]
    if "SMG_NO_FILTER" in env:
        assert st["nrequests"] == st["nemitted"]
    elif env == {"SMG_ONE_BIT_MAP": "1"}:
        assert st["nrequests"] >= st0["nrequests"]          # the second bit can only drop more


def _polyploid_vs_reference(tmp_path, k, tk, tc, L, min_entries, shards, min_rows=1000):
    """device-generated polyploid table -> (i) the engine on the resident table, as a bench step runs it, (ii) the drop-in
    executable on the FastK files, on one GPU and as `shards` prefix shards -- all byte identical to the REFERENCE binary"""
    import torch
    from conftest import REF_BIN
    from smudgeplot_amd import sharded, synth_device
    if not os.path.exists(REF_BIN):
        pytest.skip("prebuilt reference binary not present")
    n = tc.numel()
    assert n >= min_entries, n
    plot, st = sharded.hetmers_sharded(k, tk.reshape(-1), tc, symcheck="hash", eng=sharded.TorchEngine(tk.device))
    assert st["path"] == 1                                  # the generator's table is closed: the symmetry proof holds
    eng_smu = engine.smu_text(plot.cpu().numpy().reshape(1001, 501))
    synth_device.write_table_from_device(str(tmp_path / "t"), tk, tc, k, nparts=8)
    del tk, tc, plot
    torch.cuda.empty_cache()
    r = subprocess.run([REF_BIN, f"-e{L}", f"-T{min(64, os.cpu_count() or 1)}", "-oref", "t.ktab"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = (tmp_path / "ref.smu").read_text()
    assert want.count("\n") > min_rows
    assert eng_smu == want
    for gpus, out in ((1, "g1"), (shards, "gs"), (-3, "g3seq")):
        env = dict(os.environ, SMUDGEPLOT_GPUS=str(max(gpus, 1)))
        if gpus > 1:
            env["SMG_VIRTUAL_SHARDS"] = str(gpus)
        if gpus < 0:                                     # out of core: three shards one after the other, the table read twice
            env["SMG_SEQUENTIAL_SHARDS"] = str(-gpus)
        q = subprocess.run([HETMERS_BIN, f"-e{L}", "-T8", "-v", f"-o{out}", "t.ktab"], cwd=tmp_path, capture_output=True,
                           text=True, env=env)
        assert q.returncode == 0, q.stderr
        assert (gpus >= 0) or "out of core" in q.stderr
        assert (tmp_path / f"{out}.smu").read_text() == want, gpus
    return want


def test_octoploid_graded_table_of_1e8_entries_vs_reference_binary(tmp_path):
    """BASELINE configs[3] stand-in at >= 1e8 entries: 8 haplotypes with GRADED divergences (variant sets carried by 1, 2, 3
    and 4 of the 8 haplotypes: the smudges AAAAAAAB .. AAAABBBB; k-mers that span sites of two sets form the groups of
    3-4 one-away neighbours of a polyploid table) -- `bench.py --workload octoploid` at a sixth of its genome"""
    import torch
    from smudgeplot_amd import synth_device
    k, L = 31, 8
    tk, tc = synth_device.polyploid_table_graded(34_000_000, ploidy=8, cov_hap=14.0, k=k, L=L, seed=4, device=torch.device("cuda:0"))
    want = _polyploid_vs_reference(tmp_path, k, tk, tc, L, 100_000_000, 8)
    # the smudges that make it an octoploid: pairs whose minor share is about 1/8, 2/8, 3/8 and 4/8 of the pair's coverage
    rows = np.array([[int(v) for v in l.split("\t")] for l in want.splitlines()])
    share = np.round(8.0 * rows[:, 0] / (rows[:, 0] + rows[:, 1])).astype(int)
    for m in (1, 2, 3, 4):
        assert rows[share == m, 2].sum() > 0.05 * rows[:, 2].sum(), m


def test_hexaploid_k51_table_of_1e8_entries_vs_reference_binary(tmp_path):
    """BASELINE configs[4] stand-in at >= 1e8 entries: six graded haplotypes, k = 51 (two-word k-mers, the kl_part<2> /
    kl_probe<..,2> chain) -- `bench.py --workload hexaploid` at a thirteenth of its genome"""
    import torch
    from smudgeplot_amd import synth_device
    k, L = 51, 5
    tk, tc = synth_device.polyploid_table_wide(30_000_000, ploidy=6, cov_hap=10.0, k=k, L=L, seed=5, device=torch.device("cuda:0"))
    _polyploid_vs_reference(tmp_path, k, tk, tc, L, 100_000_000, 6, min_rows=300)


@pytest.mark.parametrize("k,G,rep", [(31, 3_000_000, 0.0), (31, 3_000_000, 0.05), (27, 1_000_000, 0.0), (51, 1_500_000, 0.0), (12, 400_000, 0.0)])
def test_the_tables_prefix_index_as_lookup_directory_changes_nothing(k, G, rep):
    """smg_engine_set_prefix_index: the FastK prefix index (entries up to every 3-byte prefix, libfastk.c:841) handed
    over with a bound table replaces the directory pass 1 used to build -- same plot, same requests, and the reference
    numpy oracle agrees on the small case"""
    import torch
    from smudgeplot_amd import sharded, synth_device
    dev = torch.device("cuda:0")
    if k <= 31:
        tk, tc = synth_device.diploid_table(G, k=k, het=0.01, cov=50.0, L=10, seed=9, device=dev, repeats=rep)
    else:
        tk, tc = synth_device.diploid_table_wide(G, k=k, het=0.01, cov=50.0, L=10, seed=9, device=dev)
    kw0 = tk if tk.dim() == 1 else tk[:, 0]
    index = torch.cumsum(torch.bincount((kw0 >> 40) & 0xFFFFFF, minlength=1 << 24), 0)
    flat = tk.reshape(-1)
    plots, reqs = [], []
    for use in (False, True):
        eng = sharded.TorchEngine(dev)
        eng.bind(k, flat, tc, index=index if use else None)
        for _ in range(2):                                   # (twice: the engine keeps what it knows about a bound table)
            plot, st = sharded.hetmers_sharded(k, flat, tc, symcheck="hash", eng=eng, prebound=True)
            assert st["path"] == 1
            plots.append(plot.cpu().numpy().copy()); reqs.append(st["nrequests"])
    assert all(np.array_equal(plots[0], p) for p in plots[1:])
    assert len(set(reqs)) == 1 and plots[0].sum() > 0
    if k == 12:
        keys = tk.cpu().numpy().view(np.uint64)
        assert np.array_equal(plots[0].reshape(1001, 501), brute.hetmers_plot(ktab.u64_to_packed(keys, k), tc.cpu().numpy().view(np.uint16), k))


def test_a_rerun_on_the_same_engine_takes_nothing_for_granted_that_it_does_not_check():
    """smg_engine_run on a table the engine has run before queues the whole run without reading anything back in between and
    sizes it from the last run's counts; every count is checked at the end.  A table that changed in place between two
    runs (equal size, equal pointers) must still get the right answer: here (i) the counts of one k-mer and its complement
    move so that their pairs disappear (closed table, other request count), (ii) one count moves alone (the table is no
    longer closed: general path)."""
    import torch
    from smudgeplot_amd import sharded
    k = 31
    keys, cnt = synth.diploid_table_u64(30000, k=k, seed=91, het_frac=0.4, cov=30, L=5)
    dev = torch.device("cuda:0")
    tk = torch.from_numpy(keys.view(np.int64).copy()).to(dev)
    tc = torch.from_numpy(cnt.view(np.int16).copy()).to(dev)
    eng = sharded.TorchEngine(dev)
    eng.bind(k, tk, tc)
    want = brute.hetmers_plot(ktab.u64_to_packed(keys, k), cnt, k)
    for _ in range(3):                                        # an engine is reused: every run reads its own counts
        plot, st = sharded.hetmers_sharded(k, tk, tc, symcheck="hash", eng=eng, prebound=True)
        assert st["path"] == 1 and np.array_equal(plot.cpu().numpy().reshape(1001, 501), want)
    rc = ktab.revcomp_u64(keys, k)
    pos = {int(v): i for i, v in enumerate(keys)}
    # an entry that has a pair, and its complement: counts up to 900 -> no sum stays <= 1000 next to them
    j = int(np.nonzero(rc != keys)[0][len(keys) // 3])
    cnt2 = cnt.copy(); cnt2[j] = 900; cnt2[pos[int(rc[j])]] = 900
    tc.copy_(torch.from_numpy(cnt2.view(np.int16).copy()))
    plot, st = sharded.hetmers_sharded(k, tk, tc, symcheck="hash", eng=eng, prebound=True)
    assert st["path"] == 1 and np.array_equal(plot.cpu().numpy().reshape(1001, 501), brute.hetmers_plot(ktab.u64_to_packed(keys, k), cnt2, k))
    cnt3 = cnt2.copy(); cnt3[j] = 17                           # ... and now the table is not closed any more
    tc.copy_(torch.from_numpy(cnt3.view(np.int16).copy()))
    plot, st = sharded.hetmers_sharded(k, tk, tc, symcheck="hash", eng=eng, prebound=True)
    assert st["path"] == 2 and np.array_equal(plot.cpu().numpy().reshape(1001, 501), brute.hetmers_plot(ktab.u64_to_packed(keys, k), cnt3, k))
    tc.copy_(torch.from_numpy(cnt.view(np.int16).copy()))    # back to the first table
    for _ in range(2):
        plot, st = sharded.hetmers_sharded(k, tk, tc, symcheck="hash", eng=eng, prebound=True)
        assert st["path"] == 1 and np.array_equal(plot.cpu().numpy().reshape(1001, 501), want)


@pytest.mark.parametrize("nshards", [2, 5, 13])
@pytest.mark.parametrize("name", ["k31_i1", "k31_i3_p4", "k21_i2_p2", "k51_i1_p3", "k32_i1_p2", "k65_i1", "k17_i1"])
@pytest.mark.parametrize("mode", ["hash", "exact"])
def test_out_of_core_shards_one_after_the_other(name, nshards, mode, monkeypatch):
    """a table whose shards do not fit the device together is run shard after shard, the table read twice (the reference
    streams what its cache does not hold, PloidyPlot.c:931-1038): only a code byte per entry and the requests stay between
    the two rounds.  SMG_SEQUENTIAL_SHARDS forces the mode on the reference's golden vectors."""
    g = load_golden(name)
    monkeypatch.setenv("SMG_SEQUENTIAL_SHARDS", str(nshards))
    plot, st = engine.hetmers_run(make_table(g), symcheck=mode)
    assert engine.smu_text(plot) == g["smu"]
    assert st["nels"] == len(g["counts"]) and st["path"] == 1


def test_out_of_core_mode_is_chosen_from_the_free_memory_and_says_what_it_cannot_do(monkeypatch, tmp_path):
    """SMG_HBM_LIMIT (bytes) stands in for the free device memory: the executable picks the number of shards itself and says
    so; a table that is not closed gets a precise refusal instead of a wrong answer (the extract leg runs out of core since
    round 5: tests/test_extract.py; a RAW table and k > 85 since round 6: test_raw_table_out_of_core, test_out_of_core_above_k_85)"""
    g = load_golden("k31_i3_p4")
    n = len(g["counts"])
    ktab.write_ktab(str(tmp_path / "t"), g["k"], g["packed"], g["counts"], ibyte=g["ibyte"], nparts=g["nparts"])
    # all together: n * 16 bytes + the candidate map; a code byte + requests of every entry (3.9 bytes) + a third of the table (16 bytes per entry) must fit
    env = dict(os.environ, SMG_HBM_LIMIT=str(int(n * 3.4 + n / 3 * 22 + 1000)))
    r = subprocess.run([HETMERS_BIN, "-oout", f"-e{g['L']}", "-T4", "-v", "t.ktab"], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "prefix shards one after the other" in r.stderr and "out of core" in r.stderr
    assert (tmp_path / "out.smu").read_text() == g["smu"]
    r = subprocess.run([HETMERS_BIN, "-oout2", f"-e{g['L']}", "-T4", "t.ktab"], cwd=tmp_path, capture_output=True, text=True,
                       env=dict(os.environ, SMG_HBM_LIMIT=str(int(n * 3.0))))
    assert r.returncode == 1 and "does not fit the device even shard by shard" in r.stderr
    monkeypatch.setenv("SMG_SEQUENTIAL_SHARDS", "3")
    packed, cnt = synth.adversarial_table(31, 3000, 4, seed=5, low_complexity=40, dense=1)
    keep = np.ones(len(cnt), bool); keep[len(cnt) // 2] = False
    with pytest.raises(engine.EngineError, match="not closed under reverse complement"):
        engine.hetmers_run(table_from(packed[keep], cnt[keep], 31), symcheck="hash")


@pytest.mark.parametrize("shards", [2, 3, 5])
@pytest.mark.parametrize("name", ["k31_i1", "k31_i3_p4", "k21_i2_p2", "k51_i1_p3", "k32_i1_p2", "k65_i1", "k17_i1"])
def test_raw_table_out_of_core(name, shards, monkeypatch):
    """PloidyPlot.c:931-1038 + 1381-1414: the reference conditions and streams a table of any size.  A RAW table (canonical
    k-mers, some below the threshold) that does not fit the device: trimmed and closed under reverse complement shard by shard
    (host_condition_sequential: two sweeps over the part files, the conditioned shards kept in host memory), then run out of
    core from there -- the plot must be the oracle's on the numpy-conditioned table, whatever the number of shards."""
    k, L, (rp, rcnt), (cp, cc) = _golden_raw(name)
    want = brute.hetmers_plot(cp, cc, k)
    assert want.sum() > 0
    monkeypatch.setenv("SMG_SEQUENTIAL_SHARDS", str(shards))
    for mode in ("hash", "exact"):
        plot, st = engine.hetmers_run(table_from(rp, rcnt, k), symcheck=mode, condition=engine.COND_TRIM | engine.COND_SYMM, ethresh=L)
        assert st["path"] == 1 and st["nels"] == len(cc), (name, shards, mode)
        assert np.array_equal(plot, want), (name, shards, mode)
    # trim only (a table that is closed but not trimmed), symmetrise only
    sp, sc = ktab.symmetrize(rp, rcnt, k)
    keep = sc >= L
    plot, st = engine.hetmers_run(table_from(sp, sc, k), symcheck="hash", condition=engine.COND_TRIM, ethresh=L)
    assert np.array_equal(plot, brute.hetmers_plot(sp[keep], sc[keep], k)) and st["nels"] == int(keep.sum())
    keep = rcnt >= L
    plot, st = engine.hetmers_run(table_from(rp[keep], rcnt[keep], k), symcheck="hash", condition=engine.COND_SYMM, ethresh=L)
    assert np.array_equal(plot, want) and st["nels"] == len(cc)


@pytest.mark.parametrize("shards", [2, 3, 5])
@pytest.mark.parametrize("name", ["k100_i1", "k100_wrap"])
def test_out_of_core_above_k_85(name, shards, monkeypatch):
    """k > 85 (the counted kernels: real uint8 degrees with the reference's wrap, PloidyPlot.c:163, 535) out of core: what a shard
    leaves behind between its two rounds is its degree bytes; the goldens are the reference binary's, k100_wrap is the table on
    which a degree really wraps.  Fresh tables and a raw one against the oracle as well."""
    if name not in golden_names():
        pytest.skip("no such golden")
    g = load_golden(name)
    monkeypatch.setenv("SMG_SEQUENTIAL_SHARDS", str(shards))
    for mode in ("hash", "exact"):
        plot, st = engine.hetmers_run(make_table(g), symcheck=mode)
        assert st["path"] == 1 and engine.smu_text(plot) == g["smu"], (name, shards, mode)
    for k, seed in ((96, 3), (128, 4)):
        packed, cnt = synth.adversarial_table(k, 1500, 4, seed, low_complexity=60, dense=1)
        plot, st = engine.hetmers_run(table_from(packed, cnt, k), symcheck="hash")
        assert np.array_equal(plot, brute.hetmers_plot(packed, cnt, k)), (k, shards)
    (rp, rcnt), (cp, cc) = _raw_table(97, 12, 5)
    plot, st = engine.hetmers_run(table_from(rp, rcnt, 97), symcheck="hash", condition=engine.COND_TRIM | engine.COND_SYMM, ethresh=5)
    assert np.array_equal(plot, brute.hetmers_plot(cp, cc, 97)) and st["nels"] == len(cc)


def test_raw_table_out_of_core_through_the_executable_picks_its_shards_from_the_memory(tmp_path):
    """the drop-in on a raw table with SMG_HBM_LIMIT standing in for a small device: it says that it conditions out of core, and the
    .smu is the reference binary's on the numpy-conditioned table"""
    from conftest import REF_BIN
    k, L, (rp, rcnt), (cp, cc) = _golden_raw("k31_i3_p4")
    ktab.write_ktab(str(tmp_path / "raw"), k, rp, rcnt, ibyte=3, nparts=3)
    ktab.write_ktab(str(tmp_path / "cond"), k, cp, cc, ibyte=3, nparts=2)
    n2 = 2 * len(rcnt)
    env = dict(os.environ, SMG_HBM_LIMIT=str(int(n2 * 3.4 + n2 / 3 * 54 + 1000)))      # a third of the closed table at a time
    r = subprocess.run([HETMERS_BIN, f"-e{L}", "-T4", "-v", "-ogpu", "raw"], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "prefix shards one after the other" in r.stderr and "conditioned out of core" in r.stderr
    q = subprocess.run([REF_BIN, f"-e{L}", "-T4", "-oref", "cond"], cwd=tmp_path, capture_output=True, text=True)
    assert q.returncode == 0, q.stderr
    assert (tmp_path / "gpu.smu").read_text() == (tmp_path / "ref.smu").read_text() != ""


# ---- the bench tables at FULL size against the reference binary's output (tests/golden/bench_*.smu) ------------------------
def _bench_golden(workload):
    import json
    tj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_tables.json")
    if not os.path.exists(tj):
        pytest.skip("tests/golden/bench_tables.json not committed")
    g = json.load(open(tj)).get(workload)
    if not g:
        pytest.skip(f"no golden for the {workload} bench table")
    g["smu"] = open(os.path.join(os.path.dirname(tj), g["smu_file"])).read()
    return g


@pytest.mark.parametrize("workload", ["uniform", "octoploid", "hexaploid", "repeats", "uniform_k30_1000Mbp", "uniform_k51_500Mbp", "uniform_k51_1000Mbp"])
def test_full_size_bench_table_vs_reference_golden(workload, tmp_path):
    """BASELINE.md section 3's gate where the driver can see it: the table `bench.py --workload <w>` times (BASELINE
    configs[2]: 2 535 258 108 entries at k = 31; the octoploid / hexaploid k = 51 stand-ins of configs[3] / [4]; the repeats
    table), generated here by the bench's own generator and identified by its checksum, must give the .smu that the
    REFERENCE binary (PloidyPlot.c:1603-1617) wrote for it -- tests/golden/bench_<w>.smu, made by
    tools/make_bench_goldens.py on the GPU box.  Three ways: the engine on the resident table (what a bench step runs), and --
    for the headline table -- the drop-in executable on the table's files as 8 virtual prefix shards (the N > 1 protocol on
    one device) and out of core (4 shards one after the other)."""
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from smudgeplot_amd import sharded, synth_device
    g = _bench_golden(workload)              # (the key of the golden: a workload, or <workload>_k<k>_<genome>Mbp for the k = 30 / 51 lines)
    workload = g["workload"]
    dev = torch.device("cuda:0")
    if torch.cuda.mem_get_info(dev)[1] < 250e9:
        pytest.skip("needs a 288 GB device")
    k = g["k"]
    keys, cnt, L, _ = bench.make_table(workload, g["genome"], k, dev)
    n = cnt.numel()
    assert n == g["entries"]
    assert synth_device.table_hash_text(n, *synth_device.table_hash(keys, cnt)) == g["table_hash"]
    index = torch.cumsum(torch.bincount(((keys if keys.dim() == 1 else keys[:, 0]) >> 40) & 0xFFFFFF, minlength=1 << 24), 0)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()        # (the generator's scratch goes back to the driver: the engine allocates outside torch's pool, as in bench.py)
    eng = sharded.TorchEngine(dev)
    eng.bind(k, keys.reshape(-1), cnt, index=index)
    for _ in range(2):          # (an engine is reused by every bench step: the second run must give the same)
        plot, st = sharded.hetmers_sharded(k, keys.reshape(-1), cnt, symcheck="hash", eng=eng, prebound=True)
        assert st["path"] == 1
        assert engine.smu_text(plot.cpu().numpy().reshape(1001, 501)) == g["smu"]
    del eng, plot, index
    if workload != "uniform" or k != 31:
        return
    synth_device.write_table_from_device(str(tmp_path / "t"), keys, cnt, k, nparts=4)
    del keys, cnt
    torch.cuda.empty_cache()
    for name, env in (("vs8", {"SMG_VIRTUAL_SHARDS": "8"}), ("seq4", {"SMG_SEQUENTIAL_SHARDS": "4"}), ("one", {})):
        r = subprocess.run([HETMERS_BIN, f"-e{L}", "-T4", "-o" + name, "t.ktab"], cwd=tmp_path, capture_output=True, text=True,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        assert (tmp_path / (name + ".smu")).read_text() == g["smu"], name
