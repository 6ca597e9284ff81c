"""bench.py --gpus N: every rank generates its own prefix shard (synth_device key_range), and the parity block of the bench
line (plot of the timed run vs the reference binary's golden .smu).  CPU only: the generators are torch code."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smudgeplot_amd import engine, synth_device as sd  # noqa: E402


def _shards(gen, world, **kw):
    ks, cs = [], []
    for r in range(world):
        k, c = gen(key_range=sd.key_range_of(r, world), **kw)
        ks.append(k.reshape(c.numel(), -1)); cs.append(c)
    return ks, cs


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("name", ["uniform", "repeats", "octoploid", "hexaploid", "k51"])
def test_rank_shards_are_the_table_of_one_rank(name, world):
    kw = dict(device="cpu")
    if name in ("uniform", "repeats"):
        gen, kw = sd.diploid_table, dict(kw, G=120000 if name == "uniform" else 250000, k=31, repeats=0.05 if name == "repeats" else 0.0)
    elif name == "octoploid":
        gen, kw = sd.polyploid_table_graded, dict(kw, G=20000, ploidy=8, k=31, seed=4)
    elif name == "hexaploid":
        gen, kw = sd.polyploid_table_wide, dict(kw, G=15000, ploidy=6, k=51, seed=5, max_chunk=2.0e4)     # (several chunks, cut ones too)
    else:
        gen, kw = sd.polyploid_table_wide, dict(kw, G=30000, ploidy=2, rates=(0.01,), cov_hap=25.0, k=51, L=10)
    k0, c0 = gen(**kw)
    k0 = k0.reshape(c0.numel(), -1)
    ks, cs = _shards(gen, world, **kw)
    assert torch.equal(torch.cat(ks), k0) and torch.equal(torch.cat(cs), c0)
    # every shard lies inside its range of the leading 16 key bits (so the cut values are the splitters of the sharded run)
    first = 0
    hk = hc = 0
    for r in range(world):
        lo, hi = sd.key_range_of(r, world)
        if cs[r].numel():
            lead = (ks[r][:, 0] >> 48) & 0xFFFF
            assert int(lead.min()) >= lo and int(lead.max()) < hi
        a, b = sd.table_hash(ks[r], cs[r], first_entry=first, piece=4099)
        hk, hc = (hk + a) & (2 ** 64 - 1), (hc + b) & (2 ** 64 - 1)
        first += cs[r].numel()
    assert (hk, hc) == sd.table_hash(k0, c0)          # the shards' checksums add up to the table's
    assert max(c.numel() for c in cs) < 1.25 * c0.numel() / world      # balanced (uniform random genome)


@pytest.mark.parametrize("workload,k,G", [("uniform", 51, 30000), ("uniform", 31, 60000), ("hexaploid", 51, 15000)])
def test_bench_make_table_is_one_table_whatever_the_number_of_ranks(workload, k, G):
    """bench.make_table itself (not only the generators behind it): the shards that N ranks generate are, in rank order, the
    table of N = 1 -- also for two-word k-mers of the uniform workload, where rounds 1-5 picked another generator at N = 1
    than at N > 1 (advisor finding, round 5): one table_hash whatever the launch"""
    import bench
    dev = torch.device("cpu")
    k0, c0, L0, _ = bench.make_table(workload, G, k, dev)
    k0 = k0.reshape(c0.numel(), -1)
    for world in (2, 4):
        ks, cs = [], []
        for r in range(world):
            kr, cr, L, _ = bench.make_table(workload, G, k, dev, key_range=sd.key_range_of(r, world))
            ks.append(kr.reshape(cr.numel(), -1)); cs.append(cr)
            assert L == L0
        assert torch.equal(torch.cat(ks), k0) and torch.equal(torch.cat(cs), c0)


def test_table_hash_sees_every_change():
    k, c = sd.diploid_table(50000, k=31, device="cpu")
    h = sd.table_hash(k, c)
    k2 = k.clone(); k2[1234] ^= 4
    c2 = c.clone(); c2[777] += 1
    sw = k.clone(); sw[[10, 11]] = sw[[11, 10]]
    assert len({h, sd.table_hash(k2, c), sd.table_hash(k, c2), sd.table_hash(sw, c)}) == 4


def test_parity_block_against_a_golden(tmp_path, monkeypatch):
    import bench
    plot = torch.zeros(engine.PLOT_CELLS, dtype=torch.int64)
    plot[30 * engine.PLOT_COLS + 12] = 8
    plot[51 * engine.PLOT_COLS + 25] = 2
    smu = engine.smu_text(plot.numpy().reshape(engine.PLOT_ROWS, engine.PLOT_COLS))
    assert smu == "12\t18\t8\n25\t26\t2\n"
    (tmp_path / "bench_uniform.smu").write_text(smu)
    key = bench.golden_key("uniform", 1000, 31)            # (not the default size: <workload>_k<k>_<genome>Mbp)
    tab = {key: {"workload": "uniform", "genome": 1000, "k": 31, "entries": 99, "table_hash": sd.table_hash_text(99, 5, 6), "smu_file": "bench_uniform.smu",
                       "smu_sha256": hashlib.sha256(smu.encode()).hexdigest()}}
    (tmp_path / "bench_tables.json").write_text(json.dumps(tab))
    monkeypatch.setattr(bench, "GOLDEN_DIR", str(tmp_path))
    p = bench.parity_against_golden("uniform", 1000, 31, 99, 5, 6, plot)
    assert p["ok"] is True and p["smu_sha256"] == tab[key]["smu_sha256"]
    plot[51 * engine.PLOT_COLS + 25] = 3
    assert bench.parity_against_golden("uniform", 1000, 31, 99, 5, 6, plot)["ok"] is False
    plot[51 * engine.PLOT_COLS + 25] = 2
    assert bench.parity_against_golden("uniform", 1000, 31, 99, 5, 7, plot)["ok"] is False        # another table
    assert bench.parity_against_golden("uniform", 2000, 31, 99, 5, 6, plot)["ok"] is None         # no golden at that size
    assert bench.parity_against_golden("octoploid", 1000, 31, 99, 5, 6, plot)["ok"] is None


def test_committed_bench_goldens_are_consistent():
    tj = os.path.join(ROOT, "tests", "golden", "bench_tables.json")
    if not os.path.exists(tj):
        pytest.skip("no bench goldens committed")
    import bench
    tab = json.load(open(tj))
    for wl, g in tab.items():
        assert bench.golden_key(g["workload"], g["genome"], g["k"]) == wl       # (the default size and k of a workload, or <w>_k<k>_<G>Mbp)
        smu = open(os.path.join(ROOT, "tests", "golden", g["smu_file"])).read()
        assert hashlib.sha256(smu.encode()).hexdigest() == g["smu_sha256"]
        assert g["engine_identical"] is True
        rows = [tuple(int(v) for v in ln.split("\t")) for ln in smu.splitlines()]
        assert rows == sorted(rows, key=lambda r: (r[0] + r[1], r[0])) and all(r[0] <= r[1] and r[0] < 500 and r[2] > 0 for r in rows)
