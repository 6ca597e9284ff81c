"""`extract_kmer_pairs` (the first "next" row of the scope table): the pairs behind
annotated pixels.  Golden vectors come from the REFERENCE binary (tests/golden/make_golden_extract.py);
a smudge file is compared as a sorted list of lines -- the reference writes lines in thread-schedule order."""
import glob
import json
import os
import subprocess

import numpy as np
import pytest

import brute
from conftest import GOLDEN, ROOT, load_golden, make_table
from smudgeplot_amd import engine, ktab, synth

EXTRACT_BIN = os.path.join(ROOT, "smudgeplot_amd", "bin", "extract_kmer_pairs")
REF_EXTRACT = os.path.join(ROOT, "oracle", "_ref", "extract_ref")


def extract_goldens():
    return sorted(os.path.basename(p)[len("extract_"):-5] for p in glob.glob(os.path.join(GOLDEN, "extract_*.json")))


def load_extract(name):
    j = json.load(open(os.path.join(GOLDEN, f"extract_{name}.json")))
    labels = {(int(b), int(a)): lab for b, a, _, lab in j["labels"]}
    lines = {lab: [l + "\n" for l in v] for lab, v in j["lines"].items()}
    return load_golden(name), labels, lines, j["labels"]


def write_sma(path, rows):
    with open(path, "w") as f:
        f.write("covB\tcovA\tfreq\tsmudge\n")
        for covb, cova, freq, lab in rows:
            f.write(f"{covb}\t{cova}\t{freq}\t{lab}\n")


# ---------------------------------------------------------------------------------------------- CPU

@pytest.mark.parametrize("name", extract_goldens())
def test_numpy_oracle_extract_matches_reference_golden(name):
    g, labels, lines, _ = load_extract(name)
    got = brute.extract_lines(g["packed"], g["counts"], g["k"], labels)
    assert got == lines
    assert sum(len(v) for v in lines.values()) > (500 if name != "k100_wrap" else 1)     # (the wrap table has one pixel)


def test_extract_golden_set_is_not_trivial():
    assert len(extract_goldens()) >= 5
    g, labels, lines, _ = load_extract("k31_i1")
    text = "".join(lines["1A1B"])
    assert "(a/" in text and "(c/" in text and "(g/" in text and "(t/" in text
    # every line: k bases + "(x/y)" at one position
    for l in lines["2A1B"]:
        assert len(l) == g["k"] + 5 and l.count("(") == 1


def run_x(args, cwd, env=None):
    return subprocess.run([EXTRACT_BIN, *args], cwd=cwd, capture_output=True, text=True, env=env)


def test_extract_usage_and_sma_errors(tmp_path):
    r = run_x([], tmp_path)
    assert r.returncode == 1 and r.stderr.startswith("\nUsage: extract_kmer_pairs  [-v] [-T<int(4)>] [-P<dir(/tmp)>]")
    assert " [-o<output>] [-e<int(4)>] <source>[.ktab] <smudges>[.sma]\n" in r.stderr
    r = run_x(["t"], tmp_path)
    assert r.returncode == 1 and "Usage: extract_kmer_pairs" in r.stderr
    r = run_x(["-x", "t", "s"], tmp_path)
    assert r.returncode == 1 and r.stderr == "extract_kmer_pairs: -x is an illegal option\n"
    r = run_x(["-e4", "t", "nosuch"], tmp_path)
    assert r.returncode == 1 and r.stderr == "\nextract_kmer_pairs: Could not open smudge file nosuch.sma"
    (tmp_path / "bad.sma").write_text("covB\tcovA\tfreq\tsmudge\n1\t2\tx\n")
    r = run_x(["-e4", "t", "bad.sma"], tmp_path)
    assert r.returncode == 1 and r.stderr == "extract_kmer_pairs: Cannot parse line '1\t2\tx\n'\n"
    (tmp_path / "lab.sma").write_text("h\n3\t4\t9\t1A2B\n")
    r = run_x(["-e4", "t", "lab"], tmp_path)
    assert r.returncode == 1 and r.stderr == "extract_kmer_pairs: 1A2B is not a valid smudge label'\n"
    (tmp_path / "pix.sma").write_text("h\n9\t4\t9\t1A1B\n")
    r = run_x(["-e4", "t", "pix"], tmp_path)
    assert r.returncode == 1 and r.stderr == "extract_kmer_pairs: (9,4) is not a valid pixel coordinate\n"
    # a good .sma: the smudge files are created before the table is opened, like the reference
    (tmp_path / "ok.sma").write_text("h\n20\t21\t9\t1A1B\n")
    r = run_x(["-e4", "-oout", "missing", "ok"], tmp_path)
    assert r.returncode == 1 and r.stderr == "extract_kmer_pairs: Cannot open k-mer table missing\n"
    assert (tmp_path / "out.1A1B.txt").exists()


def test_extract_without_gpu_fails_loudly(tmp_path):
    if engine.device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu tests")
    packed, cnt = synth.adversarial_table(31, 200, 6, seed=5)
    ktab.write_ktab(str(tmp_path / "t"), 31, packed, cnt, ibyte=1)
    (tmp_path / "ok.sma").write_text("h\n20\t21\t9\t1A1B\n")
    r = run_x(["-e6", "-v", "t", "ok"], tmp_path)
    assert r.returncode == 1
    assert r.stderr.endswith("extract_kmer_pairs: no HIP device available (this engine has no CPU fallback)\n")


# ---------------------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
@pytest.mark.parametrize("name", extract_goldens())
def test_engine_extract_matches_reference_golden(name):
    g, labels, lines, _ = load_extract(name)
    plot, got = engine.hetmers_extract(make_table(g), labels)
    assert engine.smu_text(plot) == g["smu"]
    assert {lab: sorted(v) for lab, v in got.items()} == lines


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["k31_i1", "k51_i1_p3", "k65_i1", "k100_i1", "k100_wrap"])
def test_extract_executable_matches_reference_golden(name, tmp_path):
    g, labels, lines, rows = load_extract(name)
    ktab.write_ktab(str(tmp_path / "t"), g["k"], g["packed"], g["counts"], ibyte=min(g["ibyte"], 2), nparts=g["nparts"])
    write_sma(tmp_path / "s.sma", rows)
    r = run_x(["-oout", f"-e{g['L']}", "-T4", "-v", "t.ktab", "s.sma"], tmp_path)
    assert r.returncode == 0, r.stderr
    assert "  The input table is trimmed and symmetric\n" in r.stderr
    for lab, want in lines.items():
        assert sorted(open(tmp_path / f"out.{lab}.txt").readlines()) == want, lab


@pytest.mark.gpu
@pytest.mark.parametrize("k,seed", [(19, 1), (32, 2), (33, 3), (47, 4), (64, 5), (85, 6), (86, 7), (97, 8), (128, 9)])
def test_extract_fresh_tables_vs_oracle(k, seed):
    packed, cnt = synth.adversarial_table(k, 2000, 4, seed, low_complexity=100, dense=1)
    want_plot = brute.hetmers_plot(packed, cnt, k)
    s, m = np.nonzero(want_plot[:, :500])
    labels = {(int(mm), int(ss - mm)): ("1A1B", "3A1B", "2A2B")[(ss + mm) % 3]
              for ss, mm in zip(s.tolist(), m.tolist()) if (ss * 7 + mm) % 4}
    want = brute.extract_lines(packed, cnt, k, labels)
    plot, got = engine.hetmers_extract(make_table(dict(packed=packed, counts=cnt, k=k, ibyte=1, nparts=1)), labels)
    assert np.array_equal(plot, want_plot)
    assert {lab: sorted(v) for lab, v in got.items()} == want
    assert sum(len(v) for v in want.values()) > 300


@pytest.mark.gpu
@pytest.mark.parametrize("seq", [0, 3], ids=["in-core", "out-of-core"])
def test_extract_on_raw_table_conditions_first(seq, tmp_path, monkeypatch):
    """raw canonical table with erroneous k-mers: conditioned on the device, then extracted; compared with
    the reference extract binary (or the numpy oracle) on the table conditioned by numpy -- also when the table "does not
    fit" and is conditioned and run shard by shard (round 6)"""
    if seq:
        monkeypatch.setenv("SMG_SEQUENTIAL_SHARDS", str(seq))
    k, L = 31, 6
    packed, cnt = synth.adversarial_table(k, 2500, L, 77, low_complexity=100, dense=1)
    rc = ktab.revcomp_packed(packed, k)
    canon = np.array([bytes(a) <= bytes(b) for a, b in zip(packed, rc)])
    rp, rcnt = packed[canon], cnt[canon].copy()
    rng = np.random.default_rng(3)
    low = rng.random(len(rcnt)) < 0.2
    rcnt[low] = rng.integers(1, L, size=int(low.sum()))
    keep = rcnt >= L
    cp, cc = ktab.symmetrize(rp[keep], rcnt[keep], k)
    plot = brute.hetmers_plot(cp, cc, k)
    s, m = np.nonzero(plot[:, :500])
    rows = [(int(mm), int(ss - mm), int(plot[ss, mm]), "1A1B" if ss % 2 else "2A1B") for ss, mm in zip(s.tolist(), m.tolist())]
    want = brute.extract_lines(cp, cc, k, {(b, a): lab for b, a, _, lab in rows})
    ktab.write_ktab(str(tmp_path / "raw"), k, rp, rcnt, ibyte=1, nparts=2)
    write_sma(tmp_path / "s.sma", rows)
    r = run_x([f"-e{L}", "-v", "-ogpu", "raw", "s"], tmp_path)
    assert r.returncode == 0, r.stderr
    assert "  Making trimmed table symmetric\n" in r.stderr
    for lab, w in want.items():
        assert sorted(open(tmp_path / f"gpu.{lab}.txt").readlines()) == w
    if os.path.exists(REF_EXTRACT):
        ktab.write_ktab(str(tmp_path / "cond"), k, cp, cc, ibyte=1, nparts=2)
        q = subprocess.run([REF_EXTRACT, f"-e{L}", "-T4", "-oref", "cond", "s"], cwd=tmp_path, capture_output=True, text=True)
        assert q.returncode == 0, q.stderr
        for lab in want:
            assert sorted(open(tmp_path / f"ref.{lab}.txt").readlines()) == sorted(open(tmp_path / f"gpu.{lab}.txt").readlines())


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [3, 6])
@pytest.mark.parametrize("name", ["k31_i1", "k51_i1_p3", "k65_i1", "k100_i1", "k100_wrap"])
def test_extract_over_prefix_shards_matches_reference_golden(name, shards, monkeypatch):
    """PloidyList.c:1207-1583 has no size or device limit: the extract leg over a table that is cut into prefix shards
    (several GPUs, or more than 2^32 entries on one) -- every shard lists the pairs among its own entries, the host
    puts the lists together; same lines as the reference, whatever the cut"""
    g, labels, lines, _ = load_extract(name)
    monkeypatch.setenv("SMG_VIRTUAL_SHARDS", str(shards))
    plot, got = engine.hetmers_extract(make_table(g), labels)
    assert engine.smu_text(plot) == g["smu"]
    assert {lab: sorted(v) for lab, v in got.items()} == lines
    monkeypatch.delenv("SMG_VIRTUAL_SHARDS")
    monkeypatch.setenv("SMG_SHARD_LIMIT", str(len(g["counts"]) // shards + 1))          # the automatic shards of a big table
    plot, got = engine.hetmers_extract(make_table(g), labels)
    assert {lab: sorted(v) for lab, v in got.items()} == lines


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [2, 5])
@pytest.mark.parametrize("name", ["k31_i1", "k51_i1_p3", "k65_i1", "k100_i1", "k100_wrap"])       # (k > 85 out of core: round 6)
def test_extract_out_of_core_matches_reference_golden(name, shards, monkeypatch):
    """... and over a table that does not fit the device (prefix shards one after the other, the table read twice: smg_multi.hpp,
    host_run_sequential): every shard lists the pairs behind the labelled pixels while it is resident for its second round"""
    g, labels, lines, _ = load_extract(name)
    monkeypatch.setenv("SMG_SEQUENTIAL_SHARDS", str(shards))
    plot, got = engine.hetmers_extract(make_table(g), labels)
    assert engine.smu_text(plot) == g["smu"]
    assert {lab: sorted(v) for lab, v in got.items()} == lines
