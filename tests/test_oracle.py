"""CPU tests: the oracle (C restatement + numpy restatement) against the committed golden
vectors produced by the reference binary, and -- where the compiled reference is present --
against the reference itself on fresh random tables."""
import os
import subprocess

import numpy as np
import pytest

import brute
from conftest import ORACLE_BIN, REF_BIN, golden_names, load_golden
from smudgeplot_amd import ktab, synth


@pytest.mark.parametrize("name", golden_names())
def test_numpy_oracle_matches_reference_golden(name):
    g = load_golden(name)
    plot = brute.hetmers_plot(g["packed"], g["counts"], g["k"])
    assert brute.smu_text(plot) == g["smu"]


@pytest.mark.parametrize("name", golden_names())
def test_c_oracle_matches_reference_golden(name, tmp_path):
    g = load_golden(name)
    if g["ibyte"] == 3:
        pytest.skip("134 MB stub: covered by the numpy oracle and by test_reference_live")
    ktab.write_ktab(str(tmp_path / "t"), g["k"], g["packed"], g["counts"], ibyte=g["ibyte"],
                    nparts=g["nparts"])
    subprocess.run([ORACLE_BIN, f"-e{g['L']}", f"-o{tmp_path}/o", str(tmp_path / "t.ktab")], check=True)
    assert open(tmp_path / "o.smu").read() == g["smu"]


def test_golden_set_is_not_trivial():
    names = golden_names()
    assert len(names) >= 10
    ks = {load_golden(n)["k"] for n in names}
    assert {17, 31, 32, 51, 64, 65, 100} <= ks
    assert load_golden("k100_wrap")["smu"].strip() != ""      # the uint8 wrap changes the answer


def test_examine_decisions(tmp_path):
    """trim / symm probe (PloidyPlot.c:1167-1230) restated by the C oracle."""
    packed, cnt = synth.adversarial_table(31, 300, 6, seed=3)
    ktab.write_ktab(str(tmp_path / "a"), 31, packed, cnt, ibyte=1)
    out = subprocess.run([ORACLE_BIN, "-e6", "-x", str(tmp_path / "a")], check=True,
                         capture_output=True, text=True).stdout
    assert out.strip() == "trim=1 symm=1"
    out = subprocess.run([ORACLE_BIN, "-e7", "-x", str(tmp_path / "a")], check=True,
                         capture_output=True, text=True).stdout
    assert out.strip() == "trim=0 symm=1"
    # drop the complement of entry #1 -> not symmetric
    rc1 = ktab.revcomp_packed(packed[1:2], 31)[0]
    keep = ~(packed == rc1).all(axis=1)
    ktab.write_ktab(str(tmp_path / "b"), 31, packed[keep], cnt[keep], ibyte=1)
    out = subprocess.run([ORACLE_BIN, "-e6", "-x", str(tmp_path / "b")], check=True,
                         capture_output=True, text=True).stdout
    assert out.strip() == "trim=1 symm=0"


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="compiled reference not present")
@pytest.mark.parametrize("k,ibyte,nparts,L,T,seed", [
    (31, 1, 1, 5, 1, 101), (31, 2, 3, 5, 8, 102), (25, 1, 2, 4, 4, 103), (48, 2, 2, 4, 5, 104),
    (72, 1, 1, 4, 2, 105),
])
def test_reference_live(k, ibyte, nparts, L, T, seed, tmp_path):
    """three-way on fresh tables: reference binary == C oracle == numpy oracle"""
    packed, cnt = synth.adversarial_table(k, 2000, L, seed, low_complexity=120, dense=2)
    ktab.write_ktab(str(tmp_path / "t"), k, packed, cnt, ibyte=ibyte, nparts=nparts)
    r = subprocess.run([REF_BIN, f"-e{L}", f"-T{T}", "-oref", "t.ktab"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ref = open(tmp_path / "ref.smu").read()
    subprocess.run([ORACLE_BIN, f"-e{L}", f"-o{tmp_path}/orc", str(tmp_path / "t")], check=True)
    assert open(tmp_path / "orc.smu").read() == ref
    assert brute.smu_text(brute.hetmers_plot(packed, cnt, k)) == ref


def test_empty_and_tiny_tables():
    for n in (0, 1, 2):
        packed = ktab.pack_bases(np.zeros((n, 31), np.uint8) + np.arange(n, dtype=np.uint8)[:, None] % 4)
        plot = brute.hetmers_plot(packed, np.full(n, 10, np.uint16), 31)
        assert plot.sum() == 0
