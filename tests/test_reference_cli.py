"""The drop-in boundary, pinned by the REFERENCE's own command line (src/smudgeplot/cli.py:18-72, 348-382).

* In the build container (/root/reference present, no GPU): the reference's `smudgeplot hetmers` / `extract` are run
  as they are, with smudgeplot_amd/bin first on PATH.  They must find OUR executables, hand them their argument
  vectors, and surface our exit status the way they surface the reference binary's: the run gets as far as the
  engine's "no HIP device" (there is no CPU fallback) after the reference's own messages.
* On the GPU box (/root/reference absent): tests/golden/cli_argv.json holds the argument vectors the reference
  command line produced (tests/golden/make_cli_argv.py); they are replayed against the drop-in executables and the
  `.smu` / smudge files are compared with the golden vectors made by the reference binaries."""
import json
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT, load_golden
from smudgeplot_amd import ktab

REF_SRC = "/root/reference/src"
BIN = os.path.join(ROOT, "smudgeplot_amd", "bin")
ARGV = json.load(open(os.path.join(GOLDEN, "cli_argv.json")))

RUN_REF_CLI = """
import importlib.metadata, sys
importlib.metadata.version = lambda name: "0.5.4"
sys.path.insert(0, %r)
import smudgeplot.cli as cli
cli.version = lambda name: "0.5.4"
sys.argv = ["smudgeplot"] + sys.argv[1:]
cli.main()
""" % REF_SRC


def _ref_cli(args, cwd):
    env = dict(os.environ, PATH=BIN + os.pathsep + os.environ.get("PATH", ""), MPLBACKEND="Agg")
    return subprocess.run([sys.executable, "-c", RUN_REF_CLI, *args], cwd=cwd, capture_output=True, text=True, env=env)


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference tree is only present in the build container")
def test_reference_cli_execs_the_drop_in_hetmers(tmp_path):
    g = load_golden("k31_i1")
    ktab.write_ktab(str(tmp_path / "t"), g["k"], g["packed"], g["counts"], ibyte=1, nparts=2)
    r = _ref_cli(["hetmers", "-L", str(g["L"]), "-t", "3", "-o", "pairs", "--verbose", "t.ktab"], tmp_path)
    assert f"Calling: {BIN}/hetmers -opairs -e{g['L']} -T3 -v t.ktab" in r.stderr, r.stderr
    assert "  The input table is trimmed and symmetric\n" in r.stderr          # our executable ran, and read the table
    from smudgeplot_amd import engine
    if engine.device_count() == 0:
        assert r.returncode != 0
        assert "hetmers: no HIP device available (this engine has no CPU fallback)" in r.stderr
        assert "CalledProcessError" in r.stderr                                 # run_binary's check=True, cli.py:72
        assert not (tmp_path / "pairs.smu").exists()
    else:
        assert r.returncode == 0, r.stderr
        assert (tmp_path / "pairs.smu").read_text() == g["smu"]


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference tree is only present in the build container")
def test_reference_cli_execs_the_drop_in_extract_and_reports_its_errors(tmp_path):
    r = _ref_cli(["extract", "-o", "x", "missing.ktab", "smudges.sma"], tmp_path)
    assert f"Calling: {BIN}/extract_kmer_pairs -ox -T4 missing.ktab smudges" in r.stderr, r.stderr
    assert r.returncode != 0 and "CalledProcessError" in r.stderr
    r = _ref_cli(["hetmers", "-L", "4", "missing_table"], tmp_path)
    assert "hetmers: Cannot open k-mer table missing_table" in r.stderr         # the reference's message, PloidyPlot.c:1351-1354
    assert r.returncode != 0


def test_recorded_argv_is_what_the_documented_grammar_says():
    assert ARGV["hetmers_default"]["exec"] == ["hetmers", "-oOUT", "-e5", "-T4", "TABLE.ktab"]
    assert ARGV["hetmers_threads_verbose"]["exec"] == ["hetmers", "-oOUT", "-e12", "-T16", "-v", "TABLE.ktab"]
    assert ARGV["hetmers_tmp"]["exec"] == ["hetmers", "-oOUT", "-e4", "-T4", "-PTMPDIR", "TABLE"]
    assert ARGV["extract_verbose"]["exec"] == ["extract_kmer_pairs", "-oOUT", "-T8", "-v", "TABLE.ktab", "SMA"]


def _subst(argv, table, out, sma=None, tmp=None):
    m = {"TABLE.ktab": table + ".ktab", "TABLE": table, "SMA": sma or "SMA", "TMPDIR": tmp or "."}
    res = []
    for a in argv[1:]:
        if a.startswith("-o"):
            res.append("-o" + out)
        elif a.startswith("-P"):
            res.append("-P" + m["TMPDIR"])
        else:
            res.append(m.get(a, a))
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["hetmers_default", "hetmers_threads_verbose", "hetmers_tmp"])
@pytest.mark.parametrize("name", ["k31_i1", "k51_i1_p3"])
def test_reference_cli_argv_replayed_on_the_drop_in_hetmers(case, name, tmp_path):
    g = load_golden(name)
    ktab.write_ktab(str(tmp_path / "t"), g["k"], g["packed"], g["counts"], ibyte=g["ibyte"], nparts=g["nparts"])
    argv = _subst(ARGV[case]["exec"], "t", "pairs", tmp=str(tmp_path))
    argv = [f"-e{g['L']}" if a.startswith("-e") else a for a in argv]         # (the golden tables are trimmed at their own L)
    r = subprocess.run([os.path.join(BIN, "hetmers"), *argv], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "pairs.smu").read_text() == g["smu"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["extract_default", "extract_verbose"])
def test_reference_cli_argv_replayed_on_the_drop_in_extract(case, tmp_path):
    from test_extract import load_extract, write_sma
    g, labels, lines, rows = load_extract("k31_i1")
    ktab.write_ktab(str(tmp_path / "t"), g["k"], g["packed"], g["counts"], ibyte=g["ibyte"], nparts=g["nparts"])
    write_sma(tmp_path / "s.sma", rows)
    argv = _subst(ARGV[case]["exec"], "t", "x", sma="s")
    r = subprocess.run([os.path.join(BIN, "extract_kmer_pairs"), *argv], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for lab, want in lines.items():
        got = sorted(open(tmp_path / f"x.{lab}.txt").read().splitlines(keepends=True))
        assert got == sorted(want), lab
