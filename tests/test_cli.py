"""smudgeplot_amd/cli.py mirrors the reference's `hetmers` / `extract` tasks (src/smudgeplot/cli.py:140-174,
210-232, 348-382): same options, same argv for the executables, same report file."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT, load_golden
from smudgeplot_amd import cli, ktab


def test_hetmers_argv_matches_the_reference_cli():
    a = cli.hetmers_parser().parse_args(["-L", "12", "-t", "8", "-o", "out", "tab.ktab"])
    assert cli.hetmers_argv(a) == ["-oout", "-e12", "-T8", "tab.ktab"]
    a = cli.hetmers_parser().parse_args(["tab", "-L", "5", "--verbose", "-tmp", "/scratch"])
    assert cli.hetmers_argv(a) == ["-okmerpairs", "-e5", "-T4", "-v", "-P/scratch", "tab"]
    with pytest.raises(SystemExit) as e:                      # -L is required, like in the reference
        cli.hetmers_parser().parse_args(["tab"])
    assert e.value.code == 2


def test_extract_argv_matches_the_reference_cli():
    a = cli.extract_parser().parse_args(["tab.ktab", "smudges.sma", "-o", "x", "-t", "2", "--verbose"])
    assert cli.extract_argv(a) == ["-ox", "-T2", "-v", "tab.ktab", "smudges"]
    a = cli.extract_parser().parse_args(["tab", "s", "-tmp", "/t"])
    assert cli.extract_argv(a) == ["-okmerpairs", "-T4", "-P/t", "tab", "s"]


def test_task_dispatch_errors(capsys):
    assert cli.main([]) == 1
    assert "No task provided" in capsys.readouterr().err
    assert cli.main(["plot"]) == 1
    assert '"plot" is not a valid task name' in capsys.readouterr().err
    assert cli.main(["--version"]) == 0
    assert cli.get_binary_path("hetmers").endswith("smudgeplot_amd/bin/hetmers")
    with pytest.raises(FileNotFoundError):
        cli.get_binary_path("no_such_binary_xyz")


def test_failing_binary_raises_like_the_reference(tmp_path):
    r = subprocess.run([sys.executable, "-m", "smudgeplot_amd", "hetmers", "-L", "4", "missing_table"], cwd=tmp_path,
                       capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode != 0
    assert "Task: hetmers" in r.stderr and "Calling: " in r.stderr
    assert "hetmers: Cannot open k-mer table missing_table" in r.stderr
    assert "CalledProcessError" in r.stderr


@pytest.mark.gpu
def test_cli_hetmers_end_to_end_with_report(tmp_path):
    g = load_golden("k31_i1")
    ktab.write_ktab(str(tmp_path / "t"), g["k"], g["packed"], g["counts"], ibyte=1, nparts=2)
    r = subprocess.run([sys.executable, "-m", "smudgeplot_amd", "hetmers", "-L", str(g["L"]), "-o", "pairs", "--json_report",
                        "t.ktab"], cwd=tmp_path, capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stderr
    assert r.stderr.rstrip().endswith("Done!")
    assert (tmp_path / "pairs.smu").read_text() == g["smu"]
    rep = json.loads((tmp_path / "pairs_report.json").read_text())
    assert rep["input_parameters"]["L"] == g["L"] and rep["input_parameters"]["o"] == "pairs"
    assert "hetmers" in rep["commandline_arguments"]
