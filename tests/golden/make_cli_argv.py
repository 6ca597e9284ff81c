#!/usr/bin/env python3
"""Record what the REFERENCE command line (src/smudgeplot/cli.py:57-72 run_binary, 348-382) passes to its two
executables.  Run in the build container (where /root/reference exists); the GPU box replays the recorded argument
vectors against the drop-in executables (tests/test_gpu_parity.py::test_reference_cli_argv_*), so the boundary is
pinned by the reference's own code without the reference having to travel.

usage: python tests/golden/make_cli_argv.py   ->  tests/golden/cli_argv.json
"""
import importlib.metadata
import json
import os
import subprocess
import sys

REF = "/root/reference/src"
sys.path.insert(0, REF)
importlib.metadata.version = lambda name: "0.0.0-test"          # (the package is not installed here)

import smudgeplot.cli as cli                                      # noqa: E402

cli.version = lambda name: "0.0.0-test"
calls = []


def fake_run(cmd, check=True, **kw):
    calls.append(list(cmd))
    return subprocess.CompletedProcess(cmd, 0)


cli.subprocess.run = fake_run
cli.get_binary_path = lambda name: name                          # resolution is tested separately (PATH)

CASES = {
    "hetmers_default": ["smudgeplot", "hetmers", "-L", "5", "-o", "OUT", "TABLE.ktab"],
    "hetmers_threads_verbose": ["smudgeplot", "hetmers", "-L", "12", "-t", "16", "-o", "OUT", "--verbose", "TABLE.ktab"],
    "hetmers_tmp": ["smudgeplot", "hetmers", "-L", "4", "-o", "OUT", "-tmp", "TMPDIR", "TABLE"],
    "extract_default": ["smudgeplot", "extract", "-o", "OUT", "TABLE.ktab", "SMA.sma"],
    "extract_verbose": ["smudgeplot", "extract", "-o", "OUT", "-t", "8", "--verbose", "TABLE.ktab", "SMA.sma"],
}
out = {}
for name, argv in CASES.items():
    calls.clear()
    sys.argv = argv
    try:
        cli.main()
    except SystemExit:
        pass
    assert len(calls) == 1, (name, calls)
    out[name] = {"cli": argv[1:], "exec": calls[0]}
    print(name, calls[0])
here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "cli_argv.json"), "w") as f:
    json.dump(out, f, indent=1)
