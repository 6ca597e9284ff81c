"""`python -m smudgeplot_amd hetmers|extract ...` -- host-side mirror of the reference's CLI for the two tasks that
exec the hot-path binaries (reference: src/smudgeplot/cli.py: Parser.hetmers 140-174, Parser.extract 210-232,
main 348-382, get_binary_path/run_binary 18-72).

Same task names, same arguments with the same meaning and defaults, same argv handed to the executables, same
"Calling: ..." / "Task: ..." / "Done!" lines, same CalledProcessError on a non-zero exit; the executables are the
MI355X drop-ins in smudgeplot_amd/bin (built by `make -C smudgeplot_amd/csrc`).  Every other smudgeplot task
(cutoff, peak_aggregation, plot, all) is untouched Python downstream of the `.smu` file: use the reference for
those.

One addition: `--gpus N` (exported as SMUDGEPLOT_GPUS) shards the table over N GPUs of the node.
"""

from __future__ import annotations

import argparse
import json
import os
import shlex
import shutil
import subprocess
import sys
from pathlib import Path

from . import __version__

TASKS = ("hetmers", "extract")


def get_binary_path(name: str) -> str:
    """bundled binary in the package first, then $PATH (reference: cli.py:18-56)"""
    bundled = Path(__file__).parent / "bin" / name
    if bundled.exists() and os.access(bundled, os.X_OK):
        return str(bundled)
    system_binary = shutil.which(name)
    if system_binary:
        return system_binary
    raise FileNotFoundError(
        f"Binary '{name}' not found. Build it with `make -C smudgeplot_amd/csrc`.\n"
        f"Checked locations:\n  - Package: {bundled.parent}\n  - System PATH: {os.get_exec_path()}\n")


def run_binary(name: str, args: list, env=None) -> None:
    """reference: cli.py:57-72 -- raises subprocess.CalledProcessError on a non-zero exit"""
    cmd_line = [get_binary_path(name)] + [str(x) for x in args]
    sys.stderr.write(f"Calling: {shlex.join(cmd_line)}\n")
    subprocess.run(cmd_line, check=True, env=env)


def hetmers_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="smudgeplot hetmers",
                                description="Calculate unique kmer pairs from FastK k-mer database.")
    p.add_argument("infile", help="Input FastK database (.ktab) file.")
    p.add_argument("-L", help="Count threshold below which k-mers are considered erroneous.", type=int, required=True)
    p.add_argument("-t", help="Number of threads (default 4).", type=int, default=4)
    p.add_argument("-o", help="The pattern used to name the output (kmerpairs).", default="kmerpairs")
    p.add_argument("-tmp", help="Directory where all temporary files will be stored (default /tmp).", default=".")
    p.add_argument("--verbose", action="store_true", default=False, help="Verbose mode.")
    p.add_argument("--json_report", action="store_true", default=False,
                   help="Write a JSON format report recording the selected parameters (default False)")
    p.add_argument("--gpus", type=int, default=0, help="[smudgeplot_amd] number of GPUs of the node to use (default 1).")
    return p


def extract_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="smudgeplot extract",
                                description="Extract kmer pair sequences from a FastK k-mer database.")
    p.add_argument("infile", help="Input FastK database (.ktab) file.")
    p.add_argument("sma", help="Input annotated k-mer pair file (.sma).")
    p.add_argument("-t", help="Number of threads (default 4)", type=int, default=4)
    p.add_argument("-o", help="The pattern used to name the output (kmerpairs).", default="kmerpairs")
    p.add_argument("-tmp", help="Directory where all temporary files will be stored (default /tmp).", default=".")
    p.add_argument("--verbose", action="store_true", default=False, help="verbose mode")
    return p


def hetmers_argv(args) -> list:
    """reference: cli.py:350-359"""
    out = [f"-o{args.o}", f"-e{args.L}", f"-T{args.t}"]
    if args.verbose:
        out.append("-v")
    if args.tmp != ".":
        out.append(f"-P{args.tmp}")
    out.append(args.infile)
    return out


def extract_argv(args) -> list:
    """reference: cli.py:369-378 (no -e: the executable's default threshold applies, as in the reference)"""
    out = [f"-o{args.o}", f"-T{args.t}"]
    if args.verbose:
        out.append("-v")
    if args.tmp != ".":
        out.append(f"-P{args.tmp}")
    out.append(args.infile)
    out.append(args.sma.removesuffix(".sma"))
    return out


def save_hetmers_json_report(outfile: str, input_params: dict) -> None:
    """reference: smudgeplot.py:411-421"""
    report = {"version": __version__, "commandline_arguments": shlex.join(sys.argv[1:]),
              "input_parameters": input_params}
    Path(f"{outfile}_report.json").write_text(json.dumps(report, indent=2) + "\n")


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    top = argparse.ArgumentParser(prog="smudgeplot", usage="smudgeplot <task> [options]\n\ntasks: hetmers  extract\n")
    top.add_argument("task", nargs="?", default="")
    if argv and argv[0] in ("-v", "--version"):
        sys.stderr.write(f"Running smudgeplot_amd v{__version__}\n")
        return 0
    task = argv[0] if argv else ""
    if task not in TASKS:
        top.print_usage(sys.stderr)
        sys.stderr.write("No task provided\n" if task == "" else f'"{task}" is not a valid task name\n')
        return 1
    sys.stderr.write(f"Running smudgeplot_amd v{__version__}\n")
    sys.stderr.write("Task: " + task + "\n")
    if task == "hetmers":
        args = hetmers_parser().parse_args(argv[1:])
        env = dict(os.environ, SMUDGEPLOT_GPUS=str(args.gpus)) if args.gpus > 1 else None
        run_binary("hetmers", hetmers_argv(args), env=env)
        if args.json_report:
            params = {k: v for k, v in vars(args).items() if k != "gpus" or v}
            save_hetmers_json_report(args.o, params)
    else:
        args = extract_parser().parse_args(argv[1:])
        run_binary("extract_kmer_pairs", extract_argv(args))
    sys.stderr.write("\nDone!\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
