"""The drop-in `hetmers` is ONE process as far as its caller can tell -- like the reference (PloidyPlot.c:1232-1630: one
process, no signal handlers; the Python CLI waits for it with subprocess.run(check=True), cli.py:57-72) -- although by
default it forks a GPU worker (hetmers_main.c).  Both process modes give the same bytes; a killed `hetmers` leaves no process
behind and no .smu written later; a worker that dies costs the caller exit status 1 and a message, not a half-written file.
"""
import os
import signal
import stat
import subprocess
import sys
import time

import numpy as np
import psutil
import pytest

from conftest import HETMERS_BIN, REF_BIN, ROOT, load_golden
from smudgeplot_amd import engine, ktab, synth

MODES = [pytest.param({}, id="two-processes"), pytest.param({"SMUDGEPLOT_ONE_PROCESS": "1"}, id="one-process")]


def _env(extra):
    return dict(os.environ, **extra)


def _descendants_gone(pids, within=1.0):
    t0 = time.time()
    while time.time() - t0 < within:
        if not any(psutil.pid_exists(p) and psutil.Process(p).status() != psutil.STATUS_ZOMBIE for p in pids):
            return True
        time.sleep(0.02)
    return False


# ---- no GPU: both modes fail the same way, and nothing stays behind -----------------------------------------------------

@pytest.mark.parametrize("mode", MODES)
def test_without_a_device_both_modes_exit_1_with_the_message_and_leave_nothing(mode, tmp_path):
    if engine.device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu tests")
    packed, cnt = synth.adversarial_table(31, 200, 6, seed=5)
    ktab.write_ktab(str(tmp_path / "t"), 31, packed, cnt, ibyte=1)
    p = subprocess.Popen([HETMERS_BIN, "-e6", "-v", "-oout", "t"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, env=_env(mode))
    kids = [c.pid for c in psutil.Process(p.pid).children(recursive=True)]
    out, err = p.communicate(timeout=60)
    assert p.returncode == 1
    assert err.startswith("\n  The input table is trimmed and symmetric\n")
    assert err.endswith("hetmers: no HIP device available (this engine has no CPU fallback)\n")
    assert not (tmp_path / "out.smu").exists()
    assert _descendants_gone(kids)


def test_a_starter_that_cannot_open_the_table_takes_its_worker_along(tmp_path):
    """the starter exits 1 with the reference's message (PloidyPlot.c:1351-1354); the worker, which waits for the table, must go too"""
    p = subprocess.Popen([HETMERS_BIN, "-e4", "-oout", "missing_table"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True)
    kids = [c.pid for c in psutil.Process(p.pid).children(recursive=True)]
    out, err = p.communicate(timeout=60)
    assert p.returncode == 1 and err == "hetmers: Cannot open k-mer table missing_table\n"
    assert _descendants_gone(kids)


# ---- with a GPU --------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["k31_i1", "k31_i3_p4", "k51_i1_p3", "k100_wrap"])
def test_golden_tables_through_the_executable_in_both_process_modes(name, mode, tmp_path):
    g = load_golden(name)
    ktab.write_ktab(str(tmp_path / "t"), g["k"], g["packed"], g["counts"], ibyte=g["ibyte"], nparts=g["nparts"])
    r = subprocess.run([HETMERS_BIN, "-okmerpairs", f"-e{g['L']}", "-T4", "-v", "t.ktab"], cwd=tmp_path, capture_output=True, text=True,
                       env=_env(mode))
    assert r.returncode == 0, r.stderr
    assert "  The input table is trimmed and symmetric\n" in r.stderr and "  Count complete, outputting table\n" in r.stderr
    assert (tmp_path / "kmerpairs.smu").read_text() == g["smu"]


def _big_enough_table(tmp_path, m=400000):
    """a table whose run takes the worker a few hundred milliseconds at least (runtime start-up included)"""
    packed, cnt = synth.adversarial_table(31, m, 5, seed=11)
    ktab.write_ktab(str(tmp_path / "t"), 31, packed, cnt, ibyte=3, nparts=2)


def _start(tmp_path, extra=None):
    p = subprocess.Popen([HETMERS_BIN, "-e5", "-T4", "-oout", "t"], cwd=tmp_path, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                         text=True, env=_env(extra or {}))
    worker = None
    t0 = time.time()
    while worker is None and time.time() - t0 < 2.0 and p.poll() is None:      # the fork is the first thing the starter does
        kids = psutil.Process(p.pid).children()
        worker = kids[0].pid if kids else None
    assert worker is not None, "no worker process seen (two-process mode is the default)"
    return p, worker


@pytest.mark.gpu
@pytest.mark.parametrize("sig", [signal.SIGTERM, signal.SIGINT, signal.SIGKILL])
def test_killing_hetmers_mid_run_leaves_no_worker_and_no_smu(sig, tmp_path):
    """what a caller's timeout (SIGTERM / SIGKILL) or a Ctrl-C handed on by a non-tty parent (SIGINT) does: the reference dies
    and nothing writes its output later.  SIGTERM / SIGINT are handed on by the starter; SIGKILL cannot be caught -- the
    kernel tells the worker (PR_SET_PDEATHSIG)."""
    _big_enough_table(tmp_path)
    p, worker = _start(tmp_path)
    time.sleep(0.03)                                           # (the worker is starting the HIP runtime: 80 ms at the very least)
    p.send_signal(sig)
    p.wait(timeout=30)
    assert p.returncode == -sig                                 # dies OF the signal, like a process without handlers
    assert _descendants_gone([worker], within=1.0), "the GPU worker outlived the process the caller started"
    time.sleep(1.0)
    assert not (tmp_path / "out.smu").exists(), "a .smu appeared after hetmers had been killed"


@pytest.mark.gpu
def test_a_worker_that_dies_costs_exit_status_1_and_a_message(tmp_path):
    _big_enough_table(tmp_path)
    p, worker = _start(tmp_path)
    time.sleep(0.03)
    os.kill(worker, signal.SIGKILL)
    _, err = p.communicate(timeout=60)
    assert p.returncode == 1
    assert err.endswith("hetmers: the GPU worker process ended unexpectedly (signal 9)\n")
    assert not (tmp_path / "out.smu").exists()


@pytest.mark.gpu
def test_a_finished_run_leaves_no_process_behind_for_long(tmp_path):
    """the starter returns as soon as the .smu is closed; the worker only hands the device back after that (tens of ms)"""
    _big_enough_table(tmp_path, m=50000)
    p, worker = _start(tmp_path)
    _, err = p.communicate(timeout=120)
    assert p.returncode == 0, err
    assert (tmp_path / "out.smu").stat().st_size > 0
    assert _descendants_gone([worker], within=5.0)


FAKE_TOOL = r'''#!%(py)s
"""stand-in for FastK's %(name)s (not installed here; thegenemyers/FASTK): enough of it for hetmers' shell-outs"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from smudgeplot_amd import ktab
args = [a for a in sys.argv[1:] if not a.startswith("-")]
name = %(name)r
if name == "Fastrm":
    for a in args:
        ktab.remove_ktab(a)
elif name == "Symmex":
    src, dst = args
    t = ktab.read_ktab(src)
    p, c = ktab.symmetrize(t.packed, t.counts, t.k)
    ktab.write_ktab(dst, t.k, p, c, ibyte=1, nparts=1)
elif name == "Logex":
    expr, src = args                                   # '.trim=A[6-]'
    dst, rest = expr.split("=")
    e = int(rest[rest.index("[") + 1: rest.index("-")])
    t = ktab.read_ktab(src)
    keep = t.counts >= e
    ktab.write_ktab(dst, t.k, t.packed[keep], t.counts[keep], ibyte=1, nparts=1)
'''


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_the_reference_s_shell_outs_in_both_process_modes(mode, tmp_path):
    """SMUDGEPLOT_USE_FASTK_TOOLS=1 on a raw table: Logex, Symmex, Fastrm are called as the reference calls them
    (PloidyPlot.c:1381-1414, 1584-1592) -- stand-ins on PATH here -- and the temporary table's name, which the starter hands to
    the worker through the shared mapping, is released by the worker without touching the mapping (round 5 free()d it there)."""
    from test_gpu_parity import _raw_table
    k, L = 31, 6
    (rp, rcnt), (cp, cc) = _raw_table(k, 71, L)
    ktab.write_ktab(str(tmp_path / "raw"), k, rp, rcnt, ibyte=1, nparts=2)
    ktab.write_ktab(str(tmp_path / "cond"), k, cp, cc, ibyte=1, nparts=2)
    bindir = tmp_path / "fastk_bin"
    bindir.mkdir()
    for name in ("Logex", "Symmex", "Fastrm"):
        f = bindir / name
        f.write_text(FAKE_TOOL % {"py": sys.executable, "root": ROOT, "name": name})
        f.chmod(f.stat().st_mode | stat.S_IXUSR)
    env = _env(dict(mode, SMUDGEPLOT_USE_FASTK_TOOLS="1", PATH=str(bindir) + os.pathsep + os.environ["PATH"]))
    r = subprocess.run([HETMERS_BIN, f"-e{L}", "-T4", "-v", "-ogpu", "raw"], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "  Trimming k-mers in table with count < 6\n" in r.stderr and "  Making trimmed table symmetric\n" in r.stderr
    assert not (tmp_path / ".trim.ktab").exists() and not (tmp_path / ".symx.ktab").exists()      # Fastrm ran (twice)
    q = subprocess.run([REF_BIN, f"-e{L}", "-T4", "-oref", "cond"], cwd=tmp_path, capture_output=True, text=True)
    assert q.returncode == 0, q.stderr
    assert (tmp_path / "gpu.smu").read_text() == (tmp_path / "ref.smu").read_text() != ""
