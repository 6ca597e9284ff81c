import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))      # checkers: tests may import them

GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_BIN = os.path.join(ROOT, "oracle", "_build", "hetmers_oracle")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "hetmers_ref")
HETMERS_BIN = os.path.join(ROOT, "smudgeplot_amd", "bin", "hetmers")
LIB = os.path.join(ROOT, "smudgeplot_amd", "libsmg_hetmers.so")
AGG_LIB = os.path.join(ROOT, "smudgeplot_amd", "libsmg_aggregate.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a HIP device: the GPU tests are skipped, not failed (the engine has no CPU
    fallback and says so; -m "not gpu" is the suite that is meant to run there)."""
    have = None
    for it in items:
        if "gpu" not in it.keywords:
            continue
        if have is None:
            try:
                from smudgeplot_amd import engine
                have = os.path.exists(LIB) and engine.device_count() > 0
            except Exception:
                have = False
        if not have:
            it.add_marker(pytest.mark.skip(reason="no HIP device (the engine has no CPU fallback)"))


def golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    smu = open(os.path.join(GOLDEN, name + ".smu")).read()
    return dict(packed=d["packed"], counts=d["counts"], k=int(d["k"]), ibyte=int(d["ibyte"]),
                nparts=int(d["nparts"]), L=int(d["L"]), smu=smu)


@pytest.fixture(scope="session", autouse=True)
def built():
    """Make sure the native pieces exist (no-op when they were built already)."""
    if not (os.path.exists(LIB) and os.path.exists(HETMERS_BIN) and os.path.exists(AGG_LIB)):
        subprocess.run(["make", "-C", os.path.join(ROOT, "smudgeplot_amd", "csrc"), "all"], check=True)
    if not os.path.exists(ORACLE_BIN):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)


def make_table(g):
    from smudgeplot_amd import ktab
    pre = np.zeros(len(g["counts"]), dtype=np.int64)
    for j in range(g["ibyte"]):
        pre = (pre << 8) | g["packed"][:, j].astype(np.int64)
    index = np.cumsum(np.bincount(pre, minlength=1 << (8 * g["ibyte"]))).astype(np.int64)
    n = len(g["counts"])
    # split into nparts on prefix boundaries, like write_ktab
    cuts = [0]
    for p in range(1, g["nparts"]):
        b = int(np.searchsorted(index, (n * p) // g["nparts"], side="left"))
        cuts.append(max(int(index[min(b, len(index) - 1)]), cuts[-1]))
    cuts.append(n)
    return ktab.KTable(g["k"], g["ibyte"], g["nparts"], 1, g["packed"], g["counts"], index,
                       np.diff(np.array(cuts, dtype=np.int64)))
