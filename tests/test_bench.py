"""bench.py under the launchers the driver uses (a 1-rank torch.distributed.run included), on small tables: the JSON
line carries the contract's fields, the roofline and the CPU baseline -- for two-word k-mers (BASELINE configs[4] is
k = 51) as well."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
NEED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline", "e2e", "parity"}


def _line(cmd, env=None):
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert NEED <= set(j), sorted(NEED - set(j))
    assert j["value"] > 0 and j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1
    assert "workload" in j["config"] and "model" not in j["config"]
    return j


def test_bench_default_launcher_small_table_with_cpu_baseline():
    j = _line([sys.executable, "bench.py", "--genome", "3e6", "--steps", "2", "--warmup", "1"])
    assert j["n_gpus"] == 1 and j["dtype"] == "u64" and j["steps"] == 2
    assert j["roofline"]["kernel"] == "kf_pass1_d<1, 1, true, true, 2>"           # the name rocprofv3 prints for the hot form
    cb = j["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["value"] > 0 and cb["cores"] >= 1 and "entry table" in cb["sample"]
    # the end-to-end block: both programs ran in this bench process on the same files, byte-identical .smu
    e = j["e2e"]
    assert e["smu_identical"] is True and e["entries"] > 0 and e["hetmers_wall_s"] > 0 and e["reference_wall_s"] > 0
    assert len(e["hetmers_runs_s"]) == 2 and e["cores"] == cb["cores"]
    assert abs(cb["value"] - e["entries"] / e["reference_wall_s"]) < 0.02 * cb["value"]      # one and the same reference run


def test_bench_under_torch_distributed_run_one_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    j = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--genome", "3e6", "--no-cpu"])
    assert j["n_gpus"] == 1 and j["cpu_baseline"] is None and j["e2e"] is None


def test_bench_k51_and_the_forced_exchange_protocol():
    j = _line([sys.executable, "bench.py", "--k", "51", "--genome", "2e6", "--steps", "2", "--warmup", "1"])
    assert j["dtype"] == "u64x2" and "kf_pass1_d<2" in j["roofline"]["kernel"]
    assert j["cpu_baseline"] is not None and j["cpu_baseline"]["value"] > 0            # the reference runs k = 51 too
    j2 = _line([sys.executable, "bench.py", "--genome", "3e6", "--steps", "2", "--warmup", "1", "--no-cpu"],
               env=dict(os.environ, SMG_FORCE_EXCHANGE="1"))
    j1 = _line([sys.executable, "bench.py", "--genome", "3e6", "--steps", "2", "--warmup", "1", "--no-cpu"])
    assert j1["pairs_in_plot"] == j2["pairs_in_plot"]


def test_bench_fails_loudly_without_the_reference_binary(tmp_path):
    """a fresh clone has no oracle/_ref (git-ignored): the line must not silently lose its baseline"""
    import bench
    ref = bench.REF_BIN
    hidden = ref + ".hidden_by_test"
    os.rename(ref, hidden)
    try:
        r = subprocess.run([sys.executable, "bench.py", "--genome", "3e6", "--steps", "1", "--warmup", "0"], cwd=ROOT,
                           capture_output=True, text=True, timeout=600)
    finally:
        os.rename(hidden, ref)
    assert r.returncode != 0 and "hetmers_ref is missing" in r.stderr
    # ... loudly, but without losing the measurement: the line of the timed region is printed, the error in place of the block
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["value"] > 0 and j["cpu_baseline"] is None and "hetmers_ref is missing" in j["e2e"]["error"]


def test_bench_repeats_workload_reaches_the_rare_paths():
    j = _line([sys.executable, "bench.py", "--workload", "repeats", "--genome", "3e6", "--steps", "2", "--warmup", "1", "--no-cpu"])
    assert "repeats" in j["config"]["workload"]
    assert j["roofline"]["deferred_entries"]["count"] > 0            # window blocks beyond +-30 entries: kf_bigfix ran
