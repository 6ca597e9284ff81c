#!/usr/bin/env python3
"""Golden .smu files of the bench tables, made by the REFERENCE binary (oracle/_ref/hetmers_ref = PloidyPlot.c compiled from
the reference's own sources) on the GPU box:

   tools/make_bench_goldens.py [outdir] [workload[@genome][:k] ...]        default: gpurun_out/goldens, all four bench workloads
                                                                            (e.g. uniform:30, uniform@5e8:51 -- the k = 30 / k = 51 lines)

For every workload of bench.py at its default size: the bench's generator puts the table into HBM, its checksum is taken
(synth_device.table_hash), the engine runs on the resident table, the table is written as a FastK table from the device and
the reference runs on those files (-T64; the four reference runs overlap -- the box has 256 hardware threads -- so their
wall times are NOT the CPU baseline, profiles/r04_e2e_*_full.json has those).  Written: <outdir>/bench_<workload>.smu (the
reference's output, PloidyPlot.c:1603-1617) and <outdir>/bench_tables.json (entries, table checksum, sha256 of the .smu,
and whether the engine's text was identical).  Copy both into tests/golden/: bench.py and the -m gpu suite compare the
plot of a run with them.  Test infrastructure: the reference binary is the judge here, never part of the product path."""
import hashlib, json, os, shutil, subprocess, sys, tempfile, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smudgeplot_amd import engine, sharded, synth_device
import bench

outdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "goldens")
workloads = sys.argv[2:] or list(bench.WORKLOADS)
os.makedirs(outdir, exist_ok=True)
dev = torch.device("cuda:0")
ref = os.path.join(ROOT, "oracle", "_ref", "hetmers_ref")
cores = min(64, os.cpu_count() or 1)
tmp = os.environ.get("E2E_TMP") or tempfile.gettempdir()
jobs, meta = [], {}
for spec in workloads:
    wl, G, k = spec, None, None
    if ":" in wl: wl, k = wl.split(":"); k = int(k)
    if "@" in wl: wl, G = wl.split("@"); G = int(float(G))
    k = k or bench.default_k(wl)
    G = G or bench.default_genome(wl, k)
    name = wl
    wl_name = bench.golden_key(wl, G, k)
    t0 = time.time()
    keys, cnt, L, desc = bench.make_table(wl, G, k, dev)
    torch.cuda.synchronize()
    n = cnt.numel()
    hk, hc = synth_device.table_hash(keys, cnt)
    m = {"workload": wl, "key": wl_name, "genome": G, "k": k, "L": L, "entries": int(n), "description": desc,
         "table_hash": synth_device.table_hash_text(n, hk, hc), "generate_s": round(time.time() - t0, 1)}
    eng = sharded.TorchEngine(dev)
    plot, st = sharded.hetmers_sharded(k, keys.reshape(-1), cnt, symcheck="hash", eng=eng)
    torch.cuda.synchronize()
    m["engine_smu"] = engine.smu_text(plot.cpu().numpy().reshape(1001, 501))
    m["engine_path"] = int(st.get("path", 0))
    del eng, plot
    d = tempfile.mkdtemp(prefix="smg_gold_" + wl_name, dir=tmp)
    t0 = time.time()
    m["table_bytes"] = synth_device.write_table_from_device(os.path.join(d, "t"), keys, cnt, k, nparts=4)
    m["write_s"] = round(time.time() - t0, 1)
    del keys, cnt
    torch.cuda.empty_cache()
    p = subprocess.Popen([ref, f"-e{L}", f"-T{cores}", "-oref", "t.ktab"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    jobs.append((wl_name, d, p, time.time()))
    meta[wl_name] = m
    print(f"[goldens] {wl_name}: {n} entries, table {m['table_hash']}, reference started", file=sys.stderr, flush=True)

table = {}
for wl, d, p, t0 in jobs:
    out, err = p.communicate()
    m = meta[wl]
    if p.returncode != 0:
        m["error"] = err[-2000:]
        table[wl] = {kk: v for kk, v in m.items() if kk != "engine_smu"}
        continue
    m["reference_wall_s_overlapped"] = round(time.time() - t0, 1)
    smu = open(os.path.join(d, "ref.smu")).read()
    with open(os.path.join(outdir, f"bench_{wl}.smu"), "w") as f:
        f.write(smu)
    m["smu_file"] = f"bench_{wl}.smu"
    m["smu_bytes"] = len(smu)
    m["smu_sha256"] = hashlib.sha256(smu.encode()).hexdigest()
    m["engine_identical"] = m.pop("engine_smu") == smu
    m["made_by"] = f"oracle/_ref/hetmers_ref -e{m['L']} -T{cores} (the reference's PloidyPlot.c) on the table written by synth_device.write_table_from_device"
    table[wl] = m
    shutil.rmtree(d, ignore_errors=True)
    print(f"[goldens] {wl}: {m['smu_bytes']} bytes, engine identical: {m['engine_identical']}", file=sys.stderr, flush=True)
old = os.path.join(outdir, "bench_tables.json")
if os.path.exists(old):                      # (goldens made earlier stay: this call adds or replaces its own)
    table = dict(json.load(open(old)), **table)
with open(old, "w") as f:
    json.dump(table, f, indent=1)
print(json.dumps({w: {kk: v for kk, v in m.items() if kk in ("entries", "table_hash", "smu_sha256", "engine_identical", "error")} for w, m in table.items()}, indent=1))
