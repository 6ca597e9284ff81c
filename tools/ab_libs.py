#!/usr/bin/env python3
"""Whole runs of several builds / settings of the engine on ONE device-resident bench table (A/B on one box, one table).

usage: ab_libs.py <workload>[@genome] spec [spec ...]       spec = lib.so[:ENV=val[:ENV=val ...]]   ("-" = the shipped library)
Every spec gets a fresh engine (own dlopen of the library: own statics, shared HIP runtime), the table bound with its prefix
index as bench.py does, 1 + 3 runs (smg_engine_run, hash proof); printed: mean times of the last three, the plot's weight and
whether the plot equals the first spec's, cell for cell.  Tuning infrastructure (tools/build_variants.sh makes the builds)."""
import hashlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smudgeplot_amd import engine
import bench
wl = sys.argv[1]
G = None
if "@" in wl:
    wl, g = wl.split("@"); G = int(float(g))
k = bench.default_k(wl)
G = G or bench.default_genome(wl)
dev = torch.device("cuda:0")
keys, cnt, L, desc = bench.make_table(wl, G, k, dev)
kw0 = keys if keys.dim() == 1 else keys[:, 0]
index = torch.cumsum(torch.bincount((kw0 >> 40) & 0xFFFFFF, minlength=1 << 24), 0)
keys = keys.reshape(-1)
torch.cuda.synchronize(); torch.cuda.empty_cache()
n = cnt.numel()
plot = torch.zeros(engine.PLOT_CELLS, dtype=torch.int64, device=dev)
print(f"# {wl} G={G} k={k} n={n}", flush=True)
first = None
golden = None
try:
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_tables.json"))).get(wl)
    if g and g["genome"] == G and g["k"] == k:
        golden = open(os.path.join(ROOT, "tests", "golden", g["smu_file"])).read()
except Exception:
    pass
for spec in sys.argv[2:]:
    parts = spec.split(":")
    path, envs = parts[0], dict(p.split("=", 1) for p in parts[1:])
    for a, b in envs.items(): os.environ[a] = b
    engine._lib = None
    engine.LIB_PATH = os.path.join(ROOT, "smudgeplot_amd", "libsmg_hetmers.so") if path == "-" else os.path.join(ROOT, path)
    try:
        e = engine.Engine(0)
        e.bind(k, n, keys.data_ptr(), cnt.data_ptr())
        if "NOINDEX" not in envs: e.set_prefix_index(index.data_ptr(), 3, 0)
        res = []
        for it in range(int(os.environ.get("AB_RUNS", "4"))):
            st = e.run(plot.data_ptr(), "hash")
            res.append(st)
        if os.environ.get("AB_PER_RUN"):          # every run on its own line (is a time bimodal from run to run?)
            for it, x in enumerate(res):
                print(f"    run {it}: total {x['ms_total']:7.3f} p1 {x['ms_pass1']:6.3f} lookup {x['ms_rclookup']:6.3f} part {x['ms_filter']:5.3f} p2 {x['ms_pass2']:5.3f}", flush=True)
        torch.cuda.synchronize()
        r = res[1:]
        m = lambda key: sum(x[key] for x in r) / len(r)
        h = hashlib.sha256(plot.cpu().numpy().tobytes()).hexdigest()[:12]
        if first is None: first = h
        gold = "" if golden is None else (" golden-ok" if engine.smu_text(plot.cpu().numpy().reshape(1001, 501)) == golden else " GOLDEN-MISMATCH")
        print(f"{spec:58s} total {m('ms_total'):7.3f} p1 {m('ms_pass1'):6.3f} (bigfix {m('ms_bigfix'):5.3f}) lookup {m('ms_rclookup'):6.3f} (part {m('ms_filter'):5.3f}) "
              f"p2 {m('ms_pass2'):5.3f}  kept {int(m('nrequests'))} of {int(m('nemitted'))} path {r[-1]['path']} plot {h} {'same' if h == first else 'DIFFERENT'}{gold}", flush=True)
        e.close()
    except Exception as ex:
        print(f"{spec:58s} FAILED {ex}", flush=True)
    for a in envs: os.environ.pop(a, None)
