#!/usr/bin/env python3
"""Time pass 1 (and optionally a whole run) of several builds of the engine on ONE device-resident table.

usage: p1_ablate.py <genome> <k> lib1.so[:env=val,...] lib2.so ...      (":full" after a lib = whole runs instead)
Each library is dlopen'ed separately (own statics, shared HIP runtime)."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smudgeplot_amd import engine, synth_device
G = int(float(sys.argv[1])); k = int(sys.argv[2])
dev = torch.device("cuda:0")
if k <= 31:
    tk, tc = synth_device.diploid_table(G, k=k, het=0.01, cov=50.0, L=10, seed=1, device=dev)
else:
    tk, tc = synth_device.diploid_table_wide(G, k=k, het=0.01, cov=50.0, L=10, seed=1, device=dev)
    tk = tk.reshape(-1)
torch.cuda.synchronize(); torch.cuda.empty_cache()
n = tc.numel()
plot = torch.zeros(engine.PLOT_CELLS, dtype=torch.int64, device=dev)
print(f"# n={n} k={k}", flush=True)
for spec in sys.argv[3:]:
    parts = spec.split(":")
    path, full, envs = parts[0], False, {}
    for p in parts[1:]:
        if p == "full": full = True
        elif "=" in p:
            a, b = p.split("=", 1); envs[a] = b
    for a, b in envs.items(): os.environ[a] = b
    engine._lib = None
    engine.LIB_PATH = os.path.join(ROOT, path)
    try:
        e = engine.Engine(0)
        e.bind(k, n, tk.data_ptr(), tc.data_ptr())
        res = []
        for it in range(4):
            if full:
                st = e.run(plot.data_ptr(), "hash")
                res.append((st["ms_pass1"], st["ms_rclookup"], st["ms_pass2"], st["ms_total"], st["nrequests"], st.get("ms_filter", 0),
                            st.get("ms_bigfix", 0), st.get("nbig", 0)))
            else:
                e.set_blockmap_bits(32)        # what a whole single-GPU run uses (the phase API defaults to the 30-bit exchange map)
                e.pass1("hash"); st = e.stats(); res.append((st["ms_pass1"], st.get("ms_bigfix", 0), st.get("nbig", 0)))
        torch.cuda.synchronize()
        r = res[1:]
        mean = [sum(x[i] for x in r) / len(r) for i in range(len(r[0]))]
        if full:
            print(f"{spec:60s} pass1 {mean[0]:7.3f} (bigfix {mean[6]:5.3f}, {int(mean[7])} entries) lookup {mean[1]:7.3f} (filter {mean[5]:6.3f}) pass2 {mean[2]:6.3f} total {mean[3]:7.3f} ms  kept {int(mean[4])} pairs {int(plot.sum().item())}", flush=True)
        else:
            print(f"{spec:60s} pass1 {mean[0]:7.3f} ms (bigfix {mean[1]:5.3f}, {int(mean[2])} entries)", flush=True)
        e.close()
    except Exception as ex:
        print(f"{spec:60s} FAILED {ex}", flush=True)
    for a in envs: os.environ.pop(a, None)
