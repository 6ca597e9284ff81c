#!/usr/bin/env python3
"""BASELINE configs[0] + configs[1] end to end, on the GPU box:
   a yeast-scale stand-in table (synthetic tetraploid, k=31, L=12; the real SRR3265401 data and FastK
   are not available offline) is written to disk in FastK format, then
     configs[0]: the REFERENCE hetmers (oracle/_ref/hetmers_ref) at -T1 and -T<cores>   (CPU)
     configs[1]: the drop-in smudgeplot_amd/bin/hetmers on the SAME files                (1x MI355X)
   and the two .smu files are compared byte for byte.  Prints one JSON object.
   Test / measurement infrastructure: the product path is only the `hetmers` binary."""
import json, os, subprocess, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smudgeplot_amd import ktab, synth, synth_device

G = int(float(sys.argv[1])) if len(sys.argv) > 1 else 12_000_000
KIND = sys.argv[2] if len(sys.argv) > 2 else "tetraploid"      # or "diploid" (BASELINE configs[2] generator)
SKIP_T1 = len(sys.argv) > 3 and sys.argv[3] in ("skipT1", "oursOnly")
OURS_ONLY = len(sys.argv) > 3 and sys.argv[3] == "oursOnly"
k, L = 31, (12 if KIND == "tetraploid" else 10)
dev = torch.device("cuda:0")
if KIND == "tetraploid":
    tk, tc = synth_device.polyploid_table(G, ploidy=4, div=0.01, cov_hap=25.0, k=k, L=L, seed=3265401, device=dev)
else:
    tk, tc = synth_device.diploid_table(G, k=k, het=0.01, cov=50.0, L=L, seed=1, device=dev)
keys = tk.cpu().numpy().view(np.uint64); cnt = tc.cpu().numpy().view(np.uint16)
ref = os.path.join(ROOT, "oracle", "_ref", "hetmers_ref")
ours = os.path.join(ROOT, "smudgeplot_amd", "bin", "hetmers")
out = {"workload": (f"synthetic tetraploid {G} bp, 1% divergence, 25x per haplotype, k={k}, L={L}" if KIND == "tetraploid"
                    else f"synthetic diploid {G} bp, 50x, 1% het, k={k}, L={L}"), "entries": int(len(cnt)),
       "host_cores": os.cpu_count()}
with tempfile.TemporaryDirectory(prefix="smg_e2e") as d:
    t0 = time.time()
    synth.write_u64_table(os.path.join(d, "t"), keys, cnt, k, ibyte=3, nparts=4)
    out["table_bytes"] = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
    out["write_s"] = round(time.time() - t0, 2)

    def run(cmd, name, env=None):
        p = os.path.join(d, name + ".smu")
        if os.path.exists(p):
            os.remove(p)
        t = time.time()
        r = subprocess.run(cmd, cwd=d, capture_output=True, text=True, env=env)
        t1 = time.time()
        dt = t1 - t
        assert r.returncode == 0, r.stderr
        err = r.stderr
        import re
        m = re.search(r"main\(\) entered at ([0-9.]+), left at ([0-9.]+)", err)
        if m:       # what lies outside main(): exec + dynamic loader in front, exit() (HIP runtime shutdown) behind
            err += f"  [smg] outside main(): {1e3 * (float(m.group(1)) - t):.1f} ms before (exec, loader), {1e3 * (t1 - float(m.group(2))):.1f} ms after (exit)\n"
        return dt, err

    cores = min(64, os.cpu_count() or 1)
    if not OURS_ONLY:
        run([ref, f"-e{L}", f"-T{cores}", "-owarm", "t.ktab"], "warm")          # page cache
    if not SKIP_T1:
        dt, _ = run([ref, f"-e{L}", "-T1", "-oref1", "t.ktab"], "ref1")
        out["reference_T1"] = {"wall_s": round(dt, 3), "kmers_per_s": len(cnt) / dt}
    if not OURS_ONLY:
        dt, _ = run([ref, f"-e{L}", f"-T{cores}", "-orefN", "t.ktab"], "refN")
        out[f"reference_T{cores}"] = {"wall_s": round(dt, 3), "kmers_per_s": len(cnt) / dt}
    # -T = host threads that read the part files (CLI default: 4).  The executable works in two processes (a worker that
    # starts the HIP runtime while the starter opens and probes the table, and whose release of the device nobody waits
    # for: hetmers_main.c); SMUDGEPLOT_ONE_PROCESS=1 is the round-4 form.  Best of three; a pause between two runs lets the
    # worker of the run before finish handing its device context back.
    for T, one in ((4, False), (32, False), (4, True)):
        best, walls = None, []
        for _ in range(3):
            time.sleep(0.4)
            dt, err = run([ours, f"-e{L}", f"-T{T}", "-v", "-ogpu", "t.ktab"], "gpu",
                          env=dict(os.environ, SMUDGEPLOT_ONE_PROCESS="1") if one else None)
            walls.append(round(dt, 3))
            if best is None or dt < best[0]:
                best = (dt, err)
        out[f"mi355x_hetmers_end_to_end_T{T}" + ("_one_process" if one else "")] = {
            "wall_s": round(best[0], 3), "all_runs_s": walls, "kmers_per_s": len(cnt) / best[0],
            "engine_line": [l.strip() for l in best[1].splitlines() if "[smg]" in l]}
    time.sleep(0.4)
    a = open(os.path.join(d, "gpu.smu"), "rb").read()
    out["smu_bytes"] = len(a)
    if OURS_ONLY:
        print(json.dumps(out, indent=1)); sys.exit(0)
    if not SKIP_T1:
        out["byte_identical_vs_reference_T1"] = a == open(os.path.join(d, "ref1.smu"), "rb").read()
    out["byte_identical_vs_reference_TN"] = a == open(os.path.join(d, "refN.smu"), "rb").read()
print(json.dumps(out, indent=1))
