"""k > 32 timing: synthetic diploid table with two-word k-mers (default k=51), engine.run in hash mode."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smudgeplot_amd import engine, synth_device
G = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 51
dev = torch.device("cuda:0")
tk, tc = synth_device.diploid_table_wide(G, k=k, device=dev)
torch.cuda.synchronize(); torch.cuda.empty_cache()
n = tc.numel()
e = engine.Engine(0)
e.bind(k, n, tk.data_ptr(), tc.data_ptr())
plot = torch.zeros(engine.PLOT_CELLS, dtype=torch.int64, device=dev)
for mode in ("hash", "hash", "exact"):
    st = e.run(plot.data_ptr(), mode)
    torch.cuda.synchronize()
    print(f"k={k} n={n} mode={mode} path={st['path']} pass1={st['ms_pass1']:.2f} lookup={st['ms_rclookup']:.2f} "
          f"pass2={st['ms_pass2']:.2f} total={st['ms_total']:.2f} ms  => {n / st['ms_total'] / 1e6:.2f} G k-mers/s, "
          f"pairs={int(plot.sum().item())} nreq={st['nrequests']}")
