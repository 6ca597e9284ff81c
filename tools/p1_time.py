"""pass-1 timing experiments: SMG_DBG_SKIP bit mask (1 scan, 2 directory, 4 emit+fp, 8 code store)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smudgeplot_amd import engine, synth_device
G = float(sys.argv[1]) if len(sys.argv) > 1 else 2e8
dev = torch.device("cuda:0")
tk, tc = synth_device.diploid_table(int(G), k=31, seed=1, device=dev)
torch.cuda.synchronize(); torch.cuda.empty_cache()
e = engine.Engine(0)
e.bind(31, tc.numel(), tk.data_ptr(), tc.data_ptr())
for mode in ("hash", "exact"):
    for skip in (0, 1, 2, 4, 8, 15):
        os.environ["SMG_DBG_SKIP"] = str(skip)
        t = []
        for _ in range(3):
            try:
                e.pass1(mode)
            except Exception as ex:
                pass
            t.append(e.stats()["ms_pass1"])
        print(f"n={tc.numel()} mode={mode} skip={skip:2d} pass1 ms = {min(t):.3f}")
