#!/usr/bin/env python3
"""Time k_decode alone: format F records (ibyte = 3) of a synthetic diploid table, resident in HBM, through
smg_engine_decode (one launch over the whole shard; the executable decodes piece by piece behind the copies).
usage: decode_time.py <genome> [k]     -- prints ms per launch, entries/s and the fraction of the HBM roofline
(pbyte bytes read + 8 W + 2 bytes written per entry)."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smudgeplot_amd import engine, synth_device
G = int(float(sys.argv[1])); k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
dev = torch.device("cuda:0")
tk, tc = synth_device.diploid_table(G, k=k, het=0.01, cov=50.0, L=10, seed=1, device=dev)
n = tc.numel(); kb = (k + 3) // 4; hb = kb - 3
rec = torch.empty((n, hb + 2), dtype=torch.uint8, device=dev)
for j in range(hb):
    rec[:, j] = ((tk >> (8 * (7 - (3 + j)))) & 0xFF).to(torch.uint8)
c = tc.to(torch.int32) & 0xFFFF
rec[:, hb] = (c & 0xFF).to(torch.uint8); rec[:, hb + 1] = (c >> 8).to(torch.uint8)
index = torch.cumsum(torch.bincount((tk >> 40) & 0xFFFFFF, minlength=1 << 24), 0).to(torch.int64)
del c
e = engine.Engine(0)
ms = []
for _ in range(4):
    e.decode(k, 3, n, rec.data_ptr(), index.data_ptr())
    ms.append(e.stats()["ms_decode"])
torch.cuda.synchronize()
nn, pk, pc = e.table()
ok = nn == n
# spot check against the generator's table
import ctypes
back = torch.empty(n, dtype=torch.int64, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy(ctypes.c_void_p(back.data_ptr()), ctypes.c_void_p(pk), ctypes.c_size_t(8 * n), 3)
ok = ok and bool(torch.equal(back, tk))
best = min(ms[1:])
byts = n * (hb + 2 + 8 + 2)
print(f"k_decode k={k} n={n}: {best:.3f} ms per launch = {n / best / 1e6:.0f} G entries/s, {byts / best / 1e6:.0f} GB/s = "
      f"{byts / best / 1e6 / 8000:.3f} of the HBM roofline; equal to the source table: {ok}")
