#!/usr/bin/env python3
"""Randomised soak of the engine against the numpy oracle (test infrastructure, run on the GPU box):
random k (5..128), table size, block structure, symmetry-proof mode, virtual multi-GPU shards, out-of-core shards (round 6:
also for raw tables), on-device conditioning from a raw canonical table.  Prints one line per failure and a summary; exit code 1 on failure."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import brute
from conftest import make_table
from smudgeplot_amd import engine, ktab, synth

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
fails, t0 = 0, time.time()
tot_n = tot_pairs = nraw = nshard = nseq = 0
for case in range(ncases):
    rng = np.random.default_rng(seed0 * 100003 + case)
    k = int(rng.choice([5, 7, 11, 15, 16, 17, 21, 24, 27, 31, 31, 31, 32, 33, 39, 47, 51, 63, 64, 65, 77, 85, 86, 101, 128]))
    m = int(rng.choice([3, 40, 400, 1500, 4000, 4000, 20000, 60000]))
    if k <= 8:
        m = min(m, 4 ** k // 6)
    L = int(rng.integers(2, 9))
    lc = int(rng.choice([0, 0, 60, 300]))
    dense = int(rng.choice([0, 0, 1, 3]))
    packed, cnt = synth.adversarial_table(k, max(m, 1), L, int(rng.integers(1 << 30)), low_complexity=lc, dense=dense)
    mode = str(rng.choice(["hash", "exact", "none"]))
    ibyte = int(rng.choice([1, 1, 2])) if (k + 3) // 4 > 2 else 1
    nparts = int(rng.integers(1, 4))
    shards = int(rng.choice([0, 0, 0, 2, 3, 7])) if (k <= 85 and mode != "none") else 0
    raw = bool(rng.random() < 0.25) and shards == 0
    seq = int(rng.choice([0, 0, 2, 3, 5])) if (mode != "none" and shards == 0) else 0       # out of core (round 6: any k, raw tables too)
    try:
        if seq:
            os.environ["SMG_SEQUENTIAL_SHARDS"] = str(seq)
        if raw:
            rc = ktab.revcomp_packed(packed, k)
            canon = np.array([bytes(a) <= bytes(b) for a, b in zip(packed, rc)])
            rp, rcnt = packed[canon], cnt[canon].copy()
            low = rng.random(len(rcnt)) < 0.15
            rcnt[low] = rng.integers(1, L, size=int(low.sum())) if L > 1 else rcnt[low]
            keep = rcnt >= L
            cp, cc = ktab.symmetrize(rp[keep], rcnt[keep], k)
            want = brute.hetmers_plot(cp, cc, k)
            tab = make_table(dict(packed=rp, counts=rcnt, k=k, ibyte=ibyte, nparts=nparts))
            plot, st = engine.hetmers_run(tab, symcheck="hash" if mode == "none" else mode,
                                          condition=engine.COND_TRIM | engine.COND_SYMM, ethresh=L)
        else:
            want = brute.hetmers_plot(packed, cnt, k)
            tab = make_table(dict(packed=packed, counts=cnt, k=k, ibyte=ibyte, nparts=nparts))
            if shards:
                os.environ["SMG_VIRTUAL_SHARDS"] = str(shards)
            else:
                os.environ.pop("SMG_VIRTUAL_SHARDS", None)
            plot, st = engine.hetmers_run(tab, symcheck=mode)
        ok = np.array_equal(plot, want)
        tot_n += len(cnt); tot_pairs += int(want.sum()); nraw += raw; nshard += shards > 0; nseq += seq > 0
    except Exception as ex:                                   # noqa: BLE001
        ok, st = False, {"error": str(ex)}
    finally:
        os.environ.pop("SMG_VIRTUAL_SHARDS", None)
        os.environ.pop("SMG_SEQUENTIAL_SHARDS", None)
    if not ok:
        fails += 1
        print(f"FAIL case={case} seed0={seed0} k={k} m={m} L={L} lc={lc} dense={dense} mode={mode} ibyte={ibyte} "
              f"nparts={nparts} shards={shards} seq={seq} raw={raw} n={len(cnt)} info={st if 'error' in st else st.get('path')}")
print(f"soak: {ncases - fails}/{ncases} cases agree with the oracle ({time.time() - t0:.0f} s; {tot_n} entries, "
      f"{tot_pairs} plot weight, {nraw} conditioned from raw tables, {nshard} as virtual multi-GPU shards, {nseq} out of core)")
sys.exit(1 if fails else 0)
