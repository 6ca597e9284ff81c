#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz + *.smu.

Each fixture is a small conditioned (trimmed + symmetric) table stored compactly (packed k-mers
+ counts; the on-disk .ktab with its 2^(8*ibyte) index is rebuilt by the tests) together with
the `.smu` bytes that the REFERENCE hetmers binary (oracle/_ref/hetmers_ref, compiled from
/root/reference/src/lib by oracle/Makefile) produced for it.  Run from the repo root, in the
build container (the reference sources do not exist on the GPU box):

    make -C oracle ref && python tests/golden/make_golden.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from smudgeplot_amd import ktab, synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "hetmers_ref")
OUT = os.path.dirname(os.path.abspath(__file__))

# name, k, ibyte, nparts, L, m, seed, low_complexity, dense, threads
CASES = [
    ("k31_i1", 31, 1, 1, 5, 1500, 11, 150, 2, 1),
    ("k31_i3_p4", 31, 3, 4, 5, 1500, 12, 100, 2, 4),
    ("k32_i1_p2", 32, 1, 2, 4, 1500, 13, 100, 2, 4),
    ("k21_i2_p2", 21, 2, 2, 4, 1500, 14, 100, 2, 32),
    ("k17_i1", 17, 1, 1, 4, 1500, 15, 100, 2, 64),
    ("k33_i1", 33, 1, 1, 4, 1200, 16, 100, 2, 3),
    ("k40_i2_p5", 40, 2, 5, 6, 1200, 17, 100, 2, 3),
    ("k51_i1_p3", 51, 1, 3, 4, 1200, 18, 100, 2, 8),
    ("k64_i1", 64, 1, 1, 4, 1000, 19, 80, 1, 2),
    ("k65_i1", 65, 1, 1, 4, 1000, 20, 80, 1, 2),
    ("k100_i1", 100, 1, 1, 4, 800, 21, 60, 1, 2),
]


def wrap_case(k=100, seed=5):
    """A k-mer with exactly 256 one-away partners: its uint8 degree wraps to 0
    (PloidyPlot.c:163,535), so the reference DOES count its pair with a partner of degree 1."""
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 4, size=k, dtype=np.uint8)
    rows = [x]
    for p in range(85):                       # 85 positions x 3 variants = 255 partners
        for d in (1, 2, 3):
            y = x.copy(); y[p] = (y[p] + d) & 3; rows.append(y)
    y = x.copy(); y[90] = (y[90] + 1) & 3; rows.append(y)      # 256th partner, degree 1
    z = rng.integers(0, 4, size=(50, k), dtype=np.uint8)       # bystanders
    bases = np.concatenate([np.array(rows), z])
    packed = ktab.pack_bases(bases)
    cnt = rng.integers(5, 60, size=len(packed)).astype(np.uint16)
    packed, cnt = ktab.sort_unique_packed(packed, cnt)
    return ktab.symmetrize(packed, cnt, k)


def run_ref(packed, cnt, k, ibyte, nparts, L, threads):
    with tempfile.TemporaryDirectory(prefix="smg_gold") as d:
        ktab.write_ktab(os.path.join(d, "t"), k, packed, cnt, ibyte=ibyte, nparts=nparts)
        r = subprocess.run([REF, f"-e{L}", f"-T{threads}", "-v", "-oout", "t.ktab"], cwd=d,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "trimmed and symmetric" in r.stderr, r.stderr
        return open(os.path.join(d, "out.smu")).read()


def main():
    for name, k, ib, parts, L, m, seed, lc, dn, T in CASES:
        packed, cnt = synth.adversarial_table(k, m, L, seed, low_complexity=lc, dense=dn)
        smu = run_ref(packed, cnt, k, ib, parts, L, T)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), packed=packed, counts=cnt,
                            k=k, ibyte=ib, nparts=parts, L=L)
        open(os.path.join(OUT, name + ".smu"), "w").write(smu)
        print(f"{name}: n={len(cnt)} lines={smu.count(chr(10))}")
    packed, cnt = wrap_case()
    smu = run_ref(packed, cnt, 100, 1, 1, 4, 2)
    np.savez_compressed(os.path.join(OUT, "k100_wrap.npz"), packed=packed, counts=cnt,
                        k=100, ibyte=1, nparts=1, L=4)
    open(os.path.join(OUT, "k100_wrap.smu"), "w").write(smu)
    print(f"k100_wrap: n={len(cnt)} lines={smu.count(chr(10))}")


if __name__ == "__main__":
    main()
