#!/usr/bin/env python3
"""Regenerate tests/golden/extract_*.json: for some of the golden tables, a pixel annotation (.sma content)
and the lines the REFERENCE extract_kmer_pairs (oracle/_ref/extract_ref, compiled from
/root/reference/src/lib/PloidyList.c by oracle/Makefile) wrote for it, sorted per smudge file (the
reference's line order depends on its thread schedule).  Run in the build container:

    make -C oracle ref && python tests/golden/make_golden_extract.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from smudgeplot_amd import ktab  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "extract_ref")
OUT = os.path.dirname(os.path.abspath(__file__))
NAMES = ["k31_i1", "k32_i1_p2", "k21_i2_p2", "k51_i1_p3", "k65_i1", "k100_i1", "k100_wrap"]
SMUDGES = ["1A1B", "2A1B", "2A2B"]


def labels_from_smu(smu_text):
    """annotate two thirds of the non-empty pixels, round-robin over three smudge labels"""
    rows = [tuple(int(v) for v in line.split("\t")) for line in smu_text.splitlines()]
    labels = []
    for n, (covb, cova, freq) in enumerate(rows):
        if n % 3 == 2:
            continue
        labels.append((covb, cova, freq, SMUDGES[(covb + cova) % 3]))
    return labels


for name in NAMES:
    d = np.load(os.path.join(OUT, name + ".npz"))
    smu = open(os.path.join(OUT, name + ".smu")).read()
    k, ibyte, nparts, L = int(d["k"]), int(d["ibyte"]), int(d["nparts"]), int(d["L"])
    labels = labels_from_smu(smu)
    with tempfile.TemporaryDirectory() as tmp:
        ktab.write_ktab(os.path.join(tmp, "t"), k, d["packed"], d["counts"], ibyte=min(ibyte, 2), nparts=nparts)
        with open(os.path.join(tmp, "s.sma"), "w") as f:
            f.write("covB\tcovA\tfreq\tsmudge\n")
            for covb, cova, freq, lab in labels:
                f.write(f"{covb}\t{cova}\t{freq}\t{lab}\n")
        r = subprocess.run([REF, f"-e{L}", "-T3", "-oout", "t.ktab", "s.sma"], cwd=tmp, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        lines = {}
        for lab in sorted({l[3] for l in labels}):
            lines[lab] = sorted(open(os.path.join(tmp, f"out.{lab}.txt")).read().splitlines())
    with open(os.path.join(OUT, f"extract_{name}.json"), "w") as f:
        json.dump({"table": name, "labels": labels, "lines": lines}, f, separators=(",", ":"))
    print(name, {k_: len(v) for k_, v in lines.items()})
