#!/usr/bin/env python3
"""profiles/hbm_traffic.json entry from two PMC passes (tools/pmc_summary.py outputs: one with FETCH_SIZE, one with WRITE_SIZE).

usage: make_traffic.py <key> <entries> <pmc_fetch.txt> <pmc_write.txt> <label> [json]
HBM bytes of a kernel = (2 x FETCH_SIZE + WRITE_SIZE) KiB: the counters are in KiB and FETCH_SIZE reports one half of a
coalesced read on gfx950 (MI355X_MICROARCH.md, section HBM; profiles/r01_calibration_fetch_size.txt).  The entry carries
the hash of the engine's gfx950 code object the counters were read on: bench.py gives no traffic figure for another one."""
import collections, hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
key, n, fc, fw, label = sys.argv[1], int(float(sys.argv[2])), sys.argv[3], sys.argv[4], sys.argv[5]
out = sys.argv[6] if len(sys.argv) > 6 else os.path.join(ROOT, "profiles", "hbm_traffic.json")


def read(path, counter):
    res, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
        else:
            m = re.match(r"\s+(\S+)\s+mean=\s*([0-9.]+)", line)
            if m and m.group(1) == counter and cur:
                res[cur] = float(m.group(2))
    return res


f, w = read(fc, "FETCH_SIZE"), read(fw, "WRITE_SIZE")
kern = {}
for k in sorted(set(f) | set(w)):
    kern[k] = round((2 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024 / n, 3)
p1 = sum(v for k, v in kern.items() if k.startswith(("kf_pass1", "kf_collect", "kf_bigfix")))
p2 = sum(v for k, v in kern.items() if k.startswith("kf_pass2"))
sys.path.insert(0, ROOT)
from smudgeplot_amd import codeobj
h = codeobj.code_object_hash()
doc = json.load(open(out)) if os.path.exists(out) else {}
doc[key] = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, {label}: (2 x FETCH_SIZE + WRITE_SIZE) KiB / {n} entries "
                      "(the counters are in KiB and FETCH_SIZE reports 1/2 on gfx950: profiles/r01_calibration_fetch_size.txt); "
                      "ms_pass1 = kf_pass1* + kf_collect + kf_bigfix (the launches inside the pass-1 event bracket); ms_pass2 = kf_pass2 + kf_pass2_far",
            "code_object_sha256_16": h,
            "bytes_per_entry": {"ms_pass1": round(p1, 3), "ms_pass2": round(p2, 3)},
            "kernels": {k: v for k, v in kern.items() if v >= 0.001}}
json.dump(doc, open(out, "w"), indent=1)
print(key, "pass1 %.3f pass2 %.3f B/entry" % (p1, p2), "lib", h)
