#!/usr/bin/env python3
"""Static instruction mix per source phase of a kernel.

Lines `//@mark NAME` in smudgeplot_amd/csrc/*.hpp are turned into `asm volatile("; ##NAME")` in a scratch copy,
the library is compiled to gfx950 assembly, and the instructions between markers are counted
(VALU / SALU / LDS / VMEM).  Usage: tools/isa_phase_count.py <mangled-kernel-prefix>
e.g. tools/isa_phase_count.py _Z10kf_pass1_rILi1ELb1EE
"""
import collections, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "smudgeplot_amd", "csrc")
tmp = tempfile.mkdtemp(prefix="isa")
for f in os.listdir(src):
    text = open(os.path.join(src, f)).read()
    text = re.sub(r'//@mark (\w+)', r'asm volatile("; ##\1");', text)
    open(os.path.join(tmp, f), "w").write(text)
out = os.path.join(tmp, "k.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                "-S", "--cuda-device-only", "-o", out, os.path.join(tmp, "smg_hetmers.hip")], check=True,
               stderr=subprocess.DEVNULL)
s = open(out).read()
m = re.search(r'\n' + re.escape(sys.argv[1]) + r'[^\n]*:\s*;[^\n]*\n', s)
body = s[m.end():]
body = body[:body.index('.Lfunc_end')]
# a marker that occurs several times (a template instantiated twice inside one kernel: interior and edge tiles) is
# numbered per occurrence: P3_RC, P3_RC#2, ...
cur, cnt, seen = 'PRE', collections.OrderedDict(), collections.Counter()
for l in body.split('\n'):
    l = l.strip()
    if l.startswith('; ##'):
        name = l[4:]
        seen[name] += 1
        cur = name if seen[name] == 1 else f"{name}#{seen[name]}"
        continue
    if not l or l.startswith((';', '.')) or l.endswith(':'):
        continue
    op = l.split()[0]
    k = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else \
        'vmem' if op.startswith(('global_', 'flat_', 'scratch_', 'buffer_')) else 'other'
    cnt.setdefault(cur, collections.Counter())[k] += 1
for k, v in cnt.items():
    print(f"{k:12s}", dict(v))
for key in ("NumVgprs", "ScratchSize", "Occupancy"):
    mm = re.search(r'; ' + key + r': (\d+)', s[m.end():])
    print(key, mm.group(1) if mm else "?")
shutil.rmtree(tmp)
