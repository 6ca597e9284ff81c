"""Timeline of the last bench step from a rocprofv3 --kernel-trace CSV: every launch with its start (relative to the
step's first kernel), duration and the idle gap in front of it.  Usage: step_timeline.py <dir> <first-kernel-substring>"""
import csv
import glob
import sys

d, first = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if first in r[2]]
if len(starts) < 2:
    sys.exit("no two steps found")
a, b = starts[-2], starts[-1]
# a step begins a few launches (memsets are not kernels; fills are) before its pass-1 kernel: take launch to launch
t0 = rows[a][0]
busy = 0
prev_end = None
print("# one step: from the start of %s to the start of the next one: %.3f ms" % (first, (rows[b][0] - t0) / 1e6))
for s, e, n in rows[a:b]:
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    busy += e - s
    print("%9.1f us  +%8.1f us  gap %7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n[:110]))
    prev_end = max(e, prev_end or e)
print("# kernels busy %.3f ms of %.3f ms" % (busy / 1e6, (rows[b][0] - t0) / 1e6))
