#!/usr/bin/env python3
"""Dev: from a rocprofv3 --kernel-trace csv, the mean duration of the k-th launch of a kernel within a step (a kernel launched m times per
step: m positions).  usage: ktrace_by_position.py <dir> <kernel substring> <launches per step>"""
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not fs:
    sys.exit("no kernel_trace.csv under " + sys.argv[1])
sub, m = sys.argv[2], int(sys.argv[3])
rows = [r for r in csv.DictReader(open(fs[0])) if sub in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 5 // m * m:]
for k in range(m):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[k::m]]
    print(f"{sub} launch {k} of {m}: mean {sum(d) / len(d) / 1e3:.1f} us  min {min(d) / 1e3:.1f}  max {max(d) / 1e3:.1f}  (n={len(d)})")
