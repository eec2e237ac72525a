#!/usr/bin/env python3
"""Dev: print the kernel_stats.csv of a rocprofv3 --kernel-trace --stats run (first csv found under the directory given)."""
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not fs:
    sys.exit("no kernel_stats.csv under " + sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in list(csv.DictReader(open(fs[0])))[:n]:
    print(f'{r["Name"][:110]:110s} calls {r["Calls"]:>6s}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  total% {r["Percentage"]}')
