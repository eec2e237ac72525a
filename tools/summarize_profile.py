#!/usr/bin/env python3
"""Condenses a tools/collect.sh directory into a text summary (committed under profiles/)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
print(f"# profile summary for {d}")
for f in glob.glob(f"{d}/stats/*/*_kernel_stats.csv"):
    print("\n## rocprofv3 --kernel-trace --stats (kernels >= 0.5 % of GPU time)")
    print("name | calls | avg_ns | min_ns | max_ns | pct")
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) >= 0.5:
            print(f'{r["Name"][:110]} | {r["Calls"]} | {float(r["AverageNs"]):.0f} | {r["MinNs"]} | {r["MaxNs"]} | {r["Percentage"]}')
print("\n## PMC passes (mean per dispatch over the timed launches, per kernel; nrx kernels only)")
for f in sorted(glob.glob(f"{d}/pmc*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "at::native" not in k and "rocprim" not in k and "Cijk" not in k:        # this library's kernels
            agg[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            v = v[len(v) // 5:]            # drop the warm-up fifth
            print(f"{k} | {c} | n={len(v)} | mean={sum(v) / len(v):.1f}")
print("""
## how to read (guides/MI355X_MICROARCH.md, HBM section)
DRAM bytes per launch = TCC_EA0_RDREQ_DRAM_32B_sum * 32 (reads) + TCC_EA0_WRREQ_WRITE_DRAM_32B_sum * 32 (writes).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies each 128-byte read request as 64 bytes
(so it reads ~half of the true fetched bytes when, as here, every request is a 128-byte one:
TCC_EA0_RDREQ_128B_sum == TCC_EA0_RDREQ_sum) -- use 2 x FETCH_SIZE, or the DRAM_32B counter, for bytes.""")
