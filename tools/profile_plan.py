#!/usr/bin/env python3
"""Dev: the row-sparse backward's planning step alone (nrx_sparse_plan) at a bench workload's shape, for rocprofv3 --kernel-trace --stats.
usage: profile_plan.py [c2|c4|c5] [steps] [uniform|zipf]     (NRX_PLAN_SORT=rocprim for the library sort)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dist = sys.argv[3] if len(sys.argv) > 3 else "uniform"
B = 65536
shapes = {"c2": ([B] * 26, [1_000_000] * 26, list(range(26))),
          "c4": ([B, B * 50, B], [10_000_000, 200_000, 200_000], [0, 1, 1]),
          "c5": ([B] * 40, [int(x) for x in np.geomspace(1e3, 5e8, 40)], list(range(40)))}
lens, rows, tab = shapes[wl]
rng = np.random.default_rng(0)
ids = []
for n, r in zip(lens, rows):
    x = np.minimum(rng.zipf(1.05, n) - 1, r - 1) if dist == "zipf" else rng.integers(0, r, n)
    ids.append(torch.from_numpy(x.astype(np.int64)).cuda())
nt = max(tab) + 1
# the placement form (nrx_sparse_plan_place: dest / walk outputs), as the bench's backward calls it; PLACE=0: the plain plan
pm = None
if os.environ.get("PLACE", "1") != "0":
    pm = ops.place_mask([0 if n == B else 2 for n in lens], [1 if n == B else n // B for n in lens])
for _ in range(5): ops.sparse_plan(ids, tab, rows, nt, pm)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(steps): ops.sparse_plan(ids, tab, rows, nt, pm)
b.record(); torch.cuda.synchronize()
print(f"{wl} ({dist} ids) plan [{os.environ.get('NRX_PLAN_SORT', 'segmented')}]: {a.elapsed_time(b) / steps * 1e3:.1f} us per call (includes the wrapper's allocations)")
