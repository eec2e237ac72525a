#!/usr/bin/env python3
"""Dev: nrx_route_ids_dedup and nrx_unique_inverse under the two sorts (NRX_PLAN_SORT unset = the planner's tile kernels, =rocprim = the library
sort), alternated in one process; C2-shaped exchange (26 x 65536 ids over 26 tables x 1M rows, world 8) and a 1.7 M-element unique."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
B, F, rows, world = 65536, 26, 1_000_000, 8
rng = np.random.default_rng(0)
ids = [torch.from_numpy(rng.integers(0, rows, B)).cuda() for _ in range(F)]
zipf = [torch.from_numpy(np.minimum(rng.zipf(1.05, B) - 1, rows - 1).astype(np.int64)).cuda() for _ in range(F)]
flat = torch.cat(ids)
lrows = [(rows + world - 1) // world + 1] * F
cap = int(B * F / world * 1.05) + 256

def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for rep in range(2):
    for sort in ("segmented", "rocprim"):
        if sort == "rocprim": os.environ["NRX_PLAN_SORT"] = "rocprim"
        else: os.environ.pop("NRX_PLAN_SORT", None)
        d_u = t(lambda: ops.route_ids_dedup(ids, list(range(F)), lrows, world, cap))
        d_z = t(lambda: ops.route_ids_dedup(zipf, list(range(F)), lrows, world, cap))
        u = t(lambda: ops.unique_inverse(flat))
        print(f"rep {rep} {sort:9s}: route_ids_dedup C2 uniform {d_u:.1f} us, Zipf {d_z:.1f} us | unique_inverse of {flat.numel()} int64 {u:.1f} us  (incl. the wrappers' allocations)")
