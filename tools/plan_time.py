#!/usr/bin/env python3
"""Dev: time nrx_sparse_plan alone on the C4 tower's lookups (history [B, 50] half padding + item id over 200 k rows, user id over 10 M rows).
usage: plan_time.py [B=65536] [steps=100]     (under rocprofv3 --kernel-trace --stats: the planner's kernels)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda:0")
L = 50
gen = torch.Generator(device=dev).manual_seed(5)
lens = torch.randint(0, L + 1, (B,), device=dev, generator=gen)
mask = torch.arange(L, device=dev)[None] < lens[:, None]
hist = (torch.randint(1, 200_000, (B, L), device=dev, generator=gen) * mask).int().reshape(-1)
iid = torch.randint(1, 200_000, (B,), device=dev, generator=gen).int()
uid = torch.randint(1, 10_000_000, (B,), device=dev, generator=gen).int()
ids, tabs, rows = [hist, iid, uid], [1, 1, 0], [200_000, 200_000, 10_000_000]
for _ in range(5):
    ops.sparse_plan(ids, tabs, rows, 2)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(steps):
    ops.sparse_plan(ids, tabs, rows, 2)
b.record(); torch.cuda.synchronize()
print(f"C4 tower plan, B = {B} ({hist.numel() + 2 * B} lookups, {int((hist == 0).sum())} of them padding): {a.elapsed_time(b) / steps * 1e3:.1f} us per plan")
