#!/usr/bin/env python3
"""Dev: where does the row-sparse backward spend its time?  rocprof-free: torch profiler table (GPU kernels)
plus cProfile (host) over a few C2 train steps with sparse_grad=True."""
import os, sys, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_SPARSE
ops.set_index_check("off")
dev = torch.device("cuda:0")
B, F, D, rows = 65536, 26, 16, 1_000_000
gen = torch.Generator(device=dev).manual_seed(1)
tables = [torch.randn(rows, D, device=dev).requires_grad_(True) for _ in range(F)]
plan = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=1) for i in range(F)], out_width=F * D, use_fm=True)
ids = [torch.randint(1, rows, (B,), device=dev, generator=gen) for _ in range(F)]
def step():
    out, _, fm = ops.embed_apply(plan, tables, ids, [None] * F, sparse_grad=True)
    (out.sum() * 1e-6 + fm.sum()).backward()
    for t in tables: t.grad = None
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as p:
    for _ in range(5): step()
    torch.cuda.synchronize()
print(p.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
