#!/usr/bin/env python3
"""Dev: GPU-kernel and host breakdown of the fused C2 training step (fwd + sorted bwd + nrx_sparse_adam_step)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_SPARSE
from news_recsys_amd.model.model_utils.optim import FusedSparseAdam
ops.set_index_check("off")
dev = torch.device("cuda:0")
B, F, D, rows = 65536, 26, 16, 1_000_000
gen = torch.Generator(device=dev).manual_seed(1)
tables = [torch.randn(rows, D, device=dev).requires_grad_(True) for _ in range(F)]
plan = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=1) for i in range(F)], out_width=F * D, use_fm=True)
ids = [torch.randint(1, rows, (B,), device=dev, generator=gen) for _ in range(F)]
up = torch.randn(B, F * D, device=dev, generator=gen) * 1e-3
sink = ops.SparseGradSink(); opt = FusedSparseAdam(sink, lr=1e-3)
def step():
    out, _, fm = ops.embed_apply(plan, tables, ids, [None] * F, sparse_grad=sink)
    ((out * up).sum() + fm.sum()).backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as p:
    for _ in range(5): step()
    torch.cuda.synchronize()
print(p.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=70))
