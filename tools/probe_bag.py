#!/usr/bin/env python3
"""Dev probe: the LDS-staged bag (history) path of the generic kernel vs bag length / table size / fill."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_SPARSE
dev = torch.device("cuda:0"); B = 65536
gen = torch.Generator(device=dev).manual_seed(3)

def run(L, rows, D=16, fill=1.0, extra_sparse=0, steps=50):
    table = torch.randn(rows, D, device=dev)
    slots = [ops.Slot("h", NRX_BAG_MASKED_MEAN, 0, D, L, 0)]
    ids = torch.randint(1, rows, (B, L), device=dev, generator=gen)
    lens = (torch.rand(B, device=dev, generator=gen) * L * fill * 2).clamp(max=L).long() if fill < 1 else torch.full((B,), L, device=dev)
    mask = (torch.arange(L, device=dev)[None] < lens[:, None]).float()
    ids = ids * mask.long()
    inputs, weights, tables = [ids], [mask], [table]
    for i in range(extra_sparse):
        t = torch.randn(1_000_000, D, device=dev); tables.append(t)
        slots.append(ops.Slot(f"s{i}", NRX_SPARSE, i + 1, D, 0, (i + 1) * D))
        inputs.append(torch.randint(1, 1_000_000, (B,), device=dev, generator=gen)); weights.append(None)
    plan = ops.EmbedPlan(slots, out_width=(1 + extra_sparse) * D)
    call = ops.PreparedEmbed(plan, tables, inputs, weights)
    for _ in range(3): call.run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(steps): call.run()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / steps * 1e3
    n_valid = mask.sum().item()
    alg = B * L * 12 + n_valid * 4 * D + B * 4 * D + extra_sparse * B * (8 + 8 * D)
    print(f"L={L:4d} rows={rows:>9d} D={D} fill={fill:.2f} sparse={extra_sparse}: {us:8.1f} us  {alg / us / 1e3:7.1f} GB/s alg  {n_valid / us:8.1f} M rows/s", flush=True)

run(50, 200_000)
run(50, 200_000, fill=0.5)
run(50, 10_000_000)
run(50, 2_000)
run(10, 200_000)
run(200, 200_000)
run(50, 200_000, D=64)
run(50, 200_000, extra_sparse=2)
