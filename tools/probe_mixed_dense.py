#!/usr/bin/env python3
"""Dev: 24 features of widths 16 / 32 / 64 interleaved, B = 65536, with and without a dense value in the middle of the sorted order
(which misaligns every later feature's first column).  Prints us per launch and the fraction of the 8 TB/s roofline."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_DENSE, NRX_SPARSE
dev, B = "cuda:0", 65536
for dense_at in (None, 8):
    dims = [16, 32, 64] * 8
    slots, tables, ins, col, byts = [], [], [], 0, 0
    for i, d in enumerate(dims):
        if dense_at is not None and i == dense_at:
            slots.append(ops.Slot("dense", NRX_DENSE, -1, 1, 0, col)); ins.append(torch.rand(B, device=dev)); col += 1; byts += 8
        tables.append(torch.randn(1_000_000, d, device=dev))
        slots.append(ops.Slot(f"f{i:02d}", NRX_SPARSE, len(tables) - 1, d, 0, col)); col += d; byts += 8 + 8 * d
        ins.append(torch.randint(1, 1_000_000, (B,), device=dev))
    plan = ops.EmbedPlan(slots, out_width=col)
    call = ops.PreparedEmbed(plan, tables, ins, [None] * len(slots))
    for _ in range(30): call.run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100): call.run()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 10
    print(f"24 features (16/32/64), dense value at position {dense_at}: {us:.1f} us per call, {byts * B / us / 1e6 / 8000 * 1e3:.3f} of 8 TB/s ({byts} B/impression)")
