#!/usr/bin/env python3
"""Dev probe: sweep the fused gather over (F, D, rows, id dtype, outputs) and print us / GB/s.
Used to find what bounds the kernel (row size vs fetch granularity, store cost, id width)."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_SPARSE
ops.set_index_check("off")
dev = torch.device("cuda:0")
B = 65536

def run(F, D, rows, idt=torch.int64, fm=False, need_out=True, steps=50, pool=4):
    gen = torch.Generator(device=dev).manual_seed(1)
    tables = [torch.randn(rows, D, device=dev) for _ in range(F)]
    plan = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=int(fm)) for i in range(F)], out_width=F * D, use_fm=fm)
    ids = [[torch.randint(1, rows, (B,), device=dev, generator=gen).to(idt) for _ in range(F)] for _ in range(pool)]
    with torch.no_grad():
        for i in range(5):
            ops.embed_apply(plan, tables, ids[i % pool], [None] * F, need_out=need_out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(steps):
            ops.embed_apply(plan, tables, ids[i % pool], [None] * F, need_out=need_out)
        e.record()
        torch.cuda.synchronize()
    us = s.elapsed_time(e) / steps * 1e3
    rd = B * F * D * 4
    wr = B * F * D * 4 if need_out else 0
    ix = B * F * ids[0][0].element_size()
    print(f"F={F:3d} D={D:4d} rows={rows:>9d} id={str(idt)[-5:]} fm={int(fm)} out={int(need_out)}  {us:8.1f} us   "
          f"alg {(rd + wr + ix) / us / 1e3:7.1f} GB/s   rows/s {B * F / us:8.1f} M/s  row-read {rd / us / 1e3:7.1f} GB/s", flush=True)
    del tables, ids

if __name__ == "__main__":
    run(26, 16, 1_000_000)
    run(26, 16, 1_000_000, fm=True)
    run(26, 16, 1_000_000, fm=True, need_out=False)      # gather only (4 B out per sample)
    run(26, 16, 1_000_000, idt=torch.int32)
    run(26, 32, 1_000_000)                                # 128 B rows: same row count, 2x bytes
    run(26, 64, 1_000_000)
    run(26, 16, 100_000)                                  # 6.4 MB tables: L2/MALL resident
    run(26, 16, 10_000)
    run(13, 16, 1_000_000)
    run(52, 16, 1_000_000)
    run(5, 64, 10_000_000)
    run(40, 32, 1_000_000)
    run(8, 128, 1_000_000)
    run(4, 256, 1_000_000)
