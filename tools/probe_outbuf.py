#!/usr/bin/env python3
"""Dev probe: does a re-used output buffer (what torch's caching allocator gives a training loop) change
the C2 kernel time vs distinct output buffers per step?  (256 MiB Infinity Cache vs 109 MB output.)"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_SPARSE
dev = torch.device("cuda:0")
B, F, D, rows = 65536, 26, 16, 1_000_000
gen = torch.Generator(device=dev).manual_seed(1)
tables = [torch.randn(rows, D, device=dev) for _ in range(F)]
plan = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=1) for i in range(F)], out_width=F * D, use_fm=True)
pool = [[torch.randint(1, rows, (B,), device=dev, generator=gen) for _ in range(F)] for _ in range(8)]

def timeit(calls, steps=200, spacer=None):
    for c in calls: c.run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(steps):
        calls[i % len(calls)].run()
        if spacer is not None: spacer()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / steps * 1e3

sep = [ops.PreparedEmbed(plan, tables, ins, [None] * F) for ins in pool]
shared_out = torch.empty(B, F * D, device=dev); shared_fm = torch.empty(B, device=dev)
shr = [ops.PreparedEmbed(plan, tables, ins, [None] * F, out=shared_out, fm=shared_fm) for ins in pool]
two = [torch.empty(B, F * D, device=dev) for _ in range(2)]
alt = [ops.PreparedEmbed(plan, tables, ins, [None] * F, out=two[i % 2], fm=shared_fm) for i, ins in enumerate(pool)]
print(f"separate out buffers (8 x 109 MB): {timeit(sep):7.1f} us")
print(f"shared out buffer   (1 x 109 MB): {timeit(shr):7.1f} us")
print(f"two out buffers     (2 x 109 MB): {timeit(alt):7.1f} us")
# a consumer that reads the output (like the MLP head would) between steps
w = torch.randn(F * D, 128, device=dev)
def consume(): torch.mm(shared_out, w)
t_mm = timeit([type("X", (), {"run": staticmethod(consume)})()])
print(f"consumer GEMM [B,416]x[416,128] alone: {t_mm:7.1f} us")
print(f"shared out + consumer GEMM each step : {timeit(shr, spacer=consume):7.1f} us (sum if serial: kernel + {t_mm:.1f})")
