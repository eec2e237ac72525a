#!/usr/bin/env python3
"""Dev: C5 table set with the Wide&Deep split (10 smallest tables wide): forward + row-sparse backward timing."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from news_recsys_amd import ops
os.environ.setdefault("NRX_BENCH_C5_SMALL", "1")
dev = torch.device("cuda:0")
path = bench.SingleGpuPath("c5", dev, 1)
calls = path.wide_split_calls()[:2]
gen = torch.Generator(device=dev).manual_seed(3)
g_out = torch.randn(calls[0].out.shape, device=dev, generator=gen)
g_wide = torch.randn(calls[0].wide.shape, device=dev, generator=gen)
bwd = [ops.PreparedSparseBackward(f, g_out, None, g_wide) for f in calls]
def step(i):
    calls[i % 2].run(); bwd[i % 2].run()
for i in range(10): step(i)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(30): step(i)
b.record(); torch.cuda.synchronize()
print(f"c5 (27 tables, wide split on the 10 smallest): fwd+bwd {a.elapsed_time(b) / 30 * 1e3:.1f} us per step")
