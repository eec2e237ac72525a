#!/usr/bin/env python3
"""Dev: run only the bound training pass (forward training form + row-sparse backward) of a bench workload, for
rocprofv3 --kernel-trace --stats.   usage: profile_fwd_bwd.py [c2|c4|c5] [steps] [uniform|zipf]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
# c5: all 40 tables (224 GB) when the device holds them, like bench.py (NRX_BENCH_C5_SMALL=1 keeps the 27 tables of <= 16M rows)
dev = torch.device("cuda:0")
dist = sys.argv[3] if len(sys.argv) > 3 else "uniform"
path = bench.SingleGpuPath(wl, dev, 1, id_dist=dist)
fwd, bwd = path.train_pass()
for i in range(steps):
    if not os.environ.get("NO_PLAN_AHEAD"): bwd[i % 2].plan_ahead()
    fwd[i % 2].run()
    bwd[i % 2].run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(steps):
    if not os.environ.get("NO_PLAN_AHEAD"): bwd[i % 2].plan_ahead()      # NO_PLAN_AHEAD=1: the planning runs inline, after the forward
    fwd[i % 2].run()
    bwd[i % 2].run()
b.record()
torch.cuda.synchronize()
print(f"{wl} ({dist} ids): fwd+bwd {a.elapsed_time(b) / steps * 1e3:.1f} us per step")
