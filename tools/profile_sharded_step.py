#!/usr/bin/env python3
"""Dev: run only the sharded engine's bound training step (forward + backward, row layout, world 1) of a bench workload, for
rocprofv3 --kernel-trace --stats.   usage: profile_sharded_step.py [c2|c3|c4|c5] [steps] [fwd|step]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from news_recsys_amd import ops
from news_recsys_amd.sharding import ShardedBenchPath
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
what = sys.argv[3] if len(sys.argv) > 3 else "step"
ops.set_index_check("off")
dev = torch.device("cuda:0")
path = ShardedBenchPath(wl, dev, 20260116, 0, 1, bench.BATCH, "row")
if what == "step":
    assert path.train_setup()
    fn = path.train_step
else:
    fn = path.step
for i in range(6):
    fn(i)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import time
a.record()
t0 = time.perf_counter()
for i in range(steps):
    fn(i)
host = (time.perf_counter() - t0) / steps * 1e6
b.record()
torch.cuda.synchronize()
print(f"{wl} sharded world 1 ({path.engine} engine), {what}: {a.elapsed_time(b) / steps * 1e3:.1f} us per step   (host: {host:.1f} us of enqueueing per step)")
