#!/usr/bin/env python3
"""Dev: one feature per ring step against two (NRX_WIDE_PAIR / NRX_FWD_PAIR), alternated inside ONE process on the same tables: the C5 gather with
the Wide&Deep column routing (embed_fwd_ring_wide) and the plain C5 concat (embed_fwd_ring, D = 32).  usage: probe_pair.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
path = bench.SingleGpuPath("c5", dev, 1)
wide, plain = path.calls, path.plain_calls
def t(calls, n=200):
    for i in range(20): calls[i % len(calls)].run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): calls[i % len(calls)].run()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for r in range(reps):
    os.environ["NRX_WIDE_PAIR"] = "0"; w0 = t(wide)
    os.environ["NRX_WIDE_PAIR"] = "1"; w1 = t(wide)
    os.environ["NRX_FWD_PAIR"] = "0"; p0 = t(plain)
    os.environ["NRX_FWD_PAIR"] = "1"; p1 = t(plain)
    print(f"rep {r}: split one/two per step {w0:.1f} / {w1:.1f} us | plain concat one/two per step {p0:.1f} / {p1:.1f} us", flush=True)
