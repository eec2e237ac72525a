#!/usr/bin/env python3
"""Dev: the C5 gather with the Wide&Deep column routing, aligned-chunk stores (NRX_WIDE_ALIGNED=1) against the dword-aligned store form (=0) and
the plain concat, alternated inside ONE process on the same tables (two processes place 224 GB of tables differently: +-5 %).
usage: probe_wide.py [reps]      (NRX_BENCH_C5_SMALL=1: the 27-table set)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
path = bench.SingleGpuPath("c5", dev, 1)
wide, plain = path.calls, path.plain_calls
unpadded = None
if os.environ.get("UNPADDED", "1") == "1":       # the round-3 form: deep row stride 1270 (rows only 8-byte aligned)
    from news_recsys_amd import ops
    p0 = wide[0].plan
    out = torch.empty((bench.BATCH, p0.out_width), dtype=torch.float32, device=dev)
    unpadded = [ops.PreparedEmbed(p0, path.tables, ins, ws, out=out) for ins, ws in path.pool]

from news_recsys_amd import ops as _ops
p0 = wide[0].plan
out128 = torch.empty((bench.BATCH, 1280), dtype=torch.float32, device=dev)
ld1280 = [_ops.PreparedEmbed(p0, path.tables, ins, ws, out_ld=1280, out=out128) for ins, ws in path.pool]

def t(calls, n=200):
    for i in range(20): calls[i % len(calls)].run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): calls[i % len(calls)].run()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for r in range(reps):
    os.environ["NRX_WIDE_ALIGNED"] = "1"; a1 = t(wide)
    os.environ["NRX_WIDE_ALIGNED"] = "0"; a0 = t(wide)
    u = t(unpadded) if unpadded else float("nan")
    pl = t(plain)
    os.environ["NRX_WIDE_ALIGNED"] = "1"; b1 = t(ld1280)
    os.environ["NRX_WIDE_ALIGNED"] = "0"; b0 = t(ld1280)
    print(f"rep {r}: split, stride 1272, aligned chunks {a1:.1f} us | split, stride 1272, dword-aligned stores {a0:.1f} us | split, stride 1270 (round 3) {u:.1f} us | plain concat {pl:.1f} us | split, stride 1280: aligned {b1:.1f} / dword {b0:.1f} us")
