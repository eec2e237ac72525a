#!/usr/bin/env python3
"""Dev: features per ring step (NRX_WIDE_STEP = 1, 2, 4, 8: separate instantiations) of the C5 gather with the Wide&Deep column routing (embed_fwd_ring_wide),
alternated inside ONE process on the same tables, every mode measured several times in rotation (the first measurement after a different kernel
ran reads a few us high: compare within the rotation).  usage: probe_pair.py [rotations]"""
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
t(wide); t(wide)
for r in range(reps):
    out = []
    for m in os.environ.get("MODES", "1,2,4,8,1,2,4,8").split(","):
        os.environ["NRX_WIDE_STEP"] = m
        out.append(f"{m}: {t(wide):.1f}")
    print(f"rotation {r}: features per step -> us   " + "   ".join(out), flush=True)
os.environ["NRX_WIDE_STEP"] = "1"
for r in range(2):
    os.environ["NRX_FWD_PAIR"] = "0"; p0 = t(plain)
    os.environ["NRX_FWD_PAIR"] = "1"; p1 = t(plain)
    os.environ["NRX_FWD_PAIR"] = "0"; p0b = t(plain)
    print(f"plain concat one/two/one per step {p0:.1f} / {p1:.1f} / {p0b:.1f} us", flush=True)
