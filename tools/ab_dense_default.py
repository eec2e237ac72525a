#!/usr/bin/env python3
"""Dev: the DEFAULT (dense-gradient) embedding step of a C2-shaped launch (26 single-valued features), float-atomic scatter against the
deterministic planned reduction with either planner, the modes ALTERNATED inside one process (separate runs of this host-bound step differ by
more than the modes do).  Eager forward + backward (ops.embed_apply + autograd), and the same step captured in a HIP graph (GPU time).
usage: ab_dense_default.py [rows=1000000] [batches=8192,32768]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_SPARSE
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
batches = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8192,32768").split(",")]
dev = torch.device("cuda:0")
ops.set_index_check("deferred")
F, D = 26, 16
gen = torch.Generator(device=dev).manual_seed(5)
tabs = [torch.randn(rows, D, device=dev).requires_grad_() for _ in range(F)]
MODES = {"atomic": (False, "0"), "planned, one-kernel planner": (True, "auto"), "planned, sorted planner": (True, "0")}
for B in batches:
    ids = [torch.randint(1, rows, (B,), device=dev, generator=gen) for _ in range(F)]
    plan = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D) for i in range(F)], out_width=F * D)
    up = torch.randn(B, F * D, device=dev)
    def step():
        o = ops.embed_apply(plan, tabs, ids, [None] * F)[0]
        o.backward(up)
        for t in tabs:
            t.grad = None
    res = {k: [] for k in MODES}
    gres = {k: [] for k in MODES}
    for rnd in range(3):
        for name, (srt, lds) in MODES.items():
            ops.DENSE_BWD_SORTED, ops.PLAN_LDS = srt, lds
            plan.__dict__.pop("_sg", None)
            for _ in range(8):
                step()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(40):
                step()
            b.record(); torch.cuda.synchronize()
            res[name].append(a.elapsed_time(b) / 40 * 1e3)
            if rnd == 0:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for _ in range(3):
                        step()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    step()
                for _ in range(5):
                    g.replay()
                torch.cuda.synchronize()
                a.record()
                for _ in range(40):
                    g.replay()
                b.record(); torch.cuda.synchronize()
                gres[name].append(a.elapsed_time(b) / 40 * 1e3)
                del g
    print(f"26 x {rows} rows, B = {B}:")
    for name in MODES:
        print(f"  {name:30s} eager step {min(res[name]):7.1f} us (rounds: {', '.join('%.0f' % x for x in res[name])})   graph replay {gres[name][0]:7.1f} us")
