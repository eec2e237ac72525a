#!/usr/bin/env python3
"""Dev: the DEFAULT (dense-gradient) embedding step of the C4 tower launch (history bag L = 50 masked mean + item id over a 200 k-row table,
user id over a 1 M-row table, D = 16): float-atomic scatter against the deterministic planned reduction, alternated inside one process; eager
forward + backward and the same step captured in a HIP graph (GPU time).  usage: ab_dense_tower.py [batches=8192,16384]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_SPARSE
batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8192,16384").split(",")]
dev = torch.device("cuda:0")
ops.set_index_check("deferred")
L, D = 50, 16
gen = torch.Generator(device=dev).manual_seed(5)
tabs = [torch.randn(1_000_000, D, device=dev).requires_grad_(), torch.randn(200_000, D, device=dev).requires_grad_()]
MODES = {"atomic": False, "planned (sorted planner)": True}
for B in batches:
    lens = torch.randint(0, L + 1, (B,), device=dev, generator=gen)
    mask = (torch.arange(L, device=dev)[None] < lens[:, None]).float()
    ids = torch.randint(1, 200_000, (B, L), device=dev, generator=gen) * mask.long()
    uid = torch.randint(1, 1_000_000, (B,), device=dev, generator=gen)
    iid = torch.randint(1, 200_000, (B,), device=dev, generator=gen)
    plan = ops.EmbedPlan([ops.Slot("h", NRX_BAG_MASKED_MEAN, 1, D, L, 0), ops.Slot("i", NRX_SPARSE, 1, D, 0, D), ops.Slot("u", NRX_SPARSE, 0, D, 0, 2 * D)], out_width=3 * D)
    up = torch.randn(B, 3 * D, device=dev)
    def step():
        o = ops.embed_apply(plan, tabs, [ids, iid, uid], [mask, None, None])[0]
        o.backward(up)
        for t in tabs:
            t.grad = None
    res = {k: [] for k in MODES}
    gres = {}
    for rnd in range(3):
        for name, srt in MODES.items():
            ops.DENSE_BWD_SORTED = srt
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
                gres[name] = a.elapsed_time(b) / 40 * 1e3
                del g
    print(f"C4 tower, B = {B}:")
    for name in MODES:
        print(f"  {name:28s} eager step {min(res[name]):7.1f} us (rounds: {', '.join('%.0f' % x for x in res[name])})   graph replay {gres[name]:7.1f} us")
