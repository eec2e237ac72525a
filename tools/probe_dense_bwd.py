#!/usr/bin/env python3
"""Dev probe: the DEFAULT (dense-gradient) backward of the fused gather -- what a drop-in user gets with sparse_grad off, i.e. the
reference's nn.Embedding(sparse=False) -- at the C4 tower shape (history bag L = 50, masked mean) and the C2 shape (26 single ids).
Prints the autograd backward alone (in the sorted mode the plan made at forward time is reused: reduction + store only), the zero
fill of the gradient tables (the part no kernel change can remove), and a whole forward + backward step.  NRX_DENSE_BWD=atomic
selects the float-atomic scatter."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_SPARSE
ops.set_index_check(os.environ.get("NRX_PROBE_INDEX_CHECK", "deferred"))      # "sync" (the module default) reads a status word per forward: the host then waits for the GPU every step
dev = torch.device("cuda:0"); B, L, D = int(os.environ.get("NRX_PROBE_B", 65536)), 50, 16
gen = torch.Generator(device=dev).manual_seed(5)
import time
HOST = [0.0]
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(n): fn()
    e.record()
    HOST[0] = (time.perf_counter() - t0) / n * 1e6          # host time per call of the un-synchronised loop
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
def case(name, plan, tables, ins, ws):
    ts = [t.clone().requires_grad_(True) for t in tables]
    out = ops.embed_apply(plan, ts, ins, ws)[0]
    up = torch.randn_like(out)
    us = timed(lambda: torch.autograd.grad(out, ts, up, retain_graph=True))
    uz = timed(lambda: [torch.zeros_like(t) for t in ts])
    def step():
        o = ops.embed_apply(plan, ts, ins, ws)[0]
        o.backward(up)
        for t in ts: t.grad = None
    ust = timed(step); hst = HOST[0]
    with torch.no_grad():
        uf = timed(lambda: ops.embed_apply(plan, ts, ins, ws))
    print(f"{name}: backward alone (plan reused) {us:7.1f} us [zero fill {uz:6.1f}]   forward {uf:6.1f} us   forward + backward step (plans each step) {ust:7.1f} us (host {hst:6.1f} us per step)", flush=True)
news = torch.randn(200_000, D, device=dev); users = torch.randn(1_000_000, D, device=dev)
lens = torch.randint(0, L + 1, (B,), device=dev, generator=gen)
mask = (torch.arange(L, device=dev)[None] < lens[:, None]).float()
ids = torch.randint(1, 200_000, (B, L), device=dev, generator=gen) * mask.long()
uid = torch.randint(1, 1_000_000, (B,), device=dev, generator=gen); iid = torch.randint(1, 200_000, (B,), device=dev, generator=gen)
p4 = ops.EmbedPlan([ops.Slot("h", NRX_BAG_MASKED_MEAN, 1, D, L, 0), ops.Slot("i", NRX_SPARSE, 1, D, 0, D), ops.Slot("u", NRX_SPARSE, 0, D, 0, 2 * D)], out_width=3 * D)
F = 26
t2 = [torch.randn(100_000, D, device=dev) for _ in range(F)]
i2 = [torch.randint(1, 100_000, (B,), device=dev, generator=gen) for _ in range(F)]
p2 = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D) for i in range(F)], out_width=F * D)
for _ in range(2):
    case("C4 tower (1 M users, 200 k news)", p4, [users, news], [ids, iid, uid], [mask, None, None])
    case("C2 (26 x 100 k rows)            ", p2, t2, i2, [None] * F)
