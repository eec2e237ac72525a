#!/usr/bin/env python3
"""Dev probe: embedding forward + dense-gradient backward of the C2-shaped launch (26 single-valued features, FM folded) at the reference's
batch sizes -- eager and captured in a HIP graph -- with the table gradients formed by (auto) the one-launch deterministic kernel
(nrx_embed_bwd_small), (atomic) the float-atomic scatter, (sorted) the planned reduction.  NRX_DENSE_BWD selects; one process per mode."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_SPARSE
ops.set_index_check("off")
dev = torch.device("cuda:0"); D, F = 16, 26
mode = os.environ.get("NRX_DENSE_BWD", "auto")
gen = torch.Generator(device=dev).manual_seed(5)
tabs = [torch.randn(100_000, D, device=dev).requires_grad_(True) for _ in range(F)]
plan = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=1) for i in range(F)], out_width=F * D, use_fm=True)
def timed(fn, n=200, reps=5):
    best = 1e9
    for _ in range(reps):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best
for B in [int(x) for x in os.environ.get("NRX_PROBE_BATCHES", "512,1024,2048,4096").split(",")]:
    ids = [torch.randint(1, 100_000, (B,), device=dev, generator=gen) for _ in range(F)]
    up, upf = torch.randn(B, F * D, device=dev), torch.randn(B, device=dev)
    def step():
        out, _, fm = ops.embed_apply(plan, tabs, ids, [None] * F)
        torch.autograd.backward([out, fm], [up, upf])
        return tabs[0].grad
    for t in tabs: t.grad = None
    step(); torch.cuda.synchronize()
    def eager():
        for t in tabs: t.grad = None
        step()
    te = timed(eager)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): eager()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    for t in tabs: t.grad = None
    with torch.cuda.graph(g):
        step()
    tg = timed(g.replay)
    g.replay(); torch.cuda.synchronize(); a = [t.grad.clone() for t in tabs]
    g.replay(); torch.cuda.synchronize()
    same = all(torch.equal(x.view(torch.int32), t.grad.view(torch.int32)) for x, t in zip(a, tabs))
    print(f"{mode:7s} B={B:5d}: eager {te:7.1f} us   graph replay {tg:7.1f} us   replays bit-identical: {same}", flush=True)
