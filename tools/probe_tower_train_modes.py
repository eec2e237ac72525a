#!/usr/bin/env python3
"""Dev probe: the DSSM user-tower launch (history bag L = 50 masked mean + item id sharing the 200 k-row news table, user id over 1 M rows, D = 16)
at the reference's batch sizes: forward + backward + optimizer step, dense gradients + AdamW(fused kernel) against SparseGradSink + FusedSparseAdam."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_SPARSE
from news_recsys_amd.model.model_utils.optim import FusedSparseAdam, dense_adamw
ops.set_index_check("off")
dev = torch.device("cuda:0"); D, L = 16, 50
gen = torch.Generator(device=dev).manual_seed(5)
plan = ops.EmbedPlan([ops.Slot("h", NRX_BAG_MASKED_MEAN, 1, D, L, 0), ops.Slot("i", NRX_SPARSE, 1, D, 0, D), ops.Slot("u", NRX_SPARSE, 0, D, 0, 2 * D)], out_width=3 * D)
def timed(fn, n=100, reps=3):
    best = 1e9
    for _ in range(reps):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best
for B in [int(x) for x in os.environ.get("NRX_PROBE_BATCHES", "512,1024").split(",")]:
    lens = torch.randint(0, L + 1, (B,), device=dev, generator=gen)
    mask = (torch.arange(L, device=dev)[None] < lens[:, None]).float()
    ids = [torch.randint(1, 200_000, (B, L), device=dev, generator=gen) * mask.long(), torch.randint(1, 200_000, (B,), device=dev, generator=gen),
           torch.randint(1, 1_000_000, (B,), device=dev, generator=gen)]
    ws = [mask, None, None]
    up = torch.randn(B, 3 * D, device=dev) * 1e-3
    for mode in ("dense", "fused"):
        tabs = [torch.randn(1_000_000, D, device=dev).requires_grad_(True), torch.randn(200_000, D, device=dev).requires_grad_(True)]
        if mode == "dense":
            opt = dense_adamw(tabs, lr=1e-3, capturable=True)
            def step():
                opt.zero_grad(set_to_none=True)
                out = ops.embed_apply(plan, tabs, ids, ws)[0]
                out.backward(up)
                opt.step()
        else:
            sink = ops.SparseGradSink(); opt = FusedSparseAdam(sink, lr=1e-3)
            def step():
                out = ops.embed_apply(plan, tabs, ids, ws, sparse_grad=sink)[0]
                out.backward(up)
                opt.step()
        te = timed(step); tg = float("nan")
        try:
            s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3): step()
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step()
            tg = timed(g.replay)
        except Exception as e:
            print(f"   ({mode}: capture failed: {type(e).__name__}: {str(e)[:120]})")
        print(f"B={B:5d} {mode:6s}: eager {te:7.1f} us   graph replay {tg:7.1f} us", flush=True)
