#!/usr/bin/env python3
"""Dev probe: embedding forward + backward + optimizer step of the C2-shaped launch (26 x 100 k rows, D = 16, FM folded) at the reference's batch
sizes, in the two training modes: (dense) default dense gradients + torch.optim.Adam(foreach) over the whole tables, (fused) row-sparse sink +
FusedSparseAdam touching only the looked-up rows.  Eager wall time per step and, where the step captures, HIP-graph replay."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_SPARSE
from news_recsys_amd.model.model_utils.optim import ExactDenseAdamW, FusedSparseAdam
ops.set_index_check("off")
dev = torch.device("cuda:0"); D, F = 16, 26
gen = torch.Generator(device=dev).manual_seed(5)
plan = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=1) for i in range(F)], out_width=F * D, use_fm=True)
def timed(fn, n=100, reps=3):
    best = 1e9
    for _ in range(reps):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best
for B in [int(x) for x in os.environ.get("NRX_PROBE_BATCHES", "512,2048").split(",")]:
    ids = [torch.randint(1, 100_000, (B,), device=dev, generator=gen) for _ in range(F)]
    up, upf = torch.randn(B, F * D, device=dev) * 1e-3, torch.randn(B, device=dev) * 1e-3
    for mode in ("dense", "dense_fusedkernel", "exact", "fused"):
        tabs = [torch.randn(100_000, D, device=dev).requires_grad_(True) for _ in range(F)]
        if mode.startswith("dense"):
            opt = (torch.optim.AdamW(tabs, lr=1e-3, capturable=True, foreach=True) if mode == "dense" else
                   torch.optim.AdamW(tabs, lr=1e-3, capturable=True, fused=True))      # torch's one-pass multi-tensor kernel
            def step():
                opt.zero_grad(set_to_none=True)
                out, _, fm = ops.embed_apply(plan, tabs, ids, [None] * F)
                torch.autograd.backward([out, fm], [up, upf])
                opt.step()
        else:
            sink = ops.SparseGradSink()
            opt = FusedSparseAdam(sink, lr=1e-3, capturable=True) if mode == "fused" else ExactDenseAdamW(sink, tabs, lr=1e-3, capturable=True)      # exact: dense AdamW over every row, fed from the sink
            def step():
                out, _, fm = ops.embed_apply(plan, tabs, ids, [None] * F, sparse_grad=sink)
                torch.autograd.backward([out, fm], [up, upf])
                opt.step()
        te = timed(step)
        tg = float("nan")
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
        print(f"B={B:5d} {mode:17s}: eager {te:7.1f} us   graph replay {tg:7.1f} us", flush=True)
