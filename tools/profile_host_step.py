#!/usr/bin/env python3
"""Dev: where the HOST time of an eager embedding forward + dense-gradient backward goes at the reference's batch sizes (cProfile over 300
steps of the C2-shaped launch, B = 512).  The step is host-bound there (~140 us of Python / autograd per step for ~25 us of GPU work)."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_SPARSE
ops.set_index_check(os.environ.get("NRX_PROBE_INDEX_CHECK", "deferred"))
dev = torch.device("cuda:0"); D, F, B = 16, 26, int(os.environ.get("NRX_PROBE_B", 512))
gen = torch.Generator(device=dev).manual_seed(5)
tabs = [torch.randn(100_000, D, device=dev).requires_grad_(True) for _ in range(F)]
plan = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=1) for i in range(F)], out_width=F * D, use_fm=True)
ids = [torch.randint(1, 100_000, (B,), device=dev, generator=gen) for _ in range(F)]
up, upf = torch.randn(B, F * D, device=dev), torch.randn(B, device=dev)
def step():
    for t in tabs: t.grad = None
    out, _, fm = ops.embed_apply(plan, tabs, ids, [None] * F)
    torch.autograd.backward([out, fm], [up, upf])
for _ in range(50): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize()
print(f"B={B}: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per step (wall, eager)")
def fwd_only():
    with torch.no_grad():
        ops.embed_apply(plan, tabs, ids, [None] * F)
for _ in range(50): fwd_only()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): fwd_only()
torch.cuda.synchronize()
print(f"forward only (no grad): {(time.perf_counter() - t0) / 300 * 1e6:.1f} us")
def fwd_grad():
    out, _, fm = ops.embed_apply(plan, tabs, ids, [None] * F)
for _ in range(50): fwd_grad()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): fwd_grad()
torch.cuda.synchronize()
print(f"forward only (autograd node built): {(time.perf_counter() - t0) / 300 * 1e6:.1f} us")
torch.autograd.set_multithreading_enabled(False)       # the backward on THIS thread (what BaseModel.backward does): visible to the profiler
for _ in range(50): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize()
print(f"same-thread backward: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
