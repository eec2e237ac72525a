#!/usr/bin/env python3
"""Host-side cost per call of the module path (tiny batch so the kernel time is negligible)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_SPARSE
dev = torch.device("cuda:0")
F, D, rows, B = 26, 16, 1000, 64
tables = [torch.randn(rows, D, device=dev) for _ in range(F)]
plan = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=1) for i in range(F)], out_width=F * D, use_fm=True)
ids = [torch.randint(1, rows, (B,), device=dev) for _ in range(F)]
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for mode in ("sync", "off"):
    ops.set_index_check(mode)
    with torch.no_grad():
        print(f"embed_apply (26 feats, no_grad, index check {mode:4s}): {t(lambda: ops.embed_apply(plan, tables, ids, [None] * F)):7.1f} us/call")
call = ops.PreparedEmbed(plan, tables, ids, [None] * F)
print(f"PreparedEmbed.run                                  : {t(call.run):7.1f} us/call")
tg = [x.clone().requires_grad_(True) for x in tables]
print(f"embed_apply with grad graph (index check off)      : {t(lambda: ops.embed_apply(plan, tg, ids, [None] * F)):7.1f} us/call")
