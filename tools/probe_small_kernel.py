#!/usr/bin/env python3
"""Dev probe: the one-block-per-sample forward kernel (embed_fwd_small_kernel) at B = 512 on full-size tables, for rocprofv3 --kernel-trace:
200 back-to-back launches over 8 rotating input sets, then 100 launches each behind a 1 GiB fill (cold caches).
usage: probe_small_kernel.py [tower | fm | fm_plain | fm_small | fm_nosums]   (tower: user id + history L = 50 + item id over 10 M / 200 k rows;
fm: 26 x 1 M x 16 with the FM epilogue and field sums; fm_plain: the same gather without FM; fm_small: 100 k-row tables)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_SPARSE, NRX_BAG_MASKED_MEAN
dev = "cuda:0"; B = 512; L = 50
ut, nt = torch.randn(10_000_000, 16, device=dev), torch.randn(200_000, 16, device=dev)
calls = []
for s in range(8):
    uid = torch.randint(1, 10_000_000, (B,), device=dev); iid = torch.randint(1, 200_000, (B,), device=dev)
    hist = torch.randint(1, 200_000, (B, L), device=dev); mask = (torch.rand(B, L, device=dev) < 0.7).float()
    plan4 = ops.EmbedPlan([ops.Slot("u", NRX_SPARSE, 0, 16, 0, 0), ops.Slot("h", NRX_BAG_MASKED_MEAN, 1, 16, L, 16), ops.Slot("i", NRX_SPARSE, 1, 16, 0, 32)], out_width=48)
    calls.append(ops.PreparedEmbed(plan4, [ut, nt], [uid, hist, iid], [None, mask, None]))
if len(sys.argv) > 1 and sys.argv[1].startswith("fm"):
    F, D, rows = 26, 16, (100_000 if "small" in sys.argv[1] else 1_000_000)
    use_fm = "plain" not in sys.argv[1]
    tabs = [torch.randn(rows, D, device=dev) for _ in range(F)]
    plan = ops.EmbedPlan([ops.Slot(f"f{i:02d}", NRX_SPARSE, i, D, 0, i * D, fm_field=int(use_fm)) for i in range(F)], out_width=F * D, use_fm=use_fm)
    calls = []
    for s in range(8):
        ids = [torch.randint(1, rows, (B,), device=dev) for _ in range(F)]
        calls.append(ops.PreparedEmbed(plan, tabs, ids, [None] * F, out=torch.empty(B, F * D, device=dev), fm=torch.empty(B, device=dev) if use_fm else None, fm_sums=torch.empty(B, D, device=dev) if (use_fm and "nosums" not in sys.argv[1]) else None))
big = torch.empty(1 << 28, device=dev)          # 1 GiB: a fill between launches evicts the caches
for i in range(200):
    calls[i % 8].run()
torch.cuda.synchronize()
for i in range(100):
    big.fill_(1.0)
    calls[i % 8].run()
torch.cuda.synchronize()
