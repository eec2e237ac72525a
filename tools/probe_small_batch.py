#!/usr/bin/env python3
"""Dev: the path's kernels at the reference's own batch size (B = 512) -- run under rocprofv3 --kernel-trace --stats to see each
kernel's fixed cost (a per-launch constant hides at B = 65536 and dominates here)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_SPARSE, NRX_BAG_MASKED_MEAN
dev = "cuda:0"; B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
gen = torch.Generator(device=dev).manual_seed(1)
# DeepFM-like: 26 x 100k x 16 with FM, row-sparse backward
F, D, rows = 26, 16, 100_000
tabs = [torch.randn(rows, D, device=dev) for _ in range(F)]
ids = [torch.randint(1, rows, (B,), device=dev, generator=gen) for _ in range(F)]
plan = ops.EmbedPlan([ops.Slot(f"f{i:02d}", NRX_SPARSE, i, D, 0, i * D, fm_field=1) for i in range(F)], out_width=F * D, use_fm=True)
out = torch.empty(B, F * D, device=dev); fm = torch.empty(B, device=dev); sums = torch.empty(B, D, device=dev)
fwd = ops.PreparedEmbed(plan, tabs, ids, [None] * F, out=out, fm=fm, fm_sums=sums)
bwd = ops.PreparedSparseBackward(fwd, torch.randn(B, F * D, device=dev), torch.randn(B, device=dev))
# DSSM tower-like: id + history L = 50 + id
ut, nt = torch.randn(100_000, 16, device=dev), torch.randn(20_000, 16, device=dev)
uid = torch.randint(1, 100_000, (B,), device=dev); iid = torch.randint(1, 20_000, (B,), device=dev)
hist = torch.randint(1, 20_000, (B, 50), device=dev); mask = (torch.rand(B, 50, device=dev) < 0.7).float()
plan4 = ops.EmbedPlan([ops.Slot("u", NRX_SPARSE, 0, 16, 0, 0), ops.Slot("h", NRX_BAG_MASKED_MEAN, 1, 16, 50, 16), ops.Slot("i", NRX_SPARSE, 1, 16, 0, 32)], out_width=48)
fwd4 = ops.PreparedEmbed(plan4, [ut, nt], [uid, hist, iid], [None, mask, None])
bwd4 = ops.PreparedSparseBackward(fwd4, torch.randn(B, 48, device=dev))
x = torch.randn(B, 112, device=dev); W = torch.randn(3, 112, 112, device=dev) / 10; b = torch.zeros(3, 112, device=dev)
w1 = torch.randn(3, 112, device=dev) / 10; b1 = torch.zeros(3, 112, device=dev)
xg = x.clone().requires_grad_(True); W2 = W.clone().requires_grad_(True); b2 = b.clone().requires_grad_(True)
w1g = w1.clone().requires_grad_(True); b1g = b1.clone().requires_grad_(True)
for _ in range(60):
    fwd.run(); bwd.run(); fwd4.run(); bwd4.run()
    with torch.no_grad():
        ops.dcn_v2(x, W, b); ops.dcn_v1(x, w1, b1)
    o = ops.dcn_v2(xg, W2, b2); torch.autograd.grad(o, (xg, W2, b2), x)
    o = ops.dcn_v1(xg, w1g, b1g); torch.autograd.grad(o, (xg, w1g, b1g), x)
torch.cuda.synchronize()
