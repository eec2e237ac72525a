#!/usr/bin/env python3
"""Dev probe: the C4 tower (user id + history bag L = 50 over a 200 k-row table + item id, D = 16, B = 65 536) with the history as padded ids + mask
and as CSR values + offsets (what ColumnarLoader(csr_bags=True) delivers); bag lengths ~ U{0..50}."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_SPARSE, NRX_FEAT_BAG_CSR
dev = torch.device("cuda:0"); B, L, D = 65536, 50, 16
gen = torch.Generator(device=dev).manual_seed(5)
news = torch.randn(200_000, D, device=dev); users = torch.randn(10_000_000, D, device=dev)
lens = torch.randint(0, L + 1, (B,), device=dev, generator=gen)
mask = (torch.arange(L, device=dev)[None] < lens[:, None]).float()
ids = torch.randint(1, 200_000, (B, L), device=dev, generator=gen) * mask.long()
offsets = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), lens.cumsum(0)])
values = ids[mask.bool()].contiguous()
uid = torch.randint(1, 10_000_000, (B,), device=dev, generator=gen); iid = torch.randint(1, 200_000, (B,), device=dev, generator=gen)
def timed(call, n=200):
    for _ in range(20): call.run()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): call.run()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
pp = ops.EmbedPlan([ops.Slot("h", NRX_BAG_MASKED_MEAN, 1, D, L, 0), ops.Slot("i", NRX_SPARSE, 1, D, 0, D), ops.Slot("u", NRX_SPARSE, 0, D, 0, 2 * D)], out_width=3 * D)
pc = ops.EmbedPlan([ops.Slot("h", NRX_BAG_MASKED_MEAN, 1, D, L, 0, flags=NRX_FEAT_BAG_CSR), ops.Slot("i", NRX_SPARSE, 1, D, 0, D), ops.Slot("u", NRX_SPARSE, 0, D, 0, 2 * D)], out_width=3 * D)
cp = ops.PreparedEmbed(pp, [users, news], [ids, iid, uid], [mask, None, None])
cc = ops.PreparedEmbed(pc, [users, news], [values, iid, uid], [offsets, None, None])
op, oc = cp.run(), cc.run()
torch.cuda.synchronize()
print("CSR == padded:", torch.equal(op[0] if isinstance(op, tuple) else op, oc[0] if isinstance(oc, tuple) else oc))
for _ in range(2):
    print(f"padded ids + mask: {timed(cp):.1f} us   CSR values + offsets: {timed(cc):.1f} us")
