"""Dev: the reference's own feature set (train_cf_deep.yaml: 5 features of widths 16 / 32) through one generic launch vs one uniform
launch per width, by batch size -- where the split threshold of nrx_embed_fwd (NRX_SPLIT_MIN_LOOKUPS) comes from."""
import torch, sys, os
sys.path.insert(0, ".")
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_SPARSE
dev="cuda:0"
dims = [16, 32, 16, 16, 32]
rows = [18, 65239, 270, 18, 94058]
for B in (512, 4096, 16384, 65536):
    gen = torch.Generator(device=dev).manual_seed(6)
    tables = [torch.randn(r, d, device=dev) for r, d in zip(rows, dims)]
    ids = [torch.randint(1, r, (B,), device=dev, generator=gen) for r in rows]
    col, slots = 0, []
    for i, d in enumerate(dims):
        slots.append(ops.Slot(f"f{i}", NRX_SPARSE, i, d, 0, col)); col += d
    plan = ops.EmbedPlan(slots, out_width=col)
    prep = ops.PreparedEmbed(plan, tables, ids, [None] * 5)
    res = {}
    for m in ("6", "2"):
        os.environ["NRX_SPLIT_MIN_LOOKUPS"] = "0" if m == "2" else str(1 << 60)
        for _ in range(200): prep.run()
        torch.cuda.synchronize()
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(500): prep.run()
        b.record(); torch.cuda.synchronize()
        res[m]=a.elapsed_time(b)*2
    print(f"B={B}: generic (one launch) {res['6']:.1f} us   split (one uniform launch per width: 2) {res['2']:.1f} us")
