"""Rows looked up thousands of times in one launch (hot ids of a skewed click log) are reduced as SEVERAL 256-entry work items by
different wavefronts -- possibly on different XCDs -- and the wavefront that finishes a row's last item adds the items' partial sums
(`sorted_long_kernel`, csrc/nrx_embed.hip).  The hand-over of the partials inside the launch uses no fence (round 5's `__threadfence()`
pair cost a write-back of the XCD's L2 per item: Zipf forward + backward 281 -> 414 us): the partials are written through the L2 with
agent-scope stores and read with agent-scope loads behind a device-scope count (rows of more than 32 items count their items in groups
of 32, whose partials are added by each group's last finisher: one counter for ~1000 items serialises at the memory side).  What must hold -- and what a stale or torn partial would
break -- is checked here at the bench's own batch size on Zipf(1.05) ids (SURVEY 8d's second distribution):

  * the row-sparse gradient equals a float64 index_add of the upstream rows (autograd of the reference's nn.Embedding lookups,
    src/model/BaseModel/base_model.py:262-271) to fp32 summation tolerance, for every row including the hottest;
  * ten runs of the same launch give the same bits (the reduction order is fixed; a race on the hand-over would show here)."""
import numpy as np
import pytest
import torch

from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_SPARSE

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _zipf(rng, rows, shape, alpha=1.05):
    # inverse-CDF draw of a Zipf(alpha) law clipped to the table (rank r with weight r^-alpha; bench.py's `--ids zipf` draws the continuous form)
    w = np.arange(1, rows, dtype=np.float64) ** -alpha
    cdf = np.cumsum(w) / w.sum()
    return (1 + np.searchsorted(cdf, rng.random(shape))).astype(np.int64).clip(1, rows - 1)


def _coo_grads(plan, tables, inputs, weights, up):
    ts = [t.clone().requires_grad_() for t in tables]
    out = ops.embed_apply(plan, ts, inputs, weights, sparse_grad=True)[0]
    (out * up).sum().backward()
    torch.cuda.synchronize()
    return [t.grad.to_dense() for t in ts]


@pytest.mark.parametrize("D", [16, 64])
def test_multi_item_rows_single_valued_zipf(D):
    rng = np.random.default_rng(41 + D)
    n, rows, B = 6, 200_000, 65_536
    slots = [ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D) for i in range(n)]
    plan = ops.EmbedPlan(slots, out_width=n * D)
    tables = [torch.from_numpy(rng.standard_normal((rows, D)).astype(np.float32)).to(DEV) for _ in range(n)]
    ids = [_zipf(rng, rows, (B,)) for _ in range(n)]
    hottest = max(int(np.bincount(x).max()) for x in ids)
    assert hottest > 4 * 256, hottest                       # several 256-entry items per hot row: the path under test
    ids[0][rng.random(B) < 0.3] = 7                         # one row with ~77 items: its items are counted in groups of 32 (two-level hand-over)
    assert int(np.bincount(ids[0]).max()) > 40 * 256
    inputs = [torch.from_numpy(x).to(DEV) for x in ids]
    up = torch.from_numpy(rng.standard_normal((B, n * D)).astype(np.float32)).to(DEV)
    first = _coo_grads(plan, tables, inputs, [None] * n, up)
    for i in range(n):
        ref = torch.zeros((rows, D), dtype=torch.float64, device=DEV)
        ref.index_add_(0, inputs[i], up[:, i * D:(i + 1) * D].double())
        scale = float(ref.abs().max())
        torch.testing.assert_close(first[i].double(), ref, rtol=1e-5, atol=2e-6 * scale)
    for _ in range(9):
        again = _coo_grads(plan, tables, inputs, [None] * n, up)
        for a, b in zip(first, again):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_multi_item_rows_history_bag_zipf():
    """The DSSM tower shape at a batch where the news table's hot rows take hundreds of items (C4 with Zipf ids: 3.3 M lookups of one table)."""
    rng = np.random.default_rng(77)
    D, L, B, news, users = 16, 50, 32_768, 200_000, 1_000_000
    slots = [ops.Slot("item_id", NRX_SPARSE, 0, D, 0, 0), ops.Slot("user_history", NRX_BAG_MASKED_MEAN, 0, D, L, D),
             ops.Slot("user_id", NRX_SPARSE, 1, D, 0, 2 * D)]
    plan = ops.EmbedPlan(slots, out_width=3 * D)
    tables = [torch.from_numpy(rng.standard_normal((r, D)).astype(np.float32)).to(DEV) for r in (news, users)]
    hist = _zipf(rng, news, (B, L))
    lens = rng.integers(1, L + 1, B)
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.float32)
    hist = np.where(mask > 0, hist, 0)
    assert int(np.bincount(hist.reshape(-1))[1:].max()) > 64 * 256
    inputs = [torch.from_numpy(_zipf(rng, news, (B,))).to(DEV), torch.from_numpy(hist).to(DEV),
              torch.from_numpy(rng.integers(1, users, (B,))).to(DEV)]
    weights = [None, torch.from_numpy(mask).to(DEV), None]
    up = torch.from_numpy(rng.standard_normal((B, 3 * D)).astype(np.float32)).to(DEV)
    first = _coo_grads(plan, tables, inputs, weights, up)
    # float64 restatement of autograd through array_feature_pooling (base_model.py:273-282): d out / d row = mask / (sum mask + 1e-8)
    w = torch.from_numpy(mask).to(DEV).double()
    scale_l = w / (w.sum(1, keepdim=True) + 1e-8)
    ref = torch.zeros((news, D), dtype=torch.float64, device=DEV)
    ref.index_add_(0, inputs[0], up[:, :D].double())
    g_h = (up[:, D:2 * D].double()[:, None, :] * scale_l[:, :, None]).reshape(-1, D)
    ref.index_add_(0, inputs[1].reshape(-1), g_h)
    ref[0] = 0
    torch.testing.assert_close(first[0].double(), ref, rtol=1e-5, atol=2e-6 * float(ref.abs().max()))
    for _ in range(5):
        again = _coo_grads(plan, tables, inputs, weights, up)
        for a, b in zip(first, again):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
