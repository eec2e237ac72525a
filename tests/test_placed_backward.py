"""Row-sparse backward in placement form (nrx_sparse_plan_place + nrx_embed_bwd_placed) vs the plain sorted walk
(nrx_sparse_plan + nrx_embed_bwd_sorted), which earlier tests tie to the reference's gradients (goldens, fp64 restatements,
the dense-gradient path).  Backward of base_model.py:262-308 (+ fm/model.py:18-26, widedeep/model.py:53-69).

Bar: BIT FOR BIT -- same unique rows, same row gradients down to the sign of zero (compared as int32 words): a placed
row is the very value the walk would have formed (0 + upstream row), every other row is walked in the same sorted order."""
import numpy as np
import pytest
import torch

from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM, NRX_SPARSE

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ids(rng, rows, shape, dist):
    if dist == "uniform":
        x = rng.integers(0, rows, shape)
    elif dist == "unique":                                   # (nearly) every row looked up once: everything is placed
        x = rng.permutation(rows)[:int(np.prod(shape))].reshape(shape)
    else:                                                    # zipf: rows looked up hundreds of times -> the work lists
        x = np.minimum(rng.zipf(1.2, shape) - 1, rows - 1)
    x = np.asarray(x, np.int64)
    x.reshape(-1)[:3] = 0                                    # the padding row is looked up too
    return x


def _grads(plan, tables, inputs, weights, g_out, g_wide, g_fm, place, monkeypatch):
    monkeypatch.setattr(ops, "SPARSE_PLACE", place)
    ts = [t.clone().requires_grad_() for t in tables]
    out, wide, fm = ops.embed_apply(plan, ts, inputs, weights, sparse_grad=True)
    loss = (out * g_out).sum()
    if wide is not None:
        loss = loss + (wide * g_wide).sum()
    if fm is not None:
        loss = loss + (fm * g_fm).sum()
    loss.backward()
    torch.cuda.synchronize()
    return [t.grad.coalesce() for t in ts]


def _same(ga, gb):
    for a, b in zip(ga, gb):
        assert torch.equal(a.indices(), b.indices())
        assert torch.equal(a.values().view(torch.int32), b.values().view(torch.int32))      # bit for bit, -0 / +0 included


CASES = [
    # name, D, n_feats, rows per table, B, fm, wide feature indices, dist
    ("c2_like_fm", 16, 26, 3000, 1500, True, (), "uniform"),
    ("fm_zipf", 16, 9, 40000, 4000, True, (), "zipf"),
    ("fm_unique", 16, 5, 9000, 1700, True, (), "unique"),
    ("plain32", 32, 6, 2000, 1300, False, (), "uniform"),
    ("plain64_zipf", 64, 5, 100000, 5000, False, (), "zipf"),
    ("plain16_few", 16, 3, 500, 700, False, (), "uniform"),
    ("wide32", 32, 7, 2500, 1100, False, (1, 4), "uniform"),
    ("wide16_zipf", 16, 6, 30000, 3000, False, (0, 5), "zipf"),
]


@pytest.mark.parametrize("name,D,n,rows,B,fm,wide,dist", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("idx", [torch.int64, torch.int32])
def test_placed_backward_equals_sorted_walk(name, D, n, rows, B, fm, wide, dist, idx, monkeypatch):
    rng = np.random.default_rng(len(name) * 1000 + D + n)
    slots, col = [], 0
    for i in range(n):
        w = wide.index(i) if i in wide else -1
        slots.append(ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, col, fm_field=int(fm), wide_col=w))
        col += D - 1 if w >= 0 else D
    plan = ops.EmbedPlan(slots, out_width=col, use_fm=fm, wide_width=len(wide))
    tables = [torch.from_numpy(rng.standard_normal((rows, D)).astype(np.float32)).to(DEV) for _ in range(n)]
    inputs = [torch.from_numpy(_ids(rng, rows, (B,), dist)).to(DEV).to(idx) for _ in range(n)]
    g_out = torch.from_numpy(rng.standard_normal((B, col)).astype(np.float32)).to(DEV)
    g_wide = torch.from_numpy(rng.standard_normal((B, max(len(wide), 1))).astype(np.float32)).to(DEV)[:, :len(wide)]
    g_fm = torch.from_numpy(rng.standard_normal((B,)).astype(np.float32)).to(DEV)
    a = _grads(plan, tables, inputs, [None] * n, g_out, g_wide, g_fm, True, monkeypatch)
    b = _grads(plan, tables, inputs, [None] * n, g_out, g_wide, g_fm, False, monkeypatch)
    _same(a, b)
    assert all(x._nnz() > 0 for x in a)


@pytest.mark.parametrize("kind", [NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM])
@pytest.mark.parametrize("dist", ["uniform", "zipf"])
def test_placed_backward_with_bag_features_equals_sorted_walk(kind, dist, monkeypatch):
    """The DSSM tower shape: item id + history bag (sharing the news table) + user id.  Only the single-valued features'
    lookups may be placed; a news row met once by the HISTORY is walked (its upstream row is scaled, not copied)."""
    rng = np.random.default_rng(17 + kind)
    D, L, B, news, users = 16, 5, 2100, 6000, 50000        # (2 single-valued lookups of 7 per sample: above the quarter from which the plan places)
    slots = [ops.Slot("item_id", NRX_SPARSE, 0, D, 0, 0), ops.Slot("user_history", kind, 0, D, L, D), ops.Slot("user_id", NRX_SPARSE, 1, D, 0, 2 * D)]
    plan = ops.EmbedPlan(slots, out_width=3 * D)
    tables = [torch.from_numpy(rng.standard_normal((r, D)).astype(np.float32)).to(DEV) for r in (news, users)]
    hist = _ids(rng, news, (B, L), dist)
    lens = rng.integers(0, L + 1, B)
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.float32)
    hist = np.where(mask > 0, hist, 0)
    if kind == NRX_BAG_SUM:
        mask = mask * rng.random((B, L)).astype(np.float32)                      # non-binary weights: the per-lookup scale array
    inputs = [torch.from_numpy(_ids(rng, news, (B,), dist)).to(DEV), torch.from_numpy(hist).to(DEV),
              torch.from_numpy(_ids(rng, users, (B,), "unique")).to(DEV)]
    weights = [None, None if kind == NRX_BAG_MEAN else torch.from_numpy(mask).to(DEV), None]
    g_out = torch.from_numpy(rng.standard_normal((B, 3 * D)).astype(np.float32)).to(DEV)
    a = _grads(plan, tables, inputs, weights, g_out, None, None, True, monkeypatch)
    b = _grads(plan, tables, inputs, weights, g_out, None, None, False, monkeypatch)
    _same(a, b)
    # ... and the pre-scaled-row form of the 0/1-weight bags (one row request per lookup) against the per-lookup factor form
    monkeypatch.setenv("NRX_BAG_PRESCALE", "0")
    c = _grads(plan, tables, inputs, weights, g_out, None, None, True, monkeypatch)
    d = _grads(plan, tables, inputs, weights, g_out, None, None, False, monkeypatch)
    monkeypatch.delenv("NRX_BAG_PRESCALE")
    _same(a, c)
    _same(a, d)


@pytest.mark.parametrize("dest", ["row_sparse", "dense"])
def test_padded_history_steps_switch_to_the_padding_split_and_keep_their_gradients(dest, monkeypatch):
    """The DSSM tower on histories padded with id 0 (the reference's DataReader pads every multi-valued feature to max_len): the first step plans
    with the padding lookups in the sort and records how many there were (PadPolicy: a mapped host word, no synchronisation), the next step of
    the same shape sets them aside before the sort (nrx_sparse_plan_ex, NRX_PLAN_SPLIT_PADDING) -- same gradients bit for bit, and equal to
    NRX_PAD_SPLIT=0.  A batch without padding switches back."""
    rng = np.random.default_rng(23)
    D, L, B, news, users = 16, 20, 9000, 50000, 300000
    monkeypatch.setattr(ops, "PAD_SPLIT_MIN", 1000)
    monkeypatch.setattr(ops, "PLAN_AHEAD_MIN", 1000)                 # (the dense mode plans such a launch ahead too: the path that carries the policy)
    monkeypatch.setattr(ops, "DENSE_BWD_SORTED", True)
    slots = [ops.Slot("item_id", NRX_SPARSE, 0, D, 0, 0), ops.Slot("user_history", NRX_BAG_MASKED_MEAN, 0, D, L, D), ops.Slot("user_id", NRX_SPARSE, 1, D, 0, 2 * D)]
    plan = ops.EmbedPlan(slots, out_width=3 * D)
    tables = [torch.from_numpy(rng.standard_normal((r, D)).astype(np.float32)).to(DEV) for r in (news, users)]
    lens = rng.integers(0, L + 1, B)
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.float32)
    hist = np.where(mask > 0, rng.integers(1, news, (B, L)), 0)
    full = rng.integers(1, news, (B, L))
    item, user = rng.integers(1, news, B), rng.integers(1, users, B)
    g_out = torch.from_numpy(rng.standard_normal((B, 3 * D)).astype(np.float32)).to(DEV)

    ts = [t.clone().requires_grad_() for t in tables]               # (the launch groups -- and their policies -- are kept per table set)

    def step(h, m, split):
        monkeypatch.setattr(ops, "PAD_SPLIT", split)
        for t in ts:
            t.grad = None
        out = ops.embed_apply(plan, ts, [torch.from_numpy(item).to(DEV), torch.from_numpy(h).to(DEV), torch.from_numpy(user).to(DEV)],
                              [None, torch.from_numpy(m).to(DEV), None], sparse_grad=dest == "row_sparse")[0]
        (out * g_out).sum().backward()
        torch.cuda.synchronize()
        return [t.grad.coalesce() if dest == "row_sparse" else t.grad for t in ts]

    def same(ga, gb):
        if dest == "row_sparse":
            _same(ga, gb)
        else:
            for a, b in zip(ga, gb):
                assert torch.equal(a.view(torch.int32), b.view(torch.int32))

    ref = step(hist, mask, "0")
    pol = ops._group_pad(ops._sparse_group_cache(plan, ts)[0], plan, B)
    assert pol is None                                              # NRX_PAD_SPLIT=0: no policy, the plain planner calls
    first = step(hist, mask, "auto")
    pol = ops._group_pad(ops._sparse_group_cache(plan, ts)[0], plan, B)
    assert pol is not None and int(pol.stats[4]) == int((hist == 0).sum()) and pol.choose()      # recorded by the first step: the next one splits
    second = step(hist, mask, "auto")
    same(ref, first)
    same(ref, second)
    ones = np.ones_like(mask)
    a = step(full, ones, "auto")                                    # (planned with the split; finds no padding)
    assert int(pol.stats[4]) == 0 and not pol.choose()
    b = step(full, ones, "auto")
    c = step(full, ones, "0")
    same(a, c)
    same(b, c)


@pytest.mark.parametrize("live_zero", [False, True])
@pytest.mark.parametrize("D", [16, 32, 64])
def test_bag_backward_prescaled_rows_with_and_without_weight_bits(live_zero, D, monkeypatch):
    """Masked-mean history: when every zero weight sits on a padding id (DataReader's masks) the reduction reads no weight bits;
    a zero weight on a REAL id (live_zero) must still contribute nothing.  Both against the per-lookup factor form, bit for bit,
    and against a float64 restatement of the pooling backward (base_model.py:273-282)."""
    rng = np.random.default_rng(3 + D + int(live_zero))
    L, B, news = 12, 1800, 700
    plan = ops.EmbedPlan([ops.Slot("user_history", NRX_BAG_MASKED_MEAN, 0, D, L, 0)], out_width=D)
    table = torch.from_numpy(rng.standard_normal((news, D)).astype(np.float32)).to(DEV)
    lens = rng.integers(0, L + 1, B)
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.float32)
    hist = np.where(mask > 0, rng.integers(1, news, (B, L)), 0)
    if live_zero:
        kill = (rng.random((B, L)) < 0.2) & (mask > 0)
        mask = np.where(kill, 0.0, mask).astype(np.float32)              # ids stay: zero weight on live rows
    inputs, weights = [torch.from_numpy(hist).to(DEV)], [torch.from_numpy(mask).to(DEV)]
    g_out = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)).to(DEV)
    a = _grads(plan, [table], inputs, weights, g_out, None, None, True, monkeypatch)
    monkeypatch.setenv("NRX_BAG_PRESCALE", "0")
    b = _grads(plan, [table], inputs, weights, g_out, None, None, True, monkeypatch)
    monkeypatch.delenv("NRX_BAG_PRESCALE")
    _same(a, b)
    den = mask.sum(1, keepdims=True).astype(np.float64) + 1e-8
    contrib = (mask.astype(np.float64) / den)[:, :, None] * g_out.cpu().numpy().astype(np.float64)[:, None, :]
    ref = np.zeros((news, D))
    np.add.at(ref, hist.reshape(-1), contrib.reshape(-1, D))
    ref[0] = 0
    np.testing.assert_allclose(a[0].to_dense().cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


def test_placed_backward_c2_shape_full_batch(monkeypatch):
    """B = 65536, 26 x 1M-row tables (the BASELINE C2 shape), FM folded in: placement form == sorted walk, bit for bit; and the
    placement plan accounts for every unique row exactly once."""
    rng = np.random.default_rng(5)
    D, n, rows, B = 16, 26, 1_000_000, 65536
    slots = [ops.Slot(f"C{i:02d}", NRX_SPARSE, i, D, 0, i * D, fm_field=1) for i in range(n)]
    plan = ops.EmbedPlan(slots, out_width=n * D, use_fm=True)
    gen = torch.Generator(device=DEV).manual_seed(3)
    tables = [torch.randn((rows, D), device=DEV, generator=gen) for _ in range(n)]
    inputs = [torch.randint(0, rows, (B,), device=DEV, generator=gen) for _ in range(n)]
    g_out = torch.randn((B, n * D), device=DEV, generator=gen)
    g_fm = torch.randn((B,), device=DEV, generator=gen)
    a = _grads(plan, tables, inputs, [None] * n, g_out, None, g_fm, True, monkeypatch)
    b = _grads(plan, tables, inputs, [None] * n, g_out, None, g_fm, False, monkeypatch)
    _same(a, b)
    order, uniq, seg, counts, dest, walk, n_walk = ops.sparse_plan(inputs, list(range(n)), [rows] * n, n, place_feats=(1 << n) - 1)
    nu, nw = int(counts[0].item()), int(n_walk.item())
    d = dest.cpu().numpy()
    placed = d[d >= 0]
    assert len(np.unique(placed)) == len(placed) and len(placed) + nw == nu
    assert len(np.intersect1d(placed, walk[:nw].cpu().numpy())) == 0


@pytest.mark.parametrize("n", [1, 2, 3, 7, 9, 26, 27])
@pytest.mark.parametrize("fm", [False, True])
@pytest.mark.parametrize("dist", ["uniform", "zipf"])
def test_full_line_placement_form_on_padded_rows_odd_and_even_feature_counts(n, fm, dist, monkeypatch):
    """The full-line form of the placement pass (embed_bwd_place_lines_kernel: 8 lanes per sample take features 2j and 2j + 1 together) needs
    rows of whole 128-byte lines: D = 16 features on a row stride that is a multiple of 32 floats.  With an odd feature count the last
    pair is half empty.  Row-sparse (placed == walked, bit for bit) and dense destination (== the row-sparse result)."""
    rng = np.random.default_rng(100 + n + 7 * fm)
    D, rows, B = 16, 5000, 1300
    ld = (16 * n + 31) // 32 * 32
    slots = [ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=int(fm)) for i in range(n)]
    plan = ops.EmbedPlan(slots, out_width=n * D, use_fm=fm)
    tables = [torch.from_numpy(rng.standard_normal((rows, D)).astype(np.float32)).to(DEV) for _ in range(n)]
    inputs = [torch.from_numpy(_ids(rng, rows, (B,), dist)).to(DEV) for _ in range(n)]
    g_out = torch.from_numpy(rng.standard_normal((B, n * D)).astype(np.float32)).to(DEV)
    g_fm = torch.from_numpy(rng.standard_normal((B,)).astype(np.float32)).to(DEV)

    def run(place, sparse, dense_sorted=None):
        monkeypatch.setattr(ops, "SPARSE_PLACE", place)
        monkeypatch.setattr(ops, "DENSE_BWD_SORTED", dense_sorted)
        ts = [t.clone().requires_grad_() for t in tables]
        out, _, fmv = ops.embed_apply(plan, ts, inputs, [None] * n, sparse_grad=sparse, out_ld=ld)
        assert out.shape[1] == ld and out.stride(0) == ld
        loss = (out[:, :n * D] * g_out).sum()
        if fmv is not None:
            loss = loss + (fmv * g_fm).sum()
        loss.backward()
        torch.cuda.synchronize()
        return [t.grad.coalesce() if sparse else t.grad for t in ts]

    a = run(True, True)
    b = run(False, True)
    _same(a, b)
    d = run(True, False, dense_sorted=True)                   # the dense destination of the same reduction
    for x, y in zip(a, d):
        assert torch.equal(x.to_dense().view(torch.int32), y.view(torch.int32))
