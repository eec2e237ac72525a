"""Full-size property tests on the REAL BASELINE.json table shapes: identity-valued tables whose
every element is a closed form of (row, column), so the gathered output can be checked bit for bit at any size without a
CPU copy of the table.

  C3  DCN: item_id table 100 000 000 x 64 fp32 (25.6 GB; element offsets far beyond 2^31), B = 65 536
  C4  DSSM: user_id table 10 000 000 x 16 + history L = 50 over the 200 000-row news table, B = 65 536
  C5  Wide&Deep: ALL 40 tables (1 k .. 500 M rows x 32, 224 GB) resident on one 288 GB GPU, wide split on the 10
      smallest, B = 65 536 -- skipped when the device cannot hold them

Reference arithmetic: BaseModel.get_feature_embedding / get_embeddings_from_batch (src/model/BaseModel/base_model.py:
262-271, 284-308), WideDeep.get_inp_embedding (src/model/sort/widedeep/model.py:53-69), DSSM tower inputs
(src/model/recall/DSSM/model.py:148-180).  Gather / concat / wide split: bit-exact; pooled history: rtol 1e-6."""
import pytest
import torch

from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_SPARSE

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B = 65536
SLAB = 1 << 24


def pattern(rows_idx: torch.Tensor, D: int, salt: int) -> torch.Tensor:
    """Closed-form table content: even columns (r & 8191) + k + salt, odd columns (r >> 13) + k -- exact in fp32
    (all values < 2^24) and (even, odd) together identify r.  rows_idx int64 [n] -> float32 [n, D]."""
    k = torch.arange(D, device=rows_idx.device)
    lo = (rows_idx & 8191)[:, None] + k[None] + salt
    hi = (rows_idx >> 13)[:, None] + k[None]
    return torch.where((k % 2 == 0)[None], lo, hi).to(torch.float32)


def make_table(rows: int, D: int, salt: int) -> torch.Tensor:
    t = torch.empty((rows, D), dtype=torch.float32, device=DEV)
    for r0 in range(0, rows, SLAB):                        # in slabs: the int64 temporaries stay ~1 GB
        r1 = min(rows, r0 + SLAB)
        t[r0:r1] = pattern(torch.arange(r0, r1, device=DEV), D, salt)
    t[0].zero_()                                           # padding row (nn.Embedding(padding_idx=0))
    return t


def expect(ids: torch.Tensor, D: int, salt: int) -> torch.Tensor:
    e = pattern(ids.reshape(-1).long(), D, salt)
    e[ids.reshape(-1) == 0] = 0
    return e


def need_free(bytes_):
    free = torch.cuda.mem_get_info(torch.device(DEV))[0]
    if free < bytes_:
        pytest.skip(f"needs {bytes_ >> 30} GiB of free HBM, device has {free >> 30} GiB")


def draw(gen, rows, shape=(B,)):
    ids = torch.randint(1, rows, shape, device=DEV, generator=gen)
    flat = ids.view(-1)
    flat[0], flat[1], flat[2] = rows - 1, 0, rows - 1      # last row (max offset), padding, duplicate
    return ids


def test_c3_real_100m_row_table_gather_and_fused_cross():
    need_free(40 << 30)
    gen = torch.Generator(device=DEV).manual_seed(20260301)
    rows = dict(category=18, item_id=100_000_000, subcategory=270, user_click_category=18, user_id=1_000_000)
    names = sorted(rows)
    D = 64
    tables = [make_table(rows[n], D, 3 * i) for i, n in enumerate(names)]
    assert tables[1].numel() > 2 ** 31                                     # element offsets do not fit 32 bits
    ids = [draw(gen, rows[n]) for n in names]
    ids[1][3] = 33_554_432                                                 # first row whose element offset is exactly 2^31
    ids[1][4] = 99_999_999
    plan = ops.EmbedPlan([ops.Slot(n, NRX_SPARSE, i, D, 0, i * D) for i, n in enumerate(names)], out_width=5 * D)
    want = torch.cat([expect(x, D, 3 * i) for i, x in enumerate(ids)], dim=1)
    for dt in (torch.int64, torch.int32):
        out = ops.embed_apply(plan, tables, [x.to(dt) for x in ids], [None] * 5)[0]
        assert torch.equal(out, want)
    # the C3 bench path: one fused launch gather -> cat[x, cross(x)]; x half bit-exact, cross half vs fp64 and vs the two-launch form on every row
    W = 5 * D
    w = torch.randn(2, W, device=DEV, generator=gen) / W ** 0.5 * 1e-3
    b = torch.randn(2, W, device=DEV, generator=gen)
    buf = ops.embed_dcn_v1(plan, tables, ids, w, b)
    assert torch.equal(buf[:, :W], want)
    x0 = want.double()                                                     # every row of the batch, in fp64
    xl = x0
    for l in range(2):
        xl = x0 * (xl @ w[l].double())[:, None] + b[l].double() + xl
    torch.testing.assert_close(buf[:, W:].double(), xl, rtol=1e-5, atol=1e-5 * xl.abs().max().item())
    del x0, xl
    two = ops.dcn_v1(want, w, b)                                           # the two-launch form of the same cross, every row
    torch.testing.assert_close(buf[:, W:], two, rtol=1e-6, atol=1e-6 * two.abs().max().item())


def test_c4_real_10m_user_table_and_history_pooling():
    need_free(4 << 30)
    gen = torch.Generator(device=DEV).manual_seed(20260302)
    D, L = 16, 50
    users, news = make_table(10_000_000, D, 0), make_table(200_000, D, 5)
    uid = draw(gen, 10_000_000)
    iid = draw(gen, 200_000)
    lens = torch.randint(0, L + 1, (B,), device=DEV, generator=gen)
    lens[0], lens[1] = L, 0
    mask = (torch.arange(L, device=DEV)[None] < lens[:, None]).float()
    hist = torch.randint(1, 200_000, (B, L), device=DEV, generator=gen) * mask.long()
    # user tower input: sorted names -> [user_history (pooled, news table), user_id]; item tower: [item_id]
    plan_u = ops.EmbedPlan([ops.Slot("user_history", NRX_BAG_MASKED_MEAN, 1, D, L, 0), ops.Slot("user_id", NRX_SPARSE, 0, D, 0, D)],
                           out_width=2 * D)
    out = ops.embed_apply(plan_u, [users, news], [hist, uid], [mask, None])[0]
    assert torch.equal(out[:, D:], expect(uid, D, 0))                                     # gather: bit-exact
    rows = expect(hist, D, 5).view(B, L, D).double()
    ref = (rows * mask.double()[..., None]).sum(1) / (mask.double().sum(1, keepdim=True) + 1e-8)
    assert torch.all(out[lens == 0, :D] == 0)                                             # all-masked bag: exact zeros
    rel = ((out[:, :D].double() - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
    assert rel < 1e-6                                                                      # stated pooling tolerance
    plan_i = ops.EmbedPlan([ops.Slot("item_id", NRX_SPARSE, 0, D, 0, 0)], out_width=D)
    assert torch.equal(ops.embed_apply(plan_i, [news], [iid], [None])[0], expect(iid, D, 5))


def test_c5_all_40_tables_on_one_gpu_with_wide_split():
    sizes = [int(round(1e3 * (5e5) ** (i / 39))) for i in range(40)]                      # 1 k .. 500 M rows
    D = 32
    need_free(sum(sizes) * D * 4 + (10 << 30))
    gen = torch.Generator(device=DEV).manual_seed(20260303)
    order = sorted(range(40), key=lambda i: -sizes[i])                                    # allocate the big ones first
    tables = [None] * 40
    for i in order:
        tables[i] = make_table(sizes[i], D, i)
    assert sum(t.numel() for t in tables) * 4 > 220e9
    ids = [draw(gen, sizes[i]) for i in range(40)]
    ids[39][5] = 499_999_999
    # wide features = the 10 smallest tables: column 0 -> wide tensor, columns 1.. -> deep concat (widedeep/model.py:53-69)
    slots, col = [], 0
    for i in range(40):
        wide = i < 10
        slots.append(ops.Slot(f"W{i:02d}", NRX_SPARSE, i, D, 0, col, wide_col=i if wide else -1))
        col += D - 1 if wide else D
    plan = ops.EmbedPlan(slots, out_width=col, wide_width=10)
    deep, wide, _ = ops.embed_apply(plan, tables, ids, [None] * 40)
    assert tuple(deep.shape) == (B, 40 * D - 10) and tuple(wide.shape) == (B, 10)
    c = 0
    for i in range(40):
        e = expect(ids[i], D, i)
        if i < 10:
            assert torch.equal(wide[:, i], e[:, 0]) and torch.equal(deep[:, c:c + D - 1], e[:, 1:])
            c += D - 1
        else:
            assert torch.equal(deep[:, c:c + D], e)
            c += D
    # plain concat (no split): the uniform ring kernel over all 40 tables
    plan2 = ops.EmbedPlan([ops.Slot(f"W{i:02d}", NRX_SPARSE, i, D, 0, i * D) for i in range(40)], out_width=40 * D)
    out = ops.embed_apply(plan2, tables, ids, [None] * 40)[0]
    for i in (0, 9, 10, 38, 39):
        assert torch.equal(out[:, i * D:(i + 1) * D], expect(ids[i], D, i))


@pytest.mark.parametrize("shape", ["c2", "c4", "c5"])
@pytest.mark.parametrize("skew", [False, True])
def test_backward_plan_at_baseline_shapes_matches_definition(shape, skew):
    """nrx_sparse_plan (the row-sparse backward's planning step: table-segmented stable sort of the lookups, unique rows,
    segment starts, per-table bounds) at the REAL lookup counts of the BASELINE shapes, B = 65 536, against its numpy
    definition (oracle.ref_np.sparse_plan) bit for bit.  C2: 26 one-chunk segments; C4: an 800-tile segment (history, which
    shares the news table with item_id) next to a 16-tile one; C5: 40 tables of 1 k .. 500 M rows -> 64-bit keys, three
    digit passes.  Reference: autograd of nn.Embedding over every lookup feature (src/model/BaseModel/base_model.py:262-308)."""
    import numpy as np
    from oracle import ref_np as R
    rng = np.random.default_rng({"c2": 2, "c4": 4, "c5": 5}[shape] + (100 if skew else 0))
    if shape == "c2":
        lens, rows, tab = [B] * 26, [1_000_000] * 26, list(range(26))
    elif shape == "c4":
        lens, rows, tab = [B, B * 50, B], [10_000_000, 200_000, 200_000], [0, 1, 1]
    else:
        rows = [int(round(1e3 * (5e5) ** (i / 39))) for i in range(40)]
        lens, tab = [B] * 40, list(range(40))
    ids = []
    for n, r in zip(lens, rows):
        x = np.minimum(rng.zipf(1.05, n) - 1, r - 1) if skew else rng.integers(0, r, n)
        ids.append(x.astype(np.int64))
    nt = max(tab) + 1
    order, uniq, seg, counts = ops.sparse_plan([torch.from_numpy(x).to(DEV) for x in ids], tab, rows, nt)
    o_r, u_r, s_r, c_r = R.sparse_plan(ids, tab, rows, nt)
    c = counts.cpu().numpy()
    assert np.array_equal(c, c_r)
    nu = int(c[0])
    assert np.array_equal(order.cpu().numpy(), o_r)
    assert np.array_equal(uniq.cpu().numpy()[:nu], u_r)
    assert np.array_equal(seg.cpu().numpy()[:nu + 1], s_r)


@pytest.mark.parametrize("skew", [False, True])
def test_c4_row_sparse_backward_equals_dense_backward_at_full_size(skew, monkeypatch):
    """The deterministic row-sparse backward (planning + sorted segmented reduction: what `embeddings.sparse_grad` and the
    bench's fwd_bwd leg run) against the dense-gradient backward in both its forms -- the default at this size (the same
    reduction + nrx_rows_to_dense: bit for bit the row-sparse result made dense) and the float-atomic scatter (NRX_DENSE_BWD=atomic:
    a different kernel and no planning at all) -- on the REAL C4 shape: user_id over the 10 M-row table, history L = 50 + item_id sharing the 200 k-row news table,
    B = 65 536 -- 3.4 M lookups, an 800-tile segment next to a 16-tile one.  Densified, the two gradients agree to the
    atomics' summation-order noise; two sparse runs are bit-identical.  (Autograd of nn.Embedding + array_feature_pooling:
    src/model/BaseModel/base_model.py:262-282.)"""
    need_free(6 << 30)
    gen = torch.Generator(device=DEV).manual_seed(20261002 + int(skew))
    D, L = 16, 50
    users = torch.randn((10_000_000, D), device=DEV, generator=gen)
    news = torch.randn((200_000, D), device=DEV, generator=gen)

    def ids_of(rows, shape):
        if not skew:
            return torch.randint(1, rows, shape, device=DEV, generator=gen)
        u = torch.rand(shape, device=DEV, generator=gen)              # popularity-skewed: rank ~ rows ** u
        return (torch.pow(float(rows), u).long() - 1).clamp_(1, rows - 1)

    uid, iid = ids_of(10_000_000, (B,)), ids_of(200_000, (B,))
    lens = torch.randint(0, L + 1, (B,), device=DEV, generator=gen)
    lens[0], lens[1] = L, 0
    mask = (torch.arange(L, device=DEV)[None] < lens[:, None]).float()
    hist = ids_of(200_000, (B, L)) * mask.long()
    plan = ops.EmbedPlan([ops.Slot("user_id", NRX_SPARSE, 0, D, 0, 0), ops.Slot("user_history", NRX_BAG_MASKED_MEAN, 1, D, L, D),
                          ops.Slot("item_id", NRX_SPARSE, 1, D, 0, 2 * D)], out_width=3 * D)
    up = torch.randn((B, 3 * D), device=DEV, generator=gen)
    grads = {}
    for mode in ("dense", "dense_auto", "sparse", "sparse2"):
        monkeypatch.setattr(ops, "DENSE_BWD_SORTED", False if mode == "dense" else None)      # "dense": the atomic scatter; "dense_auto": the default
        tabs = [users.clone().requires_grad_(True), news.clone().requires_grad_(True)]
        out = ops.embed_apply(plan, tabs, [uid, hist, iid], [None, mask, None], sparse_grad=mode.startswith("sparse"))[0]
        (out * up).sum().backward()
        grads[mode] = [t.grad for t in tabs]
        del out, tabs
    for a, b, da in zip(grads["sparse"], grads["sparse2"], grads["dense_auto"]):
        a, b = a.coalesce(), b.coalesce()
        assert torch.equal(a.indices(), b.indices()) and torch.equal(a.values(), b.values())           # bit-reproducible
        assert not da.is_sparse and torch.equal(da[a.indices()[0]], a.values())                        # default dense mode (3.4 M lookups: sorted) == row-sparse
        assert int((da != 0).any(1).sum()) <= a.indices().shape[1]                                     # ... and nothing outside its rows
    # float64 restatement of the definition (index_add of every lookup's contribution) + the L1 mass per row: fp32 sums of n
    # terms in ANY order stay within n * eps * sum|x|; hot rows of the skewed draw collect ~1e5 terms, so the bound is per row
    w = mask.double() / (mask.double().sum(1, keepdim=True) + 1e-8)
    upd = up.double()
    contrib = [(0, uid, upd[:, :D]), (1, hist.reshape(-1), (w[..., None] * upd[:, None, D:2 * D]).reshape(-1, D)), (1, iid, upd[:, 2 * D:])]
    for t, (d, sp) in enumerate(zip(grads["dense"], grads["sparse"])):
        ref = torch.zeros(d.shape, dtype=torch.float64, device=DEV)
        mass = torch.zeros(d.shape, dtype=torch.float64, device=DEV)
        for tt, ids, c in contrib:
            if tt == t:
                ref.index_add_(0, ids, c)
                mass.index_add_(0, ids, c.abs())
        ref[0] = 0                                                                                   # padding row never trains
        sp = sp.coalesce()
        rows = sp.indices()[0]
        assert torch.all(rows[1:] > rows[:-1])                                                       # every row once, ascending
        touched = torch.zeros(d.shape[0], dtype=torch.bool, device=DEV)
        touched[rows] = True
        assert torch.all(d[~touched] == 0) and torch.all(ref[~touched] == 0)                         # same support
        tol = 2e-6 * mass[rows] + 1e-6
        assert torch.all((sp.values().double() - ref[rows]).abs() <= tol)                            # sorted reduction
        assert torch.all((d[rows].double() - ref[rows]).abs() <= 4 * tol)                            # atomic scatter (order noise)
        del ref, mass


def test_c2_row_sparse_backward_with_fm_gradient_at_full_size():
    """C2 (DeepFM: 26 tables x 1 M rows x 16, B = 65 536, FM logit fused into the gather): the row-sparse backward with the FM
    gradient folded into the reduction, against float64 autograd of the reference arithmetic restated in torch
    (embedding rows -> concat; column 0 of every field is its first-order weight, columns 1.. its factors:
    fm = sum_f w + 0.5 * sum_k[(sum_f v)^2 - sum_f v^2], src/model/sort/fm/model.py:18-26, 48-59) -- every touched row of every table."""
    need_free(24 << 30)
    gen = torch.Generator(device=DEV).manual_seed(20261003)
    F, D, rows = 26, 16, 1_000_000
    tables = [torch.randn((rows, D), device=DEV, generator=gen) * 0.1 for _ in range(F)]
    for t in tables:
        t[0].zero_()
    ids = [torch.randint(1, rows, (B,), device=DEV, generator=gen) for _ in range(F)]
    up = torch.randn((B, F * D), device=DEV, generator=gen)
    upf = torch.randn((B,), device=DEV, generator=gen)
    plan = ops.EmbedPlan([ops.Slot(f"f{i:02d}", NRX_SPARSE, i, D, 0, i * D, fm_field=1) for i in range(F)], out_width=F * D, use_fm=True)
    tabs = [t.clone().requires_grad_(True) for t in tables]
    out, _, fm = ops.embed_apply(plan, tabs, ids, [None] * F, sparse_grad=True)
    ((out * up).sum() + (fm * upf).sum()).backward()
    # float64 autograd of the definition
    rows64 = [t[i].double().requires_grad_(True) for t, i in zip(tables, ids)]          # [B, D] per field
    v = torch.stack(rows64, 1)                                                             # [B, F, D]
    fm64 = v[:, :, 0].sum(1) + 0.5 * (v[:, :, 1:].sum(1).pow(2) - v[:, :, 1:].pow(2).sum(1)).sum(1)
    loss = (torch.cat(rows64, 1) * up.double()).sum() + (fm64 * upf.double()).sum()
    g64 = torch.autograd.grad(loss, rows64)
    assert torch.allclose(fm.double(), fm64, rtol=2e-5, atol=2e-5)
    for f in range(F):
        sp = tabs[f].grad.coalesce()
        r = sp.indices()[0]
        ref = torch.zeros((rows, D), dtype=torch.float64, device=DEV).index_add_(0, ids[f], g64[f])
        touched = torch.zeros(rows, dtype=torch.bool, device=DEV)
        touched[r] = True
        assert torch.all(ref[~touched] == 0)
        assert torch.allclose(sp.values().double(), ref[r], rtol=1e-5, atol=2e-5), f"table {f}"
        del ref
