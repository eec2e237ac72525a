"""The DEFAULT (dense-gradient) backward of the gather -- what autograd gives the reference's nn.Embedding(size, dim, padding_idx=0)
tables (src/model/BaseModel/base_model.py:164; backward of :262-308, + fm/model.py:18-26, widedeep/model.py:53-69) -- formed by the
sorted reduction + nrx_rows_to_dense (the default from ops.DENSE_SORTED_MIN lookups per launch on; forced here) against

  * the row-sparse mode's COO gradients made dense: the same reduction, so BIT FOR BIT;
  * itself, run twice: bit for bit (the mode is deterministic; float atomics are not);
  * the float-atomic scatter (nrx_embed_bwd, NRX_DENSE_BWD=atomic), which test_hip_parity ties to the oracle and the goldens:
    rtol 1e-5 / atol 1e-5 -- the two differ only in the order fp32 terms of a row are added (1e-4 for the Zipf cases, whose
    hottest row sums thousands of terms: the two fp32 orders drift apart by a few ulp of the partial sums)."""
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
    else:                                                    # zipf: rows looked up hundreds of times
        x = np.minimum(rng.zipf(1.2, shape) - 1, rows - 1)
    x = np.asarray(x, np.int64)
    x.reshape(-1)[:3] = 0                                    # the padding row is looked up too
    return x


def _grads(plan, tables, inputs, weights, ups, mode, monkeypatch):
    monkeypatch.setattr(ops, "DENSE_BWD_SORTED", mode != "atomic")
    ts = [t.clone().requires_grad_() for t in tables]
    res = ops.embed_apply(plan, ts, inputs, weights, sparse_grad=(mode == "coo"))
    loss = sum((r * u).sum() for r, u in zip(res, ups) if r is not None)
    loss.backward()
    torch.cuda.synchronize()
    return [t.grad.to_dense() if mode == "coo" else t.grad for t in ts]


def _check(plan, tables, inputs, weights, ups, monkeypatch, tol=1e-5):
    a = _grads(plan, tables, inputs, weights, ups, "sorted", monkeypatch)
    b = _grads(plan, tables, inputs, weights, ups, "sorted", monkeypatch)
    c = _grads(plan, tables, inputs, weights, ups, "coo", monkeypatch)
    d = _grads(plan, tables, inputs, weights, ups, "atomic", monkeypatch)
    for ga, gb, gc, gd in zip(a, b, c, d):
        assert not ga.is_sparse and ga.shape == gd.shape
        assert torch.equal(ga.view(torch.int32), gb.view(torch.int32))          # run to run: bit for bit
        assert torch.equal(ga, gc)                                              # == the row-sparse reduction
        torch.testing.assert_close(ga, gd, rtol=tol, atol=tol)                  # ~ the atomic scatter (addition order only)
        assert float(ga[0].abs().max()) == 0.0                                  # padding_idx = 0: the padding row gets no gradient


CASES = [
    # name, D, n_feats, rows per table, B, fm, wide feature indices, dist
    ("c2_like_fm", 16, 26, 3000, 1500, True, (), "uniform"),
    ("fm_zipf", 16, 9, 40000, 4000, True, (), "zipf"),
    ("plain32", 32, 6, 2000, 1300, False, (), "uniform"),
    ("plain64_zipf", 64, 5, 100000, 5000, False, (), "zipf"),
    ("odd_dim_10", 10, 4, 700, 900, False, (), "uniform"),
    ("wide16_zipf", 16, 6, 30000, 3000, False, (0, 5), "zipf"),
]


@pytest.mark.parametrize("name,D,n,rows,B,fm,wide,dist", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("idx", [torch.int64, torch.int32])
def test_dense_backward_sorted_single_valued(name, D, n, rows, B, fm, wide, dist, idx, monkeypatch):
    rng = np.random.default_rng(sum(map(ord, name)) + 3)
    slots, col = [], 0
    for i in range(n):
        if i in wide:
            slots.append(ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, col, wide_col=len([w for w in wide if w < i])))
            col += D - 1
        else:
            slots.append(ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, col, fm_field=1 if fm else 0))
            col += D
    plan = ops.EmbedPlan(slots, out_width=col, wide_width=len(wide), use_fm=fm)
    tables = [torch.from_numpy(rng.standard_normal((rows, D)).astype(np.float32)).to(DEV) for _ in range(n)]
    inputs = [torch.from_numpy(_ids(rng, rows, (B,), dist)).to(DEV).to(idx) for _ in range(n)]
    ups = (torch.from_numpy(rng.standard_normal((B, col)).astype(np.float32)).to(DEV),
           torch.from_numpy(rng.standard_normal((B, max(len(wide), 1))).astype(np.float32)).to(DEV),
           torch.from_numpy(rng.standard_normal((B,)).astype(np.float32)).to(DEV))
    _check(plan, tables, inputs, [None] * n, ups, monkeypatch, tol=1e-4 if dist == "zipf" else 1e-5)


@pytest.mark.parametrize("kind", [NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM])
@pytest.mark.parametrize("dist", ["uniform", "zipf"])
def test_dense_backward_sorted_tower_with_history_bag(kind, dist, monkeypatch):
    """The DSSM tower shape: item id + history bag (sharing the news table) + user id."""
    rng = np.random.default_rng(23 + kind)
    D, L, B, news, users = 16, 7, 2100, 6000, 50000
    slots = [ops.Slot("item_id", NRX_SPARSE, 0, D, 0, 0), ops.Slot("user_history", kind, 0, D, L, D), ops.Slot("user_id", NRX_SPARSE, 1, D, 0, 2 * D)]
    plan = ops.EmbedPlan(slots, out_width=3 * D)
    tables = [torch.from_numpy(rng.standard_normal((r, D)).astype(np.float32)).to(DEV) for r in (news, users)]
    hist = _ids(rng, news, (B, L), dist)
    lens = rng.integers(0, L + 1, B)
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.float32)
    hist = np.where(mask > 0, hist, 0)
    if kind == NRX_BAG_SUM:
        mask = mask * rng.random((B, L)).astype(np.float32)
    inputs = [torch.from_numpy(_ids(rng, news, (B,), dist)).to(DEV), torch.from_numpy(hist).to(DEV),
              torch.from_numpy(_ids(rng, users, (B,), "uniform")).to(DEV)]
    weights = [None, None if kind == NRX_BAG_MEAN else torch.from_numpy(mask).to(DEV), None]
    ups = (torch.from_numpy(rng.standard_normal((B, 3 * D)).astype(np.float32)).to(DEV), None, None)
    _check(plan, tables, inputs, weights, ups, monkeypatch, tol=1e-4 if dist == "zipf" else 1e-5)


def test_dense_backward_sorted_two_widths_and_a_table_fed_by_two_reductions(monkeypatch):
    """70 features of width 16 over 3 shared tables (two launch groups: the second ADDS into rows the first stored) + 4 features of
    width 32 over their own table (a third reduction, another row width in the same table list)."""
    rng = np.random.default_rng(5)
    B, rows = 900, 400
    slots, col = [], 0
    for i in range(70):
        slots.append(ops.Slot(f"a{i}", NRX_SPARSE, i % 3, 16, 0, col)); col += 16
    for i in range(4):
        slots.append(ops.Slot(f"b{i}", NRX_SPARSE, 3, 32, 0, col)); col += 32
    plan = ops.EmbedPlan(slots, out_width=col)
    tables = [torch.from_numpy(rng.standard_normal((rows, 16)).astype(np.float32)).to(DEV) for _ in range(3)]
    tables.append(torch.from_numpy(rng.standard_normal((rows, 32)).astype(np.float32)).to(DEV))
    inputs = [torch.from_numpy(_ids(rng, rows, (B,), "uniform")).to(DEV) for _ in slots]
    ups = (torch.from_numpy(rng.standard_normal((B, col)).astype(np.float32)).to(DEV), None, None)
    a = _grads(plan, tables, inputs, [None] * len(slots), ups, "sorted", monkeypatch)
    b = _grads(plan, tables, inputs, [None] * len(slots), ups, "sorted", monkeypatch)
    d = _grads(plan, tables, inputs, [None] * len(slots), ups, "atomic", monkeypatch)
    for ga, gb, gd in zip(a, b, d):
        assert torch.equal(ga.view(torch.int32), gb.view(torch.int32))
        torch.testing.assert_close(ga, gd, rtol=1e-5, atol=2e-5)


def test_rows_to_dense_c_abi_direct():
    """nrx_rows_to_dense through the C-ABI: store and accumulate forms, the device-side count, a key list longer than the count."""
    import ctypes as C
    from news_recsys_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(1)
    for D in (16, 10, 128):
        t0 = torch.zeros(50, D, device=DEV); t1 = torch.zeros(70, D, device=DEV)
        keys = torch.tensor([(0 << 40) | 3, (0 << 40) | 49, (1 << 40) | 0, (1 << 40) | 69, (1 << 40) | 5, (0 << 40) | 7], dtype=torch.int64, device=DEV)
        rows = torch.randn(6, D, device=DEV, generator=g)
        n_dev = torch.tensor([5], dtype=torch.int64, device=DEV)            # the 6th entry is beyond the count: not stored
        ptrs = (C.c_void_p * 2)(t0.data_ptr(), t1.data_ptr())
        st = torch.cuda.current_stream().cuda_stream
        for acc in (0, 1, 1):
            assert lib.nrx_rows_to_dense(ptrs, 2, D, keys.data_ptr(), rows.data_ptr(), 6, n_dev.data_ptr(), acc, st) == 0
        torch.cuda.synchronize()
        e0 = torch.zeros_like(t0); e1 = torch.zeros_like(t1)
        e0[3], e0[49], e1[0], e1[69], e1[5] = rows[0] * 3, rows[1] * 3, rows[2] * 3, rows[3] * 3, rows[4] * 3
        torch.testing.assert_close(t0, e0, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(t1, e1, rtol=1e-6, atol=1e-6)
        assert float(t0[7].abs().max()) == 0.0


def test_row0_is_data_keeps_the_atomic_scatter_and_refuses_the_row_sparse_mode():
    """A routed-row buffer (sharding step 4: NRX_FEAT_ROW0_IS_DATA) has no padding row: its slot 0 must receive its gradient.  The sorted
    reduction gives row 0 of every table a zero gradient, so the default mode keeps nrx_embed_bwd for such a plan and sparse_grad refuses it."""
    from news_recsys_amd._lib import NRX_FEAT_ROW0_IS_DATA
    g = torch.Generator(device=DEV).manual_seed(3)
    buf = torch.randn(64, 16, device=DEV, generator=g).requires_grad_()
    slot = torch.randint(0, 64, (500,), device=DEV, generator=g)
    slot[:5] = 0
    plan = ops.EmbedPlan([ops.Slot("f", NRX_SPARSE, 0, 16, 0, 0, flags=NRX_FEAT_ROW0_IS_DATA)], out_width=16)
    up = torch.randn(500, 16, device=DEV, generator=g)
    out = ops.embed_apply(plan, [buf], [slot], [None])[0]
    (out * up).sum().backward()
    ref = torch.zeros(64, 16, device=DEV).index_add_(0, slot, up)
    torch.testing.assert_close(buf.grad, ref, rtol=1e-5, atol=1e-5)
    assert float(buf.grad[0].abs().max()) > 0
    with pytest.raises(NotImplementedError):
        ops.embed_apply(plan, [buf], [slot], [None], sparse_grad=True)


@pytest.mark.parametrize("kind", [NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM])
def test_dense_backward_sorted_csr_bags(kind, monkeypatch):
    """A bag delivered as CSR values + offsets (ColumnarLoader(csr_bags=True)): the forward runs on the CSR form, the backward's planner on its
    padded expansion (nrx_csr_to_padded at forward time) -- the same dense gradients, bit for bit, as the padded launch; bags longer than L are
    cut to their first L entries in both."""
    from news_recsys_amd._lib import NRX_FEAT_BAG_CSR
    rng = np.random.default_rng(41 + kind)
    D, L, B, news, users = 16, 6, 1900, 5000, 30000
    lens = rng.integers(0, L + 3, B)                                     # some bags longer than L
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    values = rng.integers(1, news, int(offsets[-1])).astype(np.int64)
    hist = np.zeros((B, L), np.int64); mask = np.zeros((B, L), np.float32)
    for b in range(B):
        n = min(int(lens[b]), L)
        hist[b, :n] = values[offsets[b]:offsets[b] + n]
        mask[b, :n] = 1.0
    tables = [torch.from_numpy(rng.standard_normal((r, D)).astype(np.float32)).to(DEV) for r in (news, users)]
    iid = torch.from_numpy(_ids(rng, news, (B,), "uniform")).to(DEV); uid = torch.from_numpy(_ids(rng, users, (B,), "uniform")).to(DEV)
    ups = (torch.from_numpy(rng.standard_normal((B, 3 * D)).astype(np.float32)).to(DEV), None, None)
    def plan_of(flags):
        return ops.EmbedPlan([ops.Slot("item_id", NRX_SPARSE, 0, D, 0, 0), ops.Slot("user_history", kind, 0, D, L, D, flags=flags),
                              ops.Slot("user_id", NRX_SPARSE, 1, D, 0, 2 * D)], out_width=3 * D)
    w_pad = None if kind == NRX_BAG_MEAN else torch.from_numpy(mask).to(DEV)
    pad = _grads(plan_of(0), tables, [iid, torch.from_numpy(hist).to(DEV), uid], [None, w_pad, None], ups, "sorted", monkeypatch)
    csr_in = [iid, torch.from_numpy(values).to(DEV), uid]
    csr_w = [None, torch.from_numpy(offsets).to(DEV), None]
    csr = _grads(plan_of(NRX_FEAT_BAG_CSR), tables, csr_in, csr_w, ups, "sorted", monkeypatch)
    atom = _grads(plan_of(NRX_FEAT_BAG_CSR), tables, csr_in, csr_w, ups, "atomic", monkeypatch)
    for gp, gc, ga in zip(pad, csr, atom):
        assert torch.equal(gp.view(torch.int32), gc.view(torch.int32))
        torch.testing.assert_close(gc, ga, rtol=1e-5, atol=1e-5)


def test_auto_mode_switches_on_the_lookup_count(monkeypatch):
    """NRX_DENSE_BWD=auto (the default): the sorted path from DENSE_SORTED_MIN lookups per launch on (bag positions count), the single atomic launch
    below; forced modes ignore the count; a plan reading a routed-row buffer never qualifies."""
    from news_recsys_amd._lib import NRX_DENSE, NRX_FEAT_ROW0_IS_DATA
    t = [torch.zeros(10, 16, device=DEV)]
    plan = ops.EmbedPlan([ops.Slot("a", NRX_SPARSE, 0, 16, 0, 0), ops.Slot("h", NRX_BAG_MEAN, 0, 16, 7, 16), ops.Slot("d", NRX_DENSE, -1, 1, 0, 32)], out_width=33)
    monkeypatch.setattr(ops, "DENSE_BWD_SORTED", None)
    monkeypatch.setattr(ops, "DENSE_SORTED_MIN", 800)
    assert ops._dense_sorted_ok(plan, t, False, 100) and not ops._dense_sorted_ok(plan, t, False, 99)      # 8 lookups per sample
    assert not ops._dense_sorted_ok(plan, t, True, 100)                                                   # row-sparse mode: not this switch
    monkeypatch.setattr(ops, "DENSE_BWD_SORTED", True)
    assert ops._dense_sorted_ok(plan, t, False, 1)
    monkeypatch.setattr(ops, "DENSE_BWD_SORTED", False)
    assert not ops._dense_sorted_ok(plan, t, False, 1 << 20)
    monkeypatch.setattr(ops, "DENSE_BWD_SORTED", True)
    routed = ops.EmbedPlan([ops.Slot("a", NRX_SPARSE, 0, 16, 0, 0, flags=NRX_FEAT_ROW0_IS_DATA)], out_width=16)
    assert not ops._dense_sorted_ok(routed, t, False, 1 << 20)


@pytest.mark.parametrize("place", [1, 0])
@pytest.mark.parametrize("B,dist", [(700, "uniform"), (3000, "zipf"), (1, "uniform")])
def test_dense_sorted_backward_one_call_c_abi(B, dist, place):
    """nrx_embed_bwd_dense_sorted through the C-ABI: plan + reduction into zero-filled dense gradient tables in ONE call with ONE workspace, against
    a float64 index_add restatement of nn.Embedding's backward (src/model/BaseModel/base_model.py:262-308; the padding row gets no gradient),
    tolerance 1e-5 * sum|contribution| (fp32 sums in a fixed order), and against itself run twice: bit for bit."""
    import ctypes as C
    from news_recsys_amd import _lib
    from news_recsys_amd._lib import NrxFeature
    lib = _lib.load()
    rng = np.random.default_rng(B + place)
    D, L = 16, 5
    rows = [900, 4000]
    ids = [_ids(rng, rows[0], (B,), dist), _ids(rng, rows[1], (B, L), dist), _ids(rng, rows[0], (B,), dist)]      # table 0 fed by two features
    mask = (rng.random((B, L)) < 0.7).astype(np.float32)
    ids[1] = np.where(mask > 0, np.maximum(ids[1], 1), 0)
    table_of = [0, 1, 0]
    kinds = [NRX_SPARSE, NRX_BAG_MASKED_MEAN, NRX_SPARSE]
    cols = [0, D, 2 * D]
    g_out = rng.standard_normal((B, 3 * D)).astype(np.float32)
    d_ids = [torch.from_numpy(x).to(DEV) for x in ids]
    d_mask = torch.from_numpy(mask).to(DEV)
    d_g = torch.from_numpy(g_out).to(DEV)
    results = []
    for _ in range(2):
        grads = [torch.zeros(r, D, device=DEV) for r in rows]
        feats = (NrxFeature * 3)()
        for i in range(3):
            f = feats[i]
            f.table, f.index, f.rows, f.dim, f.kind, f.index_bits = grads[table_of[i]].data_ptr(), d_ids[i].data_ptr(), rows[table_of[i]], D, kinds[i], 64
            f.bag_len, f.out_col, f.wide_col, f.fm_field, f.flags = (L if kinds[i] != NRX_SPARSE else 0), cols[i], -1, 0, 0
            f.weight = d_mask.data_ptr() if kinds[i] == NRX_BAG_MASKED_MEAN else None
        tof = (C.c_int32 * 3)(*table_of)
        gp = (C.c_void_p * 2)(*[g.data_ptr() for g in grads])
        nbytes = lib.nrx_embed_bwd_dense_sorted_workspace(feats, 3, B, D, 2)
        assert nbytes > 0
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        rc = lib.nrx_embed_bwd_dense_sorted(feats, tof, 3, 2, B, D, d_g.data_ptr(), 3 * D, None, 0, None, gp, 0, place, ws.data_ptr(), nbytes,
                                            torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.nrx_last_error()
        torch.cuda.synchronize()
        results.append([g.cpu().numpy() for g in grads])
    for a_, b_ in zip(*results):
        assert np.array_equal(a_.view(np.int32), b_.view(np.int32))
    want = [np.zeros((r, D), np.float64) for r in rows]
    mag = [np.zeros((r, D), np.float64) for r in rows]
    for i in (0, 2):
        np.add.at(want[0], ids[i], g_out[:, cols[i]:cols[i] + D].astype(np.float64))
        np.add.at(mag[0], ids[i], np.abs(g_out[:, cols[i]:cols[i] + D]).astype(np.float64))
    den = mask.sum(1, keepdims=True).astype(np.float32) + np.float32(1e-8)
    sc = (mask / den).astype(np.float64)                                   # d pooled / d row: w / (sum w + 1e-8)  (base_model.py:278-282)
    for l in range(L):
        contrib = g_out[:, D:2 * D].astype(np.float64) * sc[:, l:l + 1]
        np.add.at(want[1], ids[1][:, l], contrib)
        np.add.at(mag[1], ids[1][:, l], np.abs(contrib))
    for t in range(2):
        want[t][0] = 0
        got = results[0][t]
        assert np.all(got[0] == 0)
        assert np.all(np.abs(got - want[t]) <= 1e-5 * mag[t] + 1e-7)
    # an undersized workspace is refused
    assert lib.nrx_embed_bwd_dense_sorted(feats, tof, 3, 2, B, D, d_g.data_ptr(), 3 * D, None, 0, None, gp, 0, place, ws.data_ptr(), 64,
                                          torch.cuda.current_stream().cuda_stream) == -1
