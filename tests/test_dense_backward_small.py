"""The one-launch DETERMINISTIC dense backward for the reference's own batch sizes (nrx_embed_bwd_small: block per table, LDS sort,
in-order sums) -- the default for launches whose tables are each fed by <= 4096 lookups.  What autograd gives the reference's
nn.Embedding(size, dim, padding_idx=0) tables (src/model/BaseModel/base_model.py:164; backward of :262-308, + fm/model.py:18-26,
widedeep/model.py:53-69), checked against

  * itself, run twice: BIT FOR BIT (the point of the kernel; the float-atomic scatter it replaces is not);
  * the float-atomic scatter (nrx_embed_bwd, NRX_DENSE_BWD=atomic), which test_hip_parity ties to the oracle and the goldens:
    rtol / atol 1e-5 (addition order only; 1e-4 where a row sums hundreds of terms);
  * a float64 restatement of the gradient for the single-valued cases."""
import ctypes as C

import numpy as np
import pytest
import torch

from news_recsys_amd import _lib, ops
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM, NRX_ERR_UNSUPPORTED, NRX_FEAT_ROW0_IS_DATA, NRX_SPARSE

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ids(rng, rows, shape, dist):
    if dist == "uniform":
        x = rng.integers(0, rows, shape)
    elif dist == "hot":                                      # a handful of rows: every one summed by a whole wavefront
        x = rng.integers(0, min(rows, 6), shape)
    else:                                                    # zipf
        x = np.minimum(rng.zipf(1.2, shape) - 1, rows - 1)
    x = np.asarray(x, np.int64)
    x.reshape(-1)[:3] = 0                                    # the padding row is looked up too
    return x


class _Counting:
    """Wraps the library so a test can see which backward entry a step took."""

    def __init__(self, lib):
        self._lib, self.small, self.small_ok, self.atomic = lib, 0, 0, 0

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name == "nrx_embed_bwd_small":
            def wrapped(*a):
                rc = fn(*a)
                self.small += 1
                self.small_ok += rc == 0
                return rc
            return wrapped
        if name == "nrx_embed_bwd":
            def wrapped(*a):
                self.atomic += 1
                return fn(*a)
            return wrapped
        return fn


def _grads(plan, tables, inputs, weights, ups, mode, monkeypatch):
    monkeypatch.setattr(ops, "DENSE_BWD_SORTED", {"auto": None, "atomic": False, "det": "det"}[mode])
    cnt = _Counting(_lib.load())
    monkeypatch.setattr(_lib, "load", lambda: cnt)
    ts = [t.clone().requires_grad_() for t in tables]
    res = ops.embed_apply(plan, ts, inputs, weights)
    loss = sum((r * u).sum() for r, u in zip(res, ups) if r is not None)
    loss.backward()
    torch.cuda.synchronize()
    monkeypatch.undo()
    return [t.grad for t in ts], cnt


def _check(plan, tables, inputs, weights, ups, monkeypatch, tol=1e-5, expect_small=True):
    a, ca = _grads(plan, tables, inputs, weights, ups, "auto", monkeypatch)
    b, _ = _grads(plan, tables, inputs, weights, ups, "auto", monkeypatch)
    d, cd = _grads(plan, tables, inputs, weights, ups, "atomic", monkeypatch)
    assert cd.small == 0 and cd.atomic >= 1
    if expect_small:
        assert ca.small_ok >= 1 and ca.atomic == 0, "the step did not take the deterministic small kernel"
    else:
        assert ca.small_ok == 0 and ca.atomic >= 1
    for ga, gb, gd in zip(a, b, d):
        if expect_small:
            assert torch.equal(ga.view(torch.int32), gb.view(torch.int32))      # run to run: bit for bit
        torch.testing.assert_close(ga, gd, rtol=tol, atol=tol)                  # ~ the atomic scatter (addition order only)
        assert float(ga[0].abs().max()) == 0.0                                  # padding_idx = 0: the padding row gets no gradient
    return a


CASES = [
    # name, D, n_feats, rows per table, B, fm, wide feature indices, dist
    ("deep_b512", 16, 5, 3000, 512, False, (), "uniform"),
    ("c2_like_fm_b512", 16, 26, 3000, 512, True, (), "uniform"),
    ("c2_like_fm_b4096", 16, 26, 100000, 4096, True, (), "uniform"),
    ("fm_zipf", 16, 9, 40000, 2000, True, (), "zipf"),
    ("fm_hot", 16, 3, 40, 3000, True, (), "hot"),
    ("plain32", 32, 6, 2000, 1300, False, (), "uniform"),
    ("plain64_zipf", 64, 5, 100000, 1000, False, (), "zipf"),
    ("dim4_hot", 4, 3, 50, 700, False, (), "hot"),
    ("dim8", 8, 4, 500, 257, False, (), "uniform"),
    ("dim128_hot", 128, 2, 64, 300, False, (), "hot"),
    ("dim256", 256, 2, 900, 100, False, (), "zipf"),
    ("wide16_zipf", 16, 6, 30000, 3000, False, (0, 5), "zipf"),
    ("wide16_hot", 16, 4, 20, 1000, False, (1,), "hot"),
    ("one_sample", 16, 3, 10, 1, True, (), "uniform"),
    # widths that are not 4 * 2^k (the reference's wide features are 16 + 1 columns: cf_widedeep_small.yaml) and rows that are not
    # 16-byte aligned: the element-by-element form of the kernel
    ("odd_dim_10", 10, 4, 700, 900, False, (), "uniform"),
    ("wide17_zipf", 17, 5, 300, 1000, False, (1, 2, 4), "zipf"),
    ("wide17_hot", 17, 3, 30, 2000, False, (0,), "hot"),
    ("dim_1", 1, 3, 50, 400, False, (), "uniform"),
    ("dim_200", 200, 2, 40, 150, False, (), "hot"),
    ("lr_dim_1_fm", 1, 6, 60, 500, True, (), "uniform"),          # LR: dim-1 tables, the row sum is the FM epilogue's first-order term
    ("fm_dim_12", 12, 4, 300, 600, True, (), "zipf"),
]


@pytest.mark.parametrize("name,D,n,rows,B,fm,wide,dist", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("idx", [torch.int64, torch.int32])
def test_small_deterministic_backward_single_valued(name, D, n, rows, B, fm, wide, dist, idx, monkeypatch):
    rng = np.random.default_rng(sum(map(ord, name)) + 11)
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
    ids = [_ids(rng, rows, (B,), dist) for _ in range(n)]
    inputs = [torch.from_numpy(x).to(DEV).to(idx) for x in ids]
    up = rng.standard_normal((B, col)).astype(np.float32)
    upw = rng.standard_normal((B, max(len(wide), 1))).astype(np.float32)
    upf = rng.standard_normal((B,)).astype(np.float32)
    ups = (torch.from_numpy(up).to(DEV), torch.from_numpy(upw).to(DEV), torch.from_numpy(upf).to(DEV))
    loose = dist != "uniform"
    got = _check(plan, tables, inputs, [None] * n, ups, monkeypatch, tol=1e-4 if loose else 1e-5)
    if not wide:
        # float64 restatement: d loss / d row = sum over the lookups of the row of (upstream block + g_fm * d fm / d field)
        tabs = [t.double().cpu().numpy() for t in tables]
        emb = np.stack([tabs[i][ids[i]] for i in range(n)], 1)                  # [B, n, D]
        g = up.astype(np.float64).reshape(B, n, D).copy()
        if fm:
            S = emb.sum(1)                                                      # [B, D]
            dfm = S[:, None, :] - emb
            dfm[:, :, 0] = 1.0                                                  # column 0: the first-order term
            g += upf.astype(np.float64)[:, None, None] * dfm
        for i in range(n):
            want = np.zeros((rows, D))
            np.add.at(want, ids[i], g[:, i])
            want[0] = 0.0
            np.testing.assert_allclose(got[i].cpu().numpy(), want, rtol=2e-4 if loose else 2e-5, atol=2e-4 if loose else 2e-5)


@pytest.mark.parametrize("kind", [NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM])
@pytest.mark.parametrize("dist", ["uniform", "zipf"])
@pytest.mark.parametrize("B,L", [(500, 7), (60, 50)])
def test_small_deterministic_backward_tower_with_history_bag(kind, dist, B, L, monkeypatch):
    """The DSSM tower shape: item id + history bag (sharing the news table) + user id."""
    rng = np.random.default_rng(29 + kind + B)
    D, news, users = 16, 6000, 50000
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


def test_small_backward_two_launch_groups_add_into_shared_tables(monkeypatch):
    """70 features over 3 shared tables: two launches of <= 64 features, the second ADDS into rows the first stored."""
    rng = np.random.default_rng(6)
    B, rows = 50, 400
    slots, col = [], 0
    for i in range(70):
        slots.append(ops.Slot(f"a{i}", NRX_SPARSE, i % 3, 16, 0, col)); col += 16
    plan = ops.EmbedPlan(slots, out_width=col)
    tables = [torch.from_numpy(rng.standard_normal((rows, 16)).astype(np.float32)).to(DEV) for _ in range(3)]
    inputs = [torch.from_numpy(_ids(rng, rows, (B,), "uniform")).to(DEV) for _ in slots]
    ups = (torch.from_numpy(rng.standard_normal((B, col)).astype(np.float32)).to(DEV), None, None)
    _check(plan, tables, inputs, [None] * len(slots), ups, monkeypatch, tol=2e-5)


def test_launches_outside_the_small_shapes_keep_the_atomic_scatter(monkeypatch):
    """A table fed by more than 4096 lookups of the launch (here 3 features x 2000 samples): NRX_ERR_UNSUPPORTED, nothing enqueued,
    the step takes nrx_embed_bwd as before; a row wider than 256 columns likewise."""
    rng = np.random.default_rng(8)
    for D, B, share in ((16, 2000, True), (264, 300, False)):
        n, rows = 3, 900
        slots = [ops.Slot(f"f{i}", NRX_SPARSE, 0 if share else i, D, 0, i * D) for i in range(n)]
        plan = ops.EmbedPlan(slots, out_width=n * D)
        tables = [torch.from_numpy(rng.standard_normal((rows, D)).astype(np.float32)).to(DEV) for _ in range(1 if share else n)]
        inputs = [torch.from_numpy(_ids(rng, rows, (B,), "uniform")).to(DEV) for _ in range(n)]
        ups = (torch.from_numpy(rng.standard_normal((B, n * D)).astype(np.float32)).to(DEV), None, None)
        _check(plan, tables, inputs, [None] * n, ups, monkeypatch, tol=2e-5, expect_small=False)


def test_small_backward_c_abi_direct_row0_data_and_out_of_range_ids():
    """Straight through the C ABI: NRX_FEAT_ROW0_IS_DATA keeps row 0's gradient, ids outside the table are dropped (the forward
    reported them), accumulate = 1 ADDS into the gradient it is given (0 stores), an unsupported shape returns NRX_ERR_UNSUPPORTED untouched."""
    lib = _lib.load()
    rng = np.random.default_rng(12)
    B, D, rows = 300, 16, 50
    ids = rng.integers(0, rows, (2, B)).astype(np.int64)
    ids[0, :5] = 0
    ids[1, 7] = rows + 3                      # out of range
    ids[1, 8] = -1
    up = rng.standard_normal((B, 2 * D)).astype(np.float32)
    t_ids = [torch.from_numpy(x).to(DEV) for x in ids]
    g_out = torch.from_numpy(up).to(DEV)
    grads = [torch.ones((rows, D), device=DEV), torch.zeros((rows, D), device=DEV)]
    arr = (_lib.NrxFeature * 2)()
    for i in range(2):
        f = arr[i]
        f.table, f.index, f.weight, f.rows, f.dim = grads[i].data_ptr(), t_ids[i].data_ptr(), None, rows, D
        f.bag_len, f.kind, f.index_bits, f.out_col, f.wide_col, f.fm_field = 0, NRX_SPARSE, 64, i * D, -1, 0
        f.flags = NRX_FEAT_ROW0_IS_DATA if i == 0 else 0
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.nrx_embed_bwd_small(arr, 2, B, g_out.data_ptr(), 2 * D, None, 0, None, 1, stream) == 0
    torch.cuda.synchronize()
    for i in range(2):
        want = np.zeros((rows, D))
        ok = (ids[i] >= 0) & (ids[i] < rows) & ((ids[i] != 0) | (i == 0))
        np.add.at(want, ids[i][ok], up[ok, i * D:(i + 1) * D].astype(np.float64))
        want += 1.0 if i == 0 else 0.0
        np.testing.assert_allclose(grads[i].cpu().numpy(), want, rtol=2e-5, atol=2e-5)
    assert float(grads[0][0].abs().max()) > 1.0 and float(grads[1][0].abs().max()) == 0.0
    first = grads[1].clone()
    grads[1].fill_(7.0)                       # accumulate = 0: touched rows are stored over whatever was there, the rest is left alone
    arr1 = (_lib.NrxFeature * 1)()
    C.memmove(C.addressof(arr1[0]), C.addressof(arr[1]), C.sizeof(_lib.NrxFeature))
    assert lib.nrx_embed_bwd_small(arr1, 1, B, g_out.data_ptr(), 2 * D, None, 0, None, 0, stream) == 0
    torch.cuda.synchronize()
    touched = torch.zeros(rows, dtype=torch.bool, device=DEV)
    ok1 = (ids[1] > 0) & (ids[1] < rows)
    touched[torch.from_numpy(ids[1][ok1]).to(DEV)] = True
    assert torch.equal(grads[1][touched], first[touched]) and bool((grads[1][~touched] == 7.0).all())
    before = [g.clone() for g in grads]
    arr[0].dim = arr[1].dim = 260             # wider than 256 columns
    assert lib.nrx_embed_bwd_small(arr, 2, B, g_out.data_ptr(), 2 * D, None, 0, None, 1, stream) == NRX_ERR_UNSUPPORTED
    arr[0].dim = arr[1].dim = D
    assert lib.nrx_embed_bwd_small(arr, 2, 5000, g_out.data_ptr(), 2 * D, None, 0, None, 1, stream) == NRX_ERR_UNSUPPORTED
    torch.cuda.synchronize()
    for g, b in zip(grads, before):
        assert torch.equal(g, b)


def test_deterministic_mode_takes_the_small_kernel_where_it_applies_and_the_planned_reduction_elsewhere(monkeypatch):
    """NRX_DENSE_BWD=deterministic (what GraphedStep(deterministic=True) captures): bit-reproducible gradients by the cheaper kernel --
    nrx_embed_bwd_small for launches inside its shapes, the planned sorted reduction for the others; never the atomics."""
    rng = np.random.default_rng(14)
    for B, share, expect_small in ((700, False, True), (2000, True, False)):
        n, rows, D = 3, 900, 16
        slots = [ops.Slot(f"f{i}", NRX_SPARSE, 0 if share else i, D, 0, i * D) for i in range(n)]
        plan = ops.EmbedPlan(slots, out_width=n * D)
        tables = [torch.from_numpy(rng.standard_normal((rows, D)).astype(np.float32)).to(DEV) for _ in range(1 if share else n)]
        inputs = [torch.from_numpy(_ids(rng, rows, (B,), "zipf")).to(DEV) for _ in range(n)]
        ups = (torch.from_numpy(rng.standard_normal((B, n * D)).astype(np.float32)).to(DEV), None, None)
        a, ca = _grads(plan, tables, inputs, [None] * n, ups, "det", monkeypatch)
        b, _ = _grads(plan, tables, inputs, [None] * n, ups, "det", monkeypatch)
        d, _ = _grads(plan, tables, inputs, [None] * n, ups, "atomic", monkeypatch)
        assert ca.atomic == 0 and (ca.small_ok >= 1) == expect_small
        for ga, gb, gd in zip(a, b, d):
            assert torch.equal(ga.view(torch.int32), gb.view(torch.int32))
            torch.testing.assert_close(ga, gd, rtol=1e-4, atol=1e-4)


def test_reference_models_take_the_small_kernel_and_match_the_reference_gradients(monkeypatch):
    """The reference's own models on the reference's golden batches (tests/golden/model_*.npz: state_dict, batch and the gradients its
    autograd produced on the CPU -- tests/golden/gen_golden.py): in the default mode every embedding gradient of Deep, FM, DCN,
    Wide&Deep, LR and the array-feature Deep is formed by nrx_embed_bwd_small -- no float-atomic launch -- and matches the reference's."""
    from tests.test_models_gpu import CASES as MODEL_CASES, batch_of, gold, load_model
    monkeypatch.setattr(ops, "DENSE_BWD_SORTED", None)
    for cls, cfg_name, gname in MODEL_CASES:
        g = gold(gname)
        m = load_model(cls, cfg_name, g)
        batch = batch_of(g)
        cnt = _Counting(_lib.load())
        monkeypatch.setattr(_lib, "load", lambda: cnt)
        loss = m.bceLoss(m(batch), batch["label"][:, 0])
        loss.backward()
        torch.cuda.synchronize()
        monkeypatch.setattr(_lib, "load", cnt._lib.__class__ and (lambda lib=cnt._lib: lib))
        assert cnt.small_ok >= 1 and cnt.atomic == 0, (gname, cnt.small, cnt.small_ok, cnt.atomic)
        for name, emb in m.embedding_tables.items():
            want = g[f"grad/embedding_tables.{name}.weight"]
            np.testing.assert_allclose(emb.weight.grad.cpu().numpy(), want, rtol=2e-3, atol=2e-6 + 1e-4 * np.abs(want).max(), err_msg=f"{gname}:{name}")
            assert torch.all(emb.weight.grad[0] == 0), name


def _sink_dense(plan, tables, inputs, weights, ups, monkeypatch):
    """The same step through the row-sparse sink (what FusedSparseAdam drains), its (key, row) pairs scattered into dense tensors."""
    monkeypatch.setattr(ops, "DENSE_BWD_SORTED", None)
    ts = [t.clone().requires_grad_() for t in tables]
    sink = ops.SparseGradSink()
    res = ops.embed_apply(plan, ts, inputs, weights, sparse_grad=sink)
    loss = sum((r * u).sum() for r, u in zip(res, ups) if r is not None)
    loss.backward()
    torch.cuda.synchronize()
    out = [torch.zeros_like(t) for t in tables]
    seen = [torch.zeros(t.shape[0], dtype=torch.int32, device=DEV) for t in tables]
    lookups = 0
    for e in sink.pending:
        assert e.get("filler") is True, "the step did not take the one-launch row-sparse form"
        k, v = e["uniq"], e["values"]
        ok = k >= 0
        lookups += k.numel()
        t_of, row = (k[ok] >> 40), (k[ok] & ((1 << 40) - 1))
        for t in range(len(tables)):
            m = t_of == t
            if bool(m.any()):
                assert tables[t].shape[1] == e["dim"]
                out[t][row[m]] = v[ok][m]
                seen[t].index_add_(0, row[m], torch.ones_like(row[m], dtype=torch.int32))
    for sn in seen:
        assert int(sn.max()) <= 1                                               # every row appears once
    return out, lookups


SINK_CASES = ["deep_b512", "c2_like_fm_b512", "c2_like_fm_b4096", "fm_zipf", "fm_hot", "plain64_zipf", "dim128_hot", "wide16_zipf",
              "odd_dim_10", "wide17_zipf", "lr_dim_1_fm", "one_sample"]


@pytest.mark.parametrize("name", SINK_CASES)
def test_small_row_sparse_form_leaves_the_same_rows_as_the_dense_form(name, monkeypatch):
    """nrx_embed_bwd_small_sparse (the sink of the fused row-sparse optimizer): the (key, row) pairs, scattered, ARE the dense gradient of
    nrx_embed_bwd_small -- bit for bit (same kernel, another destination) -- every row once, fillers keyed -1."""
    _, D, n, rows, B, fm, wide, dist = [c for c in CASES if c[0] == name][0]
    rng = np.random.default_rng(sum(map(ord, name)) + 11)
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
    inputs = [torch.from_numpy(_ids(rng, rows, (B,), dist)).to(DEV) for _ in range(n)]
    ups = (torch.from_numpy(rng.standard_normal((B, col)).astype(np.float32)).to(DEV),
           torch.from_numpy(rng.standard_normal((B, max(len(wide), 1))).astype(np.float32)).to(DEV),
           torch.from_numpy(rng.standard_normal((B,)).astype(np.float32)).to(DEV))
    dense, cnt = _grads(plan, tables, inputs, [None] * n, ups, "auto", monkeypatch)
    assert cnt.small_ok >= 1
    got, lookups = _sink_dense(plan, tables, inputs, [None] * n, ups, monkeypatch)
    assert lookups == n * B
    for a, b in zip(dense, got):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize("kind", [NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM])
def test_small_row_sparse_form_tower_with_history_bag(kind, monkeypatch):
    rng = np.random.default_rng(41 + kind)
    D, L, B, news, users = 16, 7, 500, 6000, 50000
    slots = [ops.Slot("item_id", NRX_SPARSE, 0, D, 0, 0), ops.Slot("user_history", kind, 0, D, L, D), ops.Slot("user_id", NRX_SPARSE, 1, D, 0, 2 * D)]
    plan = ops.EmbedPlan(slots, out_width=3 * D)
    tables = [torch.from_numpy(rng.standard_normal((r, D)).astype(np.float32)).to(DEV) for r in (news, users)]
    hist = _ids(rng, news, (B, L), "zipf")
    lens = rng.integers(0, L + 1, B)
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.float32)
    hist = np.where(mask > 0, hist, 0)
    if kind == NRX_BAG_SUM:
        mask = mask * rng.random((B, L)).astype(np.float32)
    inputs = [torch.from_numpy(_ids(rng, news, (B,), "zipf")).to(DEV), torch.from_numpy(hist).to(DEV),
              torch.from_numpy(_ids(rng, users, (B,), "uniform")).to(DEV)]
    weights = [None, None if kind == NRX_BAG_MEAN else torch.from_numpy(mask).to(DEV), None]
    ups = (torch.from_numpy(rng.standard_normal((B, 3 * D)).astype(np.float32)).to(DEV), None, None)
    dense, cnt = _grads(plan, tables, inputs, weights, ups, "auto", monkeypatch)
    assert cnt.small_ok >= 1
    got, _ = _sink_dense(plan, tables, inputs, weights, ups, monkeypatch)
    for a, b in zip(dense, got):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize("case", ["all_distinct_4096", "one_row_4096", "multi_64", "multi_65", "multi_256_of_3", "two_rows_alternating"])
def test_small_backward_at_the_edges_of_its_paths(case, monkeypatch):
    """The kernel's internal switches: a full table of 4096 distinct rows (hash table at its densest), ONE row looked up 4096 times (a
    single wavefront-summed run), exactly 64 / 65 lookups sharing rows (the last size wavefront 0 sorts alone / the first the block sorts),
    runs of three, two rows taking turns."""
    rng = np.random.default_rng(17)
    D, rows = 16, 6000
    B = 4096
    if case == "all_distinct_4096":
        ids = rng.permutation(np.arange(1, rows))[:B]
    elif case == "one_row_4096":
        ids = np.full(B, 77)
    elif case in ("multi_64", "multi_65"):
        ids = rng.permutation(np.arange(200, rows))[:B]
        ids[:64] = np.repeat(np.arange(1, 33), 2)                             # 32 rows looked up twice: 64 lookups that need ordering
        if case == "multi_65":
            ids[64] = 1                                                       # ... one of them a third time
        ids = rng.permutation(ids)
    elif case == "multi_256_of_3":
        ids = rng.permutation(np.arange(400, rows))[:B]
        ids[:768] = np.repeat(np.arange(1, 257), 3)
        ids = rng.permutation(ids)
    else:
        ids = np.where(np.arange(B) % 2 == 0, 5, 9)
    ids = np.asarray(ids, np.int64)
    plan = ops.EmbedPlan([ops.Slot("f", NRX_SPARSE, 0, D, 0, 0, fm_field=1)], out_width=D, use_fm=True)
    tables = [torch.from_numpy(rng.standard_normal((rows, D)).astype(np.float32)).to(DEV)]
    inputs = [torch.from_numpy(ids).to(DEV)]
    up = rng.standard_normal((B, D)).astype(np.float32)
    upf = rng.standard_normal((B,)).astype(np.float32)
    ups = (torch.from_numpy(up).to(DEV), None, torch.from_numpy(upf).to(DEV))
    got = _check(plan, tables, inputs, [None], ups, monkeypatch, tol=2e-4)
    # one FM field: S = v, so d fm / d v_k = 0 for k >= 1 and 1 for k = 0
    g = up.astype(np.float64).copy()
    g[:, 0] += upf
    want = np.zeros((rows, D))
    np.add.at(want, ids, g)
    np.testing.assert_allclose(got[0].cpu().numpy(), want, rtol=3e-4, atol=3e-4)
    sink_rows, lookups = _sink_dense(plan, tables, inputs, [None], ups, monkeypatch)
    assert lookups == B and torch.equal(sink_rows[0].view(torch.int32), got[0].view(torch.int32))
