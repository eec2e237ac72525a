"""GPU parity tests: the HIP path (through the C-ABI, via news_recsys_amd.ops) against
 (1) golden vectors captured from the reference's own Python (tests/golden/*.npz),
 (2) the CPU oracle (oracle/ref_np.py) on seeded random inputs at sizes the oracle finishes in seconds,
 (3) size-independent properties at BASELINE.json's full sizes (identity tables, linearity).

Bars: bit-exact for gather / concat / column routing / integer utilities; for fp32 reductions the
tolerance is written next to each assert (the oracle and the kernels sum in different orders)."""
import os

import numpy as np
import pytest
import torch
import yaml

from news_recsys_amd import _lib, ops
from news_recsys_amd._lib import (NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM, NRX_DENSE, NRX_SPARSE)
from oracle import ref_np as R
from tests.conftest import CONFIGS, GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def params_of(g):
    return {k[len("param/"):]: v for k, v in g.items() if k.startswith("param/")}


def batch_of(g):
    return {k[len("batch/"):]: v for k, v in g.items() if k.startswith("batch/")}


def tables_of(p):
    pre = "embedding_tables."
    return {k[len(pre):-len(".weight")]: v for k, v in p.items() if k.startswith(pre)}


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def build_plan(space, np_tables, batch, names, wide_names=(), fm=False):
    """Test-side plan builder (the product one lives in BaseModel): sorted feature order, skip
    features missing from the batch, exactly like base_model.py:284-308."""
    tnames = sorted(np_tables)
    slots, inputs, weights = [], [], []
    col, wcol = 0, 0
    for fname in sorted(names):
        if fname not in batch:
            continue
        if fname in space.dense:
            slots.append(ops.Slot(fname, NRX_DENSE, -1, 1, 0, col))
            inputs.append(dev(batch[fname]))
            weights.append(None)
            col += 1
            continue
        tname = R.emb_table_name(fname, space.share)
        t = np_tables[tname]
        D = t.shape[1]
        ti = tnames.index(tname)
        if fname in space.array:
            m = batch.get(fname + "_mask")
            kind = NRX_BAG_MASKED_MEAN if m is not None else NRX_BAG_MEAN
            slots.append(ops.Slot(fname, kind, ti, D, batch[fname].shape[1], col, fm_field=int(fm)))
            weights.append(None if m is None else dev(m.astype(np.float32)))
        else:
            wide = fname in wide_names
            slots.append(ops.Slot(fname, NRX_SPARSE, ti, D, 0, col, wide_col=wcol if wide else -1, fm_field=int(fm)))
            weights.append(None)
            if wide:
                wcol += 1
                col -= 1
        inputs.append(dev(batch[fname]))
        col += D
    plan = ops.EmbedPlan(slots, out_width=col, wide_width=wcol, use_fm=fm)
    tables = [dev(np_tables[n]).requires_grad_(True) for n in tnames]
    return plan, tables, inputs, weights, tnames


@pytest.fixture(params=["small_kernel", "big_kernels"])
def kernel_family(request):
    """Both forward families in front of the same expectation: nrx_embed_fwd serves a batch up to the small-batch limit with
    the one-block-per-sample kernel (embed_fwd_small_kernel) and anything larger with the lane-group kernels (embed_fwd_ring /
    embed_fwd_generic / the wide-split kernels).  The limit is a run-time knob of the C-ABI (nrx_set_small_batch_max), so every
    shape below -- ragged last blocks included -- meets the oracle through BOTH."""
    lib = _lib.load()
    prev = lib.nrx_set_small_batch_max((1 << 20) if request.param == "small_kernel" else 0)
    yield request.param
    lib.nrx_set_small_batch_max(prev)


def space_of(cfg_name):
    cfg = yaml.safe_load(open(os.path.join(CONFIGS, cfg_name)))
    names = set(cfg["features"]["user_feature_names"]) | set(cfg["features"]["item_feature_names"])
    return R.FeatureSpace.from_yaml_dict(cfg), names, cfg


# ----------------------------------------------------------------------------- goldens
@pytest.mark.parametrize("gname,cfg", [("model_deep", "cf_deep_small.yaml"), ("model_fm", "cf_fm_small.yaml"),
                                       ("model_dcn", "cf_dcn_small.yaml"), ("model_widedeep", "cf_widedeep_small.yaml"),
                                       ("model_lr", "cf_lr_small.yaml")])
def test_golden_embed_concat_bit_exact(gname, cfg, kernel_family):
    g = load(gname)
    space, names, _ = space_of(cfg)
    plan, tables, inputs, weights, _ = build_plan(space, tables_of(params_of(g)), batch_of(g), names)
    out, _, _ = ops.embed_apply(plan, tables, inputs, weights)
    assert np.array_equal(out.detach().cpu().numpy(), g["out/features"])      # gather + concat: bit-exact


def test_golden_arrays_dense_shared(kernel_family):
    g = load("model_deep_array")
    space, names, _ = space_of("cf_array_small.yaml")
    p, b = params_of(g), batch_of(g)
    plan, tables, inputs, weights, _ = build_plan(space, tables_of(p), b, names)
    out = ops.embed_apply(plan, tables, inputs, weights)[0].detach().cpu().numpy()
    np.testing.assert_allclose(out, g["out/features"], rtol=1e-6, atol=1e-6)   # pooled: fp32 sum order
    assert np.array_equal(out[:, :40], g["out/features"][:, :40])            # single-valued cols exact
    assert np.all(out[1, 52:84] == 0.0)                                      # all-masked bag -> exact 0
    b2 = dict(b)
    b2["ctr"] = g["case2/batch/ctr"]
    b2["user_history_mask"] = g["case2/batch/user_history_mask"]
    del b2["user_click_cats_mask"]
    plan, tables, inputs, weights, _ = build_plan(space, tables_of(p), b2, set(g["case2/names_in"].tolist()))
    out2 = ops.embed_apply(plan, tables, inputs, weights)[0].detach().cpu().numpy()
    np.testing.assert_allclose(out2, g["case2/features"], rtol=1e-6, atol=1e-6)
    b3 = {k: v for k, v in b.items() if k != "category"}
    plan, tables, inputs, weights, _ = build_plan(space, tables_of(p), b3, {"user_id", "category", "item_id"})
    assert np.array_equal(ops.embed_apply(plan, tables, inputs, weights)[0].detach().cpu().numpy(), g["case3/features"])


def test_golden_fm_fused_and_standalone(kernel_family):
    g = load("model_fm")
    space, names, _ = space_of("cf_fm_small.yaml")
    plan, tables, inputs, weights, _ = build_plan(space, tables_of(params_of(g)), batch_of(g), names, fm=True)
    out, _, fm = ops.embed_apply(plan, tables, inputs, weights)
    bias = dev(params_of(g)["score_fc.bias"])
    pred = torch.sigmoid(fm[:, None] + bias)
    # sum-square identity in a different order than ATen: rtol 1e-5, atol 1e-5
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g["out/forward"], rtol=1e-5, atol=1e-5)
    fm2 = ops.fm_interaction(out.detach(), len(plan.slots), 16)
    np.testing.assert_allclose(fm2.detach().cpu().numpy(), fm.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)
    # training parity: loss + table grads through fused FM
    labels = dev(g["batch/label"][:, 0])
    loss = torch.nn.functional.binary_cross_entropy(pred.view(-1), labels)
    np.testing.assert_allclose(loss.item(), g["out/loss"], rtol=1e-5)
    loss.backward()
    tn = sorted(tables_of(params_of(g)))
    for t, name in zip(tables, tn):
        want = g[f"grad/embedding_tables.{name}.weight"]
        np.testing.assert_allclose(t.grad.detach().cpu().numpy(), want, rtol=1e-4, atol=1e-6)
        assert np.all(t.grad[0].detach().cpu().numpy() == 0)


def test_golden_widedeep_split_bit_exact():
    g = load("model_widedeep")
    space, names, cfg = space_of("cf_widedeep_small.yaml")
    wide = set(cfg["wide_and_deep_cfg"]["wide_feature_names"])
    plan, tables, inputs, weights, _ = build_plan(space, tables_of(params_of(g)), batch_of(g), names, wide_names=wide)
    deep_x, wide_x, _ = ops.embed_apply(plan, tables, inputs, weights)
    assert np.array_equal(wide_x.detach().cpu().numpy(), g["out/wide_x"])
    assert np.array_equal(deep_x.detach().cpu().numpy(), g["out/deep_x"])


@pytest.mark.parametrize("gname,cfg", [("model_deep", "cf_deep_small.yaml"), ("model_deep_array", "cf_array_small.yaml")])
def test_golden_deep_model_grads(gname, cfg):
    """Whole Deep model: HIP embed path + torch MLP head (rocBLAS) vs the reference's loss and grads."""
    g = load(gname)
    space, names, _ = space_of(cfg)
    p, b = params_of(g), batch_of(g)
    plan, tables, inputs, weights, tn = build_plan(space, tables_of(p), b, names)
    x = ops.embed_apply(plan, tables, inputs, weights)[0]
    ws, bs = R.mlp_params(p, "score_fc.network.network")
    h = x
    for i, (W, bb) in enumerate(zip(ws, bs)):
        h = torch.nn.functional.linear(h, dev(W), dev(bb))
        if i < len(ws) - 1:
            h = torch.relu(h)
    pred = torch.sigmoid(h)
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g["out/forward"], rtol=1e-4, atol=1e-5)
    loss = torch.nn.functional.binary_cross_entropy(pred.view(-1), dev(b["label"][:, 0]))
    np.testing.assert_allclose(loss.item(), g["out/loss"], rtol=1e-4)
    loss.backward()
    for t, name in zip(tables, tn):
        want = g[f"grad/embedding_tables.{name}.weight"]
        # dense grad via float atomics: order differs from autograd's index_add -> rtol 1e-4
        np.testing.assert_allclose(t.grad.detach().cpu().numpy(), want, rtol=1e-4, atol=1e-6)
        assert np.all(t.grad[0].detach().cpu().numpy() == 0)            # padding row never trains


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
@pytest.mark.parametrize("mtag", ["none", "bin", "w"])
def test_golden_bag_pool(tag, mtag):
    g = load("ops")
    emb = dev(g[f"pool/{tag}/emb"]).requires_grad_(True)
    mask = {"none": None, "bin": g[f"pool/{tag}/mask"], "w": g[f"pool/{tag}/wmask"]}[mtag]
    out = ops.bag_pool(emb, None if mask is None else dev(mask))
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"pool/{tag}/{mtag}/out"], rtol=1e-6, atol=1e-6)
    (out * dev(g[f"pool/{tag}/up"])).sum().backward()
    np.testing.assert_allclose(emb.grad.detach().cpu().numpy(), g[f"pool/{tag}/{mtag}/gemb"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_golden_fm_op(tag):
    g = load("ops")
    w, v = g[f"fm/{tag}/w"], g[f"fm/{tag}/v"]
    B, Fn, K = v.shape
    feat = np.concatenate([w[:, :, None], v], axis=2).reshape(B, Fn * (K + 1))   # field = [w | v]
    x = dev(feat).requires_grad_(True)
    pred = torch.sigmoid(ops.fm_interaction(x, Fn, K + 1)[:, None] + dev(g[f"fm/{tag}/bias"]))
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g[f"fm/{tag}/out"], rtol=1e-5, atol=1e-5)
    (pred * dev(g[f"fm/{tag}/up"])).sum().backward()
    gx = x.grad.detach().cpu().numpy().reshape(B, Fn, K + 1)
    np.testing.assert_allclose(gx[:, :, 0], g[f"fm/{tag}/gw"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gx[:, :, 1:], g[f"fm/{tag}/gv"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_golden_dcn_v1(tag):
    g = load("ops")
    x = dev(g[f"dcn1/{tag}/x"]).requires_grad_(True)
    w = dev(g[f"dcn1/{tag}/w"]).requires_grad_(True)
    b = dev(g[f"dcn1/{tag}/b"]).requires_grad_(True)
    out = ops.dcn_v1(x, w, b)
    ref = g[f"dcn1/{tag}/out"]
    # algebraic form vs the reference's outer-product form (SURVEY hard part 4): rtol 1e-5, atol 2e-6*max|out|
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-5, atol=2e-6 * np.abs(ref).max())
    (out * dev(g[f"dcn1/{tag}/up"])).sum().backward()
    for got, key in ((x.grad, "gx"), (w.grad, "gw"), (b.grad, "gb")):
        want = g[f"dcn1/{tag}/{key}"]
        np.testing.assert_allclose(got.detach().cpu().numpy(), want, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(want).max()))


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_golden_dcn_v1_layer_by_layer_with_separate_x0(tag):
    """DCNLayer.forward(x_l, x_0) for every layer (reference dcn_arch.py:14-30): chaining single-layer calls that take
    x_0 separately reproduces the reference stack's output AND all its gradients (the goldens are the reference's
    DCNNet run layer by layer through DCNLayer.forward)."""
    from news_recsys_amd.model.sort.dcn.dcn_arch import DCNLayer
    g = load("ops")
    x = dev(g[f"dcn1/{tag}/x"]).requires_grad_(True)
    w, b = g[f"dcn1/{tag}/w"], g[f"dcn1/{tag}/b"]
    layers = []
    for l in range(w.shape[0]):
        lay = DCNLayer(w.shape[1]).to(DEV)
        with torch.no_grad():
            lay.w.copy_(dev(w[l])[:, None])
            lay.b.copy_(dev(b[l])[:, None])
        layers.append(lay)
    xl = x
    for lay in layers:
        xl = lay(xl, x)                      # x_l is x_0 only for the first layer
    ref = g[f"dcn1/{tag}/out"]
    np.testing.assert_allclose(xl.detach().cpu().numpy(), ref, rtol=1e-5, atol=2e-6 * np.abs(ref).max())
    fused = ops.dcn_v1(x.detach(), dev(w), dev(b))
    np.testing.assert_allclose(xl.detach().cpu().numpy(), fused.cpu().numpy(), rtol=1e-6, atol=1e-6 * np.abs(ref).max())
    (xl * dev(g[f"dcn1/{tag}/up"])).sum().backward()
    tol = lambda want: dict(rtol=1e-4, atol=1e-5 * max(1.0, np.abs(want).max()))
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[f"dcn1/{tag}/gx"], **tol(g[f"dcn1/{tag}/gx"]))
    gw = np.stack([lay.w.grad[:, 0].cpu().numpy() for lay in layers])
    gb = np.stack([lay.b.grad[:, 0].cpu().numpy() for lay in layers])
    np.testing.assert_allclose(gw, g[f"dcn1/{tag}/gw"], **tol(g[f"dcn1/{tag}/gw"]))
    np.testing.assert_allclose(gb, g[f"dcn1/{tag}/gb"], **tol(g[f"dcn1/{tag}/gb"]))


def test_dcn_v1_stack_from_a_later_layer_matches_oracle():
    """nrx_dcn_v1_fwd / _bwd with x != x0 and SEVERAL layers, odd width (scalar path) and width 320 (C3), vs torch
    autograd on the algebraic form (fp32, same device)."""
    gen = torch.Generator(device=DEV).manual_seed(3)
    for B, D, NL in ((37, 22, 3), (513, 320, 2), (64, 112, 1)):
        x0 = torch.randn(B, D, device=DEV, generator=gen)
        xs = torch.randn(B, D, device=DEV, generator=gen)
        w = (torch.randn(NL, D, device=DEV, generator=gen) * 0.1)
        b = torch.randn(NL, D, device=DEV, generator=gen) * 0.1
        up = torch.randn(B, D, device=DEV, generator=gen)
        a = [t.clone().requires_grad_(True) for t in (xs, x0, w, b)]
        out = ops.dcn_v1(a[0], a[2], a[3], x0=a[1])
        r = [t.clone().double().requires_grad_(True) for t in (xs, x0, w, b)]
        xl = r[0]
        for l in range(NL):
            xl = r[1] * (xl * r[2][l]).sum(1, keepdim=True) + r[3][l] + xl
        torch.testing.assert_close(out.detach().double(), xl.detach(), rtol=1e-5, atol=1e-5)
        (out * up).sum().backward()
        (xl * up.double()).sum().backward()
        for got, want in zip(a, r):
            torch.testing.assert_close(got.grad.double(), want.grad, rtol=1e-4, atol=1e-4 * max(1.0, want.grad.abs().max().item()))


def test_dcn_v1_cat_inplace_matches_separate():
    g = load("ops")
    x = g["dcn1/a/x"]
    B, D = x.shape
    buf = torch.empty(B, 2 * D, device=DEV)
    buf[:, :D] = dev(x)
    w, b = dev(g["dcn1/a/w"]), dev(g["dcn1/a/b"])
    out = ops.dcn_v1_cat_(buf, w, b)
    sep = ops.dcn_v1(dev(x), w, b)
    assert torch.equal(out[:, D:], sep) and torch.equal(out[:, :D], dev(x))


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_golden_dcn_v2_mfma(tag):
    g = load("ops")
    x = dev(g[f"dcn2/{tag}/x"]).requires_grad_(True)
    W = dev(g[f"dcn2/{tag}/W"]).requires_grad_(True)
    b = dev(g[f"dcn2/{tag}/b"]).requires_grad_(True)
    out = ops.dcn_v2(x, W, b)
    ref = g[f"dcn2/{tag}/out"]
    # fp32 GEMM, accumulation order differs from the CPU BLAS: rtol 1e-4
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref).max()))
    (out * dev(g[f"dcn2/{tag}/up"])).sum().backward()
    for got, key in ((x.grad, "gx"), (W.grad, "gW"), (b.grad, "gb")):
        want = g[f"dcn2/{tag}/{key}"]
        np.testing.assert_allclose(got.detach().cpu().numpy(), want, rtol=1e-3, atol=5e-5 * max(1.0, np.abs(want).max()))


# ----------------------------------------------------------------------------- oracle, seeded random
def _rand_case(rng, B, specs, idx_dtype=np.int64):
    """specs: list of (kind, rows, D, L).  Returns numpy tables / batch + FeatureSpace."""
    sparse, dense, array = [], [], []
    tables, batch = {}, {}
    for i, (kind, rows, D, L) in enumerate(specs):
        name = f"f{i:02d}"
        if kind == NRX_DENSE:
            dense.append(name)
            batch[name] = rng.random(B)
            continue
        t = rng.standard_normal((rows, D)).astype(np.float32)
        t[0] = 0
        tables[name] = t
        if kind == NRX_SPARSE:
            sparse.append(name)
            ids = rng.integers(0, rows, B)
            ids[: min(B, 3)] = [rows - 1, 0, rows - 1][: min(B, 3)]
            batch[name] = ids.astype(idx_dtype)
        else:
            array.append(name)
            ids = rng.integers(1, rows, (B, L))
            lens = rng.integers(0, L + 1, B)
            if B > 1:
                lens[0], lens[1] = L, 0
            m = (np.arange(L)[None] < lens[:, None]).astype(np.float32)
            batch[name] = (ids * m.astype(np.int64)).astype(idx_dtype)
            if kind == NRX_BAG_MASKED_MEAN:
                batch[name + "_mask"] = m
    return R.FeatureSpace(sparse, dense, array), tables, batch


UNIFORM_CASES = [(1, 5, 16), (63, 26, 16), (64, 26, 16), (1000, 26, 16), (257, 13, 16), (300, 40, 32), (129, 5, 64),
                 (77, 8, 128), (33, 3, 256), (500, 27, 16), (100, 1, 32), (90, 14, 64), (4096, 9, 16),
                 # partial last blocks above the default small-batch limit (2048): the ring kernel's tail, with and without the FM epilogue
                 (2049, 26, 16), (4097, 26, 16), (5000, 13, 32), (2051, 5, 64), (2500, 7, 256)]


@pytest.mark.parametrize("B,F,D", UNIFORM_CASES)
@pytest.mark.parametrize("idx_dtype", [np.int64, np.int32])
def test_uniform_gather_concat_bit_exact_vs_oracle(B, F, D, idx_dtype, kernel_family):
    rng = np.random.default_rng(B * 1000 + F * 10 + D)
    space, tables, batch = _rand_case(rng, B, [(NRX_SPARSE, 50 + 7 * i, D, 0) for i in range(F)], idx_dtype)
    want, _, _ = R.embed_concat(space, tables, batch, set(tables))
    plan, tt, inputs, weights, _ = build_plan(space, tables, batch, set(tables))
    out = ops.embed_apply(plan, tt, inputs, weights)[0]
    assert np.array_equal(out.detach().cpu().numpy(), want)
    # FM epilogue on the same inputs (fused, uniform kernel)
    plan, tt, inputs, weights, _ = build_plan(space, tables, batch, set(tables), fm=True)
    out, _, fm = ops.embed_apply(plan, tt, inputs, weights)
    assert np.array_equal(out.detach().cpu().numpy(), want)
    w, v = R.fm_split(want, [D] * F)
    ref = R.fm_logit(w.astype(np.float64), v.astype(np.float64), 0.0)[:, 0]
    # fp32 accumulation over F*D terms against a float64 oracle value: rtol 1e-5, atol 1e-5*scale
    scale = max(1.0, np.abs(ref).max())
    np.testing.assert_allclose(fm.detach().cpu().numpy(), ref, rtol=2e-5, atol=2e-5 * scale)
    # FM-only inference (no concat written)
    fm_only = ops.embed_apply(plan, tt, inputs, weights, need_out=False)[2]
    assert torch.equal(fm_only, fm)


GENERIC_CASES = {
    "mixed_dims": [(NRX_SPARSE, 97, 32, 0), (NRX_SPARSE, 61, 32, 0), (NRX_SPARSE, 18, 16, 0), (NRX_SPARSE, 27, 16, 0)],
    "odd_dims": [(NRX_SPARSE, 40, 17, 0), (NRX_SPARSE, 30, 1, 0), (NRX_SPARSE, 50, 5, 0), (NRX_SPARSE, 9, 33, 0)],
    # single-valued features of several 4Q widths: one uniform launch per width into the same concat (not the generic kernel)
    "many_mixed_dims": [(NRX_SPARSE, 300 + 7 * i, (16, 32, 64, 16, 128)[i % 5], 0) for i in range(13)],
    # a ranker's feature set: eight + two single-valued features of widths 16 / 32 (uniform launches) next to a dense value, a history
    # bag and an odd width (generic kernel) -- one concat
    # (the unaligned ones last: a feature whose first column is not a multiple of 4 cannot take the 16-byte stores)
    "deep_like_hybrid": [(NRX_SPARSE, 90 + i, 16, 0) for i in range(4)] + [(NRX_BAG_MASKED_MEAN, 70, 16, 20)] +
                        [(NRX_SPARSE, 50 + i, 16, 0) for i in range(4)] + [(NRX_SPARSE, 33, 32, 0), (NRX_SPARSE, 44, 32, 0),
                                                                            (NRX_SPARSE, 21, 5, 0), (NRX_DENSE, 0, 1, 0)],
    "lr_dim1": [(NRX_SPARSE, 40, 1, 0)] * 5,
    "bags": [(NRX_SPARSE, 100, 16, 0), (NRX_BAG_MASKED_MEAN, 200, 16, 50), (NRX_BAG_MEAN, 30, 16, 7)],
    "bag_long_odd": [(NRX_BAG_MASKED_MEAN, 64, 12, 333), (NRX_DENSE, 0, 1, 0), (NRX_SPARSE, 11, 8, 0)],
    # DSSM-tower shape: at B = 4096 the history table gets 204 800 lookups (50 planner tiles: chunked scans) next to one-tile tables
    "tower_large": [(NRX_SPARSE, 50000, 16, 0), (NRX_BAG_MASKED_MEAN, 20000, 16, 50), (NRX_SPARSE, 3000, 16, 0)],
    "wide_row": [(NRX_SPARSE, 20, 300, 0), (NRX_BAG_MASKED_MEAN, 20, 260, 9)],
    "dense_only_plus_one": [(NRX_DENSE, 0, 1, 0), (NRX_DENSE, 0, 1, 0), (NRX_SPARSE, 5, 4, 0)],
}


@pytest.mark.parametrize("case", sorted(GENERIC_CASES))
@pytest.mark.parametrize("B", [1, 37, 256, 1031, 2049, 5000])
def test_generic_embed_vs_oracle(case, B, monkeypatch, kernel_family):
    if case in ("many_mixed_dims", "deep_like_hybrid"):
        monkeypatch.setenv("NRX_SPLIT_MIN_LOOKUPS", "0")      # small batches: force the per-width split these cases are about
    rng = np.random.default_rng(sum(map(ord, case)) + B)
    space, tables, batch = _rand_case(rng, B, GENERIC_CASES[case])
    names = set(tables) | space.dense
    want, dims, _, used = R.embed_concat_ex(space, tables, batch, names)
    plan, tt, inputs, weights, _ = build_plan(space, tables, batch, names)
    out = ops.embed_apply(plan, tt, inputs, weights)[0].detach().cpu().numpy()
    col = 0
    for fname, d in zip(used, dims):
        blk, ref = out[:, col:col + d], want[:, col:col + d]
        if fname in space.array:
            np.testing.assert_allclose(blk, ref, rtol=1e-6, atol=1e-6)     # pooled: fp32 sum order
        elif fname in space.dense:
            assert np.array_equal(blk, ref)
        else:
            assert np.array_equal(blk, ref)                                # copies: bit-exact
        col += d


@pytest.mark.parametrize("B", [63, 2049, 4097, 5000])
@pytest.mark.parametrize("shape", ["uniform_fm", "uniform_plain", "masked_bag_mix", "fm_with_bag"])
def test_nothing_is_written_past_the_batch(shape, B, kernel_family):
    """The tail guard of every forward kernel, seen from outside: the C-ABI call gets buffers with 300 spare rows behind the batch, filled
    with a sentinel, and a batch that ends inside a block.  Rows [0, B) must equal the oracle, rows [B, B + 300) must still hold the
    sentinel -- `out`, the FM logits and the FM field sums alike.  (A guard that lets the last block's idle lanes through reads ids past the
    id arrays and stores rows past the batch: this is the test that sees it.)"""
    import ctypes as C
    rng = np.random.default_rng(B + len(shape))
    D = 16
    if shape.startswith("uniform"):
        specs = [(NRX_SPARSE, 60 + 5 * i, D, 0) for i in range(26)]
    elif shape == "masked_bag_mix":
        specs = [(NRX_SPARSE, 300, D, 0), (NRX_BAG_MASKED_MEAN, 200, D, 50), (NRX_SPARSE, 90, D, 0), (NRX_BAG_MEAN, 40, D, 7)]
    else:
        specs = [(NRX_SPARSE, 300, D, 0), (NRX_BAG_MASKED_MEAN, 200, D, 20), (NRX_SPARSE, 90, D, 0)]
    fm = shape in ("uniform_fm", "fm_with_bag")
    PAD = 300
    space, tables, batch = _rand_case(rng, B + PAD, specs)
    # ids of the spare samples are LEGAL ids: whatever a broken guard stores there is a real row, not a fault -- the sentinel must catch it
    names = set(tables)
    plan, tt, inputs, weights, _ = build_plan(space, tables, batch, names, fm=fm)
    head = {k: v[:B] for k, v in batch.items()}
    want, dims, _, used = R.embed_concat_ex(space, tables, head, names)
    lib = _lib.load()
    W = plan.out_width
    SENT = 12345.5
    out = torch.full((B + PAD, W), SENT, device=DEV)
    fm_out = torch.full((B + PAD,), SENT, device=DEV)
    sums = torch.full((B + PAD, D), SENT, device=DEV)
    status = torch.zeros(4, dtype=torch.int32, device=DEV)
    arr = ops._fill_features(plan, 0, len(plan.slots), [t.detach() for t in tt], inputs, weights)
    rc = lib.nrx_embed_fwd_train(arr, len(plan.slots), B, out.data_ptr(), W, None, 0, fm_out.data_ptr() if fm else None,
                                 sums.data_ptr() if fm else None, D, status.data_ptr(), None)
    assert rc == 0, lib.nrx_last_error()
    torch.cuda.synchronize()
    assert status.tolist()[0] == 0
    got = out.cpu().numpy()
    assert np.all(got[B:] == SENT), "rows past the batch were written"
    col = 0
    for fname, d in zip(used, dims):
        if fname in space.array:
            np.testing.assert_allclose(got[:B, col:col + d], want[:, col:col + d], rtol=1e-6, atol=1e-6)     # pooled: fp32 sum order
        else:
            assert np.array_equal(got[:B, col:col + d], want[:, col:col + d])
        col += d
    if fm:
        assert np.all(fm_out.cpu().numpy()[B:] == SENT) and np.all(sums.cpu().numpy()[B:] == SENT)
        w, v = R.fm_split(got[:B].astype(np.float64), dims)
        ref = R.fm_logit(w, v, 0.0)[:, 0]
        # fp32 accumulation against float64 over the same (pooled) field values: rtol 2e-5, atol 2e-5 * scale
        np.testing.assert_allclose(fm_out.cpu().numpy()[:B], ref, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(ref).max()))
        S = np.concatenate([w.sum(1, keepdims=True), v.sum(1)], axis=1)
        np.testing.assert_allclose(sums.cpu().numpy()[:B], S, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(S).max()))


@pytest.mark.parametrize("F,D,L", [(32, 256, 0), (31, 256, 0), (1, 64, 127), (1, 64, 128)])
def test_small_batch_plans_at_the_lds_limit(F, D, L, kernel_family):
    """Plans at the edge of what the one-block-per-sample kernel may take (<= 2048 work items AND an LDS image within a block's 64 KB):
    32 x D = 256 is 2048 items + 2048 output chunks = 74 KB -- it must go to the lane-group kernels instead of failing the launch."""
    rng = np.random.default_rng(F * D + L)
    B = 5
    specs = [(NRX_SPARSE, 40 + i, D, 0) for i in range(F)] if L == 0 else [(NRX_BAG_MASKED_MEAN, 90, D, L), (NRX_SPARSE, 33, D, 0)]
    space, tables, batch = _rand_case(rng, B, specs)
    want, dims, _, used = R.embed_concat_ex(space, tables, batch, set(tables))
    plan, tt, inputs, weights, _ = build_plan(space, tables, batch, set(tables))
    out = ops.embed_apply(plan, tt, inputs, weights)[0].detach().cpu().numpy()
    if L == 0:
        assert np.array_equal(out, want)
    else:
        np.testing.assert_allclose(out, want, rtol=1e-6, atol=1e-6)          # pooled: fp32 sum order


def test_out_of_range_report_names_the_callers_feature_in_split_launches(monkeypatch):
    """A feature set served by several launches (uniform launches per width + the generic kernel for the rest): the IndexError
    still names the feature by its position in the caller's list, whichever launch met the bad id."""
    monkeypatch.setenv("NRX_SPLIT_MIN_LOOKUPS", "0")
    rng = np.random.default_rng(11)
    B = 64
    space, tables, batch = _rand_case(rng, B, GENERIC_CASES["deep_like_hybrid"])
    names = set(tables) | space.dense
    plan, tt, inputs, weights, _ = build_plan(space, tables, batch, names)
    order = [s.name for s in plan.slots]
    for victim in (order.index("f07"), order.index("f10"), order.index("f04"), order.index("f11")):    # width 16, width 32, the bag, the odd width
        bad = [x.clone() for x in inputs]
        flat = bad[victim].view(-1)
        flat[3] = 10 ** 6
        w = list(weights)
        if w[victim] is not None:
            w[victim] = w[victim].clone()
            w[victim].view(-1)[3] = 1.0
        with pytest.raises(IndexError, match=plan.slots[victim].name):
            ops.embed_apply(plan, tt, bad, w, index_check="sync")


def test_bag_sum_kind_and_weights():
    rng = np.random.default_rng(5)
    B, L, D, rows = 70, 13, 16, 40
    t = rng.standard_normal((rows, D)).astype(np.float32)
    ids = rng.integers(0, rows, (B, L))
    w = rng.random((B, L)).astype(np.float32)
    w[rng.random((B, L)) < 0.3] = 0
    for weights in (None, w):
        plan = ops.EmbedPlan([ops.Slot("h", NRX_BAG_SUM, 0, D, L, 0)], out_width=D)
        out = ops.embed_apply(plan, [dev(t)], [dev(ids)], [None if weights is None else dev(weights)])[0]
        ref = (t[ids] * (1.0 if weights is None else weights[:, :, None])).sum(axis=1)
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


def test_oob_index_raises_indexerror_like_torch_cpu():
    t = torch.randn(10, 16, device=DEV)
    plan = ops.EmbedPlan([ops.Slot("a", NRX_SPARSE, 0, 16)], out_width=16)
    for bad in (10, -1, 2 ** 40):
        with pytest.raises(IndexError):
            ops.embed_apply(plan, [t], [torch.tensor([1, bad, 3], device=DEV)], [None])
    plan = ops.EmbedPlan([ops.Slot("a", NRX_SPARSE, 0, 16), ops.Slot("h", NRX_BAG_MEAN, 0, 16, 4, 16)], out_width=32)
    with pytest.raises(IndexError):     # generic kernel, bag position
        ops.embed_apply(plan, [t], [torch.tensor([1, 2], device=DEV), torch.tensor([[1, 2, 3, 4], [1, 99, 3, 4]], device=DEV)],
                        [None, None])


def test_empty_batch():
    t = torch.randn(10, 16, device=DEV)
    plan = ops.EmbedPlan([ops.Slot("a", NRX_SPARSE, 0, 16)], out_width=16)
    out = ops.embed_apply(plan, [t], [torch.zeros(0, dtype=torch.long, device=DEV)], [None])[0]
    assert out.shape == (0, 16)


def test_more_than_64_features_split_over_launches():
    rng = np.random.default_rng(9)
    space, tables, batch = _rand_case(rng, 150, [(NRX_SPARSE, 30 + i, 16, 0) for i in range(70)])
    want, _, _ = R.embed_concat(space, tables, batch, set(tables))
    plan, tt, inputs, weights, _ = build_plan(space, tables, batch, set(tables), fm=True)
    out, _, fm = ops.embed_apply(plan, tt, inputs, weights)
    assert np.array_equal(out.detach().cpu().numpy(), want)
    w, v = R.fm_split(want, [16] * 70)
    ref = R.fm_logit(w.astype(np.float64), v.astype(np.float64), 0.0)[:, 0]
    np.testing.assert_allclose(fm.detach().cpu().numpy(), ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())


@pytest.mark.parametrize("case", ["mixed_dims", "odd_dims", "bags", "bag_long_odd"])
def test_embed_backward_dense_grads_vs_oracle(case):
    rng = np.random.default_rng(77)
    B = 300
    space, tables, batch = _rand_case(rng, B, GENERIC_CASES[case])
    names = set(tables) | space.dense
    plan, tt, inputs, weights, tn = build_plan(space, tables, batch, names)
    out = ops.embed_apply(plan, tt, inputs, weights)[0]
    up = rng.standard_normal(out.shape).astype(np.float32)
    (out * dev(up)).sum().backward()
    _, dims, _, used = R.embed_concat_ex(space, tables, batch, names)
    col = 0
    want = {n: np.zeros_like(tables[n]) for n in tables}
    for fname, d in zip(used, dims):
        u = up[:, col:col + d]
        col += d
        if fname in space.dense:
            continue
        if fname in space.array:
            rows_up = R.array_pool_bwd(tables[fname][batch[fname]], batch.get(fname + "_mask"), u)
            want[fname] += R.embedding_grad_dense(batch[fname], rows_up, tables[fname].shape[0])
        else:
            want[fname] += R.embedding_grad_dense(batch[fname], u, tables[fname].shape[0])
    for t, name in zip(tt, tn):
        # float atomics accumulate in arbitrary order: rtol 1e-4, atol 1e-5
        np.testing.assert_allclose(t.grad.detach().cpu().numpy(), want[name], rtol=1e-4, atol=1e-5)
        assert np.all(t.grad[0].detach().cpu().numpy() == 0)


@pytest.mark.parametrize("B,D,NL", [(1, 4, 1), (100, 320, 2), (257, 112, 3), (65, 37, 3), (31, 1000, 8), (50, 2048, 1), (9, 6, 0),
                                    (41, 96, 6), (70, 320, 5), (33, 640, 2)])      # deeper / wider than the register-accumulation forms
def test_dcn_v1_vs_oracle(B, D, NL):
    rng = np.random.default_rng(B + D)
    x = rng.standard_normal((B, D)).astype(np.float32)
    w = (rng.standard_normal((NL, D)) / np.sqrt(D)).astype(np.float32)
    b = (rng.standard_normal((NL, D)) * 0.1).astype(np.float32)
    xt, wt, bt = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    out = ops.dcn_v1(xt, wt, bt)
    ref = R.dcn_v1(x, w, b)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-5, atol=2e-6 * max(1.0, np.abs(ref).max()))
    up = rng.standard_normal((B, D)).astype(np.float32)
    (out * dev(up)).sum().backward()
    gx, gw, gb = R.dcn_v1_bwd(x, w, b, up)
    for got, want in ((xt.grad, gx), (wt.grad, gw), (bt.grad, gb)):
        np.testing.assert_allclose(got.detach().cpu().numpy(), want, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(want).max() if want.size else 0.0))


def test_dcn_v1_ordered_mode_says_so_when_a_stack_is_beyond_it(monkeypatch):
    """8 layers x 1000 columns do not fit the LDS slabs of the fixed-order block sum: nrx_dcn_v1_bwd_ordered refuses (NRX_ERR_UNSUPPORTED, nothing
    enqueued) and ops takes the atomic launch -- with a warning and a count that a deterministic capture refuses on, not silently."""
    import warnings
    monkeypatch.setattr(ops, "WGRAD_ORDERED", True)
    monkeypatch.setattr(ops, "_atomic_warned", set())
    rng = np.random.default_rng(8)
    B, D, NL = 300, 1000, 8
    x = dev(rng.standard_normal((B, D)).astype(np.float32)).requires_grad_(True)
    w = dev((rng.standard_normal((NL, D)) / np.sqrt(D)).astype(np.float32)).requires_grad_(True)
    b = dev((rng.standard_normal((NL, D)) * 0.1).astype(np.float32)).requires_grad_(True)
    before = ops.dense_bwd_paths["atomic"]
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        ops.dcn_v1(x, w, b).sum().backward()
        torch.cuda.synchronize()
    assert any("beyond the ordered mode" in str(r.message) for r in rec)
    assert ops.dense_bwd_paths["atomic"] == before + 1
    gx, gw, gb = R.dcn_v1_bwd(x.detach().cpu().numpy(), w.detach().cpu().numpy(), b.detach().cpu().numpy(), np.ones((B, D), np.float32))
    np.testing.assert_allclose(w.grad.cpu().numpy(), gw, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(gw).max()))


@pytest.mark.parametrize("B,D,NL", [(1, 4, 1), (100, 320, 2), (9000, 112, 3), (6500, 37, 3), (3100, 320, 8), (5000, 2048, 1), (30000, 96, 6),
                                    (70000, 320, 5), (20011, 640, 2), (65536, 64, 4), (12345, 128, 2)])
@pytest.mark.parametrize("sep", [False, True])
def test_dcn_v1_ordered_cross_gradients_are_bit_reproducible(B, D, NL, sep, monkeypatch):
    """ops.WGRAD_ORDERED (nrx_dcn_v1_bwd_ordered): every block leaves its sums of g_w / g_b in its own slot and a second launch adds the blocks in
    block order -- three runs word for word the same, every launch form of the backward (register accumulation 1 .. 4 layers, the slab body for
    deeper / wider stacks, two rows per wavefront for dim <= 128, a separate layer-0 input), and equal to the atomic mode within the order of
    its sums; g_x is untouched by the switch."""
    rng = np.random.default_rng(B + D + NL)
    x = dev(rng.standard_normal((B, D)).astype(np.float32)).requires_grad_(True)
    x0 = dev(rng.standard_normal((B, D)).astype(np.float32)).requires_grad_(True) if sep else None
    w = dev((rng.standard_normal((NL, D)) / np.sqrt(D)).astype(np.float32)).requires_grad_(True)
    b = dev((rng.standard_normal((NL, D)) * 0.1).astype(np.float32)).requires_grad_(True)
    up = dev(rng.standard_normal((B, D)).astype(np.float32))

    def grads(ordered):
        monkeypatch.setattr(ops, "WGRAD_ORDERED", ordered)
        monkeypatch.setattr(ops, "WGRAD_ATOMIC", not ordered)          # (the default, "auto", takes the ordered mode too)
        if sep:
            gs = torch.autograd.grad(ops.dcn_v1(x, w, b, x0=x0), [x, x0, w, b], up)
        else:
            gs = torch.autograd.grad(ops.dcn_v1(x, w, b), [x, w, b], up)
        torch.cuda.synchronize()
        return [t.clone() for t in gs]
    ref = grads(False)
    runs = [grads(True) for _ in range(3)]
    for r in runs[1:]:
        for t0, t1 in zip(runs[0], r):
            assert torch.equal(t0.view(torch.int32), t1.view(torch.int32))
    n_x = 2 if sep else 1
    for k in range(n_x):
        assert torch.equal(runs[0][k], ref[k])
    for t0, t1 in zip(runs[0][n_x:], ref[n_x:]):
        torch.testing.assert_close(t0, t1, rtol=2e-4, atol=2e-5 * max(1.0, B ** 0.5) * max(1.0, t1.abs().max().item()) * 0.1)


@pytest.mark.parametrize("B,D,NL", [(1, 4, 1), (200, 320, 2), (130, 112, 3), (65, 37, 2), (1000, 64, 1), (127, 129, 1),
                                    # 64 < D <= 128, D % 4 == 0: the narrow kernel (W in registers, persistent 64-row tiles) at every
                                    # group count, partial last tiles, and batches larger than one pass of the persistent grid
                                    (63, 68, 2), (64, 80, 1), (1, 96, 1), (257, 100, 2), (40000, 112, 1), (131, 128, 2), (70000, 124, 1)])
def test_dcn_v2_vs_oracle(B, D, NL):
    rng = np.random.default_rng(B * 3 + D)
    x = rng.standard_normal((B, D)).astype(np.float32)
    W = (rng.standard_normal((NL, D, D)) / np.sqrt(D)).astype(np.float32)
    # asymmetric W (transpose-detecting, guide rule 16): scale the upper triangle
    W = W * (1.0 + np.triu(np.ones((D, D), np.float32)))[None]
    b = (rng.standard_normal((NL, D)) * 0.1).astype(np.float32)
    out = ops.dcn_v2(dev(x), dev(W), dev(b))
    ref = R.dcn_v2(x, W, b)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref).max()))
    out_nr = ops.dcn_v2(dev(x), dev(W[:1]), dev(b[:1]), relu=False)
    lin = x.astype(np.float64) @ W[0].astype(np.float64).T + b[0]
    np.testing.assert_allclose(out_nr.detach().cpu().numpy(), x * lin + x, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(lin).max()))
    # the matrix-core kernel is a pinned fp32 order (k-ascending fma chain, then fma(x0, lin + b, x_l)):
    # value-for-value equal to the C oracle that restates that order, for every layer count and odd D
    from oracle import ref_c
    assert np.array_equal(out.detach().cpu().numpy(), ref_c.dcn_v2(x, W, b))
    assert np.array_equal(out_nr.detach().cpu().numpy(), ref_c.dcn_v2(x, W[:1], b[:1], relu=False))


def _dcn_v2_bwd_f64(x, W, b, up, relu, masks):
    """fp64 autograd of DCNv2Net written out (dcn_arch.py:33-50, 73-91), with the ReLU decisions taken from the fp32 forward
    (`masks[l]` = layer l's fp32 output > 0): a pre-activation within rounding of zero must not flip the comparison."""
    n = W.shape[0]
    x0 = x.astype(np.float64)
    xs, lins = [x0], []
    for l in range(n):
        lin = xs[-1] @ W[l].astype(np.float64).T + b[l][None, :]
        p = x0 * lin + xs[-1]
        lins.append(lin)
        xs.append(np.where(masks[l], p, 0.0) if relu else p)
    g = up.astype(np.float64)
    gx0 = np.zeros_like(x0)
    gW, gb = np.zeros(W.shape), np.zeros(b.shape)
    for l in reversed(range(n)):
        if relu:
            g = g * masks[l]
        glin = g * x0
        gx0 += g * lins[l]
        gW[l] = glin.T @ xs[l]
        gb[l] = glin.sum(axis=0)
        g = g + glin @ W[l].astype(np.float64)
    return g + gx0, gW, gb


@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("B,D,NL", [(1, 4, 1), (200, 320, 2), (130, 112, 3), (65, 37, 2), (1000, 64, 1), (127, 129, 1), (33, 16, 3),
                                    # whole 128-row tiles (the batched epilogue), a partial last tile, a last 32-row mask group that is
                                    # not full, more row groups than one sweep of the elementwise kernel, odd D with whole tiles
                                    (4133, 112, 2), (2048, 320, 2), (20011, 64, 1), (300, 1000, 1), (515, 37, 2), (40000, 16, 1)])
def test_dcn_v2_bwd_vs_fp64(B, D, NL, relu):
    from oracle import ref_c
    rng = np.random.default_rng(B * 7 + D + NL)
    x = rng.standard_normal((B, D)).astype(np.float32)
    W = (rng.standard_normal((NL, D, D)) / np.sqrt(D)).astype(np.float32)
    W = W * (1.0 + np.triu(np.ones((D, D), np.float32)))[None]          # asymmetric: a transposed operand cannot pass
    b = (rng.standard_normal((NL, D)) * 0.1).astype(np.float32)
    up = rng.standard_normal((B, D)).astype(np.float32)
    xt, Wt, bt = dev(x).requires_grad_(True), dev(W).requires_grad_(True), dev(b).requires_grad_(True)
    out = ops.dcn_v2(xt, Wt, bt, relu=relu)
    out.backward(dev(up))
    masks = [ref_c.dcn_v2(x, W[:l + 1], b[:l + 1], relu=relu) > 0 for l in range(NL)]      # value-exact with the device forward
    assert np.array_equal(out.detach().cpu().numpy() > 0, masks[-1])
    gx, gW, gb = _dcn_v2_bwd_f64(x, W, b, up, relu, masks)
    # fp32 matrix-core sums over D (g_x) and over the batch (g_W, g_b; atomics-ordered): tolerance grows with the sum length
    for got, want, length in ((xt.grad, gx, D * NL), (Wt.grad, gW, B), (bt.grad, gb, B)):
        tol = 3e-6 * max(1.0, length ** 0.5)
        err = np.abs(got.detach().cpu().numpy().astype(np.float64) - want).max()
        assert err <= tol * max(1.0, np.abs(want).max()), (err, np.abs(want).max())


@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("B,D", [(700, 112), (1, 8), (259, 320)])
def test_dcn_v2_layer_bwd_separate_x0(B, D, relu):
    """DCNv2Layer.forward(x_l, x_0) with x_0 != x_l (dcn_arch.py:39-50): all four gradients against fp64."""
    rng = np.random.default_rng(B + D)
    x0, xl, up = (rng.standard_normal((B, D)).astype(np.float32) for _ in range(3))
    W = (rng.standard_normal((D, D)) / np.sqrt(D)).astype(np.float32)
    b = (rng.standard_normal(D) * 0.1).astype(np.float32)
    t = [dev(a).requires_grad_(True) for a in (x0, xl, W, b)]
    out = ops.dcn_v2_layer(t[0], t[1], t[2], t[3], relu=relu)
    out.backward(dev(up))
    m = out.detach().cpu().numpy() > 0 if relu else np.ones((B, D), bool)
    g = up.astype(np.float64) * m
    lin = xl.astype(np.float64) @ W.astype(np.float64).T + b
    glin = g * x0
    want = (g * lin, g + glin @ W.astype(np.float64), glin.T @ xl.astype(np.float64), glin.sum(0))
    for got, w in zip(t, want):
        err = np.abs(got.grad.detach().cpu().numpy().astype(np.float64) - w).max()
        assert err <= 3e-6 * max(1.0, B ** 0.5, D ** 0.5) * max(1.0, np.abs(w).max()), err


@pytest.mark.parametrize("acc", [0, 1, 3])
@pytest.mark.parametrize("B,D", [(300, 112), (129, 37), (1000, 320)])
def test_dcn_v2_layer_bwd_capi_padded_leading_dims(B, D, acc):
    """nrx_dcn_v2_layer_bwd called directly with every operand inside a wider allocation (ld > dim) and all accumulate_x0
    modes: same results as the contiguous call (g_xl, g_x0 value for value -- their sums run in a fixed order)."""
    from news_recsys_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device=DEV).manual_seed(B + D + acc)
    rnd = lambda *shape: torch.randn(*shape, device=DEV, generator=gen)
    x0, xl, g = rnd(B, D), rnd(B, D), rnd(B, D)
    W, b = rnd(D, D) / D ** 0.5, rnd(D) * 0.1
    gx0_init = rnd(B, D)
    st = torch.cuda.current_stream().cuda_stream

    def run(pad):
        def wide(t, extra):                           # t's values in the first D columns of a [B, D + extra] allocation
            w = torch.full((B, D + extra), float("nan"), device=DEV)
            w[:, :D] = t
            return w
        X0, XL = wide(x0, pad), wide(xl, pad)
        OUT, LIN = wide(torch.zeros(B, D, device=DEV), pad), wide(torch.zeros(B, D, device=DEV), pad)
        ld = D + pad
        assert lib.nrx_dcn_v2_layer_fwd(X0.data_ptr(), XL.data_ptr(), ld, B, D, W.data_ptr(), b.data_ptr(), 1, OUT.data_ptr(), ld,
                                        LIN.data_ptr(), st) == 0
        G, GXL, GX0 = wide(g, 2 * pad), wide(torch.zeros(B, D, device=DEV), 3 * pad), wide(gx0_init, pad)
        gW, gb = torch.empty(D, D, device=DEV), torch.empty(D, device=DEV)
        ws = torch.empty(lib.nrx_dcn_v2_layer_bwd_workspace(B, D), dtype=torch.uint8, device=DEV)
        assert lib.nrx_dcn_v2_layer_bwd(X0.data_ptr(), XL.data_ptr(), ld, LIN.data_ptr(), OUT.data_ptr(), 1, B, D, W.data_ptr(), G.data_ptr(),
                                        D + 2 * pad, GXL.data_ptr(), D + 3 * pad, GX0.data_ptr(), D + pad, acc, gW.data_ptr(), gb.data_ptr(),
                                        ws.data_ptr(), st) == 0, lib.nrx_last_error()
        if pad:                                       # nothing outside the first D columns was touched
            assert torch.isnan(GXL[:, D:]).all() and torch.isnan(GX0[:, D:]).all()
        return GXL[:, :D].clone(), GX0[:, :D].clone(), gW, gb, OUT[:, :D].clone()

    ref, got = run(0), run(4)
    assert torch.equal(ref[4], got[4]) and torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
    torch.testing.assert_close(got[2], ref[2], rtol=1e-4, atol=1e-4)          # atomics-ordered sums over the batch
    torch.testing.assert_close(got[3], ref[3], rtol=1e-4, atol=1e-4)
    # and against the definition (fp64)
    m = (ref[4] > 0).double()
    gm = g.double() * m
    lin = xl.double() @ W.double().t() + b.double()
    gx0 = gm * lin + (gx0_init.double() if acc & 1 else 0)
    gxl = gm + (gm * x0.double()) @ W.double() + (gx0 if acc & 2 else 0)
    assert (got[1].double() - gx0).abs().max().item() <= 1e-4 * max(1.0, gx0.abs().max().item())
    assert (got[0].double() - gxl).abs().max().item() <= 1e-4 * max(1.0, gxl.abs().max().item())


@pytest.mark.parametrize("relu", [1, 0])
@pytest.mark.parametrize("acc", [0, 1, 3])
@pytest.mark.parametrize("B,D", [(1, 8), (63, 16), (64, 112), (65, 112), (1000, 112), (4133, 112), (777, 128), (300, 100), (129, 64), (500, 36), (2000, 124)])
def test_dcn_v2_layer_bwd_panel_form_equals_three_launch_path(B, D, acc, relu):
    """Narrow layers (dim <= 128): preparation + dgrad in one launch (dcn2_bwd_panel_kernel) against the three-launch path in a child process
    started with NRX_DCN2_PANEL=0 (the switch is read once per process) -- g_xl and g_x0 value for value (same sums in the same order), g_W /
    g_b within the order of their batch-wide atomics; and all four against the fp64 definition (dcn_arch.py:33-50, 73-91)."""
    import subprocess, sys, os, tempfile
    from news_recsys_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device=DEV).manual_seed(B * 3 + D + acc)
    rnd = lambda *shape: torch.randn(*shape, device=DEV, generator=gen)
    x0, xl, g = rnd(B, D), rnd(B, D), rnd(B, D)
    W, b = rnd(D, D) / D ** 0.5, rnd(D) * 0.1
    W = W * (1.0 + torch.triu(torch.ones(D, D, device=DEV)))          # asymmetric: a transposed operand cannot pass
    gx0_init = rnd(B, D)
    with tempfile.TemporaryDirectory() as td:
        torch.save(dict(x0=x0.cpu(), xl=xl.cpu(), g=g.cpu(), W=W.cpu(), b=b.cpu(), gx0=gx0_init.cpu(), acc=acc, relu=relu), os.path.join(td, "in.pt"))
        code = (
            "import sys, torch\n"
            "sys.path.insert(0, %r)\n"
            "from news_recsys_amd import _lib\n"
            "lib = _lib.load()\n"
            "d = torch.load(%r)\n"
            "dev = 'cuda:0'\n"
            "x0, xl, g, W, b, gx0 = (d[k].to(dev) for k in ('x0', 'xl', 'g', 'W', 'b', 'gx0'))\n"
            "B, D = x0.shape\n"
            "out, lin = torch.empty_like(x0), torch.empty_like(x0)\n"
            "st = torch.cuda.current_stream().cuda_stream\n"
            "assert lib.nrx_dcn_v2_layer_fwd(x0.data_ptr(), xl.data_ptr(), D, B, D, W.data_ptr(), b.data_ptr(), d['relu'], out.data_ptr(), D, lin.data_ptr(), st) == 0\n"
            "gxl, gW, gb = torch.empty_like(x0), torch.empty(D, D, device=dev), torch.empty(D, device=dev)\n"
            "ws = torch.empty(lib.nrx_dcn_v2_layer_bwd_workspace(B, D), dtype=torch.uint8, device=dev)\n"
            "assert lib.nrx_dcn_v2_layer_bwd(x0.data_ptr(), xl.data_ptr(), D, lin.data_ptr(), out.data_ptr(), d['relu'], B, D, W.data_ptr(), g.data_ptr(), D,"
            " gxl.data_ptr(), D, gx0.data_ptr(), D, d['acc'], gW.data_ptr(), gb.data_ptr(), ws.data_ptr(), st) == 0\n"
            "torch.cuda.synchronize()\n"
            "torch.save(dict(gxl=gxl.cpu(), gx0=gx0.cpu(), gW=gW.cpu(), gb=gb.cpu(), out=out.cpu()), %r)\n"
        ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(td, "in.pt"), os.path.join(td, "out.pt"))
        env = dict(os.environ, NRX_DCN2_PANEL="0")
        subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=300)
        ref = torch.load(os.path.join(td, "out.pt"))
    out, lin = torch.empty_like(x0), torch.empty_like(x0)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.nrx_dcn_v2_layer_fwd(x0.data_ptr(), xl.data_ptr(), D, B, D, W.data_ptr(), b.data_ptr(), relu, out.data_ptr(), D, lin.data_ptr(), st) == 0
    gxl, gx0, gW, gb = torch.empty_like(x0), gx0_init.clone(), torch.empty(D, D, device=DEV), torch.empty(D, device=DEV)
    ws = torch.empty(lib.nrx_dcn_v2_layer_bwd_workspace(B, D), dtype=torch.uint8, device=DEV)
    assert lib.nrx_dcn_v2_layer_bwd(x0.data_ptr(), xl.data_ptr(), D, lin.data_ptr(), out.data_ptr(), relu, B, D, W.data_ptr(), g.data_ptr(), D,
                                    gxl.data_ptr(), D, gx0.data_ptr(), D, acc, gW.data_ptr(), gb.data_ptr(), ws.data_ptr(), st) == 0, lib.nrx_last_error()
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref["out"])
    assert torch.equal(gxl.cpu(), ref["gxl"]) and torch.equal(gx0.cpu(), ref["gx0"])
    torch.testing.assert_close(gW.cpu(), ref["gW"], rtol=1e-4, atol=1e-4 * max(1.0, B ** 0.5))
    torch.testing.assert_close(gb.cpu(), ref["gb"], rtol=1e-4, atol=1e-4 * max(1.0, B ** 0.5))
    m = (out > 0).double() if relu else torch.ones_like(out).double()
    gm = g.double() * m
    lin64 = xl.double() @ W.double().t() + b.double()
    want_gx0 = gm * lin64 + (gx0_init.double() if acc & 1 else 0)
    glin = gm * x0.double()
    want_gxl = gm + glin @ W.double() + (want_gx0 if acc & 2 else 0)
    tol = 1e-4
    assert (gx0.double() - want_gx0).abs().max().item() <= tol * max(1.0, want_gx0.abs().max().item())
    assert (gxl.double() - want_gxl).abs().max().item() <= tol * max(1.0, want_gxl.abs().max().item())
    want_gW, want_gb = glin.t() @ xl.double(), glin.sum(0)
    assert (gW.double() - want_gW).abs().max().item() <= 3e-6 * max(1.0, B ** 0.5) * max(1.0, want_gW.abs().max().item())
    assert (gb.double() - want_gb).abs().max().item() <= 3e-6 * max(1.0, B ** 0.5) * max(1.0, want_gb.abs().max().item())


# ----------------------------------------------------------------------------- integer utilities (bit-exact)
@pytest.mark.parametrize("n,world", [(0, 2), (1, 1), (63, 2), (2048, 8), (2049, 8), (100000, 8), (77777, 3), (5000, 64)])
@pytest.mark.parametrize("dtype", [torch.int64, torch.int32])
def test_bucketize_by_owner_bit_exact(n, world, dtype):
    rng = np.random.default_rng(n + world)
    ids = rng.integers(0, 1 << 20, n)
    counts, local_rows, slot = ops.bucketize_by_owner(torch.from_numpy(ids).to(DEV).to(dtype), world)
    c_ref, perm = R.bucketize_by_owner(ids, world)
    assert np.array_equal(counts.detach().cpu().numpy(), c_ref)
    assert np.array_equal(local_rows.detach().cpu().numpy(), (ids // world)[perm])     # stable send buffer
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    assert np.array_equal(slot.detach().cpu().numpy(), inv)


def test_gather_rows_segmented_bit_exact():
    rng = np.random.default_rng(3)
    D = 16
    tabs = [rng.standard_normal((r, D)).astype(np.float32) for r in (50, 7, 300)]
    seg_table = np.array([0, 2, 1, 0, 2], np.int32)
    seg_len = np.array([100, 0, 33, 1, 257])
    seg_start = np.concatenate([[0], np.cumsum(seg_len)])
    rows = np.concatenate([rng.integers(0, tabs[t].shape[0], l) for t, l in zip(seg_table, seg_len)])
    out = ops.gather_rows_segmented([dev(t) for t in tabs], dev(seg_start), dev(seg_table), dev(rows), int(seg_start[-1]))
    want = np.concatenate([tabs[t][rows[s:e]] for t, s, e in zip(seg_table, seg_start[:-1], seg_start[1:])])
    assert np.array_equal(out.detach().cpu().numpy(), want)
    bad = rows.copy()
    bad[120] = 7            # table 1 has 7 rows -> out of range
    with pytest.raises(IndexError):
        ops.gather_rows_segmented([dev(t) for t in tabs], dev(seg_start), dev(seg_table), dev(bad), int(seg_start[-1]))


def test_mask_lengths():
    rng = np.random.default_rng(4)
    m = (rng.random((1000, 50)) < 0.4).astype(np.float32)
    off, _ = R.csr_from_mask(m)
    assert np.array_equal(ops.mask_lengths(dev(m)).detach().cpu().numpy(), np.diff(off))


# ----------------------------------------------------------------------------- full-size properties
def test_full_size_c2_identity_tables_and_linearity():
    """BASELINE config 2 shape (26 x 1M rows x 16, B=65536): tables whose row r holds the value r in
    every column make the expected output computable from the ids alone (bit-exact, no oracle pass)."""
    F, rows, D, B = 26, 1_000_000, 16, 65536
    gen = torch.Generator(device=DEV).manual_seed(20260116)
    base = torch.arange(rows, device=DEV, dtype=torch.float32)[:, None].expand(rows, D).contiguous()
    tables = [base + f for f in range(F)]
    for t in tables:
        t[0] = 0
    ids = [torch.randint(1, rows, (B,), device=DEV, generator=gen) for _ in range(F)]
    slots = [ops.Slot(f"C{f:02d}", NRX_SPARSE, f, D, 0, f * D, fm_field=1) for f in range(F)]
    plan = ops.EmbedPlan(slots, out_width=F * D, use_fm=True)
    out, _, fm = ops.embed_apply(plan, tables, ids, [None] * F)
    want = torch.stack([ids[f].float() + f for f in range(F)], dim=1)[:, :, None].expand(B, F, D).reshape(B, F * D)
    assert torch.equal(out, want)
    # checksum of checksums: every row of the table is reachable and summed once per lookup
    assert out.double().sum().item() == want.double().sum().item()
    # FM on constant-per-field rows: closed form in float64
    vals = torch.stack([ids[f].double() + f for f in range(F)], dim=1)          # [B, F]
    ref = vals.sum(1) + 0.5 * (D - 1) * (vals.sum(1) ** 2 - (vals ** 2).sum(1))
    rel = ((fm.double() - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
    assert rel < 1e-5        # fp32 accumulation of ~1e13-sized terms


def test_full_size_c3_fused_gather_cross_properties():
    """Config 3 shape at full batch (5 features x D=64 -> x[65536, 320], 2 cross layers; the news table scaled to 20 M
    rows = 5 GB): the fused launch's left half must be the gathered rows exactly (identity tables: row r holds
    r * 2^-20 + feature), and its right half the cross of those rows -- checked in float64 on every 97th sample and, over
    the whole batch, through the identity cross(x; w = 0, b) = x + b_0 + b_1 (bit-exact in the kernel's order)."""
    D, B = 64, 65536
    rows = [18, 20_000_000, 270, 18, 1_000_000]
    gen = torch.Generator(device=DEV).manual_seed(33)
    tables = []
    for f, r in enumerate(rows):
        t = (torch.arange(r, device=DEV, dtype=torch.float32) * 2.0 ** -20 + f)[:, None].expand(r, D).contiguous()
        t[0] = 0
        tables.append(t)
    ids = [torch.randint(1, r, (B,), device=DEV, generator=gen) for r in rows]
    W = 5 * D
    plan = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D) for i in range(5)], out_width=W)
    assert ops.fused_cross_is_fast(plan)
    w = torch.randn(2, W, device=DEV, generator=gen) / W ** 0.5
    b = torch.randn(2, W, device=DEV, generator=gen) * 0.1
    buf = ops.embed_dcn_v1(plan, tables, ids, w, b)
    want_x = torch.stack([ids[f].float() * 2.0 ** -20 + f for f in range(5)], dim=1)[:, :, None].expand(B, 5, D).reshape(B, W)
    assert torch.equal(buf[:, :W], want_x)
    sub = slice(0, B, 97)
    x0 = want_x[sub].double()
    xl = x0
    for l in range(2):
        xl = x0 * (xl @ w[l].double())[:, None] + b[l].double() + xl
    torch.testing.assert_close(buf[sub, W:].double(), xl, rtol=1e-5, atol=1e-5)
    lin = ops.embed_dcn_v1(plan, tables, ids, torch.zeros_like(w), b)
    assert torch.equal(lin[:, W:], b[1] + (b[0] + want_x))          # the kernel's own order of the two additions: exact


def test_full_size_c4_history_pooling_property():
    """Config 4 shape (history L=50, D=16, B=65536, 200k-row news table): mean-pooling rows that all
    equal the row id gives mean(ids over valid positions); all-masked bags give exact zeros."""
    rows, D, B, L = 200_000, 16, 65536, 50
    gen = torch.Generator(device=DEV).manual_seed(20260120)
    table = torch.arange(rows, device=DEV, dtype=torch.float32)[:, None].expand(rows, D).contiguous()
    table[0] = 0
    lens = torch.randint(0, L + 1, (B,), device=DEV, generator=gen)
    mask = (torch.arange(L, device=DEV)[None] < lens[:, None]).float()
    ids = torch.randint(1, rows, (B, L), device=DEV, generator=gen) * mask.long()
    plan = ops.EmbedPlan([ops.Slot("user_history", NRX_BAG_MASKED_MEAN, 0, D, L, 0)], out_width=D)
    out = ops.embed_apply(plan, [table], [ids], [mask])[0]
    ref = (ids.double() * mask.double()).sum(1) / (mask.double().sum(1) + 1e-8)
    assert torch.all(out[lens == 0] == 0)
    rel = ((out.double() - ref[:, None]).abs() / ref[:, None].clamp_min(1.0)).max().item()
    assert rel < 1e-6        # stated fp32 pooling tolerance
    assert torch.equal(out[:, 0], out[:, D - 1])


def test_prepared_embed_matches_embed_apply_and_reruns():
    rng = np.random.default_rng(21)
    space, tables, batch = _rand_case(rng, 500, [(NRX_SPARSE, 90 + i, 16, 0) for i in range(26)])
    plan, tt, inputs, weights, _ = build_plan(space, tables, batch, set(tables), fm=True)
    ref_out, _, ref_fm = ops.embed_apply(plan, tt, inputs, weights)
    call = ops.PreparedEmbed(plan, tt, inputs, weights, check_index=True)
    for _ in range(3):
        out, _, fm = call.run()
    call.check()
    assert torch.equal(out, ref_out.detach()) and torch.equal(fm, ref_fm.detach())
    inputs[3].fill_(10 ** 6)                 # ids are re-read from the same tensors on every run
    call.run()
    with pytest.raises(IndexError):
        call.check()


@pytest.mark.parametrize("fm", [False, True])
def test_sorted_backward_hot_rows_long_segment_path(fm):
    """Skewed ids: rows looked up hundreds or thousands of times in one batch (a 40-row table, one id repeated 5000
    times) take the wavefront-per-chunk path of nrx_embed_bwd_sorted (work lists, partial sums, combine) -- same result as
    the dense scatter within float tolerance, identical bits from run to run."""
    rng = np.random.default_rng(77)
    B, D = 20000, 16
    rows = [40, 3000, 500000]
    tabs = [torch.randn(r, D, device=DEV) for r in rows]
    ids = [torch.from_numpy(rng.integers(0, r, B)).to(DEV) for r in rows]
    ids[2][:5000] = 4242                                        # one hot row: 20 chunks of 256 entries
    ids[1][100:130] = 7                                         # a 30-entry segment (single work item)
    slots = [ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=int(fm)) for i in range(3)]
    plan = ops.EmbedPlan(slots, out_width=3 * D, use_fm=fm)
    up = torch.randn(B, 3 * D, device=DEV)
    upf = torch.randn(B, device=DEV)

    def run(mode):
        ts = [t.clone().requires_grad_(True) for t in tabs]
        out, _, f_ = ops.embed_apply(plan, ts, ids, [None] * 3, sparse_grad=mode)
        loss = (out * up).sum() + ((f_ * upf).sum() if fm else 0)
        loss.backward()
        return [t.grad.to_dense() if t.grad.is_sparse else t.grad for t in ts]

    dense = run(False)
    s1, s2 = run(True), run(True)
    for a, b, c in zip(dense, s1, s2):
        assert torch.equal(b, c)                                # bit-reproducible
        scale = a.abs().max().item()
        torch.testing.assert_close(b, a, rtol=2e-4, atol=2e-5 * max(1.0, scale))
        assert torch.all(b[0] == 0)                             # padding row


@pytest.mark.parametrize("mix", ["uniform16", "with_bag"])
def test_fm_gradient_folded_into_embed_backward_all_three_modes(mix):
    """The FM epilogue's gradient rides in the embedding backward (nrx_fm_grad_t: field sums from nrx_embed_fwd_train,
    values from the forward concat) instead of a separate nrx_fm_bwd pass.  Dense, COO and the bound
    PreparedSparseBackward must all equal torch autograd on the reference formula (fm/model.py:18-26) in fp64."""
    rng = np.random.default_rng(33)
    B = 700
    feats = [(NRX_SPARSE, 40 + 3 * i, 16, 0) for i in range(9)]
    if mix == "with_bag":
        feats[4] = (NRX_BAG_MASKED_MEAN, 61, 16, 7)          # a pooled field inside the FM (generic kernel)
    space, tables, batch = _rand_case(rng, B, feats)
    plan, tt, inputs, weights, _ = build_plan(space, tables, batch, set(tables), fm=True)
    up = torch.randn(B, plan.out_width, device=DEV)
    upf = torch.randn(B, device=DEV)

    # reference: gather/pool in torch fp64, FM by the reference's formula
    t64 = [t.detach().double().requires_grad_(True) for t in tt]
    cols = []
    for s_, x, w in zip(plan.slots, inputs, weights):
        e = t64[s_.table][x.long()]
        if s_.kind == NRX_BAG_MASKED_MEAN:
            e = (e * w.double()[..., None]).sum(1) / (w.double().sum(1, keepdim=True) + 1e-8)
        cols.append(e)
    feat = torch.cat(cols, 1)
    fv = torch.stack(cols, 1)
    first = fv[:, :, 0].sum(1)
    v = fv[:, :, 1:]
    fmr = first + 0.5 * ((v.sum(1) ** 2) - (v ** 2).sum(1)).sum(1)
    ((feat * up.double()).sum() + (fmr * upf.double()).sum()).backward()
    want = [t.grad.clone() for t in t64]
    for w_ in want:
        w_[0].zero_()                                          # padding row never trains

    def check(grads, tag):
        for g_, w_ in zip(grads, want):
            gd = g_.to_dense() if g_.is_sparse else g_
            torch.testing.assert_close(gd.double(), w_, rtol=2e-4, atol=2e-4, msg=lambda m: f"{tag}: {m}")

    for mode in (False, True):
        ts = [t.detach().clone().requires_grad_(True) for t in tt]
        out, _, fm = ops.embed_apply(plan, ts, inputs, weights, sparse_grad=mode)
        torch.testing.assert_close(fm.double(), fmr.detach(), rtol=1e-5, atol=1e-4)
        ((out * up).sum() + (fm * upf).sum()).backward()
        check([t.grad for t in ts], f"sparse_grad={mode}")
    # only the FM output is used downstream (g_out is None)
    ts = [t.detach().clone().requires_grad_(True) for t in tt]
    _, _, fm = ops.embed_apply(plan, ts, inputs, weights)
    (fm * upf).sum().backward()
    t64b = [t.grad for t in ts]
    ts2 = [t.detach().clone().requires_grad_(True) for t in tt]
    out2, _, fm2 = ops.embed_apply(plan, ts2, inputs, weights)
    ((out2 * 0).sum() + (fm2 * upf).sum()).backward()
    for a_, b_ in zip(t64b, ts2):
        torch.testing.assert_close(a_, b_.grad, rtol=1e-4, atol=1e-5)       # float-atomic order differs between runs

    # bound form: PreparedEmbed(fm_sums=...) + PreparedSparseBackward, scattered into dense for the comparison
    D = 16
    sums = torch.empty(B, D, device=DEV)
    fwd = ops.PreparedEmbed(plan, tt, inputs, weights, fm_sums=sums)
    out, _, fm = fwd.run()
    fv32 = torch.stack([c.float() for c in cols], 1).detach()
    torch.testing.assert_close(sums[:, 1:], fv32[:, :, 1:].sum(1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(sums[:, 0], fv32[:, :, 0].sum(1), rtol=1e-5, atol=1e-5)
    bwd = ops.PreparedSparseBackward(fwd, up, upf)
    for _ in range(2):
        groups = bwd.run()
    dense = [torch.zeros_like(t) for t in tt]
    for g_ in groups:
        nu = int(g_["counts"][0])
        keys = g_["uniq"][:nu]
        for k_, v_ in zip(keys.tolist(), g_["values"][:nu]):
            dense[k_ >> 40][k_ & ((1 << 40) - 1)] += v_
    check(dense, "PreparedSparseBackward")


@pytest.mark.parametrize("world,lens,cap", [(1, [100], 100), (2, [63, 1, 300], 256), (8, [4096, 4096, 5000, 17], 2048),
                                            (3, [0, 777, 0, 2050], 1024), (8, [65536] * 5, 44000), (4, [5000], 64)])
@pytest.mark.parametrize("dtype", [torch.int64, torch.int32])
def test_route_ids_bit_exact_vs_oracle(world, lens, cap, dtype):
    """Fixed-capacity routing incl. feature boundaries inside a 64-id group, empty features and an
    overflowing block (last case)."""
    rng = np.random.default_rng(sum(lens) + world)
    arrays = [rng.integers(0, 1 << 20, n) for n in lens]
    send, slot, counts2d, overflow = ops.route_ids([torch.from_numpy(a).to(DEV).to(dtype) for a in arrays], world, cap)
    r_send, r_slot, r_counts, r_worst = R.route_ids(arrays, world, cap)
    assert np.array_equal(counts2d.cpu().numpy(), r_counts)
    assert int(overflow.item()) == r_worst
    assert np.array_equal(slot.cpu().numpy(), r_slot)
    valid = r_send >= 0
    assert np.array_equal(send.cpu().numpy()[valid], r_send[valid])      # unused slots are unspecified


@pytest.mark.parametrize("world,lens,cap", [(1, [100], 100), (2, [63, 1, 300], 256), (8, [4096, 4096, 5000, 17], 2048), (3, [2050, 0, 777], 1024)])
def test_route_ids_pos_and_one_sided_gather_c_abi(world, lens, cap):
    """nrx_route_ids_pos: the same send blocks / slots / counts as nrx_route_ids (oracle: ref_np.route_ids) plus, per sent id, its position inside
    its feature -- checked through the slot map (send_pos[slot[p]] == p's position).  Then nrx_gather_inbox_place with this process as every
    'source' (the send blocks taken as the inbox of ONE owner that holds whole tables): each row lands at peer_out[s][pos, col(feature)]."""
    import ctypes as C
    rng = np.random.default_rng(sum(lens) + world)
    rows = 5000
    arrays = [rng.integers(0, rows, n) for n in lens]
    ids = [torch.from_numpy(a).to(DEV) for a in arrays]
    send, slot, c2, over, pos = ops.route_ids(ids, world, cap, want_pos=True)
    r_send, r_slot, r_counts, r_worst = R.route_ids(arrays, world, cap)
    assert np.array_equal(c2.cpu().numpy(), r_counts) and int(over.item()) == r_worst and np.array_equal(slot.cpu().numpy(), r_slot)
    valid = r_send >= 0
    assert np.array_equal(send.cpu().numpy()[valid], r_send[valid])
    sl, ps = slot.cpu().numpy(), pos.cpu().numpy()
    off = 0
    for a in arrays:
        s_ = sl[off:off + len(a)]
        ok = s_ >= 0
        assert np.array_equal(ps[s_[ok]], np.arange(len(a))[ok])
        off += len(a)
    if r_worst > cap:
        return
    # owner side: treat block s of the send buffer as what "source s" sent to an owner whose local row l of feature f is table_f[l]
    lib = _lib.load()
    D, F = 16, len(lens)
    B = max(lens) if max(lens) else 1
    tabs = [torch.from_numpy(rng.standard_normal((rows // world + 2, D)).astype(np.float32)).to(DEV) for _ in range(F)]
    outs = [torch.full((B, F * D), -7.5, device=DEV) for _ in range(world)]
    tp = (C.c_void_p * F)(*[t.data_ptr() for t in tabs]); tr = (C.c_int64 * F)(*[t.shape[0] for t in tabs])
    ft = (C.c_int32 * F)(*range(F)); cols = (C.c_int32 * F)(*[f * D for f in range(F)])
    po = (C.c_void_p * world)(*[o.data_ptr() for o in outs])
    rc = lib.nrx_gather_inbox_place(tp, tr, F, ft, F, world, cap, c2.data_ptr(), send.data_ptr(), pos.data_ptr(), D, po, F * D, B, cols, None,
                                    torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.nrx_last_error()
    torch.cuda.synchronize()
    want = [np.full((B, F * D), -7.5, np.float32) for _ in range(world)]
    for f, a in enumerate(arrays):
        t = tabs[f].cpu().numpy()
        for i, v in enumerate(a):
            want[v % world][i, f * D:(f + 1) * D] = t[v // world]
    for o, w in zip(outs, want):
        assert np.array_equal(o.cpu().numpy(), w)
    # shapes the placing kernel does not take are refused, not mis-served
    po1 = (C.c_void_p * world)(*[o.data_ptr() for o in outs])
    assert lib.nrx_gather_inbox_place(tp, tr, F, ft, F, world, cap, c2.data_ptr(), send.data_ptr(), pos.data_ptr(), 8, po1, F * D, B, cols, None,
                                      torch.cuda.current_stream().cuda_stream) == -3          # NRX_ERR_UNSUPPORTED: rows of 8 floats


@pytest.mark.parametrize("world,lens,tables,cap,hi", [(1, [100], [0], 100, 50), (2, [63, 1, 300], [0, 1, 0], 256, 40),
                                                      (8, [4096, 4096, 5000, 17], [0, 1, 2, 1], 2048, 3000),
                                                      (3, [0, 777, 0, 2050], [1, 0, 0, 1], 1024, 1 << 20), (4, [5000], [0], 64, 100000)])
@pytest.mark.parametrize("dtype", [torch.int64, torch.int32])
def test_route_ids_dedup_bit_exact_vs_oracle(world, lens, tables, cap, hi, dtype):
    """Per-destination de-duplication (nrx_route_ids_dedup): unique (owner, table, row) lists, the slot of every lookup
    (duplicates share one), per-(owner, table) counts and the overflow word, all equal to the definition; small id
    ranges = heavy duplication, the last case overflows its blocks."""
    rng = np.random.default_rng(sum(lens) + world)
    arrays = [rng.integers(0, hi, n) for n in lens]
    nt = max(tables) + 1
    lrows = [(hi + world - 1) // world + 1] * nt
    send, slot, c2, over = ops.route_ids_dedup([torch.from_numpy(a).to(DEV).to(dtype) for a in arrays], tables, lrows, world, cap)
    r_send, r_slot, r_c2, r_worst = R.route_ids_dedup(arrays, tables, lrows, world, cap)
    assert np.array_equal(c2.cpu().numpy(), r_c2) and int(over.item()) == r_worst
    assert np.array_equal(slot.cpu().numpy(), r_slot)
    valid = r_send >= 0
    assert np.array_equal(send.cpu().numpy()[valid], r_send[valid])
    # un-permute: the row a lookup gets back is the row it asked for
    if r_worst <= cap and sum(lens):
        ids = np.concatenate(arrays)
        sl = slot.cpu().numpy()
        got_local = send.cpu().numpy()[sl]
        assert np.array_equal(got_local, ids // world) and np.array_equal(sl // cap, ids % world)


@pytest.mark.parametrize("sort", ["segmented", "segmented-bins", "rocprim"])
def test_dedup_and_unique_inverse_under_every_sort_at_multi_tile_sizes(sort, monkeypatch):
    """nrx_route_ids_dedup and nrx_unique_inverse sort with the planner's own tile kernels (seg_sort_generic: one segment, every scan path --
    <= 32 tiles direct, chunked, and the per-segment bin scan forced by NRX_PLAN_SORT=segmented-bins) or, NRX_PLAN_SORT=rocprim, the library
    sort: the same results, bit for bit equal to the definitions (oracle/ref_np.py: route_ids_dedup; np.unique)."""
    monkeypatch.setenv("NRX_PLAN_SORT", sort)
    rng = np.random.default_rng(99)
    for world, lens, tables, hi in [(8, [70000, 70000, 30001], [0, 1, 0], 90000), (4, [300000], [0], 1 << 19), (2, [4096 * 33 + 5, 17], [1, 0], 3000)]:
        arrays = [rng.integers(0, hi, n) for n in lens]
        nt = max(tables) + 1
        lrows = [(hi + world - 1) // world + 1] * nt
        cap = sum(lens)
        send, slot, c2, over = ops.route_ids_dedup([torch.from_numpy(a).to(DEV) for a in arrays], tables, lrows, world, cap)
        r_send, r_slot, r_c2, r_worst = R.route_ids_dedup(arrays, tables, lrows, world, cap)
        assert np.array_equal(c2.cpu().numpy(), r_c2) and int(over.item()) == r_worst
        assert np.array_equal(slot.cpu().numpy(), r_slot)
        valid = r_send >= 0
        assert np.array_equal(send.cpu().numpy()[valid], r_send[valid])
    for a in (rng.integers(-2 ** 62, 2 ** 62, 150001), rng.integers(0, 5000, 4096 * 40 + 3), rng.integers(-7, 7, 4097)):
        u, inv = ops.unique_inverse(torch.from_numpy(a).to(DEV))
        ru, rinv = np.unique(a, return_inverse=True)
        assert np.array_equal(u.cpu().numpy(), ru) and np.array_equal(inv.cpu().numpy(), rinv.reshape(a.shape))


def test_unique_inverse_matches_numpy():
    """nrx_unique_inverse == np.unique(return_inverse=True): int64 incl. negatives and extremes, int32, empty, all-equal."""
    rng = np.random.default_rng(5)
    cases = [rng.integers(-50, 50, 10000), rng.integers(-2 ** 62, 2 ** 62, 5000), np.array([7] * 3000), np.zeros(0, np.int64),
             np.array([np.iinfo(np.int64).min, np.iinfo(np.int64).max, 0, -1, np.iinfo(np.int64).max]), rng.integers(0, 1 << 20, 200003)]
    for a in cases:
        for dt in (torch.int64, torch.int32):
            if dt == torch.int32 and a.size and (a.min() < -2 ** 31 or a.max() >= 2 ** 31):
                continue
            u, inv = ops.unique_inverse(torch.from_numpy(a).to(DEV).to(dt))
            ru, rinv = np.unique(a, return_inverse=True)
            assert np.array_equal(u.cpu().numpy(), ru) and np.array_equal(inv.cpu().numpy(), rinv.reshape(a.shape))
    x = torch.randint(0, 100, (37, 5), device=DEV)
    u, inv = ops.unique_inverse(x)
    assert torch.equal(u[inv], x)


@pytest.mark.parametrize("world,B,Ls,cap,dtype", [(1, 50, [7], 400, torch.int64), (2, 300, [9, 4], 2048, torch.int64),
                                                  (3, 1000, [50], 20000, torch.int32), (8, 513, [5, 5, 12], 2048, torch.int64),
                                                  (4, 200, [6], 64, torch.int64)])
def test_pooled_bag_channel_vs_oracle(world, B, Ls, cap, dtype):
    """Owner-side partial pooling: nrx_bag_norm_weights, nrx_route_bags (bit-exact vs the definition,
    incl. dropped zero-weight lookups, feature boundaries inside a chunk and an overflowing block = last case) and
    nrx_pool_inbox_fwd / _bwd on every owner's shard; the partials summed over owners reproduce
    array_feature_pooling (base_model.py:273-282) to the stated 1e-6."""
    rng = np.random.default_rng(world * 1000 + B)
    D, rows = 16, 211
    kinds = [("masked_mean", NRX_BAG_MASKED_MEAN), ("mean", NRX_BAG_MEAN), ("sum", NRX_BAG_SUM)]
    table = rng.standard_normal((rows, D)).astype(np.float32)
    table[0] = 0
    ids, masks, wn_ref, kk = [], [], [], []
    for f, L in enumerate(Ls):
        name, kind = kinds[f % 3]
        lens = rng.integers(0, L + 1, B)
        lens[0] = 0
        m = (np.arange(L)[None] < lens[:, None]).astype(np.float32)
        x = rng.integers(1, rows, (B, L)) * (m.astype(np.int64) if name != "mean" else 1)
        ids.append(x)
        masks.append(None if name == "mean" else m)
        wn_ref.append(R.bag_norm_weights(masks[-1], B, L, name))
        kk.append(kind)
    wn = [ops.bag_norm_weights(None if m is None else dev(m), B, L, k, DEV) for m, L, k in zip(masks, Ls, kk)]
    for a_, b_ in zip(wn, wn_ref):
        np.testing.assert_allclose(a_.cpu().numpy(), b_, rtol=1e-6, atol=0)
    send, tag, sw, c2, over = ops.route_bags([dev(x).to(dtype) for x in ids], wn, world, cap)
    r_send, r_tag, r_sw, r_c2, r_worst = R.route_bags(ids, [w.cpu().numpy() for w in wn], world, cap)
    assert np.array_equal(c2.cpu().numpy(), r_c2) and int(over.item()) == r_worst
    valid = r_send >= 0
    assert np.array_equal(send.cpu().numpy()[valid], r_send[valid]) and np.array_equal(tag.cpu().numpy()[valid], r_tag[valid])
    assert np.array_equal(sw.cpu().numpy()[valid], r_sw[valid])
    if r_worst > cap:
        return                                           # overflow detected: the engine redoes the step exactly
    # every owner pools its block (this process plays all owners of ONE source): inbox block 0 = the block sent to o
    nf = len(Ls)
    total = np.zeros((nf * B, D), np.float64)
    gtab = torch.zeros((rows, D), device=DEV)
    up = rng.standard_normal((nf * B, D)).astype(np.float32)
    for o in range(world):
        shard = table[o::world]
        recv = np.zeros((world, nf), np.int64)
        recv[0] = r_c2[o]
        ib = torch.full((world * cap,), 10 ** 9, dtype=torch.int32, device=DEV)
        it = torch.full((world * cap,), -5, dtype=torch.int32, device=DEV)
        iw = torch.zeros(world * cap, device=DEV)
        ib[:cap], it[:cap], iw[:cap] = send[o * cap:(o + 1) * cap], tag[o * cap:(o + 1) * cap], sw[o * cap:(o + 1) * cap]
        status = torch.zeros(4, dtype=torch.int32, device=DEV)
        part = ops.pool_inbox([dev(shard)], [0] * nf, B, world, cap, dev(recv), ib, it, iw, status)
        want = R.pool_inbox([shard], [0] * nf, B, world, cap, recv, ib.cpu().numpy(), it.cpu().numpy(), iw.cpu().numpy(), D)
        np.testing.assert_allclose(part.cpu().numpy(), want, rtol=1e-6, atol=1e-6)
        assert status[0].item() == 0 and torch.all(part[1:] == 0)
        total += part[0].cpu().numpy()
        gshard = torch.zeros_like(dev(shard))
        gp = torch.zeros((world, nf * B, D), device=DEV)
        gp[0] = dev(up)
        ops.pool_inbox_bwd([gshard], [0] * nf, B, world, cap, dev(recv), ib, it, iw, gp, skip_row0=(o == 0))
        gtab[o::world] += gshard
    for f, (L, (name, _)) in enumerate(zip(Ls, [kinds[f % 3] for f in range(nf)])):
        ref = R.array_pool(table[ids[f]], masks[f]) if name != "sum" else (table[ids[f]] * masks[f][..., None]).sum(1)
        np.testing.assert_allclose(total[f * B:(f + 1) * B], ref, rtol=1e-6, atol=1e-6)
    # backward: d/dtable of sum(up * pooled) in fp64
    t64 = torch.from_numpy(table).double().requires_grad_(True)
    loss = 0
    for f, L in enumerate(Ls):
        w = torch.from_numpy(wn_ref[f]).double()
        loss = loss + ((t64[torch.from_numpy(ids[f])] * w[..., None]).sum(1) * torch.from_numpy(up[f * B:(f + 1) * B]).double()).sum()
    loss.backward()
    g_ref = t64.grad.clone()
    g_ref[0] = 0
    torch.testing.assert_close(gtab.cpu().double(), g_ref, rtol=1e-4, atol=1e-4)


def test_inbox_gather_and_scatter_vs_oracle():
    rng = np.random.default_rng(8)
    world, cap, D = 3, 512, 16
    tabs = [rng.standard_normal((r, D)).astype(np.float32) for r in (40, 300)]
    feat_table = [0, 1, 0]
    recv2d = rng.integers(0, 170, (world, 3))
    recv2d[1] = [0, 0, 0]
    inbox = np.full(world * cap, 10 ** 9, np.int64)                      # garbage past the valid prefixes
    for s in range(world):
        j = 0
        for f in range(3):
            n = recv2d[s, f]
            inbox[s * cap + j: s * cap + j + n] = rng.integers(0, tabs[feat_table[f]].shape[0], n)
            j += n
    status = torch.zeros(4, dtype=torch.int32, device=DEV)
    out = ops.gather_inbox([dev(t) for t in tabs], feat_table, world, cap, dev(recv2d), dev(inbox), status)
    want = R.gather_inbox(tabs, feat_table, world, cap, recv2d, inbox, D)
    got = out.cpu().numpy()
    for s in range(world):
        n = recv2d[s].sum()
        assert np.array_equal(got[s * cap: s * cap + n], want[s * cap: s * cap + n])
    assert status[0].item() == 0
    g_rows = rng.standard_normal((world * cap, D)).astype(np.float32)
    grads = [torch.zeros_like(dev(t)) for t in tabs]
    ops.scatter_add_inbox(grads, feat_table, world, cap, dev(recv2d), dev(inbox), dev(g_rows), True)
    ref = [np.zeros_like(t) for t in tabs]
    for s in range(world):
        j = 0
        for f in range(3):
            n = recv2d[s, f]
            rows = inbox[s * cap + j: s * cap + j + n]
            np.add.at(ref[feat_table[f]], rows, g_rows[s * cap + j: s * cap + j + n])
            j += n
    for g, r in zip(grads, ref):
        r[0] = 0                                                          # skip_row0
        np.testing.assert_allclose(g.cpu().numpy(), r, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", ["mixed_dims", "odd_dims", "bags", "bag_long_odd", "tower_large"])
def test_sorted_sparse_backward_matches_dense_and_is_deterministic(case):
    """sparse_grad=True: sorted segmented reduction -> COO grads.  Densified they equal the oracle's dense
    grads (tighter than the atomic path: a fixed summation order), two runs are bit-identical, and the
    padding row carries an explicit zero."""
    rng = np.random.default_rng(78)
    B = 4096 if case == "tower_large" else 300
    space, tables, batch = _rand_case(rng, B, GENERIC_CASES[case])
    names = set(tables) | space.dense
    up = None
    runs = []
    for rep in range(3):
        # third run: the sync-free form (worst-case launch, unique count read on the device) must give the same bits
        ops.SPARSE_BWD_SYNC_FREE = rep == 2
        plan, tt, inputs, weights, tn = build_plan(space, tables, batch, names)
        out = ops.embed_apply(plan, tt, inputs, weights, sparse_grad=True)[0]
        if up is None:
            up = rng.standard_normal(tuple(out.shape)).astype(np.float32)
        (out * dev(up)).sum().backward()
        assert all(t.grad.is_sparse for t in tt)
        runs.append([t.grad.coalesce() for t in tt])
    ops.SPARSE_BWD_SYNC_FREE = False
    for a, b, c in zip(*runs):
        assert torch.equal(a.indices(), b.indices()) and torch.equal(a.values(), b.values())      # bit-reproducible
        assert torch.equal(a.indices(), c.indices()) and torch.equal(a.values(), c.values())
    _, dims, _, used = R.embed_concat_ex(space, tables, batch, names)
    col = 0
    want = {n: np.zeros_like(tables[n]) for n in tables}
    for fname, d in zip(used, dims):
        u = up[:, col:col + d]
        col += d
        if fname in space.dense:
            continue
        if fname in space.array:
            rows_up = R.array_pool_bwd(tables[fname][batch[fname]], batch.get(fname + "_mask"), u)
            want[fname] += R.embedding_grad_dense(batch[fname], rows_up, tables[fname].shape[0])
        else:
            want[fname] += R.embedding_grad_dense(batch[fname], u, tables[fname].shape[0])
    for g, name in zip(runs[0], tn):
        dense = g.to_dense().cpu().numpy()
        np.testing.assert_allclose(dense, want[name], rtol=1e-5, atol=1e-6)
        assert np.all(dense[0] == 0)
        assert g.values().shape[0] <= B * 400 and g.indices().min().item() >= 0


@pytest.mark.parametrize("B,dims,NL", [(1, [16], 1), (300, [64] * 5, 2), (257, [32, 32, 16, 16, 16], 3), (100, [256] * 8, 1),
                                       (77, [128, 4, 4, 64], 8), (65, [16] * 26, 2), (1000, [32] * 8, 3), (129, [64] * 2, 1),
                                       (513, [32] * 3, 4)])
def test_fused_gather_cross_matches_two_launches(B, dims, NL):
    """One launch vs gather + cross: the gathered half is bit-exact always; the cross half is bit-identical when
    the one-wave-per-sample kernel runs (same lane layout and reduction tree as dcn_v1_fwd_kernel) and equal to fp32
    rounding when the grouped kernel runs (uniform 32/64-wide rows, <= 8 features: its dot product is reduced
    inside an 8/16-lane group, a different summation tree)."""
    rng = np.random.default_rng(B + len(dims))
    tables = [rng.standard_normal((40 + 3 * i, d)).astype(np.float32) for i, d in enumerate(dims)]
    ids = [rng.integers(0, t.shape[0], B) for t in tables]
    W = sum(dims)
    cols = np.concatenate([[0], np.cumsum(dims)])
    plan = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, d, 0, int(cols[i])) for i, d in enumerate(dims)], out_width=W)
    tt = [dev(t) for t in tables]
    ii = [dev(x) for x in ids]
    w = dev((rng.standard_normal((NL, W)) / np.sqrt(W)).astype(np.float32))
    b = dev((rng.standard_normal((NL, W)) * 0.1).astype(np.float32))
    fused = ops.embed_dcn_v1(plan, tt, ii, w, b)
    buf = ops.embed_apply(plan, tt, ii, [None] * len(dims), out_ld=2 * W)[0]
    two = ops.dcn_v1_cat_(buf, w, b)
    grouped = len(set(dims)) == 1 and dims[0] in (32, 64) and 2 <= len(dims) <= 8
    if grouped:
        assert torch.equal(fused[:, :W], two[:, :W])
        torch.testing.assert_close(fused[:, W:], two[:, W:], rtol=2e-5, atol=2e-6 * max(1.0, two.abs().max().item()))
    else:
        assert torch.equal(fused, two)                                     # same layout, same reduction order
    x = np.concatenate([t[i] for t, i in zip(tables, ids)], axis=1)
    assert np.array_equal(fused[:, :W].cpu().numpy(), x)                   # gather + concat: bit-exact vs numpy
    ref = R.dcn_v1(x, w.cpu().numpy(), b.cpu().numpy())
    np.testing.assert_allclose(fused[:, W:].cpu().numpy(), ref, rtol=1e-5, atol=2e-6 * max(1.0, np.abs(ref).max()))
    with pytest.raises(IndexError):
        bad = [t.clone() for t in ii]
        bad[0][0] = 10 ** 6
        ops.embed_dcn_v1(plan, tt, bad, w, b)


def test_fused_gather_cross_rejects_what_it_does_not_cover():
    t = torch.randn(10, 6, device=DEV)                                        # dim 6: not a multiple of 4
    plan = ops.EmbedPlan([ops.Slot("a", NRX_SPARSE, 0, 6)], out_width=6)
    with pytest.raises(ops.FusedUnsupported):
        ops.embed_dcn_v1(plan, [t], [torch.tensor([1, 2], device=DEV)], torch.zeros(1, 6, device=DEV), torch.zeros(1, 6, device=DEV))
    t16 = torch.randn(10, 16, device=DEV)
    plan = ops.EmbedPlan([ops.Slot("h", NRX_BAG_MEAN, 0, 16, 3)], out_width=16)
    with pytest.raises(ops.FusedUnsupported):
        ops.embed_dcn_v1(plan, [t16], [torch.zeros(2, 3, dtype=torch.long, device=DEV)], torch.zeros(1, 16, device=DEV), torch.zeros(1, 16, device=DEV))


def test_dcn_v2_property_random_shapes_value_exact():
    """Hypothesis sweep over batch / width (odd, tiny, non-multiples of the 128x64 tile and of the 32-deep slab): the
    matrix-core layer equals the C oracle's fp32 fma chain value for value."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    from oracle import ref_c

    @settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(B=st.integers(1, 400), D=st.integers(1, 200), NL=st.integers(1, 2), relu=st.booleans(), seed=st.integers(0, 10 ** 6))
    def run(B, D, NL, relu, seed):
        rng = np.random.default_rng(seed)
        x = rng.standard_normal((B, D)).astype(np.float32)
        W = (rng.standard_normal((NL, D, D)) / np.sqrt(D)).astype(np.float32)
        b = (rng.standard_normal((NL, D)) * 0.1).astype(np.float32)
        out = ops.dcn_v2(dev(x), dev(W), dev(b), relu=relu)
        assert np.array_equal(out.detach().cpu().numpy(), ref_c.dcn_v2(x, W, b, relu=relu))

    run()


@pytest.mark.parametrize("B,F,D,wide_every", [(1, 3, 16, 1), (300, 27, 32, 3), (257, 14, 16, 2), (1000, 40, 32, 4), (129, 9, 64, 1),
                                              (65, 13, 32, 13), (4096, 26, 16, 5)])
@pytest.mark.parametrize("idx_dtype", [np.int64, np.int32])
def test_uniform_wide_split_bit_exact_vs_oracle(B, F, D, wide_every, idx_dtype):
    """Wide&Deep split on uniform features (embed_fwd_uniform_wide): every `wide_every`-th feature sends column 0 to
    the wide tensor and columns 1.. to the deep concat, so the deep row loses 16-byte alignment -- copies stay
    bit-exact vs the oracle (widedeep/model.py:53-69), the dense backward equals the oracle's too, out-of-range ids
    are reported."""
    rng = np.random.default_rng(B + 31 * F + D + wide_every)
    space, tables, batch = _rand_case(rng, B, [(NRX_SPARSE, 40 + 5 * i, D, 0) for i in range(F)], idx_dtype)
    names = sorted(tables)
    wide = set(names[::wide_every])
    feats, dims, used = R.embed_concat(space, tables, batch, set(tables))
    want_wide, want_deep = R.wide_split(feats, dims, used, wide)
    plan, tt, inputs, weights, _ = build_plan(space, tables, batch, set(tables), wide_names=wide)
    deep, wd, _ = ops.embed_apply(plan, tt, inputs, weights)
    assert np.array_equal(deep.detach().cpu().numpy(), want_deep)
    assert np.array_equal(wd.detach().cpu().numpy(), want_wide)
    # backward through both outputs == scatter-add of the re-joined upstream rows
    up_d = rng.standard_normal(want_deep.shape).astype(np.float32)
    up_w = rng.standard_normal(want_wide.shape).astype(np.float32)
    ((deep * dev(up_d)).sum() + (wd * dev(up_w)).sum()).backward()
    col, wc = 0, 0
    for n, t in zip(names, tt):
        if n in wide:
            g_rows = np.concatenate([up_w[:, wc:wc + 1], up_d[:, col:col + D - 1]], axis=1)
            col, wc = col + D - 1, wc + 1
        else:
            g_rows = up_d[:, col:col + D]
            col += D
        want = R.embedding_grad_dense(batch[n], g_rows, tables[n].shape[0])
        np.testing.assert_allclose(t.grad.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    # the deterministic row-sparse backward takes the same routing (dword-aligned 16-byte reads of the shifted deep columns,
    # column 0 from the wide gradient): densified it equals the dense result, and two runs are bit-identical
    dense = [t.grad.clone() for t in tt]
    runs = []
    for _ in range(2):
        t2 = [t.detach().clone().requires_grad_() for t in tt]
        deep2, wd2, _ = ops.embed_apply(plan, t2, inputs, weights, sparse_grad=True)
        ((deep2 * dev(up_d)).sum() + (wd2 * dev(up_w)).sum()).backward()
        runs.append([t.grad.coalesce() for t in t2])
    for g0, g1, gd in zip(runs[0], runs[1], dense):
        assert torch.equal(g0.indices(), g1.indices()) and torch.equal(g0.values(), g1.values())
        torch.testing.assert_close(g0.to_dense(), gd, rtol=1e-5, atol=1e-5)
    with pytest.raises(IndexError):
        bad = [x.clone() for x in inputs]
        bad[-1][0] = 10 ** 6
        ops.embed_apply(plan, [t.detach() for t in tt], bad, weights)


@pytest.mark.parametrize("B,F,D,wide_every", [(1, 9, 16, 1), (300, 27, 32, 3), (257, 14, 16, 2), (2049, 40, 32, 4), (129, 9, 64, 1), (63, 12, 32, 12),
                                              (500, 11, 16, 5), (1000, 40, 32, 1)])
@pytest.mark.parametrize("aligned", ["1", "0"])
def test_wide_split_aligned_chunk_stores_bit_exact_vs_oracle(B, F, D, wide_every, aligned, monkeypatch):
    """WideDeep.get_inp_embedding's column routing (src/model/sort/widedeep/model.py:53-69) with the deep row's stride padded to a multiple of 4
    floats (what the WideDeep model asks for): the ring-form split kernel then writes the row as aligned 16-byte chunks assembled across lanes
    (nrx_embed_wide.hip: WideAl).  Bit-exact against the oracle for every pattern of wide features (every 1st ... every 12th feature: 0-3
    floats pending at every feature boundary), the dword-store form (NRX_WIDE_ALIGNED=0) beside it; the pad columns are never written."""
    monkeypatch.setenv("NRX_WIDE_ALIGNED", aligned)
    rng = np.random.default_rng(B + 31 * F + D + wide_every)
    space, tables, batch = _rand_case(rng, B, [(NRX_SPARSE, 40 + 3 * i, D, 0) for i in range(F)])
    names = sorted(tables)
    wide_names = set(names[wide_every - 1::wide_every])
    full, dims, used = R.embed_concat(space, tables, batch, set(names))
    want_wide, want_deep = R.wide_split(full, dims, used, wide_names)          # the oracle's restatement of widedeep/model.py:53-69
    plan, tt, inputs, weights, _ = build_plan(space, tables, batch, set(names), wide_names=wide_names)
    W = plan.out_width
    ld = (W + 3) // 4 * 4 + 4                               # padded stride (+ one whole spare chunk: must stay untouched)
    out, wide, _ = ops.embed_apply(plan, [t.detach() for t in tt], inputs, weights, out_ld=ld)
    assert out.shape == (B, ld)
    assert np.array_equal(out[:, :W].cpu().numpy(), want_deep)
    assert np.array_equal(wide.cpu().numpy(), want_wide)
    # the pad columns: run into a sentinel-filled buffer through the bound call and check they are intact
    buf = torch.full((B, ld), 777.25, device=DEV)
    ops.PreparedEmbed(plan, [t.detach() for t in tt], inputs, weights, out_ld=ld, out=buf).run()
    torch.cuda.synchronize()
    assert np.array_equal(buf[:, :W].cpu().numpy(), want_deep)
    assert np.all(buf[:, W:].cpu().numpy() == 777.25)
    # narrow=True: the [B, W] view of the padded buffer, differentiable
    tt2 = [t.detach().clone().requires_grad_() for t in tt]
    deep, wide2, _ = ops.embed_apply(plan, tt2, inputs, weights, out_ld=ld, narrow=True)
    assert deep.shape == (B, W) and deep.stride(0) == ld
    assert np.array_equal(deep.detach().cpu().numpy(), want_deep)
    up = torch.from_numpy(rng.standard_normal((B, W)).astype(np.float32)).to(DEV)
    upw = torch.from_numpy(rng.standard_normal(tuple(wide2.shape)).astype(np.float32)).to(DEV)
    ((deep * up).sum() + (wide2 * upw).sum()).backward()
    tt3 = [t.detach().clone().requires_grad_() for t in tt]
    deep3, wide3, _ = ops.embed_apply(plan, tt3, inputs, weights)           # the unpadded launch: same gradients
    ((deep3 * up).sum() + (wide3 * upw).sum()).backward()
    for a_, b_ in zip(tt2, tt3):
        torch.testing.assert_close(a_.grad, b_.grad, rtol=1e-5, atol=1e-5)  # float atomics order (small batch: the scatter kernel)


@pytest.mark.parametrize("B", [3000, 70000])
@pytest.mark.parametrize("idx", [torch.int64, torch.int32])
def test_dense_value_mid_order_keeps_later_features_on_the_ring_kernel(B, idx, monkeypatch):
    """A dense value (one column) in the middle of the sorted feature order makes every later feature's first column a
    non-multiple of 4 floats.  Those features now take the ring kernel's dword-aligned store form (UniformArgs::unal) instead of
    the generic kernel; the concat (base_model.py:284-308) must be the same bits either way: checked against the numpy oracle's
    gather / concat (verbatim copies) for widths 16 / 32 / 64 around two dense values, also with an `out` whose row stride and
    base are not 16-byte aligned."""
    monkeypatch.setenv("NRX_SPLIT_MIN_LOOKUPS", "1")            # take the per-width uniform launches at any batch
    rng = np.random.default_rng(B)
    dims = [16, 32, 16, 64, 32, 16, 16, 64, 32, 32, 16, 64]
    slots, tables, inputs, col = [], [], [], 0
    for i, d in enumerate(dims):
        if i in (3, 8):                                         # dense values in the middle of the order
            slots.append(ops.Slot(f"d{i}", NRX_DENSE, -1, 1, 0, col))
            inputs.append(dev(rng.random(B).astype(np.float32)))
            col += 1
        rows = 500 + 37 * i
        tables.append(dev(rng.standard_normal((rows, d)).astype(np.float32)))
        slots.append(ops.Slot(f"f{i:02d}", NRX_SPARSE, len(tables) - 1, d, 0, col))
        inputs.append(dev(rng.integers(0, rows, B)).to(idx))
        col += d
    plan = ops.EmbedPlan(slots, out_width=col)
    out = ops.embed_apply(plan, tables, inputs, [None] * len(slots), index_check="sync")[0]
    out_pad = ops.embed_apply(plan, tables, inputs, [None] * len(slots), out_ld=col + 3, index_check="sync")[0]      # odd row stride
    ref = np.zeros((B, col), np.float32)
    for s, x in zip(slots, inputs):
        if s.kind == NRX_DENSE:
            ref[:, s.out_col] = x.cpu().numpy()
        else:
            ref[:, s.out_col:s.out_col + s.dim] = tables[s.table].cpu().numpy()[x.cpu().numpy()]
    assert np.array_equal(out.cpu().numpy(), ref)
    assert np.array_equal(out_pad.cpu().numpy()[:, :col], ref)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 1000, 65536 + 3])
def test_fm_head_forward_and_backward_vs_float64(B):
    """ops.fm_head = the last line of FMModel.forward, sigmoid(bias + first + second) (src/model/sort/fm/model.py:25-26), and its autograd: one
    launch each way.  Against torch in float64 (forward rtol 1e-6; gradients rtol 1e-5), with a per-sample upstream gradient and with the expanded
    scalar a `.sum()` hands down (read once, not materialised); the bias gradient is summed in a fixed order: same bits run to run."""
    from news_recsys_amd import ops
    gen = torch.Generator(device="cuda:0").manual_seed(B)
    logit = (torch.randn(B, device="cuda:0", generator=gen) * 3).requires_grad_()
    bias = torch.tensor([0.37], device="cuda:0").requires_grad_()
    up = torch.randn(B, 1, device="cuda:0", generator=gen)
    l64, b64 = logit.detach().double().requires_grad_(), bias.detach().double().requires_grad_()
    ref = torch.sigmoid(b64 + l64.unsqueeze(1))
    out = ops.fm_head(logit, bias)
    assert out.shape == (B, 1)
    torch.testing.assert_close(out.double(), ref, rtol=1e-6, atol=1e-7)
    (ref * up.double()).sum().backward()
    (out * up).sum().backward()
    # (p (1 - p) from the SAVED fp32 p, as torch's sigmoid_backward forms it: where p rounds towards 1 the factor 1 - p carries p's rounding, an
    # absolute error of ~1e-7 on a gradient of ~1e-3 -- hence the absolute term)
    torch.testing.assert_close(logit.grad.double(), l64.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bias.grad.double(), b64.grad, rtol=1e-5, atol=1e-6 * max(1.0, float(up.abs().sum()) ** 0.5))
    gb = bias.grad.clone()
    for _ in range(3):                                   # the expanded-scalar path, and bit-reproducibility of the bias sum
        logit.grad = bias.grad = None
        (ops.fm_head(logit, bias) * up).sum().backward()
        assert torch.equal(bias.grad.view(torch.int32), gb.view(torch.int32))
    logit.grad = bias.grad = None
    l64.grad = b64.grad = None
    ops.fm_head(logit, bias).sum().backward()
    torch.sigmoid(b64 + l64.unsqueeze(1)).sum().backward()
    torch.testing.assert_close(logit.grad.double(), l64.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bias.grad.double(), b64.grad, rtol=1e-5, atol=1e-6 * B ** 0.5 + 1e-5)
