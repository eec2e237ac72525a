"""CSR-form bag features (NRX_FEAT_BAG_CSR; SURVEY 8b "CSR-input bag variant"): the fused launch reads an array feature
as ids [nnz] + offsets [B + 1] instead of the padded ids + mask that DataReader.__getitem__ builds per sample
(src/dataset/DataReader/data_reader.py:96-109).  Definition: oracle.ref_np.csr_bag_to_padded followed by the reference's
masked mean pooling (src/model/BaseModel/base_model.py:273-282).

Bars: CSR launch == padded launch BIT FOR BIT (same kernel, same summation order); vs the numpy oracle rtol/atol 1e-6
(fp32 sum order; 1e-5 for the un-normalised NRX_BAG_SUM); dense table gradients rtol 1e-4 / atol 1e-5 (float atomics); row-sparse gradients equal to the padded launch's."""
import numpy as np
import pytest
import torch

from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM, NRX_FEAT_BAG_CSR, NRX_SPARSE
from oracle import ref_np as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def rand_csr(rng, B, L, rows, dtype, long_tail=True):
    lens = rng.integers(0, L + 1, B)
    if long_tail:
        lens[rng.random(B) < 0.1] = L + rng.integers(1, 9)        # longer than L: truncated like DataReader does
    lens[0] = 0
    if B > 1:
        lens[-1] = 0                                              # empty bags at both ends
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    values = rng.integers(0, rows, offsets[-1]).astype(dtype)     # id 0 (the padding row) may appear as a real entry
    return values, offsets


def plans(kind, D, L, extra_sparse=True):
    slots_c = [ops.Slot("hist", kind, 0, D, L, 0, flags=NRX_FEAT_BAG_CSR)]
    slots_p = [ops.Slot("hist", kind, 0, D, L, 0)]
    if extra_sparse:
        slots_c.append(ops.Slot("uid", NRX_SPARSE, 1, D, 0, D))
        slots_p.append(ops.Slot("uid", NRX_SPARSE, 1, D, 0, D))
    w = D * len(slots_c)
    return ops.EmbedPlan(slots_c, out_width=w), ops.EmbedPlan(slots_p, out_width=w)


@pytest.mark.parametrize("kind", [NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM])
@pytest.mark.parametrize("B,L,D,dtype", [(1, 5, 16, np.int64), (257, 50, 16, np.int32), (1000, 7, 64, np.int64), (300, 130, 8, np.int64),
                                         (65, 33, 6, np.int32),
                                         # dims <= 4 put 256 samples in a block: the offsets staging needs 257 entries
                                         (513, 5, 1, np.int64), (700, 9, 4, np.int32), (256, 3, 1, np.int64), (257, 6, 2, np.int64)])
def test_csr_bag_forward_equals_padded_and_oracle(kind, B, L, D, dtype):
    rng = np.random.default_rng(B * 131 + L + D + kind)
    rows = 200
    t = rng.standard_normal((rows, D)).astype(np.float32)
    t[0] = 0 if kind != NRX_BAG_MEAN else t[0]                   # NRX_BAG_MEAN reads row 0 at padded positions: leave it non-zero
    u = rng.standard_normal((50, D)).astype(np.float32)
    values, offsets = rand_csr(rng, B, L, rows, dtype)
    uid = rng.integers(0, 50, B)
    ids, mask = R.csr_bag_to_padded(values, offsets, L)
    plan_c, plan_p = plans(kind, D, L)
    tt = [dev(t), dev(u)]
    out_c = ops.embed_apply(plan_c, tt, [dev(values), dev(uid)], [dev(offsets), None])[0]
    w_p = None if kind == NRX_BAG_MEAN else dev(mask)
    out_p = ops.embed_apply(plan_p, tt, [dev(ids), dev(uid)], [w_p, None])[0]
    assert torch.equal(out_c, out_p)
    e = t[ids].astype(np.float64)
    if kind == NRX_BAG_MASKED_MEAN:
        ref = R.array_pool(t[ids], mask)
    elif kind == NRX_BAG_MEAN:
        ref = R.array_pool(t[ids], None)
    else:
        ref = (e * mask[:, :, None]).sum(1)
    tol = 1e-5 if kind == NRX_BAG_SUM else 1e-6                  # plain sums of up to L terms cancel: same bar as test_bag_sum_kind_and_weights
    np.testing.assert_allclose(out_c[:, :D].cpu().numpy(), ref, rtol=tol, atol=tol)
    assert np.array_equal(out_c[:, D:].cpu().numpy(), u[uid])


def test_csr_bag_all_empty_and_empty_values_tensor():
    D, L, B = 16, 9, 33
    t = torch.randn(20, D, device=DEV)
    plan_c, _ = plans(NRX_BAG_MASKED_MEAN, D, L, extra_sparse=False)
    off = torch.zeros(B + 1, dtype=torch.int64, device=DEV)
    out = ops.embed_apply(plan_c, [t], [torch.zeros(0, dtype=torch.int64, device=DEV)], [off])[0]
    assert out.shape == (B, D) and torch.all(out == 0)           # 0 / (0 + 1e-8): exact zeros, like an all-masked padded bag


def test_csr_bag_out_of_range_id_raises():
    D, L = 16, 4
    t = torch.randn(20, D, device=DEV)
    plan_c, _ = plans(NRX_BAG_MASKED_MEAN, D, L, extra_sparse=False)
    vals = torch.tensor([1, 2, 99, 3], device=DEV)
    off = torch.tensor([0, 2, 4], device=DEV)
    with pytest.raises(IndexError):
        ops.embed_apply(plan_c, [t], [vals], [off], index_check="sync")
    # ... but an out-of-range id BEYOND the first L entries is never read (DataReader cut it off)
    vals = torch.tensor([1, 2, 3, 4, 99, 5], device=DEV)
    off = torch.tensor([0, 5, 6], device=DEV)
    ops.embed_apply(plan_c, [t], [vals], [off], index_check="sync")


def test_csr_bag_argument_errors():
    D, L = 16, 4
    t = torch.randn(20, D, device=DEV)
    plan_c, _ = plans(NRX_BAG_MASKED_MEAN, D, L, extra_sparse=False)
    vals = torch.tensor([1, 2, 3], device=DEV)
    with pytest.raises(ValueError):
        ops.embed_apply(plan_c, [t], [vals], [None])                                            # no offsets
    with pytest.raises(ValueError):
        ops.embed_apply(plan_c, [t], [vals], [torch.tensor([0.0, 3.0], device=DEV)])            # offsets must be int64
    bad = ops.EmbedPlan([ops.Slot("a", NRX_SPARSE, 0, D, 0, 0, flags=NRX_FEAT_BAG_CSR)], out_width=D)
    with pytest.raises(ValueError):
        ops.embed_apply(bad, [t], [vals], [torch.tensor([0, 3], device=DEV)])


@pytest.mark.parametrize("sparse_grad", [False, True])
@pytest.mark.parametrize("kind", [NRX_BAG_MASKED_MEAN, NRX_BAG_SUM])
@pytest.mark.parametrize("B,D", [(400, 16), (600, 1), (513, 4)])        # D <= 4: 256 samples per block (257 staged offsets)
def test_csr_bag_backward_matches_padded(kind, sparse_grad, B, D):
    rng = np.random.default_rng(77 + kind + B)
    L, rows = 20, 120
    t0 = rng.standard_normal((rows, D)).astype(np.float32)
    u0 = rng.standard_normal((50, D)).astype(np.float32)
    values, offsets = rand_csr(rng, B, L, rows, np.int64)
    uid = rng.integers(0, 50, B)
    ids, mask = R.csr_bag_to_padded(values, offsets, L)
    g = dev(rng.standard_normal((B, 2 * D)).astype(np.float32))
    plan_c, plan_p = plans(kind, D, L)
    grads = []
    for plan, x, w in ((plan_c, dev(values), dev(offsets)), (plan_p, dev(ids), dev(mask))):
        tt = [dev(t0).requires_grad_(), dev(u0).requires_grad_()]
        out = ops.embed_apply(plan, tt, [x, dev(uid)], [w, None], sparse_grad=sparse_grad)[0]
        out.backward(g)
        grads.append([p.grad.to_dense() if p.grad.is_sparse else p.grad for p in tt])
    for a, b in zip(*grads):
        if sparse_grad:
            assert torch.equal(a, b)                                     # same planner input after the expansion: deterministic
        else:
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)       # float atomics: order differs between runs
    # and against the oracle's pooling backward (dense form)
    if kind == NRX_BAG_MASKED_MEAN:
        ge = R.array_pool_bwd(t0[ids], mask, g[:, :D].cpu().numpy())
        ref = np.zeros_like(t0, dtype=np.float64)
        np.add.at(ref, ids.reshape(-1), ge.reshape(-1, D))
        ref[0] = 0
        np.testing.assert_allclose(grads[0][0].cpu().numpy(), ref, rtol=1e-4, atol=1e-5)


def _csr_batch(batch, arrays):
    csr = dict(batch)
    for n in arrays:
        m = batch[f"{n}_mask"].cpu().numpy()
        ids = batch[n].cpu().numpy()
        lens = (m != 0).sum(1)
        assert all((m[b, :lens[b]] != 0).all() for b in range(len(lens)))          # DataReader masks are prefixes
        csr[n] = dev(np.concatenate([ids[b, :lens[b]] for b in range(len(lens))]))
        csr[f"{n}_offsets"] = dev(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64))
        del csr[f"{n}_mask"]
    return csr


@pytest.mark.parametrize("sparse_grad", [False, True])
def test_module_surface_takes_csr_batches(sparse_grad):
    """Deep-with-arrays golden (generated by the reference's own code): a batch that carries `name` [nnz] +
    `name_offsets` reproduces the reference's features, loss and embedding-table gradients."""
    import os
    from tests.conftest import CONFIGS, GOLDEN
    from news_recsys_amd.model.sort.deep.model import Deep
    g = dict(np.load(os.path.join(GOLDEN, "model_deep_array.npz"), allow_pickle=False))
    model = Deep(os.path.join(CONFIGS, "cf_array_small.yaml"))
    model.load_state_dict({k[len("param/"):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param/")}, strict=True)
    model = model.to(DEV)
    if sparse_grad:
        model.sparse_grad = True
    batch = {k[len("batch/"):]: dev(v) for k, v in g.items() if k.startswith("batch/")}
    csr = _csr_batch(batch, ["user_history", "user_click_cats"])
    names = model.user_feature_names | model.item_feature_names
    with torch.no_grad():
        fa, da, na = model.get_embeddings_from_batch(batch, names)
        fb, db, nb = model.get_embeddings_from_batch(csr, names)
    assert da == db and na == nb and torch.equal(fa, fb)
    np.testing.assert_allclose(fb.cpu().numpy(), g["out/features"], rtol=1e-6, atol=1e-6)
    loss = model.training_step(csr, 0)
    np.testing.assert_allclose(loss.item(), g["out/loss"], rtol=1e-5)
    loss.backward()
    for k, v in g.items():
        if k.startswith("grad/embedding_tables."):
            p = dict(model.named_parameters())[k[len("grad/"):]]
            got = p.grad.to_dense() if p.grad.is_sparse else p.grad
            np.testing.assert_allclose(got.cpu().numpy(), v, rtol=1e-4, atol=1e-6)
