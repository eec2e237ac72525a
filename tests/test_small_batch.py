"""Small batches (the reference trains with batch 512: src/model/sort/deep/train_cf_deep.yaml:48) take a one-block-per-sample
kernel (embed_fwd_small_kernel) instead of the lane-group walks tuned for B = 65536.  Bar: BIT-IDENTICAL to the big kernels --
checked by running the same samples once as a small batch and once as the head of a batch large enough for the big kernels
(samples are independent), for gather / concat, masked-mean / mean / sum pooling, dense values and the FM epilogue
(src/model/BaseModel/base_model.py:262-308, src/model/sort/fm/model.py:18-26,48-59), and against the numpy oracle."""
import numpy as np
import pytest
import torch

from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM, NRX_DENSE, NRX_SPARSE
from oracle import ref_np as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BIG = 4096          # > the small-batch limit (2048): served by the ring / generic kernels


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def _bits(t):
    return t.contiguous().view(torch.int32)


def _case(name, rng, B):
    """(plan, tables, inputs, weights) of BIG samples; the first B are the small batch."""
    if name == "c2_fm":                        # 26 x D = 16, FM epilogue (the C2 / train_cf_fm shape)
        F, D, rows = 26, 16, 5000
        slots = [ops.Slot(f"C{i:02d}", NRX_SPARSE, i, D, 0, i * D, fm_field=1) for i in range(F)]
        plan = ops.EmbedPlan(slots, out_width=F * D, use_fm=True)
        tables = [dev(rng.standard_normal((rows, D)).astype(np.float32)) for _ in range(F)]
        return plan, tables, [dev(rng.integers(0, rows, BIG)) for _ in range(F)], [None] * F
    if name == "c1_mixed":                     # train_cf_deep.yaml: dims 16 / 32 interleaved in sorted order
        dims, rows = [16, 32, 16, 16, 32], [18, 65239, 270, 18, 94058]
        slots, col = [], 0
        for i, d in enumerate(dims):
            slots.append(ops.Slot(f"f{i}", NRX_SPARSE, i, d, 0, col))
            col += d
        plan = ops.EmbedPlan(slots, out_width=col)
        tables = [dev(rng.standard_normal((r, d)).astype(np.float32)) for r, d in zip(rows, dims)]
        return plan, tables, [dev(rng.integers(0, r, BIG).astype(np.int32)) for r in rows], [None] * 5
    if name == "dssm_user":                    # user id + masked-mean history (L = 50) sharing the news table + a dense value
        D, L = 16, 50
        slots = [ops.Slot("age", NRX_DENSE, -1, 1, 0, 0), ops.Slot("user_history", NRX_BAG_MASKED_MEAN, 0, D, L, 1),
                 ops.Slot("user_id", NRX_SPARSE, 1, D, 0, 1 + D)]
        plan = ops.EmbedPlan(slots, out_width=1 + 2 * D)
        tables = [dev(rng.standard_normal((3000, D)).astype(np.float32)), dev(rng.standard_normal((9000, D)).astype(np.float32))]
        lens = rng.integers(0, L + 1, BIG)
        mask = (np.arange(L)[None] < lens[:, None]).astype(np.float32)
        hist = np.where(mask > 0, rng.integers(1, 3000, (BIG, L)), 0)
        return plan, tables, [dev(rng.random(BIG).astype(np.float32)), dev(hist), dev(rng.integers(0, 9000, BIG))], [None, dev(mask), None]
    if name == "bags_mean_sum":                # mean over L incl. padding, weighted sum, D = 64
        D, L = 64, 7
        slots = [ops.Slot("a", NRX_BAG_MEAN, 0, D, L, 0), ops.Slot("b", NRX_BAG_SUM, 0, D, L, D), ops.Slot("c", NRX_SPARSE, 0, D, 0, 2 * D)]
        plan = ops.EmbedPlan(slots, out_width=3 * D)
        tables = [dev(rng.standard_normal((400, D)).astype(np.float32))]
        w = rng.random((BIG, L)).astype(np.float32) * (rng.random((BIG, L)) < 0.7)
        ins = [dev(rng.integers(0, 400, (BIG, L))), dev(rng.integers(0, 400, (BIG, L))), dev(rng.integers(0, 400, BIG))]
        return plan, tables, ins, [None, dev(w), None]
    raise KeyError(name)


@pytest.mark.parametrize("B", [1, 64, 512, 2048])
@pytest.mark.parametrize("name", ["c2_fm", "c1_mixed", "dssm_user", "bags_mean_sum"])
def test_small_batch_kernel_is_bit_identical_to_the_big_kernels(name, B):
    rng = np.random.default_rng(len(name) + B)
    plan, tables, inputs, weights = _case(name, rng, B)
    with torch.no_grad():
        big = ops.embed_apply(plan, tables, inputs, weights, index_check="sync")
        small = ops.embed_apply(plan, tables, [x[:B].contiguous() for x in inputs],
                                [None if w is None else w[:B].contiguous() for w in weights], index_check="sync")
    for a, b in zip(big, small):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(_bits(a[:B]), _bits(b)), name


@pytest.mark.parametrize("B", [1, 63, 257, 2049, 4096])
@pytest.mark.parametrize("name", ["c2_fm", "c1_mixed", "dssm_user", "bags_mean_sum"])
def test_both_families_on_the_same_batch_through_the_limit_knob(name, B):
    """The same batch -- ragged last blocks on both sides of the default limit -- through nrx_set_small_batch_max(large) and (0): the
    one-block-per-sample kernel and the lane-group kernels must agree bit for bit (outputs, FM logits)."""
    from news_recsys_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(7 * len(name) + B)
    plan, tables, inputs, weights = _case(name, rng, B)
    ins = [x[:B].contiguous() for x in inputs]
    ws = [None if w is None else w[:B].contiguous() for w in weights]
    res = []
    prev = lib.nrx_set_small_batch_max(-1)
    try:
        for limit in (1 << 20, 0):
            lib.nrx_set_small_batch_max(limit)
            assert lib.nrx_set_small_batch_max(-1) == limit
            with torch.no_grad():
                res.append(ops.embed_apply(plan, tables, ins, ws, index_check="sync"))
    finally:
        lib.nrx_set_small_batch_max(prev)
    for a, b in zip(*res):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(_bits(a), _bits(b)), name


def test_small_batch_training_form_field_sums_and_gradients():
    """FM plan with gradients: the small kernel writes the same field sums (consumed by the backward) -- loss gradients of the
    small batch equal those of the same samples inside a big batch, within fp32 atomics order for the dense scatter."""
    rng = np.random.default_rng(9)
    B = 300
    plan, tables, inputs, weights = _case("c2_fm", rng, B)
    grads = []
    for n in (B, BIG):
        ts = [t.clone().requires_grad_() for t in tables]
        ids = [x[:n].contiguous() for x in inputs]
        out, _, fm = ops.embed_apply(plan, ts, ids, weights, sparse_grad=True)
        ((out[:B] * out[:B]).sum() + (fm[:B] * fm[:B]).sum()).backward()
        grads.append([t.grad.coalesce().to_dense() for t in ts])
    for a, b in zip(*grads):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)


def test_small_batch_out_of_range_id_raises_and_oracle_parity():
    rng = np.random.default_rng(3)
    plan, tables, inputs, weights = _case("dssm_user", rng, 128)
    B = 128
    ins = [x[:B].contiguous() for x in inputs]
    ws = [None if w is None else w[:B].contiguous() for w in weights]
    out = ops.embed_apply(plan, tables, ins, ws, index_check="sync")[0].cpu().numpy()
    hist, mask = ins[1].cpu().numpy(), ws[1].cpu().numpy()
    ref = R.array_pool(tables[0].cpu().numpy()[hist], mask)
    np.testing.assert_allclose(out[:, 1:17], ref, rtol=1e-6, atol=1e-6)
    assert np.array_equal(out[:, 17:], tables[1].cpu().numpy()[ins[2].cpu().numpy()])
    assert np.array_equal(out[:, 0], ins[0].cpu().numpy())
    bad = [x.clone() for x in ins]
    bad[2][5] = 10 ** 7
    with pytest.raises(IndexError):
        ops.embed_apply(plan, tables, bad, ws, index_check="sync")
