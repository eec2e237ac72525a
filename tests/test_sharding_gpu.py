"""GPU tests of the row-sharded engine with the product (HIP) backend in a single process (world=1:
the routing, owner-side segmented gather / scatter-add and the slot-addressed final launch all run;
the all-to-alls degenerate to copies).  Multi-rank behaviour is covered by tests/test_sharding_gloo.py."""
import os

import numpy as np
import pytest
import torch

from news_recsys_amd import ops, sharding
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_DENSE, NRX_SPARSE
from news_recsys_amd.model.sort.deep.model import Deep
from news_recsys_amd.model.sort.fm.model import FM
from news_recsys_amd.sharding import RowShardedEmbedding, ShardedFeature
from tests.conftest import CONFIGS, GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(B=777, fm=False):
    g = torch.Generator(device=DEV).manual_seed(5)
    rows = {"a": 1000, "b": 77, "c": 5000}
    dims = {"a": 16, "b": 16, "c": 32}
    tables = {n: torch.randn(rows[n], dims[n], device=DEV, generator=g) for n in rows}
    for t in tables.values():
        t[0] = 0
    feats = [ShardedFeature("f_a", NRX_SPARSE, "a", 16, fm=fm), ShardedFeature("f_b", NRX_SPARSE, "b", 16, fm=fm),
             ShardedFeature("f_hist", NRX_BAG_MASKED_MEAN, "a", 16, 9, fm=fm)]
    if not fm:
        feats += [ShardedFeature("f_c", NRX_SPARSE, "c", 32), ShardedFeature("f_cm", NRX_BAG_MEAN, "c", 32, 4),
                  ShardedFeature("f_d", NRX_DENSE, "", 1)]
    inputs, weights = [], []
    for f in feats:
        if f.kind == NRX_DENSE:
            inputs.append(torch.rand(B, device=DEV, generator=g))
            weights.append(None)
            continue
        shape = (B, f.bag_len) if f.bag_len else (B,)
        ids = torch.randint(0, rows[f.table], shape, device=DEV, generator=g)
        if f.kind == NRX_BAG_MASKED_MEAN:
            lens = torch.randint(0, f.bag_len + 1, (B,), device=DEV, generator=g)
            m = (torch.arange(f.bag_len, device=DEV)[None] < lens[:, None]).float()
            ids = ids * m.long()
            weights.append(m)
        else:
            weights.append(None)
        inputs.append(ids)
    return tables, feats, inputs, weights


def _direct(tables, feats, inputs, weights, fm):
    names = sorted(tables)
    slots, col = [], 0
    for f in feats:
        if f.kind == NRX_DENSE:
            slots.append(ops.Slot(f.name, NRX_DENSE, -1, 1, 0, col))
            col += 1
        else:
            slots.append(ops.Slot(f.name, f.kind, names.index(f.table), f.dim, f.bag_len, col, fm_field=int(fm)))
            col += f.dim
    plan = ops.EmbedPlan(slots, out_width=col, use_fm=fm)
    leaves = [tables[n].clone().requires_grad_(True) for n in names]
    return ops.embed_apply(plan, leaves, inputs, weights), leaves, names


@pytest.mark.parametrize("fm", [False, True])
@pytest.mark.parametrize("mode", ["capacity", "exact"])
def test_world1_engine_equals_direct_path(fm, mode):
    tables, feats, inputs, weights = _case(fm=fm)
    (out_d, _, fm_d), leaves_d, names = _direct(tables, feats, inputs, weights, fm)
    shards = {n: tables[n].clone().requires_grad_(True) for n in names}
    eng = RowShardedEmbedding(0, 1, mode=mode)
    out_s, _, fm_s = eng.forward(feats, inputs, weights, shards)
    # single-valued columns are routed copies: bit-exact.  Bag columns: in capacity mode they are pooled AT THE OWNER with
    # the normalisation folded into the weights (sum_l (w_l/den) row_l instead of (sum_l w_l row_l)/den): stated pooling
    # tolerance rtol 1e-6; in exact mode the rows travel and the pooling is the direct path's, bit for bit
    col = 0
    for f in feats:
        w = 1 if f.kind == NRX_DENSE else f.dim
        a, b = out_s[:, col:col + w], out_d[:, col:col + w]
        if f.bag_len and mode == "capacity":
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)
        else:
            assert torch.equal(a, b), f.name
        col += w
    up = torch.randn_like(out_d)
    loss_d = (out_d * up).sum() + (fm_d.sum() if fm else 0)
    loss_s = (out_s * up).sum() + (fm_s.sum() if fm else 0)
    if fm:
        np.testing.assert_allclose(fm_s.detach().cpu().numpy(), fm_d.detach().cpu().numpy(), rtol=1e-6, atol=1e-5)
    loss_d.backward()
    loss_s.backward()
    for n, leaf in zip(names, leaves_d):
        np.testing.assert_allclose(shards[n].grad.cpu().numpy(), leaf.grad.cpu().numpy(), rtol=1e-4, atol=1e-5)
        assert torch.all(shards[n].grad[0] == 0)          # global padding row (rank 0, local row 0)


def test_oob_raises_on_the_routed_path():
    tables, feats, inputs, weights = _case(B=50)
    inputs[1] = inputs[1].clone()
    inputs[1][7] = 77           # table b has 77 rows
    for mode in ("capacity", "exact"):
        eng = RowShardedEmbedding(0, 1, mode=mode)
        with pytest.raises(IndexError):
            eng.forward(feats, inputs, weights, {n: t for n, t in tables.items()})


@pytest.mark.parametrize("cls,cfg,gname", [(Deep, "cf_array_small.yaml", "model_deep_array"), (FM, "cf_fm_small.yaml", "model_fm")])
def test_shard_model_world1_matches_reference_golden(cls, cfg, gname):
    g = dict(np.load(os.path.join(GOLDEN, gname + ".npz"), allow_pickle=False))
    m = cls(os.path.join(CONFIGS, cfg))
    m.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param/")}, strict=True)
    m = m.to(DEV)
    keys_before = sorted(m.state_dict())
    sharding.shard_model_(m, 0, 1)
    assert sorted(m.state_dict()) == keys_before                      # checkpoint keys unchanged
    batch = {k[6:]: torch.from_numpy(v).to(DEV) for k, v in g.items() if k.startswith("batch/")}
    out = m(batch)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["out/forward"], rtol=1e-4, atol=1e-6)
    loss = m.bceLoss(out, batch["label"][:, 0])
    loss.backward()
    for name, emb in m.embedding_tables.items():
        want = g[f"grad/embedding_tables.{name}.weight"]
        np.testing.assert_allclose(emb.weight.grad.cpu().numpy(), want, rtol=2e-3, atol=2e-6 + 1e-4 * np.abs(want).max())


def test_prepared_sharded_forward_matches_engine():
    tables, feats, inputs, weights = _case(B=900)
    eng = RowShardedEmbedding(0, 1, overflow_policy="defer")
    ref_out, _, _ = eng.forward(feats, inputs, weights, tables)
    call = sharding.PreparedShardedForward(eng, feats, inputs, weights, tables)
    for _ in range(2):
        out, _, _ = call.run()
    assert torch.equal(out, ref_out)
    assert not call.overflowed() and not eng.overflowed()


def test_planner_replicated_tables_match_direct_path():
    """Planner layout at world=1 with the HIP backend: replicated features read their full table with the
    original ids, routed features go through the exchange; output and grads equal the direct path."""
    tables, feats, inputs, weights = _case(B=500)
    (out_d, _, _), leaves_d, names = _direct(tables, feats, inputs, weights, False)
    feats_p = [ShardedFeature(f.name, f.kind, f.table, f.dim, f.bag_len, f.wide, f.fm, f.table == "a") for f in feats]
    shards = {n: tables[n].clone().requires_grad_(True) for n in names}
    eng = RowShardedEmbedding(0, 1)
    out_s, _, _ = eng.forward(feats_p, inputs, weights, shards)
    assert torch.equal(out_s, out_d)
    up = torch.randn_like(out_d)
    (out_d * up).sum().backward()
    (out_s * up).sum().backward()
    for n, leaf in zip(names, leaves_d):
        np.testing.assert_allclose(shards[n].grad.cpu().numpy(), leaf.grad.cpu().numpy(), rtol=1e-4, atol=1e-5)
    call = sharding.PreparedShardedForward(RowShardedEmbedding(0, 1, overflow_policy="defer"), feats_p, inputs, weights, tables)
    assert torch.equal(call.run()[0], out_d.detach())


@pytest.mark.parametrize("fm", [False, True])
def test_one_sided_placement_world1_equals_the_direct_path(fm, monkeypatch):
    """PreparedShardedForward(one_sided=True) at world 1: the owner's gather (nrx_gather_inbox_place) writes every routed row straight into the
    concat -- bit-exact against the direct fused launch; with an FM epilogue the logit comes from a pass over the finished concat
    (nrx_fm_fwd), equal to the fused epilogue within fp32 summation order (rtol 1e-5: SURVEY 8a a5)."""
    from news_recsys_amd import ops
    from news_recsys_amd._lib import NRX_SPARSE
    monkeypatch.setenv("NRX_SHARD_ONE_SIDED_MIN", "0")          # small test groups: lift the world-1 size threshold
    g = torch.Generator(device=DEV).manual_seed(5)
    B, F, D, rows = 3001, 9, 16, 5000
    tables = {f"t{i}": torch.randn(rows + i, D, device=DEV, generator=g) for i in range(F)}
    feats = [ShardedFeature(f"f{i}", NRX_SPARSE, f"t{i}", D, 0, False, fm) for i in range(F)]
    inputs = [torch.randint(0, rows, (B,), device=DEV, generator=g) for _ in range(F)]
    weights = [None] * F
    plan = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=int(fm)) for i in range(F)], out_width=F * D, use_fm=fm)
    want, _, want_fm = ops.embed_apply(plan, [tables[f"t{i}"] for i in range(F)], inputs, weights)
    eng = RowShardedEmbedding(0, 1, overflow_policy="defer")
    call = sharding.PreparedShardedForward(eng, feats, inputs, weights, tables, one_sided=True)
    assert len(call.placed) == F and call.final is None
    for _ in range(2):
        out, _, fmv = call.run()
    assert torch.equal(out, want)
    if fm:
        torch.testing.assert_close(fmv, want_fm, rtol=1e-5, atol=1e-5 * float(want_fm.abs().max()))
    assert not call.overflowed()
    # a mix: one 16-wide bag and a dense value next to the placed features (they take the final launch, into the same concat)
    tables2, feats2, inputs2, weights2 = _case(B=700)
    ref = sharding.PreparedShardedForward(RowShardedEmbedding(0, 1, overflow_policy="defer"), feats2, inputs2, weights2, tables2).run()[0]
    ld = (ref.shape[1] + 3) // 4 * 4
    c2 = sharding.PreparedShardedForward(RowShardedEmbedding(0, 1, overflow_policy="defer"), feats2, inputs2, weights2, tables2, out_ld=ld,
                                         one_sided=True)
    got = c2.run()[0]
    assert torch.equal(got[:, :ref.shape[1]], ref)
