"""Fused row-sparse Adam (nrx_sparse_adam_step + ops.SparseGradSink + optim.FusedSparseAdam) against
torch.optim.SparseAdam fed with the COO gradients of the sparse_grad=True path: same weights after several
steps (the update rule is SparseAdam's, restated; the reference itself trains the tables with dense AdamW,
src/model/sort/deep/model.py:54-65 -- the row-sparse optimizer is a documented deviation)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(seed, shared):
    from news_recsys_amd import ops
    from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_SPARSE
    g = torch.Generator(device=DEV).manual_seed(seed)
    rows = [50, 70, 31]
    dims = [16, 16, 8]
    tables = [torch.randn(r, d, device=DEV, generator=g) for r, d in zip(rows, dims)]
    for t in tables:
        t[0].zero_()
    B, L = 200, 6
    slots = [ops.Slot("a", NRX_SPARSE, 0, 16, 0, 0), ops.Slot("h", NRX_BAG_MASKED_MEAN, 0 if shared else 1, 16, L, 16),
             ops.Slot("b", NRX_SPARSE, 1, 16, 0, 32), ops.Slot("c", NRX_SPARSE, 2, 8, 0, 48)]
    plan = ops.EmbedPlan(slots, out_width=56)
    def batch():
        ins = [torch.randint(0, rows[0], (B,), device=DEV, generator=g),
               torch.randint(0, rows[0 if shared else 1], (B, L), device=DEV, generator=g),
               torch.randint(0, rows[1], (B,), device=DEV, generator=g), torch.randint(0, rows[2], (B,), device=DEV, generator=g)]
        ws = [None, (torch.rand(B, L, device=DEV, generator=g) < 0.7).float(), None, None]
        up = torch.randn(B, 56, device=DEV, generator=g)
        return ins, ws, up
    return plan, tables, batch


@pytest.mark.parametrize("shared", [False, True])
def test_fused_sparse_adam_matches_torch_sparse_adam(shared):
    from news_recsys_amd import ops
    from news_recsys_amd.model.model_utils.optim import FusedSparseAdam
    plan, tables, batch = _setup(5, shared)
    ref = [t.clone().requires_grad_(True) for t in tables]
    fus = [t.clone().requires_grad_(True) for t in tables]
    opt_ref = torch.optim.SparseAdam(ref, lr=0.05, betas=(0.9, 0.999), eps=1e-8)
    sink = ops.SparseGradSink()
    opt_fus = FusedSparseAdam(sink, lr=0.05, betas=(0.9, 0.999), eps=1e-8)
    for step in range(4):
        ins, ws, up = batch()
        opt_ref.zero_grad()
        (ops.embed_apply(plan, ref, ins, ws, sparse_grad=True)[0] * up).sum().backward()
        opt_ref.step()
        (ops.embed_apply(plan, fus, ins, ws, sparse_grad=sink)[0] * up).sum().backward()
        assert all(t.grad is None for t in fus) and len(sink.pending) == 2        # one entry per embedding dim
        opt_fus.step()
        assert not sink.pending
    for a, b in zip(ref, fus):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=2e-5, atol=2e-6)
        assert torch.equal(b[0], torch.zeros_like(b[0]))                         # the padding row never moves


def test_fused_sparse_adam_merges_two_backward_groups_on_one_table():
    """Two embed calls reading the SAME table in one step (DSSM towers): one Adam update per row with the summed
    gradient, exactly what SparseAdam does with the accumulated COO grad."""
    from news_recsys_amd import ops
    from news_recsys_amd._lib import NRX_SPARSE
    from news_recsys_amd.model.model_utils.optim import FusedSparseAdam
    g = torch.Generator(device=DEV).manual_seed(2)
    t0 = torch.randn(40, 16, device=DEV, generator=g)
    t1 = torch.randn(30, 16, device=DEV, generator=g)
    planA = ops.EmbedPlan([ops.Slot("x", NRX_SPARSE, 0, 16, 0, 0), ops.Slot("y", NRX_SPARSE, 1, 16, 0, 16)], out_width=32)
    planB = ops.EmbedPlan([ops.Slot("z", NRX_SPARSE, 0, 16, 0, 0)], out_width=16)
    ref = [t0.clone().requires_grad_(True), t1.clone().requires_grad_(True)]
    fus = [t0.clone().requires_grad_(True), t1.clone().requires_grad_(True)]
    opt_ref = torch.optim.SparseAdam(ref, lr=0.1)
    sink = ops.SparseGradSink()
    opt_fus = FusedSparseAdam(sink, lr=0.1)
    for _ in range(3):
        ia = [torch.randint(1, 40, (64,), device=DEV, generator=g), torch.randint(1, 30, (64,), device=DEV, generator=g)]
        ib = [torch.randint(1, 40, (64,), device=DEV, generator=g)]
        ua, ub = torch.randn(64, 32, device=DEV, generator=g), torch.randn(64, 16, device=DEV, generator=g)
        opt_ref.zero_grad()
        loss = (ops.embed_apply(planA, ref, ia, [None, None], sparse_grad=True)[0] * ua).sum() + \
               (ops.embed_apply(planB, [ref[0]], ib, [None], sparse_grad=True)[0] * ub).sum()
        loss.backward()
        opt_ref.step()
        loss = (ops.embed_apply(planA, fus, ia, [None, None], sparse_grad=sink)[0] * ua).sum() + \
               (ops.embed_apply(planB, [fus[0]], ib, [None], sparse_grad=sink)[0] * ub).sum()
        loss.backward()
        assert len(sink.pending) == 2
        opt_fus.step()
    for a, b in zip(ref, fus):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=2e-5, atol=2e-6)


def test_model_trains_with_fused_sparse_grad(tmp_path):
    """`embeddings.sparse_grad: fused` through a model class: configure_optimizers returns the composite optimizer,
    a few steps reduce the loss, tables get no .grad."""
    import os
    import yaml
    import torch.nn.functional as F
    from news_recsys_amd.model.sort.deep.model import Deep
    from tests.conftest import CONFIGS
    cfg = yaml.safe_load(open(os.path.join(CONFIGS, "cf_array_small.yaml")))
    cfg["embeddings"]["sparse_grad"] = "fused"
    cfg["train_hparams"]["lr_milestones"] = [2000, 5000]
    p = tmp_path / "fused.yaml"
    p.write_text(yaml.safe_dump(cfg))
    torch.manual_seed(0)
    m = Deep(str(p)).to(DEV)
    opt = m.configure_optimizers()["optimizer"]
    g = torch.Generator(device=DEV).manual_seed(1)
    b = {}
    for n in m.sparse_feature_names:
        b[n] = torch.randint(1, m.embedding_tables[m._get_emb_feature_name(n)].weight.shape[0], (128,), device=DEV, generator=g)
    for n in m.array_feature_names:
        b[n] = torch.randint(1, m.embedding_tables[m._get_emb_feature_name(n)].weight.shape[0], (128, 9), device=DEV, generator=g)
        b[n + "_mask"] = (torch.rand(128, 9, device=DEV, generator=g) < 0.6).float()
    b["label"] = (torch.rand(128, 2, device=DEV, generator=g) < 0.4).float()
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = F.binary_cross_entropy(m(b).view(-1), b["label"][:, 0])
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(e.weight.grad is None for e in m.embedding_tables.values())
    assert losses[-1] < losses[0] - 0.05
    with torch.no_grad():
        assert torch.isfinite(m(b)).all()


def test_sparse_dense_adam_fused_checkpoint_resume():
    """optimizer.state_dict() must carry FusedSparseAdam's moments and step count -- a run restored from it
    continues bit-identically (the sorted backward and the fused step are deterministic), a run restarted without it
    does not."""
    import copy
    from news_recsys_amd import ops
    from news_recsys_amd.model.model_utils.optim import SparseDenseAdam
    plan, tables, batch = _setup(9, shared=True)
    lin = torch.nn.Linear(56, 1).to(DEV)
    batches = [batch() for _ in range(5)]

    def build(tabs, lin_):
        sink = ops.SparseGradSink()
        ps = [t.clone().requires_grad_(True) for t in tabs]
        return ps, sink, SparseDenseAdam(ps, list(lin_.parameters()), lr=0.05, fused_sink=sink)

    def run(ps, sink, opt, lin_, bs):
        for ins, ws, up in bs:
            opt.zero_grad()
            (lin_(ops.embed_apply(plan, ps, ins, ws, sparse_grad=sink)[0]) * up[:, :1]).sum().backward()
            opt.step()

    pa, sa, oa = build(tables, lin)
    run(pa, sa, oa, lin, batches[:3])
    sd = oa.state_dict()
    assert sd["sparse"]["t"] == 3 and len(sd["sparse"]["tables"]) == 3 and sd["dense"]["state"]
    sd = copy.deepcopy(sd)
    lin_b, lin_c = copy.deepcopy(lin), copy.deepcopy(lin)
    pb, sb, ob = build([p.detach() for p in pa], lin_b)          # restored
    ob.load_state_dict(sd)
    pc, sc, oc = build([p.detach() for p in pa], lin_c)          # weights only: Adam restarts
    run(pa, sa, oa, lin, batches[3:])
    run(pb, sb, ob, lin_b, batches[3:])
    run(pc, sc, oc, lin_c, batches[3:])
    for a, b, c in zip(pa, pb, pc):
        assert torch.equal(a.detach(), b.detach())
        assert not torch.equal(a.detach(), c.detach())
    assert torch.equal(lin.weight, lin_b.weight)
