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


@pytest.mark.parametrize("pair_merge", [True, False])
def test_fused_sparse_adam_merges_two_backward_groups_on_one_table(pair_merge):
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
    opt_fus.pair_merge = pair_merge              # True (default): nrx_rows_mark / nrx_rows_merge; False: the sort-based merge
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
        assert all(int((m >= 0).sum()) == 0 for m in opt_fus._maps)               # the slot maps are clean between steps
    assert (len(opt_fus._maps) > 0) == pair_merge
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


@pytest.mark.parametrize("exact", [False, True])
def test_sparse_dense_adam_fused_checkpoint_resume(exact):
    """optimizer.state_dict() must carry FusedSparseAdam's moments and step count -- a run restored from it
    continues bit-identically (the sorted backward and the fused step are deterministic), a run restarted without it
    does not.  exact: the same for ExactDenseAdamW (the reference's dense AdamW fed from the sink)."""
    import copy
    from news_recsys_amd import ops
    from news_recsys_amd.model.model_utils.optim import SparseDenseAdam
    plan, tables, batch = _setup(9, shared=True)
    lin = torch.nn.Linear(56, 1).to(DEV)
    batches = [batch() for _ in range(5)]

    def build(tabs, lin_):
        sink = ops.SparseGradSink()
        ps = [t.clone().requires_grad_(True) for t in tabs]
        return ps, sink, SparseDenseAdam(ps, list(lin_.parameters()), lr=0.05, fused_sink=sink, exact=exact)

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


@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("B", [200, 6000])
def test_exact_dense_adamw_from_the_sink_matches_torch_adamw_on_dense_gradients(shared, B):
    """optim.ExactDenseAdamW (nrx_rows_mark + nrx_dense_adamw_rows) = the reference's optimizer for the tables, one dense torch.optim.AdamW
    over every parameter (src/model/sort/deep/model.py:54-65): EVERY row of every table that has a gradient moves every step (weight decay,
    decaying moments); a table no launch looked up has no gradient and stays, as under torch.  Same weights and moments as torch.optim.AdamW fed with the dense gradients of the default mode; B = 200
    goes through the one-launch small backward (filler keys), B = 6000 through the planned reduction (device-side count)."""
    from news_recsys_amd import ops
    from news_recsys_amd.model.model_utils.optim import ExactDenseAdamW
    plan, tables, batch = _setup(7, shared)
    extra = torch.randn(23, 16, device=DEV)                                       # a table of the model that this step never looks up
    ref = [t.clone().requires_grad_(True) for t in tables + [extra]]
    exa = [t.clone().requires_grad_(True) for t in tables + [extra]]
    opt_ref = torch.optim.AdamW(ref, lr=0.03, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    sink = ops.SparseGradSink()
    opt_exa = ExactDenseAdamW(sink, exa, lr=0.03, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    g = torch.Generator(device=DEV).manual_seed(11)
    for step in range(4):
        ins, ws, up = batch()
        if B != 200:                                                              # a larger batch of the same plan
            rep = B // 200
            ins = [x.repeat((rep,) + (1,) * (x.dim() - 1)) for x in ins]
            ws = [None if w is None else w.repeat(rep, 1) for w in ws]
            up = torch.randn(B, 56, device=DEV, generator=g)
        if step == 2:
            opt_ref.param_groups[0]["lr"] = opt_exa.lr = 0.01                     # a scheduler's edit
        opt_ref.zero_grad()
        (ops.embed_apply(plan, ref[:3], ins, ws)[0] * up).sum().backward()
        assert ref[3].grad is None                                                # torch.optim.AdamW skips a parameter without a gradient ...
        opt_ref.step()
        (ops.embed_apply(plan, exa[:3], ins, ws, sparse_grad=sink)[0] * up).sum().backward()
        assert all(t.grad is None for t in exa)
        opt_exa.step()
        assert not sink.pending and all(int((m >= 0).sum()) == 0 for m in opt_exa.maps)      # the slot maps are clean again
    for i, (a, b) in enumerate(zip(ref, exa)):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=2e-5, atol=2e-6)
        st = opt_ref.state[a]
        if i == 3:                                                                # the table nobody looked up: torch never made its state,
            assert "exp_avg" not in st and not opt_exa.moments[i][0].any() and not opt_exa.moments[i][1].any()      # ours never moved
            continue
        # (the moments see the two backwards' fp32 addition orders directly: rows that sum ~100 terms differ by ~1e-5 absolute)
        torch.testing.assert_close(st["exp_avg"], opt_exa.moments[i][0], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(st["exp_avg_sq"], opt_exa.moments[i][1], rtol=1e-4, atol=1e-5)
    assert torch.equal(exa[3].detach(), extra) and torch.equal(ref[3].detach(), extra)      # ... and so does the sink-fed form: the table nobody
                                                                                             # looked up has not moved (round-4 advice)


def test_model_trains_with_exact_sparse_grad_like_the_default_mode(tmp_path):
    """`embeddings.sparse_grad: exact` through a model class against the default mode (dense .grad + torch.optim.AdamW over every parameter,
    the reference's configure_optimizers): same parameters after a few steps from the same initialisation."""
    import os
    import yaml
    import torch.nn.functional as F
    from news_recsys_amd.model.sort.deep.model import Deep
    from tests.conftest import CONFIGS
    models = []
    for mode in (False, "exact"):
        cfg = yaml.safe_load(open(os.path.join(CONFIGS, "cf_array_small.yaml")))
        cfg["embeddings"]["sparse_grad"] = mode
        cfg["train_hparams"]["lr_milestones"] = [2000, 5000]
        p = tmp_path / f"m_{mode}.yaml"
        p.write_text(yaml.safe_dump(cfg))
        torch.manual_seed(0)
        models.append(Deep(str(p)).to(DEV))
    models[1].load_state_dict(models[0].state_dict())
    m0 = models[0]
    g = torch.Generator(device=DEV).manual_seed(1)
    b = {}
    for n in m0.sparse_feature_names:
        b[n] = torch.randint(1, m0.embedding_tables[m0._get_emb_feature_name(n)].weight.shape[0], (128,), device=DEV, generator=g)
    for n in m0.array_feature_names:
        b[n] = torch.randint(1, m0.embedding_tables[m0._get_emb_feature_name(n)].weight.shape[0], (128, 9), device=DEV, generator=g)
        b[n + "_mask"] = (torch.rand(128, 9, device=DEV, generator=g) < 0.6).float()
    b["label"] = (torch.rand(128, 2, device=DEV, generator=g) < 0.4).float()
    for m in models:
        opt = m.configure_optimizers()["optimizer"]
        for _ in range(5):
            opt.zero_grad()
            F.binary_cross_entropy(m(b).view(-1), b["label"][:, 0]).backward()
            opt.step()
    assert all(e.weight.grad is None for e in models[1].embedding_tables.values())
    for (k, p0), (_, p1) in zip(models[0].named_parameters(), models[1].named_parameters()):
        torch.testing.assert_close(p0.detach(), p1.detach(), rtol=2e-4, atol=2e-6, msg=k)


def test_exact_dense_adamw_captured_in_a_graph_advances_its_bias_corrections():
    """ExactDenseAdamW(capturable=True): forward + backward + step() captured once, replayed on new batches -- the step count lives on the
    device, so replay k uses the bias corrections of step k: same weights as torch.optim.AdamW stepping eagerly on the same batches."""
    from news_recsys_amd import ops
    from news_recsys_amd.model.model_utils.optim import ExactDenseAdamW
    plan, tables, batch = _setup(9, True)
    ref = [t.clone().requires_grad_(True) for t in tables]
    exa = [t.clone().requires_grad_(True) for t in tables]
    opt_ref = torch.optim.AdamW(ref, lr=0.02, weight_decay=0.01)
    sink = ops.SparseGradSink()
    opt_exa = ExactDenseAdamW(sink, exa, lr=0.02, weight_decay=0.01, capturable=True)
    batches = [batch() for _ in range(5)]
    static = [[x.clone() for x in batches[0][0]], [None if w is None else w.clone() for w in batches[0][1]], batches[0][2].clone()]
    prev = ops._INDEX_CHECK
    ops.set_index_check("off")
    try:
        def step():
            (ops.embed_apply(plan, exa, static[0], static[1], sparse_grad=sink)[0] * static[2]).sum().backward()
            opt_exa.step()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()                                                               # warm-up = training step 1 (on batch 0)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()                                                               # captured; the capture itself runs nothing
        for ins, ws, up in batches[1:]:
            for d, x in zip(static[0], ins):
                d.copy_(x)
            for d, x in zip(static[1], ws):
                if d is not None:
                    d.copy_(x)
            static[2].copy_(up)
            g.replay()
        torch.cuda.synchronize()
    finally:
        ops.set_index_check(prev)
    for ins, ws, up in batches:
        opt_ref.zero_grad()
        (ops.embed_apply(plan, ref, ins, ws)[0] * up).sum().backward()
        opt_ref.step()
    for a, b in zip(ref, exa):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=5e-5, atol=5e-6)
