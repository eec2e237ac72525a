"""A whole training step on the fused path captured in a HIP graph (news_recsys_amd.graph.GraphedStep) gives the
losses of the eager loop and -- up to float-atomic reordering in the dense-gradient scatter -- its parameters."""
import os

import pytest
import torch
import torch.nn.functional as F

from tests.conftest import CONFIGS

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(m, B, gen):
    b = {}
    for n in m.sparse_feature_names:
        rows = m.embedding_tables[m._get_emb_feature_name(n)].weight.shape[0]
        b[n] = torch.randint(1, rows, (B,), device=DEV, generator=gen)
    for n in m.array_feature_names:
        rows = m.embedding_tables[m._get_emb_feature_name(n)].weight.shape[0]
        b[n] = torch.randint(1, rows, (B, 9), device=DEV, generator=gen)
        b[n + "_mask"] = (torch.rand(B, 9, device=DEV, generator=gen) < 0.6).float()
    b["label"] = (torch.rand(B, 2, device=DEV, generator=gen) < 0.3).float()
    return b


def test_graphed_training_step_matches_eager():
    from news_recsys_amd import ops
    from news_recsys_amd.graph import GraphedStep
    from news_recsys_amd.model.sort.deep.model import Deep
    cfg = os.path.join(CONFIGS, "cf_array_small.yaml")
    torch.manual_seed(3)
    m_e = Deep(cfg).to(DEV)
    m_g = Deep(cfg).to(DEV)
    m_g.load_state_dict(m_e.state_dict())
    opt_e = torch.optim.Adam(m_e.parameters(), lr=1e-3, capturable=True)
    opt_g = torch.optim.Adam(m_g.parameters(), lr=1e-3, capturable=True)
    gen = torch.Generator(device=DEV).manual_seed(0)
    batches = [_batch(m_e, 256, gen) for _ in range(5)]

    def make_step(m, opt):
        def step(b):
            opt.zero_grad(set_to_none=False)
            loss = F.binary_cross_entropy(m(b).view(-1), b["label"][:, 0])
            loss.backward()
            opt.step()
            return loss
        return step

    snapshot = {k: v.clone() for k, v in m_g.state_dict().items()}
    mode_before = ops._INDEX_CHECK
    gs = GraphedStep(make_step(m_g, opt_g), batches[0], warmup=2)
    assert ops._INDEX_CHECK == mode_before                      # the capture restores the caller's check mode
    m_g.load_state_dict(snapshot)                               # undo the warm-up / capture updates
    for st in opt_g.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                v.zero_()
    step_e = make_step(m_e, opt_e)
    ops.set_index_check("off")
    try:
        for b in batches:
            le = step_e(b).item()
            lg = gs(b).item()
            assert abs(le - lg) <= 1e-5 * max(1.0, abs(le))      # float atomics in the dense scatter: order differs run to run
    finally:
        ops.set_index_check(mode_before)
    # Adam's first steps move a weight by ~lr * sign(gradient): where a gradient is within float-atomic reordering of zero, the
    # two runs may step in different directions (seen: 3 of 16 384 weights of one layer 6e-5 apart, about once in 35 runs).
    # A replay that was NOT the eager step (stale inputs, a skipped or doubled update) moves every weight, by up to lr per step.
    lr, n_steps = 1e-3, len(batches)
    for p, q in zip(m_e.parameters(), m_g.parameters()):
        d = (p - q).abs()
        assert d.max().item() <= 2 * lr * n_steps
        assert (d > 5e-5).float().mean().item() <= 2e-3, (d > 5e-5).sum().item()
    with pytest.raises(ValueError):
        gs(_batch(m_e, 128, gen))                               # a different batch shape cannot be replayed


def test_graphed_fused_sparse_training_step_matches_eager():
    """The fused row-sparse path (sorted backward left on the device + nrx_sparse_adam_step with a device-side step
    size) is capturable too: replaying the graph == running the same step eagerly."""
    import yaml
    from news_recsys_amd import ops
    from news_recsys_amd.graph import GraphedStep
    from news_recsys_amd.model.model_utils.optim import SparseDenseAdam
    from news_recsys_amd.model.sort.deep.model import Deep
    cfg = os.path.join(CONFIGS, "cf_array_small.yaml")
    torch.manual_seed(5)

    def build():
        m = Deep(cfg).to(DEV)
        m.sparse_grad = "fused"
        m._sparse_sink = ops.SparseGradSink()
        tabs = [e.weight for e in m.embedding_tables.values()]
        ids = {id(p) for p in tabs}
        opt = SparseDenseAdam(tabs, [p for p in m.parameters() if id(p) not in ids], lr=1e-2, fused_sink=m._sparse_sink, capturable=True)
        return m, opt

    m_e, opt_e = build()
    m_g, opt_g = build()
    m_g.load_state_dict(m_e.state_dict())
    gen = torch.Generator(device=DEV).manual_seed(0)
    batches = [_batch(m_e, 256, gen) for _ in range(5)]

    def make_step(m, opt):
        def step(b):
            opt.zero_grad(set_to_none=False)
            loss = F.binary_cross_entropy(m(b).view(-1), b["label"][:, 0])
            loss.backward()
            opt.step()
            return loss
        return step

    mode_before = ops._INDEX_CHECK
    ops.set_index_check("off")
    try:
        # identical histories: GraphedStep runs its 2 warm-up steps eagerly on m_g (the capture itself only records),
        # so m_e takes the same 2 eager steps first
        gs = GraphedStep(make_step(m_g, opt_g), batches[0], warmup=2)
        step_e = make_step(m_e, opt_e)
        for _ in range(2):
            step_e(batches[0])
        for b in batches:
            le = step_e(b).item()
            lg = gs(b).item()
            assert abs(le - lg) <= 2e-5 * max(1.0, abs(le)), (le, lg)
    finally:
        ops.set_index_check(mode_before)
    for p, q in zip(m_e.parameters(), m_g.parameters()):
        torch.testing.assert_close(p, q, rtol=1e-4, atol=2e-5)


def test_out_of_range_id_inside_a_replay_raises_indexerror_naming_the_feature():
    """The reference raises IndexError for an out-of-range id (torch on CPU, src/model/BaseModel/base_model.py:271).  A captured step keeps that
    contract one call late: the gather kernels of the graph record the offence in the host-mapped status word, the next replay (or check())
    raises IndexError with the feature's name; clean replays before and after are unaffected."""
    from news_recsys_amd import ops
    from news_recsys_amd.graph import GraphedStep
    from news_recsys_amd.model.sort.deep.model import Deep
    cfg = os.path.join(CONFIGS, "cf_array_small.yaml")
    torch.manual_seed(11)
    m = Deep(cfg).to(DEV)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)
    gen = torch.Generator(device=DEV).manual_seed(1)
    batches = [_batch(m, 128, gen) for _ in range(4)]

    def step(b):
        opt.zero_grad(set_to_none=False)
        loss = F.binary_cross_entropy(m(b).view(-1), b["label"][:, 0])
        loss.backward()
        opt.step()
        return loss

    gs = GraphedStep(step, batches[0], warmup=2)
    gs(batches[1])
    gs.check()                                                  # clean so far
    victim = sorted(m.sparse_feature_names)[0]
    bad = dict(batches[2])
    bad[victim] = bad[victim].clone()
    bad[victim][5] = 10 ** 7
    gs(bad)                                                     # the replay itself does not raise (nothing is read back) ...
    with pytest.raises(IndexError, match=victim):
        gs(batches[3])                                          # ... the next call does, naming the feature
    gs(batches[3])                                              # the word was cleared by the raise: clean batches replay again
    gs.check()
    bad2 = dict(batches[1])
    arr = sorted(m.array_feature_names)[0]
    bad2[arr] = bad2[arr].clone()
    bad2[arr][3, 2] = -4
    bad2[arr + "_mask"] = bad2[arr + "_mask"].clone()
    bad2[arr + "_mask"][3, 2] = 1.0
    gs(bad2)
    with pytest.raises(IndexError, match=arr):
        gs.check()


def test_deterministic_graphed_step_has_bit_reproducible_table_gradients():
    """GraphedStep(deterministic=True): the dense table gradients of the captured step come from the sorted reduction at any batch size (planned
    inside the graph, count read on the device) -- two replays from the same state give the same bits, and they equal the eager sorted path;
    the default capture (float-atomic scatter at this batch size) only agrees within the order noise."""
    from news_recsys_amd import ops
    from news_recsys_amd.graph import GraphedStep
    from news_recsys_amd.model.sort.deep.model import Deep
    cfg = os.path.join(CONFIGS, "cf_array_small.yaml")
    torch.manual_seed(13)
    m = Deep(cfg).to(DEV)
    gen = torch.Generator(device=DEV).manual_seed(2)
    batches = [_batch(m, 512, gen) for _ in range(3)]
    for b in batches:                                           # duplicates: rows that several samples hit, in every table
        for n in m.sparse_feature_names:
            b[n][::3] = b[n][0]
    tabs = [e.weight for e in m.embedding_tables.values()]

    def step(b):
        for p in m.parameters():
            if p.grad is not None:
                p.grad.zero_()
        loss = F.binary_cross_entropy(m(b).view(-1), b["label"][:, 0])
        loss.backward()
        return loss

    gs = GraphedStep(step, batches[0], warmup=2, deterministic=True)
    runs = []
    for _ in range(3):
        gs(batches[1])
        torch.cuda.synchronize()
        runs.append([t.grad.clone() for t in tabs])
    for g0, g1, g2 in zip(*runs):
        assert torch.equal(g0.view(torch.int32), g1.view(torch.int32)) and torch.equal(g0.view(torch.int32), g2.view(torch.int32))
    prev = ops.DENSE_BWD_SORTED
    ops.DENSE_BWD_SORTED = "det"
    try:
        step(batches[1])                                        # eager, the same deterministic mode
        torch.cuda.synchronize()
        for g0, t in zip(runs[0], tabs):
            assert torch.equal(g0.view(torch.int32), t.grad.view(torch.int32))
    finally:
        ops.DENSE_BWD_SORTED = prev
    ops.DENSE_BWD_SORTED = False
    try:
        step(batches[1])                                        # eager, float atomics: same gradient up to the order of the additions
        torch.cuda.synchronize()
        for g0, t in zip(runs[0], tabs):
            torch.testing.assert_close(g0, t.grad, rtol=1e-4, atol=1e-5)
    finally:
        ops.DENSE_BWD_SORTED = prev


def test_deterministic_graphed_step_of_a_dcn_v2_model_reproduces_every_gradient(tmp_path, monkeypatch):
    """GraphedStep(deterministic=True) on a DCN model with the v2 cross stack and the package's MLP weight gradient (utils.MLP_WGRAD): the captured step
    takes the ORDERED weight gradients (ops.WGRAD_ORDERED: per-slice partial tiles added in slice order) next to the deterministic table gradients --
    EVERY parameter's gradient is the same bits over three replays of the same batch; the default capture only promises that of nothing that sums
    over the batch with atomics.  B = 6000: several batch slices per weight gradient, more than 4096 lookups per table."""
    import yaml
    from news_recsys_amd import ops
    from news_recsys_amd.graph import GraphedStep
    from news_recsys_amd.model.model_utils import utils as mlp_utils
    from news_recsys_amd.model.sort.dcn.model import DCN
    cfg = yaml.safe_load(open(os.path.join(CONFIGS, "cf_dcn_small.yaml")))
    cfg.setdefault("dcn_cfg", {})["version"] = 2
    cfg["dcn_cfg"]["cross_num_layers"] = 2
    path = tmp_path / "dcn_v2.yaml"
    path.write_text(yaml.safe_dump(cfg))
    monkeypatch.setattr(mlp_utils, "MLP_WGRAD", True)
    torch.manual_seed(29)
    m = DCN(str(path)).to(DEV)
    gen = torch.Generator(device=DEV).manual_seed(4)
    batches = [_batch(m, 6000, gen) for _ in range(2)]

    def step(b):
        for p in m.parameters():
            if p.grad is not None:
                p.grad.zero_()
        loss = F.binary_cross_entropy(m(b).view(-1), b["label"][:, 0])
        loss.backward()
        return loss

    before = ops.WGRAD_ORDERED
    gs = GraphedStep(step, batches[0], warmup=2, deterministic=True)
    assert ops.WGRAD_ORDERED == before                          # (the switch is the capture's, not the process's)
    runs = []
    for _ in range(3):
        gs(batches[1])
        torch.cuda.synchronize()
        runs.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    assert len(runs[0]) >= 8
    for n, g0 in runs[0].items():
        for r in runs[1:]:
            assert torch.equal(g0.view(torch.int32), r[n].view(torch.int32)), n
    step(batches[1])                                            # eager, default modes: the same gradients up to the order of the additions
    torch.cuda.synchronize()
    for n, p in m.named_parameters():
        if p.grad is not None:
            torch.testing.assert_close(runs[0][n], p.grad, rtol=2e-3, atol=2e-5)


def test_deterministic_mode_says_so_when_it_cannot_be_deterministic(monkeypatch):
    """A deterministic mode that falls back to float atomics must not do so silently (round-4 advice): with more than 64 tables neither the
    planned reduction nor -- beyond 4096 lookups per table -- the one-launch kernel serves the launch: the eager backward warns, and
    GraphedStep(deterministic=True) refuses the capture."""
    import warnings
    from news_recsys_amd import ops
    from news_recsys_amd._lib import NRX_SPARSE
    from news_recsys_amd.graph import GraphedStep
    n, D, rows, B = 66, 16, 600, 4100
    slots = [ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D) for i in range(n)]
    plan = ops.EmbedPlan(slots, out_width=n * D)
    gen = torch.Generator(device=DEV).manual_seed(9)
    tables = [torch.randn((rows, D), device=DEV, generator=gen).requires_grad_() for _ in range(n)]
    ids = [torch.randint(0, rows, (B,), device=DEV, generator=gen) for _ in range(n)]

    def step(_b=None):
        for t in tables:
            t.grad = None
        out = ops.embed_apply(plan, tables, ids, [None] * n)[0]
        out.sum().backward()
        return out

    monkeypatch.setattr(ops, "DENSE_BWD_SORTED", "det")
    ops._atomic_warned.clear()
    before = ops.dense_bwd_paths["atomic"]
    with pytest.warns(UserWarning, match="float atomics"):
        step()
    assert ops.dense_bwd_paths["atomic"] > before
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(RuntimeError, match="float atomics"):
            GraphedStep(step, {"x": ids[0]}, warmup=1, deterministic=True)
    # ... and the supported shape (<= 64 tables) still takes a deterministic form, silently
    monkeypatch.setattr(ops, "DENSE_BWD_SORTED", "det")
    plan2 = ops.EmbedPlan(slots[:8], out_width=8 * D)
    before = dict(ops.dense_bwd_paths)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for t in tables[:8]:
            t.grad = None
        ops.embed_apply(plan2, tables[:8], ids[:8], [None] * 8)[0].sum().backward()
    assert ops.dense_bwd_paths["atomic"] == before.get("atomic", 0) and ops.dense_bwd_paths["sorted"] + ops.dense_bwd_paths["small"] > before.get("sorted", 0) + before.get("small", 0)
