"""SparseDenseAdam checkpointing and closure handling: all Adam state lives in the inner optimizers, so
state_dict() / load_state_dict() -- what Lightning saves and restores -- must carry it; step(closure) must run the closure
with gradients enabled.  CPU: the COO path (torch SparseAdam inside); the fused path is covered on the GPU in
tests/test_fused_sparse_adam_gpu.py."""
import copy

import torch

from news_recsys_amd.model.model_utils.optim import SparseDenseAdam


def _model(seed):
    g = torch.Generator().manual_seed(seed)
    emb = torch.nn.Embedding(30, 4, sparse=True)
    lin = torch.nn.Linear(4, 1)
    with torch.no_grad():
        emb.weight.copy_(torch.randn(30, 4, generator=g))
        lin.weight.copy_(torch.randn(1, 4, generator=g))
        lin.bias.zero_()
    return emb, lin


def _loss(emb, lin, ids, y):
    return ((lin(emb(ids)).squeeze(1) - y) ** 2).mean()


def test_state_dict_round_trip_resumes_identically():
    g = torch.Generator().manual_seed(0)
    batches = [(torch.randint(0, 30, (16,), generator=g), torch.randn(16, generator=g)) for _ in range(5)]
    emb, lin = _model(1)
    opt = SparseDenseAdam([emb.weight], list(lin.parameters()), lr=0.05)
    for ids, y in batches[:3]:
        opt.zero_grad()
        _loss(emb, lin, ids, y).backward()
        opt.step()
    sd = copy.deepcopy(opt.state_dict())
    assert sd["sparse"]["state"] and sd["dense"]["state"], "moments must be in the checkpoint"
    emb2, lin2 = _model(2)                                # a freshly built model + optimizer, then restore
    emb2.load_state_dict(emb.state_dict())
    lin2.load_state_dict(lin.state_dict())
    opt2 = SparseDenseAdam([emb2.weight], list(lin2.parameters()), lr=0.01)
    opt2.load_state_dict(sd)
    assert opt2.param_groups[0]["lr"] == 0.05
    for ids, y in batches[3:]:
        for e, l, o in ((emb, lin, opt), (emb2, lin2, opt2)):
            o.zero_grad()
            _loss(e, l, ids, y).backward()
            o.step()
    assert torch.equal(emb.weight, emb2.weight) and torch.equal(lin.weight, lin2.weight)
    # and a fresh optimizer WITHOUT the restore diverges (bias correction restarts, moments are zero)
    emb3, lin3 = _model(3)
    emb3.load_state_dict(emb2.state_dict())
    assert opt2.state_dict()["sparse"]["state"][0]["step"] == 5


def test_step_runs_closure_with_grad_enabled():
    emb, lin = _model(4)
    opt = SparseDenseAdam([emb.weight], list(lin.parameters()), lr=0.05)
    ids, y = torch.arange(8), torch.ones(8)
    before = emb.weight.detach().clone()

    def closure():
        opt.zero_grad()
        loss = _loss(emb, lin, ids, y)
        loss.backward()                   # fails under no_grad if step() does not re-enable gradients
        return loss

    loss = opt.step(closure)
    assert loss is not None and loss.item() > 0
    assert not torch.equal(before[:8], emb.weight.detach()[:8]) and torch.equal(before[8:], emb.weight.detach()[8:])


def test_exact_dense_adamw_restores_step_counts_from_a_checkpoint_without_them():
    """A checkpoint written before ExactDenseAdamW kept per-table step counts has `t` but no `steps`: on load every registered table takes
    `t` (each had moved on every step), so the bias corrections continue where torch.optim.AdamW's would; step counts saved under
    an `unlisted:` key (a table that was not in `params`) are refused instead of dropped.  Host logic only: no kernel runs."""
    import pytest
    from news_recsys_amd import ops
    from news_recsys_amd.model.model_utils.optim import ExactDenseAdamW
    tables = [torch.zeros(5, 4), torch.zeros(7, 4)]
    opt = ExactDenseAdamW(ops.SparseGradSink(), tables, lr=0.1)
    old = {"t": 9, "t_dev": None, "tables": {i: {"exp_avg": torch.ones_like(t), "exp_avg_sq": torch.ones_like(t)} for i, t in enumerate(tables)}}
    opt.load_state_dict(old)                                   # (no "steps" entry: the pre-round-5 format)
    assert opt.t == 9 and opt._steps == {0: 9, 1: 9}
    assert torch.equal(opt.moments[0][0], torch.ones(5, 4))
    new = dict(old, steps={0: 9, 1: 4})                        # the current format: a table that was looked up in 4 of the 9 steps
    opt.load_state_dict(new)
    assert opt._steps == {0: 9, 1: 4}
    with pytest.raises(ValueError, match="unlisted"):
        opt.load_state_dict(dict(old, steps={0: 9, "unlisted:1": 4}))
