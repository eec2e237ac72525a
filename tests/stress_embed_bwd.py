#!/usr/bin/env python3
"""Dev (GPU box): random gather launches -- 1 .. 30 features (single-valued, masked-mean / mean / sum bags, tables shared, one dim per launch from 16 / 32 / 64 or
mixed), batches 1 .. 30 000, uniform / skewed ids with padding ids and padded histories -- forward and backward through ops.embed_apply in every mode of the
backward (row-sparse; dense: auto, planned, deterministic, atomic; one-kernel planner allowed / forbidden; padding split always / never), against a float64
restatement in torch (F.embedding + the pooling of base_model.py:262-282, autograd for the gradients); the deterministic modes against each other word for word.
usage: python tests/stress_embed_bwd.py [seconds=120] [seed=1]   (a checker like the tests next to it; not collected by pytest)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _poison
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM, NRX_SPARSE
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
DEV = "cuda:0"
ops.set_index_check("deferred")
t0, n_done = time.time(), 0
while time.time() - t0 < budget:
    _poison.poison()
    nf = int(rng.choice([1, 2, 3, 5, 9, 26, 30]))
    nt = int(rng.integers(1, nf + 1))
    B = int(rng.choice([1, 3, 64, 700, 4097, 9000, 30000]))
    mixed = bool(rng.integers(0, 4) == 0)
    dim0 = int(rng.choice([16, 32, 64]))
    tdim = [int(rng.choice([16, 32, 64])) if mixed else dim0 for _ in range(nt)]
    trows = [int(rng.choice([2, 50, 3000, 200000, 1500000])) for _ in range(nt)]
    slots, ins, ws, col, look = [], [], [], 0, 0
    for f in range(nf):
        t = int(rng.integers(0, nt))
        kind = int(rng.choice([NRX_SPARSE, NRX_SPARSE, NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM]))
        L = 0 if kind == NRX_SPARSE else int(rng.choice([1, 4, 17, 50]))
        look += B * max(L, 1)
        slots.append(ops.Slot(f"f{f}", kind, t, tdim[t], L, col))
        col += tdim[t]
        shape = (B,) if L == 0 else (B, L)
        skew = rng.integers(0, 3) == 0
        x = rng.integers(0, trows[t], shape) if not skew else np.minimum(rng.zipf(1.3, shape) - 1, trows[t] - 1)
        w = None
        if L:
            n_valid = rng.integers(0, L + 1, (B, 1))
            m = (np.arange(L)[None, :] < n_valid).astype(np.float32)
            if kind == NRX_BAG_MASKED_MEAN or (kind == NRX_BAG_SUM and rng.integers(0, 2)):
                x = x * m.astype(np.int64)                  # padded histories: id 0 behind the valid entries
                w = m * (rng.random((B, L)).astype(np.float32) if kind == NRX_BAG_SUM and rng.integers(0, 2) else 1.0)
        ins.append(torch.from_numpy(np.asarray(x, np.int64)).to(DEV))
        ws.append(None if w is None else torch.from_numpy(np.asarray(w, np.float32)).to(DEV))
    if look > 1_600_000:
        continue
    plan = ops.EmbedPlan(slots, out_width=col)
    tables = [torch.randn(trows[t], tdim[t], device=DEV) for t in range(nt)]
    up = torch.randn(B, col, device=DEV)
    # float64 restatement
    t64 = [t.double().requires_grad_() for t in tables]
    outs = []
    for s_, x, w in zip(slots, ins, ws):
        e = torch.nn.functional.embedding(x, t64[s_.table])
        if s_.kind == NRX_SPARSE:
            outs.append(e)
        elif s_.kind == NRX_BAG_MASKED_MEAN:
            wd = w.double()
            outs.append((e * wd.unsqueeze(-1)).sum(1) / (wd.sum(1, keepdim=True) + 1e-8))
        elif s_.kind == NRX_BAG_MEAN:
            outs.append(e.mean(1))
        else:
            outs.append((e * w.double().unsqueeze(-1)).sum(1) if w is not None else e.sum(1))
    ref_out = torch.cat(outs, 1)
    ref_g = torch.autograd.grad(ref_out, t64, up.double(), allow_unused=True)
    ref_g = [g if g is not None else torch.zeros_like(t) for g, t in zip(ref_g, t64)]
    for g in ref_g:                                            # the padding row never trains: its gradient is zero in every mode
        g[0] = 0
    n_max = [1] * nt                                           # lookups of the hottest row of every table: the length of the longest fp32 sum
    for s_, x in zip(slots, ins):
        c = torch.bincount(x.reshape(-1)[x.reshape(-1) > 0], minlength=1)
        n_max[s_.table] += int(c.max().item()) if c.numel() else 0

    def run(sparse_grad, dense_mode, lds, split):
        ops.DENSE_BWD_SORTED, ops.PLAN_LDS, ops.PAD_SPLIT = dense_mode, lds, split
        plan.__dict__.pop("_sg", None)
        ts = [t.clone().requires_grad_() for t in tables]
        out = ops.embed_apply(plan, ts, ins, ws, sparse_grad=sparse_grad)[0]
        out.backward(up)
        torch.cuda.synchronize()
        return out.detach(), [(t.grad.to_dense() if t.grad is not None and t.grad.is_sparse else (t.grad if t.grad is not None else torch.zeros_like(t))) for t in ts]
    modes = [("row-sparse", True, None, "auto", "auto"), ("row-sparse, sorted planner, split", True, None, "0", "1"), ("dense auto", False, None, "auto", "auto"),
             ("dense planned", False, True, "auto", "0"), ("dense deterministic", False, "det", "0", "1"), ("dense atomic", False, False, "auto", "auto")]
    res = {}
    for name, sg, dm, lds, split in modes:
        out, gs = run(sg, dm, lds, split)
        assert (out.double() - ref_out.detach()).abs().max().item() <= 1e-5 * max(1.0, ref_out.abs().max().item()), (name, "forward")
        for t, (g, r) in enumerate(zip(gs, ref_g)):
            err = (g.double() - r).abs().max().item()
            # an fp32 sum of n_max terms: every addition rounds at eps x |running sum| -- a random walk over the additions (the atomic mode adds in
            # any order: a 2-row table hit a million times is a sum of a million terms); 50 standard deviations allowed on top of the relative bound
            rmax = max(1.0, r.abs().max().item())
            assert err <= 2e-5 * rmax + 50 * 6e-8 * rmax * n_max[t] ** 0.5, dict(mode=name, table=t, err=err, B=B, nf=nf, rows=trows[t], dim=tdim[t], n_max=n_max[t], rmax=rmax)
        res[name] = gs
    for a, b in (("row-sparse", "row-sparse, sorted planner, split"), ("row-sparse", "dense planned")):      # the planned reductions agree word for word
        for g0, g1 in zip(res[a], res[b]):
            assert torch.equal(g0.view(torch.int32), g1.view(torch.int32)), (a, b, dict(B=B, nf=nf, nt=nt, mixed=mixed))
    n_done += 1
ops.DENSE_BWD_SORTED, ops.PLAN_LDS, ops.PAD_SPLIT = None, "auto", "auto"
print(f"stress_embed_bwd: {n_done} random launches x 6 backward modes: forward and gradients within tolerance of float64, the planned modes word for word equal ({time.time() - t0:.0f} s)")
