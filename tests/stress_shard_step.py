#!/usr/bin/env python3
"""Dev (GPU box): random towers through the BOUND SHARDED STEP (news_recsys_amd/shard_step.py) at world 1 -- 1 .. 12 single-valued features over
shared tables of one to three widths (16 / 32 / 64), zero to two bag groups (masked-mean / mean / sum bags of 1 .. 130 entries, one pooled table per
width; 0/1 masks or float weights, empty bags, padded histories), batches 1 .. 20 000, int64 / int32 ids, uniform / skewed with padding ids -- in every form
of the step (one tower in four is an FM plan: fields of one width, the logit and its gradient checked too): one-sided placement on / off, the requester's pack as the owner's placement pass on / off, the pooled channel's three routings
(NRX_ROUTE_BAGS = runs | one | legacy), binary-mask fast path on / off, exchange groups side by side or one after the other.
Checked against a float64 restatement in torch (F.embedding + the pooling of src/model/BaseModel/base_model.py:262-282, autograd for the gradients):
the single-valued columns of the concat bit for bit, the pooled columns and every table's gradient (the sum of the step's (key, value) lists) within
the fp32 summation tolerance; two runs of a form word for word equal; all forms of one tower word for word equal in their single-valued columns.
usage: python tests/stress_shard_step.py [seconds=120] [seed=1]   (a checker like the tests next to it; not collected by pytest)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _poison
from news_recsys_amd import ops, shard_step
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM, NRX_SPARSE
from news_recsys_amd.sharding import RowShardedEmbedding, ShardedFeature
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
DEV = "cuda:0"
t0, n_done, n_forms = time.time(), 0, 0
while time.time() - t0 < budget:
    _poison.poison()
    B = int(rng.choice([1, 3, 64, 81, 700, 4097, 9000, 20000]))
    idt = torch.int64 if rng.integers(0, 3) else torch.int32             # (every feature of a tower: the ids of an exchange group share a dtype)
    fm = bool(rng.integers(0, 4) == 0)                                   # an FM tower: single-valued fields of ONE width, the logit + its gradient too
    dims = sorted(set(int(d) for d in rng.choice([16, 32, 64], 1 if fm else int(rng.integers(1, 4)))))
    tables, feats, ins, ws, look = {}, [], [], [], 0
    for d in dims:
        for t in range(int(rng.integers(1, 4))):
            tables[f"t{d}_{t}"] = (int(rng.choice([2, 50, 3000, 200000, 1500000])), d)
    names = list(tables)
    for f in range(int(rng.integers(1, 13))):
        t = names[int(rng.integers(0, len(names)))]
        rows, d = tables[t]
        skew = rng.integers(0, 3) == 0
        x = rng.integers(0, rows, B) if not skew else np.minimum(rng.zipf(1.3, B) - 1, rows - 1)
        feats.append(ShardedFeature(f"s{f}", NRX_SPARSE, t, d, 0, False, fm))
        ins.append(torch.from_numpy(np.asarray(x, np.int64)).to(DEV).to(idt))
        ws.append(None)
        look += B
    binary_ok = True
    for d in dims:
        if fm or rng.integers(0, 2) == 0:
            continue
        bag_table = [n for n in names if tables[n][1] == d][0]          # the bag features of one pooled group share ONE table
        rows = tables[bag_table][0]
        for f in range(int(rng.integers(1, 4))):
            kind = int(rng.choice([NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM]))
            L = int(rng.choice([1, 2, 4, 17, 50, 81, 130]))
            x = rng.integers(0, rows, (B, L))
            w = None
            if kind != NRX_BAG_MEAN:
                n_valid = rng.integers(0, L + 1, (B, 1))
                m = (np.arange(L)[None, :] < n_valid).astype(np.float32)
                x = x * m.astype(np.int64)                  # padded histories: id 0 behind the valid entries
                if kind == NRX_BAG_SUM and rng.integers(0, 2):
                    m = m * rng.random((B, L)).astype(np.float32)
                    binary_ok = False
                w = m
            feats.append(ShardedFeature(f"b{d}_{f}", kind, bag_table, d, L))
            ins.append(torch.from_numpy(np.asarray(x, np.int64)).to(DEV).to(idt))
            ws.append(None if w is None else torch.from_numpy(np.asarray(w, np.float32)).to(DEV))
            look += B * L
    if look > 1_200_000:
        continue
    gen = torch.Generator(device=DEV).manual_seed(int(rng.integers(0, 1 << 30)))
    arenas = {n: shard_step.make_arena(r, d, 0, 1, DEV, generator=gen) for n, (r, d) in tables.items()}
    width = sum(f.dim for f in feats)
    up = torch.randn(B, width, device=DEV, generator=gen)
    up_fm = torch.randn(B, device=DEV, generator=gen) if fm else None
    # ---- float64 restatement
    t64 = {n: shard_step.arena_shard(a).double().requires_grad_() for n, a in arenas.items()}
    outs, col, cols = [], 0, []
    for f, x, w in zip(feats, ins, ws):
        e = torch.nn.functional.embedding(x.long(), t64[f.table])
        if f.kind == NRX_SPARSE:
            outs.append(e)
        elif f.kind == NRX_BAG_MASKED_MEAN:
            wd = w.double()
            outs.append((e * wd.unsqueeze(-1)).sum(1) / (wd.sum(1, keepdim=True) + 1e-8))
        elif f.kind == NRX_BAG_MEAN:
            outs.append(e.mean(1))
        else:
            outs.append((e * w.double().unsqueeze(-1)).sum(1) if w is not None else e.sum(1))
        cols.append((col, f.dim, f.kind))
        col += f.dim
    ref_out = torch.cat(outs, 1)
    ref_fm = None
    if fm:      # sort/fm/model.py:18-26,48-59: column 0 of every field = its first-order weight, the rest its factor vector
        e3 = torch.stack(outs, 1)                                            # [B, fields, D]
        v = e3[:, :, 1:]
        ref_fm = e3[:, :, 0].sum(1) + 0.5 * ((v.sum(1) ** 2) - (v ** 2).sum(1)).sum(1)
    loss = (ref_out * up.double()).sum() + ((ref_fm * up_fm.double()).sum() if fm else 0.0)
    ref_g = dict(zip(t64, torch.autograd.grad(loss, list(t64.values()), allow_unused=True)))
    n_max = {n: 1 for n in tables}
    for f, x in zip(feats, ins):
        v = x.reshape(-1).long()
        c = torch.bincount(v[v > 0], minlength=1)
        n_max[f.table] += int(c.max().item()) if c.numel() else 0
    first_out = None
    forms = [(os_, dg, rb, bn, ov) for os_ in (False, True) for dg in (False, True) for rb in ("runs", "one", "legacy") for bn in (False, True) for ov in ("1", "0")]
    for k in rng.permutation(len(forms))[:4]:
        one_sided, direct, route, binary, overlap = forms[int(k)]
        binary = binary and binary_ok
        os.environ["NRX_ROUTE_BAGS"], os.environ["NRX_SHARD_OVERLAP"] = route, overlap
        eng = RowShardedEmbedding(0, 1, overflow_policy="defer")
        step = shard_step.PreparedShardedStep(eng, feats, ins, ws, arenas, one_sided=one_sided, binary_masks=binary, check_index=True)
        step.bind_backward(up, up_fm, direct_grad=direct)
        what = dict(B=B, feats=[(f.name, f.kind, f.table, f.dim, f.bag_len) for f in feats], tables=tables, one_sided=one_sided, direct=direct, route=route,
                    binary=binary, overlap=overlap)
        runs = []
        for _ in range(2):
            out, _, fmv = step.run()
            entries = step.backward()
            torch.cuda.synchronize()
            if fm:
                assert (fmv.double() - ref_fm.detach()).abs().max().item() <= 2e-5 * max(1.0, ref_fm.abs().max().item()) + 1e-5 * len(feats), ("FM logit", B, len(feats))
            runs.append((out.clone(), [(e["uniq"][:int(e["counts"][0])].clone(), e["values"][:int(e["counts"][0])].clone()) for e in entries]))
        step.check()
        assert torch.equal(runs[0][0].view(torch.int32), runs[1][0].view(torch.int32)), ("two runs, forward", what)
        for (k0, v0), (k1, v1) in zip(runs[0][1], runs[1][1]):
            assert torch.equal(k0, k1) and torch.equal(v0.view(torch.int32), v1.view(torch.int32)), ("two runs, gradient", what)
        out = runs[0][0]
        for c0, d, kind in cols:
            got, want = out[:, c0:c0 + d], ref_out[:, c0:c0 + d].detach()
            if kind == NRX_SPARSE:
                assert torch.equal(got, want.float()), ("single-valued columns", what)
            else:
                assert (got.double() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()), ("pooled columns", what)
        if first_out is None:
            first_out = out
        for c0, d, kind in cols:
            if kind == NRX_SPARSE:
                assert torch.equal(first_out[:, c0:c0 + d], out[:, c0:c0 + d])
        got_g = {n: torch.zeros_like(t) for n, t in t64.items()}
        for e, (keys, vals) in zip(entries, runs[0][1]):
            for ti, arena in enumerate(e["tables"]):
                name = next(n for n, a in arenas.items() if a is arena)
                sel = (keys >> 40) == ti
                rows = keys[sel] & ((1 << 40) - 1)
                live = rows > 0                                            # (arena row 0: the dummy row)
                got_g[name].index_add_(0, rows[live] - 1, vals[sel][live].double())
        for n in tables:
            r = ref_g[n] if ref_g[n] is not None else torch.zeros_like(t64[n])
            r = r.clone()
            r[0] = 0                                                       # the padding row never trains (padding_idx = 0, base_model.py:164)
            assert got_g[n][0].abs().max().item() == 0, ("padding row", what)
            err = (got_g[n] - r).abs().max().item()
            rmax = max(1.0, r.abs().max().item())
            assert err <= 2e-5 * rmax + 50 * 6e-8 * rmax * n_max[n] ** 0.5, dict(table=n, err=err, n_max=n_max[n], rmax=rmax, **what)
        n_forms += 1
    n_done += 1
print(f"stress_shard_step: {n_done} random towers x 4 of the step's 48 forms ({n_forms} steps bound): single-valued columns bit for bit, pooled columns and "
      f"gradients within tolerance of float64, two runs word for word equal ({time.time() - t0:.0f} s)")
