"""PreparedShardedStep (news_recsys_amd/shard_step.py) at world 2 and 3 with the PRODUCT kernels: the rank processes share cuda:0 and exchange
through gloo with host-staged buffers (RowShardedEmbedding(host_staged=True) -- a test transport: RCCL refuses two ranks on one device).
Everything else is what an 8-GPU node runs: nrx_route_feat for W > 1 owners, the equal-split exchanges, nrx_inbox_transpose, the owner's fused
forward over its pseudo-batch, the slot-addressed final launch, nrx_embed_bwd_scatter, the owner-side planned reduction.

Truth = the DIRECT (unsharded) bound path over the rank-major concatenation of the ranks' batches, on full tables, in the parent process:
  * every rank's forward concat (and FM logit) equals its rows of the direct result bit for bit (row copies; the FM epilogue sums a sample's
    fields in the same order);
  * the UNION of the ranks' (key, value) sets equals the direct row-sparse gradient: same set of (table, global row) keys, every value bit for
    bit -- an owner adds a row's lookups in (feature, source rank, sample) order, which is the direct reduction's order on the concatenation;
  * two runs of the sharded step give the same bits (no atomics anywhere on the path).
No reference counterpart (single-device reference: src/model/sort/deep/train.py:38-44); the arithmetic is autograd of
src/model/BaseModel/base_model.py:262-308."""
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from news_recsys_amd import ops, shard_step
from news_recsys_amd._lib import NRX_SPARSE
from news_recsys_amd.sharding import RowShardedEmbedding, ShardedFeature
from tests.test_sharding_gloo import _free_port

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SPECS = {
    # name -> (features: (name, table, dim, rows), per-rank batch, FM epilogue)
    "fm16": ([(f"C{i:02d}", f"C{i:02d}", 16, 40_000 + 997 * i) for i in range(7)], 3000, True),
    "mixed": ([("a", "a", 16, 5000), ("b", "b", 32, 70_000), ("item_id", "item_id", 16, 9000), ("last_click", "item_id", 16, 9000),
               ("tiny", "tiny", 32, 5)], 2100, False),
}


def _full_tables(spec):
    rng = np.random.default_rng(11)
    tabs = {}
    for _, t, d, r in spec:
        if t not in tabs:
            x = rng.standard_normal((r, d)).astype(np.float32)
            x[0] = 0
            tabs[t] = x
    return tabs


def _batch(spec, B, rank):
    rng = np.random.default_rng(500 + rank)
    ids = []
    for _, t, d, r in sorted(spec):
        x = rng.integers(0, r, B)
        x[: 4] = 0                                        # padding ids on every rank
        if r > 1000:
            x[rng.random(B) < 0.05] = 17                  # a hot row: looked up by every rank (cross-source summation order matters)
        ids.append(x)
    width = sum(d for _, _, d, _ in spec)
    return ids, rng.standard_normal((B, width)).astype(np.float32), rng.standard_normal((B,)).astype(np.float32)


def _worker(rank, world, port, q, case, one_sided, direct_grad, fail_map=False):
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if fail_map:
        os.environ["NRX_DEBUG_FAIL_PEER_MAP"] = "1"           # rank 1 pretends it cannot map its peers' buffers: EVERY rank must fall back
        import warnings
        warnings.simplefilter("ignore")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import _poison
    _poison.poison()          # (NRX_TEST_POISON=1: this rank's buffers start from 0xFF bytes)
    try:
        spec, B, fm = SPECS[case]
        tabs = _full_tables(spec)
        arenas = {t: shard_step.make_arena(x.shape[0], x.shape[1], rank, world, DEV, full=torch.from_numpy(x).to(DEV)) for t, x in tabs.items()}
        feats = [ShardedFeature(nm, NRX_SPARSE, t, d, 0, False, fm) for nm, t, d, _ in sorted(spec)]
        ids, up, up_fm = _batch(spec, B, rank)
        inputs = [torch.from_numpy(x).to(DEV) for x in ids]
        g_out = torch.from_numpy(up).to(DEV)
        g_fm = torch.from_numpy(up_fm).to(DEV) if fm else None
        eng = RowShardedEmbedding(rank, world, slack=0.5, host_staged=True, overflow_policy="defer")
        step = shard_step.PreparedShardedStep(eng, feats, inputs, [None] * len(feats), arenas, one_sided=one_sided).bind_backward(g_out, g_fm, direct_grad=direct_grad)
        assert all(g["placed"] == (one_sided and not fail_map) for g in step.groups) and all(b["direct"] == (direct_grad and not fail_map) for b in step.bwd)
        runs = []
        for _ in range(2):
            out, _, fmv = step.run()
            entries = step.backward()
            torch.cuda.synchronize()
            dist.barrier()                                # (one-sided: every peer's placing launch has finished before anyone reads its buffer)
            got = []
            for e in entries:
                nu = int(e["counts"][0])
                names = [next(n for n, a in arenas.items() if a is t) for t in e["tables"]]
                got.append((names, e["dim"], e["uniq"][:nu].cpu().numpy(), e["values"][:nu].cpu().numpy()))
            runs.append((out.cpu().numpy().copy(), None if fmv is None else fmv.cpu().numpy().copy(), got))
        over = step.overflowed()
        same = np.array_equal(runs[0][0], runs[1][0]) and all(
            np.array_equal(a[2], b[2]) and np.array_equal(a[3].view(np.int32), b[3].view(np.int32)) for a, b in zip(runs[0][2], runs[1][2]))
        q.put((rank, runs[1][0], runs[1][1], runs[1][2], bool(over), bool(same)))
        dist.barrier()                                    # nobody unmaps a buffer a peer may still be writing
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case,one_sided,direct_grad,fail_map", [("fm16", False, False, False), ("mixed", False, False, False), ("mixed", True, False, False),
                                                                 ("fm16", False, True, False), ("mixed", True, True, False), ("fm16", True, True, False),
                                                                 ("fm16", True, True, True)])
def test_sharded_step_equals_the_direct_path_on_the_concatenated_batch(world, case, one_sided, direct_grad, fail_map):
    """one_sided: the owners write the rows straight into the requesters' concat buffers (nrx_gather_place_feat; the rank processes map each
    other's buffers through hipIpc) -- same outputs, same gradients.  direct_grad: the requesters write the gradient rows that need no reduction
    straight into the owners' values[] (the owners' plans came back first; the arenas are mapped the same way) -- same (keys, values)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    # (fail_map: one rank cannot map its peers -- PeerMappingError is raised on every rank together and all take the all-to-all forms: same results)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, case, one_sided, direct_grad, fail_map)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        item = q.get(timeout=300)
        res[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    spec, B, fm = SPECS[case]
    tabs = _full_tables(spec)
    names = sorted(tabs)
    # ---- the direct path on the rank-major concatenation
    batches = [_batch(spec, B, r) for r in range(world)]
    inputs = [torch.from_numpy(np.concatenate([b[0][k] for b in batches])).to(DEV) for k in range(len(spec))]
    g_out = torch.from_numpy(np.concatenate([b[1] for b in batches])).to(DEV)
    g_fm = torch.from_numpy(np.concatenate([b[2] for b in batches])).to(DEV) if fm else None
    slots, col = [], 0
    for nm, t, d, _ in sorted(spec):
        slots.append(ops.Slot(nm, NRX_SPARSE, names.index(t), d, 0, col, fm_field=int(fm)))
        col += d
    plan = ops.EmbedPlan(slots, out_width=col, use_fm=fm)
    sums = torch.empty((world * B, 16), dtype=torch.float32, device=DEV) if fm else None
    fwd = ops.PreparedEmbed(plan, [torch.from_numpy(tabs[t]).to(DEV) for t in names], inputs, [None] * len(spec), fm_sums=sums)
    d_out, _, d_fm = fwd.run()
    d_groups = ops.PreparedSparseBackward(fwd, g_out, g_fm).run()
    torch.cuda.synchronize()
    want = {}
    for g in d_groups:
        nu = int(g["counts"][0])
        for k, v in zip(g["uniq"][:nu].cpu().numpy(), g["values"][:nu].cpu().numpy()):
            if k & ((1 << 40) - 1):                                  # (the padding row's zero entry has no counterpart: owner id 0 is never keyed per table)
                want[(names[k >> 40], int(k & ((1 << 40) - 1)))] = v
    got = {}
    for r in range(world):
        out, fmv, entries, over, same = res[r]
        assert not over and same
        assert np.array_equal(out, d_out[r * B:(r + 1) * B].cpu().numpy())
        if fm and not one_sided:
            assert np.array_equal(fmv, d_fm[r * B:(r + 1) * B].cpu().numpy())
        elif fm:
            np.testing.assert_allclose(fmv, d_fm[r * B:(r + 1) * B].cpu().numpy(), rtol=1e-5, atol=1e-5 * float(d_fm.abs().max()))
        for tnames, dim, keys, vals in entries:
            for k, v in zip(keys, vals):
                row = int(k & ((1 << 40) - 1))
                if row == 0:                                         # the arena's dummy row (empty slots, padding ids): zeros, never trained
                    assert not v.any()
                    continue
                key = (tnames[k >> 40], (row - 1) * world + r)
                assert key not in got
                got[key] = v
    assert set(got) == set(want)
    for key, v in want.items():
        assert np.array_equal(got[key].view(np.int32), v.view(np.int32)), key


# ---------------------------------------------------------------------------------------------- the tower with a history bag (pooled channel)
def _tower_batch(rank, B, L, news, users):
    rng = np.random.default_rng(900 + rank)
    hist = rng.integers(1, news, (B, L))
    hist[rng.random((B, L)) < 0.1] = 23                    # a hot news row in many bags of every rank
    lens = rng.integers(0, L + 1, B)
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.float32)
    hist = np.where(mask > 0, hist, 0)
    return rng.integers(0, news, B), hist, mask, rng.integers(1, users, B), rng.standard_normal((B, 48)).astype(np.float32)


def _tower_worker(rank, world, port, q, binary):
    import os
    from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import _poison
    _poison.poison()          # (NRX_TEST_POISON=1: this rank's buffers start from 0xFF bytes)
    try:
        D, L, B, news, users = 16, 9, 2500, 6000, 50_000
        rng = np.random.default_rng(3)
        tabs = {"item_id": rng.standard_normal((news, D)).astype(np.float32), "user_id": rng.standard_normal((users, D)).astype(np.float32)}
        for t in tabs.values():
            t[0] = 0
        arenas = {t: shard_step.make_arena(x.shape[0], D, rank, world, DEV, full=torch.from_numpy(x).to(DEV)) for t, x in tabs.items()}
        feats = [ShardedFeature("item_id", NRX_SPARSE, "item_id", D), ShardedFeature("user_history", NRX_BAG_MASKED_MEAN, "item_id", D, L),
                 ShardedFeature("user_id", NRX_SPARSE, "user_id", D)]
        item, hist, mask, user, up = _tower_batch(rank, B, L, news, users)
        inputs = [torch.from_numpy(item).to(DEV), torch.from_numpy(hist).to(DEV), torch.from_numpy(user).to(DEV)]
        weights = [None, torch.from_numpy(mask).to(DEV), None]
        eng = RowShardedEmbedding(rank, world, slack=0.5, host_staged=True, overflow_policy="defer")
        step = shard_step.PreparedShardedStep(eng, feats, inputs, weights, arenas, one_sided=False, binary_masks=binary).bind_backward(torch.from_numpy(up).to(DEV))
        res = []
        for _ in range(2):
            out, _, _ = step.run()
            entries = step.backward()
            torch.cuda.synchronize()
            dist.barrier()
            got = []
            for e in entries:
                nu = int(e["counts"][0])
                names = [next(n for n, a in arenas.items() if a is t) for t in e["tables"]]
                got.append((names, e["uniq"][:nu].cpu().numpy(), e["values"][:nu].cpu().numpy()))
            res.append((out.cpu().numpy().copy(), got))
        same = np.array_equal(res[0][0], res[1][0]) and all(np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.int32), b[2].view(np.int32))
                                                           for a, b in zip(res[0][1], res[1][1]))
        q.put((rank, res[1][0], res[1][1], bool(step.overflowed()), bool(same)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,binary", [(2, False), (3, False), (2, True), (3, True)])
def test_sharded_tower_with_a_pooled_history_bag(world, binary):
    """The DSSM tower at world 2 / 3: the history bag is pooled at the owners (forward) and its gradient comes back as row-sparse (keys, values)
    from the owners' planned reduction (backward).  Against the direct path on the concatenated batch: single-valued columns bit for bit, the
    pooled columns rtol 1e-6; per (table, global row) the sum over ranks and lists equals the direct gradient to fp32 summation tolerance; two
    runs of the sharded step give the same bits."""
    from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tower_worker, args=(r, world, port, q, binary)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        item = q.get(timeout=300)
        res[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    D, L, B, news, users = 16, 9, 2500, 6000, 50_000
    rng = np.random.default_rng(3)
    tabs = {"item_id": rng.standard_normal((news, D)).astype(np.float32), "user_id": rng.standard_normal((users, D)).astype(np.float32)}
    for t in tabs.values():
        t[0] = 0
    bs = [_tower_batch(r, B, L, news, users) for r in range(world)]
    inputs = [torch.from_numpy(np.concatenate([b[0] for b in bs])).to(DEV), torch.from_numpy(np.concatenate([b[1] for b in bs])).to(DEV),
              torch.from_numpy(np.concatenate([b[3] for b in bs])).to(DEV)]
    weights = [None, torch.from_numpy(np.concatenate([b[2] for b in bs])).to(DEV), None]
    g_out = torch.from_numpy(np.concatenate([b[4] for b in bs])).to(DEV)
    slots = [ops.Slot("item_id", NRX_SPARSE, 0, D, 0, 0), ops.Slot("user_history", NRX_BAG_MASKED_MEAN, 0, D, L, D), ops.Slot("user_id", NRX_SPARSE, 1, D, 0, 2 * D)]
    fwd = ops.PreparedEmbed(ops.EmbedPlan(slots, out_width=3 * D), [torch.from_numpy(tabs["item_id"]).to(DEV), torch.from_numpy(tabs["user_id"]).to(DEV)],
                            inputs, weights)
    d_out = fwd.run()[0].cpu().numpy()
    d_groups = ops.PreparedSparseBackward(fwd, g_out).run()
    torch.cuda.synchronize()
    got = {"item_id": {}, "user_id": {}}
    for r in range(world):
        out, entries, over, same = res[r]
        assert not over and same
        want = d_out[r * B:(r + 1) * B]
        assert np.array_equal(out[:, :D], want[:, :D]) and np.array_equal(out[:, 2 * D:], want[:, 2 * D:])
        np.testing.assert_allclose(out[:, D:2 * D], want[:, D:2 * D], rtol=1e-6, atol=1e-6)
        for tnames, keys, vals in entries:
            for k, v in zip(keys, vals):
                row = int(k & ((1 << 40) - 1))
                if row == 0:
                    assert not v.any()
                    continue
                d = got[tnames[k >> 40]]
                key = (row - 1) * world + r
                d[key] = d.get(key, 0) + v.astype(np.float64)
    names = ["item_id", "user_id"]
    n_want = {n: 0 for n in names}
    for g in d_groups:
        nu = int(g["counts"][0])
        keys, vals = g["uniq"][:nu].cpu().numpy(), g["values"][:nu].double().cpu().numpy()
        scale = float(np.abs(vals).max())
        for k, v in zip(keys, vals):
            r = int(k & ((1 << 40) - 1))
            if r:
                n_want[names[k >> 40]] += 1
                np.testing.assert_allclose(got[names[k >> 40]][r], v, rtol=1e-5, atol=2e-6 * scale)
    assert all(n_want[n] == len(got[n]) for n in names)


# ---------------------------------------------------------------------------------------------- a whole model on two ranks
def _model_worker(rank, world, port, q, cls_name, cfg, gname):
    import os
    from news_recsys_amd import sharding
    from news_recsys_amd.model.sort.deep.model import Deep
    from news_recsys_amd.model.sort.fm.model import FM
    from tests.conftest import CONFIGS, GOLDEN
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import _poison
    _poison.poison()          # (NRX_TEST_POISON=1: this rank's buffers start from 0xFF bytes)
    try:
        g = dict(np.load(os.path.join(GOLDEN, gname + ".npz"), allow_pickle=False))
        m = {"Deep": Deep, "FM": FM}[cls_name](os.path.join(CONFIGS, cfg))
        m.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param/")}, strict=True)
        m = m.to(DEV)
        shard_step.shard_model_step_(m, rank, world, host_staged=True, slack=1.0)
        full = {k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("batch/")}
        n = next(iter(full.values())).shape[0] // world
        batch = {k: v[rank * n:(rank + 1) * n].contiguous().to(DEV) for k, v in full.items()}
        opt = m.configure_optimizers()["optimizer"]
        for _ in range(2):
            opt.zero_grad()
            loss = m.bceLoss(m(batch), batch["label"][:, 0])
            loss.backward()
            dense = sharding.data_parallel_params(m)
            grads = [p.grad for p in dense if p.grad is not None]
            flat = torch.cat([x.reshape(-1) for x in grads]).cpu()          # (test transport: the dense all-reduce through the host)
            dist.all_reduce(flat)
            flat /= world
            off = 0
            for x in grads:
                x.copy_(flat[off:off + x.numel()].view_as(x))
                off += x.numel()
            opt.step()
        torch.cuda.synchronize()
        dist.barrier()
        q.put((rank, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cls_name,cfg,gname", [("Deep", "cf_array_small.yaml", "model_deep_array"), ("FM", "cf_fm_small.yaml", "model_fm")])
def test_bound_sharded_model_on_two_ranks_trains_like_the_unsharded_model(cls_name, cfg, gname):
    """shard_model_step_ at world 2 (every rank takes half of the golden batch; tables see the gradient of the global-batch mean: grad_average)
    against the unsharded `sparse_grad: fused` model on the whole batch: after two optimizer steps the arenas hold the unsharded tables' rows
    (rank::2) and the dense parameters agree (rtol 1e-5: the halves' losses are averaged in another order than one mean over the batch)."""
    import os
    from news_recsys_amd.model.sort.deep.model import Deep
    from news_recsys_amd.model.sort.fm.model import FM
    from tests.conftest import CONFIGS, GOLDEN
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_model_worker, args=(r, world, port, q, cls_name, cfg, gname)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        item = q.get(timeout=300)
        res[item[0]] = item[1]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = dict(np.load(os.path.join(GOLDEN, gname + ".npz"), allow_pickle=False))
    ref = {"Deep": Deep, "FM": FM}[cls_name](os.path.join(CONFIGS, cfg))
    ref.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param/")}, strict=True)
    ref = ref.to(DEV)
    ref.sparse_grad = "fused"
    full = {k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("batch/")}
    n = next(iter(full.values())).shape[0] // world * world
    batch = {k: v[:n].to(DEV) for k, v in full.items()}
    opt = ref.configure_optimizers()["optimizer"]
    for _ in range(2):
        opt.zero_grad()
        ref.bceLoss(ref(batch), batch["label"][:, 0]).backward()
        opt.step()
    want = {k: v.detach().cpu().numpy() for k, v in ref.state_dict().items()}
    for k, w in want.items():
        if k.startswith("embedding_tables."):
            for r in range(world):
                got = res[r][k][1:]                                          # the arena without its dummy row = global rows r::world
                np.testing.assert_allclose(got, w[r::world], rtol=1e-5, atol=1e-6, err_msg=f"{k} rank {r}")
                assert not res[r][k][0].any()
        else:
            for r in range(world):
                np.testing.assert_allclose(res[r][k], w, rtol=1e-5, atol=1e-6, err_msg=f"{k} rank {r}")


# ---------------------------------------------------------------------------------------------- a table with fewer rows than ranks: empty shards
def _tiny_worker(rank, world, port, q):
    import os
    from news_recsys_amd._lib import NRX_BAG_MEAN
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import _poison
    _poison.poison()
    try:
        D, L, B = 16, 2, 500
        rng = np.random.default_rng(5)
        tabs = {"tiny": rng.standard_normal((2, D)).astype(np.float32), "big": rng.standard_normal((300, D)).astype(np.float32)}
        for t in tabs.values():
            t[0] = 0
        arenas = {t: shard_step.make_arena(x.shape[0], D, rank, world, DEV, full=torch.from_numpy(x).to(DEV)) for t, x in tabs.items()}
        assert arenas["tiny"].shape[0] == (2 if rank < 2 else 1)            # rank 2 owns no row of the 2-row table: its arena is the dummy row alone
        feats = [ShardedFeature("a", NRX_SPARSE, "big", D), ShardedFeature("cats", NRX_BAG_MEAN, "tiny", D, L)]
        r2 = np.random.default_rng(50 + rank)
        inputs = [torch.from_numpy(r2.integers(0, 300, B)).to(DEV), torch.from_numpy(r2.integers(0, 2, (B, L))).to(DEV)]
        up = torch.from_numpy(r2.standard_normal((B, 2 * D)).astype(np.float32)).to(DEV)
        eng = RowShardedEmbedding(rank, world, slack=3.0, host_staged=True, overflow_policy="defer")
        step = shard_step.PreparedShardedStep(eng, feats, inputs, [None, None], arenas, one_sided=False, binary_masks=True).bind_backward(up)
        out, _, _ = step.run()
        entries = step.backward()
        torch.cuda.synchronize()
        dist.barrier()
        step.check()
        got = {}
        for e in entries:
            nu = int(e["counts"][0])
            for ti, arena in enumerate(e["tables"]):
                name = next(n for n, a in arenas.items() if a is arena)
                keys, vals = e["uniq"][:nu].cpu().numpy(), e["values"][:nu].cpu().numpy()
                for k, v in zip(keys, vals):
                    if (k >> 40) == ti and (k & ((1 << 40) - 1)) > 0:
                        key = (name, int((k & ((1 << 40) - 1)) - 1) * world + rank)
                        got[key] = got.get(key, 0) + v.astype(np.float64)
        q.put((rank, out.cpu().numpy(), got))
    finally:
        dist.destroy_process_group()


def test_a_table_with_fewer_rows_than_ranks_leaves_an_empty_shard_that_works():
    """World 3, a bag feature over a 2-row table (a tiny category table on many ranks): rank 2's shard of it is EMPTY -- its pooled channel handed
    the pooling launch a null table (found by tests/stress_shard_step_multirank.py).  Forward: single-valued columns bit for bit, the mean-pooled
    columns rtol 1e-6 against torch on the full tables; gradient of the tiny table's one trainable row = the sum over all ranks' bags."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tiny_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        item = q.get(timeout=300)
        res[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    D, L, B = 16, 2, 500
    rng = np.random.default_rng(5)
    tabs = {"tiny": rng.standard_normal((2, D)).astype(np.float32), "big": rng.standard_normal((300, D)).astype(np.float32)}
    for t in tabs.values():
        t[0] = 0
    g_tiny = np.zeros(D)
    seen = {}
    for r in range(world):
        r2 = np.random.default_rng(50 + r)
        a, cats = r2.integers(0, 300, B), r2.integers(0, 2, (B, L))
        up = r2.standard_normal((B, 2 * D)).astype(np.float32)
        out, got = res[r]
        assert np.array_equal(out[:, :D], tabs["big"][a])
        np.testing.assert_allclose(out[:, D:], tabs["tiny"][cats].mean(1), rtol=1e-6, atol=1e-6)
        g_tiny += ((cats == 1).sum(1, keepdims=True) * up[:, D:].astype(np.float64) / L).sum(0)      # d mean / d row 1 = (its count in the bag) / L
        for k, v in got.items():
            assert k not in seen
            seen[k] = v
    assert ("tiny", 1) in seen and ("tiny", 0) not in seen                 # row 0 is the padding row: it never trains
    np.testing.assert_allclose(seen[("tiny", 1)], g_tiny, rtol=1e-5, atol=1e-5)
