"""world_size-2 (and 3) CPU tests of the row-sharded path over gloo: the id routing, the three
all-to-alls and the un-permute must reproduce the single-process result bit-exactly for single-valued
features (pure copies) and to fp32 tolerance for pooled ones; the backward must deliver every rank's
gradient rows to the owning shard.  The local kernels are stood in by tests/sharding_checker.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from news_recsys_amd import sharding
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_DENSE, NRX_SPARSE
from news_recsys_amd.sharding import ShardedFeature, RowShardedEmbedding
from oracle import ref_np as R
from tests.sharding_checker import CheckerBackend

FEATS = [ShardedFeature("category", NRX_SPARSE, "category", 8), ShardedFeature("ctr", NRX_DENSE, "", 1),
         ShardedFeature("item_id", NRX_SPARSE, "item_id", 16), ShardedFeature("user_click_cats", NRX_BAG_MEAN, "category", 8, 5),
         ShardedFeature("user_history", NRX_BAG_MASKED_MEAN, "item_id", 16, 7), ShardedFeature("user_id", NRX_SPARSE, "user_id", 16)]
ROWS = {"category": 19, "item_id": 101, "user_id": 57}
B = 33


def full_tables():
    rng = np.random.default_rng(123)
    t = {n: rng.standard_normal((r, 8 if n == "category" else 16)).astype(np.float32) for n, r in ROWS.items()}
    for v in t.values():
        v[0] = 0
    return t


def batch_for(rank):
    rng = np.random.default_rng(1000 + rank)
    b = {"category": rng.integers(0, ROWS["category"], B), "item_id": rng.integers(0, ROWS["item_id"], B),
         "user_id": rng.integers(0, ROWS["user_id"], B), "ctr": rng.random(B),
         "user_click_cats": rng.integers(0, ROWS["category"], (B, 5))}
    lens = rng.integers(0, 8, B)
    lens[0] = 0
    m = (np.arange(7)[None] < lens[:, None]).astype(np.float32)
    b["user_history"] = rng.integers(1, ROWS["item_id"], (B, 7)) * m.astype(np.int64)
    b["user_history_mask"] = m
    b["_up"] = rng.standard_normal((B, 8 + 1 + 16 + 8 + 16 + 16)).astype(np.float32)
    return b


SPACE = R.FeatureSpace(["category", "item_id", "user_id"], ["ctr"], ["user_click_cats", "user_history"],
                       {"user_click_cats": "category", "user_history": "item_id"})


def oracle_forward_and_grads(world):
    """Single-process truth: per-rank outputs and the FULL dense table grads summed over all ranks."""
    tabs = full_tables()
    outs, grads = [], {n: np.zeros_like(t) for n, t in tabs.items()}
    for r in range(world):
        b = batch_for(r)
        names = {f.name for f in FEATS}
        out, dims, _, used = R.embed_concat_ex(SPACE, tabs, b, names)
        outs.append(out)
        col = 0
        for fname, d in zip(used, dims):
            up = b["_up"][:, col:col + d]
            col += d
            if fname in SPACE.dense:
                continue
            tname = R.emb_table_name(fname, SPACE.share)
            if fname in SPACE.array:
                rows_up = R.array_pool_bwd(tabs[tname][b[fname]], b.get(fname + "_mask"), up)
                grads[tname] += R.embedding_grad_dense(b[fname], rows_up, tabs[tname].shape[0])
            else:
                grads[tname] += R.embedding_grad_dense(b[fname], up, tabs[tname].shape[0])
    return outs, grads


def _worker(rank, world, port, q, mode="capacity", slack=0.5, replicate=(), pool_bags=True, dedup=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tabs = full_tables()
        shards = {n: (torch.from_numpy(t).clone() if n in replicate else sharding.shard_table(torch.from_numpy(t), rank, world)
                      ).requires_grad_(True) for n, t in tabs.items()}
        feats = [ShardedFeature(f.name, f.kind, f.table, f.dim, f.bag_len, f.wide, f.fm, f.table in replicate) for f in FEATS]
        eng = RowShardedEmbedding(rank, world, backend=CheckerBackend(), mode=mode, slack=slack, pool_bags=pool_bags, dedup=dedup,
                                  grad_average=(mode == "exact"))
        if slack < 0:          # force tiny blocks: every exchange overflows and must fall back to exact
            eng.capacity_for = lambda n: 8
        b = batch_for(rank)
        inputs = [torch.from_numpy(np.asarray(b[f.name])) for f in FEATS]
        weights = [torch.from_numpy(b["user_history_mask"]) if f.kind == NRX_BAG_MASKED_MEAN else None for f in FEATS]
        out, wide, fm = eng.forward(feats, inputs, weights, shards)
        (out * torch.from_numpy(b["_up"])).sum().backward()
        # a dense-parameter all-reduce on the side
        p = torch.nn.Parameter(torch.zeros(3))
        p.grad = torch.full((3,), float(rank + 1))
        sharding.allreduce_dense_grads([p], world)
        q.put((rank, out.detach().numpy(), {n: s.grad.numpy() for n, s in shards.items()}, p.grad.numpy()))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,mode,slack,replicate,pool_bags,dedup",
                         [(2, "capacity", 0.5, (), True, False), (3, "capacity", 0.5, (), True, False), (2, "capacity", 0.5, (), False, False),
                          (2, "exact", 0.0, (), True, False), (2, "capacity", -1.0, (), True, False),
                          (2, "capacity", 0.5, ("category",), True, False), (2, "exact", 0.0, ("category", "user_id"), True, False),
                          (3, "capacity", 0.5, (), True, True), (2, "capacity", 0.5, (), False, True), (2, "capacity", -1.0, (), False, True)])
def test_row_sharded_forward_backward_over_gloo(world, mode, slack, replicate, pool_bags, dedup):
    """capacity = sync-free fixed blocks; exact = variable splits; slack -1 = capacity forced to
    overflow, which must be detected on every rank and redone exactly.  pool_bags: the row-sharded bag features
    (user_click_cats: plain mean, user_history: masked mean) are pooled AT THE OWNER and come back as one partial
    vector per (sample, owner) -- stated tolerance for pooled columns rtol 1e-6.  dedup: per-destination de-duplication
    (every distinct (owner, table, row) travels once; duplicates share a slot) -- routed copies stay bit-exact."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode, slack, replicate, pool_bags, dedup)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, out, grads, pg = q.get(timeout=120)
        results[rank] = (out, grads, pg)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_outs, want_grads = oracle_forward_and_grads(world)
    for r in range(world):
        out, grads, pg = results[r]
        # columns: category 0:8 | ctr 8 | item_id 9:25 | user_click_cats 25:33 | user_history 33:49 | user_id 49:65
        for lo, hi in ((0, 8), (8, 9), (9, 25), (49, 65)):
            assert np.array_equal(out[:, lo:hi], want_outs[r][:, lo:hi])               # routed copies: bit-exact
        np.testing.assert_allclose(out[:, 25:49], want_outs[r][:, 25:49], rtol=1e-6, atol=1e-6)   # pooled
        # grad_average (on for the exact-mode cases): routed table grads are the gradient of the GLOBAL-batch mean, i.e.
        # the summed grads / world; the replicated tables' local grads stay unscaled (their all-reduce averages)
        scale = world if mode == "exact" else 1
        for n, g in grads.items():
            if n in replicate:
                continue            # replicated: local data-parallel grads, summed below
            want = want_grads[n][r::world] / scale
            np.testing.assert_allclose(g, want, rtol=1e-5, atol=1e-6, err_msg=f"rank {r} table {n}")
    for n in replicate:          # planner-replicated tables: the ranks' local grads sum to the full grad
        total = sum(results[r][1][n] for r in range(world))
        np.testing.assert_allclose(total, want_grads[n], rtol=1e-5, atol=1e-6, err_msg=f"replicated table {n}")
    for r in range(world):
        out, grads, pg = results[r]
        assert np.all(results[0][1]["item_id"][0] == 0)                                # global padding row: no grad
        np.testing.assert_allclose(pg, np.full(3, sum(range(1, world + 1)) / world))


def test_partition_helpers_roundtrip():
    for rows in (1, 2, 7, 64, 101):
        full = torch.arange(rows * 3, dtype=torch.float32).view(rows, 3)
        for world in (1, 2, 3, 8):
            shards = [sharding.shard_table(full, r, world) for r in range(world)]
            assert [s.shape[0] for s in shards] == [sharding.local_row_count(rows, r, world) for r in range(world)]
            assert torch.equal(sharding.unshard_tables(shards), full)
            for r in range(world):                      # row g lives on g % world at g // world
                for lr in range(shards[r].shape[0]):
                    assert shards[r][lr, 0].item() == (lr * world + r) * 3


def test_world_one_engine_is_identity_routing():
    tabs = full_tables()
    shards = {n: torch.from_numpy(t).clone().requires_grad_(True) for n, t in tabs.items()}
    eng = RowShardedEmbedding(0, 1, backend=CheckerBackend())
    b = batch_for(0)
    inputs = [torch.from_numpy(np.asarray(b[f.name])) for f in FEATS]
    weights = [torch.from_numpy(b["user_history_mask"]) if f.kind == NRX_BAG_MASKED_MEAN else None for f in FEATS]
    out, _, _ = eng.forward(FEATS, inputs, weights, shards)
    want, _ = oracle_forward_and_grads(1)
    np.testing.assert_allclose(out.detach().numpy(), want[0], rtol=1e-6, atol=1e-6)


def _ckpt_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from news_recsys_amd.model.sort.deep.model import Deep
        from tests.conftest import CONFIGS
        torch.manual_seed(7)
        m = Deep(os.path.join(CONFIGS, "cf_array_small.yaml"))
        ref = {k: v.detach().clone() for k, v in m.state_dict().items()}          # the reference-shaped checkpoint
        sharding.shard_model_(m, rank, world, backend=CheckerBackend())
        got = sharding.full_state_dict(m)
        ok_save = set(got) == set(ref) and all(torch.equal(got[k], ref[k]) for k in ref)
        # train-like change of the local shards only, then save again: the gather must see the CURRENT rows
        with torch.no_grad():
            for emb in m.embedding_tables.values():
                emb.weight.mul_(2.0)
        got2 = sharding.full_state_dict(m)
        ok_cur = all(torch.equal(got2[k], ref[k] * 2.0) for k in ref if k.startswith("embedding_tables."))
        # scatter-on-load into a fresh sharded model (other init)
        torch.manual_seed(99 + rank)
        m2 = Deep(os.path.join(CONFIGS, "cf_array_small.yaml"))
        sharding.shard_model_(m2, rank, world, backend=CheckerBackend())
        sharding.load_full_state_dict_(m2, ref)
        ok_load = True
        for name, emb in m2.embedding_tables.items():
            want = sharding.shard_table(ref[f"embedding_tables.{name}.weight"], rank, world)
            ok_load &= torch.equal(emb.weight.detach()[:want.shape[0]], want)
        ok_load &= all(torch.equal(v, ref[k]) for k, v in m2.state_dict().items() if not k.startswith("embedding_tables."))
        q.put((rank, ok_save, ok_cur, ok_load))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_checkpoint_gather_on_save_and_scatter_on_load(world):
    """A model converted by shard_model_ saves the REFERENCE's state_dict (full tables, base_model.py:531-536 loads it strictly) from
    every rank and loads one back, whatever the world size."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ckpt_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == list(range(world))
    for rank, ok_save, ok_cur, ok_load in res:
        assert ok_save, f"rank {rank}: gathered state_dict differs from the unsharded one"
        assert ok_cur, f"rank {rank}: gather did not see the updated shards"
        assert ok_load, f"rank {rank}: scatter-on-load left wrong shards"
