"""world_size 2 and 3 with the PRODUCT (HIP) backend: all ranks share cuda:0 and exchange through gloo with
host-staged buffers (RowShardedEmbedding(host_staged=True), a test transport -- RCCL refuses two ranks on one device).
Everything else is the code an 8-GPU node runs: nrx_route_ids placing ids for W > 1 owners, the equal-split block
exchange, nrx_gather_inbox / nrx_scatter_add_inbox on each owner's shard, the slot-addressed final fused launch, the
overflow / out-of-range agreement.  Truth = the single-process oracle of tests/test_sharding_gloo.py."""
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from news_recsys_amd import sharding
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN
from news_recsys_amd.sharding import RowShardedEmbedding, ShardedFeature
from tests.test_sharding_gloo import FEATS, _free_port, batch_for, full_tables, oracle_forward_and_grads

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _worker(rank, world, port, q, mode, slack, replicate, prepared, dedup=False, pool_bags=True, one_sided=False):
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import _poison
    _poison.poison()          # (NRX_TEST_POISON=1: this rank's buffers start from 0xFF bytes)
    try:
        tabs = full_tables()
        shards = {n: (torch.from_numpy(t).clone() if n in replicate else sharding.shard_table(torch.from_numpy(t), rank, world)
                      ).to(DEV).requires_grad_(True) for n, t in tabs.items()}
        feats = [ShardedFeature(f.name, f.kind, f.table, f.dim, f.bag_len, f.wide, f.fm, f.table in replicate) for f in FEATS]
        eng = RowShardedEmbedding(rank, world, mode=mode, slack=slack, host_staged=True,
                                  overflow_policy="defer" if prepared else "check", dedup=dedup, pool_bags=pool_bags,
                                  grad_average=False)
        if slack < 0:
            eng.capacity_for = lambda n: 64
        b = batch_for(rank)
        inputs = [torch.from_numpy(np.asarray(b[f.name])).to(DEV) for f in FEATS]
        inputs = [x.float() if x.dtype == torch.float64 else x for x in inputs]
        weights = [torch.from_numpy(b["user_history_mask"]).to(DEV) if f.kind == NRX_BAG_MASKED_MEAN else None for f in FEATS]
        if prepared:        # the sync-free bound forward bench.py uses at N > 1
            if one_sided:
                _one_sided_worker(eng, rank, shards, replicate, q)
                return
            call = sharding.PreparedShardedForward(eng, feats, inputs, weights, {n: s.detach() for n, s in shards.items()})
            call.run()
            call.run()                                   # re-launchable: same buffers, same result
            out = call.final.out
            torch.cuda.synchronize()
            q.put((rank, out.detach().cpu().numpy(), None, bool(call.overflowed())))
            return
        out, wide, fm = eng.forward(feats, inputs, weights, shards)
        (out * torch.from_numpy(b["_up"]).to(DEV)).sum().backward()
        torch.cuda.synchronize()
        q.put((rank, out.detach().cpu().numpy(), {n: s.grad.cpu().numpy() for n, s in shards.items()}, False))
    finally:
        dist.destroy_process_group()


# The one-sided layout: the single-valued 16-wide features first (16-byte aligned columns: what the owners can place), then the bags, the
# 8-wide feature (too narrow for the placing kernel: it takes the buffer path in the same call) and the dense value.
ORDER_1S = ["item_id", "user_id", "user_history", "user_click_cats", "category", "ctr"]
COLS_SORTED = {"category": (0, 8), "ctr": (8, 9), "item_id": (9, 25), "user_click_cats": (25, 33), "user_history": (33, 49), "user_id": (49, 65)}


def _one_sided_worker(eng, rank, shards, replicate, q):
    by_name = {f.name: f for f in FEATS}
    feats = [ShardedFeature(n, by_name[n].kind, by_name[n].table, by_name[n].dim, by_name[n].bag_len, False, False, by_name[n].table in replicate)
             for n in ORDER_1S]
    b = batch_for(rank)
    inputs = [torch.from_numpy(np.asarray(b[n])).to(DEV) for n in ORDER_1S]
    inputs = [x.float() if x.dtype == torch.float64 else x for x in inputs]
    weights = [torch.from_numpy(b["user_history_mask"]).to(DEV) if f.kind == NRX_BAG_MASKED_MEAN else None for f in feats]
    width = 65
    ld = 68
    call = sharding.PreparedShardedForward(eng, feats, inputs, weights, {n: s.detach() for n, s in shards.items()}, out_ld=ld, one_sided=True)
    placed = sorted(feats[i].name for i in call.placed)
    assert call.peers is not None and placed == [n for n in ("item_id", "user_id") if by_name[n].table not in replicate], placed
    call.run()
    res = call.run()                                     # re-launchable: same buffers, same result
    torch.cuda.synchronize()
    dist.barrier()                                       # test transport: every peer's placing launch has finished before anyone reads
    torch.cuda.synchronize()
    out = res[0][:, :width].detach().cpu().numpy()
    # back to the sorted-name column order the oracle (and _check_outputs) use
    cols, c = {}, 0
    for n in ORDER_1S:
        w = COLS_SORTED[n][1] - COLS_SORTED[n][0]
        cols[n] = (c, c + w)
        c += w
    resorted = np.concatenate([out[:, cols[n][0]:cols[n][1]] for n in sorted(ORDER_1S)], axis=1)
    q.put((rank, resorted, None, bool(call.overflowed())))
    dist.barrier()                                       # nobody unmaps a buffer a peer may still be writing


def _run(world, mode, slack, replicate, prepared, dedup=False, pool_bags=True, one_sided=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode, slack, replicate, prepared, dedup, pool_bags, one_sided)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, out, grads, over = q.get(timeout=300)
        results[rank] = (out, grads, over)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return results


def _check_outputs(results, world):
    want_outs, want_grads = oracle_forward_and_grads(world)
    for r in range(world):
        out = results[r][0]
        for lo, hi in ((0, 8), (9, 25), (49, 65)):
            assert np.array_equal(out[:, lo:hi], want_outs[r][:, lo:hi])               # routed copies: bit-exact
        np.testing.assert_allclose(out[:, 8:9], want_outs[r][:, 8:9], rtol=1e-6)         # dense feature (float32 here)
        np.testing.assert_allclose(out[:, 25:49], want_outs[r][:, 25:49], rtol=1e-6, atol=1e-6)   # pooled
    return want_grads


@pytest.mark.parametrize("world,mode,slack,replicate,dedup,pool_bags",
                         [(2, "capacity", 0.5, (), False, True), (3, "capacity", 0.5, (), False, True), (2, "exact", 0.0, (), False, True),
                          (2, "capacity", -1.0, (), False, True), (2, "capacity", 0.5, ("category",), False, True),
                          (2, "capacity", 0.5, (), False, False),        # bags travel as rows (pool at the source)
                          (3, "capacity", 0.5, (), True, True),          # per-destination dedup + owner-side pooling
                          (2, "capacity", 0.5, (), True, False)])        # dedup of bag lookups too
def test_hip_backend_forward_backward_with_several_ranks(world, mode, slack, replicate, dedup, pool_bags):
    results = _run(world, mode, slack, replicate, prepared=False, dedup=dedup, pool_bags=pool_bags)
    want_grads = _check_outputs(results, world)
    for r in range(world):
        for n, g in results[r][1].items():
            if n in replicate:
                continue
            np.testing.assert_allclose(g, want_grads[n][r::world], rtol=1e-5, atol=1e-6, err_msg=f"rank {r} table {n}")
    for n in replicate:
        total = sum(results[r][1][n] for r in range(world))
        np.testing.assert_allclose(total, want_grads[n], rtol=1e-5, atol=1e-6)
    assert np.all(results[0][1]["item_id"][0] == 0)


@pytest.mark.parametrize("world,replicate", [(2, ()), (3, ("category",))])
def test_hip_backend_prepared_sync_free_forward_with_several_ranks(world, replicate):
    results = _run(world, "capacity", 0.5, replicate, prepared=True)
    _check_outputs(results, world)
    assert not any(results[r][2] for r in range(world))


@pytest.mark.parametrize("world,replicate", [(2, ()), (3, ("category",))])
def test_one_sided_placement_with_several_ranks_on_one_gpu(world, replicate):
    """ONE-SIDED PLACEMENT (PreparedShardedForward(one_sided=True)): every rank maps the other ranks' concat buffers (hipIpc through torch's
    CUDA-IPC sharing -- the rank processes share cuda:0 here, on a node each maps its peers over xGMI), ids AND their sample positions go to the
    owners, and the owner's gather (nrx_gather_inbox_place) writes each row straight into the requester's concat: no row buffer, no row
    all-to-all, no second pass.  Same outputs as the buffer path: routed single-valued features bit-exact, pooled bags / dense values as before."""
    results = _run(world, "capacity", 0.5, replicate, prepared=True, one_sided=True)
    _check_outputs(results, world)
    assert not any(results[r][2] for r in range(world))
