"""The bound sharded training step (news_recsys_amd/shard_step.py) and its new kernels on one GPU.

  * nrx_route_feat / nrx_inbox_transpose against their definitions (oracle/ref_np.py route_feat, owner_ids_from_inbox): integer work, bit-exact --
    worlds 1 .. 8 and 64, int32 / int64 ids, ids that cannot be rows, the padding id, blocks that overflow their capacity, repeated launches on
    one state block (the chain re-arms itself);
  * nrx_embed_bwd_scatter against "values[dest[p]] = upstream row of lookup p" with and without the FM term (fp32, value-exact: copies and one
    fused multiply-add chain restated in numpy), dest < 0 skipped;
  * PreparedShardedStep at world 1 against the DIRECT path on the same tables: forward concat and FM logit bit for bit; the row-sparse gradient
    (keys, values, counts) bit for bit with keys shifted by the arena's dummy row; one FusedSparseAdam step leaves the same weights.
    (world 2 / 3 with rank processes sharing the GPU: tests/test_shard_step_multirank_one_gpu.py.)
No reference counterpart: the reference is single-device (src/model/sort/deep/train.py:38-44); the arithmetic checked is that of
src/model/BaseModel/base_model.py:262-308 and its autograd."""
import ctypes as C

import numpy as np
import pytest
import torch

from news_recsys_amd import _lib, ops, shard_step
from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM, NRX_SPARSE, NrxFmGrad
from news_recsys_amd.sharding import RowShardedEmbedding, ShardedFeature
from oracle import ref_np

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _route(ids, world, capf, want_pos=True, state=None, overflow=None):
    lib = _lib.load()
    n, B = len(ids), ids[0].numel()
    dev = ids[0].device
    send = torch.full((world, n, capf), -7, dtype=torch.int32, device=dev)
    pos = torch.full((world, n, capf), -7, dtype=torch.int32, device=dev) if want_pos else None
    slot = torch.full((n, B), -9, dtype=torch.int32, device=dev)
    counts = torch.full((world, n), -1, dtype=torch.int64, device=dev)
    overflow = torch.zeros(1, dtype=torch.int64, device=dev) if overflow is None else overflow
    if state is None:
        state = torch.zeros(lib.nrx_route_feat_state_bytes(n, B, world), dtype=torch.uint8, device=dev)
    ptrs = (C.c_void_p * n)(*[x.data_ptr() for x in ids])
    ops.check(lib.nrx_route_feat(ptrs, n, B, ids[0].element_size() * 8, world, capf, send.data_ptr(), None if pos is None else pos.data_ptr(),
                                 slot.data_ptr(), counts.data_ptr(), overflow.data_ptr(), state.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream), "nrx_route_feat")
    torch.cuda.synchronize()
    return send, pos, slot, counts, overflow, state


@pytest.mark.parametrize("world", [1, 2, 3, 5, 8, 64])
@pytest.mark.parametrize("dt", [torch.int64, torch.int32])
def test_route_feat_equals_its_definition(world, dt):
    rng = np.random.default_rng(100 + world)
    for n, B, rows in ((1, 1, 50), (3, 777, 90), (5, 4096, 100_000), (4, 9001, 1 << 20), (26, 12_345, 1_000_000)):
        ids_np = [rng.integers(0, rows, B) for _ in range(n)]
        ids_np[0][: min(B, 3)] = 0                                     # the padding id
        if dt is torch.int64 and B > 10:
            ids_np[-1][5] = -4                                         # cannot be rows: rank 0, reported there
            ids_np[-1][6] = (1 << 31) + 5
            ids_np[-1][7] = (1 << 31) - 1
        if dt is torch.int32 and B > 10:
            ids_np[-1][5] = -4
        capf = B if world == 1 else int(B / world * 1.3) + 64
        ids = [torch.from_numpy(x).to(DEV).to(dt) for x in ids_np]
        send, pos, slot, counts, overflow, _ = _route(ids, world, capf)
        w_send, w_pos, w_slot, w_counts, w_max = ref_np.route_feat(ids_np, world, capf)
        assert np.array_equal(counts.cpu().numpy(), w_counts)
        assert int(overflow.item()) == w_max
        assert np.array_equal(send.cpu().numpy(), w_send)
        assert np.array_equal(pos.cpu().numpy(), w_pos)
        assert np.array_equal(slot.cpu().numpy(), w_slot)


def test_route_feat_overflow_repeated_launches_and_skew():
    """Blocks that exceed capf drop their surplus (slot = -1) and raise the running maximum; the same state block serves launch after launch."""
    rng = np.random.default_rng(5)
    world, n, B, capf = 4, 6, 20_000, 5_200
    state = None
    overflow = torch.zeros(1, dtype=torch.int64, device=DEV)
    worst = 0
    for it in range(6):
        ids_np = [rng.integers(1, 50_000, B) for _ in range(n)]
        if it % 2:
            ids_np[2][rng.random(B) < 0.4] = 8                         # a hot id: its owner's block of feature 2 overflows
        ids = [torch.from_numpy(x).to(DEV) for x in ids_np]
        send, pos, slot, counts, overflow, state = _route(ids, world, capf, state=state, overflow=overflow)
        w_send, w_pos, w_slot, w_counts, w_max = ref_np.route_feat(ids_np, world, capf)
        worst = max(worst, w_max)
        assert np.array_equal(send.cpu().numpy(), w_send) and np.array_equal(slot.cpu().numpy(), w_slot)
        assert np.array_equal(pos.cpu().numpy(), w_pos) and np.array_equal(counts.cpu().numpy(), w_counts)
        assert int(overflow.item()) == worst                           # a running maximum since the caller zeroed it
        assert ((w_slot == -1).sum() > 0) == bool(it % 2)               # the hot id's block drops its surplus, the uniform launches drop nothing
    assert worst > capf


def test_inbox_transpose_equals_its_definition():
    lib = _lib.load()
    rng = np.random.default_rng(9)
    for world, n, capf in ((2, 3, 64), (3, 26, 2048), (8, 5, 8256)):
        a = rng.integers(-5, 1 << 30, (world, n, capf)).astype(np.int32)
        b = rng.integers(0, 1 << 20, (world, n, capf)).astype(np.int32)
        ta, tb = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)
        oa, ob = torch.empty((n, world * capf), dtype=torch.int32, device=DEV), torch.empty((n, world * capf), dtype=torch.int32, device=DEV)
        st = torch.cuda.current_stream().cuda_stream
        ops.check(lib.nrx_inbox_transpose(ta.data_ptr(), oa.data_ptr(), tb.data_ptr(), ob.data_ptr(), world, n, capf, st), "t")
        assert np.array_equal(oa.cpu().numpy(), ref_np.owner_ids_from_inbox(a)) and np.array_equal(ob.cpu().numpy(), ref_np.owner_ids_from_inbox(b))
        oa.zero_()
        ops.check(lib.nrx_inbox_transpose(ta.data_ptr(), oa.data_ptr(), None, None, world, n, capf, st), "t")
        assert np.array_equal(oa.cpu().numpy(), ref_np.owner_ids_from_inbox(a))


@pytest.mark.parametrize("D,n,fm", [(16, 26, True), (16, 5, False), (32, 40, False), (64, 3, False), (16, 7, True)])
def test_embed_bwd_scatter_places_every_upstream_row(D, n, fm):
    lib = _lib.load()
    rng = np.random.default_rng(D + n)
    B = 3001
    ld = n * D
    g_out = torch.from_numpy(rng.standard_normal((B, ld)).astype(np.float32)).to(DEV)
    perm = rng.permutation(n * B + 100)[: n * B].astype(np.int32)
    perm[rng.random(n * B) < 0.01] = -1                                # skipped lookups (an overflowed block)
    dest = torch.from_numpy(perm).to(DEV)
    values = torch.full((n * B + 100, D), 7.0, dtype=torch.float32, device=DEV)
    slots = [ops.Slot(f"f{i}", NRX_SPARSE, 0, D, 0, i * D, fm_field=int(fm)) for i in range(n)]
    plan = ops.EmbedPlan(slots, out_width=ld, use_fm=fm)
    ins = [dest[i * B:(i + 1) * B] for i in range(n)]
    arr = ops._fill_features(plan, 0, n, [None], ins, [None] * n, table_ptrs=[values.data_ptr()], fm=fm)
    fmg = None
    if fm:
        feat = torch.from_numpy(rng.standard_normal((B, ld)).astype(np.float32)).to(DEV)
        sums = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)).to(DEV)
        g_fm = torch.from_numpy(rng.standard_normal((B,)).astype(np.float32)).to(DEV)
        fmg = NrxFmGrad(g_fm.data_ptr(), sums.data_ptr(), D, feat.data_ptr(), ld)
    ops.check(lib.nrx_embed_bwd_scatter(arr, n, B, D, g_out.data_ptr(), ld, None, 0, fmg, dest.data_ptr(), values.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), "nrx_embed_bwd_scatter")
    torch.cuda.synchronize()
    want = np.full((n * B + 100, D), 7.0, np.float32)
    g = g_out.cpu().numpy()
    for i in range(n):
        rows = g[:, i * D:(i + 1) * D].copy()
        if fm:      # nrx_fm_grad_t: g + g_fm * (k == 0 ? 1 : sums[b, k] - feat[b, col + k]), one fma as the kernels form it
            f_ = feat.cpu().numpy()[:, i * D:(i + 1) * D]
            s_ = sums.cpu().numpy()
            d_ = np.concatenate([np.ones((B, 1), np.float32), (s_[:, 1:] - f_[:, 1:]).astype(np.float32)], axis=1)
            rows = (rows.astype(np.float64) + g_fm.cpu().numpy()[:, None].astype(np.float64) * d_.astype(np.float64)).astype(np.float32)
        d = perm[i * B:(i + 1) * B]
        want[d[d >= 0]] = rows[d >= 0]
    got = values.cpu().numpy()
    if fm:
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)   # (fma vs the float64 restatement: last-place differences)
    else:
        assert np.array_equal(got, want)                              # pure copies


def _direct(feats, tabs_full, inputs, fm, g_out, g_fm):
    """The direct (unsharded) bound path on full tables: forward + row-sparse backward."""
    names = sorted({f.table for f in feats})
    slots, col = [], 0
    for f in feats:
        slots.append(ops.Slot(f.name, NRX_SPARSE, names.index(f.table), f.dim, 0, col, fm_field=int(fm)))
        col += f.dim
    plan = ops.EmbedPlan(slots, out_width=col, use_fm=fm)
    sums = torch.empty((inputs[0].numel(), feats[0].dim), dtype=torch.float32, device=DEV) if fm else None
    fwd = ops.PreparedEmbed(plan, [tabs_full[t] for t in names], inputs, [None] * len(feats), fm_sums=sums)
    out, _, fmv = fwd.run()
    bwd = ops.PreparedSparseBackward(fwd, g_out, g_fm)
    groups = bwd.run()
    torch.cuda.synchronize()
    return out, fmv, groups, names


@pytest.mark.parametrize("one_sided,direct_grad", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("case", ["c2_like_fm", "two_dims", "shared_table", "tiny_tables"])
def test_world_1_step_equals_the_direct_path_bit_for_bit(case, one_sided, direct_grad):
    """direct_grad: the requester's pack writes the rows that need no reduction straight into the owner's values[] (the owner's plan came back
    first) and the owner only walks the listed rows -- the same (keys, values), bit for bit, as the buffered form and as the direct path."""
    rng = np.random.default_rng(sum(map(ord, case)))
    gen = torch.Generator(device=DEV).manual_seed(3)
    fm = case == "c2_like_fm"
    if case == "c2_like_fm":
        B, spec = 8192, [(f"C{i:02d}", f"C{i:02d}", 16, 300_000) for i in range(26)]
    elif case == "two_dims":
        B, spec = 5000, [("a", "a", 16, 7000), ("b", "b", 32, 90_000), ("c", "c", 16, 50), ("d", "d", 32, 1200), ("e", "e", 64, 40_000)]
    elif case == "shared_table":
        B, spec = 3000, [("item_id", "item_id", 16, 20_000), ("last_click", "item_id", 16, 20_000), ("user_id", "user_id", 16, 70_000)]
    else:
        B, spec = 2500, [("x", "x", 16, 3), ("y", "y", 16, 18), ("z", "z", 16, 270)]
    feats = [ShardedFeature(nm, NRX_SPARSE, tb, d, 0, False, fm) for nm, tb, d, _ in sorted(spec)]
    rows = {tb: r for _, tb, _, r in spec}
    dims = {tb: d for _, tb, d, _ in spec}
    arenas = {t: shard_step.make_arena(rows[t], dims[t], 0, 1, DEV, generator=gen) for t in rows}
    full = {t: shard_step.arena_shard(a).clone() for t, a in arenas.items()}      # world 1: the shard IS the table
    inputs = [torch.from_numpy(rng.integers(0, rows[f.table], B)).to(DEV) for f in feats]
    inputs[0][:5] = 0
    width = sum(f.dim for f in feats)
    g_out = torch.randn((B, width), device=DEV, generator=gen)
    g_fm = torch.randn((B,), device=DEV, generator=gen) if fm else None
    eng = RowShardedEmbedding(0, 1, overflow_policy="defer")
    step = shard_step.PreparedShardedStep(eng, feats, inputs, [None] * len(feats), arenas, one_sided=one_sided).bind_backward(g_out, g_fm, direct_grad=direct_grad)
    assert all(g["placed"] == one_sided for g in step.groups)      # (an FM epilogue over placed features = a pass over the finished concat)
    assert all(b["direct"] == direct_grad for b in step.bwd)
    for _ in range(2):                                                # re-launchable: same buffers, same result
        out, _, fmv = step.run()
        entries = step.backward()
    torch.cuda.synchronize()
    assert not step.overflowed()
    d_out, d_fm, d_groups, names = _direct(feats, full, inputs, fm, g_out, g_fm)
    assert torch.equal(out, d_out)
    if fm and not one_sided:
        assert torch.equal(fmv, d_fm)
    elif fm:       # (the pass over the finished concat adds the lanes' partial sums in another order than the fused epilogue: last-place differences;
                   #  the FIELD SUMS -- what the backward folds in -- are the same bits, or the gradient check below would fail)
        torch.testing.assert_close(fmv, d_fm, rtol=1e-5, atol=1e-5 * float(d_fm.abs().max()))
    assert len(entries) == len(d_groups)
    for e, d in zip(sorted(entries, key=lambda e: e["dim"]), sorted(d_groups, key=lambda g: g["dim"])):
        nu = int(d["counts"][0])
        assert int(e["counts"][0]) == nu
        # the direct path numbers its tables over ALL table names; the entry over the group's own table list: compare by name
        ek, dk = e["uniq"][:nu].cpu().numpy(), d["uniq"][:nu].cpu().numpy()
        e_names = [next(n for n, a in arenas.items() if a is t) for t in e["tables"]]
        e_tab = np.array([names.index(e_names[t]) for t in (ek >> 40)])
        pad = (dk & ((1 << 40) - 1)) == 0                              # the padding row: owner id 0 = the arena's dummy row
        assert np.array_equal(e_tab, dk >> 40)
        assert np.array_equal(np.where(pad, 0, (ek & ((1 << 40) - 1)) - 1), dk & ((1 << 40) - 1))
        ev, dv = e["values"][:nu], d["values"][:nu]
        assert torch.equal(ev.view(torch.int32), dv.view(torch.int32))


def test_world_1_fused_sparse_adam_step_moves_the_same_rows():
    """forward + backward + FusedSparseAdam on the arenas == the same on the full tables (the direct path): same weights afterwards."""
    from news_recsys_amd.model.model_utils.optim import FusedSparseAdam
    rng = np.random.default_rng(2)
    gen = torch.Generator(device=DEV).manual_seed(8)
    B = 4096
    spec = [(f"f{i}", 16, 50_000 + 1000 * i) for i in range(6)]
    feats = [ShardedFeature(nm, NRX_SPARSE, nm, d, 0, False, False) for nm, d, _ in spec]
    arenas = {nm: shard_step.make_arena(r, d, 0, 1, DEV, generator=gen) for nm, d, r in spec}
    full = {nm: shard_step.arena_shard(a).clone() for nm, a in arenas.items()}
    inputs = [torch.from_numpy(rng.integers(0, r, B)).to(DEV) for _, _, r in spec]
    g_out = torch.randn((B, 96), device=DEV, generator=gen)
    eng = RowShardedEmbedding(0, 1, overflow_policy="defer")
    step = shard_step.PreparedShardedStep(eng, feats, inputs, [None] * 6, arenas).bind_backward(g_out)
    sink_a, sink_b = ops.SparseGradSink(), ops.SparseGradSink()
    opt_a = FusedSparseAdam(sink_a, lr=0.05)
    opt_b = FusedSparseAdam(sink_b, lr=0.05)
    for _ in range(3):
        step.run()
        step.sink_entries(sink_a)
        opt_a.step()
        _, _, groups, names = _direct(feats, full, inputs, False, g_out, None)
        for g in groups:
            sink_b.pending.append(dict(tables=[full[n] for n in names], dim=g["dim"], uniq=g["uniq"], values=g["values"], counts=g["counts"], cap=g["cap"]))
        opt_b.step()
    torch.cuda.synchronize()
    for nm in arenas:
        assert torch.equal(arenas[nm][1:], full[nm]), nm
        assert float(arenas[nm][0].abs().max()) == 0.0 and float(arenas[nm][1].abs().max()) == 0.0      # dummy row, global padding row


def test_gather_place_feat_equals_its_definition():
    """nrx_gather_place_feat against numpy: every (feature, pseudo-sample) with a position writes its arena row (owner id 0: zeros) at
    peer[s][pos, col_f]; empty slots (position -1) write nothing; ids outside the arena and positions outside the batch are dropped and counted."""
    lib = _lib.load()
    rng = np.random.default_rng(31)
    for D, n, world, capf, B in ((16, 5, 1, 640, 640), (32, 3, 3, 256, 500), (64, 9, 2, 128, 200), (128, 2, 4, 64, 100)):
        bp = world * capf
        rows = [int(rng.integers(5, 400)) for _ in range(n)]
        arenas = [torch.from_numpy(rng.standard_normal((r, D)).astype(np.float32)).to(DEV) for r in rows]
        for a in arenas:
            a[0].zero_()
        oid = np.stack([rng.integers(0, r, bp) for r in rows]).astype(np.int32)
        # every (source block, feature) names each sample position at most once -- what nrx_route_feat produces
        opos = np.full((n, bp), -1, np.int32)
        for f in range(n):
            for s_ in range(world):
                k = int(rng.integers(capf // 2, capf + 1))
                opos[f, s_ * capf: s_ * capf + k] = rng.permutation(B)[:k] if k <= B else np.concatenate([rng.permutation(B), -np.ones(k - B, np.int64)])
        oid[opos < 0] = 0
        oid[0, 3], oid[1 % n, 7] = 0, 0                                  # padding lookups with a position: zeros are written
        bad_id, bad_pos = 0, 0
        if bp > 40:
            if opos[0, 11] >= 0:
                oid[0, 11] = rows[0] + 5; bad_id += 1
            if opos[n - 1, 13] >= 0:
                opos[n - 1, 13] = B + 2; bad_pos += 1
        ld = n * D + 4
        outs = [torch.full((B, ld), 9.0, dtype=torch.float32, device=DEV) for _ in range(world)]
        cols = [f * D for f in range(n)]
        status = torch.zeros(4, dtype=torch.int32, device=DEV)
        t_oid, t_pos = torch.from_numpy(oid).to(DEV), torch.from_numpy(opos).to(DEV)
        ops.check(lib.nrx_gather_place_feat((C.c_void_p * n)(*[a.data_ptr() for a in arenas]), (C.c_int64 * n)(*rows), (C.c_int32 * n)(*cols), n, world,
                                            capf, t_oid.data_ptr(), t_pos.data_ptr(), D, (C.c_void_p * world)(*[o.data_ptr() for o in outs]), ld, B,
                                            status.data_ptr(), torch.cuda.current_stream().cuda_stream), "nrx_gather_place_feat")
        torch.cuda.synchronize()
        want = [np.full((B, ld), 9.0, np.float32) for _ in range(world)]
        tabs = [a.cpu().numpy() for a in arenas]
        for f in range(n):
            for b in range(bp):
                p = opos[f, b]
                if p < 0 or p >= B:
                    continue
                i = oid[f, b]
                want[b // capf][p, cols[f]:cols[f] + D] = tabs[f][i] if 0 <= i < rows[f] else 0.0
        for o, w in zip(outs, want):
            assert np.array_equal(o.cpu().numpy(), w)
        assert int(status[0]) == bad_id + bad_pos


def _dense_from_entries(entries, arenas, world=1, rank=0):
    """Sum of the (key, value) lists per table as dense [global rows, D] float64 arrays (a table may be fed by a pooled and a single-valued group)."""
    out = {}
    for e in entries:
        nu = int(e["counts"][0])
        keys, vals = e["uniq"][:nu].cpu().numpy(), e["values"][:nu].double().cpu().numpy()
        for t, arena in enumerate(e["tables"]):
            name = next(n for n, a in arenas.items() if a is arena)
            sel = (keys >> 40) == t
            rows = (keys[sel] & ((1 << 40) - 1))
            live = rows > 0
            d = out.setdefault(name, {})
            for r, v in zip(rows[live], vals[sel][live]):
                g = (int(r) - 1) * world + rank
                d[g] = d.get(g, 0) + v
    return out


@pytest.mark.parametrize("binary", [False, True])
@pytest.mark.parametrize("kind", [NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN])
def test_world_1_tower_with_a_history_bag_against_the_direct_path(kind, binary):
    """The DSSM tower (recall/DSSM/model.py:148-180): item_id + a history bag sharing the news table + user_id.  The bag goes through the pooled
    channel (owner-side partial pooling); its bound backward expands the owner's inbox into pseudo-lookups and reduces them with the planned
    reduction.  Forward: single-valued columns bit for bit, the pooled columns to rtol 1e-6 (the partial-sum order differs); gradient: per
    (table, row) the sum of the lists equals the direct row-sparse gradient to fp32 summation tolerance; two runs give the same bits."""
    rng = np.random.default_rng(17 + kind)
    gen = torch.Generator(device=DEV).manual_seed(4)
    D, L, B, news, users = 16, 11, 6000, 20_000, 300_000
    feats = [ShardedFeature("item_id", NRX_SPARSE, "item_id", D), ShardedFeature("user_history", kind, "item_id", D, L),
             ShardedFeature("user_id", NRX_SPARSE, "user_id", D)]
    arenas = {"item_id": shard_step.make_arena(news, D, 0, 1, DEV, generator=gen), "user_id": shard_step.make_arena(users, D, 0, 1, DEV, generator=gen)}
    full = {t: shard_step.arena_shard(a).clone() for t, a in arenas.items()}
    hist = rng.integers(1, news, (B, L))
    lens = rng.integers(0, L + 1, B)
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.float32)
    hist = np.where(mask > 0, hist, 0)
    inputs = [torch.from_numpy(rng.integers(0, news, B)).to(DEV), torch.from_numpy(hist).to(DEV), torch.from_numpy(rng.integers(1, users, B)).to(DEV)]
    weights = [None, torch.from_numpy(mask).to(DEV) if kind == NRX_BAG_MASKED_MEAN else None, None]
    g_out = torch.randn((B, 3 * D), device=DEV, generator=gen)
    eng = RowShardedEmbedding(0, 1, overflow_policy="defer")
    # binary: the masks here are 0/1 (DataReader's): the pooled backward may skip the expansion (pre-scaled sample rows + nrx_pool_order_remap)
    step = shard_step.PreparedShardedStep(eng, feats, inputs, weights, arenas, binary_masks=binary).bind_backward(g_out)
    assert [g["pooled"] for g in step.groups].count(True) == 1
    runs = []
    for _ in range(2):
        out, _, _ = step.run()
        entries = step.backward()
        torch.cuda.synchronize()
        runs.append((out.clone(), [(e["uniq"].clone(), e["values"].clone(), int(e["counts"][0])) for e in entries]))
    assert not step.overflowed()
    assert torch.equal(runs[0][0], runs[1][0])
    for (k0, v0, n0), (k1, v1, n1) in zip(runs[0][1], runs[1][1]):
        assert n0 == n1 and torch.equal(k0[:n0], k1[:n1]) and torch.equal(v0[:n0].view(torch.int32), v1[:n1].view(torch.int32))
    # ---- the direct path
    slots = [ops.Slot("item_id", NRX_SPARSE, 0, D, 0, 0), ops.Slot("user_history", kind, 0, D, L, D), ops.Slot("user_id", NRX_SPARSE, 1, D, 0, 2 * D)]
    fwd = ops.PreparedEmbed(ops.EmbedPlan(slots, out_width=3 * D), [full["item_id"], full["user_id"]], inputs, weights)
    d_out, _, _ = fwd.run()
    d_groups = ops.PreparedSparseBackward(fwd, g_out).run()
    torch.cuda.synchronize()
    assert torch.equal(out[:, :D], d_out[:, :D]) and torch.equal(out[:, 2 * D:], d_out[:, 2 * D:])
    torch.testing.assert_close(out[:, D:2 * D], d_out[:, D:2 * D], rtol=1e-6, atol=1e-6)
    got = _dense_from_entries(entries, arenas)
    names = ["item_id", "user_id"]
    for g in d_groups:
        nu = int(g["counts"][0])
        keys, vals = g["uniq"][:nu].cpu().numpy(), g["values"][:nu].double().cpu().numpy()
        scale = float(np.abs(vals).max())
        seen = {n: 0 for n in names}
        for k, v in zip(keys, vals):
            r = int(k & ((1 << 40) - 1))
            if r == 0:
                continue
            name = names[k >> 40]
            seen[name] += 1
            np.testing.assert_allclose(got[name][r], v, rtol=1e-5, atol=2e-6 * scale)
        for n in names:
            assert seen[n] == len(got[n])


def test_plan_payload_lists_the_payload_of_every_lookup():
    """nrx_sparse_plan_ex(NRX_PLAN_PAYLOAD): the same plan (unique keys, segments, counts) whose order[] names payload[p] where the plain plan names
    p -- integer work, bit-exact against payload[order] of the plain plan; and nrx_pool_order_remap applied to the plain plan gives the same list
    for the pooled channel's payload (s * n_tags + tag)."""
    lib = _lib.load()
    rng = np.random.default_rng(77)
    for n_feats, lens, rows in ((1, [50_000], [3000]), (3, [7000, 7000, 9000], [500, 40_000, 500]), (2, [4097, 100], [1 << 21, 17])):
        ids = [torch.from_numpy(rng.integers(0, r, l)).to(DEV).to(torch.int32) for l, r in zip(lens, rows)]
        tof = list(range(n_feats)) if n_feats != 3 else [0, 1, 0]
        n_tables = max(tof) + 1
        total = sum(lens)
        payload = torch.from_numpy(rng.integers(0, 1 << 20, total).astype(np.int32)).to(DEV)

        def plan(flags, pl):
            order = torch.empty(total, dtype=torch.int64, device=DEV); uniq = torch.empty(total, dtype=torch.int64, device=DEV)
            seg = torch.empty(total + 1, dtype=torch.int64, device=DEV); counts = torch.empty(n_tables + 2, dtype=torch.int64, device=DEV)
            dest = torch.empty(total, dtype=torch.int32, device=DEV); walk = torch.empty(total, dtype=torch.int32, device=DEV)
            n_walk = torch.empty(2, dtype=torch.int64, device=DEV)
            ws = torch.empty(lib.nrx_sparse_plan_workspace(total), dtype=torch.uint8, device=DEV)
            ops.check(lib.nrx_sparse_plan_ex((C.c_void_p * n_feats)(*[x.data_ptr() for x in ids]), (C.c_int64 * n_feats)(*lens), (C.c_int32 * n_feats)(*tof),
                                             (C.c_int64 * n_feats)(*rows), n_feats, 32, n_tables, 0, flags, order.data_ptr(), uniq.data_ptr(), seg.data_ptr(),
                                             counts.data_ptr(), dest.data_ptr(), walk.data_ptr(), n_walk.data_ptr(), None if pl is None else pl.data_ptr(), None,
                                             None, ws.data_ptr(), torch.cuda.current_stream().cuda_stream), "nrx_sparse_plan_ex")
            torch.cuda.synchronize()
            return order, uniq, seg, counts, walk, n_walk
        o0, u0, s0, c0, w0, nw0 = plan(0, None)
        o1, u1, s1, c1, w1, nw1 = plan(_lib.NRX_PLAN_PAYLOAD, payload)
        nu = int(c0[0])
        assert torch.equal(c0, c1) and torch.equal(u0[:nu], u1[:nu]) and torch.equal(s0[:nu + 1], s1[:nu + 1])
        assert int(nw0[0]) == int(nw1[0]) == nu and torch.equal(w0[:nu], w1[:nu])          # nothing placeable: every row is walked
        assert torch.equal(o1, payload.long()[o0])


@pytest.mark.parametrize("world", [1, 3])
def test_pool_owner_ids_payload_and_remap(world):
    """nrx_pool_inbox_owner_ids / nrx_pool_order_remap against numpy on a routed bag batch (ops.route_bags' inbox layout)."""
    lib = _lib.load()
    rng = np.random.default_rng(world)
    B, L, n, rows_local = 700, 9, 2, 900
    cap = 12288 if world == 1 else 4096
    ids = [torch.from_numpy(rng.integers(0, rows_local * world, (B, L))).to(DEV) for _ in range(n)]
    w = [torch.from_numpy((rng.random((B, L)) < 0.7).astype(np.float32)).to(DEV) for _ in range(n)]
    send_rows, send_tag, send_w, counts2d, overflow = ops.route_bags(ids, w, world, cap)
    torch.cuda.synchronize()
    assert int(overflow.item()) <= cap
    # (the test plays owner 0 of a symmetric exchange: block s of the inbox = what THIS rank's routing addressed to owner s)
    recv2d = counts2d.clone()
    oid = torch.full((world * cap,), -5, dtype=torch.int32, device=DEV)
    pay = torch.full((world * cap,), -5, dtype=torch.int32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    for skip in (0, 1):
        ops.check(lib.nrx_pool_inbox_owner_ids(rows_local, n, B, world, cap, recv2d.data_ptr(), send_rows.data_ptr(), send_tag.data_ptr(), skip,
                                               oid.data_ptr(), pay.data_ptr(), st), "owner_ids")
        torch.cuda.synchronize()
        r, t, c = send_rows.cpu().numpy().reshape(world, cap), send_tag.cpu().numpy().reshape(world, cap), recv2d.cpu().numpy().sum(1)
        want_id, want_pay = np.zeros((world, cap), np.int32), np.zeros((world, cap), np.int32)
        for s_ in range(world):
            k = np.arange(cap) < min(c[s_], cap)
            live = k & (r[s_] >= 0) & (r[s_] < rows_local) & ~((r[s_] == 0) & bool(skip)) & (t[s_] >= 0) & (t[s_] < n * B)
            want_id[s_] = np.where(live, r[s_] + 1, 0)
            want_pay[s_] = np.where(live, s_ * n * B + t[s_], 0)
        assert np.array_equal(oid.cpu().numpy().reshape(world, cap), want_id) and np.array_equal(pay.cpu().numpy().reshape(world, cap), want_pay)
    order = torch.from_numpy(rng.permutation(world * cap)).to(DEV)
    before = order.cpu().numpy()
    ops.check(lib.nrx_pool_order_remap(order.data_ptr(), world * cap, send_tag.data_ptr(), cap, n * B, world, st), "remap")
    torch.cuda.synchronize()
    tg = np.clip(send_tag.cpu().numpy(), None, None)
    tt = np.where((tg[before] >= 0) & (tg[before] < n * B), tg[before], 0)
    assert np.array_equal(order.cpu().numpy(), (before // cap) * (n * B) + tt)


@pytest.mark.parametrize("one_sided", [False, True])
def test_check_raises_index_error_and_overflow(one_sided):
    """An id outside its table is an IndexError (the reference's nn.Embedding raises it on the CPU, base_model.py:271) -- deferred to check();
    a block that overflowed its capacity is a RuntimeError (lookups were dropped)."""
    gen = torch.Generator(device=DEV).manual_seed(1)
    feats = [ShardedFeature("a", NRX_SPARSE, "a", 16), ShardedFeature("b", NRX_SPARSE, "b", 16)]
    arenas = {"a": shard_step.make_arena(500, 16, 0, 1, DEV, generator=gen), "b": shard_step.make_arena(90, 16, 0, 1, DEV, generator=gen)}
    ids = [torch.randint(0, 500, (300,), device=DEV, generator=gen), torch.randint(0, 90, (300,), device=DEV, generator=gen)]
    eng = RowShardedEmbedding(0, 1, overflow_policy="defer")
    step = shard_step.PreparedShardedStep(eng, feats, ids, [None, None], arenas, one_sided=one_sided, check_index=True)
    step.run()
    step.check()                                         # clean
    ids[1][7] = 90                                       # one past the table
    step.run()
    with pytest.raises(IndexError):
        step.check()
    ids[1][7] = 3
    step.run()
    step.check()                                         # the record was cleared
    step.groups[0]["overflow"].fill_(10 ** 9)            # (a dropped block, as nrx_route_feat reports it)
    with pytest.raises(RuntimeError, match="overflowed"):
        step.check()


@pytest.mark.parametrize("world", [1, 2, 5, 8])
@pytest.mark.parametrize("dt", [torch.int64, torch.int32])
def test_route_bags_one_launch_equals_its_definition_and_the_three_launch_form(world, dt):
    """nrx_route_bags_one: the pooled channel's routing in one launch -- integer work (and verbatim float weights), bit-exact against
    oracle/ref_np.py route_bags on the slots in use and against nrx_route_bags on every output word it defines; repeated launches on one state."""
    lib = _lib.load()
    rng = np.random.default_rng(60 + world)
    state = None
    for rep, (n, B, Ls, rows) in enumerate(((1, 300, [7], 1000), (3, 2000, [9, 1, 50], 200_000), (2, 9000, [50, 3], 5_000_000))):
        ids_np = [rng.integers(0, rows, (B, L)) for L in Ls]
        w_np = [((rng.random((B, L)) < 0.6) * rng.random((B, L))).astype(np.float32) for L in Ls]
        if dt is torch.int64:
            ids_np[0][0, 0] = -3
            ids_np[0][1, 0] = (1 << 31) + 9
            w_np[0][0, 0] = w_np[0][1, 0] = 0.5
        total = sum(int((w != 0).sum()) for w in w_np)
        cap = total + 64 if world == 1 else int(total / world * 1.4) + 64
        ids = [torch.from_numpy(x).to(DEV).to(dt) for x in ids_np]
        ws = [torch.from_numpy(x).to(DEV) for x in w_np]
        want = ref_np.route_bags([x if dt is torch.int64 else x.astype(np.int32) for x in ids_np], w_np, world, cap)
        l_rows, l_tag, l_w, l_c2d, l_over = ops.route_bags(ids, ws, world, cap)                      # the three-launch form
        rows_o = torch.full((world * cap,), -7, dtype=torch.int32, device=DEV)
        tag_o = torch.full((world * cap,), -7, dtype=torch.int32, device=DEV)
        w_o = torch.full((world * cap,), -7.0, dtype=torch.float32, device=DEV)
        c2d = torch.full((world, n), -1, dtype=torch.int64, device=DEV)
        over = torch.zeros(1, dtype=torch.int64, device=DEV)
        bl = (C.c_int32 * n)(*Ls)
        state = torch.zeros(lib.nrx_route_bags_one_state_bytes(bl, n, B, world), dtype=torch.uint8, device=DEV)
        for _ in range(2):                                                                            # the chain re-arms itself
            ops.check(lib.nrx_route_bags_one((C.c_void_p * n)(*[x.data_ptr() for x in ids]), (C.c_void_p * n)(*[x.data_ptr() for x in ws]), bl, n,
                                             ids[0].element_size() * 8, B, world, cap, rows_o.data_ptr(), tag_o.data_ptr(), w_o.data_ptr(),
                                             c2d.data_ptr(), over.data_ptr(), state.data_ptr(), torch.cuda.current_stream().cuda_stream), "route_bags_one")
        torch.cuda.synchronize()
        assert np.array_equal(c2d.cpu().numpy(), want[3]) and int(over.item()) == want[4]
        assert torch.equal(c2d, l_c2d)
        used = np.zeros(world * cap, bool)
        for o in range(world):
            used[o * cap: o * cap + min(int(want[3][o].sum()), cap)] = True
        assert np.array_equal(rows_o.cpu().numpy()[used], want[0][used]) and np.array_equal(tag_o.cpu().numpy()[used], want[1][used])
        assert np.array_equal(w_o.cpu().numpy()[used], want[2][used])
        assert np.array_equal(rows_o.cpu().numpy()[used], l_rows.cpu().numpy()[used]) and np.array_equal(w_o.cpu().numpy()[used], l_w.cpu().numpy()[used])
        assert (rows_o.cpu().numpy()[~used] == -7).all()                                              # slots past a block's count: untouched


@pytest.mark.parametrize("world", [1, 2, 5, 8])
@pytest.mark.parametrize("dt", [torch.int64, torch.int32])
def test_route_bags_runs_equals_norm_weights_plus_route_bags_one_and_the_run_definition(world, dt):
    """nrx_route_bags_runs = nrx_bag_norm_weights(_inv) + nrx_route_bags_one in one launch, with run bounds in the place of tags: the rows, the
    travelling weights (the normalisation's bits included: arbitrary float masks), counts2d, the overflow maximum and 1 / den word for word; send_run
    against oracle/ref_np.py route_bags_runs (with the oracle's own weights where the masks are 0/1: numpy's sum order is not the launch's); repeated
    launches on one state; blocks that overflow keep their bounds inside the block."""
    lib = _lib.load()
    rng = np.random.default_rng(90 + world)
    st = torch.cuda.current_stream().cuda_stream
    shapes = ((1, 300, [7], 1000), (3, 2000, [9, 1, 50], 200_000), (2, 9000, [50, 3], 5_000_000), (1, 70, [100], 999), (1, 5, [4096], 50),
              (2, 40, [1365, 2], 777), (1, 33, [2047], 10_000))
    for rep, (n, B, Ls, rows) in enumerate(shapes):
        kinds = [(NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM)[(rep + f) % 3] for f in range(n)]
        for binary in (True, False):
            ids_np = [rng.integers(0, rows, (B, L)) for L in Ls]
            m_np = [(rng.random((B, L)) < 0.6).astype(np.float32) * (1.0 if binary else rng.random((B, L)).astype(np.float32)) for L in Ls]
            m_np[0][B // 2] = 0                                                        # an empty bag
            ids = [torch.from_numpy(x).to(DEV).to(dt) for x in ids_np]
            ms = [None if k == NRX_BAG_MEAN else torch.from_numpy(m).to(DEV) for k, m in zip(kinds, m_np)]
            inv_want = [torch.empty(B, device=DEV) for _ in range(n)]
            wn = [torch.empty((B, L), device=DEV) for L in Ls]
            for f in range(n):
                ops.check(lib.nrx_bag_norm_weights_inv(None if ms[f] is None else ms[f].data_ptr(), B, Ls[f], kinds[f], wn[f].data_ptr(),
                                                       inv_want[f].data_ptr(), st), "norm")
            total = sum(int((w != 0).sum()) for w in wn)
            for tight in (False, True):
                cap = (total + 64 if world == 1 else int(total / world * 1.4) + 64) if not tight else max(8, total // (2 * world))
                bl = (C.c_int32 * n)(*Ls)
                one = [torch.full((world * cap,), -7, dtype=torch.int32, device=DEV), torch.full((world * cap,), -7, dtype=torch.int32, device=DEV),
                       torch.full((world * cap,), -7.0, device=DEV), torch.zeros((world, n), dtype=torch.int64, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)]
                s1 = torch.zeros(lib.nrx_route_bags_one_state_bytes(bl, n, B, world), dtype=torch.uint8, device=DEV)
                idp = (C.c_void_p * n)(*[x.data_ptr() for x in ids])
                ops.check(lib.nrx_route_bags_one(idp, (C.c_void_p * n)(*[x.data_ptr() for x in wn]), bl, n, ids[0].element_size() * 8, B, world, cap,
                                                 one[0].data_ptr(), one[1].data_ptr(), one[2].data_ptr(), one[3].data_ptr(), one[4].data_ptr(), s1.data_ptr(), st), "one")
                nb = lib.nrx_route_bags_runs_state_bytes(bl, n, B, world)
                assert (nb > 0) == all(4096 // L * world <= 4096 for L in Ls)       # (short bags x many owners: too many run words per tile for this form)
                if nb == 0:
                    continue
                s2 = torch.zeros(nb, dtype=torch.uint8, device=DEV)
                rows_o = torch.full((world * cap,), -7, dtype=torch.int32, device=DEV)
                w_o = torch.full((world * cap,), -7.0, device=DEV)
                run = torch.full((world, n * B, 2), -7, dtype=torch.int32, device=DEV)
                inv = [torch.full((B,), -7.0, device=DEV) for _ in range(n)]
                c2d = torch.zeros((world, n), dtype=torch.int64, device=DEV)
                over = torch.zeros(1, dtype=torch.int64, device=DEV)
                for _ in range(2):
                    ops.check(lib.nrx_route_bags_runs(idp, (C.c_void_p * n)(*[(0 if m is None else m.data_ptr()) for m in ms]), (C.c_int32 * n)(*kinds), bl, n,
                                                      ids[0].element_size() * 8, B, world, cap, rows_o.data_ptr(), w_o.data_ptr(), run.data_ptr(),
                                                      (C.c_void_p * n)(*[x.data_ptr() for x in inv]), c2d.data_ptr(), over.data_ptr(), s2.data_ptr(), st), "runs")
                torch.cuda.synchronize()
                assert torch.equal(c2d, one[3]) and torch.equal(over, one[4])
                assert (int(over.item()) > cap) == tight
                used = np.zeros(world * cap, bool)
                for o in range(world):
                    used[o * cap: o * cap + min(int(c2d[o].sum()), cap)] = True
                u = torch.from_numpy(used).to(DEV)
                assert torch.equal(rows_o[u], one[0][u]) and torch.equal(w_o[u].view(torch.int32), one[2][u].view(torch.int32))
                assert bool((rows_o[~u] == -7).all())
                for f in range(n):
                    assert torch.equal(inv[f].view(torch.int32), inv_want[f].view(torch.int32))
                wn_np = [w.cpu().numpy() for w in wn]                                                                # (the launch's own weights: which entries travel)
                assert np.array_equal(run.cpu().numpy(), ref_np.route_bags_runs(ids_np, wn_np, world, cap))
                got_run, tags = run.cpu().numpy(), one[1].cpu().numpy()                                             # ... and the one-launch form's tags lie in their runs
                for o in range(world):
                    k = np.flatnonzero(used[o * cap:(o + 1) * cap])
                    tg = tags[o * cap + k]
                    assert (got_run[o, tg, 0] <= k).all() and (k < got_run[o, tg, 1]).all()
                if binary and not tight and dt is torch.int64:                                                       # ... and from the oracle's
                    wo = [ref_np.bag_norm_weights(None if k == NRX_BAG_MEAN else m, B, L, {NRX_BAG_MASKED_MEAN: "masked_mean", NRX_BAG_MEAN: "mean", NRX_BAG_SUM: "sum"}[k])
                          for k, m, L in zip(kinds, m_np, Ls)]
                    ref = ref_np.route_bags(ids_np, wo, world, cap)
                    assert np.array_equal(c2d.cpu().numpy(), ref[3])
                    assert np.array_equal(run.cpu().numpy(), ref_np.route_bags_runs(ids_np, wo, world, cap))
                    assert np.array_equal(rows_o.cpu().numpy()[used], ref[0][used])
    assert lib.nrx_route_bags_runs_state_bytes((C.c_int32 * 1)(5000), 1, 10, 1) == 0          # a bag longer than a tile: not this form
    assert lib.nrx_route_bags_runs_state_bytes((C.c_int32 * 1)(2), 1, 10, 8) == 0             # 2048 samples x 8 owners of run words per tile


@pytest.mark.parametrize("world", [1, 3])
@pytest.mark.parametrize("D", [16, 24, 64])
def test_pool_inbox_fwd_runs_equals_pool_inbox_fwd_and_the_runs_give_the_backwards_words(world, D):
    """nrx_pool_inbox_fwd_runs over the routing launch's run bounds: the partial sums bit for bit those of nrx_pool_inbox_fwd (memset + marking pass
    over the tags); nrx_pool_inbox_runs_words: tag_out = the tags on every entry of a run (nothing else written), oid_out / payload_out =
    nrx_pool_inbox_owner_ids' words on ALL world * cap slots."""
    lib = _lib.load()
    rng = np.random.default_rng(5 + world + D)
    st = torch.cuda.current_stream().cuda_stream
    B, Ls, rows_local = 900, [9, 30], 700
    n = len(Ls)
    ids = [torch.from_numpy(rng.integers(0, rows_local * world, (B, L))).to(DEV) for L in Ls]
    ms = [torch.from_numpy((rng.random((B, L)) < 0.7).astype(np.float32)).to(DEV) for L in Ls]
    kinds = [NRX_BAG_MASKED_MEAN, NRX_BAG_SUM]
    table = torch.randn((rows_local, D), device=DEV)
    total = sum(int((m != 0).sum()) for m in ms)
    cap = total + 64 if world == 1 else int(total / world * 1.5) + 64
    bl = (C.c_int32 * n)(*Ls)
    wn = [ops.bag_norm_weights(m, B, L, k) for m, L, k in zip(ms, Ls, kinds)]
    l_rows, l_tag, l_w, l_c2d, _ = ops.route_bags(ids, wn, world, cap)
    want = ops.pool_inbox([table], [0, 0], B, world, cap, l_c2d, l_rows, l_tag, l_w)          # (the test plays owner 0 of a symmetric exchange)
    rows_o = torch.empty(world * cap, dtype=torch.int32, device=DEV)
    w_o = torch.empty(world * cap, device=DEV)
    run = torch.empty((world, n * B, 2), dtype=torch.int32, device=DEV)
    c2d = torch.zeros((world, n), dtype=torch.int64, device=DEV)
    over = torch.zeros(1, dtype=torch.int64, device=DEV)
    state = torch.zeros(lib.nrx_route_bags_runs_state_bytes(bl, n, B, world), dtype=torch.uint8, device=DEV)
    ops.check(lib.nrx_route_bags_runs((C.c_void_p * n)(*[x.data_ptr() for x in ids]), (C.c_void_p * n)(*[m.data_ptr() for m in ms]), (C.c_int32 * n)(*kinds), bl, n,
                                      64, B, world, cap, rows_o.data_ptr(), w_o.data_ptr(), run.data_ptr(), None, c2d.data_ptr(), over.data_ptr(),
                                      state.data_ptr(), st), "runs")
    tp, tr, ft = (C.c_void_p * 1)(table.data_ptr()), (C.c_int64 * 1)(rows_local), (C.c_int32 * n)(0, 0)
    partial = torch.full((world, n * B, D), 7.0, device=DEV)
    ops.check(lib.nrx_pool_inbox_fwd_runs(tp, tr, 1, ft, n, B, world, cap, c2d.data_ptr(), rows_o.data_ptr(), w_o.data_ptr(), run.data_ptr(), D,
                                          partial.data_ptr(), None, st), "fwd_runs")
    torch.cuda.synchronize()
    assert torch.equal(partial.view(torch.int32), want.view(torch.int32))
    used = torch.zeros(world * cap, dtype=torch.bool, device=DEV)
    for o in range(world):
        used[o * cap: o * cap + int(c2d[o].sum())] = True
    for skip in (0, 1):
        tag_o = torch.full((world * cap,), -7, dtype=torch.int32, device=DEV)
        oid = torch.full((world * cap,), -7, dtype=torch.int32, device=DEV)
        pay = torch.full((world * cap,), -7, dtype=torch.int32, device=DEV)
        ops.check(lib.nrx_pool_inbox_runs_words(rows_local, n, B, world, cap, c2d.data_ptr(), rows_o.data_ptr(), run.data_ptr(), skip, tag_o.data_ptr(),
                                                oid.data_ptr(), pay.data_ptr(), st), "runs_words")
        want_id, want_pay = torch.empty_like(oid), torch.empty_like(pay)
        ops.check(lib.nrx_pool_inbox_owner_ids(rows_local, n, B, world, cap, l_c2d.data_ptr(), l_rows.data_ptr(), l_tag.data_ptr(), skip,
                                               want_id.data_ptr(), want_pay.data_ptr(), st), "owner_ids")
        torch.cuda.synchronize()
        assert torch.equal(oid, want_id) and torch.equal(pay, want_pay)
        assert torch.equal(tag_o[used], l_tag[used]) and bool((tag_o[~used] == -7).all())
        want_tags = ref_np.tags_from_runs(run.cpu().numpy(), world, cap)                        # the definition: -1 outside the runs
        assert np.array_equal(np.where(tag_o.cpu().numpy() == -7, -1, tag_o.cpu().numpy()), want_tags)
    only_tags = torch.full((world * cap,), -7, dtype=torch.int32, device=DEV)
    ops.check(lib.nrx_pool_inbox_runs_words(rows_local, n, B, world, cap, c2d.data_ptr(), rows_o.data_ptr(), run.data_ptr(), 0, only_tags.data_ptr(), None, None, st),
              "runs_words")
    torch.cuda.synchronize()
    assert torch.equal(only_tags[used], l_tag[used])


@pytest.mark.parametrize("binary", [False, True])
def test_the_three_routings_of_the_pooled_channel_give_the_same_step_bit_for_bit(binary, monkeypatch):
    """NRX_ROUTE_BAGS = runs (default) | one | legacy: the same forward and the same (keys, values) word for word."""
    rng = np.random.default_rng(23)
    D, L, B, news = 16, 50, 5000, 30_000
    feats = [ShardedFeature("item_id", NRX_SPARSE, "item_id", D), ShardedFeature("user_history", NRX_BAG_MASKED_MEAN, "item_id", D, L)]
    hist = rng.integers(1, news, (B, L))
    mask = (np.arange(L)[None, :] < rng.integers(0, L + 1, B)[:, None]).astype(np.float32)
    inputs = [torch.from_numpy(rng.integers(0, news, B)).to(DEV), torch.from_numpy(np.where(mask > 0, hist, 0)).to(DEV)]
    weights = [None, torch.from_numpy(mask).to(DEV)]
    g_out = torch.randn((B, 2 * D), device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    res = {}
    for how in ("runs", "one", "legacy"):
        monkeypatch.setenv("NRX_ROUTE_BAGS", how)
        arenas = {"item_id": shard_step.make_arena(news, D, 0, 1, DEV, generator=torch.Generator(device=DEV).manual_seed(4))}
        eng = RowShardedEmbedding(0, 1, overflow_policy="defer")
        step = shard_step.PreparedShardedStep(eng, feats, inputs, weights, arenas, binary_masks=binary)
        g = next(g for g in step.groups if g["pooled"])
        assert (g.get("runs_state") is not None) == (how == "runs") and (g.get("rstate") is not None) == (how == "one")
        out, _, _ = step.run()
        step.bind_backward(g_out)
        first = [(e["uniq"].clone(), e["values"].clone(), int(e["counts"][0])) for e in step.backward()]
        for _ in range(2):
            out, _, _ = step.run()
            entries = step.backward()
        torch.cuda.synchronize()
        assert not step.overflowed()
        cur = [(e["uniq"].clone(), e["values"].clone(), int(e["counts"][0])) for e in entries]
        for (k0, v0, n0), (k1, v1, n1) in zip(first, cur):
            assert n0 == n1 and torch.equal(k0[:n0], k1[:n1]) and torch.equal(v0[:n0].view(torch.int32), v1[:n1].view(torch.int32))
        res[how] = (out.clone(), cur)
    for how in ("one", "legacy"):
        assert torch.equal(res["runs"][0].view(torch.int32), res[how][0].view(torch.int32))
        assert len(res["runs"][1]) == len(res[how][1])
        for (k0, v0, n0), (k1, v1, n1) in zip(res["runs"][1], res[how][1]):
            assert n0 == n1 and torch.equal(k0[:n0], k1[:n1]) and torch.equal(v0[:n0].view(torch.int32), v1[:n1].view(torch.int32))


@pytest.mark.parametrize("case", ["tower_two_groups", "fm_one_group"])
def test_world_1_step_is_capturable_and_the_graph_replays_the_eager_bits(case):
    """The bound step allocates nothing, reads nothing back and forks / joins its exchange groups with wait_stream: forward + backward captured in a
    HIP graph replay the eager step's concat, keys, values and counts word for word (a DSSM tower: two groups side by side on two streams, the
    pooled channel's two-launch forward; an FM plan: one-sided placement + the pass over the finished concat)."""
    rng = np.random.default_rng(3)
    gen = torch.Generator(device=DEV).manual_seed(4)
    D, B = 16, 6000
    if case == "tower_two_groups":
        L, news, users = 11, 20_000, 300_000
        feats = [ShardedFeature("item_id", NRX_SPARSE, "item_id", D), ShardedFeature("user_history", NRX_BAG_MASKED_MEAN, "item_id", D, L),
                 ShardedFeature("user_id", NRX_SPARSE, "user_id", D)]
        arenas = {"item_id": shard_step.make_arena(news, D, 0, 1, DEV, generator=gen), "user_id": shard_step.make_arena(users, D, 0, 1, DEV, generator=gen)}
        mask = (np.arange(L)[None, :] < rng.integers(0, L + 1, B)[:, None]).astype(np.float32)
        inputs = [torch.from_numpy(rng.integers(0, news, B)).to(DEV), torch.from_numpy(np.where(mask > 0, rng.integers(1, news, (B, L)), 0)).to(DEV),
                  torch.from_numpy(rng.integers(1, users, B)).to(DEV)]
        weights = [None, torch.from_numpy(mask).to(DEV), None]
        g_fm = None
    else:
        n, rows = 7, 50_000
        feats = [ShardedFeature(f"C{i}", NRX_SPARSE, f"C{i}", D, fm=True) for i in range(n)]
        arenas = {f"C{i}": shard_step.make_arena(rows, D, 0, 1, DEV, generator=gen) for i in range(n)}
        inputs = [torch.from_numpy(rng.integers(0, rows, B)).to(DEV) for _ in range(n)]
        weights = [None] * n
        g_fm = torch.randn(B, device=DEV, generator=gen)
    g_out = torch.randn((B, sum(f.dim for f in feats)), device=DEV, generator=gen)
    eng = RowShardedEmbedding(0, 1, overflow_policy="defer")
    step = shard_step.PreparedShardedStep(eng, feats, inputs, weights, arenas, binary_masks=True).bind_backward(g_out, g_fm)
    if case == "tower_two_groups":
        import os
        assert len(step.groups) == 2 and (step._side_streams()[0] is not None or os.environ.get("NRX_SHARD_OVERLAP", "1") != "1")
    for _ in range(3):                                  # (the planners choose from the previous batch's statistics: settled before the capture)
        out, _, fm = step.run()
        entries = step.backward()
    torch.cuda.synchronize()
    want = (out.clone(), None if fm is None else fm.clone(), [(e["uniq"].clone(), e["values"].clone(), int(e["counts"][0])) for e in entries])
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out2, _, fm2 = step.run()
        entries2 = step.backward()
    out2.zero_()
    for e in entries2:
        e["values"].zero_()
        e["counts"].zero_()
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out2.view(torch.int32), want[0].view(torch.int32))
    assert fm is None or torch.equal(fm2.view(torch.int32), want[1].view(torch.int32))
    assert len(entries2) == len(want[2])
    for e, (k, v, n) in zip(entries2, want[2]):
        assert int(e["counts"][0]) == n and torch.equal(e["uniq"][:n], k[:n]) and torch.equal(e["values"][:n].view(torch.int32), v[:n].view(torch.int32))
