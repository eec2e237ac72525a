"""The one-kernel planner of the row-sparse backward (nrx_sparse_plan_lds: row bitmaps in LDS, no sort) and the pair pass behind it
(nrx_embed_bwd_placed_pairs), against
  * its definition, oracle.ref_np.sparse_plan_pairs (itself written on oracle.ref_np.sparse_plan, the sorted plan's definition):
    unique rows, per-table bounds, dest words, pair records and walk rows bit for bit;
  * the sorted planner's backward (nrx_sparse_plan_place + nrx_embed_bwd_placed / _placed_dense), which earlier tests tie to the
    reference's gradients (goldens, fp64 restatements): same unique rows, same row gradients BIT FOR BIT (int32 words);
  * a float64 restatement of autograd's index_add (oracle.ref_np.embedding_grad_dense) directly, rtol 1e-5.
Backward of src/model/BaseModel/base_model.py:262-308 (+ sort/fm/model.py:18-26)."""
import ctypes as C

import numpy as np
import pytest
import torch

from news_recsys_amd import _lib, ops
from news_recsys_amd._lib import NRX_SPARSE
from oracle import ref_np as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _plan_lds(ids_np, tab, rows, nt, dtype=torch.int64, misalign=False, state=None):
    lib = _lib.load()
    dev = torch.device(DEV)
    ids = []
    for x in ids_np:
        t = torch.from_numpy(x).to(DEV).to(dtype)
        if misalign:                                   # a view that starts 8 (or 4) bytes into its buffer: the one-id-per-load form
            buf = torch.empty(t.numel() + 1, dtype=dtype, device=DEV)
            buf[1:] = t
            t = buf[1:]
        ids.append(t)
    n = len(ids)
    total = sum(x.numel() for x in ids)
    order = torch.full((total,), -7, dtype=torch.int64, device=DEV)
    uniq = torch.full((total,), -7, dtype=torch.int64, device=DEV)
    seg = torch.full((total + 1,), -7, dtype=torch.int64, device=DEV)
    counts = torch.full((nt + 2,), -7, dtype=torch.int64, device=DEV)
    dest = torch.full((total,), 0x7f7f7f7f, dtype=torch.int32, device=DEV)
    walk = torch.full((total,), -7, dtype=torch.int32, device=DEV)
    n_walk = torch.full((2,), -7, dtype=torch.int64, device=DEV)
    pairs = torch.full((total // 2 + 1, 4), -7, dtype=torch.int32, device=DEV)
    stats = torch.zeros(4, dtype=torch.int64, device=DEV)
    if state is None:
        state = torch.zeros(lib.nrx_sparse_plan_lds_state_bytes(), dtype=torch.uint8, device=DEV)
    ws = torch.empty(lib.nrx_sparse_plan_lds_workspace(total), dtype=torch.uint8, device=DEV)
    ptrs = (C.c_void_p * n)(*[x.data_ptr() for x in ids])
    lens = (C.c_int64 * n)(*[x.numel() for x in ids])
    tof = (C.c_int32 * n)(*tab)
    rws = (C.c_int64 * n)(*rows)
    assert lib.nrx_sparse_plan_lds_ok(lens, tof, rws, n, nt) == 1
    rc = lib.nrx_sparse_plan_lds(ptrs, lens, tof, rws, n, ids[0].element_size() * 8, nt, order.data_ptr(), uniq.data_ptr(), seg.data_ptr(),
                                 counts.data_ptr(), dest.data_ptr(), walk.data_ptr(), n_walk.data_ptr(), pairs.data_ptr(), n_walk.data_ptr() + 8,
                                 stats.data_ptr(), state.data_ptr(), ws.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "nrx_sparse_plan_lds")
    torch.cuda.synchronize()
    return dict(order=order.cpu().numpy(), uniq=uniq.cpu().numpy(), seg=seg.cpu().numpy(), counts=counts.cpu().numpy(),
                dest=dest.cpu().numpy(), walk=walk.cpu().numpy(), n_walk=int(n_walk[0].item()), n_pairs=int(n_walk[1].item()),
                pairs=pairs.cpu().numpy(), stats=stats.cpu().numpy(), total=total, state=state)


def _check_against_definition(got, ids_np, tab, rows, nt):
    uniq, counts, dest, pairs, walk, walk_lookups = R.sparse_plan_pairs(ids_np, tab, rows, nt)
    nu, n = len(uniq), got["total"]
    assert np.array_equal(got["counts"], counts)
    assert np.array_equal(got["uniq"][:nu], uniq)
    assert got["n_walk"] == len(walk) and np.array_equal(got["walk"][:len(walk)], walk)
    assert np.array_equal(got["dest"][:n], dest)                               # once: u; every other lookup: -1
    assert got["n_pairs"] == len(pairs) and np.array_equal(got["pairs"][:len(pairs), :3], pairs)      # twice: {u, first, second}, ascending
    at = 0                                                                      # the walk rows' lookups: packed, in walk order, ascending
    for u in walk:
        ps = walk_lookups[int(u)]
        assert got["seg"][u] == at and got["seg"][u + 1] == at + len(ps)
        assert np.array_equal(got["order"][at:at + len(ps)], ps)
        at += len(ps)
    assert list(got["stats"]) == [nu, len(walk), at, n]


def _case_ids(rng, kind, B, rows, n):
    if kind == "uniform":
        ids = [rng.integers(1, r, B) for r in rows]
    elif kind == "dup":                                    # many rows looked up 2, 3, 4 ... times: the list, its sort, the walk rows
        ids = [rng.integers(1, max(2, min(r, B // 3)), B) for r in rows]
    elif kind == "one_range":                              # every lookup in ONE 131 072-row range: more than the LDS cache holds -> the re-scan path
        ids = [rng.integers(1, min(r, 100000), B) for r in rows]
    elif kind == "hot":                                    # a handful of very hot rows: lists beyond the LDS sort -> the in-place sort
        ids = [np.where(rng.random(B) < 0.6, rng.integers(1, 6, B), rng.integers(1, r, B)) for r in rows]
    else:
        raise AssertionError(kind)
    ids = [np.asarray(x, np.int64) for x in ids]
    for x in ids:
        x[:2] = 0                                          # the padding row
        if len(x) > 3:
            x[2] = -5                                      # out-of-range ids fall on the padding row
            x[3] = 1 << 33
    return ids


PLAN_CASES = [
    # name, kind, B, rows per feature, table of each feature, n_tables
    ("c2_like", "uniform", 4099, [300000] * 6, list(range(6)), 6),
    ("ragged_rows", "uniform", 3001, [131072, 131073, 17, 262145, 5000], [0, 1, 2, 3, 4], 5),
    ("shared_table", "uniform", 2500, [200000, 200000, 70000], [0, 0, 2], 4),            # two features on table 0; table 1 and 3 unread
    ("dup_heavy", "dup", 6000, [50000, 400000], [0, 1], 2),
    ("cache_overflow", "one_range", 30011, [1000000, 300000], [0, 1], 2),
    ("hot_rows", "hot", 20000, [500000], [0], 1),
    ("tiny_batch", "uniform", 1, [1000, 200000], [0, 1], 2),
    ("odd_batch", "uniform", 7, [1000], [0], 1),
]


@pytest.mark.parametrize("name,kind,B,rows,tab,nt", PLAN_CASES, ids=[c[0] for c in PLAN_CASES])
@pytest.mark.parametrize("dtype,misalign", [(torch.int64, False), (torch.int32, False), (torch.int64, True), (torch.int32, True)],
                         ids=["i64", "i32", "i64-unaligned", "i32-unaligned"])
def test_plan_lds_equals_its_definition(name, kind, B, rows, tab, nt, dtype, misalign):
    rng = np.random.default_rng(len(name) * 77 + B)
    ids = _case_ids(rng, kind, B, rows, len(rows))
    if dtype == torch.int32:
        for x in ids:
            if len(x) > 3:
                x[3] = -9                                  # (2^33 does not exist in 32 bits)
    got = _plan_lds(ids, tab, rows, nt, dtype, misalign)
    _check_against_definition(got, ids, tab, rows, nt)


def test_plan_lds_state_is_rearmed_between_calls_of_different_shapes():
    """The control block (ticket, epoch, per-block totals) is left ready by every call: shapes with more and fewer row ranges alternate on ONE
    state, and every plan still equals its definition (a stale total of an earlier call would shift the unique indices)."""
    rng = np.random.default_rng(5)
    lib = _lib.load()
    state = torch.zeros(lib.nrx_sparse_plan_lds_state_bytes(), dtype=torch.uint8, device=DEV)
    shapes = [([900000] * 3, [0, 1, 2], 3, 5000), ([40000], [0], 1, 777), ([2000000, 150000], [0, 1], 2, 9000), ([40000], [0], 1, 778)]
    for _ in range(3):
        for rows, tab, nt, B in shapes:
            ids = _case_ids(rng, "uniform", B, rows, len(rows))
            got = _plan_lds(ids, tab, rows, nt, state=state)
            _check_against_definition(got, ids, tab, rows, nt)


def test_plan_lds_refuses_launches_outside_its_shapes():
    lib = _lib.load()
    def ok(lens, tab, rows, nt):
        n = len(lens)
        return lib.nrx_sparse_plan_lds_ok((C.c_int64 * n)(*lens), (C.c_int32 * n)(*tab), (C.c_int64 * n)(*rows), n, nt)
    assert ok([100, 100], [0, 1], [1000, 1000], 2) == 1
    assert ok([100, 5000], [0, 1], [1000, 1000], 2) == 0            # a bag feature (its lookups are a multiple of the batch)
    assert ok([65536] * 5, [0, 1, 2, 3, 4], [100_000_000, 1_000_000, 18, 270, 18], 5) == 0      # C3's 100 M-row table: 763 ranges scanning 65 536 ids each
    assert ok([100, 100], [0, 0], [1000, 2000], 1) == 0             # one table, two row counts


def _grads(plan, tables, inputs, g_out, g_fm, lds, sparse, monkeypatch, only_fm=False):
    monkeypatch.setattr(ops, "PLAN_LDS", "1" if lds else "0")
    plan.__dict__.pop("_sg", None)                         # (the launch groups cache their planner policy)
    ts = [t.clone().requires_grad_() for t in tables]
    sink = ops.SparseGradSink() if sparse == "sink" else None
    # the row stride padded to whole 128-byte lines (what the model classes do): an odd number of 64-byte features then still goes through the
    # full-line placement pass, whose last pair has one feature
    out, _, fm = ops.embed_apply(plan, ts, inputs, [None] * len(inputs), out_ld=g_out.shape[1], sparse_grad=sink if sink is not None else sparse)
    loss = (fm * g_fm).sum() if only_fm else (out * g_out).sum()         # only_fm: the concat gets NO gradient (an FM model's loss reads the logit)
    if fm is not None and not only_fm:
        loss = loss + (fm * g_fm).sum()
    loss.backward()
    torch.cuda.synchronize()
    if sink is not None:                 # the fused optimizer's sink: (keys, rows, counts) on the device -> one dense tensor per table (every key once)
        res = [torch.zeros_like(t) for t in ts]
        keys_seen = 0
        for e in sink.pending:
            nu = int(e["counts"][0].item()) if e["counts"] is not None else e["uniq"].numel()
            k, v = e["uniq"][:nu], e["values"][:nu]
            live = k >= 0
            k, v = k[live], v[live]
            assert torch.unique(k).numel() == k.numel()
            for ti in range(len(ts)):
                m = (k >> 40) == ti
                res[ti][(k[m] & ((1 << 40) - 1))] = v[m]
            keys_seen += k.numel()
        assert keys_seen > 0
        return res
    return [t.grad.coalesce() if sparse is True else t.grad for t in ts]


BWD_CASES = [
    # name, D, n_feats, rows, B, fm, kind
    ("c2_like_fm", 16, 26, 300000, 5000, True, "uniform"),
    ("fm_dup", 16, 7, 60000, 9000, True, "dup"),
    ("plain32", 32, 6, 200000, 4097, False, "uniform"),
    ("plain64_hot", 64, 3, 500000, 6000, False, "hot"),
    ("plain16_tiny", 16, 4, 150000, 31, False, "uniform"),
    ("fm_one", 16, 2, 140000, 1, True, "uniform"),
    ("plain16_1300", 16, 8, 200000, 1300, False, "uniform"),      # D = 16, no FM: the full-line placement pass <U, false, *>, even feature count
    ("plain16_4097_odd", 16, 5, 150000, 4097, False, "dup"),      # ... odd feature count (the last pair's upper half idles), many pairs and 3+ rows
    ("fm16_1300_odd", 16, 7, 160000, 1300, True, "uniform"),      # ... <U, true, *>
    ("fm16_logit_only", 16, 26, 250000, 3000, "only", "uniform"), # the FM model class: the loss reads the logit, the concat has no gradient
    ("fm32_logit_only", 32, 5, 150000, 2049, "only", "dup"),
]


@pytest.mark.parametrize("name,D,n,rows,B,fm,kind", BWD_CASES, ids=[c[0] for c in BWD_CASES])
@pytest.mark.parametrize("dest", ["row_sparse", "dense", "dense_one_call", "sink_one_call"])
def test_pairs_backward_equals_sorted_backward_and_float64(name, D, n, rows, B, fm, kind, dest, monkeypatch):
    rng = np.random.default_rng(len(name) * 31 + D)
    monkeypatch.setattr(ops, "DENSE_BWD_SORTED", True)      # dense destination: always the planned reduction
    # "dense": planned by sparse_plan at every size; "dense_one_call": plan + reduction in ONE library call that takes the planner as an argument
    # (nrx_embed_bwd_dense_planned: what default-mode launches below PLAN_AHEAD_MIN lookups and captured steps use)
    monkeypatch.setattr(ops, "PLAN_AHEAD_MIN", (1 << 40) if dest == "dense_one_call" else 0)
    monkeypatch.setattr(ops, "DENSE_LDS_MIN", 0)
    only_fm = fm == "only"
    fm = bool(fm)
    slots = [ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=int(fm)) for i in range(n)]
    plan = ops.EmbedPlan(slots, out_width=n * D, use_fm=fm)
    tables_np = [rng.standard_normal((rows, D)).astype(np.float32) for _ in range(n)]
    tables = [torch.from_numpy(t).to(DEV) for t in tables_np]
    ids_np = _case_ids(rng, kind, B, [rows] * n, n)
    for x in ids_np:
        x[2:4] = 7                                          # (the forward rejects out-of-range ids)
        x[:1] = 0
    inputs = [torch.from_numpy(x).to(DEV) for x in ids_np]
    ld = (n * D + 31) // 32 * 32
    g_out = torch.zeros((B, ld), dtype=torch.float32, device=DEV)
    g_out[:, :n * D] = torch.from_numpy(rng.standard_normal((B, n * D)).astype(np.float32)).to(DEV)
    g_fm = torch.from_numpy(rng.standard_normal((B,)).astype(np.float32)).to(DEV)
    sparse = True if dest == "row_sparse" else ("sink" if dest == "sink_one_call" else False)
    if dest == "sink_one_call":
        monkeypatch.setattr(ops, "SPARSE_SMALL_DET", False)      # (the planned reduction at every size, not the one-launch kernel of small batches)
    a = _grads(plan, tables, inputs, g_out, g_fm, True, sparse, monkeypatch, only_fm)
    b = _grads(plan, tables, inputs, g_out, g_fm, False, sparse, monkeypatch, only_fm)
    for x, y in zip(a, b):
        if sparse is True:
            assert torch.equal(x.indices(), y.indices())
            assert torch.equal(x.values().view(torch.int32), y.values().view(torch.int32))
        else:
            assert torch.equal(x.view(torch.int32), y.view(torch.int32))
    # float64 restatement: d loss / d table = index_add of the upstream rows (+ the FM term g_fm * (S - v) per factor column, g_fm for column 0)
    go = g_out.cpu().numpy().astype(np.float64) * (0.0 if only_fm else 1.0)
    up = [go[:, i * D:(i + 1) * D].copy() for i in range(n)]          # (the padding columns carry no feature)
    if fm:
        gf = g_fm.cpu().numpy().astype(np.float64)[:, None]
        rows_v = [tables_np[i].astype(np.float64)[ids_np[i]] for i in range(n)]
        S = sum(rows_v)
        for i in range(n):
            t = gf * (S - rows_v[i])
            t[:, 0] = gf[:, 0]
            up[i] += t
    for i in range(n):
        want = R.embedding_grad_dense(ids_np[i], up[i], rows)
        assert not want[0].any()                            # the padding row is looked up (ids 0) and gets no gradient
        for res in (a, b):                                  # the one-kernel plan + pair records, and the sorted plan: each against float64 directly
            got = (res[i].to_dense() if sparse is True else res[i]).cpu().numpy()
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5 * max(1.0, np.abs(want).max()))


def test_policy_follows_the_previous_batch(monkeypatch):
    """auto mode: the first batch of a launch group is planned by the sorted planner, which leaves the duplicate statistics; near-unique ids then
    take the one-kernel planner, a duplicate-heavy batch sends the group back."""
    monkeypatch.setattr(ops, "PLAN_LDS", "auto")
    rng = np.random.default_rng(3)
    n, B, rows = 4, 3000, 400000
    lens = (C.c_int64 * n)(*([B] * n))
    pol = ops.PlanPolicy(lens, (C.c_int32 * n)(*range(n)), (C.c_int64 * n)(*([rows] * n)), n, n, B * n)
    assert pol.eligible
    kinds = []
    for kind in ["uniform", "uniform", "uniform", "dup", "dup", "uniform", "uniform"]:
        src = rows if kind == "uniform" else 300
        ids = [torch.from_numpy(rng.integers(1, src, B)).to(DEV) for _ in range(n)]
        pl = ops.sparse_plan(ids, list(range(n)), [rows] * n, n, (1 << n) - 1, policy=pol)
        torch.cuda.synchronize()
        kinds.append(len(pl) == 8)
    #          sorted (no statistics yet), lds, lds, lds (found out it was slow), sorted, sorted (found near-unique again), lds
    assert kinds == [False, True, True, True, False, False, True]


def test_plan_lds_at_the_real_c2_counts():
    """BASELINE config 2 at full size: 26 tables x 1 000 000 rows, 65 536 uniform ids each (1.7 M lookups, 208 row ranges) -- unique rows, per-table
    bounds, dest words, pair records and walk rows against the definition (vectorised here: oracle.ref_np.sparse_plan + run lengths)."""
    rng = np.random.default_rng(2026)
    T, rows, B = 26, 1_000_000, 65_536
    ids = [rng.integers(1, rows, B) for _ in range(T)]
    for x in ids:
        x[:3] = 0
    got = _plan_lds(ids, list(range(T)), [rows] * T, T)
    order, uniq, seg, counts = R.sparse_plan(ids, list(range(T)), [rows] * T, T)
    nu, n = len(uniq), T * B
    ln = seg[1:] - seg[:-1]
    row = uniq & ((1 << 40) - 1)
    once, twice = (ln == 1) & (row != 0), (ln == 2) & (row != 0)
    assert np.array_equal(got["counts"], counts) and np.array_equal(got["uniq"][:nu], uniq)
    dest = np.full(n, -1, np.int32)
    dest[order[seg[:-1][once]]] = np.nonzero(once)[0]
    assert np.array_equal(got["dest"][:n], dest)
    pu = np.nonzero(twice)[0]
    want_pairs = np.stack([pu, order[seg[:-1][twice]], order[seg[:-1][twice] + 1]], axis=1).astype(np.int32)
    assert got["n_pairs"] == len(pu) and np.array_equal(got["pairs"][:len(pu), :3], want_pairs)
    wu = np.nonzero(~once & ~twice)[0]
    assert got["n_walk"] == len(wu) and np.array_equal(got["walk"][:len(wu)], wu.astype(np.int32))
    at = 0
    for u in wu:
        k = int(ln[u])
        assert got["seg"][u] == at and got["seg"][u + 1] == at + k and np.array_equal(got["order"][at:at + k], order[seg[u]:seg[u] + k])
        at += k


def test_plan_lds_ticket_order_and_more_ranges_than_compute_units(monkeypatch):
    """Launches of more row ranges than the device has compute units take their ranges in TICKET order (a block may wait only for blocks that
    have started); NRX_PLAN_LDS_XCD=0 forces that order for any launch.  Both against the definition, repeatedly on one control block."""
    lib = _lib.load()
    rng = np.random.default_rng(8)
    state = torch.zeros(lib.nrx_sparse_plan_lds_state_bytes(), dtype=torch.uint8, device=DEV)
    T, rows, B = 40, 2_000_000, 1500                       # 40 x 16 = 640 row ranges
    for _ in range(2):
        ids = _case_ids(rng, "uniform", B, [rows] * T, T)
        got = _plan_lds(ids, list(range(T)), [rows] * T, T, state=state)
        _check_against_definition(got, ids, list(range(T)), [rows] * T, T)
    monkeypatch.setenv("NRX_PLAN_LDS_XCD", "0")
    for kind in ("uniform", "dup", "hot"):
        rws = [300000, 131072, 600000]
        ids = _case_ids(rng, kind, 6000, rws, 3)
        got = _plan_lds(ids, [0, 1, 2], rws, 3, state=state)
        _check_against_definition(got, ids, [0, 1, 2], rws, 3)


@pytest.mark.parametrize("seed", range(12))
def test_plan_lds_random_shapes(seed):
    """Random launches inside the planner's shapes: 1-9 tables of 1 .. 600 000 rows (several features may share a table), batch 1 .. 7000, ids from
    uniform to heavily repeated, int32 or int64 -- the plan equals its definition."""
    rng = np.random.default_rng(1000 + seed)
    nt = int(rng.integers(1, 10))
    trow = [int(rng.choice([1, 2, 33, 5000, 131072, 131073, 262144, 600000])) for _ in range(nt)]
    nf = int(rng.integers(nt, nt + 4))
    tab = list(range(nt)) + [int(rng.integers(0, nt)) for _ in range(nf - nt)]
    rng.shuffle(tab)
    B = int(rng.choice([1, 2, 63, 64, 65, 1000, 4097, 7000]))
    rows = [trow[t] for t in tab]
    ids = []
    for r in rows:
        span = max(1, int(r * rng.choice([1.0, 0.3, 0.01])))
        ids.append(rng.integers(0, span, B))
    dtype = torch.int32 if seed % 2 else torch.int64
    got = _plan_lds(ids, tab, rows, nt, dtype)
    _check_against_definition(got, ids, tab, rows, nt)


def test_captured_step_takes_the_one_kernel_planner_and_replays_bit_for_bit(monkeypatch):
    """A step captured in a HIP graph (forward + row-sparse backward into a sink) after eager warm-up steps on near-unique ids: the capture bakes in
    the one-kernel planner (the policy's choice from the warm-up batches); replays on NEW ids give the same unique rows and row gradients, bit for
    bit, as the eager sorted planner on those ids."""
    monkeypatch.setattr(ops, "PLAN_LDS", "auto")
    n, D, rows, B = 6, 16, 300000, 5000           # (above the one-launch small kernel's 4096 lookups per table: the planned reduction)
    slots = [ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=1) for i in range(n)]
    plan = ops.EmbedPlan(slots, out_width=n * D, use_fm=True)
    gen = torch.Generator(device=DEV).manual_seed(4)
    tables = [torch.randn((rows, D), device=DEV, generator=gen).requires_grad_() for _ in range(n)]
    static_ids = [torch.randint(1, rows, (B,), device=DEV, generator=gen) for _ in range(n)]
    up = torch.randn((B, n * D), device=DEV, generator=gen)
    upf = torch.randn((B,), device=DEV, generator=gen)
    sink = ops.SparseGradSink()
    keep = {}

    def step():
        sink.clear()
        out, _, fm = ops.embed_apply(plan, tables, static_ids, [None] * n, sparse_grad=sink, index_check="off")
        ((out * up).sum() + (fm * upf).sum()).backward()
        e = sink.pending[0]
        keep["uniq"], keep["values"], keep["counts"] = e["uniq"], e["values"], e["counts"]

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):                       # batch 1: sorted planner + statistics; batches 2, 3: the one-kernel planner
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()                     # (the warm-up plans' statistics have landed in the policy's mapped words: GraphedStep syncs here too)
    pol = ops._group_policy(ops._sparse_group_cache(plan, tables)[0], B, n)
    assert pol is not None and pol.choose()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step()
    for trial in range(3):
        new = [torch.randint(1, rows, (B,), device=DEV, generator=gen) for _ in range(n)]
        for dst, src in zip(static_ids, new):
            dst.copy_(src)
        g.replay()
        torch.cuda.synchronize()
        nu = int(keep["counts"][0].item())
        ku, kv = keep["uniq"][:nu].clone(), keep["values"][:nu].clone()
        monkeypatch.setattr(ops, "PLAN_LDS", "0")
        plan.__dict__.pop("_sg", None)
        ref_sink = ops.SparseGradSink()
        out, _, fm = ops.embed_apply(plan, tables, new, [None] * n, sparse_grad=ref_sink, index_check="off")
        ((out * up).sum() + (fm * upf).sum()).backward()
        torch.cuda.synchronize()
        e = ref_sink.pending[0]
        assert int(e["counts"][0].item()) == nu and torch.equal(e["uniq"][:nu], ku)
        assert torch.equal(e["values"][:nu].view(torch.int32), kv.view(torch.int32))
        monkeypatch.setattr(ops, "PLAN_LDS", "auto")
        plan.__dict__.pop("_sg", None)
