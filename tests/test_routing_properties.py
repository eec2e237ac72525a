"""Property tests (hypothesis) of the integer routing definitions in oracle/ref_np.py -- the
specification the HIP routing kernels are held to bit-exactly -- and, on a GPU, of the kernels
themselves against those definitions on hypothesis-generated shapes."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import ref_np as R

lens_st = st.lists(st.integers(min_value=0, max_value=700), min_size=1, max_size=9)


@settings(max_examples=60, deadline=None)
@given(lens=lens_st, world=st.integers(1, 9), slack=st.floats(0.0, 1.0), seed=st.integers(0, 2 ** 16))
def test_route_ids_definition_properties(lens, world, slack, seed):
    rng = np.random.default_rng(seed)
    arrays = [rng.integers(0, 5000, n) for n in lens]
    total = sum(lens)
    cap = max(1, int(total / world * (1 + slack)) + 1)
    send, slot, counts2d, worst = R.route_ids(arrays, world, cap)
    ids = np.concatenate(arrays) if total else np.zeros(0, np.int64)
    assert counts2d.shape == (world, len(lens)) and counts2d.sum() == total
    assert worst == (counts2d.sum(axis=1).max() if total else 0)
    placed = slot >= 0
    assert len(set(slot[placed].tolist())) == placed.sum()               # slots are unique
    assert np.array_equal(send[slot[placed]], ids[placed] // world)       # send buffer holds the local row
    assert np.all(slot[placed] // cap == ids[placed] % world)             # ... in the owner's block
    if worst <= cap:
        assert placed.all()
    for o in range(world):                                                # stable: source order kept per owner
        pos = np.flatnonzero((ids % world == o) & placed)
        assert np.all(np.diff(slot[pos]) > 0)
    # the owner can rebuild (feature, local row) of every slot from counts2d alone
    for o in range(world):
        j = 0
        for f, n in enumerate(counts2d[o]):
            src = np.flatnonzero((np.concatenate([np.full(a.size, k) for k, a in enumerate(arrays)]) == f) & (ids % world == o)) if total else []
            for k, p in enumerate(src):
                if j + k < cap:
                    assert slot[p] == o * cap + j + k
            j += int(n)


@settings(max_examples=40, deadline=None)
@given(n=st.integers(0, 3000), world=st.integers(1, 9), seed=st.integers(0, 2 ** 16))
def test_bucketize_definition_properties(n, world, seed):
    ids = np.random.default_rng(seed).integers(0, 10 ** 6, n)
    counts, perm = R.bucketize_by_owner(ids, world)
    assert counts.sum() == n and sorted(perm.tolist()) == list(range(n))
    owner = ids % world
    assert np.all(np.diff(owner[perm]) >= 0)
    start = 0
    for o in range(world):
        seg = perm[start:start + counts[o]]
        assert np.all(owner[seg] == o) and np.all(np.diff(seg) > 0)
        start += counts[o]


@pytest.mark.gpu
@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(lens=st.lists(st.integers(min_value=0, max_value=3000), min_size=1, max_size=12), world=st.integers(1, 8),
       slack=st.floats(-0.5, 0.5), seed=st.integers(0, 2 ** 16), i32=st.booleans())
def test_hip_route_ids_matches_definition(lens, world, slack, seed, i32):
    import torch
    from news_recsys_amd import ops
    rng = np.random.default_rng(seed)
    arrays = [rng.integers(0, 1 << 20, n) for n in lens]
    total = sum(lens)
    cap = max(1, int(total / world * (1 + slack)) + 1)
    dt = torch.int32 if i32 else torch.int64
    send, slot, counts2d, overflow = ops.route_ids([torch.from_numpy(a).cuda().to(dt) for a in arrays], world, cap)
    r_send, r_slot, r_counts, r_worst = R.route_ids(arrays, world, cap)
    assert np.array_equal(counts2d.cpu().numpy(), r_counts)
    assert int(overflow.item()) == r_worst
    assert np.array_equal(slot.cpu().numpy(), r_slot)
    valid = r_send >= 0
    assert np.array_equal(send.cpu().numpy()[valid], r_send[valid])


@pytest.mark.gpu
@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(n=st.integers(0, 20000), world=st.integers(1, 16), seed=st.integers(0, 2 ** 16))
def test_hip_bucketize_matches_definition(n, world, seed):
    import torch
    from news_recsys_amd import ops
    ids = np.random.default_rng(seed).integers(0, 1 << 30, n)
    counts, local_rows, slot = ops.bucketize_by_owner(torch.from_numpy(ids).cuda(), world)
    c_ref, perm = R.bucketize_by_owner(ids, world)
    assert np.array_equal(counts.cpu().numpy(), c_ref)
    assert np.array_equal(local_rows.cpu().numpy(), (ids // world)[perm])


# ----------------------------------------------------------------------------- sparse-backward planning
@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    dict(n=[0], rows=[5], tab=[0], nt=1), dict(n=[1], rows=[1], tab=[0], nt=1),
    dict(n=[300, 300, 77], rows=[50, 50, 9], tab=[0, 1, 1], nt=3),
    dict(n=[5000] * 26, rows=[1000] * 26, tab=list(range(26)), nt=26),
    dict(n=[4096, 8192], rows=[1 << 20, 1 << 27], tab=[3, 1], nt=5),            # 3 + 27 bits: 32-bit keys
    dict(n=[4096, 100], rows=[(1 << 31) + 5, 7], tab=[0, 9], nt=10),            # 36 bits: 64-bit keys
    dict(n=[70000, 30000, 5000], rows=[1 << 24, 1 << 24, 300], tab=[0, 2, 0], nt=3),     # many tiles per table, 2 x 12-bit digits,
                                                                                # a table shared by non-adjacent features
    dict(n=[50000, 50000], rows=[1 << 20, 200000], tab=[1, 0], nt=2, zipf=1.05),          # skewed ids: long runs of one row
    dict(n=[200000, 4097, 140000], rows=[1 << 19, 11, 1 << 19], tab=[1, 0, 1], nt=2),    # a segment of 83 tiles: chunked scans
    dict(n=[9000, 4097], rows=[1 << 38, 11], tab=[1, 0], nt=2),                 # 38 row bits: four digit passes, 64-bit keys
    dict(n=[9000, 4097], rows=[1 << 21, 11], tab=[69, 0], nt=70),               # more tables than segments: library sort
    dict(n=[60000, 65536, 9000], rows=[1 << 27, (1 << 24) + 77, 500], tab=[0, 2, 1], nt=3),    # three LSD passes: the MSD form by default (17 / 15 low bits
                                                                                # through the rank sort; the 500-row table done by the one scatter)
    dict(n=[50000, 30000], rows=[1 << 27, 1 << 22], tab=[0, 1], nt=2, zipf=1.05),             # ... skewed: bins of thousands of entries -> the work list,
                                                                                # two streaming passes over 17 low bits
])
@pytest.mark.parametrize("dtype", ["int64", "int32"])
@pytest.mark.parametrize("sort", ["segmented", "segmented-bins", "rocprim", "msd", "lsd"])
def test_sparse_plan_bit_exact(case, dtype, sort, monkeypatch):
    """nrx_sparse_plan == its definition (oracle.ref_np.sparse_plan): stable order, unique keys, segment
    starts, per-table bounds -- including out-of-range / negative ids (row 0) and tables with no lookups."""
    import torch
    from news_recsys_amd import ops
    from oracle import ref_np as R
    if sort != "segmented":
        monkeypatch.setenv("NRX_PLAN_SORT", sort)                  # read by the library on every call ("segmented-bins": long segments
    else:                                                          # through seg_scan_bins, the path beyond 128 chunks per segment)
        monkeypatch.delenv("NRX_PLAN_SORT", raising=False)
    rng = np.random.default_rng(sum(case["n"]) + case["nt"])
    ids = []
    for n, r in zip(case["n"], case["rows"]):
        hi = min(r, (1 << 31) - 1) if dtype == "int32" else r
        if case.get("zipf"):
            x = np.minimum(rng.zipf(case["zipf"], n) - 1, max(hi, 1) - 1).astype(dtype)
        else:
            x = rng.integers(0, max(hi, 1), n).astype(dtype)
        if n > 10:
            x[3] = -1
            if hi == r and r < (1 << 31) - 10:
                x[5] = r + 3                                       # past the table: falls on row 0
            x[7:10] = x[6]                                         # duplicates
        ids.append(x)
    order, uniq, seg, counts = ops.sparse_plan([torch.from_numpy(x).to("cuda:0") for x in ids], case["tab"], case["rows"], case["nt"])
    o_r, u_r, s_r, c_r = R.sparse_plan(ids, case["tab"], case["rows"], case["nt"])
    c = counts.cpu().numpy()
    assert np.array_equal(c, c_r)
    nu = int(c[0])
    assert np.array_equal(order.cpu().numpy(), o_r)
    assert np.array_equal(uniq.cpu().numpy()[:nu], u_r)
    assert np.array_equal(seg.cpu().numpy()[:nu + 1], s_r)
    # the placement form (nrx_sparse_plan_place): same plan + dest / walk, for "every feature placeable" and for a partial mask
    nf = len(ids)
    for feats in (list(range(nf)), [f for f in range(nf) if f % 2 == 0]):
        mask = sum(1 << f for f in feats)
        o2, u2, s2, c2, dest, walk, n_walk = ops.sparse_plan([torch.from_numpy(x).to("cuda:0") for x in ids], case["tab"], case["rows"],
                                                             case["nt"], place_feats=mask)
        _, _, _, _, d_r, w_r = R.sparse_plan_place(ids, case["tab"], case["rows"], case["nt"], feats)
        assert np.array_equal(c2.cpu().numpy(), c_r) and np.array_equal(o2.cpu().numpy(), o_r)
        assert np.array_equal(u2.cpu().numpy()[:nu], u_r) and np.array_equal(s2.cpu().numpy()[:nu + 1], s_r)
        assert int(n_walk.item()) == len(w_r)
        assert np.array_equal(walk.cpu().numpy()[:len(w_r)], w_r)
        able = np.isin(np.repeat(np.arange(nf), [len(x) for x in ids]), feats)          # dest is written for placeable lookups only
        assert np.array_equal(dest.cpu().numpy()[:len(d_r)][able], d_r[able]) and np.all(d_r[~able] == -1)


@pytest.mark.gpu
@settings(max_examples=30, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(lens=st.lists(st.integers(min_value=0, max_value=9000), min_size=1, max_size=7),
       row_bits=st.lists(st.integers(min_value=1, max_value=26), min_size=7, max_size=7),
       tabs=st.lists(st.integers(min_value=0, max_value=4), min_size=7, max_size=7),
       skew=st.booleans(), seed=st.integers(0, 2 ** 16), msd=st.booleans())
def test_sparse_plan_segmented_sort_matches_definition(lens, row_bits, tabs, skew, seed, msd):
    """The table-segmented planner sort on hypothesis-generated shapes: any mix of segment lengths (empty segments, tiles that
    hold several features, tables shared by non-adjacent features), digit plans from 1 to 26 row bits, uniform and heavily
    repeated ids -- same plan as the definition, bit for bit."""
    import os
    import torch
    from news_recsys_amd import ops
    # msd: the one-scatter + bin-sort form of the planner forced wherever the structure allows (the library reads the variable per call);
    # skew then sends the repeated row's bin through the work-list path
    if msd:
        os.environ["NRX_PLAN_SORT"] = "msd"
    else:
        os.environ.pop("NRX_PLAN_SORT", None)
    rng = np.random.default_rng(seed)
    n = len(lens)
    tab = tabs[:n]
    nt = max(tab) + 1
    rows_of_table = [1 << row_bits[t] for t in range(nt)]
    rows = [rows_of_table[t] - (seed % 3 if rows_of_table[t] > 4 else 0) for t in tab]       # not only powers of two
    rows = [max(rows[i], 1) for i in range(n)]
    for t in range(nt):                                     # features of one table agree on its row count
        r = min(rows[i] for i in range(n) if tab[i] == t) if any(tab[i] == t for i in range(n)) else 1
        rows = [r if tab[i] == t else rows[i] for i in range(n)]
    ids = []
    for ln, r in zip(lens, rows):
        x = rng.integers(0, r, ln)
        if skew and ln:
            x = np.where(rng.random(ln) < 0.6, x[0], x)     # 60 % of the lookups hit one row
        ids.append(x.astype(np.int64))
    feats = [f for f in range(n) if (seed >> f) & 1] if seed % 4 else list(range(n))
    order, uniq, seg, counts, dest, walk, n_walk = ops.sparse_plan([torch.from_numpy(x).to("cuda:0") for x in ids], tab, rows, nt,
                                                                   place_feats=sum(1 << f for f in feats))
    o_r, u_r, s_r, c_r, d_r, w_r = R.sparse_plan_place(ids, tab, rows, nt, feats)
    c = counts.cpu().numpy()
    assert np.array_equal(c, c_r)
    nu = int(c[0])
    assert np.array_equal(order.cpu().numpy(), o_r)
    assert np.array_equal(uniq.cpu().numpy()[:nu], u_r)
    assert np.array_equal(seg.cpu().numpy()[:nu + 1], s_r)
    assert int(n_walk.item()) == len(w_r) and np.array_equal(walk.cpu().numpy()[:len(w_r)], w_r)
    able = np.isin(np.repeat(np.arange(n), [len(x) for x in ids]), feats)              # dest is written for placeable lookups only
    assert np.array_equal(dest.cpu().numpy()[:len(d_r)][able], d_r[able]) and np.all(d_r[~able] == -1)
    os.environ.pop("NRX_PLAN_SORT", None)


@pytest.mark.gpu
@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(lens=st.lists(st.integers(min_value=0, max_value=9000), min_size=1, max_size=6),
       big=st.integers(min_value=0, max_value=300_000),
       row_bits=st.lists(st.integers(min_value=1, max_value=26), min_size=7, max_size=7),
       tabs=st.lists(st.integers(min_value=0, max_value=3), min_size=7, max_size=7),
       pad=st.sampled_from([0.0, 0.5, 0.9, 1.0]), seed=st.integers(0, 2 ** 16),
       sort=st.sampled_from([None, "msd", "lsd", "segmented-bins"]), place=st.booleans(), idx32=st.booleans())
def test_sparse_plan_ex_padding_split_matches_definition(lens, big, row_bits, tabs, pad, seed, sort, place, idx32):
    """nrx_sparse_plan_ex with NRX_PLAN_SPLIT_PADDING (the padding lookups set aside before the sort, every pass on the live lookups only) ==
    the definition, bit for bit, like the unsplit plan: padded histories (a share `pad` of every feature's ids is 0; 1.0: a table that sees
    padding only), one long feature (chunked segments), every sort form, with and without the placement outputs; and the statistics it leaves
    (unique rows, walk rows, lookups, padding lookups) are this plan's."""
    import os
    import torch
    from news_recsys_amd import ops
    if sort:
        os.environ["NRX_PLAN_SORT"] = sort
    else:
        os.environ.pop("NRX_PLAN_SORT", None)
    rng = np.random.default_rng(seed)
    lens = list(lens) + [big]
    n = len(lens)
    tab = tabs[:n]
    nt = max(tab) + 1
    rows_of_table = [max((1 << row_bits[t]) - (seed % 3 if row_bits[t] > 2 else 0), 1) for t in range(nt)]
    rows = [rows_of_table[t] for t in tab]
    ids = []
    for ln, r in zip(lens, rows):
        x = rng.integers(0, r, ln)
        x = np.where(rng.random(ln) < pad, 0, x)
        ids.append(x.astype(np.int32 if idx32 else np.int64))
    feats = [f for f in range(n) if (seed >> f) & 1] if seed % 4 else list(range(n))
    total = sum(lens)
    pol = ops.PadPolicy(total)
    prev = ops.PAD_SPLIT
    ops.PAD_SPLIT = "1"
    try:
        res = ops.sparse_plan([torch.from_numpy(x).to("cuda:0") for x in ids], tab, rows, nt,
                              place_feats=sum(1 << f for f in feats) if place else None, pad=pol)
        torch.cuda.synchronize()
    finally:
        ops.PAD_SPLIT = prev
        os.environ.pop("NRX_PLAN_SORT", None)
    if total == 0:
        return
    ids64 = [x.astype(np.int64) for x in ids]
    o_r, u_r, s_r, c_r, d_r, w_r = R.sparse_plan_place(ids64, tab, rows, nt, feats)
    order, uniq, seg, counts = res[:4]
    c = counts.cpu().numpy()
    assert np.array_equal(c, c_r)
    nu = int(c[0])
    assert np.array_equal(order.cpu().numpy(), o_r)
    assert np.array_equal(uniq.cpu().numpy()[:nu], u_r)
    assert np.array_equal(seg.cpu().numpy()[:nu + 1], s_r)
    if place:
        dest, walk, n_walk = res[4:]
        assert int(n_walk.item()) == len(w_r) and np.array_equal(walk.cpu().numpy()[:len(w_r)], w_r)
        able = np.isin(np.repeat(np.arange(n), lens), feats)
        assert np.array_equal(dest.cpu().numpy()[:len(d_r)][able], d_r[able])
    st_ = pol.stats.numpy()
    assert st_[0] == nu and st_[3] == total and st_[4] == sum(int((x == 0).sum()) for x in ids)
    assert st_[1] == (len(w_r) if place else -1)


@pytest.mark.gpu
@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(nt=st.sampled_from([5, 17, 40, 61]), lens=st.lists(st.integers(min_value=0, max_value=6000), min_size=64, max_size=64),
       big=st.integers(min_value=0, max_value=200_000), rb=st.lists(st.sampled_from([1, 3, 9, 14, 20, 26, 29, 31]), min_size=64, max_size=64),
       shared=st.booleans(), seed=st.integers(0, 2 ** 16), sort=st.sampled_from([None, "msd", "lsd", "segmented-bins"]), place=st.booleans(),
       split=st.booleans(), skew=st.booleans())
def test_sparse_plan_segment_local_keys_match_definition(nt, lens, big, rb, shared, seed, sort, place, split, skew):
    """Launches whose table + row bits exceed 32 (many tables next to one of 2^29 .. 2^31 rows: the Wide&Deep shape): the sorted pairs carry the row
    only and the table of a sorted position is the run it lies in (PlaceInfo::seg_off) -- same plan as the definition, bit for bit, and as the
    64-bit-key form (NRX_PLAN_SEGKEY=0): runs that end inside a tile, empty tables, tables fed by two features, every sort form, with and
    without placement and the padding split."""
    import os
    import torch
    from news_recsys_amd import ops
    rng = np.random.default_rng(seed)
    rows_t = [max((1 << rb[t]) - (seed % 3 if rb[t] > 2 else 0), 1) for t in range(nt)]
    rows_t[seed % nt] = (1 << 31) - 5 if seed % 2 else (1 << 29) + 3          # one table that alone needs 30 .. 31 row bits
    tab = list(range(nt)) + ([int(x) for x in rng.integers(0, nt, 3)] if shared else [])       # (features sharing a table)
    n = len(tab)
    ln = [lens[i % 64] for i in range(n)]
    ln[seed % n] = big
    if seed % 5 == 0:
        ln[(seed + 1) % n] = 0                                                  # an empty feature
    rows = [rows_t[t] for t in tab]
    ids = []
    for l_, r in zip(ln, rows):
        x = rng.integers(0, r, l_)
        if skew and l_:
            x = np.where(rng.random(l_) < 0.5, x[0], x)
        if split and l_:
            x = np.where(rng.random(l_) < 0.4, 0, x)
        ids.append(x.astype(np.int64))
    total = sum(ln)
    if total == 0:
        return
    feats = [f for f in range(n) if (seed >> (f % 16)) & 1] if seed % 4 else list(range(n))
    dev_ids = [torch.from_numpy(x).to("cuda:0") for x in ids]
    o_r, u_r, s_r, c_r, d_r, w_r = R.sparse_plan_place(ids, tab, rows, nt, feats)
    prev = ops.PAD_SPLIT
    try:
        if sort:
            os.environ["NRX_PLAN_SORT"] = sort
        ops.PAD_SPLIT = "1" if split else "0"
        for segkey in ("1", "0"):
            os.environ["NRX_PLAN_SEGKEY"] = segkey
            res = ops.sparse_plan(dev_ids, tab, rows, nt, place_feats=sum(1 << f for f in feats) if place else None, pad=ops.PadPolicy(total))
            torch.cuda.synchronize()
            order, uniq, seg, counts = res[:4]
            c = counts.cpu().numpy()
            assert np.array_equal(c, c_r), segkey
            nu = int(c[0])
            assert np.array_equal(order.cpu().numpy(), o_r), segkey
            assert np.array_equal(uniq.cpu().numpy()[:nu], u_r), segkey
            assert np.array_equal(seg.cpu().numpy()[:nu + 1], s_r), segkey
            if place:
                dest, walk, n_walk = res[4:]
                assert int(n_walk.item()) == len(w_r) and np.array_equal(walk.cpu().numpy()[:len(w_r)], w_r), segkey
                able = np.isin(np.repeat(np.arange(n), ln), feats)
                assert np.array_equal(dest.cpu().numpy()[:len(d_r)][able], d_r[able]), segkey
    finally:
        ops.PAD_SPLIT = prev
        os.environ.pop("NRX_PLAN_SORT", None)
        os.environ.pop("NRX_PLAN_SEGKEY", None)


@pytest.mark.gpu
def test_sparse_plan_ex_rejects_half_a_placement_and_unknown_flags():
    """nrx_sparse_plan_ex: dest / walk / n_walk go together; flag bits it does not know are an error, not ignored (NRX_ERR_BAD_ARG, nothing
    enqueued); with all three null it is nrx_sparse_plan."""
    import ctypes as C
    import torch
    from news_recsys_amd import _lib
    lib = _lib.load()
    dev = "cuda:0"
    ids = torch.tensor([3, 0, 3, 5, 0, 1], dtype=torch.int64, device=dev)
    n = ids.numel()
    ptrs, lens = (C.c_void_p * 1)(ids.data_ptr()), (C.c_int64 * 1)(n)
    tof, rws = (C.c_int32 * 1)(0), (C.c_int64 * 1)(8)
    order, uniq, seg = (torch.empty(n + 1, dtype=torch.int64, device=dev) for _ in range(3))
    counts = torch.empty(3, dtype=torch.int64, device=dev)
    dest, walk = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
    n_walk = torch.empty(1, dtype=torch.int64, device=dev)
    stats = torch.zeros(5, dtype=torch.int64, device=dev)
    ws = torch.empty(lib.nrx_sparse_plan_workspace(n), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def call(flags, d, w, nw):
        return lib.nrx_sparse_plan_ex(ptrs, lens, tof, rws, 1, 64, 1, 1, flags, order.data_ptr(), uniq.data_ptr(), seg.data_ptr(), counts.data_ptr(),
                                      d, w, nw, None, None, stats.data_ptr(), ws.data_ptr(), st)
    assert call(0, dest.data_ptr(), None, n_walk.data_ptr()) == _lib.NRX_ERR_BAD_ARG
    assert call(4, dest.data_ptr(), walk.data_ptr(), n_walk.data_ptr()) == _lib.NRX_ERR_BAD_ARG          # an unknown flag bit
    assert call(_lib.NRX_PLAN_PAIRS, dest.data_ptr(), walk.data_ptr(), n_walk.data_ptr()) == _lib.NRX_ERR_BAD_ARG      # pair records asked for, no place to put them
    for flags in (0, _lib.NRX_PLAN_SPLIT_PADDING):
        assert call(flags, None, None, None) == 0, lib.nrx_last_error()
        torch.cuda.synchronize()
        o_r, u_r, s_r, c_r = R.sparse_plan([ids.cpu().numpy()], [0], [8], 1)
        assert np.array_equal(counts.cpu().numpy(), c_r) and np.array_equal(order.cpu().numpy()[:n], o_r)
        assert stats.tolist() == [int(c_r[0]), -1, -1, n, 2]
        assert call(flags, dest.data_ptr(), walk.data_ptr(), n_walk.data_ptr()) == 0, lib.nrx_last_error()
        torch.cuda.synchronize()
        _, _, _, _, d_r, w_r = R.sparse_plan_place([ids.cpu().numpy()], [0], [8], 1, [0])
        assert np.array_equal(dest.cpu().numpy(), d_r) and int(n_walk.item()) == len(w_r) and stats.tolist()[1] == len(w_r)


def test_oracle_sparse_plan_place_definition():
    """The placement definition itself, on a case small enough to read: dest names the unique index of a row looked up once
    (not row 0, placeable feature), walk lists every other unique row."""
    #           feature 0 (table 0)   feature 1 (table 1, not placeable)   feature 2 (table 0)
    ids = [np.array([5, 7, 0, 9]), np.array([3, 3, 4]), np.array([7, 2, 11])]
    order, uniq, seg, counts, dest, walk = R.sparse_plan_place(ids, [0, 1, 0], [16, 8, 16], 2, place_feats=[0, 2])
    rows = uniq & ((1 << 40) - 1)
    tabs = uniq >> 40
    assert list(zip(tabs.tolist(), rows.tolist())) == [(0, 0), (0, 2), (0, 5), (0, 7), (0, 9), (0, 11), (1, 3), (1, 4)]
    #            p:  0  1   2   3 | 4   5   6 | 7   8  9
    assert dest.tolist() == [2, -1, -1, 4, -1, -1, -1, -1, 1, 5]          # row 7 twice, row 0 = padding, table 1 not placeable
    assert walk.tolist() == [0, 3, 6, 7]
    # every unique row is either placed exactly once or walked
    placed = sorted(d for d in dest.tolist() if d >= 0)
    assert sorted(placed + walk.tolist()) == list(range(len(uniq)))


def test_oracle_sparse_plan_pairs_definition():
    """The one-kernel planner's definition on a case small enough to read: once -> unique index, twice -> one pair record,
    three times / the padding row -> the walk list with its lookups in ascending order."""
    #           feature 0 (table 0)      feature 1 (table 0)   feature 2 (table 1)
    ids = [np.array([5, 7, 0, 9, 7]), np.array([7, 2, 9, 0, 11]), np.array([3, 3, 3, 4, 6])]
    uniq, counts, dest, pairs, walk, walk_lookups = R.sparse_plan_pairs(ids, [0, 0, 1], [16, 16, 8], 2)
    rows, tabs = uniq & ((1 << 40) - 1), uniq >> 40
    assert list(zip(tabs.tolist(), rows.tolist())) == [(0, 0), (0, 2), (0, 5), (0, 7), (0, 9), (0, 11), (1, 3), (1, 4), (1, 6)]
    assert counts.tolist() == [9, 0, 6, 9]
    #            p:  0   1   2   3   4 | 5  6   7   8  9 | 10  11  12  13 14
    assert dest.tolist() == [2, -1, -1, -1, -1, -1, 1, -1, -1, 5, -1, -1, -1, 7, 8]     # row 9 twice (u = 4): a pair; row 7 three times: walked
    assert pairs.tolist() == [[4, 3, 7]]
    assert walk.tolist() == [0, 3, 6]
    assert walk_lookups[0].tolist() == [2, 8] and walk_lookups[3].tolist() == [1, 4, 5] and walk_lookups[6].tolist() == [10, 11, 12]
