#!/usr/bin/env python3
"""Per-operator timings at BASELINE shapes (B=65536): DCN v1/v2 forward+backward, embed backward, FM,
bag pooling.  Prints us, algorithmic GB/s (HBM-bound ops) or TFLOP/s (DCN-v2, MFMA-bound)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
from news_recsys_amd._lib import NRX_SPARSE, NRX_BAG_MASKED_MEAN
ops.set_index_check("off")
dev = torch.device("cuda:0")
B = 65536
only = set(sys.argv[1:])

def timeit(fn, steps=30, warm=3, settle_ms=60.0):
    """Mean launch time AFTER the clocks have settled: the op is repeated for >= settle_ms first.  Coming out of idle the
    chip ramps its core clock over tens of milliseconds; a 3 + 30-launch burst of a matrix-core kernel measures that ramp
    (DCN-v2 forward D=320: 145-155 us in the first 5 ms, 122 us settled -- profiles/r02_dcn_v2_phase_probe.txt); kernels
    bound by HBM read the same either way."""
    import time as _t
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = _t.perf_counter()
    while (_t.perf_counter() - t0) * 1e3 < settle_ms:
        for _ in range(4): fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(steps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / steps * 1e3

def want(name): return not only or name in only

if want("dcn_v2"):
    for D in (320, 112, 256, 512):
        x = torch.randn(B, D, device=dev); W = torch.randn(1, D, D, device=dev) / D ** 0.5; b = torch.zeros(1, D, device=dev)
        with torch.no_grad():
            us = timeit(lambda: ops.dcn_v2(x, W, b))
            us_ref = timeit(lambda: torch.relu(x * torch.addmm(b[0], x, W[0].t()) + x))
        fl = 2.0 * B * D * D + 3.0 * B * D
        print(f"dcn_v2 fwd  D={D:4d}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s ({fl / us / 1e6 / 157.3 * 100:5.1f}% of 157.3 fp32-matrix peak)   [rocBLAS+eltwise: {us_ref:8.1f} us]", flush=True)
        # backward of one layer: hand-written (prep + MFMA dgrad + MFMA wgrad) vs the torch composition it replaced (3 library
        # GEMMs incl. the lin recompute + elementwise)
        xg = x.clone().requires_grad_(True); Wg = W.clone().requires_grad_(True); bg = b.clone().requires_grad_(True)
        out = ops.dcn_v2(xg, Wg, bg); up = torch.randn_like(out)
        us_b = timeit(lambda: torch.autograd.grad(out, (xg, Wg, bg), up, retain_graph=True), steps=20)
        def torch_bwd():
            g = up * (out > 0)
            lin = torch.addmm(bg[0], xg, Wg[0].t())
            glin = g * xg
            return g * lin + g + glin @ Wg[0], glin.t() @ xg, glin.sum(0)
        with torch.no_grad():
            us_t = timeit(torch_bwd, steps=20)
        flb = 4.0 * B * D * D
        print(f"dcn_v2 bwd  D={D:4d}: {us_b:8.1f} us  {flb / us_b / 1e6:7.1f} TFLOP/s of its two GEMMs ({flb / us_b / 1e6 / 157.3 * 100:5.1f}% of peak)   [torch composition: {us_t:8.1f} us]", flush=True)
if want("dcn_v1"):
    for D, NL in ((320, 2), (320, 3), (112, 3)):
        x = torch.randn(B, D, device=dev); w = torch.randn(NL, D, device=dev) / D ** 0.5; b = torch.zeros(NL, D, device=dev)
        with torch.no_grad():
            us = timeit(lambda: ops.dcn_v1(x, w, b))
        print(f"dcn_v1 fwd  D={D} L={NL}: {us:8.1f} us  {2 * B * D * 4 / us / 1e3:7.1f} GB/s algorithmic (read x + write out)", flush=True)
        xg = x.clone().requires_grad_(True); wg = w.clone().requires_grad_(True); bg = b.clone().requires_grad_(True)
        out = ops.dcn_v1(xg, wg, bg); up = torch.randn_like(out)
        us = timeit(lambda: torch.autograd.grad(out, (xg, wg, bg), up, retain_graph=True))
        print(f"dcn_v1 bwd  D={D} L={NL}: {us:8.1f} us  {3 * B * D * 4 / us / 1e3:7.1f} GB/s algorithmic (read x,g + write gx)", flush=True)
if want("embed_bwd"):
    F, D, rows = 26, 16, 1_000_000
    gen = torch.Generator(device=dev).manual_seed(1)
    tables = [torch.randn(rows, D, device=dev).requires_grad_(True) for _ in range(F)]
    plan = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=1) for i in range(F)], out_width=F * D, use_fm=True)
    ids = [torch.randint(1, rows, (B,), device=dev, generator=gen) for _ in range(F)]
    out, _, fm = ops.embed_apply(plan, tables, ids, [None] * F)
    up = torch.randn_like(out); upf = torch.randn_like(fm)
    us = timeit(lambda: torch.autograd.grad((out, fm), tables, (up, upf), retain_graph=True), steps=10)
    print(f"embed bwd (C2, dense grads incl. 1.66 GB zero-fill + FM bwd): {us:8.1f} us", flush=True)
    from news_recsys_amd import _lib
    lib = _lib.load()
    grads = [torch.zeros_like(t) for t in tables]
    arr = ops._fill_features(plan, 0, F, grads, ids, [None] * F, table_ptrs=[g.data_ptr() for g in grads])
    st = torch.cuda.current_stream().cuda_stream
    us = timeit(lambda: lib.nrx_embed_bwd(arr, F, B, up.data_ptr(), F * D, None, 0, None, st))
    print(f"embed bwd scatter kernel alone (atomics into 26 x 1M x 16): {us:8.1f} us  {B * F * (8 + 64 + 2 * 64) / us / 1e3:7.1f} GB/s algorithmic (ids + g read + row RMW)", flush=True)
if want("train_step"):
    F, D, rows = 26, 16, 1_000_000
    gen = torch.Generator(device=dev).manual_seed(1)
    tables = [torch.randn(rows, D, device=dev).requires_grad_(True) for _ in range(F)]
    plan = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=1) for i in range(F)], out_width=F * D, use_fm=True)
    ids = [torch.randint(1, rows, (B,), device=dev, generator=gen) for _ in range(F)]
    for sg in (False, True):
        def step():
            out, _, fm = ops.embed_apply(plan, tables, ids, [None] * F, sparse_grad=sg)
            (out.sum() * 1e-6 + fm.sum()).backward()
            for t in tables: t.grad = None
        us = timeit(step, steps=10)
        print(f"C2 embed+FM forward+backward, sparse_grad={sg}: {us:9.1f} us  -> {B / us:7.2f} M impressions/s", flush=True)
if want("fm"):
    x = torch.randn(B, 416, device=dev)
    with torch.no_grad():
        us = timeit(lambda: ops.fm_interaction(x, 26, 16))
    print(f"fm standalone fwd [B,416]: {us:8.1f} us  {B * 416 * 4 / us / 1e3:7.1f} GB/s", flush=True)
if want("pool"):
    emb = torch.randn(B, 50, 16, device=dev); m = (torch.rand(B, 50, device=dev) < 0.6).float()
    with torch.no_grad():
        us = timeit(lambda: ops.bag_pool(emb, m))
    print(f"bag_pool standalone [B,50,16]: {us:8.1f} us  {B * 50 * 17 * 4 / us / 1e3:7.1f} GB/s", flush=True)

if want("metrics"):
    import time, numpy as np
    from news_recsys_amd.metrics import ranking_metrics
    from oracle import ref_np as R
    N, U = 2_000_000, 50_000
    gen = torch.Generator(device=dev).manual_seed(5)
    uid = torch.randint(1, U + 1, (N,), device=dev, generator=gen)
    sc = torch.rand(N, device=dev, generator=gen)
    lb = (torch.rand(N, device=dev, generator=gen) < 0.1 + 0.5 * sc).float()
    warm = list(range(1, U // 2))
    ranking_metrics(uid, sc, lb, warm); torch.cuda.synchronize()
    t0 = time.perf_counter(); res = ranking_metrics(uid, sc, lb, warm); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"validation metrics on device: {N} samples / {U} users: {dt * 1e3:8.1f} ms  ({N / dt / 1e6:6.1f} M samples/s)", flush=True)
    n2 = 200_000
    t0 = time.perf_counter(); ref = R.validation_metrics(uid[:n2].cpu().numpy(), sc[:n2].cpu().numpy(), lb[:n2].cpu().numpy(), warm); dt2 = time.perf_counter() - t0
    print(f"reference-style Python loop (oracle port, 1 thread): {n2} samples: {dt2 * 1e3:8.1f} ms  ({n2 / dt2 / 1e6:6.3f} M samples/s)", flush=True)

if want("topk"):
    import time, numpy as np
    from oracle import ref_c
    N, Q, d, k = 200_000, 65_536, 16, 10
    gen = torch.Generator(device=dev).manual_seed(9)
    items = torch.nn.functional.normalize(torch.randn(N, d, device=dev, generator=gen), dim=1)
    q = torch.nn.functional.normalize(torch.randn(Q, d, device=dev, generator=gen), dim=1)
    with torch.no_grad():
        us = timeit(lambda: ops.topk_ip(items, q, k), steps=5, warm=1)
        us_t = timeit(lambda: torch.topk(q @ items.T, k, dim=1), steps=2, warm=1)
    fl = 2.0 * N * Q * d
    print(f"topk_ip {Q} queries x {N} items x {d}, k={k}: {us / 1e3:8.2f} ms  {fl / us / 1e6:6.1f} TFLOP/s "
          f"({fl / us / 1e6 / 157.3 * 100:4.1f}% of fp32 peak)  {Q / us:6.2f} M queries/s  [torch matmul+topk {us_t / 1e3:8.2f} ms]", flush=True)
    off = torch.arange(Q + 1, device=dev, dtype=torch.int64) * 50
    ex = torch.sort(torch.randint(0, N, (Q, 50), device=dev, generator=gen), dim=1).values.reshape(-1)
    with torch.no_grad():
        us_e = timeit(lambda: ops.topk_ip(items, q, k, exclude=(off, ex)), steps=5, warm=1)
    print(f"  with 50 excluded items per query: {us_e / 1e3:8.2f} ms", flush=True)
    nq = 2048
    t0 = time.perf_counter(); ref_c.topk_ip(items.cpu().numpy(), q[:nq].cpu().numpy(), k); dt = time.perf_counter() - t0
    print(f"  C oracle (OpenMP, {ref_c.threads()} threads): {nq} queries {dt * 1e3:8.1f} ms  ({nq / dt / 1e6:6.3f} M queries/s)", flush=True)

if want("eager"):
    # What running the reference's own module code on this GPU costs (PyTorch-ROCm eager, restated: one
    # nn.Embedding lookup per feature, masked mean pooling, torch.cat, FM from sums -- base_model.py:262-308,
    # fm/model.py:18-26) next to the fused launch on the same inputs.
    import torch.nn.functional as Fn
    gen = torch.Generator(device=dev).manual_seed(3)
    F, D, rows = 26, 16, 1_000_000
    tables = [torch.randn(rows, D, device=dev) for _ in range(F)]
    ids = [torch.randint(1, rows, (B,), device=dev, generator=gen) for _ in range(F)]
    def eager_c2():
        embs = [Fn.embedding(i, t, padding_idx=0) for i, t in zip(ids, tables)]
        x = torch.cat(embs, dim=1)
        e = torch.stack(embs, dim=1)                               # [B, F, D]
        first = e[:, :, 0].sum(dim=1)
        v = e[:, :, 1:]
        fm = first + 0.5 * (v.sum(dim=1).pow(2) - v.pow(2).sum(dim=1)).sum(dim=1)
        return x, fm
    plan = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=1) for i in range(F)], out_width=F * D, use_fm=True)
    prep = ops.PreparedEmbed(plan, tables, ids, [None] * F)
    with torch.no_grad():
        x_ref, fm_ref = eager_c2()
        prep.run(); torch.cuda.synchronize()
        assert torch.equal(prep.out, x_ref) and torch.allclose(prep.fm, fm_ref, rtol=1e-4, atol=1e-4)
        us_e = timeit(eager_c2, steps=20)
        us_f = timeit(prep.run, steps=50)
    print(f"C2 forward (gather+concat+FM), PyTorch-ROCm eager (reference module code restated): {us_e:8.1f} us   fused HIP launch: {us_f:7.1f} us   x{us_e / us_f:5.1f}", flush=True)
    # C4-like tower input: user_id + history L=50 (masked mean) + item_id, D=16
    L = 50
    t_user = torch.randn(10_000_000, D, device=dev); t_item = torch.randn(200_000, D, device=dev)
    uid = torch.randint(1, 10_000_000, (B,), device=dev, generator=gen); iid = torch.randint(1, 200_000, (B,), device=dev, generator=gen)
    hist = torch.randint(1, 200_000, (B, L), device=dev, generator=gen); mask = (torch.rand(B, L, device=dev, generator=gen) < 0.6).float()
    def eager_c4():
        h = Fn.embedding(hist, t_item, padding_idx=0) * mask.unsqueeze(-1)
        h = h.sum(dim=1) / (mask.sum(dim=1, keepdim=True) + 1e-8)
        return torch.cat([Fn.embedding(iid, t_item), h, Fn.embedding(uid, t_user)], dim=1)
    plan4 = ops.EmbedPlan([ops.Slot("item_id", NRX_SPARSE, 0, D, 0, 0), ops.Slot("user_history", NRX_BAG_MASKED_MEAN, 0, D, L, D),
                           ops.Slot("user_id", NRX_SPARSE, 1, D, 0, 2 * D)], out_width=3 * D)
    prep4 = ops.PreparedEmbed(plan4, [t_item, t_user], [iid, hist, uid], [None, mask, None])
    with torch.no_grad():
        ref4 = eager_c4(); prep4.run(); torch.cuda.synchronize()
        torch.testing.assert_close(prep4.out, ref4, rtol=1e-5, atol=1e-6)
        us_e = timeit(eager_c4, steps=20)
        us_f = timeit(prep4.run, steps=50)
    print(f"C4 tower input (id + masked-mean history L=50 + id),  PyTorch-ROCm eager: {us_e:8.1f} us   fused HIP launch: {us_f:7.1f} us   x{us_e / us_f:5.1f}", flush=True)

if want("train_opt"):
    # C2 tables: forward + backward + OPTIMIZER step, three ways.  (a) the reference's semantics: dense table grads +
    # dense AdamW over every row; (b) row-sparse COO grads + torch.optim.SparseAdam; (c) the fused path: sorted
    # reduction left on the device + nrx_sparse_adam_step (no COO tensors, no host read).
    from news_recsys_amd.model.model_utils.optim import FusedSparseAdam
    F, D, rows = 26, 16, 1_000_000
    gen = torch.Generator(device=dev).manual_seed(1)
    plan = ops.EmbedPlan([ops.Slot(f"f{i}", NRX_SPARSE, i, D, 0, i * D, fm_field=1) for i in range(F)], out_width=F * D, use_fm=True)
    ids = [torch.randint(1, rows, (B,), device=dev, generator=gen) for _ in range(F)]
    up = torch.randn(B, F * D, device=dev, generator=gen) * 1e-3
    for mode in ("dense+AdamW", "coo+SparseAdam", "fused"):
        tables = [torch.randn(rows, D, device=dev).requires_grad_(True) for _ in range(F)]
        if mode == "dense+AdamW":
            opt = torch.optim.AdamW(tables, lr=1e-3); sg = False
        elif mode == "coo+SparseAdam":
            opt = torch.optim.SparseAdam(tables, lr=1e-3); sg = True
        else:
            sg = ops.SparseGradSink(); opt = FusedSparseAdam(sg, lr=1e-3)
        def step():
            out, _, fm = ops.embed_apply(plan, tables, ids, [None] * F, sparse_grad=sg)
            ((out * up).sum() + fm.sum()).backward()
            opt.step()
            if mode != "fused": opt.zero_grad(set_to_none=True)
        us = timeit(step, steps=10)
        print(f"C2 embed+FM fwd+bwd+optimizer, {mode:15s}: {us:9.1f} us  -> {B / us:7.2f} M impressions/s", flush=True)
        del tables, opt
        torch.cuda.empty_cache()

if want("widedeep"):
    # C5's single-GPU table set (27 tables <= 16M rows, D=32) gathered (a) as a plain concat (uniform kernel, what
    # `bench.py --workload c5` times) and (b) with the Wide&Deep split of widedeep/model.py:53-69: the 10 smallest tables
    # send column 0 to wide[B,10] and columns 1..31 to the deep concat (embed_fwd_uniform_wide: dword-aligned vector stores;
    # the generic kernel took 210 us here)
    rows_all = [int(round(1e3 * (5e5) ** (i / 39))) for i in range(40)]
    rows = [r for r in rows_all if r <= 16_000_000]
    D = 32
    gen = torch.Generator(device=dev).manual_seed(5)
    tables = [torch.randn(r, D, device=dev) for r in rows]
    ids = [torch.randint(1, r, (B,), device=dev, generator=gen) for r in rows]
    F_ = len(rows)
    plain = ops.EmbedPlan([ops.Slot(f"w{i}", NRX_SPARSE, i, D, 0, i * D) for i in range(F_)], out_width=F_ * D)
    slots, col = [], 0
    for i in range(F_):
        wide = i < 10                                      # rows ascend with i: the 10 smallest tables
        slots.append(ops.Slot(f"w{i}", NRX_SPARSE, i, D, 0, col, wide_col=i if wide else -1))
        col += D - 1 if wide else D
    split = ops.EmbedPlan(slots, out_width=col, wide_width=10)
    alg = F_ * (8 + 4 * D + 4 * D) * B
    for name, plan in (("plain concat (uniform kernel)", plain), ("wide split, 10 wide features (uniform+wide kernel)", split)):
        prep = ops.PreparedEmbed(plan, tables, ids, [None] * F_)
        with torch.no_grad():
            us = timeit(prep.run, steps=50)
        print(f"C5 single-GPU tables, {name:48s}: {us:7.1f} us   {alg / us / 1e3:7.1f} GB/s algorithmic  ({alg / us / 1e3 / 8000:.3f} of peak)", flush=True)

if want("c1shape"):
    # the reference's own deep config shape (train_cf_deep.yaml: user_id/item_id D=32, three D=16 features) at B=65536:
    # mixed dims -> generic kernel
    dims = [16, 32, 16, 16, 32]
    rows = [18, 65239, 270, 18, 94058]
    gen = torch.Generator(device=dev).manual_seed(6)
    tables = [torch.randn(r, d, device=dev) for r, d in zip(rows, dims)]
    ids = [torch.randint(1, r, (B,), device=dev, generator=gen) for r in rows]
    col, slots = 0, []
    for i, d in enumerate(dims):
        slots.append(ops.Slot(f"f{i}", NRX_SPARSE, i, d, 0, col)); col += d
    plan = ops.EmbedPlan(slots, out_width=col)
    prep = ops.PreparedEmbed(plan, tables, ids, [None] * 5)
    with torch.no_grad():
        us = timeit(prep.run, steps=100)
    alg = B * sum(8 + 8 * d for d in dims)
    print(f"C1-shaped gather (5 feats, dims 16/32 mixed, cache-resident tables), B=65536: {us:7.1f} us   {alg / us / 1e3:7.1f} GB/s algorithmic", flush=True)

if want("bag_csr"):
    # the DSSM user tower input (user_id + history pooled over the 200 k news table, D = 16, L = 50) with the history in the
    # reference's padded ids + mask form vs the CSR form (NRX_FEAT_BAG_CSR: ids [nnz] + offsets [B + 1]); lengths uniform in
    # [0, 50] (MIND-like: most users have short histories) and full length.  Same pooled values, bit for bit.
    from news_recsys_amd._lib import NRX_FEAT_BAG_CSR
    D, L = 16, 50
    gen = torch.Generator(device=dev).manual_seed(8)
    t_user = torch.randn(10_000_000, D, device=dev); t_item = torch.randn(200_000, D, device=dev)
    uid = torch.randint(1, 10_000_000, (B,), device=dev, generator=gen)
    for name, lens in (("lengths uniform in [0, 50]", torch.randint(0, L + 1, (B,), device=dev, generator=gen)),
                       ("every history full (50)", torch.full((B,), L, device=dev, dtype=torch.int64))):
        mask = (torch.arange(L, device=dev)[None] < lens[:, None]).float()
        hist = torch.randint(1, 200_000, (B, L), device=dev, generator=gen) * mask.long()
        offsets = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), lens.cumsum(0)])
        values = hist[mask.bool()].contiguous()
        pp = ops.EmbedPlan([ops.Slot("user_history", NRX_BAG_MASKED_MEAN, 0, D, L, 0), ops.Slot("user_id", NRX_SPARSE, 1, D, 0, D)], out_width=2 * D)
        pc = ops.EmbedPlan([ops.Slot("user_history", NRX_BAG_MASKED_MEAN, 0, D, L, 0, flags=NRX_FEAT_BAG_CSR), ops.Slot("user_id", NRX_SPARSE, 1, D, 0, D)],
                           out_width=2 * D)
        a = ops.PreparedEmbed(pp, [t_item, t_user], [hist, uid], [mask, None])
        c64 = ops.PreparedEmbed(pc, [t_item, t_user], [values, uid], [offsets, None])
        c32 = ops.PreparedEmbed(pc, [t_item, t_user], [values.int(), uid.int()], [offsets, None])
        with torch.no_grad():
            a.run(); c64.run(); c32.run(); torch.cuda.synchronize()
            assert torch.equal(a.out, c64.out) and torch.equal(a.out, c32.out)
            us = [timeit(x.run, steps=100) for x in (a, c64, c32)]
        print(f"user tower input, {name}: padded int64 ids + mask {us[0]:6.1f} us   CSR int64 {us[1]:6.1f} us   CSR int32 {us[2]:6.1f} us", flush=True)
