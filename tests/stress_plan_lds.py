#!/usr/bin/env python3
"""Dev (GPU box): random launches through the one-kernel planner (nrx_sparse_plan_lds) against its definition (oracle.ref_np.sparse_plan_pairs), with the helpers
of tests/test_plan_lds.py: 1 .. 40 features over 1 .. 40 tables of 2 .. 3 M rows (tables shared, unread tables), batches 1 .. 70 000, int32 / int64 ids, aligned
and misaligned id arrays, uniform / duplicate-heavy / one-range / hot-row ids -- every overflow path of the kernel (LDS cache, pair slots, LDS sort) is reached by
some of them; the state block is reused from launch to launch.
usage: python tests/stress_plan_lds.py [seconds=120] [seed=1]   (a checker like the tests next to it; not collected by pytest)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _poison
import ctypes as C
from news_recsys_amd import _lib
import test_plan_lds as T
lib = _lib.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, n_done, n_skipped, state = time.time(), 0, 0, None
kinds = ["uniform", "dup", "one_range", "hot"]
while time.time() - t0 < budget:
    _poison.poison()
    nt = int(rng.choice([1, 2, 5, 13, 26, 40]))
    nf = min(40, nt + int(rng.integers(0, 3)))
    tab = [int(x) for x in rng.permutation(nt)[:min(nf, nt)]] + [int(x) for x in rng.integers(0, nt, max(0, nf - nt))]
    rows_t = [int(rng.choice([2, 17, 1000, 131072, 131073, 300000, 1000000, 3000000])) for _ in range(nt)]
    rows = [rows_t[t] for t in tab]
    B = int(rng.choice([1, 7, 100, 4099, 20000, 70000]))
    if B * len(tab) > 1_500_000:
        continue
    n = len(tab)
    lens = (C.c_int64 * n)(*([B] * n))
    if lib.nrx_sparse_plan_lds_ok(lens, (C.c_int32 * n)(*tab), (C.c_int64 * n)(*rows), n, nt) != 1:
        n_skipped += 1
        continue
    kind = kinds[int(rng.integers(0, 4))]
    ids = T._case_ids(rng, kind, B, rows, n)
    dtype = torch.int32 if rng.integers(0, 2) and max(rows) < (1 << 31) else torch.int64
    if dtype is torch.int32:
        ids = [np.clip(x, -5, (1 << 31) - 1) for x in ids]
    got = T._plan_lds(ids, tab, rows, nt, dtype=dtype, misalign=bool(rng.integers(0, 2)), state=state)
    state = got["state"]
    try:
        T._check_against_definition(got, [np.asarray(x, np.int64) for x in ids], tab, rows, nt)
    except AssertionError:
        print("MISMATCH", dict(nt=nt, tab=tab, rows=rows, B=B, kind=kind, dtype=str(dtype)))
        raise
    n_done += 1
print(f"stress_plan_lds: {n_done} random launches equal to the definition ({n_skipped} shapes outside the planner's launches skipped; {time.time() - t0:.0f} s)")
