#!/usr/bin/env python3
"""Dev (GPU box): random launches through the sorted planner in every form -- padding split on / off, segment-local keys on / off, LSD / MSD /
scan-bins sorts, with and without placement -- against the definition (oracle.ref_np.sparse_plan_place), larger than the hypothesis cases of
tests/test_routing_properties.py (several chunked segments per launch).  usage: python tests/stress_plan.py [seconds=120] [seed=1]   (a checker like the tests next to it: it may use the oracle; not collected by pytest)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _poison
from news_recsys_amd import ops
from oracle import ref_np as R
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ops.PLAN_PAIRS = True          # (the Python layer's default leaves the sorted planner's pair records off; checked here all the same)
t0, n_done = time.time(), 0
while time.time() - t0 < budget:
    _poison.poison()
    nt = int(rng.choice([1, 2, 3, 7, 26, 40, 63]))
    nf = min(64, nt + int(rng.integers(0, 3)))
    tab = list(range(nt)) + [int(x) for x in rng.integers(0, nt, nf - nt)]
    rb = rng.choice([1, 4, 10, 14, 18, 20, 24, 27, 30, 31], nt)
    rows_t = [max((1 << int(b)) - int(rng.integers(0, 3)), 1) for b in rb]
    lens = [int(x) for x in rng.choice([0, 1, 17, 900, 4096, 4097, 20000, 140000, 300000, 700000], nf, p=[.05, .05, .1, .2, .1, .1, .2, .1, .07, .03])]
    rows = [rows_t[t] for t in tab]
    pad, skew = float(rng.choice([0.0, 0.3, 0.9])), bool(rng.integers(0, 2))
    ids = []
    for l_, r in zip(lens, rows):
        x = rng.integers(0, r, l_)
        if skew and l_:
            x = np.where(rng.random(l_) < 0.4, x[0], x)
        if pad and l_:
            x = np.where(rng.random(l_) < pad, 0, x)
        ids.append(x.astype(np.int64 if rng.integers(0, 2) else np.int32))
    if len({x.dtype for x in ids}) > 1:
        ids = [x.astype(np.int64) for x in ids]
    total = sum(lens)
    if total == 0 or total >= (1 << 22):
        continue
    feats = [f for f in range(nf) if rng.integers(0, 2)] or [0]
    place = bool(rng.integers(0, 2))
    ids64 = [x.astype(np.int64) for x in ids]
    pairs_def = None
    o_r, u_r, s_r, c_r, d_r, w_r = R.sparse_plan_place(ids64, tab, rows, nt, feats)
    dev_ids = [torch.from_numpy(x).to("cuda:0") for x in ids]
    for sort in (None, "msd", "lsd", "segmented-bins"):
        for split in ("1", "0"):
            for segkey in ("1", "0"):
                if sort:
                    os.environ["NRX_PLAN_SORT"] = sort
                else:
                    os.environ.pop("NRX_PLAN_SORT", None)
                os.environ["NRX_PLAN_SEGKEY"] = segkey
                ops.PAD_SPLIT = split
                res = ops.sparse_plan(dev_ids, tab, rows, nt, place_feats=sum(1 << f for f in feats) if place else None, pad=ops.PadPolicy(total))
                torch.cuda.synchronize()
                c = res[3].cpu().numpy()
                nu = int(c_r[0])
                ok = (np.array_equal(c, c_r) and np.array_equal(res[0].cpu().numpy(), o_r) and np.array_equal(res[1].cpu().numpy()[:nu], u_r)
                      and np.array_equal(res[2].cpu().numpy()[:nu + 1], s_r))
                if ok and place:
                    able = np.isin(np.repeat(np.arange(nf), lens), feats)
                    ok = (int(res[6].item()) == len(w_r) and np.array_equal(res[5].cpu().numpy()[:len(w_r)], w_r)
                          and np.array_equal(res[4].cpu().numpy()[:len(d_r)][able], d_r[able]))
                if not ok:
                    print("MISMATCH", dict(nt=nt, tab=tab, rows=rows, lens=lens, pad=pad, skew=skew, sort=sort, split=split, segkey=segkey, place=place, feats=feats))
                    sys.exit(1)
                if split == "0":                       # ... and with pair records (NRX_PLAN_PAIRS: every feature placeable), against sparse_plan_pairs
                    res = ops.sparse_plan(dev_ids, tab, rows, nt, place_feats=(1 << nf) - 1, pairs=True)
                    torch.cuda.synchronize()
                    if pairs_def is None:
                        pairs_def = R.sparse_plan_pairs(ids64, tab, rows, nt)
                    u_p, c_p, d_p, p_p, w_p, wl_p = pairs_def
                    nu = len(u_p)
                    ok = (len(res) == 8 and np.array_equal(res[3].cpu().numpy(), c_p) and np.array_equal(res[1].cpu().numpy()[:nu], u_p)
                          and np.array_equal(res[0].cpu().numpy(), o_r) and np.array_equal(res[2].cpu().numpy()[:nu + 1], s_r)
                          and np.array_equal(res[4].cpu().numpy()[:total], d_p) and int(res[6][0].item()) == len(w_p)
                          and np.array_equal(res[5].cpu().numpy()[:len(w_p)], w_p) and int(res[6][1].item()) == len(p_p)
                          and np.array_equal(res[7].cpu().numpy()[:len(p_p), :3], p_p))
                    if not ok:
                        which = dict(n=len(res), counts=np.array_equal(res[3].cpu().numpy(), c_p), uniq=np.array_equal(res[1].cpu().numpy()[:nu], u_p),
                                     order=np.array_equal(res[0].cpu().numpy(), o_r), seg=np.array_equal(res[2].cpu().numpy()[:nu + 1], s_r),
                                     dest=np.array_equal(res[4].cpu().numpy()[:total], d_p), n_walk=(res[6].tolist(), len(w_p), len(p_p)))
                        print("MISMATCH (pair records)", which, dict(nt=nt, nf=nf, total=total, pad=pad, skew=skew, sort=sort, segkey=segkey))
                        sys.exit(1)
    n_done += 1
print(f"stress_plan: {n_done} random launches x 16 planner forms (+ 8 with pair records) each: all equal to the definition ({time.time() - t0:.0f} s)")
