#!/usr/bin/env python3
"""Dev (GPU box): random DCN-v2 layers (batch 1 .. 70 000, dim 4 .. 340 -- the panel form up to 112 wide, the three-launch form above, aligned and
unaligned widths, padded leading dimensions, every accumulate_x0 mode, ReLU on / off, ordered and atomic weight gradients) through nrx_dcn_v2_layer_bwd against
the fp64 definition (dcn_arch.py:33-50, 73-91); ordered runs are repeated and compared word for word.
usage: python tests/stress_dcn2_bwd.py [seconds=120] [seed=1]   (a checker like the tests next to it; not collected by pytest)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _poison
from news_recsys_amd import _lib
lib = _lib.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
DEV = "cuda:0"
t0, n_done, n_panel, n_big = time.time(), 0, 0, 0
st = torch.cuda.current_stream().cuda_stream
while time.time() - t0 < budget:
    _poison.poison()
    B = int(rng.choice([1, 2, 31, 63, 64, 65, 127, 129, 1000, 4097, 20011, 70000]))
    D = int(rng.choice([4, 8, 12, 16, 36, 37, 64, 96, 100, 108, 112, 116, 128, 129, 200, 320, 340]))
    if B * D > 12_000_000:
        continue
    pad = int(rng.choice([0, 4, 7]))
    relu, acc, ordered = int(rng.integers(0, 2)), int(rng.choice([0, 1, 3])), int(rng.integers(0, 2))
    gen = torch.Generator(device=DEV).manual_seed(int(rng.integers(0, 1 << 30)))
    rnd = lambda *shape: torch.randn(*shape, device=DEV, generator=gen)
    ld = D + pad
    wide = lambda t, extra: torch.cat([t, torch.full((t.shape[0], extra), float("nan"), device=DEV)], 1).contiguous() if extra else t.contiguous()
    x0, xl, g = rnd(B, D), rnd(B, D), rnd(B, D)
    W, b = rnd(D, D) / D ** 0.5 * (1.0 + torch.triu(torch.ones(D, D, device=DEV))), rnd(D) * 0.1
    gx0_init = rnd(B, D)
    X0, XL = wide(x0, pad), wide(xl, pad)
    OUT, LIN = wide(torch.zeros(B, D, device=DEV), pad), wide(torch.zeros(B, D, device=DEV), pad)
    assert lib.nrx_dcn_v2_layer_fwd(X0.data_ptr(), XL.data_ptr(), ld, B, D, W.data_ptr(), b.data_ptr(), relu, OUT.data_ptr(), ld, LIN.data_ptr(), st) == 0
    runs = []
    for _ in range(2 if ordered else 1):
        G, GXL, GX0 = wide(g, 2 * pad), wide(torch.zeros(B, D, device=DEV), 3 * pad), wide(gx0_init.clone(), pad)
        gW, gb = torch.empty(D, D, device=DEV), torch.empty(D, device=DEV)
        ws = torch.empty(lib.nrx_dcn_v2_layer_bwd_workspace(B, D), dtype=torch.uint8, device=DEV)
        rc = lib.nrx_dcn_v2_layer_bwd(X0.data_ptr(), XL.data_ptr(), ld, LIN.data_ptr(), OUT.data_ptr(), relu | (4 if ordered else 0), B, D, W.data_ptr(),
                                      G.data_ptr(), D + 2 * pad, GXL.data_ptr(), D + 3 * pad, GX0.data_ptr(), D + pad, acc, gW.data_ptr(), gb.data_ptr(),
                                      ws.data_ptr(), st)
        assert rc == 0, lib.nrx_last_error()
        torch.cuda.synchronize()
        if pad:
            assert torch.isnan(GXL[:, D:]).all() and torch.isnan(GX0[:, D:]).all()
        runs.append((GXL[:, :D].clone(), GX0[:, :D].clone(), gW, gb))
    if ordered:
        for a_, b_ in zip(runs[0], runs[1]):
            assert torch.equal(a_.view(torch.int32), b_.view(torch.int32)), dict(B=B, D=D, pad=pad, relu=relu, acc=acc)
    m = (OUT[:, :D] > 0).double() if relu else torch.ones(B, D, device=DEV, dtype=torch.float64)
    gm = g.double() * m
    lin64 = xl.double() @ W.double().t() + b.double()
    want_gx0 = gm * lin64 + (gx0_init.double() if acc & 1 else 0)
    glin = gm * x0.double()
    want_gxl = gm + glin @ W.double() + (want_gx0 if acc & 2 else 0)
    want_gW, want_gb = glin.t() @ xl.double(), glin.sum(0)
    for got, want, tol in ((runs[0][1], want_gx0, 1e-4), (runs[0][0], want_gxl, 1e-4), (runs[0][2], want_gW, 3e-6 * max(1.0, B ** 0.5)),
                           (runs[0][3], want_gb, 3e-6 * max(1.0, B ** 0.5))):
        err = (got.double() - want).abs().max().item()
        assert err <= tol * max(1.0, want.abs().max().item()), dict(B=B, D=D, pad=pad, relu=relu, acc=acc, ordered=ordered, err=err)
    n_done += 1
    n_panel += D <= 112 and D % 4 == 0 and pad % 4 == 0
    n_big += B >= 20011
print(f"stress_dcn2_bwd: {n_done} random layers ({n_panel} in the panel form, {n_big} with 20 011 or 70 000 rows): all within tolerance of the fp64 definition, ordered runs word for word equal ({time.time() - t0:.0f} s)")
