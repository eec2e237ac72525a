#!/usr/bin/env python3
"""Dev: ops.linear's weight gradient (the batch is the contraction), float atomics against the ORDERED mode, at MLP shapes, B = 65 536.
GPU time of the wgrad launches alone (torch.cuda.Event around 50 calls of the library function)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import _lib
from news_recsys_amd.ops import _stream_ptr
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for out_f, in_f in ((128, 416), (128, 128), (64, 128), (32, 64), (1, 64), (256, 256), (512, 416)):
    g = torch.randn(B, out_f, device="cuda")
    a = torch.randn(B, in_f, device="cuda")
    gW, gb = torch.empty(out_f, in_f, device="cuda"), torch.empty(out_f, device="cuda")
    ws = torch.empty(lib.nrx_linear_wgrad_ordered_workspace(B, out_f, in_f), dtype=torch.uint8, device="cuda")
    st = _stream_ptr(g)
    res = {}
    for name, fn in (("atomic", lambda: lib.nrx_linear_wgrad(g.data_ptr(), out_f, a.data_ptr(), in_f, B, out_f, in_f, gW.data_ptr(), gb.data_ptr(), st)),
                     ("ordered", lambda: lib.nrx_linear_wgrad_ordered(g.data_ptr(), out_f, a.data_ptr(), in_f, B, out_f, in_f, gW.data_ptr(), gb.data_ptr(), ws.data_ptr(), st))):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 50 * 1e3
    print(f"linear wgrad B = {B}, W [{out_f}, {in_f}]: atomic {res['atomic']:6.1f} us (fills + GEMM)   ordered {res['ordered']:6.1f} us (GEMM + reduction)   scratch {ws.numel() / 1e6:.1f} MB")
