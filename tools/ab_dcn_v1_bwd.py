#!/usr/bin/env python3
"""Dev: the DCN-v1 stack's forward + backward (ops.dcn_v1 through autograd), cross-weight gradients by float atomics against the ORDERED mode."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
B = 65536
for D, NL in ((112, 3), (320, 2), (640, 3), (64, 4), (128, 6)):
    x = torch.randn(B, D, device="cuda", requires_grad=True)
    w = (torch.randn(NL, D, device="cuda") / D ** 0.5).requires_grad_()
    b = (torch.randn(NL, D, device="cuda") * 0.1).requires_grad_()
    up = torch.randn(B, D, device="cuda")
    res = {}
    for rnd in range(2):
        for name, o in (("atomic", False), ("ordered", True)):
            ops.WGRAD_ORDERED, ops.WGRAD_ATOMIC = o, not o
            def step():
                torch.autograd.grad(ops.dcn_v1(x, w, b), [x, w, b], up)
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.graph(g, stream=s):
                step()
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                g.replay()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 50 * 1e3)
            del g
    print(f"dcn_v1 D = {D}, {NL} layers, B = {B}: forward + backward graph replay   atomic {min(res['atomic']):6.1f} us   ordered {min(res['ordered']):6.1f} us")
