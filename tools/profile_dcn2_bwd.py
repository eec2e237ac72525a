#!/usr/bin/env python3
"""Dev: DCN-v2 stack forward + backward through autograd for rocprofv3 / timing.  usage: profile_dcn2_bwd.py [D] [iters] [fp32|bf16x3] [layers]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
D = int(sys.argv[1]) if len(sys.argv) > 1 else 320
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
math = sys.argv[3] if len(sys.argv) > 3 else "fp32"
NL = int(sys.argv[4]) if len(sys.argv) > 4 else 1
B = 65536
x = torch.randn(B, D, device="cuda", requires_grad=True)
W = [(torch.randn(D, D, device="cuda") / D ** 0.5).requires_grad_() for _ in range(NL)]
b = [(torch.randn(D, device="cuda") * 0.1).requires_grad_() for _ in range(NL)]
up = torch.randn(B, D, device="cuda")
def step():
    out = ops.dcn_v2(x, W, b, math=math)
    torch.autograd.grad(out, [x] + W + b, up)
for _ in range(150):
    step()
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n):
    step()
e.record()
torch.cuda.synchronize()
print(f"dcn_v2 {math} D={D} layers={NL} B={B}: forward + backward {a.elapsed_time(e) / n * 1e3:.1f} us per step (through autograd)")
