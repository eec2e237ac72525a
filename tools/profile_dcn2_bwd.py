#!/usr/bin/env python3
"""Dev: DCN-v2 layer forward + backward a few times (for rocprofv3 --kernel-trace --stats).  usage: profile_dcn2_bwd.py [D] [iterations]  (>= 200 iterations for settled clocks)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
D = int(sys.argv[1]) if len(sys.argv) > 1 else 320
B = 65536
dev = torch.device("cuda:0")
x = torch.randn(B, D, device=dev, requires_grad=True)
W = (torch.randn(1, D, D, device=dev) / D ** 0.5).requires_grad_(True)
b = torch.zeros(1, D, device=dev, requires_grad=True)
up = torch.randn(B, D, device=dev)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    out = ops.dcn_v2(x, W, b)
    torch.autograd.grad(out, (x, W, b), up)
torch.cuda.synchronize()
