#!/usr/bin/env python3
"""Runs the DCN-v2 MFMA layer at C3 shape (B=65536, D=320) for profiling.  usage: run_dcn2.py [D] [launches]
(20 launches = the clock-ramp regime the round-1 counters were taken in; >= 400 for a settled kernel average)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
D = int(sys.argv[1]) if len(sys.argv) > 1 else 320
dev = torch.device("cuda:0")
x = torch.randn(65536, D, device=dev); W = torch.randn(1, D, D, device=dev) / D ** 0.5; b = torch.randn(1, D, device=dev)
with torch.no_grad():
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
        ops.dcn_v2(x, W, b)
torch.cuda.synchronize()
