#!/usr/bin/env python3
"""Profile driver: 5 launches of nrx_topk_ip at 65 536 queries x 200 000 items x 16, k = 10."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import ops
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(9)
items = torch.nn.functional.normalize(torch.randn(200_000, 16, device=dev, generator=gen), dim=1)
q = torch.nn.functional.normalize(torch.randn(65_536, 16, device=dev, generator=gen), dim=1)
with torch.no_grad():
    for _ in range(5):
        ops.topk_ip(items, q, 10)
torch.cuda.synchronize()
