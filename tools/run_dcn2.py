#!/usr/bin/env python3
"""Dev: back-to-back DCN-v2 layer forwards (inference form) for rocprofv3 / timing.  usage: run_dcn2.py [D] [launches] [fp32|bf16x3] [train]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import _lib
D = int(sys.argv[1]) if len(sys.argv) > 1 else 320
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
math = sys.argv[3] if len(sys.argv) > 3 else "fp32"
train = len(sys.argv) > 4
lib = _lib.load()
B = 65536
x = torch.randn(B, D, device="cuda")
x1 = torch.randn(B, D, device="cuda")
W = torch.randn(D, D, device="cuda") / D ** 0.5
b = torch.randn(D, device="cuda") * 0.1
out = torch.empty_like(x)
lin = torch.empty_like(x) if train else None
st = torch.cuda.current_stream().cuda_stream
flags = 1 | (2 if math == "bf16x3" else 0)
for same in (True, False):
    x0 = x if same else x1
    for _ in range(300):
        lib.nrx_dcn_v2_layer_fwd(x0.data_ptr(), x.data_ptr(), D, B, D, W.data_ptr(), b.data_ptr(), flags, out.data_ptr(), D,
                                 None if lin is None else lin.data_ptr(), st)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        lib.nrx_dcn_v2_layer_fwd(x0.data_ptr(), x.data_ptr(), D, B, D, W.data_ptr(), b.data_ptr(), flags, out.data_ptr(), D,
                                 None if lin is None else lin.data_ptr(), st)
    e.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(e) / n * 1e3
    fl = 2.0 * B * D * D + 3.0 * B * D
    print(f"dcn_v2 layer fwd {math} D={D} B={B} {'x0 == x_l' if same else 'x0 != x_l'}{' +lin_out' if train else ''}: {us:.1f} us  "
          f"{fl / us / 1e6:.1f} TFLOP/s (2D^2+3D flop/row)")
