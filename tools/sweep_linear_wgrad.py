#!/usr/bin/env python3
"""Dev: nrx_linear_wgrad through the C-ABI on the MLP-head shapes (B = 65 536) next to torch's g.T @ a, for one setting of the tuning
switches (NRX_WGRAD_TILE / NRX_WGRAD_BLOCKS / NRX_WGRAD_MIN_ROWS are read once per process).  usage: sweep_linear_wgrad.py [--torch]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import _lib
lib = _lib.load()
B = 65536
shapes = [(128, 416), (128, 128), (64, 128), (1, 64), (256, 512), (512, 512), (112, 112), (320, 320)]
st = torch.cuda.current_stream().cuda_stream
tag = " ".join(f"{k}={os.environ[k]}" for k in ("NRX_WGRAD_TILE", "NRX_WGRAD_BLOCKS", "NRX_WGRAD_MIN_ROWS") if k in os.environ) or "default"
res = []
for o, i in shapes:
    g = torch.randn(B, o, device="cuda"); a = torch.randn(B, i, device="cuda"); gW = torch.empty(o, i, device="cuda"); gb = torch.empty(o, device="cuda")
    def run():
        rc = lib.nrx_linear_wgrad(g.data_ptr(), o, a.data_ptr(), i, B, o, i, gW.data_ptr(), gb.data_ptr(), st)
        assert rc == 0
    def ref():
        return g.t() @ a
    f = ref if "--torch" in sys.argv else run
    t_end = time.perf_counter() + 0.06
    while time.perf_counter() < t_end: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) * 10)
print(f"{'torch g.T @ a' if '--torch' in sys.argv else tag:60s}" + " ".join(f"{o}x{i}:{r:6.1f}" for (o, i), r in zip(shapes, res)), flush=True)
