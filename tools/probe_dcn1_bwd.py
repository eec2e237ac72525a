#!/usr/bin/env python3
"""Dev: nrx_dcn_v1_bwd through the C-ABI (no autograd around it) by batch size, width and depth -- HIP-event mean of 200 launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import _lib
from news_recsys_amd.ops import check
lib = _lib.load(); dev = "cuda:0"
sep = os.environ.get("SEP") == "1"       # SEP=1: separate layer-0 input x0 (DCNLayer.forward(x_l, x_0)), its gradient to g_x0
for D, NL in (((320, 1), (112, 1), (320, 5)) if sep else ((112, 3), (112, 1), (320, 2), (320, 3), (320, 5), (640, 2))):
    for B in ([int(a) for a in sys.argv[1:]] or [4096, 16384, 65536, 262144]):
        x = torch.randn(B, D, device=dev); g = torch.randn(B, D, device=dev); gx = torch.empty_like(x)
        w = torch.randn(NL, D, device=dev) / D ** 0.5; b = torch.zeros(NL, D, device=dev)
        gw = torch.zeros_like(w); gb = torch.zeros_like(b)
        x0 = torch.randn(B, D, device=dev) if sep else None; gx0 = torch.empty_like(x) if sep else None
        st = torch.cuda.current_stream().cuda_stream
        def run():
            check(lib.nrx_dcn_v1_bwd(x.data_ptr(), D, x0.data_ptr() if sep else None, D if sep else 0, B, D, NL, w.data_ptr(), b.data_ptr(),
                                     g.data_ptr(), D, gx.data_ptr(), D, gx0.data_ptr() if sep else None, D if sep else 0,
                                     gw.data_ptr(), gb.data_ptr(), st), "bwd")
        for _ in range(50): run()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200): run()
        e.record(); torch.cuda.synchronize()
        us = a.elapsed_time(e) * 5
        print(f"D={D} L={NL} B={B:7d}: {us:7.1f} us   {3 * B * D * 4 / us / 1e3:7.0f} GB/s (x, g read + g_x written)", flush=True)
