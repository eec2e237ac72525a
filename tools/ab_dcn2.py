#!/usr/bin/env python3
"""A/B of the DCN-v2 forward kernel forms (NRX_DCN2_FORM = classic | persist1 | persist2, read once per process):
runs one layer at B = 65536 for each width, prints a digest of the outputs and the mean launch time.  Drive it once per
form and compare the digests (they must be identical: same fma chain per element)."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import _lib
from news_recsys_amd.ops import check

lib = _lib.load()
dev = torch.device("cuda:0")
form = os.environ.get("NRX_DCN2_FORM", "default")
B = int(os.environ.get("AB_BATCH", "65536"))
for D in [int(a) for a in sys.argv[1:]] or [320, 112, 256, 512, 64, 128, 96]:
    g = torch.Generator(device="cpu").manual_seed(D)
    if os.environ.get("AB_PROBE_DATA"):      # the value ranges of tools/dcn2_phase_probe.hip: everything uniform in [-0.1, 0.1)
        x = ((torch.rand(B, D, generator=g) - 0.5) * 0.2).to(dev); x0 = ((torch.rand(B, D, generator=g) - 0.5) * 0.2).to(dev)
        W = ((torch.rand(D, D, generator=g) - 0.5) * 0.2).to(dev); b = ((torch.rand(D, generator=g) - 0.5) * 0.2).to(dev)
    else:
        x = (torch.rand(B, D, generator=g) - 0.5).to(dev); x0 = (torch.rand(B, D, generator=g) - 0.5).to(dev)
        W = ((torch.rand(D, D, generator=g) - 0.5) / D ** 0.5).to(dev); b = (torch.rand(D, generator=g) - 0.5).to(dev)
    for name, xa, train in (("x0!=xl inference", x0, False), ("x0==xl inference", x, False), ("x0!=xl training", x0, True)):
        out = torch.empty_like(x); lin = torch.empty_like(x) if train else None
        st = torch.cuda.current_stream().cuda_stream
        def run():
            check(lib.nrx_dcn_v2_layer_fwd(xa.data_ptr(), x.data_ptr(), D, B, D, W.data_ptr(), b.data_ptr(), 1, out.data_ptr(), D,
                                           lin.data_ptr() if train else None, st), "fwd")
        for _ in range(int(os.environ.get('AB_WARMUP', '300'))): run()      # ~45 ms of the same launch: the clocks have settled
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 100 * 1e3
        h = hashlib.sha1(out.cpu().numpy().tobytes()); 
        if train: h.update(lin.cpu().numpy().tobytes())
        tf = (2.0 * D * D + 3.0 * D) * B / us * 1e-6
        print(f"{form:9s} D={D:4d} {name:18s} {us:8.1f} us {tf:6.1f} TF {tf / 157.3 * 100:5.1f} %  sha1 {h.hexdigest()[:12]}", flush=True)
