#!/usr/bin/env python3
"""Dev: non-temporal stores for the forward's concat (NRX_FWD_STNT) and the placement pass's rows (NRX_PLACE_STNT), alternated in ONE process per
workload (the library reads the variables per call).  usage: ab_stnt.py [c2|c4|c5] ..."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
def t(fn, n=200):
    for i in range(20): fn(i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for wl in sys.argv[1:] or ["c2"]:
    path = bench.SingleGpuPath(wl, dev, 1)
    calls = path.plain_calls or path.calls
    if os.environ.get("WIDE") == "1" and path.plain_calls is not None:
        calls = path.calls                      # c5: the headline launch (the gather with the Wide&Deep column routing)
    dcalls = path.distinct_output_calls()
    fwd, bwd = (path.train_pass() if not path.cross else (None, None))
    for rep in range(3):
        out = []
        for k in ("0", "1"):
            os.environ["NRX_FWD_STNT"] = k
            out.append(t(lambda i: calls[i % len(calls)].run()))
            out.append(t(lambda i: dcalls[i % len(dcalls)].run()))
        os.environ["NRX_FWD_STNT"] = "0"
        line = f"{wl} rep {rep}: forward recycled buffer plain {out[0]:.1f} / nt {out[2]:.1f} us | distinct buffers plain {out[1]:.1f} / nt {out[3]:.1f} us"
        if fwd is not None:
            fb = []
            for k in ("0", "1"):
                os.environ["NRX_PLACE_STNT"] = k
                def both(i):
                    fwd[i % 2].run(); bwd[i % 2].run()
                fb.append(t(both, 100))
            os.environ["NRX_PLACE_STNT"] = "0"
            line += f" | fwd+bwd place stores plain {fb[0]:.1f} / nt {fb[1]:.1f} us"
        print(line, flush=True)
    del path, calls, dcalls, fwd, bwd
    torch.cuda.empty_cache()
