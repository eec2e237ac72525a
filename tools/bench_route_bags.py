#!/usr/bin/env python3
"""Dev: time the pooled channel's routing alone (C4's history: B x 50 ids, weights all 1) -- nrx_route_bags_one against nrx_route_bags."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B, L, rows = 65536, 50, 200_000
gen = torch.Generator(device=dev).manual_seed(1)
pool = [torch.randint(1, rows, (B, L), device=dev, generator=gen) for _ in range(4)]
w = torch.ones((B, L), dtype=torch.float32, device=dev)
cap = B * L if W == 1 else (int(B * L / W * 1.05) + 256 + 63) // 64 * 64
rows_o = torch.empty(W * cap, dtype=torch.int32, device=dev); tag_o = torch.empty_like(rows_o); w_o = torch.empty(W * cap, dtype=torch.float32, device=dev)
c2d = torch.empty((W, 1), dtype=torch.int64, device=dev); over = torch.zeros(1, dtype=torch.int64, device=dev)
bl = (C.c_int32 * 1)(L)
state = torch.zeros(lib.nrx_route_bags_one_state_bytes(bl, 1, B, W), dtype=torch.uint8, device=dev)
ws = torch.empty(max(1, lib.nrx_route_workspace(B * L, W)), dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
wp = (C.c_void_p * 1)(w.data_ptr())
def timed(fn, reps=100):
    for i in range(10): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
ptrs = [(C.c_void_p * 1)(x.data_ptr()) for x in pool]
one = timed(lambda i: lib.nrx_route_bags_one(ptrs[i % 4], wp, bl, 1, 64, B, W, cap, rows_o.data_ptr(), tag_o.data_ptr(), w_o.data_ptr(), c2d.data_ptr(), over.data_ptr(), state.data_ptr(), st))
three = timed(lambda i: lib.nrx_route_bags(ptrs[i % 4], wp, bl, 1, 64, B, W, cap, rows_o.data_ptr(), tag_o.data_ptr(), w_o.data_ptr(), c2d.data_ptr(), over.data_ptr(), ws.data_ptr(), st))
nb = lib.nrx_route_bags_runs_state_bytes(bl, 1, B, W)
runs = float("nan")
if nb > 0:
    s2 = torch.zeros(nb, dtype=torch.uint8, device=dev)
    run = torch.empty((W, B, 2), dtype=torch.int32, device=dev)
    inv = torch.empty(B, dtype=torch.float32, device=dev)
    kinds = (C.c_int32 * 1)(2)       # NRX_BAG_MASKED_MEAN
    ip = (C.c_void_p * 1)(inv.data_ptr())
    wn = torch.empty_like(w)
    runs = timed(lambda i: lib.nrx_route_bags_runs(ptrs[i % 4], wp, kinds, bl, 1, 64, B, W, cap, rows_o.data_ptr(), w_o.data_ptr(), run.data_ptr(), ip, c2d.data_ptr(), over.data_ptr(), s2.data_ptr(), st))
    both = timed(lambda i: (lib.nrx_bag_norm_weights_inv(w.data_ptr(), B, L, 2, wn.data_ptr(), inv.data_ptr(), st),
                            lib.nrx_route_bags_one(ptrs[i % 4], (C.c_void_p * 1)(wn.data_ptr()), bl, 1, 64, B, W, cap, rows_o.data_ptr(), tag_o.data_ptr(), w_o.data_ptr(), c2d.data_ptr(), over.data_ptr(), state.data_ptr(), st)))
    print(f"world {W}: nrx_route_bags_runs {runs:.1f} us  against nrx_bag_norm_weights_inv + nrx_route_bags_one {both:.1f} us")
print(f"world {W}: nrx_route_bags_one {one:.1f} us, nrx_route_bags {three:.1f} us  (lib {os.path.basename(os.environ.get('NRX_LIB', 'default'))})")
