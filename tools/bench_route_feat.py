#!/usr/bin/env python3
"""Dev: time the routing launch(es) alone at a given world size on one GPU (the kernels do not care that the peers are not there):
nrx_route_feat (one launch, per-feature blocks) against nrx_route_ids (per-owner blocks: hist + scan + place).
usage: bench_route_feat.py [c2|c5|c3] [world ...]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from news_recsys_amd import _lib, ops
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
worlds = [int(x) for x in sys.argv[2:]] or [1, 2, 8]
lib = _lib.load()
dev = torch.device("cuda:0")
feats, _ = bench.workload_spec(wl)
feats = [f for f in feats if not f["bag"]]
B, n = bench.BATCH, len(feats)
gen = torch.Generator(device=dev).manual_seed(1)
pool = [[torch.randint(1, f["rows"], (B,), device=dev, generator=gen) for f in feats] for _ in range(4)]
st = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=200):
    for i in range(20): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for W in worlds:
    capf = B if W == 1 else (int(B / W * 1.05) + 64 + 63) // 64 * 64
    send = torch.empty((W, n, capf), dtype=torch.int32, device=dev); pos = torch.empty_like(send)
    slot = torch.empty((n, B), dtype=torch.int32, device=dev); counts = torch.empty((W, n), dtype=torch.int64, device=dev)
    over = torch.zeros(1, dtype=torch.int64, device=dev)
    state = torch.zeros(lib.nrx_route_feat_state_bytes(n, B, W), dtype=torch.uint8, device=dev)
    ptrs = [(C.c_void_p * n)(*[x.data_ptr() for x in ids]) for ids in pool]
    for with_pos in (False, True):
        us = timed(lambda i: lib.nrx_route_feat(ptrs[i % 4], n, B, 64, W, capf, send.data_ptr(), pos.data_ptr() if with_pos else None, slot.data_ptr(),
                                                counts.data_ptr(), over.data_ptr(), state.data_ptr(), st))
        print(f"{wl} world {W}: nrx_route_feat{' + positions' if with_pos else ''} {us:.1f} us  (lib {os.environ.get('NRX_LIB', 'default')})")
    cap = max(64, n * B) if W == 1 else (int(n * B / W * 1.05) + 256 + 63) // 64 * 64
    us = timed(lambda i: ops.route_ids(pool[i % 4], W, cap), 50)
    print(f"{wl} world {W}: ops.route_ids (allocating wrapper around nrx_route_ids) {us:.1f} us")
