#!/usr/bin/env python3
"""Dev: where the one-kernel planner's time goes.  Needs a library built with -DNRX_PL_TIMING (tools/build_variant.sh pltime -DNRX_PL_TIMING; run with
NRX_LIB=news_recsys_amd/lib/variants/libnrx_pltime.so): every block stamps the 100 MHz wall clock at its phase boundaries.
usage: plan_lds_phases.py [tables=26] [rows=1000000] [batch=65536]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import _lib
T = int(sys.argv[1]) if len(sys.argv) > 1 else 26
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
ids = [torch.randint(1, rows, (B,), device=dev, generator=g) for _ in range(T)]
n = T * B
order, uniq, seg = (torch.empty(n + 1, dtype=torch.int64, device=dev) for _ in range(3))
counts = torch.empty(T + 2, dtype=torch.int64, device=dev)
dest, walk = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
n_walk = torch.empty(2, dtype=torch.int64, device=dev)
pairs = torch.empty((n // 2 + 1, 4), dtype=torch.int32, device=dev)
stats = torch.zeros(4, dtype=torch.int64, device=dev)
state = torch.zeros(lib.nrx_sparse_plan_lds_state_bytes(), dtype=torch.uint8, device=dev)
ws = torch.empty(lib.nrx_sparse_plan_lds_workspace(n), dtype=torch.uint8, device=dev)
ptrs = (C.c_void_p * T)(*[x.data_ptr() for x in ids]); lens = (C.c_int64 * T)(*([B] * T))
tof = (C.c_int32 * T)(*range(T)); rws = (C.c_int64 * T)(*([rows] * T))
st = torch.cuda.current_stream().cuda_stream
def run():
    _lib.check(lib.nrx_sparse_plan_lds(ptrs, lens, tof, rws, T, 64, T, order.data_ptr(), uniq.data_ptr(), seg.data_ptr(), counts.data_ptr(), dest.data_ptr(),
                                       walk.data_ptr(), n_walk.data_ptr(), pairs.data_ptr(), n_walk.data_ptr() + 8, stats.data_ptr(), state.data_ptr(), ws.data_ptr(), st), "plan")
for _ in range(5): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): run()
b.record(); torch.cuda.synchronize()
print(f"{T} tables x {rows} rows, batch {B}: {a.elapsed_time(b) / 20 * 1e3:.1f} us per plan (20 back to back); stats {stats.tolist()}")
nb = T * ((rows + (1 << 17) - 1) >> 17)
fn = getattr(C.CDLL(_lib.LIB_PATH), "nrx_plan_lds_stamps", None)
if fn is None:
    sys.exit("(library without -DNRX_PL_TIMING: no phase stamps)")
buf = (C.c_ulonglong * (nb * 8))()
assert fn(buf, nb) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.float64) / 100.0
t -= t[:, 0].min()
names = ["start", "scan", "mark", "rank", "chain", "place", "list", "keys"]
print("phase END, us after the first block's start: mean / max over the blocks")
prev = t[:, 0]
for k in range(8):
    print(f"  {names[k]:6s} {t[:, k].mean():7.2f} / {t[:, k].max():7.2f}    (phase itself: mean {(t[:, k] - prev).mean():6.2f})")
    prev = t[:, k]
