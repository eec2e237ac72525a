#!/usr/bin/env python3
"""Dev: the library's streaming copy (nrx_stream_copy) under its variant knob -- which form does this box copy fastest with?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import _lib
lib = _lib.load()
n = 1 << 30
src = torch.empty(n // 4, dtype=torch.float32, device="cuda").normal_()
dst = torch.empty_like(src)
st = torch.cuda.current_stream().cuda_stream
for var in sys.argv[1:] or ["018", "0132", "048", "0416", "0432", "118", "1132", "148", "1416", "1432", "044", "144"]:
    os.environ["NRX_COPY_VARIANT"] = var
    for _ in range(3):
        lib.nrx_stream_copy(dst.data_ptr(), src.data_ptr(), n, st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        lib.nrx_stream_copy(dst.data_ptr(), src.data_ptr(), n, st)
    b.record()
    torch.cuda.synchronize()
    print(f"variant nt={var[0]} unroll={var[1]} blocks/CU={var[2:]}: {2 * n * 20 / (a.elapsed_time(b) * 1e-3) / 1e9:8.1f} GB/s (read + written)")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    dst.copy_(src)
b.record()
torch.cuda.synchronize()
print(f"torch copy_: {2 * n * 20 / (a.elapsed_time(b) * 1e-3) / 1e9:8.1f} GB/s")

# the runtime's own device-to-device copy (hipMemcpyDtoDAsync from the HIP runtime this process already has loaded): is the "box ceiling" of
# bench.py's stream_copy_GBps an artefact of nrx_copy_kernel?  (round-3 review, hygiene item)
import ctypes
hip = None
for line in open("/proc/self/maps"):
    if "libamdhip64" in line:
        hip = ctypes.CDLL(line.split()[-1])
        break
if hip is not None:
    hip.hipMemcpyDtoDAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemcpyDtoDAsync.restype = ctypes.c_int
    for _ in range(3):
        assert hip.hipMemcpyDtoDAsync(dst.data_ptr(), src.data_ptr(), n, st) == 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        hip.hipMemcpyDtoDAsync(dst.data_ptr(), src.data_ptr(), n, st)
    b.record()
    torch.cuda.synchronize()
    print(f"hipMemcpyDtoDAsync: {2 * n * 20 / (a.elapsed_time(b) * 1e-3) / 1e9:8.1f} GB/s (read + written)")
