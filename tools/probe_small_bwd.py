#!/usr/bin/env python3
"""Dev: standalone FM / bag-pool forward and backward through the C-ABI (no autograd around them), B = 65536 -- HIP-event means."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recsys_amd import _lib
from news_recsys_amd.ops import check
lib = _lib.load(); dev = "cuda:0"; B = 65536
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=200):
    for _ in range(50): fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) * 1e3 / n
F, D = 26, 16
feat = torch.randn(B, F * D, device=dev); out = torch.empty(B, device=dev); g = torch.randn(B, device=dev); gf = torch.empty_like(feat)
us = timeit(lambda: check(lib.nrx_fm_fwd(feat.data_ptr(), F * D, F, D, B, out.data_ptr(), st), "fm_fwd"))
print(f"fm_fwd  [B,{F*D}]: {us:7.1f} us  {B*F*D*4/us/1e3:7.0f} GB/s (read feat)")
us = timeit(lambda: check(lib.nrx_fm_bwd(feat.data_ptr(), F * D, F, D, B, g.data_ptr(), None, 0, gf.data_ptr(), F * D, st), "fm_bwd"))
print(f"fm_bwd  [B,{F*D}]: {us:7.1f} us  {2*B*F*D*4/us/1e3:7.0f} GB/s (read feat + write g_feat)")
L, D = 50, 16
emb = torch.randn(B, L, D, device=dev); mask = (torch.rand(B, L, device=dev) < 0.7).float(); po = torch.empty(B, D, device=dev)
gp = torch.randn(B, D, device=dev); ge = torch.empty_like(emb)
us = timeit(lambda: check(lib.nrx_bag_pool_fwd(emb.data_ptr(), mask.data_ptr(), B, L, D, po.data_ptr(), st), "pool_fwd"))
print(f"bag_pool_fwd [B,{L},{D}]: {us:7.1f} us  {B*L*(D+1)*4/us/1e3:7.0f} GB/s (read emb + mask)")
us = timeit(lambda: check(lib.nrx_bag_pool_bwd(gp.data_ptr(), mask.data_ptr(), B, L, D, ge.data_ptr(), st), "pool_bwd"))
print(f"bag_pool_bwd [B,{L},{D}]: {us:7.1f} us  {B*L*(D+1)*4/us/1e3:7.0f} GB/s (read mask + write g_emb)")
