#!/usr/bin/env python3
"""Dev: exercises the exact torch.distributed / RCCL calls bench.py and sharding.py make at N > 1, with the one GPU a
gpurun box has (world_size = 1 through a real nccl process group): init with device_id, barrier, all_reduce(MAX) of a
float64 scalar, all_to_all_single with equal splits on int64 and fp32 buffers.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/rccl_smoke.py"""
import os
import torch
import torch.distributed as dist

local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
device = torch.device("cuda", local_rank)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("nccl", device_id=device)
dist.barrier()
t = torch.tensor([1.25], dtype=torch.float64, device=device)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
a = torch.arange(1024, dtype=torch.int64, device=device)
b = torch.empty_like(a)
dist.all_to_all_single(b, a)
x = torch.randn(4096, 16, device=device)
y = torch.empty_like(x)
dist.all_to_all_single(y.view(-1), x.view(-1))
torch.cuda.synchronize()
assert t.item() == 1.25 and torch.equal(a, b) and torch.equal(x, y)
print(f"rccl smoke ok: world={dist.get_world_size()} backend={dist.get_backend()}")
dist.destroy_process_group()
