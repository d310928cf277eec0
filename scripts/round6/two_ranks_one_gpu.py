#!/usr/bin/env python3
"""Can RCCL run TWO ranks on ONE GPU here?  (NCCL refuses duplicate devices; RCCL has had builds that allow it.)
torchrun --nproc-per-node 2 scripts/round6/two_ranks_one_gpu.py"""
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
x = torch.full((1024,), float(rank + 1), device=dev)
dist.all_reduce(x)
torch.cuda.synchronize()
print(f"rank {rank}: all_reduce ok, value {float(x[0])}", flush=True)
if rank == 0:
    y = torch.empty(1024, device=dev)
    dist.recv(y, src=1)
else:
    dist.send(x, dst=0)
torch.cuda.synchronize()
print(f"rank {rank}: send/recv ok", flush=True)
dist.destroy_process_group()
