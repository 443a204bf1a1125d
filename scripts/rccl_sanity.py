#!/usr/bin/env python3
"""DEV-ONLY: exercise the RCCL code path of distributed.py with however many ranks torchrun gives
(1 on the single-GPU box): init with device_id, float64 SUM / MAX all-reduce, barrier, async KE reduce."""
import os, sys
import torch, torch.distributed as dist
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from silver2_isaacsim_amd import distributed as hd
os.environ.setdefault("RANK", "0"); os.environ.setdefault("LOCAL_RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
rank, local_rank, world = hd.env_rank_world()
torch.cuda.set_device(local_rank)
dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
dev = torch.device("cuda", local_rank)
print("backend", dist.get_backend(), "world", dist.get_world_size(), "collective device", hd.collective_device(dev))
ke = torch.tensor([1.5 + rank, 0.25], dtype=torch.float64, device=dev)
out, work = hd.global_kinetic_energy(ke, async_op=True)
if work is not None:
    work.wait()
else:                                  # world == 1: helper is a no-op by design; call RCCL directly once
    dist.all_reduce(ke, op=dist.ReduceOp.SUM)
t = torch.tensor([3.0 + rank], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("ke", ke.tolist(), "max", t.item())
dist.destroy_process_group()
print("rccl ok")
