"""Multi-GPU layer: block partition of bodies + the one collective on the path.

Bodies are independent (no body-to-body term anywhere in the wrench model), so
the N>1 path is a contiguous block partition with no per-step exchange
(SURVEY.md section 8e).  The only collective is the optional global kinetic
energy: each rank reduces its shard on device to one float64 and the ranks
all-reduce that scalar (RCCL over xGMI when the backend is "nccl"; gloo on CPU
in tests).  Payload 8-16 bytes, latency-bound, so it runs every K steps and
asynchronously, never inside the per-step path.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of n bodies owned by `rank`; the first n % world
    ranks get one extra body.  Concatenating all shards in rank order gives 0..n."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(int(n), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def env_rank_world() -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment, (0,0,1) if absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: str | None = None, device_id: int | None = None, force: bool = False) -> bool:
    """Initialise torch.distributed from the env when WORLD_SIZE>1 (or, with force=True, a one-rank group: the way a
    single-GPU box drives the real RCCL calls).  Returns True if a group is active.  backend None -> 'nccl' (= RCCL)
    on GPU, 'gloo' otherwise."""
    rank, local_rank, world = env_rank_world()
    if world <= 1 and not force:
        return False
    if dist.is_initialized():
        return True
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if backend is None:
        backend = os.environ.get("HYDRO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {}
    if backend == "nccl":
        dev = local_rank if device_id is None else device_id
        torch.cuda.set_device(dev)
        kw["device_id"] = torch.device("cuda", dev)
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return True


def collective_device(default: torch.device | str) -> torch.device:
    """Where tensors handed to a collective must live: the rank's GPU under nccl/RCCL, the CPU
    under gloo (CPU tests, or a rehearsal of the multi-rank path on a single-GPU box)."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return torch.device(default)


def _collectives_on() -> bool:
    """A process group with more than one rank - or any group when HYDRO_DIST_ALWAYS=1, which lets a single-GPU box
    drive the real RCCL calls (a one-rank communicator) through the same code the multi-GPU run takes."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("HYDRO_DIST_ALWAYS") == "1"


def all_reduce_sum_(t: torch.Tensor, async_op: bool = False):
    """In-place SUM all-reduce of a small tensor; no-op without a process group."""
    if _collectives_on():
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)
    return None


def all_reduce_max_(t: torch.Tensor):
    if _collectives_on():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def global_kinetic_energy(local_ke: torch.Tensor, async_op: bool = False):
    """All-reduce the per-rank kinetic-energy scalar(s) (float64, shape (1,) or (2,)
    = [translational, rotational]).  `local_ke` lives on the rank's device for
    nccl/RCCL and on the CPU for gloo.  Returns (tensor, work-handle-or-None)."""
    if local_ke.dtype != torch.float64:
        raise TypeError("kinetic energy partials are float64")
    work = all_reduce_sum_(local_ke, async_op=async_op)
    return local_ke, work


def barrier():
    if _collectives_on():
        dist.barrier()
