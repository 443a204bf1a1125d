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


def init_process_group(backend: str | None = None, device_id: int | None = None, force: bool = False,
                       node_barrier: bool = False) -> bool:
    """Initialise torch.distributed from the env when WORLD_SIZE>1 (or, with force=True, a one-rank group: the way a
    single-GPU box drives the real RCCL calls).  Returns True if a group is active.  backend None -> 'nccl' (= RCCL)
    on GPU, 'gloo' otherwise.
    node_barrier=True: barrier() of this module becomes the node-local shared-memory barrier (NodeBarrier) - a
    measurement tool: its ranks spin on a host core for the microseconds a timed region opens and closes in.  bench.py
    asks for it; a host that also runs a simulator does not, and gets torch.distributed's own barrier."""
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
    global _node_barrier
    _node_barrier = None            # (a new group: a barrier of an earlier one is not its barrier)
    if node_barrier:                # here, not at the first barrier(): its set-up takes milliseconds of host time, and a
        enable_node_barrier()       # GPU left idle that long right before a timed region starts it at a lower clock
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


def live_ranks(device: torch.device | str = "cpu") -> int:
    """How many ranks the LIVE process group really joins in a collective: every rank contributes a 1 to a SUM
    all-reduce (RCCL under backend "nccl").  1 without a group.  A launcher that started fewer ranks than the
    benchmark was asked for shows up here, whatever the environment variables say."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    one = torch.ones(1, dtype=torch.int64, device=collective_device(device))
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(one.item())


def gather_rows(row, device: torch.device | str = "cpu", dtype=torch.float64) -> torch.Tensor:
    """(world, k) tensor on the CPU whose row r is rank r's `row` (k numbers, the same k on every rank).  A SUM
    all-reduce in which every rank fills only its own row of zeros: x + 0 is exact, so nothing of the result depends
    on the order the ranks are added in - a gather built from the one collective the path needs anyway."""
    vals = torch.as_tensor(row, dtype=dtype).reshape(-1).cpu()
    if not _collectives_on():
        return vals.reshape(1, -1)
    world, rank = dist.get_world_size(), dist.get_rank()
    t = torch.zeros((world, vals.numel()), dtype=dtype)
    t[rank] = vals
    t = t.to(collective_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu()


def global_kinetic_energy(local_ke: torch.Tensor, async_op: bool = False):
    """All-reduce the per-rank kinetic-energy scalar(s) (float64, shape (1,) or (2,)
    = [translational, rotational]).  `local_ke` lives on the rank's device for
    nccl/RCCL and on the CPU for gloo.  Returns (tensor, work-handle-or-None)."""
    if local_ke.dtype != torch.float64:
        raise TypeError("kinetic energy partials are float64")
    work = all_reduce_sum_(local_ke, async_op=async_op)
    return local_ke, work


class NodeBarrier:
    """Barrier for the ranks of ONE host through a shared-memory page: rank r is the only writer of slot r (one cache
    line each), a barrier is "publish my epoch, spin until every slot has reached it".  The ranks of a bench run sit on
    one node by contract; a dist.barrier() over RCCL is an all-reduce launch plus a stream synchronisation (tens of
    microseconds), which is a tenth of a 20-step timed region - this one costs about a microsecond, and its ranks
    leave it within a microsecond of each other, which is what the max-over-ranks wall time wants.  Built collectively
    (every rank calls NodeBarrier.create at the same point); None when the ranks span hosts."""

    LINE = 8                                              # int64 per slot = one 64-byte line

    def __init__(self, rank: int, world: int, mm):
        import numpy as np
        self.rank, self.world, self._mm = rank, world, mm
        self.slots = np.frombuffer(mm, dtype=np.int64, count=world * self.LINE).reshape(world, self.LINE)
        self.epoch = 0

    @classmethod
    def create(cls) -> "NodeBarrier | None":
        import mmap
        import socket
        rank, world = dist.get_rank(), dist.get_world_size()
        try:
            boot = open("/proc/sys/kernel/random/boot_id").read().strip()
        except OSError:
            boot = ""
        ids = [None] * world
        dist.all_gather_object(ids, (socket.gethostname(), boot))
        if len(set(ids)) != 1 or not os.path.isdir("/dev/shm"):
            return None
        # /dev/shm is world-writable: the name is unpredictable (rank 0 draws it, the others are told), the file must not
        # exist yet (O_EXCL) and no component may be a symbolic link somebody planted (O_NOFOLLOW)
        name = [f"/dev/shm/hydro_barrier_{os.getuid()}_{os.urandom(12).hex()}" if rank == 0 else None]
        dist.broadcast_object_list(name, src=0)
        path = name[0]
        size = max(4096, world * cls.LINE * 8)
        fd, mm, ok = -1, None, True
        try:
            if rank == 0:
                fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW | os.O_RDWR, 0o600)
                os.ftruncate(fd, size)                    # zero-filled
        except OSError:
            ok = False
        dist.barrier()                                    # the file exists with its final size
        try:
            if rank != 0:
                fd = os.open(path, os.O_RDWR | os.O_NOFOLLOW)
                if os.fstat(fd).st_uid != os.getuid():
                    raise OSError("barrier page owned by another user")
            mm = mmap.mmap(fd, size)
            os.close(fd)
        except (OSError, ValueError):
            ok = False
        nb = cls(rank, world, mm) if ok else None
        if nb is not None:
            nb.slots[rank, 1] = rank + 1                  # self-test: is this page really the one the others see?
        dist.barrier()                                    # everybody has it mapped and signed
        if rank == 0:
            try:
                os.unlink(path)                           # the mappings keep the page; nothing is left behind
            except OSError:
                pass
        seen = nb is not None and nb.slots[:, 1].tolist() == list(range(1, world + 1))
        flags = [None] * world
        dist.all_gather_object(flags, bool(seen))
        return nb if all(flags) else None                 # all ranks or none: a mixed choice would deadlock

    def wait(self, timeout_s: float = 120.0) -> None:
        import time
        self.epoch += 1
        e = self.epoch
        self.slots[self.rank, 0] = e
        col = self.slots[:, 0]
        spins = 0
        t_end = None
        while int(col.min()) < e:
            spins += 1
            # ranks of a timed region arrive within microseconds of each other: spin for those; a rank that is late by
            # more (it is still setting up, or the host is oversubscribed) is waited for politely - give the core away,
            # then sleep - so that the barrier never burns a core the late rank, or a simulator, needs
            if spins > 2000:
                if spins > 20000:
                    time.sleep(50e-6)
                else:
                    os.sched_yield()
            if spins & 0xFFF == 0:                        # a rank that died must not hang the others for ever
                now = time.monotonic()
                if t_end is None:
                    t_end = now + timeout_s
                elif now > t_end:
                    raise TimeoutError(f"node barrier: rank {self.rank} waited {timeout_s:.0f} s at epoch {e}: {col.tolist()}")


_node_barrier: "NodeBarrier | None" = None                  # None = torch.distributed's own barrier


def enable_node_barrier() -> bool:
    """COLLECTIVE (every rank of the group calls it at the same point): switch barrier() to the node-local shared-memory
    barrier.  False - and barrier() stays torch.distributed's - when the ranks span hosts, /dev/shm is unusable, or
    HYDRO_BARRIER=dist is set."""
    global _node_barrier
    if not (dist.is_available() and dist.is_initialized()) or os.environ.get("HYDRO_BARRIER") == "dist":
        return False
    if _node_barrier is None:
        _node_barrier = NodeBarrier.create()
    return _node_barrier is not None


def barrier(timeout_s: float | None = None):
    """Barrier over all ranks; no-op without a process group.  torch.distributed's own unless a measurement asked for the
    node-local one (init_process_group(node_barrier=True) / enable_node_barrier()).  timeout_s (node-local barrier only): how
    long to wait for a rank that never arrives before raising TimeoutError (default 120 s)."""
    if not _collectives_on():
        return
    if _node_barrier is None:
        dist.barrier()
    elif timeout_s is None:
        _node_barrier.wait()
    else:
        _node_barrier.wait(timeout_s)


def barrier_kind() -> str:
    if not _collectives_on():
        return "none (single process)"
    return "torch.distributed.barrier" if _node_barrier is None else "node-local shared-memory epoch barrier"
