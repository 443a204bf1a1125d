"""N>1 path on CPU: world_size-2 gloo.  Bodies shard with no data-path collective; the one
collective is the kinetic-energy all-reduce.  The per-rank partial here is computed with NumPy
(on the GPU box it is the device reduction of hydro_kinetic_energy, covered by the gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from silver2_isaacsim_amd import distributed as hd
    from silver2_isaacsim_amd import scenes
    assert hd.env_rank_world() == (rank, rank, world)
    assert hd.init_process_group(backend="gloo")
    sc = scenes.scene_c4(n=10001, seed=8)                     # odd size: ragged shards
    mine = sc.shard(rank, world)
    lo, hi = hd.shard_range(sc.n, rank, world)
    assert mine.n == hi - lo
    m = mine.params[:, 10].astype(np.float64)
    ke_local = float((0.5 * m * (mine.state[:, 7:10].astype(np.float64) ** 2).sum(1)).sum())
    t = torch.tensor([ke_local, 0.0], dtype=torch.float64)
    out, work = hd.global_kinetic_energy(t, async_op=True)    # asynchronous, off the step path
    work.wait()
    mt = sc.params[:, 10].astype(np.float64)
    ke_full = float((0.5 * mt * (sc.state[:, 7:10].astype(np.float64) ** 2).sum(1)).sum())
    # max-over-ranks timing used by bench.py
    tm = torch.tensor([1.0 + rank], dtype=torch.float64)
    hd.all_reduce_max_(tm)
    hd.barrier()
    q.put((rank, float(out[0]), ke_full, float(tm[0]), mine.n))
    dist.destroy_process_group()


def test_world_size_2_shards_and_ke_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert sum(r[4] for r in res) == 10001
    for _, ke_sum, ke_full, tmax, _ in res:
        assert ke_sum == pytest.approx(ke_full, rel=1e-12)     # 8e: all-reduce == fp64 host sum
        assert tmax == 2.0


def test_single_process_helpers_are_noops():
    from silver2_isaacsim_amd import distributed as hd
    t = torch.tensor([3.0], dtype=torch.float64)
    out, work = hd.global_kinetic_energy(t)
    assert work is None and float(out[0]) == 3.0
    hd.barrier()
    with pytest.raises(TypeError):
        hd.global_kinetic_energy(torch.tensor([1.0], dtype=torch.float32))
