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


def _worker(rank, world, port, q, barrier_kind="node"):
    import sys
    sys.path.insert(0, REPO)
    if barrier_kind == "dist":
        os.environ["HYDRO_BARRIER"] = "dist"
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from silver2_isaacsim_amd import distributed as hd
    from silver2_isaacsim_amd import scenes
    assert hd.env_rank_world() == (rank, rank, world)
    assert hd.init_process_group(backend="gloo", node_barrier=(barrier_kind != "default"))
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
    # what bench.py's N > 1 line proves itself with: the ranks that really join an all-reduce, an exact gather of per-rank
    # rows (digests of the shard wrenches, host sums), the float64 host sum of the kinetic energy
    assert hd.live_ranks() == world
    rows = hd.gather_rows([rank + 0.1, -rank, 1e300 * (rank + 1)])
    assert rows.shape == (world, 3) and rows.tolist() == [[r + 0.1, -r, 1e300 * (r + 1)] for r in range(world)]
    dig = hd.gather_rows(list(range(rank, rank + 32)), dtype=torch.int64)
    assert dig.dtype == torch.int64 and dig.tolist() == [list(range(r, r + 32)) for r in range(world)]
    lin_full, rot_full = scenes.kinetic_energy_fp64(sc.state, sc.params)
    parts = hd.gather_rows(scenes.kinetic_energy_fp64(mine.state, mine.params))
    assert float(parts[:, 0].sum()) == pytest.approx(lin_full, rel=1e-14) and float(parts[:, 1].sum()) == pytest.approx(rot_full, rel=1e-14)
    assert lin_full == pytest.approx(ke_full, rel=1e-14) and rot_full > 0
    # max-over-ranks timing used by bench.py
    tm = torch.tensor([1.0 + rank], dtype=torch.float64)
    hd.all_reduce_max_(tm)
    hd.barrier()
    # the node-local barrier really orders the ranks: rank 1 dawdles before each of 50 barriers, rank 0 must not get ahead
    import time
    if barrier_kind in ("dist", "default"):
        # HYDRO_BARRIER=dist: the fallback ranks on several hosts take; "default": nobody asked for the node barrier -
        # a host that also runs a simulator gets torch.distributed's own (the spinning one is bench.py's tool)
        assert hd.barrier_kind() == "torch.distributed.barrier" and hd._node_barrier is None
    else:
        assert hd.barrier_kind() == "node-local shared-memory epoch barrier"
        nb = hd._node_barrier
        for k in range(50):
            if rank == 1 and k % 10 == 0:
                time.sleep(0.01)
            hd.barrier()
            assert int(nb.slots[:, 0].min()) >= nb.epoch and int(nb.slots[:, 0].max()) <= nb.epoch + 1
    assert not [f for f in os.listdir("/dev/shm") if f.startswith(f"hydro_barrier_{os.getuid()}_")]      # unlinked once mapped
    q.put((rank, float(out[0]), ke_full, float(tm[0]), mine.n))
    dist.destroy_process_group()


@pytest.mark.parametrize("barrier_kind", ["node", "dist", "default"])
def test_world_size_2_shards_and_ke_allreduce(barrier_kind):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, barrier_kind)) for r in range(world)]
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


def _deserter(rank, world, port, q):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from silver2_isaacsim_amd import distributed as hd
    assert hd.init_process_group(backend="gloo", node_barrier=True)
    hd.barrier()
    if rank == 1:                                   # leaves without reaching the next barrier
        q.put((rank, "left"))
        return
    try:
        hd._node_barrier.wait(timeout_s=1.0)
        q.put((rank, "passed"))
    except TimeoutError as e:
        q.put((rank, f"timeout: {e}"))


def test_node_barrier_times_out_instead_of_hanging_when_a_rank_is_gone():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_deserter, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res[1] == "left" and res[0].startswith("timeout: node barrier: rank 0 waited 1 s at epoch 2")


def test_single_process_helpers_are_noops():
    from silver2_isaacsim_amd import distributed as hd
    t = torch.tensor([3.0], dtype=torch.float64)
    out, work = hd.global_kinetic_energy(t)
    assert work is None and float(out[0]) == 3.0
    hd.barrier()
    with pytest.raises(TypeError):
        hd.global_kinetic_energy(torch.tensor([1.0], dtype=torch.float32))


def _monitor_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from silver2_isaacsim_amd import distributed as hd
    from silver2_isaacsim_amd import scenes
    from silver2_isaacsim_amd.simulate import KineticEnergyMonitor
    assert hd.init_process_group(backend="gloo")
    sc = scenes.scene_c4(n=262144 // 64, seed=4)              # BASELINE configs[3] population, scaled down
    mine = sc.shard(rank, world)
    m = mine.params[:, 10].astype(np.float64)
    vel = mine.state[:, 7:10].astype(np.float64)
    step = {"k": 0}

    def reduce_local(out):                                    # stands in for hydro_kinetic_energy on this CPU-only box
        v = vel * (1.0 + 0.01 * step["k"])                    # the "state" changes from sample to sample
        out[0] = float((0.5 * m * (v ** 2).sum(1)).sum()); out[1] = 0.0
    mon = KineticEnergyMonitor(None, every=8, device="cpu", reduce_local=reduce_local)
    for k in range(1, 41):                                    # 40 steps -> 5 samples, collected lazily
        step["k"] = k
        mon.observe(k)
    mon.collect(block=True)
    mt = sc.params[:, 10].astype(np.float64); vt = sc.state[:, 7:10].astype(np.float64)
    want = [float((0.5 * mt * ((vt * (1.0 + 0.01 * k)) ** 2).sum(1)).sum()) for k in (8, 16, 24, 32, 40)]
    q.put((rank, [s for s, _ in mon.samples], [v[0] for _, v in mon.samples], want, mon.submitted))
    hd.barrier()
    dist.destroy_process_group()


def test_kinetic_energy_monitor_two_ranks():
    """SURVEY 8e as designed: per-rank reduce every K steps, asynchronous all-reduce, results picked up later;
    the global value equals the fp64 host sum over ALL bodies to 1e-12 on every rank."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_monitor_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, steps, got, want, submitted in res:
        assert steps == [8, 16, 24, 32, 40] and submitted == 5
        assert got == pytest.approx(want, rel=1e-12)


def _deadline_worker(rank, world, port, q):
    import sys
    import time
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from silver2_isaacsim_amd import distributed as hd
    from silver2_isaacsim_amd.simulate import KineticEnergyMonitor
    assert hd.init_process_group(backend="gloo")

    def reduce_local(out):
        out[0] = 1.0 + rank; out[1] = 0.0
    mon = KineticEnergyMonitor(None, every=4, device="cpu", reduce_local=reduce_local, timeout_s=None)
    mon.observe(4)                                            # both ranks join sample 1 ...
    mon.collect(block=True, timeout_s=30.0)
    if rank == 0:
        mon.observe(8)                                        # ... only rank 0 submits sample 2
        t0 = time.monotonic()
        try:
            mon.collect(block=True, timeout_s=1.5)
            q.put((rank, "no timeout", 0.0, mon.samples))
        except TimeoutError as e:
            q.put((rank, str(e), time.monotonic() - t0, mon.samples))
    else:
        q.put((rank, "idle", 0.0, mon.samples))
        time.sleep(6.0)                                       # alive (its sockets open) while rank 0 waits in vain
    q.close(); q.join_thread()
    os._exit(0)                                               # (a collective is outstanding on rank 0: leave without a teardown barrier)


def test_a_sample_nobody_else_joins_raises_within_the_deadline():
    """VERDICT r5 item 2 on the CPU: `collect(block=True, timeout_s=)` polls the sample's work handle against a monotonic
    clock and raises TimeoutError naming the step and the rank; it does not hang, retry or re-execute anything."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_deadline_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r[0], r[1:]) for r in (q.get(timeout=180) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
    msg, waited, samples = res[0]
    assert "kinetic-energy sample of step 8 on rank 0" in msg and "did not finish within 1.5 s" in msg
    assert 1.4 <= waited < 20.0
    assert samples == [(4, [3.0, 0.0])] and res[1][2] == [(4, [3.0, 0.0])]      # the first sample was summed over both ranks


def test_kinetic_energy_monitor_single_process_cadence():
    from silver2_isaacsim_amd.simulate import KineticEnergyMonitor
    calls = []

    def reduce_local(out):
        calls.append(1); out[0] = float(len(calls)); out[1] = 0.5
    mon = KineticEnergyMonitor(None, every=4, device="cpu", reduce_local=reduce_local, slots=2)
    assert [mon.observe(k) for k in range(1, 13)] == [False, False, False, True] * 3
    mon.collect(block=True)
    assert mon.samples == [(4, [1.0, 0.5]), (8, [2.0, 0.5]), (12, [3.0, 0.5])] and mon.last()[0] == 12
    with pytest.raises(ValueError):
        KineticEnergyMonitor(None, every=0, device="cpu", reduce_local=reduce_local)
