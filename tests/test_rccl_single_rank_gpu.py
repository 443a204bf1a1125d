"""The real RCCL code path on a single-GPU box: a ONE-rank communicator (backend "nccl" = RCCL on ROCm) with
HYDRO_DIST_ALWAYS=1, so that every collective of the N > 1 path - the float64 all-reduce of the kinetic-energy monitor
issued with async_op=True on its side stream, the max-over-ranks timing, the barriers - goes through RCCL exactly as it
will on a multi-GPU node (there: same calls, more ranks).  Run in a child process: a process group is per process."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["HYDRO_REPO"])
from silver2_isaacsim_amd import distributed as hd, scenes
from silver2_isaacsim_amd.simulate import ClosedLoopSim
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and hd.collective_device(torch.device("cuda", 0)).type == "cuda"
sc = scenes.scene_c2(n=5000)
sim = ClosedLoopSim(sc, ke_every=64)
assert sim.monitor._nccl
sim.run(256, graph_steps=64)
sim.synchronize()
sim.monitor.collect(block=True)
ref = ClosedLoopSim(sc)
want = []
for _ in range(4):
    ref.run(64, graph_steps=64)
    st = ref.state().astype(np.float64); m = sc.params[:, 10].astype(np.float64)
    want.append(float((0.5 * m * (st[:, 7:10] ** 2).sum(1)).sum()))
t = torch.tensor([2.5], dtype=torch.float64, device="cuda:0")
hd.all_reduce_max_(t); hd.barrier()
print(json.dumps({"steps": [s for s, _ in sim.monitor.samples], "ke": [v[0] for _, v in sim.monitor.samples], "want": want,
                  "host_waits": sim.monitor.waited_on_host, "max": float(t.item()),
                  "captured": sim._captured_samples, "capturable": sim.monitor.graph_capturable}))
sim.close(); ref.close()
dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_monitor_all_reduce_goes_through_rccl(native_built):
    import json
    env = dict(os.environ, HYDRO_REPO=REPO, HYDRO_DIST_ALWAYS="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    res = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["steps"] == [64, 128, 192, 256] and d["max"] == 2.5
    assert d["ke"] == pytest.approx(d["want"], rel=1e-12)
    assert d["capturable"] and d["captured"] == 4      # the RCCL all-reduce of every sample ran INSIDE the replayed step graph


def _bench_over_rccl(steps, warmup):
    import json
    env = dict(os.environ, HYDRO_DIST_ALWAYS="1", HYDRO_BENCH_FORCE_GROUP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--bodies", "65536",
           "--spinup-seconds", "0.1", "--cpu-seconds", "0", "--no-extras"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    return json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("steps,warmup,every,samples", [(20, 5, 10, 2), (600, 8, 256, 2)])
def test_bench_strong_leg_over_rccl(native_built, steps, warmup, every, samples):
    """bench.py's N > 1 path (weak headline + c4_strong leg + monitor) with the nccl backend and one rank: WORLD_SIZE=1
    normally means "no process group", HYDRO_BENCH_FORCE_GROUP=1 makes the bench build a one-rank RCCL group and take
    the multi-rank code path - at the driver's own `--steps 20 --warmup 5` and at a long region.  The asynchronous
    all-reduce on the side stream runs at least twice inside the timed region and the leg's self-checks hold."""
    d = _bench_over_rccl(steps, warmup)
    cs = d["c4_strong"]
    ke = cs["ke"]
    assert cs["bodies_this_rank"] == 262144 and ke["every_steps"] == every and ke["samples"] == samples and ke["samples"] >= 2
    assert ke["host_waits"] == 0 or steps == 20        # (a 20-step region is over before the first sample has landed: collect(block=True) waits once or twice)
    assert ke["rel_err"] <= 1e-12 and cs["shards_bit_identical"] is True and d["ok"] is True
    # the same leg with each sample's pipeline - RCCL all-reduce and pinned copy included - captured into the step graph
    gr = cs["captured"]
    assert "error" not in gr and gr["samples"] == samples and gr["rel_err"] <= 1e-12, gr
    assert gr["sampled_at_steps"] == ke["sampled_at_steps"]
    if steps == 20:
        assert gr["ms_per_step"] < 1.25 * cs["ms_per_step"]    # no host work per sample: measured 9.2 vs 14.3 us (a loose bound: host timing)
    col = d["collective"]
    assert col["ke_rel_err"] <= 1e-12
    assert col["backend"] == "nccl (RCCL)" and col["rccl_ranks"] == 1 and col["ranks"] == 1
    assert col["barrier"] == "node-local shared-memory epoch barrier"      # built over the RCCL group's own collectives


CHILD_C_ROUTE = r'''
import ctypes, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.environ["HYDRO_REPO"])
from silver2_isaacsim_amd import scenes
from silver2_isaacsim_amd.engine import HydroEngine
torch.cuda.set_device(0)
sc = scenes.scene_c4(n=70000, seed=3)
eng = HydroEngine(sc.n, "cuda:0", sc.rho, sc.g)
eng.set_params(sc.params)
st = torch.from_numpy(scenes.to_tiled(sc.state)).to("cuda:0")
ke = eng.kinetic_energy(st, rotational=True)
local = ke.clone()
rccl = ctypes.CDLL(os.environ.get("HYDRO_TEST_RCCL", "librccl.so.1"))       # whichever copy the process resolves: the library binds the same one
comm = ctypes.c_void_p()
dev0 = (ctypes.c_int * 1)(0)
assert rccl.ncclCommInitAll(ctypes.byref(comm), 1, dev0) == 0
origins = [HydroEngine.bind_rccl(rccl)]                                      # the copy that made the communicator: no guessing
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
eng.ke_allreduce(comm.value, ke, stream=side)                                # 16 bytes, on a side stream
side.synchronize()
ok = torch.equal(ke, local)
origins.append(HydroEngine.bind_rccl(None))                                  # forget it: the next call looks one up by itself
eng.ke_allreduce(comm.value, ke, stream=side)
side.synchronize()
ok = ok and torch.equal(ke, local)
origins.append(eng._lib.hydro_rccl_origin().decode())
try:
    eng.ke_allreduce(0, ke)
    refused = False
except Exception as e:
    refused = "HYDRO_E_ARG" in str(e)
rccl.ncclCommDestroy(comm)
print(json.dumps({"same": bool(ok), "refused_null": refused, "ke": ke.tolist(), "origins": origins}))
eng.close()
'''


def test_ke_allreduce_from_the_c_abi(native_built):
    """hydro_ke_allreduce (ncclAllReduce(count 2, ncclDouble, ncclSum) on a side stream) with a one-rank
    communicator made through RCCL's own C API - no torch.distributed anywhere: the route a plain-C host takes."""
    import json
    env = dict(os.environ, HYDRO_REPO=REPO)
    res = subprocess.run([sys.executable, "-c", CHILD_C_ROUTE], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["same"] and d["refused_null"] and d["ke"][0] > 0
    assert d["origins"][0] == "hydro_bind_rccl" and d["origins"][1] == "unbound"
    assert d["origins"][2] in ("the copy already loaded in the process", "librccl opened by libhydro", "HYDRO_RCCL_LIBRARY")
