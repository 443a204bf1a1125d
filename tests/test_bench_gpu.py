"""bench.py contract: one JSON line with the fields the driver reads."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_bench_json_contract(native_built):
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
           "--bodies", "65536", "--cpu-seconds", "1", "--no-extras"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{")          # exactly one line on stdout, whatever libraries print
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "body-steps/s" and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    # ONE clock: `frac` is on the interval `value` and `ms_per_step` come from (bytes per launch / ms_per_step) ...
    assert r["frac"] * r["peak"] * 1e9 * d["ms_per_step"] * 1e-3 == pytest.approx(r["algorithmic_bytes_per_launch"], rel=1e-9)
    assert d["value"] == pytest.approx(65536 / (d["ms_per_step"] * 1e-3), rel=1e-9)
    # ... and the HIP-event figure of the same steps is kept beside it (shorter: no host synchronisation in it)
    assert r["frac_contract_steps"] >= r["frac"] and r["kernel_us"] <= r["step_us"] * 1.0001
    assert r["traffic_bytes_per_body"] == 122 and r["frac_traffic"] == pytest.approx(r["frac"] * 122 / 130)
    assert r["resident"] in ("hbm", "infinity-cache") and "roofline_4m" not in d      # 65 536 bodies x 4: a cache-resident test size
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    assert c["gpu_vs_oracle_max_rel_err"] <= 1e-5 and c["gpu_vs_oracle_n_over_1e-5"] == 0     # the CPU leg is also the checker
    assert d["value"] > 1e8


def test_bench_two_ranks_share_the_gpu(native_built):
    """The driver's N>1 launch (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) rehearsed
    with two ranks on this box's one GPU (gloo for the collectives, HYDRO_BENCH_SHARE_GPU=1): one JSON line from
    rank 0, whole-job aggregate over both ranks, no CPU leg, kinetic-energy all-reduce done."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, HYDRO_BENCH_SHARE_GPU="1", HYDRO_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "8",
           "--bodies", "65536", "--spinup-seconds", "0.2"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{")          # the gloo / RCCL banners go to stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["config"]["bodies_per_gpu"] == 65536 and "x2" in d["config"]["sharding"]
    assert d["value"] == pytest.approx(2 * 65536 * 40 / (d["ms_per_step"] * 1e-3 * 40), rel=1e-6)
    assert len(d["global_kinetic_energy_J"]) == 2 and d["global_kinetic_energy_J"][0] > 0
    # N > 1 also runs BASELINE configs[3] as stated: 262 144 bodies block-partitioned over the ranks (strong scaling),
    # with the global kinetic energy sampled by the asynchronous monitor
    cs = d["c4_strong"]
    assert cs["scaling"] == "strong" and cs["baseline_config"] == "configs[3]" and cs["n_gpus"] == 2
    assert cs["bodies_total"] == 262144 and cs["bodies_this_rank"] == 131072
    assert cs["value"] == pytest.approx(262144 * 40 / (cs["ms_per_step"] * 1e-3 * 40), rel=1e-6)
    ke = cs["kinetic_energy"]
    assert ke["every_steps"] == 256 and ke["samples"] == 0        # 40 steps: no sampling point reached ...
    # ... so check the monitor itself against the fp64 host sum over ALL 262 144 bodies in a second, longer run
    with socket.socket() as s2:
        s2.bind(("127.0.0.1", 0)); port2 = s2.getsockname()[1]
    cmd2 = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(port2), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "600", "--warmup", "8",
            "--bodies", "65536", "--spinup-seconds", "0.1"]
    res2 = subprocess.run(cmd2, capture_output=True, text=True, timeout=600, env=env)
    assert res2.returncode == 0, res2.stderr[-3000:]
    ke2 = json.loads([l for l in res2.stdout.splitlines() if l.strip()][0])["c4_strong"]["kinetic_energy"]
    assert ke2["samples"] == 2 and ke2["last_step"] == 512
    import numpy as np
    sys.path.insert(0, REPO)
    from silver2_isaacsim_amd import scenes
    sc = scenes.scene_c4(n=262144, seed=4)
    m = sc.params[:, 10].astype(np.float64)
    lin = float((0.5 * m * (sc.state[:, 7:10].astype(np.float64) ** 2).sum(1)).sum())
    assert ke2["global_J"][0] == pytest.approx(lin, rel=1e-12)


def test_bench_gpus_2_launches_its_own_ranks(native_built):
    """`python bench.py --gpus 2` with NO torchrun environment must not fall through to a one-GPU run: it starts the two
    ranks itself (before anything touches the GPU) and relays rank 0's line.  Rehearsed on this box's one GPU with both
    ranks sharing it and gloo for the collectives."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HYDRO_BENCH_SHARE_GPU="1", HYDRO_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "8", "--bodies", "65536",
           "--spinup-seconds", "0.2"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{")
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["collectives"] == "gloo, 2 rank(s)" and "x2" in d["config"]["sharding"]
    assert d["barrier"] == "node-local shared-memory epoch barrier"      # the ranks of one host time their region with it
    assert d["c4_strong"]["n_gpus"] == 2 and d["c4_strong"]["bodies_this_rank"] == 131072


def test_bench_refuses_more_gpus_than_visible(native_built):
    """--gpus N with fewer than N devices visible (and no rehearsal knob): non-zero exit, a message, no JSON line."""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "HYDRO_BENCH_SHARE_GPU")}
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode != 0 and res.stdout.strip() == ""
    assert f"--gpus {n} but only {n - 1} GPU(s) are visible" in res.stderr
