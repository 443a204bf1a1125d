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
    assert d["unit"] == "body-steps/s" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"])
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
