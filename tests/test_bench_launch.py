"""bench.py --gpus N can not be mis-launched: without a torchrun environment it starts its own ranks, and it refuses
to benchmark fewer GPUs than asked.  CPU-only half (no device is touched before the check)."""
import os
import subprocess
import sys

from conftest import REPO


def test_gpus_n_without_devices_exits_non_zero():
    import torch
    n = torch.cuda.device_count() + 2
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "HYDRO_BENCH_SHARE_GPU")}
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n)], capture_output=True, text=True,
                         timeout=300, env=env)
    assert res.returncode != 0
    assert res.stdout.strip() == ""                                  # no JSON line that could be mistaken for a result
    assert f"--gpus {n} but only {n - 2} GPU(s) are visible" in res.stderr


def test_world_size_mismatch_is_an_error():
    """A torchrun environment whose WORLD_SIZE differs from --gpus is refused before any device work."""
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert res.returncode != 0 and "--gpus 2 but WORLD_SIZE=3" in res.stderr and res.stdout.strip() == ""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert res.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in res.stderr
