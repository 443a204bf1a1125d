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
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
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
    # fp32 floor: a body whose buoyancy and drag cancel >40x can sit marginally above 1e-5
    assert c["gpu_vs_oracle_max_rel_err"] <= 2e-5 and c["gpu_vs_oracle_n_over_1e-5"] <= 2
    assert d["value"] > 1e8
