"""The C ABI used from plain C (no Python, no torch on the product side): examples/c_abi_demo.c is
compiled against include/hydro.h + libhydro.so and its output compared with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import REPO, load_golden
from oracle import hydro_oracle as ho

pytestmark = pytest.mark.gpu


def test_plain_c_host_program(tmp_path, native_built):
    exe = str(tmp_path / "c_abi_demo")
    lib_dir = os.path.join(REPO, "silver2_isaacsim_amd", "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", "-std=c11", "-O2", "-D__HIP_PLATFORM_AMD__", "-DHYDRO_DEMO_WITH_RCCL", os.path.join(REPO, "examples", "c_abi_demo.c"),
           "-I", os.path.join(rocm, "include"), "-I", os.path.join(REPO, "include"),
           "-L", lib_dir, "-lhydro", "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-lrccl",
           f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{os.path.join(rocm, 'lib')}", "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    fx = load_golden("c4")
    n = 1000                                      # not a multiple of 64: ragged last tile
    blob = str(tmp_path / "bodies.bin")
    with open(blob, "wb") as f:
        f.write(struct.pack("<q", n))
        f.write(np.ascontiguousarray(fx["state"][:n], np.float32).tobytes())
        f.write(np.ascontiguousarray(fx["prev"][:n], np.float32).tobytes())
        f.write(np.ascontiguousarray(fx["params"][:n], np.float32).tobytes())
        f.write(struct.pack("<f", float(fx["dt"])))
    run = subprocess.run([exe, blob], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr[-3000:]
    assert "HYDRO_E_STATE" not in run.stdout and "parameters not set" in run.stderr      # the message of the provoked error
    rows = [line.split() for line in run.stdout.strip().splitlines()]
    rows = [r for r in rows if len(r) == 6 and all(c[0] in "+-0123456789" for c in r)]       # (RCCL announces its version on stdout)
    got = np.array([[float(x) for x in r] for r in rows])
    assert got.shape == (n, 6)
    rho, g = float(fx["rho"]), float(fx["g"])
    err = ho.wrench_error(got[:, :3], got[:, 3:], fx["net_force"][:n], fx["net_torque"][:n], fx["params"][:n], rho, g)
    assert err.max() <= 1e-5
    assert "bit-identical to the eager loop" in run.stderr                         # 64 fused steps from a HIP graph, from C
    assert "resident loop: 63 + 1 steps in two launches, bit-identical" in run.stderr  # hydro_step_fused_tiled_multi from C
    assert "batched launch: 2 scenes in one launch, bit-identical" in run.stderr       # hydro_step_wrench_tiled_batch from C
    assert "global kinetic energy over 1 rank(s) through RCCL" in run.stderr          # hydro_ke_allreduce: SURVEY.md 8e, from C
    ke_line = [l for l in run.stderr.splitlines() if l.startswith("kinetic energy")][0]
    lin = float(ke_line.split()[2])
    assert lin == pytest.approx(ho.kinetic_energy(fx["state"][:n], fx["params"][:n])[0], rel=1e-8)


def test_library_before_torch_in_a_fresh_process(native_built):
    """`__graft_entry__.build()` (which loads libhydro.so) followed by `smoke()` in ONE fresh process: the binding
    must end up on the HIP runtime torch ships, whichever of the two is asked for first (two runtimes in one
    process made hipGetDeviceCount fail -> HYDRO_E_DEVICE)."""
    import sys
    code = "import __graft_entry__ as g; g.build(); g.smoke()"
    res = subprocess.run([sys.executable, "-c", code], cwd=REPO, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "[smoke]" in res.stdout, res.stderr[-2000:]
    code = ("from silver2_isaacsim_amd import _native as n; n.load(); import torch; "
            "from silver2_isaacsim_amd.engine import HydroEngine; e = HydroEngine(16, 'cuda:0'); e.close(); print('ok')")
    res = subprocess.run([sys.executable, "-c", code], cwd=REPO, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "ok" in res.stdout, res.stderr[-2000:]
