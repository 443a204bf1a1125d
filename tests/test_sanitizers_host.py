"""AddressSanitizer + UndefinedBehaviorSanitizer on the CPU builds of the native code that CAN run on a CPU: the host
instantiation of the kernel arithmetic (csrc/hydro_body.h through tests/host_emul/emul.cpp) and the C oracle
(oracle/hydro_oracle.c).  GPU sanitizers are not available on the pool; what they would look at - the per-body arithmetic
on degenerate inputs: zero dimensions, zero / non-unit quaternions, exact ties, NaN-producing divisions - is this code.
Both are rebuilt with -fsanitize=address,undefined -fno-sanitize-recover into a temporary directory and driven from a child
Python (the sanitizer runtime must be the first library of the process: LD_PRELOAD) over the edge-case table, the surface-tie
fixture and a special-value population; any report aborts the child."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

CHILD = r'''
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.environ["HYDRO_REPO"]); sys.path.insert(0, os.path.join(os.environ["HYDRO_REPO"], "tests"))
import edge_cases as ec
import populations
emul = ctypes.CDLL(os.environ["HYDRO_SAN_EMUL"]); orc = ctypes.CDLL(os.environ["HYDRO_SAN_ORACLE"])
fp = ctypes.POINTER(ctypes.c_float); dp = ctypes.POINTER(ctypes.c_double)
def f32(a): return np.ascontiguousarray(a, dtype=np.float32)
def emul_wrench(st, pv, pr, rho, g, dt):
    n = len(st); st, pv, pr = f32(st), f32(pv), f32(pr)
    f = np.empty((n, 3), np.float32); t = np.empty((n, 3), np.float32); r = np.empty(n, np.float32)
    assert emul.emul_wrench(ctypes.c_int64(n), st.ctypes.data_as(fp), pv.ctypes.data_as(fp), pr.ctypes.data_as(fp), ctypes.c_double(rho),
                            ctypes.c_double(g), ctypes.c_double(dt), f.ctypes.data_as(fp), t.ctypes.data_as(fp), r.ctypes.data_as(fp)) == 0
    return f, t, r
def emul_components(st, ac, pr, rho, g):
    n = len(st); st, ac, pr = f32(st), f32(ac), f32(pr)
    out = np.empty((n, 8, 3), np.float32); r = np.empty(n, np.float32)
    assert emul.emul_components(ctypes.c_int64(n), st.ctypes.data_as(fp), ac.ctypes.data_as(fp), pr.ctypes.data_as(fp), ctypes.c_double(rho),
                                ctypes.c_double(g), out.ctypes.data_as(fp), r.ctypes.data_as(fp)) == 0
    return out, r
total = 0
# 1. the 81-case degenerate table (zero sizes, ties, zero and non-unit quaternions), both semantics
for warp in (0, 1):
    emul.emul_set_semantics(warp)
    f, t, r = emul_wrench(ec.STATE, ec.PREV, ec.PARAMS, ec.RHO, ec.G, ec.DT)
    c, cr = emul_components(ec.STATE, ec.ACCEL32, ec.PARAMS, ec.RHO, ec.G)
    total += len(ec.STATE)
emul.emul_set_semantics(0)
# 2. quantised surface ties and a special-value population with zero / tiny dimensions and zero quaternions
st, pv, pr = populations.surface_ties(2048, seed=77)[:3]
emul_wrench(st, pv, pr, populations.RHO, populations.G, populations.DT); total += len(st)
sys.path.insert(0, os.path.join(os.environ["HYDRO_REPO"], "tests", "tools"))
import reference_fuzz
st, pv, pr, rho, g, dt, acc = reference_fuzz.population(6000, 99)
with np.errstate(all="ignore"):
    f, t, r = emul_wrench(st, pv, pr, rho, g, dt)
    emul_components(st, acc, pr, rho, g)
total += len(st)
# 3. the C oracle (OpenMP, one and several threads) on the same populations, through its ctypes binding
from oracle import c_oracle
c_oracle._LIB_PATH = os.environ["HYDRO_SAN_ORACLE"]; c_oracle.build = lambda *a, **k: c_oracle._LIB_PATH     # the sanitized build, not the shipped one
for threads in (1, 3):
    with np.errstate(all="ignore"):
        c_oracle.wrench(f32(st), f32(pv), f32(pr), rho, g, dt, threads=threads)
        c_oracle.wrench(f32(ec.STATE), f32(ec.PREV), f32(ec.PARAMS), ec.RHO, ec.G, ec.DT, threads=threads)
print("SANITIZED-OK", total)
'''


@pytest.fixture(scope="module")
def sanitized(tmp_path_factory):
    d = tmp_path_factory.mktemp("san")
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
    emul = str(d / "libemul_san.so")
    r = subprocess.run(["g++", *san, "-march=x86-64-v3", "-ffp-contract=fast", "-fPIC", "-shared", "-o", emul,
                        os.path.join(REPO, "tests", "host_emul", "emul.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    orc = str(d / "liboracle_san.so")
    r = subprocess.run(["gcc", *san, "-march=x86-64-v3", "-fPIC", "-fopenmp", "-shared", "-o", orc,
                        os.path.join(REPO, "oracle", "hydro_oracle.c"), "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not (os.path.isabs(asan) and os.path.exists(asan)):
        pytest.skip("no libasan on this box")
    return emul, orc, ":".join(p for p in (asan, ubsan) if os.path.isabs(p) and os.path.exists(p))


def test_host_arithmetic_and_oracle_are_clean_under_asan_and_ubsan(sanitized):
    emul, orc, preload = sanitized
    env = dict(os.environ, HYDRO_REPO=REPO, HYDRO_SAN_EMUL=emul, HYDRO_SAN_ORACLE=orc, LD_PRELOAD=preload,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               OMP_NUM_THREADS="3", PYTHONDONTWRITEBYTECODE="1")
    res = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-4000:])
    assert "SANITIZED-OK" in res.stdout
    assert "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr, res.stderr[-4000:]
