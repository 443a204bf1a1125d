"""ctypes binding of oracle/libhydro_oracle.so (TEST INFRASTRUCTURE ONLY; see
oracle/hydro_oracle.c).  Built by `make -C oracle` / `__graft_entry__.build()`."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhydro_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "hydro_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        dp, fp = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float)
        L.hydro_oracle_components.argtypes = [ctypes.c_int64, dp, dp, dp, ctypes.c_double, ctypes.c_double, dp, dp]
        L.hydro_oracle_components.restype = ctypes.c_int
        L.hydro_oracle_wrench.argtypes = [ctypes.c_int64, fp, fp, fp, ctypes.c_double, ctypes.c_double,
                                          ctypes.c_double, dp, dp, ctypes.c_int]
        L.hydro_oracle_wrench.restype = ctypes.c_int
        L.hydro_oracle_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def components(state, accel, params, rho, g):
    state = np.ascontiguousarray(state, dtype=np.float64)
    accel = np.ascontiguousarray(accel, dtype=np.float64)
    params = np.ascontiguousarray(params, dtype=np.float64)
    n = state.shape[0]
    comps = np.empty((n, 8, 3)); ratio = np.empty(n)
    rc = lib().hydro_oracle_components(n, _dp(state), _dp(accel), _dp(params), rho, g, _dp(comps), _dp(ratio))
    assert rc == 0
    return comps, ratio


def wrench(state, prev, params, rho, g, dt, threads=1):
    state = np.ascontiguousarray(state, dtype=np.float32)
    prev = np.ascontiguousarray(prev, dtype=np.float32)
    params = np.ascontiguousarray(params, dtype=np.float32)
    n = state.shape[0]
    f = np.empty((n, 3)); t = np.empty((n, 3))
    rc = lib().hydro_oracle_wrench(n, _fp(state), _fp(prev), _fp(params), rho, g, dt, _dp(f), _dp(t), int(threads))
    assert rc == 0
    return f, t


def max_threads() -> int:
    return int(lib().hydro_oracle_max_threads())
