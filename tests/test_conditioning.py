"""Designed cancellation cases (tests/populations.py) against the fp64 oracle, gate 1e-5.

* terminal rise: drag along z cancels buoyancy 100x and 300x.  Needs buoyancy + z-drag summed in fp64 AND the
  scene scalars handed over as doubles (g = 9.81 rounded to fp32 is already 4e-8 off).  The all-fp32 form this
  repo started the round with: max 2.2e-5 / 2.5e-5, 8 / 22 of 4 096 bodies over the gate; now max 1.7e-6.
* near-upright floaters (the buoy scenes): at rest the whole torque is the horizontal buoyancy lever arm, 1 %
  (0.1 %) of its length at 0.5 (0.05) degrees of tilt.  Benign for both forms (3.9e-7 -> 1.9e-7): the wet lattice
  is symmetric, so the arm is a single product; kept as the physical sanity case of the buoyancy torque."""
import ctypes
import os

import numpy as np
import pytest

import populations as pop
from conftest import REPO
from oracle import hydro_oracle as ho

GATE = 1e-5
CASES = [("terminal_rise_100x", lambda: pop.terminal_rise(cancel=100.0)),
         ("terminal_rise_300x", lambda: pop.terminal_rise(seed=13, cancel=300.0)),
         ("floaters_0.5deg", lambda: pop.near_upright_floaters()),
         ("floaters_0.05deg", lambda: pop.near_upright_floaters(seed=14, tilt_deg=0.05))]


def _check(name, f, t, state, prev, params):
    rf, rt, aux = ho.step_wrench(state, prev, params, pop.RHO, pop.G, pop.DT)
    err = ho.wrench_error(f, t, rf, rt, params, pop.RHO, pop.G)
    assert err.max() <= GATE, f"{name}: max {err.max():.3e}"
    if name.startswith("terminal"):
        b = aux["buoyancy_force"][:, 2]
        cancel = b / np.maximum(np.abs(rf[:, 2] / aux["scale"]), 1e-300)
        assert np.median(cancel) > 50.0                               # the population is what it claims to be
        assert np.median(err) < 2e-6
    else:
        assert np.median(err) < 1e-6
    return err


@pytest.fixture(scope="module")
def emul(native_built):
    lib = ctypes.CDLL(os.path.join(REPO, "tests", "host_emul", "libemul.so"))
    fp = ctypes.POINTER(ctypes.c_float)

    def run(state, prev, params):
        n = len(state)
        f = np.empty((n, 3), np.float32); t = np.empty((n, 3), np.float32); r = np.empty(n, np.float32)
        st, pv, pr = (np.ascontiguousarray(x, np.float32) for x in (state, prev, params))
        assert lib.emul_wrench(ctypes.c_int64(n), st.ctypes.data_as(fp), pv.ctypes.data_as(fp), pr.ctypes.data_as(fp),
                               ctypes.c_double(pop.RHO), ctypes.c_double(pop.G), ctypes.c_float(np.float32(1.0 / pop.DT)),
                               f.ctypes.data_as(fp), t.ctypes.data_as(fp), r.ctypes.data_as(fp)) == 0
        return f, t
    return run


@pytest.mark.parametrize("name,make", CASES)
def test_cancellation_cases_host_arithmetic(name, make, emul):
    state, prev, params = make()
    f, t = emul(state, prev, params)
    _check(name, f, t, state, prev, params)


@pytest.mark.gpu
@pytest.mark.parametrize("name,make", CASES)
def test_cancellation_cases_gpu(name, make, native_built):
    import torch
    from silver2_isaacsim_amd import scenes
    from silver2_isaacsim_amd.engine import HydroEngine
    state, prev, params = make()
    n = len(state)
    eng = HydroEngine(n, "cuda:0", pop.RHO, pop.G)
    eng.set_params(params)
    out = eng.step_wrench_tiled(torch.from_numpy(scenes.to_tiled(state)).to("cuda:0"), n, pop.DT,
                                prev=torch.from_numpy(scenes.to_tiled(prev)).to("cuda:0"))
    o = scenes.from_tiled(out.cpu().numpy(), n)
    eng.close()
    _check(name, o[:, :3], o[:, 3:], state, prev, params)
