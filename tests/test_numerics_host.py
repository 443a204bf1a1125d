"""fp32 arithmetic of the GPU kernels (csrc/hydro_body.h compiled for the host, test-only)
against the fp64 oracle: a pre-GPU gate on the closed forms and their conditioning.
Tolerance: the SURVEY.md 8d metric, gate 1e-5 (north_star: "within 1e-5 relative")."""
import ctypes
import os

import numpy as np
import pytest

from conftest import REPO, SCENE_FIXTURES, load_golden
from oracle import hydro_oracle as ho

GATE = 1e-5


@pytest.fixture(scope="module")
def emul(native_built):
    lib = ctypes.CDLL(os.path.join(REPO, "tests", "host_emul", "libemul.so"))
    fp = ctypes.POINTER(ctypes.c_float)

    def run(state, prev, params, rho, g, dt):
        st = np.ascontiguousarray(state, np.float32); pv = np.ascontiguousarray(prev, np.float32)
        pr = np.ascontiguousarray(params, np.float32)
        n = len(st)
        f = np.empty((n, 3), np.float32); t = np.empty((n, 3), np.float32); r = np.empty(n, np.float32)
        rc = lib.emul_wrench(ctypes.c_int64(n), st.ctypes.data_as(fp), pv.ctypes.data_as(fp), pr.ctypes.data_as(fp),
                             ctypes.c_double(rho), ctypes.c_double(g), ctypes.c_double(float(dt)),
                             f.ctypes.data_as(fp), t.ctypes.data_as(fp), r.ctypes.data_as(fp))
        assert rc == 0
        return f, t, r
    return run


@pytest.mark.parametrize("name", SCENE_FIXTURES)
def test_fp32_arithmetic_within_gate_on_fixtures(name, emul):
    fx = load_golden(name)
    rho, g, dt = float(fx["rho"]), float(fx["g"]), float(fx["dt"])
    f, t, r = emul(fx["state"], fx["prev"], fx["params"], rho, g, dt)
    rf, rt, aux = ho.step_wrench(fx["state"], fx["prev"], fx["params"], rho, g, dt)
    err = ho.wrench_error(f, t, rf, rt, fx["params"], rho, g)
    assert np.isfinite(f).all() and np.isfinite(t).all()
    assert err.max() <= GATE, f"{name}: max {err.max():.3e}"
    assert np.abs(r - aux["ratio"]).max() < 5e-7
    dry = aux["ratio"] == 0.0
    assert np.all(f[dry] == 0.0) and np.all(t[dry] == 0.0)       # exact zeros, not small numbers


def test_ungated_population(emul):
    """No branch-margin rule: 65 536 bodies straight from the C4 law.  The keypoint and face tests are taken on fp64
    heights, like the reference's, so bodies with a keypoint on the waterline within fp32 resolution no longer flip."""
    from silver2_isaacsim_amd import scenes
    sc = scenes.scene_c4(n=65536, seed=2024, margin=None)
    f, t, _ = emul(sc.state, sc.prev, sc.params, sc.rho, sc.g, sc.dt)
    rf, rt, _ = ho.step_wrench(sc.state, sc.prev, sc.params, sc.rho, sc.g, sc.dt)
    err = ho.wrench_error(f, t, rf, rt, sc.params, sc.rho, sc.g)
    assert err.max() <= 5e-7


def test_barely_wet_bodies_are_well_conditioned(emul):
    """ratio in [1e-6, 1e-2]: z_min = p_z - extent cancels catastrophically in fp32 (1e-2 at ratio 1e-5); in fp64 it
    is exact to the final rounding."""
    from silver2_isaacsim_amd import scenes
    rng = np.random.default_rng(7)
    sc = scenes.scene_c4(n=8192, seed=31)
    ext = scenes.vertical_extent(sc.state[:, 3:7], sc.params[:, :3])
    ratio = np.exp(rng.uniform(np.log(1e-6), np.log(1e-2), sc.n))
    sc.state[:, 2] = (ext * (1.0 - 2.0 * ratio)).astype(np.float32)
    f, t, r = emul(sc.state, sc.prev, sc.params, sc.rho, sc.g, sc.dt)
    rf, rt, aux = ho.step_wrench(sc.state, sc.prev, sc.params, sc.rho, sc.g, sc.dt)
    wet = aux["ratio"] > 0
    rel = np.abs(r[wet] - aux["ratio"][wet]) / aux["ratio"][wet]
    assert rel.max() < 1.2e-7
    fz = np.abs(f[wet, 2] - rf[wet, 2]) / np.maximum(np.abs(rf[wet, 2]), 1e-30)
    assert np.median(fz) < 1e-7


def test_large_world_offsets_do_not_hurt(emul):
    """|p_xy| ~ 1e4 m: lever arms are body-relative, so x/y never enter the arithmetic."""
    fx = load_golden("c2")
    rho, g, dt = float(fx["rho"]), float(fx["g"]), float(fx["dt"])
    st = fx["state"].copy()
    f0, t0, _ = emul(st, fx["prev"], fx["params"], rho, g, dt)
    st[:, 0] += 12345.0; st[:, 1] -= 54321.0
    f1, t1, _ = emul(st, fx["prev"], fx["params"], rho, g, dt)
    assert np.array_equal(f0, f1) and np.array_equal(t0, t1)


@pytest.mark.parametrize("scale", [1.0 + 1e-5, 1.001, 1.02, 0.9])
def test_non_unit_quaternions_host(scale, emul):
    """The reference uses the quaternion as given (N7): with e = |q|^2 - 1 its matrix is not orthogonal, and terms that
    vanish for a rotation carry e.  The fp64 evaluation follows the same polynomial in q, so parity holds for any |q|."""
    from silver2_isaacsim_amd import scenes
    sc = scenes.scene_c4(n=16384, seed=21)
    st = sc.state.copy()
    st[:, 3:7] = (st[:, 3:7].astype(np.float64) * scale).astype(np.float32)
    ext = scenes.vertical_extent(st[:, 3:7], sc.params[:, :3]); ext0 = scenes.vertical_extent(sc.state[:, 3:7], sc.params[:, :3])
    st[:, 2] = (sc.state[:, 2].astype(np.float64) * ext / ext0).astype(np.float32)
    keep = scenes.branch_margins(st, sc.params) > 1e-4
    f, t, _ = emul(st, sc.prev, sc.params, sc.rho, sc.g, sc.dt)
    rf, rt, _ = ho.step_wrench(st, sc.prev, sc.params, sc.rho, sc.g, sc.dt)
    err = ho.wrench_error(f, t, rf, rt, sc.params, sc.rho, sc.g)[keep]
    assert err.max() <= 5e-7


def test_degenerate_inputs_host(emul):
    import edge_cases as ec
    f, t, r = emul(ec.STATE, ec.PREV, ec.PARAMS, ec.RHO, ec.G, ec.DT)
    ec.check(f, t, r)
