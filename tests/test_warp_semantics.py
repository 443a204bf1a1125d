"""`semantics="warp"`: the two places where the reference's Warp twin (warp_hydrodynamics.py:233-335, the
calculator hydrodynamics_behavior.py:155 instantiates) differs from the Numba path.

PARITY UNPINNED for this mode: `warp` is not importable here and the reference ships no outputs of it, so
the oracle's Warp branch is a restatement from source text only.  What IS checked: the restatement against
closed forms, scalar vs vectorised oracle, and the HIP path / its host instantiation against the oracle.
The default (Numba) mode must not move."""
import ctypes
import os

import numpy as np
import pytest

from conftest import REPO, load_golden
from oracle import hydro_oracle as ho

GATE = 1e-5


def _quat_z(deg):
    h = np.deg2rad(deg) / 2.0
    return np.array([0.0, 0.0, np.sin(h), np.cos(h)])


def test_oracle_warp_added_mass_closed_form():
    """Isotropic linear added mass: Numba gives -m a for any attitude, Warp gives -m R R a
    (warp_hydrodynamics.py:216-230): a 90 degree yaw turns the horizontal part of a by 180 degrees."""
    params = np.array([1.0, 1.0, 1.0, 1.2, 0.8, 300.0, 150.0, 1.0, 0.05, 0.02, 500.0])
    p, v, w = np.array([0.0, 0.0, -5.0]), np.array([0.1, 0.0, 0.0]), np.zeros(3)
    a, alpha = np.array([1.0, 2.0, 3.0]), np.array([0.5, -1.0, 0.25])
    m = 1025.0 * 1.0 * 0.05
    nb = ho.solve_components_one(p, _quat_z(90), v, w, a, alpha, params, 1025.0, 9.81)
    wp = ho.solve_components_one(p, _quat_z(90), v, w, a, alpha, params, 1025.0, 9.81, semantics="warp")
    assert np.allclose(nb[4], -m * a, rtol=1e-12)
    assert np.allclose(wp[4], -m * np.array([-1.0, -2.0, 3.0]), rtol=1e-12, atol=1e-12)
    # cube: the angular diagonal is isotropic too (2 rho V c), same rotation rule
    mt = 1025.0 * 1.0 * 2.0 * 0.02
    assert np.allclose(nb[5], -mt * alpha, rtol=1e-12)
    assert np.allclose(wp[5], -mt * np.array([-0.5, 1.0, 0.25]), rtol=1e-12, atol=1e-12)
    for k in (0, 1, 2, 3, 6, 7):                      # everything else is common to the two calculators
        assert np.array_equal(nb[k], wp[k])
    # identity and half-turn attitudes: R = R^T, the two calculators agree
    for deg in (0, 180):
        x = ho.solve_components_one(p, _quat_z(deg), v, w, a, alpha, params, 1025.0, 9.81)
        y = ho.solve_components_one(p, _quat_z(deg), v, w, a, alpha, params, 1025.0, 9.81, semantics="warp")
        assert np.allclose(x[4], y[4], atol=1e-12) and np.allclose(x[5], y[5], atol=1e-12)


def test_oracle_warp_dry_body_centres():
    """N6: a dry body reports cob = cop = position in Warp (zeros in Numba); forces are zero in both."""
    params = np.array([1.0, 1.0, 1.0, 1.2, 0.8, 300.0, 150.0, 1.0, 0.05, 0.02, 500.0])
    p = np.array([3.0, -4.0, 5.0])
    args = (p, _quat_z(30), np.ones(3), np.ones(3), np.ones(3), np.ones(3), params, 1025.0, 9.81)
    nb = ho.solve_components_one(*args)
    wp = ho.solve_components_one(*args, semantics="warp")
    assert all(np.all(x == 0.0) for x in nb[:8]) and nb[8] == 0.0
    assert all(np.all(x == 0.0) for x in wp[:6]) and wp[8] == 0.0
    assert np.array_equal(wp[6], p) and np.array_equal(wp[7], p)


def test_oracle_warp_scalar_equals_vectorised():
    fx = load_golden("c4")
    st, pr = fx["state"].astype(np.float64), fx["params"].astype(np.float64)
    acc = (st[:, 7:13] - fx["prev"].astype(np.float64)) / float(fx["dt"])
    rho, g = float(fx["rho"]), float(fx["g"])
    vec = ho.solve_components(st, acc, pr, rho, g, semantics="warp")
    for i in range(0, len(st), 7):
        one = ho.solve_components_one(st[i, 0:3], st[i, 3:7], st[i, 7:10], st[i, 10:13], acc[i, :3], acc[i, 3:], pr[i], rho, g,
                                      semantics="warp")
        for k, name in enumerate(ho.COMPONENT_FIELDS):
            assert np.allclose(vec[name][i], one[k], rtol=1e-12, atol=1e-12), (i, name)
    # and the mode matters on this scene: added mass differs from the Numba result for tilted bodies
    nb = ho.solve_components(st, acc, pr, rho, g)
    wet = nb["ratio"] > 0
    assert np.abs(vec["added_mass_force"][wet] - nb["added_mass_force"][wet]).max() > 1e-3


@pytest.fixture(scope="module")
def emul(native_built):
    lib = ctypes.CDLL(os.path.join(REPO, "tests", "host_emul", "libemul.so"))
    fp = ctypes.POINTER(ctypes.c_float)

    def run(state, prev, params, rho, g, dt, warp):
        st = np.ascontiguousarray(state, np.float32); pv = np.ascontiguousarray(prev, np.float32)
        pr = np.ascontiguousarray(params, np.float32)
        n = len(st)
        f = np.empty((n, 3), np.float32); t = np.empty((n, 3), np.float32); r = np.empty(n, np.float32)
        lib.emul_set_semantics(int(warp))
        try:
            rc = lib.emul_wrench(ctypes.c_int64(n), st.ctypes.data_as(fp), pv.ctypes.data_as(fp), pr.ctypes.data_as(fp),
                                 ctypes.c_double(rho), ctypes.c_double(g), ctypes.c_double(float(dt)),
                                 f.ctypes.data_as(fp), t.ctypes.data_as(fp), r.ctypes.data_as(fp))
        finally:
            lib.emul_set_semantics(0)
        assert rc == 0
        return f, t
    return run


@pytest.mark.parametrize("name", ["c2", "c4", "c5"])
def test_host_arithmetic_warp_mode(name, emul):
    fx = load_golden(name)
    rho, g, dt = float(fx["rho"]), float(fx["g"]), float(fx["dt"])
    f, t = emul(fx["state"], fx["prev"], fx["params"], rho, g, dt, warp=True)
    rf, rt, _ = ho.step_wrench(fx["state"], fx["prev"], fx["params"], rho, g, dt, semantics="warp")
    assert ho.wrench_error(f, t, rf, rt, fx["params"], rho, g).max() <= GATE
    # the default mode is untouched by the switch having been used
    f0, t0 = emul(fx["state"], fx["prev"], fx["params"], rho, g, dt, warp=False)
    assert ho.wrench_error(f0, t0, fx["net_force"], fx["net_torque"], fx["params"], rho, g).max() <= GATE


# ------------------------------------------------------------------------------------------- GPU
def _engine(n, rho, g, params, coeff="f32", semantics="numba"):
    from silver2_isaacsim_amd.engine import HydroEngine
    eng = HydroEngine(n, "cuda:0", rho, g)
    eng.set_params(params, coeff)
    eng.set_semantics(semantics)
    return eng


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2", "c4", "c5"])
def test_gpu_wrench_warp_mode_every_entry(name, native_built):
    import torch
    from silver2_isaacsim_amd import scenes
    fx = load_golden(name)
    rho, g, dt = float(fx["rho"]), float(fx["g"]), float(fx["dt"])
    st, pv, pr = fx["state"], fx["prev"], fx["params"]
    n = len(st)
    rf, rt, _ = ho.step_wrench(st, pv, pr, rho, g, dt, semantics="warp")
    eng = _engine(n, rho, g, pr, "f16" if name == "c5" else "f32", "warp")
    dev = "cuda:0"
    soa = eng.step_wrench(torch.from_numpy(scenes.to_soa(st)).to(dev), dt, prev=torch.from_numpy(scenes.to_soa(pv)).to(dev)).cpu().numpy().T
    til = scenes.from_tiled(eng.step_wrench_tiled(torch.from_numpy(scenes.to_tiled(st)).to(dev), n, dt,
                                                  prev=torch.from_numpy(scenes.to_tiled(pv)).to(dev)).cpu().numpy(), n)
    eng.set_prev_velocity(pv)
    F, T = eng.step_wrench_aos(torch.from_numpy(np.ascontiguousarray(st[:, 0:3])).to(dev),
                               torch.from_numpy(np.ascontiguousarray(st[:, [6, 3, 4, 5]])).to(dev),
                               torch.from_numpy(np.ascontiguousarray(st[:, 7:13])).to(dev), dt)
    aos = np.concatenate([F.cpu().numpy(), T.cpu().numpy()], 1)
    assert np.array_equal(soa, til)
    for o in (soa, aos):
        assert ho.wrench_error(o[:, :3], o[:, 3:], rf, rt, pr, rho, g).max() <= GATE
    # back to the default: the reference-executed Numba fixtures again, and different from the Warp result
    eng.set_semantics("numba")
    nb = eng.step_wrench(torch.from_numpy(scenes.to_soa(st)).to(dev), dt, prev=torch.from_numpy(scenes.to_soa(pv)).to(dev)).cpu().numpy().T
    assert ho.wrench_error(nb[:, :3], nb[:, 3:], fx["net_force"], fx["net_torque"], pr, rho, g).max() <= GATE
    assert np.abs(nb - soa).max() > 1e-3
    eng.close()


@pytest.mark.gpu
def test_gpu_components_warp_mode_and_wrapper(native_built):
    import torch
    from silver2_isaacsim_amd.wrapper import HipHydrodynamicsWrapper
    fx = load_golden("c4")
    rho, g = float(fx["rho"]), float(fx["g"])
    st, pr = fx["state"], fx["params"]
    n = len(st)
    acc = ((st[:, 7:13].astype(np.float64) - fx["prev"].astype(np.float64)) / float(fx["dt"])).astype(np.float32)
    ref = ho.solve_components(st, acc, pr, rho, g, semantics="warp")
    w = HipHydrodynamicsWrapper(pr[:, 0], pr[:, 1], pr[:, 2], pr[:, 3], pr[:, 4], pr[:, 5], pr[:, 6], rho, g,
                                pr[:, 8], pr[:, 9], pr[:, 7], device="cuda:0", semantics="warp")
    outs = w.calculate_hydrodynamic_forces(st[:, 0:3], st[:, 3:7], st[:, 7:10], st[:, 10:13], acc[:, :3], acc[:, 3:])
    torch.cuda.synchronize()
    outs = [o.cpu().numpy() for o in outs]
    vol = pr[:, :3].astype(np.float64).prod(1)
    floor = np.maximum(1e-3 * rho * g * vol, 1e-12)
    for k, name in enumerate(ho.COMPONENT_FIELDS[:6]):
        rel = np.linalg.norm(outs[k] - ref[name], axis=1) / np.maximum(np.linalg.norm(ref[name], axis=1), floor)
        assert rel.max() < 5e-5, name
    dry = ref["ratio"] == 0
    assert dry.any()
    assert np.array_equal(outs[6][dry], st[dry, 0:3]) and np.array_equal(outs[7][dry], st[dry, 0:3])     # N6
    assert np.abs(outs[6] - ref["center_of_buoyancy"]).max() < 3e-5 and np.abs(outs[7] - ref["center_of_pressure"]).max() < 3e-5
    w.close()
    with pytest.raises(ValueError):
        HipHydrodynamicsWrapper(1, 1, 1, 1, 1, 1, 1, 1025.0, 9.81, 0, 0, 0, device="cuda:0", semantics="cuda")
