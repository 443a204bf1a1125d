"""The oracle (NumPy fp64 + C fp64) against outputs of the reference itself.

tests/golden/*.npz were produced by tests/golden/make_golden.py, which executes the
reference's own functions (numba_hydrodynamics.py / numba_hydrodynamics_wrapper.py) on
seeded fp32-exact inputs.  The same script executes the reference's `_apply_behavior` and stores its
net wrench; the K1-K5 behaviour-level numbers (net force / torque / clamp) recorded in SURVEY.md
section 8c are checked as well.
"""
import numpy as np
import pytest

from conftest import SCENE_FIXTURES, accel_of, load_golden
from oracle import c_oracle
from oracle import hydro_oracle as ho

TOL = 1e-12


def _rel(got, ref, floor=1e-12):
    return (np.linalg.norm(got - ref, axis=-1) / np.maximum(np.linalg.norm(ref, axis=-1), floor)).max()


def _floor(name):
    """Denominator floor of the relative error.  On the quantised `ties` bodies a term that is mathematically 0 (lift with
    the velocity in the plane of the up vector: sin(2 asin d) with d = 0 up to rounding) comes out as 5e-15 N from the
    reference and as exactly 0 from another summation order: rounding noise of a term whose neighbours are O(100) N is
    compared absolutely there (1e-12 N)."""
    return 1.0 if name == "ties" else 1e-12


@pytest.mark.parametrize("name", ["kat"] + SCENE_FIXTURES)
def test_numpy_oracle_matches_reference_outputs(name):
    fx = load_golden(name)
    out = ho.solve_components(fx["state"], accel_of(fx), fx["params"], float(fx["rho"]), float(fx["g"]))
    for i, field in enumerate(ho.COMPONENT_FIELDS):
        assert _rel(out[field], fx["components"][:, i, :], _floor(name)) < TOL, field
    assert np.abs(out["ratio"] - fx["ratio"]).max() < 1e-15 + 1e-12
    if "rest_completed" in fx:
        assert np.array_equal(out["rest"], fx["rest_completed"])


@pytest.mark.parametrize("name", ["kat"] + SCENE_FIXTURES)
def test_c_oracle_matches_reference_outputs(name, native_built):
    fx = load_golden(name)
    comps, ratio = c_oracle.components(fx["state"], accel_of(fx), fx["params"], float(fx["rho"]), float(fx["g"]))
    assert _rel(comps, fx["components"], _floor(name)) < TOL
    assert np.abs(ratio - fx["ratio"]).max() < 1e-12


@pytest.mark.parametrize("name", SCENE_FIXTURES)
def test_scalar_restatement_matches_batch(name):
    fx = load_golden(name)
    acc = accel_of(fx)
    rho, g = float(fx["rho"]), float(fx["g"])
    for i in range(0, len(fx["state"]), max(1, len(fx["state"]) // 64)):
        s = fx["state"][i]
        one = ho.solve_components_one(s[0:3], s[3:7], s[7:10], s[10:13], acc[i, :3], acc[i, 3:], fx["params"][i], rho, g)
        for k in range(8):
            assert np.abs(np.asarray(one[k]) - fx["components"][i, k]).max() <= 1e-9 * max(1.0, np.abs(fx["components"][i, k]).max())
        assert abs(one[8] - fx["ratio"][i]) < 1e-14


@pytest.mark.parametrize("name", SCENE_FIXTURES)
def test_c_wrench_equals_numpy_wrench(name, native_built):
    fx = load_golden(name)
    rho, g, dt = float(fx["rho"]), float(fx["g"]), float(fx["dt"])
    f, t = c_oracle.wrench(fx["state"], fx["prev"], fx["params"], rho, g, dt)
    f2, t2, _ = ho.step_wrench(fx["state"], fx["prev"], fx["params"], rho, g, dt)
    err = ho.wrench_error(f, t, f2, t2, fx["params"], rho, g)
    assert err.max() < 1e-9      # two fp64 evaluation orders; the arm x drag cancellation amplifies 1e-16


@pytest.mark.parametrize("name", ["kat"] + SCENE_FIXTURES)
def test_fused_oracle_matches_the_reference_behavior_script(name, native_built):
    """`net_force` / `net_torque` in the fixtures are what the reference's own `_apply_behavior`
    (hydrodynamics_behavior.py:176-238, executed unchanged in float64) hands to the simulator:
    quaternion reorder, finite-difference acceleration, lever arms, sum, clamp - A13-A16."""
    fx = load_golden(name)
    rho, g, dt = float(fx["rho"]), float(fx["g"]), float(fx["dt"])
    f, t, _ = ho.step_wrench(fx["state"], fx["prev"], fx["params"], rho, g, dt)
    assert ho.wrench_error(f, t, fx["net_force"], fx["net_torque"], fx["params"], rho, g).max() < 1e-9   # fp64 vs fp64; the lever-arm cancellation amplifies 1e-16
    fc, tc = c_oracle.wrench(fx["state"].astype(np.float32), fx["prev"].astype(np.float32), fx["params"].astype(np.float32), rho, g, dt)
    tol = 1e-9 if name != "kat" else 1e-5          # the C entry takes fp32 inputs; K1-K5 inputs are not fp32-exact
    assert ho.wrench_error(fc, tc, fx["net_force"], fx["net_torque"], fx["params"], rho, g).max() < tol


# ---- K1-K5: behaviour-level known answers -------------------------------
KAT_EXPECT = {
    "K1": dict(ratio=0.6310344827586206, buoy_z=6345.209482758621,
               net_f=(-151.43459442254508, -46.262195873560046, 6009.799326553225),
               net_t=(-1.77439654899338, 138.5819542233543, 83.60922169670674), scale=1.0),
    "K2": dict(ratio=1.0, buoy_z=203.92047000000002,
               net_f=(-25.16164006846788, -5.088934648536673, 207.64831560551696),
               net_t=(0.002638071814222473, -7.912149076663272, -45.1973958333495), scale=1.0),
    "K3": dict(ratio=1.0, buoy_z=3.2579009999999995,
               net_f=(-0.2845960556464743, -0.11431218924213835, 3.315417910535692),
               net_t=(-0.005489045751996897, 0.010983927080558797, -0.02742265663134303), scale=1.0),
    "K4": dict(ratio=0.30724950744484997, buoy_z=9268.411829204482,
               net_f=(-112.94046647252904, -1.6731920958893187, 487.0745850738602),
               net_t=(17.987732522593024, -17.376269320457695, 0.45117533795075376), scale=0.03630474159376046),
}


def test_known_answers_behaviour_level():
    fx = load_golden("kat")
    names = [str(x) for x in fx["names"]]
    comps = ho.solve_components(fx["state"], fx["accel"], fx["params"], 1025.0, 9.81)
    net_f, net_t, scale = ho.behavior_epilogue(fx["state"][:, 0:3], comps, fx["params"][:, 10])
    for i, nm in enumerate(names):
        if nm == "K5":           # dry: all nine outputs zero, including cob / cop (N6)
            for field in ho.COMPONENT_FIELDS:
                assert np.all(comps[field][i] == 0.0)
            assert comps["ratio"][i] == 0.0 and np.all(net_f[i] == 0.0) and np.all(net_t[i] == 0.0)
            continue
        e = KAT_EXPECT[nm]
        assert comps["ratio"][i] == pytest.approx(e["ratio"], rel=1e-13)
        assert comps["buoyancy_force"][i, 2] == pytest.approx(e["buoy_z"], rel=1e-13)
        assert net_f[i] == pytest.approx(e["net_f"], rel=1e-11, abs=1e-12)
        assert net_t[i] == pytest.approx(e["net_t"], rel=1e-9, abs=1e-12)
        assert scale[i] == pytest.approx(e["scale"], rel=1e-12)
    # K4: zero projected area -> centre of pressure falls back to the centre of buoyancy, no lift
    k4 = names.index("K4")
    assert comps["area"][k4] == 0.0
    assert np.array_equal(comps["center_of_pressure"][k4], comps["center_of_buoyancy"][k4])
    assert np.all(comps["lift_force"][k4] == 0.0)


# ---- config 1: single-buoy trajectory ---------------------------------------------------
def test_c1_single_buoy_trajectory():
    """10 000 closed-loop steps (semi-implicit Euler standing in for PhysX) reproduce the
    trajectory generated with the reference's functions; z settles towards the analytic
    equilibrium 0.5 - m/(rho*A*h)."""
    fx = load_golden("c1_trajectory")
    params, mass, dt = fx["params"], float(fx["mass"]), float(fx["dt"])
    rho, g = float(fx["rho"]), float(fx["g"])
    p = np.array([0.0, 0.0, np.float64(np.float32(0.3))]); q = np.array([0.0, 0.0, 0.0, 1.0])   # fp32-exact inputs
    v = np.array([0.0, 0.0, np.float64(np.float32(-1e-3))]); w = np.zeros(3)
    v_last = np.zeros(3); w_last = np.zeros(3)
    z = np.zeros(len(fx["z"]))
    fired = []
    for k in range(len(z)):
        a = (v - v_last) / dt
        comps = ho.solve_components_one(p, q, v, w, a, (w - w_last) / dt, params, rho, g)
        if comps[8] > 1e-9 and not np.linalg.norm(v) > 1e-6:
            fired.append(k)
        net_f, _, _ = ho.behavior_epilogue_one(p, comps, mass)
        v_last = v.copy(); w_last = w.copy()
        v = v + dt * (net_f / mass + np.array([0.0, 0.0, -g]))
        p = p + dt * v
        z[k] = p[2]
    assert np.abs(z - fx["z"]).max() < 1e-9
    assert fired == list(fx["rest_steps"])                 # the N1 rest branch really occurs
    assert z.min() == pytest.approx(-0.252, abs=2e-3) and z.max() == pytest.approx(0.298, abs=2e-3)
    assert abs(z[-1] - (0.5 - mass / rho)) < 2e-3


# ---- properties of the model -------------------------------------
def _c4_sample(n=4096, seed=123):
    from silver2_isaacsim_amd import scenes
    return scenes.scene_c4(n=n, seed=seed)


def test_dry_bodies_are_exactly_zero():
    sc = _c4_sample()
    f, t, aux = ho.step_wrench(sc.state, sc.prev, sc.params, sc.rho, sc.g, sc.dt)
    dry = aux["ratio"] == 0.0
    assert dry.sum() > 500
    assert np.all(f[dry] == 0.0) and np.all(t[dry] == 0.0)


def test_fully_submerged_buoyancy_is_rho_v_g():
    sc = _c4_sample()
    _, _, aux = ho.step_wrench(sc.state, sc.prev, sc.params, sc.rho, sc.g, sc.dt)
    full = aux["ratio"] == 1.0
    vol = sc.params[:, :3].astype(np.float64).prod(axis=1)
    assert full.sum() > 1000
    assert np.allclose(aux["buoyancy_force"][full, 2], sc.rho * vol[full] * sc.g, rtol=1e-14)


def test_yaw_equivariance():
    """Rotating the whole scene about world z rotates the wrench with it."""
    sc = _c4_sample(2048)
    th = 0.7
    c, s = np.cos(th), np.sin(th)
    rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    qz = np.array([0.0, 0.0, np.sin(th / 2), np.cos(th / 2)])
    st = sc.state.astype(np.float64).copy(); pv = sc.prev.astype(np.float64).copy()
    st[:, 0:3] = st[:, 0:3] @ rz.T; st[:, 7:10] = st[:, 7:10] @ rz.T; st[:, 10:13] = st[:, 10:13] @ rz.T
    pv[:, 0:3] = pv[:, 0:3] @ rz.T; pv[:, 3:6] = pv[:, 3:6] @ rz.T
    x, y, z, w = (sc.state[:, 3 + i].astype(np.float64) for i in range(4))
    a, b, cc, d = qz                                                     # q' = qz (x) q
    st[:, 3] = d * x + a * w + b * z - cc * y
    st[:, 4] = d * y - a * z + b * w + cc * x
    st[:, 5] = d * z + a * y - b * x + cc * w
    st[:, 6] = d * w - a * x - b * y - cc * z
    f0, t0, _ = ho.step_wrench(sc.state, sc.prev, sc.params, sc.rho, sc.g, sc.dt)
    f1, t1, _ = ho.step_wrench(st, pv, sc.params, sc.rho, sc.g, sc.dt)
    assert np.abs(f1 - f0 @ rz.T).max() < 1e-6 * max(1.0, np.abs(f0).max())
    assert np.abs(t1 - t0 @ rz.T).max() < 1e-6 * max(1.0, np.abs(t0).max())


def test_clamp_never_increases_force():
    sc = _c4_sample()
    acc = ho.finite_difference_accel(sc.state, sc.prev, sc.dt)
    comps = ho.solve_components(sc.state, acc, sc.params, sc.rho, sc.g)
    raw = comps["buoyancy_force"] + comps["drag_force"] + comps["lift_force"] + comps["added_mass_force"]
    f, _, scale = ho.behavior_epilogue(sc.state[:, :3], comps, sc.params[:, 10])
    assert np.all(scale <= 1.0) and (scale < 1.0).sum() > 10
    assert np.all(np.linalg.norm(f, axis=1) <= np.linalg.norm(raw, axis=1) * (1 + 1e-15))
    assert np.all(np.linalg.norm(f, axis=1) <= sc.params[:, 10].astype(np.float64) * 500.0 * (1 + 1e-9))


def test_quadratic_drag_scales_with_speed_squared():
    """Above 0.2 m/s, with damping switched off, drag force is proportional to speed^2."""
    p = np.array([[1, 1, 1, 1.2, 0.8, 0.0, 0.0, 0.0, 0.0, 0.0, 500.0]])
    base = np.array([[0, 0, -5.0, 0, 0, 0, 1, 0.6, -0.3, 0.5, 0, 0, 0]])
    out = []
    for k in (1.0, 2.0, 4.0):
        s = base.copy(); s[:, 7:10] *= k
        out.append(ho.solve_components(s, np.zeros((1, 6)), p, 1025.0, 9.81)["drag_force"][0])
    assert np.allclose(out[1], 4 * out[0], rtol=1e-13) and np.allclose(out[2], 16 * out[0], rtol=1e-13)


def test_kinetic_energy_oracle():
    sc = _c4_sample(1024)
    tot, per = ho.kinetic_energy(sc.state, sc.params, rotational=False)
    assert tot == pytest.approx(float((0.5 * sc.params[:, 10].astype(np.float64)
                                       * (sc.state[:, 7:10].astype(np.float64) ** 2).sum(1)).sum()), rel=1e-14)
    tot_r, _ = ho.kinetic_energy(sc.state, sc.params, rotational=True)
    assert tot_r > tot
