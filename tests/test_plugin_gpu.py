"""The reference's plugin surface on the HIP engine, driven through an in-memory simulator host:
HydrodynamicsBehavior lifecycle (hydrodynamics_behavior.py:48-245) and the calculator
(warp_hydrodynamics_wrapper.py:79-132 / numba_hydrodynamics_wrapper.py:34-53)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import hydro_oracle as ho
from silver2_isaacsim_amd import behavior as hb
from silver2_isaacsim_amd import config as cfg
from silver2_isaacsim_amd.testing import FakeHost, FakeWorld
from silver2_isaacsim_amd.wrapper import HipHydrodynamicsWrapper

pytestmark = pytest.mark.gpu
from silver2_isaacsim_amd.testing import MAIN_SCENE, build_main_scene      # noqa: E402


def build_scene(batched, config_path=None, seed=0):
    """The 20 prims of silver2_isaac_sim.usd that carry the behavior."""
    return build_main_scene(batched, config_path, seed)


def oracle_wrench(world, prims, host, prev6, dt, semantics="numba"):
    n = len(prims)
    g = lambda p, k: host.get_exposed_variable(p, cfg.full_attr_name(k))       # noqa: E731
    params = np.array([[g(p, "xDimension"), g(p, "yDimension"), g(p, "zDimension"), g(p, "linearDragCoefficient"),
                        g(p, "angularDragCoefficient"), g(p, "linearDamping"), g(p, "angularDamping"),
                        g(p, "liftCoefficient"), g(p, "linearAddedMassCoefficient"),
                        g(p, "angularAddedMassCoefficient"), float(world.masses[i])] for i, p in enumerate(prims)], np.float32)
    pos = world.positions.cpu().numpy(); q = world.orientations.cpu().numpy(); vel = world.velocities.cpu().numpy()
    state = np.concatenate([pos, q[:, [1, 2, 3, 0]], vel], axis=1)
    f, t, _ = ho.step_wrench(state, prev6, params, 1025.0, 9.81, dt, semantics=semantics)
    return f, t, params


@pytest.mark.parametrize("batched", [True, "callbacks", False])
def test_lifecycle_and_wrench_parity(batched, native_built):
    hb.REGISTRY.clear()
    world, host, prims, behaviors = build_scene(batched)
    # on_init created the 12 attributes and applied globals + part overrides
    tib = prims[MAIN_SCENE.index("Tibia_3")]
    assert host.get_exposed_variable(tib, cfg.full_attr_name("linearDamping")) == 20.0
    assert host.get_exposed_variable(prims[0], cfg.full_attr_name("zDimension")) == 3.0
    for b in behaviors:
        b.on_play()
    dt = 1.0 / 60.0
    n = len(prims)
    prev = np.zeros((n, 6), np.float32)
    for step in range(3):
        host.step(dt)
        torch.cuda.synchronize()
        f_ref, t_ref, params = oracle_wrench(world, prims, host, prev, dt)
        got_f = np.stack([world.applied[p.path][0].cpu().numpy() for p in prims])
        got_t = np.stack([world.applied[p.path][1].cpu().numpy() for p in prims])
        err = ho.wrench_error(got_f, got_t, f_ref, t_ref, params, 1025.0, 9.81)
        assert err.max() <= 1e-5, (step, err.max())
        prev = world.velocities.cpu().numpy().copy()
        world.velocities += 0.01 * torch.randn_like(world.velocities)          # "PhysX" moves the bodies
    # N prims -> one batched launch + one apply call per step (or N in per-prim mode) ...
    assert world.apply_calls == (3 if batched else 3 * n)
    if batched:
        assert len(host.views) == 1 and len(host.views[0].paths) == n
    # ... and, in scene mode, ONE physics-step subscription for the whole group: host work per step is O(1) in prims
    # (the reference subscribes once per prim, :131-132, as "callbacks" mode and per-prim mode do)
    assert len(host._subs) == (1 if batched is True else n)
    assert host.callbacks_fired == (3 if batched is True else 3 * n)
    for b in behaviors:
        b.on_stop()
    assert not hb.REGISTRY._groups and not host._subs          # every subscription released (:240-245)
    host.step(dt)                                              # nothing fires, nothing is applied after on_stop
    assert world.apply_calls == (3 if batched else 3 * n)
    for b in behaviors:
        b.on_destroy()
    assert not prims[0].has(cfg.full_attr_name("gravity"))


@pytest.mark.parametrize("batched", [True, False])
def test_warp_semantics_subclass(batched, native_built, monkeypatch):
    """SEMANTICS = "warp" on the scripted class: the plugin then follows the calculator the reference script
    instantiates (hydrodynamics_behavior.py:155, warp_hydrodynamics.py:216-230) in its added-mass rotation."""
    hb.REGISTRY.clear()
    monkeypatch.setattr(hb.HydrodynamicsBehavior, "SEMANTICS", "warp")
    world, host, prims, behaviors = build_scene(batched, seed=5)
    for b in behaviors:
        b.on_play()
    dt = 1.0 / 60.0
    prev = np.zeros((len(prims), 6), np.float32)
    for step in range(2):                       # step 0: v_last = 0, i.e. large accelerations -> added mass matters
        host.step(dt)
        torch.cuda.synchronize()
        got_f = np.stack([world.applied[p.path][0].cpu().numpy() for p in prims])
        got_t = np.stack([world.applied[p.path][1].cpu().numpy() for p in prims])
        f_w, t_w, params = oracle_wrench(world, prims, host, prev, dt, "warp")
        f_n, t_n, _ = oracle_wrench(world, prims, host, prev, dt, "numba")
        assert ho.wrench_error(got_f, got_t, f_w, t_w, params, 1025.0, 9.81).max() <= 1e-5
        assert ho.wrench_error(got_f, got_t, f_n, t_n, params, 1025.0, 9.81).max() > 1e-3      # and it is a different model
        prev = world.velocities.cpu().numpy().copy()
        world.velocities += 0.01 * torch.randn_like(world.velocities)
    for b in behaviors:
        b.on_stop()
    hb.REGISTRY.clear()


def test_state_fetch_failure_skips_the_step(native_built):
    hb.REGISTRY.clear()
    world, host, prims, behaviors = build_scene(True)
    for b in behaviors:
        b.on_play()
    host.step(1 / 60)
    calls = world.apply_calls
    host.views[0].fail_next_fetch = True               # RuntimeError inside get_world_poses (:191-192)
    host.step(1 / 60)
    assert world.apply_calls == calls                  # skipped, nothing raised
    host.step(1 / 60)
    assert world.apply_calls == calls + 1
    host.step(0.0)                                     # dt <= 1e-6 guard (:139)
    assert world.apply_calls == calls + 1
    for b in behaviors:
        b.on_stop()


def test_missing_rigid_body_and_json_override(tmp_path, native_built):
    hb.REGISTRY.clear()
    data = cfg.default_config()
    data["parts"]["coxa"]["linearDamping"] = 11.5
    import json
    path = str(tmp_path / cfg.CONFIG_FILE_NAME)
    json.dump(data, open(path, "w"))
    world = FakeWorld("cuda:0"); host = FakeHost(world, path)
    prim = cfg.AttributeStore("Coxa_9"); world.add_body(prim.path, (0, 0, -5), (1, 0, 0, 0), [0] * 6, 0.45)
    b = hb.HydrodynamicsBehavior(prim, host); b.on_init()
    assert host.get_exposed_variable(prim, cfg.full_attr_name("linearDamping")) == 11.5
    ghost = cfg.AttributeStore("Decor", rigid_body=False)
    g = hb.HydrodynamicsBehavior(ghost, host); g.on_init(); g.on_play()
    host.step(1 / 60)                                   # no RigidBodyAPI: warned, never registered, no crash
    assert g._group is None and world.apply_calls == 0
    b.on_play(); host.step(1 / 60)
    assert world.apply_calls == 1
    b.on_stop()
    # a missing config file keeps the USD values (:79-81)
    host2 = FakeHost(world, str(tmp_path / "absent.json"))
    p2 = cfg.AttributeStore("Coxa_1"); b2 = hb.HydrodynamicsBehavior(p2, host2); b2.on_init()
    assert host2.get_exposed_variable(p2, cfg.full_attr_name("linearDamping")) == 300.0


def test_calculator_surface(native_built):
    """Same ctor keywords / method as the reference calculators; values = the reference's own outputs."""
    fx = load_golden("kat")
    names = [str(x) for x in fx["names"]]
    for i, nm in enumerate(names):
        p = fx["params"][i]
        w = HipHydrodynamicsWrapper(width=p[0], depth=p[1], height=p[2], linear_drag_coefficient=p[3],
                                    angular_drag_coefficient=p[4], linear_damping=p[5], angular_damping=p[6],
                                    water_density=1025.0, gravity=9.81, linear_mass_coeff=p[8],
                                    angular_mass_coeff=p[9], lift_coefficient=p[7], device="cuda:0")
        s, a = fx["state"][i].astype(np.float32), fx["accel"][i].astype(np.float32)
        out = w.calculate_hydrodynamic_forces(s[0:3], s[3:7], s[7:10], s[10:13], a[0:3], a[3:6])
        assert len(out) == 8 and all(o.shape == (1, 3) and o.dtype == torch.float32 and o.is_cuda for o in out)
        ref = fx["components"][i]
        for k in range(6):
            got = out[k].cpu().numpy()[0]
            assert np.abs(got - ref[k]).max() <= 2e-6 * max(1.0, np.abs(ref[k]).max()), (nm, k)
        assert np.abs(out[6].cpu().numpy()[0] - ref[6]).max() < 1e-5 and np.abs(out[7].cpu().numpy()[0] - ref[7]).max() < 1e-5
        assert float(w.sub_ratio[0]) == pytest.approx(fx["ratio"][i], abs=2e-7)
        w.close()


def test_calculator_batched_and_fused(native_built):
    fx = load_golden("c2")
    n = 512
    p = fx["params"][:n]
    w = HipHydrodynamicsWrapper(p[:, 0], p[:, 1], p[:, 2], p[:, 3], p[:, 4], p[:, 5], p[:, 6], 1025.0, 9.81,
                                p[:, 8], p[:, 9], p[:, 7], device="cuda:0", mass=p[:, 10])
    st = torch.from_numpy(fx["state"][:n]).cuda()
    dt = float(fx["dt"])
    w.engine.set_prev_velocity(fx["prev"][:n])
    F, T = w.calculate_wrench(st[:, 0:3], st[:, 3:7], st[:, 7:10], st[:, 10:13], dt)
    rf, rt, _ = ho.step_wrench(fx["state"][:n], fx["prev"][:n], p, 1025.0, 9.81, dt)
    assert ho.wrench_error(F.cpu().numpy(), T.cpu().numpy(), rf, rt, p, 1025.0, 9.81).max() <= 1e-5
    buf = F.data_ptr()
    F2, _ = w.calculate_wrench(st[:, 0:3], st[:, 3:7], st[:, 7:10], st[:, 10:13], dt)
    assert F2.data_ptr() == buf                        # wrapper-owned, reused buffers (warp wrapper :123-132)
    with pytest.raises(ValueError):
        HipHydrodynamicsWrapper(1, 1, 1, 1.2, 0.8, 300, 150, [1025.0, 1000.0], 9.81, 0.05, 0.02, 1.0)
    w.close()


def test_headless_buoy_demo(tmp_path, native_built):
    """The reference's validation demo end to end: USD table -> 20 behaviors -> one launch per step ->
    RTF meter + velocity CSV; the buoy settles at its float height."""
    import csv
    import importlib.util
    import os
    from conftest import REPO
    hb.REGISTRY.clear()
    spec = importlib.util.spec_from_file_location("buoy_demo", os.path.join(REPO, "examples", "buoy_bobbing_headless.py"))
    demo = importlib.util.module_from_spec(spec); spec.loader.exec_module(demo)
    out = demo.main(["--steps", "900", "--out", str(tmp_path)])
    z = out["z"]
    float_height = 1.5 - demo.BUOY_MASS / 1025.0                 # centre height of a 1x1x3 m box floating upright
    assert z[0] > float_height and abs(z[-1] - float_height) < 0.05 and z.min() > 0.0
    assert out["apply_calls"] == 900                             # 20 prims, one batched apply per step
    assert out["stats"]["physics_steps"] == 900 and out["stats"]["rtf"] > 1.0
    rows = list(csv.reader(open(out["csv"])))
    assert len(rows) == 901 and rows[0][1] == "z_position" and float(rows[-1][1]) == pytest.approx(z[-1], abs=1e-5)


def test_calculator_is_one_launch_and_matches_the_soa_component_entry(native_built):
    """calculate_hydrodynamic_forces goes through hydro_step_components_aos: same bits as the plain-SoA
    component entry, on ragged sizes; and a per-call latency figure for the per-prim flow (N = 1)."""
    import time
    from silver2_isaacsim_amd import scenes
    from silver2_isaacsim_amd.engine import HydroEngine
    fx = load_golden("c4")
    n = 777
    p = fx["params"][:n]
    w = HipHydrodynamicsWrapper(p[:, 0], p[:, 1], p[:, 2], p[:, 3], p[:, 4], p[:, 5], p[:, 6], 1025.0, 9.81,
                                p[:, 8], p[:, 9], p[:, 7], device="cuda:0")
    st = torch.from_numpy(fx["state"][:n]).cuda()
    acc = ((fx["state"][:n, 7:13].astype(np.float64) - fx["prev"][:n].astype(np.float64)) / float(fx["dt"])).astype(np.float32)
    a = torch.from_numpy(acc).cuda()
    out = w.calculate_hydrodynamic_forces(st[:, 0:3], st[:, 3:7], st[:, 7:10], st[:, 10:13], a[:, 0:3], a[:, 3:6])
    eng = HydroEngine(n, "cuda:0"); eng.set_params(p)
    comps, ratio = eng.step_components(torch.from_numpy(scenes.to_soa(fx["state"][:n])).cuda(), torch.from_numpy(scenes.to_soa(acc)).cuda())
    torch.cuda.synchronize()
    ref = comps.cpu().numpy().T.reshape(n, 8, 3)
    for k in range(8):
        assert np.array_equal(out[k].cpu().numpy(), ref[:, k, :]), k
    assert np.array_equal(w.sub_ratio.cpu().numpy(), ratio.cpu().numpy())
    eng.close(); w.close()
    # the reference's per-prim flow: one body per call
    w1 = HipHydrodynamicsWrapper(1, 1, 1, 1.2, 0.8, 300, 150, 1025.0, 9.81, 0.05, 0.02, 1.0, device="cuda:0")
    args = [st[:1, 0:3].contiguous(), st[:1, 3:7].contiguous(), st[:1, 7:10].contiguous(), st[:1, 10:13].contiguous(),
            a[:1, 0:3].contiguous(), a[:1, 3:6].contiguous()]
    for _ in range(50):
        w1.calculate_hydrodynamic_forces(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        w1.calculate_hydrodynamic_forces(*args)
    torch.cuda.synchronize()
    print(f"calculate_hydrodynamic_forces, N=1: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per call")
    w1.close()


def test_prepared_aos_step_and_view_pointer_changes(native_built):
    """engine.prepare_step_wrench_aos == step_wrench_aos bit for bit; the plugin's stepper re-prepares when the view
    hands out different buffers, converts what is not float32 / contiguous / on the device, and gives the same wrench
    either way (hydrodynamics_behavior.py:178-194: whatever the tensor API returns is used as it comes)."""
    from silver2_isaacsim_amd.behavior import _AosStepper
    from silver2_isaacsim_amd.engine import HydroEngine
    fx = load_golden("c4")
    n, dt = 500, float(fx["dt"])
    dev = torch.device("cuda:0")
    st = fx["state"][:n]
    pos = torch.from_numpy(np.ascontiguousarray(st[:, 0:3])).to(dev)
    quat = torch.from_numpy(np.ascontiguousarray(st[:, [6, 3, 4, 5]])).to(dev)
    vel = torch.from_numpy(np.ascontiguousarray(st[:, 7:13])).to(dev)

    def engine():
        e = HydroEngine(n, dev, float(fx["rho"]), float(fx["g"]))
        e.set_params(fx["params"][:n]); e.set_prev_velocity(fx["prev"][:n])
        return e
    a, b, c = engine(), engine(), engine()
    F0, T0 = a.step_wrench_aos(pos, quat, vel, dt)
    step = b.prepare_step_wrench_aos(pos, quat, vel)
    F1, T1 = step(dt)
    torch.cuda.synchronize()
    assert torch.equal(F0, F1) and torch.equal(T0, T1)
    with pytest.raises(ValueError):
        b.prepare_step_wrench_aos(pos.double(), quat, vel)
    # the plugin's stepper: float64 CPU tensors (converted every step, never cached), then device buffers (prepared once)
    F = torch.empty((n, 3), device=dev); T = torch.empty((n, 3), device=dev)
    stepper = _AosStepper(c, F, T)
    used = stepper(pos.cpu().double(), quat.cpu().double(), vel.cpu().double(), dt)
    torch.cuda.synchronize()
    assert used.device == dev and used.dtype == torch.float32 and stepper._key is None
    assert torch.equal(F, F0) and torch.equal(T, T0)
    c.set_prev_velocity(fx["prev"][:n])
    assert stepper(pos, quat, vel, dt) is pos and stepper._key is not None
    prepared = stepper._step
    c.set_prev_velocity(fx["prev"][:n])
    stepper(pos, quat, vel, dt)
    assert stepper._step is prepared                               # same buffers: no new preparation
    pos2 = pos.clone()
    c.set_prev_velocity(fx["prev"][:n])
    stepper(pos2, quat, vel, dt)
    torch.cuda.synchronize()
    assert stepper._step is not prepared and torch.equal(F, F0) and torch.equal(T, T0)
    for e in (a, b, c):
        e.close()
    with pytest.raises(Exception):
        step(dt)                                                   # the engine behind a prepared call is gone


def test_every_mode_gives_the_same_bits(native_built):
    """Scene-level subscription, per-prim callbacks and per-prim engines apply bit-identical wrenches."""
    got = {}
    for mode in (True, "callbacks", False):
        hb.REGISTRY.clear()
        world, host, prims, behaviors = build_scene(mode, seed=11)
        for b in behaviors:
            b.on_play()
        gen = torch.Generator(device="cpu").manual_seed(1)
        for _ in range(4):
            host.step(1 / 60)
            world.velocities += (0.01 * torch.randn(world.velocities.shape, generator=gen)).to(world.device)
        torch.cuda.synchronize()
        got[mode] = (torch.stack([world.applied[p.path][0] for p in prims]).cpu(), torch.stack([world.applied[p.path][1] for p in prims]).cpu())
        for b in behaviors:
            b.on_stop()
    for mode in ("callbacks", False):
        assert torch.equal(got[True][0], got[mode][0]) and torch.equal(got[True][1], got[mode][1]), mode
    hb.REGISTRY.clear()


def test_lifecycle_mirrors_the_kit_calls_of_the_reference(native_built):
    """on_init: SimulationContext(backend="torch") first (:50-51), exposed variables (:68), then the Property-window
    rebuild (:70); on_destroy: variables removed, rebuild again (:124-126)."""
    hb.REGISTRY.clear()
    world = FakeWorld("cuda:0"); host = FakeHost(world)
    prim = cfg.AttributeStore("Femur_2"); world.add_body(prim.path, (0, 0, -5), (1, 0, 0, 0), [0] * 6, 0.75)
    b = hb.HydrodynamicsBehavior(prim, host); b.on_init()
    assert host.lifecycle == ["simulation_context(torch)", "create_exposed_variables", "request_rebuild"]
    assert b._sim_context is host
    b.on_play(); host.step(1 / 60); b.on_stop(); b.on_destroy()
    assert host.lifecycle[3:] == ["remove_exposed_variables", "request_rebuild"]


@pytest.mark.parametrize("batched", [True, False])
def test_engine_errors_surface_instead_of_skipping_the_step(batched, native_built, caplog):
    """Only the state FETCH is guarded (hydrodynamics_behavior.py:177-192).  An engine error - here a view tensor the
    C ABI refuses (not 16-byte aligned), then a closed engine - propagates out of the physics-step callback; it is not
    swallowed as "skip this step" (HydroError is a RuntimeError, and RuntimeError is one of the guarded fetch errors)."""
    from silver2_isaacsim_amd.engine import HydroError
    hb.REGISTRY.clear()
    world, host, prims, behaviors = build_scene(batched)
    for b in behaviors:
        b.on_play()
    host.step(1 / 60)
    calls = world.apply_calls
    view = host.views[0]
    n = len(view.paths)
    base = torch.empty(n * 3 + 1, device=world.device)
    view._pos = base[1:].view(n, 3)                       # contiguous float32 on the device, but 4 bytes off alignment
    with caplog.at_level("ERROR", logger="silver2_isaacsim_amd"):
        with pytest.raises(HydroError, match="16-byte aligned"):
            host.step(1 / 60)
    assert world.apply_calls == calls                      # nothing applied for the failed step ...
    if batched:
        assert sum("refused the step" in r.message for r in caplog.records) == 1
    view._pos = torch.empty((n, 3), device=world.device)
    host.step(1 / 60)                                      # ... and the plugin works again once the input is sane
    assert world.apply_calls == calls + (1 if batched else len(prims))
    (behaviors[0]._group.engine if batched else behaviors[0]._engine).close()
    with pytest.raises(HydroError, match="closed"):
        host.step(1 / 60)
    hb.REGISTRY.clear()


@pytest.mark.parametrize("buffers", ["fresh", "strided", "numpy"])
def test_views_that_do_not_hand_out_stable_torch_buffers(buffers, native_built, caplog):
    """Nothing pins `get_world_poses(clone=False)` to the same contiguous device tensors every step
    (hydrodynamics_behavior.py:178-194).  Fresh tensors per step, non-contiguous views and - a view created without the
    torch backend - NumPy arrays all give the oracle's wrench; NumPy input is announced ONCE, never skipped silently."""
    hb.REGISTRY.clear()
    hb._warned_non_torch = False
    world, host, prims, behaviors = build_main_scene(True, None, 3, view_buffers=buffers)
    for b in behaviors:
        b.on_play()
    dt = 1.0 / 60.0
    prev = np.zeros((len(prims), 6), np.float32)
    with caplog.at_level("WARNING", logger="silver2_isaacsim_amd"):
        for step in range(4):
            host.step(dt)
            torch.cuda.synchronize()
            f_ref, t_ref, params = oracle_wrench(world, prims, host, prev, dt)
            got_f = np.stack([world.applied[p.path][0].cpu().numpy() for p in prims])
            got_t = np.stack([world.applied[p.path][1].cpu().numpy() for p in prims])
            assert ho.wrench_error(got_f, got_t, f_ref, t_ref, params, 1025.0, 9.81).max() <= 1e-5, step
            prev = world.velocities.cpu().numpy().copy()
            world.velocities += 0.01 * torch.randn_like(world.velocities)
    assert world.apply_calls == 4
    stepper = behaviors[0]._group._stepper
    assert stepper.prepared == 4                                   # re-prepared every step: new buffers, new launch arguments
    assert buffers == "fresh" or stepper._key is None              # converted copies are never remembered
    warned = [r for r in caplog.records if "not torch tensors" in r.message]
    assert len(warned) == (1 if buffers == "numpy" else 0)
    for b in behaviors:
        b.on_stop()
    hb.REGISTRY.clear()


def test_config3_through_the_plugin_at_full_scale(native_built):
    """BASELINE config 3 through `HydrodynamicsBehavior`: 19 456 prims, each with its own behavior instance, ONE
    physics-step subscription and ONE launch per step; host time per step stays under 100 us (it does not depend on
    the number of prims), and the applied wrench has the bits of the engine's own array-of-structs entry."""
    import time
    from silver2_isaacsim_amd.engine import HydroEngine
    from silver2_isaacsim_amd.testing import build_c3_scene
    hb.REGISTRY.clear()
    world, host, prims, behaviors, sc = build_c3_scene(1024)
    assert len(prims) == 19456
    for b in behaviors:
        b.on_play()
    assert len(host._subs) == 1
    dt = sc.dt
    # step 1 against the engine driven directly with the same tensors (previous velocity 0 on the first step, :196-198)
    # (the plugin reads water density and gravity back from Float - 32-bit - USD attributes, as the reference does:
    # its g is float32(9.81) = 9.8100004196167, and the engine driven directly has to use the same number)
    grp = behaviors[0]._group
    assert (grp.rho, grp.g) == (1025.0, float(np.float32(9.81)))
    eng = HydroEngine(sc.n, "cuda:0", grp.rho, grp.g); eng.set_params(sc.params)
    F, T = eng.step_wrench_aos(world.positions.clone(), world.orientations.clone(), world.velocities.clone(), dt)
    host.step(dt)
    torch.cuda.synchronize()
    view = host.views[0]
    assert torch.equal(view._force, F) and torch.equal(view._torque, T)
    eng.close()
    for _ in range(200):
        host.step(dt)
    torch.cuda.synchronize()
    per_step = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(400):
            host.step(dt)
        torch.cuda.synchronize()
        per_step.append((time.perf_counter() - t0) / 400 * 1e6)
    print(f"plugin, 19 456 prims, scene mode: {sorted(per_step)[2]:.1f} us of host time per physics step (runs: {per_step})")
    assert sorted(per_step)[2] < 100.0
    assert host.callbacks_fired == 1 + 200 + 2000 and world.apply_calls == host.callbacks_fired
    for b in behaviors:
        b.on_stop()
    assert not hb.REGISTRY._groups and not host._subs


@pytest.mark.parametrize("mode", [True, "callbacks"])
def test_membership_changes_between_steps(mode, native_built):
    """Prims stop and play again while the others keep stepping: the batch is rebuilt with the members of the moment
    (no stale or duplicate rows), each member's wrench is the oracle's for ITS state - members that stay keep their own
    previous-step velocity across the rebuild, as the reference's per-prim `_last_*_velocity` do (:196-198,237-238), a prim
    that comes back starts from zero - and the subscription count follows."""
    hb.REGISTRY.clear()
    world, host, prims, behaviors = build_scene(mode, seed=7)
    for b in behaviors:
        b.on_play()
    dt = 1.0 / 60.0
    host.step(dt); torch.cuda.synchronize()
    prev = world.velocities.cpu().numpy().copy()
    gone = [3, 9, 17]
    for k in gone:
        behaviors[k].on_stop()
    assert len(host._subs) == (1 if mode is True else len(prims) - len(gone))

    def check(active, prev_rows):
        f_ref, t_ref, params = oracle_wrench(world, prims, host, prev_rows, dt)
        got_f = np.stack([world.applied[prims[k].path][0].cpu().numpy() for k in active])
        got_t = np.stack([world.applied[prims[k].path][1].cpu().numpy() for k in active])
        err = ho.wrench_error(got_f, got_t, f_ref[active], t_ref[active], params[active], 1025.0, 9.81)
        assert err.max() <= 1e-5, err.max()
    active = [k for k in range(len(prims)) if k not in gone]
    world.velocities += 0.01 * torch.randn_like(world.velocities)
    host.step(dt); torch.cuda.synchronize()
    assert len(host.views[-1].paths) == len(active)
    if mode is True or mode == "callbacks":
        check(active, prev)                                           # the batch was rebuilt; nobody's finite difference was reset
    prev = world.velocities.cpu().numpy().copy()
    behaviors[9].on_play()                                           # one comes back
    active = sorted(active + [9])
    world.velocities += 0.01 * torch.randn_like(world.velocities)
    host.step(dt); torch.cuda.synchronize()
    assert len(host.views[-1].paths) == len(active) and len(set(host.views[-1].paths)) == len(active)
    prev[9] = 0.0                                                     # ... except the newcomer's: it starts from zero
    check(active, prev)
    prev = world.velocities.cpu().numpy().copy()
    world.velocities += 0.01 * torch.randn_like(world.velocities)
    host.step(dt); torch.cuda.synchronize()
    check(active, prev)                                               # ... and from then on the finite difference
    for b in behaviors:
        b.on_stop()
    assert not hb.REGISTRY._groups and not host._subs
