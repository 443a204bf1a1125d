"""Closed-loop device simulation (wrench + integrator, HIP-graph replay) - SURVEY.md 8f row 2."""
import os

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden
from silver2_isaacsim_amd import scenes
from silver2_isaacsim_amd.simulate import ClosedLoopSim

pytestmark = pytest.mark.gpu


def test_graph_replay_equals_eager_stepping(native_built):
    sc = scenes.scene_c2(n=1000)
    a, b = ClosedLoopSim(sc), ClosedLoopSim(sc)
    a.run_eager(192)
    b.run(192, graph_steps=64)                      # 3 replays of a 64-step graph
    assert b._graph is not None
    sa, sb = a.state(), b.state()
    assert np.isfinite(sa).all() and np.array_equal(sa, sb)
    b.run(70, graph_steps=64); a.run_eager(70)       # one replay + 6 eager steps
    assert np.array_equal(a.state(), b.state()) and a.steps_done == b.steps_done == 262
    with pytest.raises(ValueError):
        b.run(63, graph_steps=63)
    a.close(); b.close()


def test_graph_is_recaptured_when_the_ping_pong_phase_changes(native_built):
    """A captured graph hard-codes which buffer is current.  An odd number of resident launches (or of eager steps)
    swaps the buffers; a replay of the old graph would then step the stale one: run -> run_resident ->
    run, and measure_rtf(resident=True) followed by measure_rtf(resident=False), give the bits of eager stepping."""
    sc = scenes.scene_c2(n=1000)
    a, b = ClosedLoopSim(sc), ClosedLoopSim(sc)
    a.run_eager(192)
    b.run(64); b.run_resident(64); b.run(64)        # one resident launch in between: odd
    assert np.array_equal(a.state(), b.state()) and b.steps_done == 192
    b.run_eager(3); b.run(64); a.run_eager(67)       # odd eager remainder, then replays again
    assert np.array_equal(a.state(), b.state())
    c, d = ClosedLoopSim(sc), ClosedLoopSim(sc)
    c.measure_rtf(640, graph_steps=64, resident=True)            # 1 warm + 10 timed launches: odd
    c.measure_rtf(128, graph_steps=64, resident=False)
    d.run_eager(64 + 640 + 64 + 128)
    assert c.steps_done == d.steps_done and np.array_equal(c.state(), d.state())
    for s_ in (a, b, c, d):
        s_.close()


@pytest.mark.parametrize("name,coeff", [("c2", "f32"), ("c5", "f16")])
def test_fused_step_equals_wrench_then_integrate(name, coeff, native_built):
    """hydro_step_fused_tiled == hydro_step_wrench_tiled + hydro_integrate_tiled, bit for bit,
    including the optional wrench output; 40 closed-loop steps."""
    sc = scenes.scene_c2(n=3000) if name == "c2" else scenes.scene_c5(n=3000)
    a, b = ClosedLoopSim(sc, coeff_dtype=coeff, fused=True), ClosedLoopSim(sc, coeff_dtype=coeff, fused=False)
    w = a.engine.alloc_tiled(6, sc.n)
    a.engine.step_fused_tiled(a.cur, a.old, sc.n, sc.dt, wrench=w)     # one step by hand, with the wrench
    a.cur, a.old = a.old, a.cur
    b.run_eager(1)
    torch.cuda.synchronize()
    assert torch.equal(w, b.wrench) and np.array_equal(a.state(), b.state())
    steps = 40 if name == "c2" else 4       # the C5 population has light bodies the explicit integrator cannot hold
    a.run_eager(steps); b.run_eager(steps)
    sa, sb = a.state(), b.state()
    assert np.array_equal(sa, sb, equal_nan=True)
    a.close(); b.close()


def test_config1_single_buoy_on_the_gpu(native_built):
    """Config 1 (1 body, 10 000 steps): the fp32 device loop follows the trajectory generated with
    the reference's functions (tests/golden/c1_trajectory.npz) and settles at the same equilibrium."""
    fx = load_golden("c1_trajectory")
    sim = ClosedLoopSim(scenes.scene_c1())
    z = []
    for _ in range(100):
        sim.run(100, graph_steps=100)
        z.append(sim.state()[0, 2])
    z = np.array(z)
    ref = fx["z"][99::100]
    print(f"[config 1 on the GPU] max |z - z_ref| over 10 000 steps: {np.abs(z - ref).max():.3e} m; final offset from the "
          f"analytic equilibrium {abs(z[-1] - (0.5 - float(fx['mass']) / float(fx['rho']))):.3e} m")
    # wrench in fp64, state and integrator in fp32; the buoy stays upright, so the only rounding on the way is that of the
    # fp32 position / velocity updates of a 0.6 m oscillation: measured 6.6e-8 m over the 10 000 steps (was allowed 2e-3)
    assert np.abs(z - ref).max() < 1e-6, np.abs(z - ref).max()
    assert abs(z[-1] - (0.5 - float(fx["mass"]) / float(fx["rho"]))) < 3e-3
    st = sim.state()[0]
    assert np.abs(st[3:6]).max() < 1e-4 and abs(st[6] - 1) < 1e-6      # the cube stays upright
    # the same 10 000 steps as ten launches with the body resident in registers: the same state, bit for bit
    res = ClosedLoopSim(scenes.scene_c1())
    res.run_resident(10000, chunk=1000)
    assert np.array_equal(res.state(), sim.state())
    sim.close(); res.close()


def test_rtf_report(native_built):
    sim = ClosedLoopSim(scenes.scene_c2())
    r = sim.measure_rtf(2048, graph_steps=64)
    assert r["physics_steps"] == 2048 and r["sim_time_s"] == pytest.approx(2048 * sim.dt)
    assert r["rtf"] > 50 and r["body_steps_per_s"] > 1e8          # 4 096 buoys at 60 Hz: far beyond real time
    ke = sim.kinetic_energy()
    assert np.isfinite(ke).all() and ke[0] >= 0
    print("C2 closed loop:", r)
    sim.close()


def test_implicit_drag_agrees_with_explicit_where_both_are_stable(native_built):
    """Heavy buoys (damping * dt / m ~ 0.01): the linearly-implicit drag update is a first-order
    perturbation of the explicit one."""
    sc = scenes.scene_c2(n=512)
    a, b = ClosedLoopSim(sc, implicit_drag=True), ClosedLoopSim(sc, implicit_drag=False)
    a.run_eager(10); b.run_eager(10)
    sa, sb = a.state(), b.state()
    assert np.abs(sa[:, 0:3] - sb[:, 0:3]).max() < 1e-2            # after 10 steps: a small O(dt) difference (cm)
    assert np.abs(sa[:, 7:10] - sb[:, 7:10]).max() < 1e-1
    a.run_eager(110); b.run_eager(110)                             # 2 s: bobbing phases drift apart, both stay bounded
    sa, sb = a.state(), b.state()
    assert np.isfinite(sa).all() and np.isfinite(sb).all()
    assert np.abs(sa[:, 7:10]).max() < 5.0 and np.abs(sb[:, 7:10]).max() < 5.0
    a.close(); b.close()


def test_config3_links_need_and_get_the_implicit_update(native_built):
    """Config 3 bodies (0.45-18 kg links, damping 10-300 N s/m, 120 Hz): damping * dt / m reaches 5.5, the
    explicit integrator diverges, the implicit one settles to terminal sinking speed."""
    sc = scenes.scene_c3(envs=64)
    sim = ClosedLoopSim(sc, implicit_drag=True)
    sim.run(1200, graph_steps=100)
    st = sim.state()
    assert np.isfinite(st).all()
    assert np.abs(st[:, 7:10]).max() < 2.0 and np.abs(st[:, 10:13]).max() < 5.0
    assert np.abs(np.linalg.norm(st[:, 3:7], axis=1) - 1).max() < 1e-5
    ke = sim.kinetic_energy()
    assert np.isfinite(ke).all()
    r = sim.measure_rtf(2400, graph_steps=100)
    print("C3 (64 envs) closed loop, implicit drag:", r)
    sim.close()
    bad = ClosedLoopSim(sc, implicit_drag=False)
    bad.run(1200, graph_steps=100)
    sb = bad.state()
    assert (not np.isfinite(sb).all()) or np.abs(sb[:, 7:10]).max() > 50.0      # explicit: blows up
    bad.close()
    with pytest.raises(ValueError):
        ClosedLoopSim(sc, fused=False, implicit_drag=True)


def test_silver2_envs_example(native_built):
    """examples/silver2_envs_headless.py: BASELINE config 3 (19 links x envs) in closed loop from a HIP graph."""
    import importlib.util
    import os
    from conftest import REPO
    spec = importlib.util.spec_from_file_location("silver2_envs", os.path.join(REPO, "examples", "silver2_envs_headless.py"))
    demo = importlib.util.module_from_spec(spec); spec.loader.exec_module(demo)
    out = demo.main(["--envs", "256", "--steps", "512"])
    assert out["bodies"] == 19 * 256 and out["finite"] and out["physics_steps"] == 512
    assert out["rtf"] > 10.0                                            # 120 Hz scene, microseconds per step
    ke0, ke1 = out["kinetic_energy_J"]["before"], out["kinetic_energy_J"]["after"]
    assert ke1[1] < ke0[1]                                              # the angular drag dissipates the initial spin
    # 64 steps per launch, the links resident in registers: the same final state, bit for bit
    res = demo.main(["--envs", "256", "--steps", "512", "--resident"])
    assert res["resident"] and res["finite"] and res["kinetic_energy_J"]["after"] == ke1
    assert (res["deepest_z"], res["highest_z"]) == (out["deepest_z"], out["highest_z"])
    # the same bodies through the plugin surface: one behavior instance per prim, one subscription, one launch per step
    via = demo.main(["--envs", "64", "--steps", "200", "--through-plugin"])
    assert via["bodies"] == 19 * 64 and via["physics_step_subscriptions"] == 1 and via["apply_calls"] == 300
    assert via["rtf"] > 10.0


def test_checkpoint_resume_is_bit_exact(native_built):
    """SURVEY section 5 (checkpoint / resume): the only state of the path is the body state and the previous-step
    velocity.  64 steps, snapshot to the host, a NEW engine resumed from the snapshot for 64 more steps == 128
    uninterrupted steps, bit for bit."""
    sc = scenes.scene_c2(n=2048, seed=21)
    a = ClosedLoopSim(sc, fused=True)
    a.run(128, graph_steps=64)
    want = a.state()
    b = ClosedLoopSim(sc, fused=True)
    b.run(64, graph_steps=64)
    cur = b.state()
    b.synchronize()
    old = scenes.from_tiled(b.old.cpu().numpy(), sc.n)              # the state one step earlier: its velocities are "prev"
    a.close(); b.close()
    snap = scenes.Scene(sc.name, cur.astype(np.float32), old[:, 7:13].astype(np.float32), sc.params, sc.rho, sc.g, sc.dt,
                        sc.coeff_dtype, dict(sc.info))
    c = ClosedLoopSim(snap, fused=True)
    c.run(64, graph_steps=64)
    got = c.state()
    c.close()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name,coeff,implicit", [("c2", "f32", False), ("c2", "f16", False), ("c3", "f32", True), ("c5", "f16", True)])
def test_resident_multi_step_launch_equals_single_steps(name, coeff, implicit, native_built):
    """hydro_step_fused_tiled_multi: the bodies are independent, so one launch carries each of them through K steps in
    registers.  Same arithmetic in the same order: the state after 1, 2, 7 and 64 + 64 + 9 resident steps has the BITS of
    as many single-step launches, ragged size, explicit and implicit drag, and the two-buffer ping-pong (previous
    velocity written into the buffer being read) hands the right previous velocity to the next launch."""
    sc = {"c2": scenes.scene_c2, "c3": lambda n: scenes.scene_c3(envs=(n + 18) // 19), "c5": scenes.scene_c5}[name](3001)
    a, b = ClosedLoopSim(sc, coeff_dtype=coeff, implicit_drag=implicit), ClosedLoopSim(sc, coeff_dtype=coeff, implicit_drag=implicit)
    done = 0
    for k, chunk in ((1, 1), (2, 2), (7, 7), (137, 64)):
        a.run_eager(k)
        b.run_resident(k, chunk=chunk)
        done += k
        sa, sb = a.state(), b.state()
        assert a.steps_done == b.steps_done == done
        # (the C5 population has light bodies no fixed-step integrator holds for 100+ steps: they diverge the same way)
        assert (name == "c5" or np.isfinite(sa).all()) and np.array_equal(sa, sb, equal_nan=True), (name, k)
        # ... and the buffer that holds "previous" carries the velocity of the step before (what the next launch reads)
        assert np.array_equal(a.old[:, 7:13].cpu().numpy(), b.old[:, 7:13].cpu().numpy(), equal_nan=True)
    # mixing the two entries mid-run changes nothing either
    a.run(64, graph_steps=64); b.run_resident(32, chunk=32); b.run_eager(32)
    assert np.array_equal(a.state(), b.state(), equal_nan=True)
    with pytest.raises(ValueError):
        ClosedLoopSim(sc, fused=False).run_resident(4)
    a.close(); b.close()


def test_resident_loop_samples_the_kinetic_energy_and_refuses_bad_calls(native_built):
    sc = scenes.scene_c2(n=3000)
    ref, res = ClosedLoopSim(sc, ke_every=64), ClosedLoopSim(sc, ke_every=64)
    ref.run(192, graph_steps=64); res.run_resident(192, chunk=64)
    ref.synchronize(); res.synchronize()
    ref.monitor.collect(block=True); res.monitor.collect(block=True)
    assert [s for s, _ in res.monitor.samples] == [64, 128, 192]
    for (_, x), (_, y) in zip(ref.monitor.samples, res.monitor.samples):
        assert x[0] == y[0] and x[1] == y[1]                      # same state bits, same fixed-order reduction
    assert np.array_equal(ref.state(), res.state())
    with pytest.raises(ValueError):
        res.run_resident(60, chunk=48)                            # ke_every = 64 is not a multiple of 48
    e = res.engine
    from silver2_isaacsim_amd._native import HydroError
    with pytest.raises(HydroError, match="steps must be"):
        e.step_fused_tiled_multi(res.cur, res.old, sc.n, sc.dt, 0)
    with pytest.raises(HydroError, match="must not alias"):
        e.step_fused_tiled_multi(res.cur, res.old, sc.n, sc.dt, 4, state_out=res.cur)
    ref.close(); res.close()


def test_kinetic_energy_monitor_in_the_closed_loop(native_built):
    """SURVEY.md 8e on one GPU: every 64 steps (= one graph replay) the shard's kinetic energy is reduced on device on
    the step stream and carried to the host on a side stream; the step loop never waits for it.  The samples equal an
    fp64 host sum over the state at those steps, and the run's bits do not depend on the monitor being there."""
    sc = scenes.scene_c2(n=3000)
    plain, watched = ClosedLoopSim(sc), ClosedLoopSim(sc, ke_every=64)
    ref = ClosedLoopSim(sc)
    expect = []
    for _ in range(4):
        ref.run(64, graph_steps=64)
        st = ref.state().astype(np.float64)
        m = sc.params[:, 10].astype(np.float64)
        expect.append(float((0.5 * m * (st[:, 7:10] ** 2).sum(1)).sum()))
    plain.run(256, graph_steps=64)
    watched.run(256, graph_steps=64)
    watched.synchronize()
    got = watched.monitor.collect(block=True)
    assert [s for s, _ in watched.monitor.samples] == [64, 128, 192, 256] and len(got) <= 4
    for (step, ke), want in zip(watched.monitor.samples, expect):
        assert ke[0] == pytest.approx(want, rel=1e-12), step
        assert ke[1] > 0.0                                            # rotational part (box inertia)
    assert np.array_equal(plain.state(), watched.state())
    # without a process group the samples ride INSIDE the replayed graphs (two sampling graphs, ring slots 0 / 1): no host work
    assert watched.monitor.graph_capturable and len(watched._graph_sampling) == 2 and watched._captured_samples == 4
    # a sample every SECOND replay: the replays in between are of the plain graph (no sampling kernel at all)
    sparse = ClosedLoopSim(sc, ke_every=128)
    sparse.run(256, graph_steps=64)
    sparse.synchronize()
    sparse.monitor.collect(block=True)
    assert [s for s, _ in sparse.monitor.samples] == [128, 256] and sparse._captured_samples == 2
    assert sparse.monitor.samples[0][1][0] == pytest.approx(expect[1], rel=1e-12)
    assert sparse.monitor.samples[1][1][0] == pytest.approx(expect[3], rel=1e-12)
    assert np.array_equal(plain.state(), sparse.state())
    sparse.close()
    # the host-driven form on request (what gloo gets anyway): same samples
    driven = ClosedLoopSim(sc, ke_every=64, graph_resident_sampling=False)
    driven.run(256, graph_steps=64)
    driven.synchronize()
    driven.monitor.collect(block=True)
    assert driven._captured_samples == 0 and len(driven._graph_sampling) == 1
    assert [(s_, v) for s_, v in driven.monitor.samples] == [(s_, v) for s_, v in watched.monitor.samples]
    driven.close()
    # eager stepping samples at the same steps
    eager = ClosedLoopSim(sc, ke_every=64)
    eager.run_eager(130)
    eager.synchronize()
    eager.monitor.collect(block=True)
    assert [s for s, _ in eager.monitor.samples] == [64, 128]
    assert eager.monitor.samples[0][1][0] == pytest.approx(expect[0], rel=1e-12)
    for s in (plain, watched, ref, eager):
        s.close()


def test_graph_resident_sampling_takes_samples_without_host_work(native_built):
    """KineticEnergyMonitor.capture_sample: the rest of a sample's pipeline (the all-reduce over the ranks - none here - and the
    copy to pinned host memory) is captured into the caller's HIP graph behind the sampling step; a replay takes the sample,
    the host only records an event (`submit_captured`).  Two graphs alternate between two ring slots; `reserve` keeps a slot's
    previous sample from being overwritten before it was picked up.  Samples equal the fp64 host sum of the state they saw."""
    import torch
    from silver2_isaacsim_amd.engine import HydroEngine
    from silver2_isaacsim_amd.simulate import KineticEnergyMonitor
    dev = torch.device("cuda:0")
    sc = scenes.scene_c4(n=50000, seed=41)
    eng = HydroEngine(sc.n, dev, sc.rho, sc.g)
    eng.set_params(sc.params)
    states = []
    for k in range(3):                                    # three different states: a stale sample would show
        st = sc.state.copy(); st[:, 7:10] *= (1.0 + 0.25 * k)
        states.append(st)
    cur = torch.from_numpy(scenes.to_tiled(states[0])).to(dev)
    prev = torch.from_numpy(scenes.to_tiled(sc.prev)).to(dev)
    out = eng.alloc_tiled(6, sc.n)
    mon = KineticEnergyMonitor(eng, every=4)
    assert mon.graph_capturable                            # a GPU, no process group
    stream = torch.cuda.Stream(dev)
    graphs = []
    with torch.cuda.stream(stream):
        mon.warm_up(stream)
        for j in (0, 1):
            eng.step_wrench_tiled(cur, sc.n, sc.dt, out=out, prev=prev, ke_out=mon.slot_buffer(j), rotational=True)
        stream.synchronize()
        for j in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                for k in range(4):
                    eng.step_wrench_tiled(cur, sc.n, sc.dt, out=out, prev=prev,
                                          ke_out=mon.slot_buffer(j) if k == 3 else None, rotational=True)
                mon.capture_sample(j)
            graphs.append(g)
        want = []
        for r in range(6):                                 # six replays back to back, the state changes under them
            stream.synchronize()
            st = states[r % 3]
            cur.copy_(torch.from_numpy(scenes.to_tiled(st)).to(dev))
            want.append(scenes.kinetic_energy_fp64(st, sc.params))
            j = r % 2
            mon.reserve(j)
            graphs[j].replay()
            mon.submit_captured(4 * (r + 1), j, stream)
    mon.collect(block=True)
    assert [s for s, _ in mon.samples] == [4, 8, 12, 16, 20, 24] and mon.submitted == 6
    for (_, ke), w in zip(mon.samples, want):
        assert ke[0] == pytest.approx(w[0], rel=1e-12) and ke[1] == pytest.approx(w[1], rel=1e-12)
    eng.close()


def test_sharded_closed_loop_example_two_ranks(native_built):
    """examples/sharded_closed_loop.py as a user would run it (torchrun, one process per GPU) - here two ranks sharing this
    box's one GPU over gloo: the asynchronous global kinetic energy of the sharded closed loop equals the fp64 host sum over
    the gathered final state, and the sharded run reproduces the unsharded one bit for bit."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, HYDRO_DIST_BACKEND="gloo", HYDRO_EXAMPLE_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "examples", "sharded_closed_loop.py"), "--bodies", "20001", "--steps", "128"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ranks"] == 2 and d["bodies"] == 20001 and d["bodies_per_rank"] == 10001
    assert [k["step"] for k in d["kinetic_energy_J"]] == [64, 128]
    assert d["last_sample_rel_err_vs_host_fp64"] <= 1e-12 and d["sharded_equals_unsharded_bit_for_bit"] is True


def test_a_rank_that_never_joins_a_sample_is_a_timeout_not_a_hang(native_built):
    """the product has the deadline the bench has.  Two ranks of the sharded example on this box's GPU
    (gloo); rank 1 stalls before its first replay, so the all-reduce of rank 0's first kinetic-energy sample is never joined:
    rank 0 raises TimeoutError naming the step and the rank within `--timeout` and the job exits non-zero - no hang, no retry."""
    import socket
    import subprocess
    import sys
    import time
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, HYDRO_DIST_BACKEND="gloo", HYDRO_EXAMPLE_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "_stalled_rank_worker.py"), "--bodies", "20001", "--steps", "128",
           "--timeout", "5"]
    t0 = time.monotonic()
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode != 0 and time.monotonic() - t0 < 120
    assert "kinetic-energy sample of step 64 on rank 0" in res.stderr and "did not finish within 5 s" in res.stderr
    assert "warm-up pass 0" in res.stderr                           # (rank 1 stalls before its first run: the first collective is the warm-up's)
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]       # no result line from a run that did not finish


def test_captured_collective_is_opt_in_with_more_than_one_rank(native_built, monkeypatch):
    """Default of ClosedLoopSim's sampling under graph replays: inside the graph only where no other rank is involved; with a
    process group of more than one rank it is the host-driven side-stream pipeline unless asked for (simulate.py)."""
    from silver2_isaacsim_amd import distributed as hd
    from silver2_isaacsim_amd.simulate import ClosedLoopSim
    sc = scenes.scene_c2(n=4096, seed=3)
    alone = ClosedLoopSim(sc, ke_every=64)
    assert alone._graph_sampling_ok                                   # no process group: the sample rides in the graph
    alone.close()
    import torch.distributed as dist
    monkeypatch.setattr(hd, "_collectives_on", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda *a, **k: 8)
    for kw, env, want in (({}, None, False), ({"graph_resident_sampling": True}, None, True), ({}, "1", True), ({}, "0", False),
                          ({"graph_resident_sampling": False}, "1", False)):
        if env is None:
            monkeypatch.delenv("HYDRO_GRAPH_SAMPLING", raising=False)
        else:
            monkeypatch.setenv("HYDRO_GRAPH_SAMPLING", env)
        sim = ClosedLoopSim(sc, ke_every=64, **kw)                    # (the constructor is local: no collective in it)
        assert sim._wants_graph_sampling is want, (kw, env)
        sim.engine.close()
