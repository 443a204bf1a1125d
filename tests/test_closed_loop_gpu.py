"""Closed-loop device simulation (wrench + integrator, HIP-graph replay) - SURVEY.md 8f row 2."""
import numpy as np
import pytest
import torch

from conftest import load_golden
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
    assert np.abs(z - ref).max() < 2e-3, np.abs(z - ref).max()
    assert abs(z[-1] - (0.5 - float(fx["mass"]) / float(fx["rho"]))) < 3e-3
    st = sim.state()[0]
    assert np.abs(st[3:6]).max() < 1e-4 and abs(st[6] - 1) < 1e-6      # the cube stays upright
    sim.close()


def test_rtf_report(native_built):
    sim = ClosedLoopSim(scenes.scene_c2())
    r = sim.measure_rtf(2048, graph_steps=64)
    assert r["physics_steps"] == 2048 and r["sim_time_s"] == pytest.approx(2048 * sim.dt)
    assert r["rtf"] > 50 and r["body_steps_per_s"] > 1e8          # 4 096 buoys at 60 Hz: far beyond real time
    ke = sim.kinetic_energy()
    assert np.isfinite(ke).all() and ke[0] >= 0
    print("C2 closed loop:", r)
    sim.close()
