#!/usr/bin/env python3
"""BASELINE config 3 without Isaac Sim: the SILVER2 hexapod (chassis + 18 leg links, parameters of the
reference's hydrodynamics_config.json) in `--envs` independent environments, every link a free rigid
body, stepped in closed loop on the device: one kernel per physics step (wrench + integrator, drag taken
implicitly - the 0.45 kg links at 120 Hz are outside the explicit scheme's stability bound), 64 steps
per HIP-graph replay.  Prints the real-time factor the way the reference's benchmark_rtf.py defines it
and the global kinetic energy before / after (the drag has to dissipate it).

    python examples/silver2_envs_headless.py --envs 1024 --steps 2048
    python examples/silver2_envs_headless.py --envs 1024 --steps 2048 --resident
    python examples/silver2_envs_headless.py --envs 1024 --steps 2048 --through-plugin

`--resident`: 64 physics steps per LAUNCH instead of per graph replay - the links are independent bodies, so each is
carried through the 64 steps in registers (hydro_step_fused_tiled_multi): same bits, no memory traffic and no launch
between the steps, about 2.7x the real-time factor at this size.

`--through-plugin` drives the same 19 x envs bodies through the PLUGIN surface instead: one `HydrodynamicsBehavior`
instance per prim on the in-memory host (silver2_isaacsim_amd.testing), one physics-step subscription for the group,
one hydro_step_wrench_aos launch and one apply per step; the bodies do not move (the host has no PhysX), what is
reported is the host time per physics step and the real-time factor it allows.
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from silver2_isaacsim_amd import scenes                         # noqa: E402
from silver2_isaacsim_amd.simulate import ClosedLoopSim         # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--through-plugin", action="store_true")
    ap.add_argument("--resident", action="store_true")
    args = ap.parse_args(argv)
    if args.through_plugin:
        return through_plugin(args)
    sc = scenes.scene_c3(envs=args.envs)
    sim = ClosedLoopSim(sc, device=args.device, fused=True, implicit_drag=True)
    ke0 = sim.kinetic_energy(rotational=True)
    stats = sim.measure_rtf(args.steps, graph_steps=64, resident=args.resident)
    ke1 = sim.kinetic_energy(rotational=True)
    state = sim.state()
    out = {"bodies": sc.n, "envs": args.envs, "dt": sc.dt, "resident": args.resident, **stats,
           "kinetic_energy_J": {"before": [float(x) for x in ke0], "after": [float(x) for x in ke1]},
           "finite": bool((state == state).all()), "deepest_z": float(state[:, 2].min()), "highest_z": float(state[:, 2].max())}
    sim.close()
    print(json.dumps(out))
    return out


def through_plugin(args):
    import time
    import torch
    from silver2_isaacsim_amd import behavior as hb
    from silver2_isaacsim_amd.testing import build_c3_scene
    hb.REGISTRY.clear()
    world, host, prims, behaviors, sc = build_c3_scene(args.envs, device=args.device)
    for b in behaviors:
        b.on_play()
    for _ in range(100):
        host.step(sc.dt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        host.step(sc.dt)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    out = {"bodies": len(prims), "envs": args.envs, "dt": sc.dt, "mode": "HydrodynamicsBehavior x %d on the in-memory host" % len(prims),
           "physics_steps": args.steps, "wall_time_s": wall, "us_per_physics_step": wall / args.steps * 1e6,
           "rtf": args.steps * sc.dt / wall, "physics_step_subscriptions": len(host._subs), "apply_calls": world.apply_calls}
    for b in behaviors:
        b.on_stop()
    hb.REGISTRY.clear()
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
