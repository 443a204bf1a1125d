#!/usr/bin/env python3
"""BASELINE config 3 without Isaac Sim: the SILVER2 hexapod (chassis + 18 leg links, parameters of the
reference's hydrodynamics_config.json) in `--envs` independent environments, every link a free rigid
body, stepped in closed loop on the device: one kernel per physics step (wrench + integrator, drag taken
implicitly - the 0.45 kg links at 120 Hz are outside the explicit scheme's stability bound), 64 steps
per HIP-graph replay.  Prints the real-time factor the way the reference's benchmark_rtf.py defines it
and the global kinetic energy before / after (the drag has to dissipate it).

    python examples/silver2_envs_headless.py --envs 1024 --steps 2048
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
    args = ap.parse_args(argv)
    sc = scenes.scene_c3(envs=args.envs)
    sim = ClosedLoopSim(sc, device=args.device, fused=True, implicit_drag=True)
    ke0 = sim.kinetic_energy(rotational=True)
    stats = sim.measure_rtf(args.steps, graph_steps=64)
    ke1 = sim.kinetic_energy(rotational=True)
    state = sim.state()
    out = {"bodies": sc.n, "envs": args.envs, "dt": sc.dt, **stats,
           "kinetic_energy_J": {"before": [float(x) for x in ke0], "after": [float(x) for x in ke1]},
           "finite": bool((state == state).all()), "deepest_z": float(state[:, 2].min()), "highest_z": float(state[:, 2].max())}
    sim.close()
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
