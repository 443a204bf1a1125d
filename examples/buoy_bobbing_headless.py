#!/usr/bin/env python3
"""Headless version of the reference's validation demo (README.md:51-56 there: the OBSEA buoy
bobbing at the surface, SILVER2 resting on the seabed), on an MI355X and without Isaac Sim.

  * per-prim parameters come from the scene's USD file (`--usd path/to/silver2_isaac_sim.usd`, decoded
    by `usd_crate.py`) or, without one, from the table committed in tests/golden/;
  * one `HydrodynamicsBehavior` per prim, exactly as Kit would instantiate them, on the in-memory
    simulator host; all 20 prims are evaluated by ONE kernel launch per physics step;
  * `BenchmarkRtf` and `LogVelocity` (the reference's two telemetry scripts) run beside them;
  * a semi-implicit Euler point mass stands in for PhysX for the buoy; the robot links are held in
    place (they are articulated and constrained in the real scene).

    python examples/buoy_bobbing_headless.py --steps 1800 --out /tmp/demo
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from silver2_isaacsim_amd import behavior as hb                      # noqa: E402
from silver2_isaacsim_amd import config as cfg                       # noqa: E402
from silver2_isaacsim_amd import usd_crate                           # noqa: E402
from silver2_isaacsim_amd.telemetry import BenchmarkRtf, LogVelocity  # noqa: E402
from silver2_isaacsim_amd.testing import FakeHost, FakeWorld         # noqa: E402

BUOY = "/World/Environment/Obsea_Buoy"
ROBOT_ORIGIN = (2.0, 10.7, -18.44)            # /World/SILVER2.xformOp:translate in the main scene
BUOY_MASS = 700.0                             # the scene derives it from density; any floating value will do


def load_table(usd_path):
    if usd_path:
        return usd_crate.hydrodynamics_table(usd_path)
    with open(os.path.join(REPO, "tests", "golden", "usd_hydrodynamics_tables.json")) as f:
        return json.load(f)["silver2_isaac_sim.usd"]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--usd", default=None)
    ap.add_argument("--steps", type=int, default=1800)
    ap.add_argument("--out", default=".")
    args = ap.parse_args(argv)

    table = load_table(args.usd)
    rate = float(table["__scene__"].get("timeStepsPerSecond", 60))
    dt = 1.0 / rate
    world = FakeWorld("cuda:0")
    host = FakeHost(world)
    rng = np.random.default_rng(0)
    behaviors, prims = [], {}
    for path, attrs in table.items():
        if path == "__scene__":
            continue
        name = path.rsplit("/", 1)[-1]
        prim = cfg.AttributeStore(name, path)
        if path == BUOY:
            pos = tuple(attrs.get("translate", (-7.0, 40.0, 0.596)))
            pos = (pos[0], pos[1], pos[2] + 0.4)                  # released 0.4 m above its mooring height
            mass = BUOY_MASS
        else:
            pos = tuple(np.array(ROBOT_ORIGIN) + rng.uniform(-0.3, 0.3, 3))
            mass = attrs.get("mass", 1.0)
        world.add_body(path, pos, (1.0, 0.0, 0.0, 0.0), [0.0] * 6, mass)
        b = hb.HydrodynamicsBehavior(prim, host)
        b.on_init()                                               # creates the 12 attributes, applies the JSON table
        for k in cfg.SCHEMA_NAMES:                                # authored USD values win where no JSON part matched
            if k in attrs and cfg.match_part(name, cfg.PART_TABLE) is None:
                host.set_exposed_variable(prim, cfg.full_attr_name(k), attrs[k])
        behaviors.append(b); prims[path] = prim

    rtf = BenchmarkRtf(host, out=lambda s: print(s, flush=True))
    logger = LogVelocity(prims[BUOY], host, directory=args.out)
    rtf.on_init(); logger.on_init()
    for b in behaviors:
        b.on_play()
    rtf.on_play(); logger.on_play()

    i_buoy = world.index(BUOY)
    g = torch.tensor([0.0, 0.0, -9.81], device=world.device)
    z_hist = []
    for k in range(args.steps):
        host.step(dt)                                             # every prim's callback; one batched launch
        force, _torque = world.applied[BUOY]
        v = world.velocities[i_buoy, 0:3] + dt * (force / BUOY_MASS + g)      # "PhysX": buoy only
        world.velocities[i_buoy, 0:3] = v
        world.positions[i_buoy] = world.positions[i_buoy] + dt * v
        logger.on_update(k * dt, dt)
        z_hist.append(float(world.positions[i_buoy, 2]))
    stats = rtf.on_stop()
    logger.on_stop()
    for b in behaviors:
        b.on_stop()
    z = np.array(z_hist)
    print(f"buoy z: start {z[0]:.3f} m, min {z.min():.3f}, max {z.max():.3f}, last {z[-1]:.3f} "
          f"(analytic float height for {BUOY_MASS:.0f} kg: {1.5 - BUOY_MASS / 1025.0:.3f} m)")
    return {"z": z, "stats": stats, "csv": os.path.join(args.out, "velocity_log.csv"), "apply_calls": world.apply_calls}


if __name__ == "__main__":
    main()
