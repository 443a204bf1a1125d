#!/usr/bin/env python3
"""GPU box: drive ONE auxiliary kernel for the profiler (rocprofv3 wants the program itself after `--`).
    python3 scripts/run_aux.py ke 1048576 [launches]        hydro_kinetic_energy_tiled, rotational, 4 rotating scenes
    python3 scripts/run_aux.py resident 1048576 [launches]  hydro_step_fused_tiled_multi, 64 steps per launch
    python3 scripts/run_aux.py batch 1048576 [launches]     hydro_step_wrench_tiled_batch, 4 scenes per launch, 2 groups
Prints one JSON line with the HIP-event time per launch."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from scripts import bench_extras  # noqa: E402
from silver2_isaacsim_amd.engine import HydroEngine  # noqa: E402
from silver2_isaacsim_amd.simulate import ClosedLoopSim  # noqa: E402

what, n = sys.argv[1], int(sys.argv[2])
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 400
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(dev)


def timed(fn, reps, warm=20):
    with torch.cuda.stream(stream):
        for k in range(warm):
            fn(k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for k in range(reps):
            fn(k)
        e1.record(stream)
        stream.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


if what == "ke":
    sets = 4 if n <= 1048576 else 2
    sc = bench.build_scene("c5", n, 5)
    reps = [bench.Replica(sc, "f16", dev, roll=r * 131071) for r in range(sets)]
    out = torch.zeros(2, dtype=torch.float64, device=dev)
    with torch.cuda.stream(stream):
        bench.spin_up(reps, stream, 0.5)
    us = timed(lambda k: reps[k % sets].engine.kinetic_energy(reps[k % sets].state, True, out=out), launches)
    print(json.dumps({"what": "hydro_kinetic_energy_tiled (rotational)", "n": n, "us_per_launch": us, "bytes_per_body": 56,
                      "frac_of_8TBs": n * 56 / (us * 1e-6) / 8e12}))
elif what == "resident":
    sim = ClosedLoopSim(bench.build_scene("c2", n, 17))
    sim.run_resident(64, 64)
    sim.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sim.stream):
        ev0.record(sim.stream)
        sim.run_resident(64 * launches, 64)
        ev1.record(sim.stream)
    sim.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / (64 * launches)
    print(json.dumps({"what": "hydro_step_fused_tiled_multi, 64 steps per launch", "n": n, "us_per_step": us,
                      "roofline": bench_extras.valu_roofline("resident closed loop, one step (", n, us)}))
    sim.close()
elif what == "batch":
    sc = bench.build_scene("c5", n, 11)
    groups = [[bench.Replica(sc, "f16", dev, roll=(g * 4 + j) * 97) for j in range(4)] for g in range(2)]
    with torch.cuda.stream(stream):
        steps = [HydroEngine.prepare_step_wrench_tiled_batch([r.engine for r in grp], [r.state for r in grp], sc.dt,
                                                             outs=[r.out for r in grp], prevs=[r.prev for r in grp])[0] for grp in groups]
        bench.spin_up(groups[0], stream, 0.5)
    us = timed(lambda k: steps[k % 2](), launches)
    print(json.dumps({"what": "hydro_step_wrench_tiled_batch, 4 scenes per launch", "n_per_scene": n, "us_per_launch": us,
                      "frac_of_8TBs": 4 * n * 130 / (us * 1e-6) / 8e12}))
else:
    raise SystemExit(__doc__)
