#!/usr/bin/env python3
"""DEV-ONLY, to be run under rocprofv3 --pmc FETCH_SIZE (or WRITE_SIZE): the fused closed-loop step of ONE
1 048 576-body scene, 200 eager steps with temporal and 200 with non-temporal accesses (the two show up as
step_fused_tiled_kernel<..., false/true, ...> in the counter file)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import bench
from silver2_isaacsim_amd.simulate import ClosedLoopSim
sc = bench.build_scene("c2", 1048576, 17)
for nt in (0, 1):
    sim = ClosedLoopSim(sc, fused=True, implicit_drag=True)
    sim.engine.set_tuning(0, 0, nt)
    sim.run_eager(200)
    sim.synchronize()
    sim.close()
