#!/usr/bin/env python3
"""DEV-ONLY: does a single scene stepping on itself profit from the 256 MiB Infinity Cache when the
accesses are NOT marked non-temporal?  Closed loop (fused step, hipGraph x64) at several sizes, nt = 0 / 1.
python scripts/diag_mall.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import bench
from silver2_isaacsim_amd.simulate import ClosedLoopSim
for n in [int(x) for x in sys.argv[1:]] or (262144, 524288, 786432, 1048576, 1572864, 2097152):
    sc = bench.build_scene("c2", n, 17)
    row = []
    for nt in (0, 1):
        sim = ClosedLoopSim(sc, fused=True, implicit_drag=True)
        sim.engine.set_tuning(0, 0, nt)
        r = sim.measure_rtf(1024, graph_steps=64)
        r = sim.measure_rtf(2048, graph_steps=64)
        row.append(r["us_per_step"])
        sim.close()
    ws = n * (2 * 52 + 44 + 0) / 2**20
    print(f"n={n:8d}  working set ~{ws:6.0f} MiB   temporal {row[0]:7.2f} us/step   non-temporal {row[1]:7.2f} us/step   "
          f"({n * 172 / row[0] / 1e3:6.0f} vs {n * 172 / row[1] / 1e3:6.0f} GB/s of 172 B/body)", flush=True)
