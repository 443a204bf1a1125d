#!/usr/bin/env python3
"""DEV-ONLY: throughput of the AoS entry point and per-step cost of the plugin path."""
import os, sys, time, statistics
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import bench
from silver2_isaacsim_amd import scenes
from silver2_isaacsim_amd.engine import HydroEngine
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(dev)

def aos_rate(n, coeff="f32", sets=4, steps=200):
    sc = bench.build_scene("c4", n, 3)
    R = []
    for k in range(sets):
        e = HydroEngine(n, dev, sc.rho, sc.g); e.set_params(sc.params, coeff); e.set_prev_velocity(sc.prev)
        pos = torch.from_numpy(np.ascontiguousarray(sc.state[:, 0:3])).to(dev)
        q = torch.from_numpy(np.ascontiguousarray(sc.state[:, [6, 3, 4, 5]])).to(dev)
        vel = torch.from_numpy(np.ascontiguousarray(sc.state[:, 7:13])).to(dev)
        F = torch.empty((n, 3), device=dev); T = torch.empty((n, 3), device=dev)
        R.append((e, pos, q, vel, F, T))
    with torch.cuda.stream(stream):
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 0.3:
            e, pos, q, vel, F, T = R[k % sets]; e.step_wrench_aos(pos, q, vel, sc.dt, forces=F, torques=T); k += 1
            if k % 64 == 0: stream.synchronize()
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for k in range(steps):
            e, pos, q, vel, F, T = R[k % sets]; e.step_wrench_aos(pos, q, vel, sc.dt, forces=F, torques=T)
        e1.record(stream); stream.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / steps
    for r in R: r[0].close()
    print(f"AoS entry n={n} {coeff}: {us:.2f} us/step, {n / us * 1e6:.3e} body-steps/s, {n * 168 / us / 1e3:.0f} GB/s (168 B/body incl. prev update)", flush=True)

for n in (4096, 262144, 1048576, 4194304):
    aos_rate(n)

# plugin path: the 20 prims of the main scene through HydrodynamicsBehavior + FakeHost
sys.path.insert(0, os.path.join(REPO, 'tests'))
from test_plugin_gpu import build_scene
from silver2_isaacsim_amd import behavior as hb
for batched in (True, False):
    hb.REGISTRY.clear()
    world, host, prims, behaviors = build_scene(batched)
    for b in behaviors: b.on_play()
    for _ in range(50): host.step(1 / 60)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); N = 500
    for _ in range(N): host.step(1 / 60)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    print(f"plugin path, 20 prims, batched={batched}: {dt * 1e6:.1f} us per physics step ({1 / dt:.0f} steps/s, RTF at 60 Hz = {1 / dt / 60:.0f}x)", flush=True)
    for b in behaviors: b.on_stop()
