#!/usr/bin/env python3
"""DEV-ONLY: achieved bandwidth of the auxiliary kernels at 1M bodies (tiled layout)."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import bench
from silver2_isaacsim_amd import scenes
from silver2_isaacsim_amd.engine import HydroEngine
dev = torch.device("cuda:0"); stream = torch.cuda.Stream(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
sc = bench.build_scene("c4", n, 3)
sets = 4
R = []
for k in range(sets):
    e = HydroEngine(n, dev, sc.rho, sc.g); e.set_params(sc.params)
    S = torch.from_numpy(scenes.to_tiled(sc.state)).to(dev); P = torch.from_numpy(scenes.to_tiled(np.concatenate([np.zeros((n, 7), np.float32), sc.prev], 1))).to(dev)
    W = e.alloc_tiled(6, n); O = e.alloc_tiled(13, n); ke = torch.empty(2, dtype=torch.float64, device=dev)
    e.step_wrench_tiled(S, n, sc.dt, out=W, prev=P)
    R.append((e, S, P, W, O, ke))

def timeit(name, fn, bytes_per_body, steps=200):
    with torch.cuda.stream(stream):
        t0 = time.perf_counter(); k = 0
        while time.perf_counter() - t0 < 0.2:
            fn(R[k % sets]); k += 1
            if k % 64 == 0: stream.synchronize()
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for k in range(steps): fn(R[k % sets])
        e1.record(stream); stream.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / steps
    print(f"{name:34s} {us:8.2f} us  {n * bytes_per_body / us / 1e3:7.0f} GB/s of {bytes_per_body} B/body ({n * bytes_per_body / us / 1e3 / 80:5.1f}% of 8 TB/s)", flush=True)

timeit("wrench_tiled (fp32)", lambda r: r[0].step_wrench_tiled(r[1], n, sc.dt, out=r[3], prev=r[2]), 144)
timeit("integrate_tiled", lambda r: r[0].integrate_tiled(r[1], r[3], n, sc.dt, state_out=r[4]), 52 + 24 + 16 + 52)
timeit("step_fused_tiled", lambda r: r[0].step_fused_tiled(r[1], r[2], n, sc.dt, state_out=r[4]), 52 + 24 + 44 + 52)
timeit("kinetic_energy_tiled (lin)", lambda r: r[0].kinetic_energy(r[1], False, out=r[5]), 12 + 4)
timeit("kinetic_energy_tiled (lin+rot)", lambda r: r[0].kinetic_energy(r[1], True, out=r[5]), 40 + 16)
pos = torch.from_numpy(np.ascontiguousarray(sc.state[:, 0:3])).to(dev); q = torch.from_numpy(np.ascontiguousarray(sc.state[:, [6, 3, 4, 5]])).to(dev); vel = torch.from_numpy(np.ascontiguousarray(sc.state[:, 7:13])).to(dev)
F = torch.empty((n, 3), device=dev); T = torch.empty((n, 3), device=dev)
timeit("pack_state_aos", lambda r: r[0].pack_state_aos(pos, q, vel, out=r[4]), 52 + 52)
timeit("unpack_wrench_aos", lambda r: r[0].unpack_wrench_aos(r[3], n, forces=F, torques=T), 24 + 24)
