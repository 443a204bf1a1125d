#!/usr/bin/env python3
"""GPU box, one-off confidence run for the single-launch kinetic-energy reduction: thousands of launches on states that
CHANGE before every launch, several sizes and engines interleaved, stand-alone and in-kernel sampling mixed - every result
compared with the fp64 sum of the state it was launched on (1e-12).  A stale partial or class sum, a lost ticket or a
counter left non-zero would show up as a wrong value or a hang (the script runs under `timeout`).
    python scripts/diag_ke_stress.py [launches_per_size]   -> gpurun_out/ke_stress.log"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from silver2_isaacsim_amd import scenes  # noqa: E402
from silver2_isaacsim_amd.engine import HydroEngine  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = "cuda:0"
sc = scenes.scene_c4(n=65536, seed=23)
cases = []
for n in (1048576, 300000, 4194304, 70001, 257, 16384):
    k = -(-n // sc.n)
    state = np.tile(sc.state, (k, 1))[:n]; params = np.tile(sc.params, (k, 1))[:n]; prev = np.tile(sc.prev, (k, 1))[:n]
    eng = HydroEngine(n, dev, sc.rho, sc.g)
    eng.set_params(params)
    st = torch.from_numpy(scenes.to_tiled(state)).to(dev)
    pv = torch.from_numpy(scenes.to_tiled(prev)).to(dev)
    mass = torch.from_numpy(params[:, 10].astype(np.float64)).to(dev)
    cases.append((n, eng, st, pv, mass, eng.alloc_tiled(6, n)))
t0 = time.time()
worst, checked = 0.0, 0
ke = torch.zeros(2, dtype=torch.float64, device=dev)
for it in range(reps):
    for n, eng, st, pv, mass, out in cases:
        st[:, 7:10, :] *= (1.002 if (it + n) % 3 else 0.995)
        v = st[:, 7:10, :].permute(1, 0, 2).reshape(3, -1)[:, :n].double()
        want = (0.5 * mass * (v * v).sum(0)).sum()
        if it % 2:
            got = eng.kinetic_energy(st, rotational=False)[0]
        else:
            eng.step_wrench_tiled(st, n, sc.dt, out=out, prev=pv, ke_out=ke, rotational=False)
            got = ke[0]
        rel = abs((got - want) / want).item()
        worst = max(worst, rel); checked += 1
        assert rel < 1e-12, (it, n, got.item(), want.item(), rel)
    if it % 500 == 499:
        print(f"{it + 1} rounds, {checked} launches checked, worst relative difference {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
print(f"done: {checked} launches on changing states, 6 sizes, stand-alone and in-kernel sampling alternating: all within 1e-12 (worst {worst:.2e})")
