#!/usr/bin/env python3
"""DEV-ONLY: interleaved A/B of the stand-alone kinetic-energy entry (hydro_kinetic_energy_tiled, rotational) between whole
libhydro.so builds made by scripts/ab_variants.py build, next to the memory-only probe of its 56 B per body.
    python scripts/ab_variants.py build base= ke512=-DHYDRO_AB_KE512=1@ke512.patch       (CPU container)
    python scripts/ab_ke.py base ke512                                                  (GPU box)  -> gpurun_out/ab_ke.log
Rotating state sets as the bench's kinetic-energy probe (nothing cache-resident); the results must have IDENTICAL bits."""
import os
import statistics
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import numpy as np      # noqa: E402
import torch            # noqa: E402

from silver2_isaacsim_amd import _native as nat, scenes      # noqa: E402
from silver2_isaacsim_amd.engine import HydroEngine          # noqa: E402
import bench                                                 # noqa: E402
from scripts import probes                                   # noqa: E402

VARDIR = os.path.join(REPO, "scripts", "_variants")
names = sys.argv[1:] or ["base", "ke512"]
dev = torch.device("cuda:0"); stream = torch.cuda.Stream(dev)
LOG = open(os.path.join(REPO, "gpurun_out", "ab_ke.log"), "a")


def say(*a):
    s = " ".join(str(x) for x in a); print(s, flush=True); LOG.write(s + "\n"); LOG.flush()


for n, sets, K in ((1048576, 6, 400), (4194304, 2, 120), (262144, 16, 400), (100003, 16, 400)):
    sc = bench.build_scene("c4", n, 11)
    states = [torch.from_numpy(scenes.to_tiled(sc.state[np.roll(np.arange(sc.n), 7919 * k)])).to(dev) for k in range(sets)]
    engines, outs = {}, {}
    for nm in names:
        nat._lib = nat.load(os.path.join(VARDIR, f"libvar_{nm}.so"))
        engines[nm] = [HydroEngine(sc.n, dev, sc.rho, sc.g) for _ in range(sets)]
        for k, e in enumerate(engines[nm]):
            e.set_params(sc.params[np.roll(np.arange(sc.n), 7919 * k)])
        outs[nm] = torch.zeros(2, dtype=torch.float64, device=dev)
    with torch.cuda.stream(stream):
        for nm in names:
            engines[nm][0].kinetic_energy(states[0], True, out=outs[nm])
    stream.synchronize()
    host = scenes.kinetic_energy_fp64(sc.state, sc.params)
    for nm in names:
        same = torch.equal(outs[nm], outs[names[0]])
        rel = max(abs(outs[nm][k].item() - host[k]) / host[k] for k in range(2))
        say(f"n={n} {nm:8s}: bits identical to {names[0]}: {same}; vs fp64 host sum {rel:.2e}")
    # replay-stability and order independence of every variant: 50 launches, same bits
    with torch.cuda.stream(stream):
        for nm in names:
            o = torch.zeros(2, dtype=torch.float64, device=dev)
            ok = True
            for _ in range(50):
                engines[nm][0].kinetic_energy(states[0], True, out=o)
                stream.synchronize()
                ok = ok and torch.equal(o, outs[names[0]])
            say(f"n={n} {nm:8s}: 50 launches, same bits every time: {ok}")
    res = {nm: [] for nm in names}
    with torch.cuda.stream(stream):
        t_end = __import__("time").perf_counter() + 1.0
        while __import__("time").perf_counter() < t_end:          # clocks up
            for k in range(64):
                engines[names[0]][k % sets].kinetic_energy(states[k % sets], True, out=outs[names[0]])
            stream.synchronize()
        for r in range(11):
            for nm in names:
                E, o = engines[nm], outs[nm]
                for k in range(40):
                    E[k % sets].kinetic_energy(states[k % sets], True, out=o)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for k in range(K):
                    E[k % sets].kinetic_energy(states[k % sets], True, out=o)
                e1.record(stream); stream.synchronize()
                res[nm].append(e0.elapsed_time(e1) * 1e3 / K)
    for nm in names:
        v = res[nm]
        say(f"n={n} {nm:8s}: median {statistics.median(v):7.2f} us  min {min(v):7.2f}  max {max(v):7.2f}   ke_frac {n * 56 / (statistics.median(v) * 1e-6) / 8e12:.3f}   rounds: "
            + " ".join(f"{x:.2f}" for x in v))
    for nm in names:
        for e in engines[nm]:
            e.close()
    del states, engines
    torch.cuda.empty_cache()
nat._lib = nat.load(os.path.join(VARDIR, f"libvar_{names[0]}.so"))
for n in (1048576, 4194304):
    r = probes.bound_probes(n, dev, stream, rounds=5, which=(), with_aos=False, with_ke=True)
    for name, v in r["us"].items():
        say(f"n={n} probe run: {name:58s}: {v:8.2f} us")
