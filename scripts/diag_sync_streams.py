#!/usr/bin/env python3
"""DEV-ONLY (GPU box): does torch.cuda.synchronize() at the end of a 20-step region cost more when more HIP streams exist on
the device (every HydroEngine owns a private one)?  Region overhead = wall - HIP events, for 0 / 8 / 32 extra used-once streams."""
import os, statistics, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
import bench

dev = torch.device("cuda:0"); stream = torch.cuda.Stream(dev)
sc = bench.build_scene("c5", 1048576, 5)
reps = [bench.Replica(sc, "f16", dev, roll=r * 131071) for r in range(4)]
bench.spin_up(reps, stream, 1.0)
K = 20
extra = []


def region():
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        ev0.record(stream); ev1.record(stream)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ev0.record(stream)
        for k in range(K): reps[k % 4].step()
        ev1.record(stream)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
    return wall * 1e6, ev0.elapsed_time(ev1) * 1e3


for n_extra in (0, 8, 32, 0):
    while len(extra) < n_extra:
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            torch.zeros(16, device=dev).add_(1)          # used once: the stream has a hardware queue
        extra.append(s)
    if n_extra == 0:
        extra.clear()
    torch.cuda.synchronize(dev)
    for _ in range(5): region()
    w, e = zip(*[region() for _ in range(40)])
    print(f"{len(extra):3d} extra streams (+ 4 engine streams): wall {statistics.median(w):7.1f} us  events {statistics.median(e):7.1f} us  "
          f"host adds {statistics.median([a - b for a, b in zip(w, e)]):6.1f} us per {K}-step region", flush=True)
