#!/usr/bin/env python3
"""DEV-ONLY: per-launch durations of the C5 step under different host queueing patterns."""
import os, sys, time, statistics
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import bench
dev = torch.device("cuda:0")
sc = bench.build_scene("c5", 1048576, 5)
reps = [bench.Replica(sc, "f16", dev, roll=r * 131071) for r in range(4)]
stream = torch.cuda.Stream(dev)
bench.spin_up(reps, stream, 1.0)

def run(n, sync_every=0, label=""):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        evs[0].record(stream)
        for k in range(n):
            reps[k % 4].step()
            evs[k + 1].record(stream)
            if sync_every and (k + 1) % sync_every == 0:
                stream.synchronize()
    stream.synchronize()
    wall = time.perf_counter() - t0
    d = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(n)]
    tot = evs[0].elapsed_time(evs[n]) * 1e3 / n
    h = np.histogram(d, bins=[0, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 35, 1e9])[0]
    print(f"{label:28s} n={n} mean(ev total) {tot:6.2f} us  median {statistics.median(d):6.2f}  host {wall / n * 1e6:6.2f} us/launch  hist {list(h)}", flush=True)
    return d

for rnd in range(2):
    run(400, 0, "deep queue 400")
    run(4000, 0, "deep queue 4000")
    run(400, 64, "sync every 64")
    run(400, 8, "sync every 8")
    run(400, 1, "sync every launch")
d = run(600, 0, "deep queue 600 (sequence)")
print([round(x, 1) for x in d[:120]])
