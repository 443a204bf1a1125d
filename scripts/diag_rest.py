#!/usr/bin/env python3
"""DEV-ONLY: boxes whose power management throttles under SUSTAINED combined load - how long a rest
after sustained load restores the full rate, and what a rest costs a box that does not throttle (clocks falling back to
idle).  After 0.5 s of back-to-back C5 steps: rest r, then 5 warm-up + 20 timed steps (the driver's region) and
straight after them 200 timed steps; HIP events.   -> gpurun_out/rest.log"""
import os, statistics, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
import bench

dev = torch.device("cuda:0"); stream = torch.cuda.Stream(dev)
out = open(os.path.join(REPO, "gpurun_out", "rest.log"), "a")
def say(s):
    print(s, flush=True); out.write(s + "\n"); out.flush()
sc = bench.build_scene("c5", 1048576, 5)
reps = [bench.Replica(sc, "f16", dev, roll=r * 131071) for r in range(4)]

def burst(k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for j in range(k): reps[j % 4].step()
    e1.record(stream); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / k

with torch.cuda.stream(stream):
    bench.spin_up(reps, stream, 1.0)
    sustained = statistics.median(burst(2000) for _ in range(3))
    say(f"sustained (3 x 2000 steps back to back after 1 s): {sustained:.2f} us per step")
    for rest_ms in (0, 1, 3, 10, 30, 100, 300, 1000, 3000):
        v20, v200 = [], []
        for rep in range(3):
            bench.spin_up(reps, stream, 0.5)
            torch.cuda.synchronize(dev)
            if rest_ms: time.sleep(rest_ms * 1e-3)
            for j in range(5): reps[j % 4].step()
            v20.append(burst(20)); v200.append(burst(200))
        say(f"rest {rest_ms:5d} ms: 20 steps {statistics.median(v20):6.2f} us/step   next 200 steps {statistics.median(v200):6.2f}   "
            + " ".join(f"{a:.1f}/{b:.1f}" for a, b in zip(v20, v200)))
