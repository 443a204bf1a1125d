#!/usr/bin/env python3
"""DEV-ONLY (GPU box): what one HIP-graph replay of G steps and one KineticEnergyMonitor.observe() cost - the host call and the
step stream's own time (HIP events between them) - at G = 10 and 64 on the 262 144-body configs[3] scene.  Round 5: this is where
the 0.4 ms first pass of a fresh monitor showed up (first line of the output), inside bench.py's 20-step region until
KineticEnergyMonitor.warm_up() moved it out.      python scripts/diag_replay_cost.py"""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from silver2_isaacsim_amd.simulate import KineticEnergyMonitor
dev = torch.device("cuda:0"); stream = torch.cuda.Stream(dev)
full = bench.build_scene("c4", 262144, 4)
reps = [bench.Replica(full, "f32", dev, roll=0) for _ in range(2)]
ke_dev = torch.zeros(2, dtype=torch.float64, device=dev)
mon = KineticEnergyMonitor(reps[0].engine, every=10)
bench.spin_up(reps, stream, 0.3)
for G in (10, 64):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        reps[1].step_sampling(ke_dev); reps[0].step_sampling(ke_dev); stream.synchronize()
        with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
            for k in range(G):
                (reps[k % 2].step_sampling(ke_dev) if k == G - 1 else reps[k % 2].step())
        g.replay(); stream.synchronize()
        for trial in range(3):
            torch.cuda.synchronize()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            for x in e: x.record(stream)
            torch.cuda.synchronize()
            t = [time.perf_counter()]
            e[0].record(stream)
            g.replay(); t.append(time.perf_counter()); e[1].record(stream)
            mon.observe(mon.every * (mon.submitted + 1), stream=stream, sampled=ke_dev); t.append(time.perf_counter()); e[2].record(stream)
            g.replay(); t.append(time.perf_counter()); e[3].record(stream)
            mon.observe(mon.every * (mon.submitted + 1), stream=stream, sampled=ke_dev); t.append(time.perf_counter()); e[4].record(stream)
            torch.cuda.synchronize(); t.append(time.perf_counter())
            mon.collect(block=True)
            print(json.dumps({"graph_steps": G, "host_us": [round((b - a) * 1e6, 1) for a, b in zip(t, t[1:])],
                              "gpu_us_between_events": [round(e[i].elapsed_time(e[i + 1]) * 1e3, 1) for i in range(4)]}), flush=True)
# eager steps for comparison
with torch.cuda.stream(stream):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(20): reps[k % 2].step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(json.dumps({"eager20_host_us": round((t1 - t0) * 1e6, 1), "eager20_total_us": round((t2 - t0) * 1e6, 1)}))
