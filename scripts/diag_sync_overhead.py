#!/usr/bin/env python3
"""DEV-ONLY: what the host adds to a short timed region (K = 20 steps of the C5 headline, ~460 us of GPU work):
wall time between the synchronize pairs minus the HIP-event time of the same steps, for different ways of ending the
region.   python scripts/diag_sync_overhead.py   -> gpurun_out/sync_overhead.log"""
import gc, os, statistics, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
import bench

dev = torch.device("cuda:0"); stream = torch.cuda.Stream(dev)
sc = bench.build_scene("c5", 1048576, 5)
reps = [bench.Replica(sc, "f16", dev, roll=r * 131071) for r in range(4)]
bench.spin_up(reps, stream, 1.0)
K = 20


def region(mode):
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if mode.startswith("pre"):                          # round 5: events created (first record) BEFORE the region, as bench.py does now
        ctx = torch.cuda.stream(stream); ctx.__enter__()
        ev0.record(stream); ev1.record(stream)
        torch.cuda.synchronize(dev)
        gc_was = gc.isenabled()
        if "nogc" in mode: gc.disable()
        t0 = time.perf_counter()
        ev0.record(stream)
        for k in range(K): reps[k % 4].step()
        ev1.record(stream)
        if "evsync" in mode: ev1.synchronize()
        if "streamsync" in mode: stream.synchronize()
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        if gc_was: gc.enable()
        ctx.__exit__(None, None, None)
        return wall * 1e6, ev0.elapsed_time(ev1) * 1e3
    if mode in ("ctx_outside", "ctx_outside+poll", "ctx_outside+streamsync"):
        ctx = torch.cuda.stream(stream); ctx.__enter__()
        t0 = time.perf_counter()
        ev0.record(stream)
        for k in range(K): reps[k % 4].step()
        ev1.record(stream)
        if mode.endswith("poll"):
            while not ev1.query(): pass
        if mode.endswith("streamsync"):
            stream.synchronize()
        torch.cuda.synchronize(dev)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        ctx.__exit__(None, None, None)
    else:
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            ev0.record(stream)
            for k in range(K): reps[k % 4].step()
            ev1.record(stream)
            if mode == "poll":
                while not ev1.query(): pass
        torch.cuda.synchronize(dev)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
    return wall * 1e6, ev0.elapsed_time(ev1) * 1e3


out = open(os.path.join(REPO, "gpurun_out", "sync_overhead.log"), "a")
for mode in ("ctx_outside", "pre", "pre+nogc", "pre+evsync", "pre+streamsync", "ctx_outside", "pre"):
    for _ in range(5): region(mode)
    w, e = zip(*[region(mode) for _ in range(40)])
    line = (f"{mode:26s}: wall {statistics.median(w):7.1f} us  events {statistics.median(e):7.1f} us  host adds {statistics.median([a - b for a, b in zip(w, e)]):6.1f} us "
            f"per {K}-step region ({statistics.median(w) / K:.2f} vs {statistics.median(e) / K:.2f} us per step)")
    print(line, flush=True); out.write(line + "\n")
