#!/usr/bin/env python3
"""DEV-ONLY: interleaved A/B of libhydro builds on the SMALL configs (C2 4 096, C3 19 456, C4 shard 32 768), 64 steps
per HIP-graph replay so that the kernel, not the host call, is timed.   python scripts/ab_small.py r1 n3 ginf"""
import ctypes, os, statistics, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
from silver2_isaacsim_amd import _native as nat
from silver2_isaacsim_amd.engine import HydroEngine
import bench
names = sys.argv[1:]
dev = torch.device("cuda:0"); stream = torch.cuda.Stream(dev)
so = lambda n: os.path.join(REPO, "scripts", "_variants", f"libvar_{n}.so")
for kind, n in (("c2", 4096), ("c3", 19456), ("c4", 32768), ("c4", 262144)):
    sc = bench.build_scene(kind, n, 11)
    graphs, reps = {}, {}
    for nm in names:
        raw = ctypes.CDLL(so(nm)); full = dict(nat.SIGNATURES)
        for k in [k for k in full if not hasattr(raw, k)]:
            del nat.SIGNATURES[k]
        nat._lib = nat.load(so(nm)); nat.SIGNATURES.update(full)
        reps[nm] = [bench.Replica(sc, "f32", dev, roll=7919 * k) for k in range(4)]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            for k in range(8): reps[nm][k % 4].step()
            stream.synchronize()
            with torch.cuda.graph(g, stream=stream):
                for k in range(64): reps[nm][k % 4].step()
        graphs[nm] = g
    bench.spin_up(reps[names[0]], stream, 0.5)
    res = {nm: [] for nm in names}
    with torch.cuda.stream(stream):
        for r in range(9):
            for nm in names:
                for _ in range(5): graphs[nm].replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(40): graphs[nm].replay()
                e1.record(stream); stream.synchronize()
                res[nm].append(e0.elapsed_time(e1) * 1e3 / (40 * 64))
    for nm in names:
        v = res[nm]
        print(f"{kind} n={n} graph {nm:8s}: median {statistics.median(v):6.3f} us/step  min {min(v):6.3f}  max {max(v):6.3f}", flush=True)
    del graphs
    for nm in names:
        for R in reps[nm]: R.engine.close()
