#!/usr/bin/env python3
"""DEV-ONLY: does updating the previous velocity IN PLACE cost the array-of-structs entry anything?  The memory-only
probe of its 168 B/body with the 24 B/body of previous velocity written (a) where they were read, as the product does,
(b) into a second buffer, (c) / (d) the same two with write-through stores.   -> gpurun_out/aos_inplace.log"""
import ctypes, os, statistics, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "scripts"))
import torch
import probes

dev = torch.device("cuda:0"); stream = torch.cuda.Stream(dev); L = probes.lib(); sp = ctypes.c_void_p(stream.cuda_stream)
out = open(os.path.join(REPO, "gpurun_out", "aos_inplace.log"), "a")
for n in (1048576, 4194304):
    sets = max(2, -(-(410 << 20) // (n * 52)))
    tiles = n // 64
    A = {k: [] for k in ("in place", "second buffer")}
    keep = []
    for r in range(sets):
        t = [torch.rand(n * 3, device=dev), torch.rand(n * 4, device=dev), torch.rand(n * 6, device=dev), torch.empty(n * 3, device=dev),
             torch.empty(n * 3, device=dev), torch.rand(tiles * 384, device=dev), torch.rand(tiles * 704, device=dev), torch.empty(tiles * 384, device=dev)]
        keep.append(t)
        p = [x.data_ptr() for x in t]
        A["in place"].append(probes.AArgs(p[0], p[1], p[2], p[3], p[4], p[5], p[6], n, p[5]))
        A["second buffer"].append(probes.AArgs(p[0], p[1], p[2], p[3], p[4], p[5], p[6], n, p[7]))
    cases = {(k, w): (A[k], w) for k in A for w in (0, 2)}
    res = {c: [] for c in cases}
    with torch.cuda.stream(stream):
        for rnd in range(7):
            for c, (args, w) in cases.items():
                for k in range(20): L.probe_launch_aos(w, ctypes.byref(args[k % sets]), sp)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for k in range(100): L.probe_launch_aos(w, ctypes.byref(args[k % sets]), sp)
                e1.record(stream); e1.synchronize()
                res[c].append(e0.elapsed_time(e1) * 10.0)
    for (k, w), v in res.items():
        line = f"n={n:8d} previous velocity {k:14s} {'write-through' if w else 'nt':14s}: {statistics.median(v):7.2f} us  (min {min(v):.2f})"
        print(line, flush=True); out.write(line + "\n")
