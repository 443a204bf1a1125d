#!/usr/bin/env python3
"""DEV-ONLY: the tiled entry takes a tile stride per buffer, so a caller may keep state, previous velocity and wrench of a
tile in ONE record ([13 + 6 + 6][64] floats = 6 400 B per tile) instead of three arrays.  Does the memory side care?
C5 headline (fp16 coefficients, four rotating replicas), interleaved A/B of the two layouts through the C ABI.
   -> gpurun_out/interleaved_records.log"""
import ctypes, os, statistics, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import numpy as np, torch
from silver2_isaacsim_amd import _native as nat, scenes
from silver2_isaacsim_amd.engine import HydroEngine
import bench

dev = torch.device("cuda:0"); stream = torch.cuda.Stream(dev); sp = ctypes.c_void_p(stream.cuda_stream)
out = open(os.path.join(REPO, "gpurun_out", "interleaved_records.log"), "a")
L = nat.load()
for n, coeff, sets in ((1048576, "f16", 4), (4194304, "f16", 2)):
    sc = bench.build_scene("c5", n, 11)
    tiles = n // 64
    calls = {"three arrays": [], "one record per tile": [], "one record per tile, params too far": []}
    keep = []
    for r in range(sets):
        idx = np.roll(np.arange(sc.n), 7919 * r)
        e = HydroEngine(sc.n, dev, sc.rho, sc.g); e.set_params(sc.params[idx], coeff)
        st = torch.from_numpy(scenes.to_tiled(sc.state[idx])).to(dev); pv = torch.from_numpy(scenes.to_tiled(sc.prev[idx])).to(dev)
        w = e.alloc_tiled(6, sc.n)
        rec = torch.zeros((tiles, 25, 64), dtype=torch.float32, device=dev)
        rec[:, 0:13] = st; rec[:, 13:19] = pv
        keep.append((e, st, pv, w, rec))
        base = rec.data_ptr()
        def mk(args):
            return lambda: L.hydro_step_wrench_tiled(*args)
        calls["three arrays"].append(mk((e._h, ctypes.c_int64(n), ctypes.c_void_p(st.data_ptr()), ctypes.c_int64(13 * 64), ctypes.c_void_p(pv.data_ptr()),
                                         ctypes.c_int64(6 * 64), ctypes.c_double(sc.dt), ctypes.c_void_p(w.data_ptr()), ctypes.c_int64(6 * 64), sp)))
        calls["one record per tile"].append(mk((e._h, ctypes.c_int64(n), ctypes.c_void_p(base), ctypes.c_int64(25 * 64), ctypes.c_void_p(base + 13 * 256),
                                                ctypes.c_int64(25 * 64), ctypes.c_double(sc.dt), ctypes.c_void_p(base + 19 * 256), ctypes.c_int64(25 * 64), sp)))
    del calls["one record per tile, params too far"]
    # same bits
    with torch.cuda.stream(stream):
        assert calls["three arrays"][0]() == 0 and calls["one record per tile"][0]() == 0
    stream.synchronize()
    assert torch.equal(keep[0][3], keep[0][4][:, 19:25])
    res = {k: [] for k in calls}
    with torch.cuda.stream(stream):
        for rnd in range(9):
            for k, fs in calls.items():
                for j in range(40): fs[j % sets]()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for j in range(200): fs[j % sets]()
                e1.record(stream); e1.synchronize()
                res[k].append(e0.elapsed_time(e1) * 5.0)
    for k, v in res.items():
        line = f"n={n:8d} {coeff} {k:22s}: median {statistics.median(v):7.2f} us  min {min(v):7.2f}   rounds: " + " ".join(f"{x:.1f}" for x in v)
        print(line, flush=True); out.write(line + "\n")
    for t in keep: t[0].close()
    del keep, calls
    torch.cuda.empty_cache()
