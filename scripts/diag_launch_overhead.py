#!/usr/bin/env python3
"""DEV-ONLY: host cost of one step call (4 096 bodies) - engine method vs the raw ctypes call with prebuilt
arguments vs HIP-graph replay.   python scripts/diag_launch_overhead.py"""
import ctypes, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from silver2_isaacsim_amd import scenes, _native as nat
from silver2_isaacsim_amd.engine import HydroEngine
dev = torch.device("cuda:0")
sc = scenes.scene_c2()
eng = HydroEngine(sc.n, dev, sc.rho, sc.g); eng.set_params(sc.params)
S = torch.from_numpy(scenes.to_tiled(sc.state)).to(dev); P = torch.from_numpy(scenes.to_tiled(sc.prev)).to(dev)
O = eng.alloc_tiled(6, sc.n)
stream = torch.cuda.Stream(dev)


def bench(fn, k=20000):
    with torch.cuda.stream(stream):
        for _ in range(200): fn()
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(k): fn()
        t_issue = time.perf_counter() - t0
        stream.synchronize()
        t_all = time.perf_counter() - t0
    return t_issue / k * 1e6, t_all / k * 1e6


with torch.cuda.stream(stream):
    a = bench(lambda: eng.step_wrench_tiled(S, sc.n, sc.dt, out=O, prev=P))
    print(f"engine.step_wrench_tiled      host issue {a[0]:6.2f} us/call   wall {a[1]:6.2f} us/step")
    lib = eng._lib
    args = (eng._h, ctypes.c_int64(sc.n), ctypes.c_void_p(S.data_ptr()), ctypes.c_int64(13 * 64), ctypes.c_void_p(P.data_ptr()),
            ctypes.c_int64(6 * 64), ctypes.c_double(sc.dt), ctypes.c_void_p(O.data_ptr()), ctypes.c_int64(6 * 64),
            ctypes.c_void_p(stream.cuda_stream))
    fn = lib.hydro_step_wrench_tiled
    b = bench(lambda: fn(*args))
    print(f"raw ctypes, prebuilt args     host issue {b[0]:6.2f} us/call   wall {b[1]:6.2f} us/step")
    if hasattr(eng, "prepare_step_wrench_tiled"):
        step = eng.prepare_step_wrench_tiled(S, sc.n, sc.dt, out=O, prev=P, stream=stream)
        c = bench(step)
        print(f"prepared step                 host issue {c[0]:6.2f} us/call   wall {c[1]:6.2f} us/step")
