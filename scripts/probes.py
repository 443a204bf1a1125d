#!/usr/bin/env python3
"""Which bound binds?  Memory-only and compute-only probes of the wrench kernels (scripts/probes.hip) timed
interleaved with the product kernels themselves, on the GPU box.

    python scripts/probes.py build                      (here: hipcc cross-compiles)  -> scripts/_variants/libprobes.so
    python scripts/probes.py [bodies ...]               (GPU box)  -> gpurun_out/probes.log

`bound_probes(n, dev, stream)` is what bench.py reports as extras.bound_probes_*: {memory_only_us, compute_only_us,
compute_l2_us, kernel_us} for the headline kernel (tiled, fp16 coefficients), and the same pair for the
array-of-structs entry and the kinetic-energy reduction.  Nothing here is imported by the package."""
from __future__ import annotations

import ctypes
import os
import statistics
import subprocess
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "gpurun_out")
SO = os.path.join(REPO, "scripts", "_variants", "libprobes.so")
SRC = os.path.join(REPO, "scripts", "probes.hip")


def build(force: bool = False) -> str:
    """hipcc -> scripts/_variants/libprobes.so (same code-generation flags as libhydro.so: the compute-only probes
    instantiate the product's hydro_body.h)."""
    deps = [SRC, os.path.join(REPO, "silver2_isaacsim_amd", "csrc", "hydro_body.h")]
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(d) for d in deps):
        return SO
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
                    "-ffp-contract=on", "-o", SO, SRC], check=True)
    return SO


class PArgs(ctypes.Structure):
    _fields_ = [("st", ctypes.c_void_p), ("pv", ctypes.c_void_p), ("prm", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("tiles", ctypes.c_uint32)]


class AArgs(ctypes.Structure):
    _fields_ = [("pos", ctypes.c_void_p), ("quat", ctypes.c_void_p), ("vel", ctypes.c_void_p), ("force", ctypes.c_void_p),
                ("torque", ctypes.c_void_p), ("pv", ctypes.c_void_p), ("prm", ctypes.c_void_p), ("n", ctypes.c_uint32),
                ("pvo", ctypes.c_void_p)]


NAMES = {0: "memory-only: product pattern (4-byte loads/stores)", 1: "memory-only: same bytes, 16-byte loads/stores",
         2: "memory-only: same reads, 16-byte, no stores", 3: "float4 copy 1:1, same total bytes",
         4: "compute-only: lane-generated inputs", 5: "compute-only: 64 L2-resident tiles", 6: "memory-only: KE reads (56 B/body)",
         7: "memory-only: product pattern, write-through stores"}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise OSError(f"{SO} not built (python scripts/probes.py build, or __graft_entry__.build())")
        import torch  # noqa: F401  (one HIP runtime per process: torch's first)
        _lib = ctypes.CDLL(SO)
        _lib.probe_launch.argtypes = [ctypes.c_int, ctypes.POINTER(PArgs), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        _lib.probe_launch_aos.argtypes = [ctypes.c_int, ctypes.POINTER(AArgs), ctypes.c_void_p]
    return _lib


def _f16_param_records(params: np.ndarray) -> np.ndarray:
    """(n,11) parameters -> the engine's fp16-coefficient tile records as float32 words: [dimx dimy dimz mass][64] f32,
    then [7][64] f16 (hydro_kernels.hip kPrmTileF16 = 480 words per tile); n a multiple of 64."""
    n = params.shape[0]
    t = n // 64
    rec = np.zeros((t, 480), np.float32)
    p = params.reshape(t, 64, 11)
    for k, f in enumerate((0, 1, 2, 10)):
        rec[:, k * 64:(k + 1) * 64] = p[:, :, f]
    half = np.ascontiguousarray(p[:, :, 3:10].transpose(0, 2, 1)).astype(np.float16).reshape(t, 448)
    rec[:, 256:] = half.view(np.float32).reshape(t, 224)
    return rec


def _tiled_buffers(n: int, sets: int, dev, real_scene=None):
    """Record buffers of the tiled kernel's shape; with `real_scene` the first 64 tiles hold real bodies (what the
    L2-resident compute probe reads)."""
    from silver2_isaacsim_amd import scenes
    tiles = n // 64
    bufs = []
    for _ in range(sets):
        st = torch.rand(tiles * 832, device=dev); pv = torch.rand(tiles * 384, device=dev)
        prm = torch.rand(tiles * max(480, 704), device=dev); out = torch.empty(tiles * 384, device=dev)
        if real_scene is not None:
            m = 64 * 64
            st[:64 * 832] = torch.from_numpy(scenes.to_tiled(real_scene.state[:m]).reshape(-1)).to(dev)
            pv[:64 * 384] = torch.from_numpy(scenes.to_tiled(real_scene.prev[:m]).reshape(-1)).to(dev)
            prm[:64 * 480] = torch.from_numpy(_f16_param_records(real_scene.params[:m]).reshape(-1)).to(dev)
        bufs.append((st, pv, prm, out, PArgs(st.data_ptr(), pv.data_ptr(), prm.data_ptr(), out.data_ptr(), tiles)))
    return bufs


def _time(fn, stream, reps: int, warm: int = 20) -> float:
    for r in range(warm):
        fn(r)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for r in range(reps):
        fn(r)
    e1.record(stream)
    stream.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def bound_probes(n: int, dev, stream, rounds: int = 5, which=(0, 7, 4, 5), with_aos: bool = True, with_ke: bool = True,
                 reps: int | None = None) -> dict:
    """{memory-only, compute-only, kernel} microseconds per launch at n bodies, interleaved round by round (medians).
    Headline kernel = hydro_step_wrench_tiled, fp16 coefficients, rotating replicas as bench.py steps them."""
    import bench
    L = lib()
    sets = 4 if n <= 1048576 else 2
    reps = reps or (200 if n <= 1048576 else 60)
    sc = bench.build_scene("c5", n, 5)
    bufs = _tiled_buffers(n, sets, dev, real_scene=sc)
    sp = ctypes.c_void_p(stream.cuda_stream)
    replicas = [bench.Replica(sc, "f16", dev, roll=r * 131071) for r in range(sets)]
    cases = {}
    for k in which:
        cases[NAMES[k]] = (lambda r, k=k: L.probe_launch(k, ctypes.byref(bufs[r % sets][4]), None, None, 0, sp))
    cases["kernel: hydro_step_wrench_tiled (fp16 coefficients)"] = lambda r: replicas[r % sets].step()
    aos = None
    if with_aos:
        aos = _aos_case(sc, n, sets, dev, L, sp)
        cases.update(aos["cases"])
    if with_ke:
        cases[NAMES[6]] = lambda r: L.probe_launch(6, ctypes.byref(bufs[r % sets][4]), None, None, 0, sp)
        ke_out = torch.zeros(2, dtype=torch.float64, device=dev)
        cases["kernel: hydro_kinetic_energy_tiled (rotational)"] = \
            lambda r: replicas[r % sets].engine.kinetic_energy(replicas[r % sets].state, True, out=ke_out)
    res = {name: [] for name in cases}
    with torch.cuda.stream(stream):
        bench.spin_up(replicas, stream, 0.3)
        for _ in range(rounds):
            for name, fn in cases.items():
                res[name].append(_time(fn, stream, reps))
    out = {"n": n, "rotating_sets": sets, "us": {name: statistics.median(v) for name, v in res.items()}}
    for r in replicas:
        r.engine.close()
    if aos:
        for e in aos["engines"]:
            e.close()
    return out


def clock_probes(n: int, dev, stream, seconds: float = 2.0) -> dict:
    """The shader clock the chip holds under each kind of work (DVFS), read in-kernel: every wave stamps s_memtime /
    s_memrealtime around its body.  Each kind is run back to back for `seconds` first (the clock needs that long to
    settle), then 50 stamped launches are evaluated: median over waves and launches of cycles / real time.
    Returns {kind: {"ghz", "wave_lifetime_us", "us_per_launch"}} for the whole body (what the tiled wrench kernel does),
    memory only, compute only, and sustained arithmetic (64 passes of the body per wave: the resident closed loop's load)."""
    import time
    import bench
    L = lib()
    L.probe_launch_clock.argtypes = [ctypes.c_int, ctypes.POINTER(PArgs), ctypes.c_void_p, ctypes.c_void_p]
    sets = 4 if n <= 1048576 else 2
    sc = bench.build_scene("c5", n, 5)
    tiles = n // 64
    from silver2_isaacsim_amd import scenes
    bufs = _tiled_buffers(n, sets, dev, real_scene=None)
    # real bodies everywhere: the whole-body probe must do the product's arithmetic on the product's data
    st = torch.from_numpy(scenes.to_tiled(sc.state).reshape(-1)).to(dev); pv = torch.from_numpy(scenes.to_tiled(sc.prev).reshape(-1)).to(dev)
    prm = torch.from_numpy(_f16_param_records(sc.params).reshape(-1)).to(dev)
    for b in bufs:
        b[0][:st.numel()] = st; b[1][:pv.numel()] = pv; b[2][:prm.numel()] = prm
    stamps = torch.zeros(2 * tiles, dtype=torch.int64, device=dev)
    sp = ctypes.c_void_p(stream.cuda_stream)
    out = {}
    with torch.cuda.stream(stream):
        for kind, name in ((0, "whole_body"), (1, "memory_only"), (2, "compute_only"), (3, "sustained_arithmetic")):
            t0 = time.perf_counter(); r = 0
            while time.perf_counter() - t0 < seconds:
                for _ in range(64):
                    L.probe_launch_clock(kind, ctypes.byref(bufs[r % sets][4]), stamps.data_ptr(), sp); r += 1
                stream.synchronize()
            ghz, life = [], []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for k in range(50):
                L.probe_launch_clock(kind, ctypes.byref(bufs[k % sets][4]), stamps.data_ptr(), sp)
                if k % 10 == 9:                          # (reading the stamps back drains the queue: every tenth launch only)
                    stream.synchronize()
                    v = stamps.view(tiles, 2).double()
                    ok = v[:, 1] > 0
                    ghz.append(float((v[ok, 0] / v[ok, 1]).median()) * 0.1)
                    life.append(float(v[ok, 1].median()) * 0.01)
            e1.record(stream); stream.synchronize()
            out[name] = {"ghz": statistics.median(ghz), "wave_lifetime_us": statistics.median(life)}
    return out


def _aos_case(sc, n, sets, dev, L, sp):
    """The array-of-structs entry (fp32 parameters, engine-owned previous velocity) beside memory-only probes of its
    168 B/body in both access shapes.  Enough rotating sets that the simulator's rows (52 B per body, read with temporal
    loads) cannot stay in the Infinity Cache between two uses."""
    sets = max(sets, -(-(410 << 20) // (n * 52)))
    from silver2_isaacsim_amd.engine import HydroEngine
    tiles = n // 64
    engines, tens, args = [], [], []
    for r in range(sets):
        idx = np.roll(np.arange(sc.n), r * 97)
        e = HydroEngine(sc.n, dev, sc.rho, sc.g)
        e.set_params(sc.params[idx]); e.set_prev_velocity(sc.prev[idx])
        st = sc.state[idx]
        pos = torch.from_numpy(np.ascontiguousarray(st[:, 0:3])).to(dev)
        quat = torch.from_numpy(np.ascontiguousarray(st[:, [6, 3, 4, 5]])).to(dev)
        vel = torch.from_numpy(np.ascontiguousarray(st[:, 7:13])).to(dev)
        f, t = torch.empty((sc.n, 3), device=dev), torch.empty((sc.n, 3), device=dev)
        pv, prm = torch.rand(tiles * 384, device=dev), torch.rand(tiles * 704, device=dev)
        engines.append(e); tens.append((pos, quat, vel, f, t, pv, prm))
        args.append(AArgs(pos.data_ptr(), quat.data_ptr(), vel.data_ptr(), f.data_ptr(), t.data_ptr(), pv.data_ptr(), prm.data_ptr(), n,
                          pv.data_ptr()))
    steps = [engines[r].prepare_step_wrench_aos(*tens[r][:3], forces=tens[r][3], torques=tens[r][4]) for r in range(sets)]
    cases = {
        "memory-only: AoS traffic, 16-byte chunks per wave": lambda r: L.probe_launch_aos(1, ctypes.byref(args[r % sets]), sp),
        "memory-only: AoS traffic, one row per lane": lambda r: L.probe_launch_aos(0, ctypes.byref(args[r % sets]), sp),
        "kernel: hydro_step_wrench_aos (fp32 parameters)": lambda r: steps[r % sets](sc.dt),
    }                                   # (`sets` here is this function's own, possibly larger, count)
    return {"cases": cases, "engines": engines, "keep": tens}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        print(build(force=True)); sys.exit(0)
    os.makedirs(OUT, exist_ok=True)
    dev = torch.device("cuda:0"); stream = torch.cuda.Stream(dev)
    log = open(os.path.join(OUT, "probes.log"), "a")
    for n in [int(x) for x in sys.argv[1:]] or [1048576, 4194304]:
        r = bound_probes(n, dev, stream, rounds=7, which=(0, 7, 1, 2, 4, 5))
        for name, us in r["us"].items():
            line = f"n={n:9d} {name:58s}: {us:8.2f} us"
            print(line, flush=True); log.write(line + "\n")
        log.flush()
        torch.cuda.empty_cache()
        c = clock_probes(n, dev, stream)
        for name, v in c.items():
            line = f"n={n:9d} clock held under {name:13s}: {v['ghz']:.3f} GHz   (median wave lifetime {v['wave_lifetime_us']:.2f} us)"
            print(line, flush=True); log.write(line + "\n")
        log.flush()
        torch.cuda.empty_cache()
