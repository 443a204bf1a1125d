#!/usr/bin/env python3
"""DEV-ONLY: interleaved A/B of whole libhydro.so builds (prebuilt here, see `build` below; the variants travel to the
GPU box under scripts/_variants/).  Times the C5 headline step (tiled, fp16 coeffs, 4 rotating replicas) and the fp32 4M step.
The product source has ONE code path per kernel: a variant is the product source plus PATCHES from scripts/ab/ (applied to
a scratch copy, never to the tree) and / or extra compiler flags.
  build (CPU container):  python scripts/ab_variants.py build base= name1=-mllvm,-foo name2=-DHYDRO_AB_TILED_LDS=1@knobs_r03.patch ...
                          (flags comma-separated; @patch[+patch...] = files under scripts/ab/.  knobs_r03.patch brings back
                          the compile-time knobs of rounds 1-3 - HYDRO_AB_* - and the rejected kernels in scripts/ab/*.h)
  run   (GPU box):        python scripts/ab_variants.py run name1 name2 ...        -> gpurun_out/ab.log"""
import os, statistics, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "gpurun_out"); os.makedirs(OUT, exist_ok=True)


VARDIR = os.path.join(REPO, "scripts", "_variants"); os.makedirs(VARDIR, exist_ok=True)   # *.so is git-ignored, not gpurun-ignored


def so(name):
    return os.path.join(VARDIR, f"libvar_{name}.so")


def scratch_source(name, patches):
    """Copy of the product's csrc/ + include/hydro.h (+ the lab headers of scripts/ab/) with `patches` applied."""
    import glob, shutil, subprocess
    root = os.path.join(VARDIR, f"src_{name}")
    shutil.rmtree(root, ignore_errors=True)
    os.makedirs(os.path.join(root, "pkg", "csrc")); os.makedirs(os.path.join(root, "include"))
    for f in glob.glob(os.path.join(REPO, "silver2_isaacsim_amd", "csrc", "*")) + glob.glob(os.path.join(REPO, "scripts", "ab", "*.h")):
        shutil.copy(f, os.path.join(root, "pkg", "csrc"))
    shutil.copy(os.path.join(REPO, "include", "hydro.h"), os.path.join(root, "include"))
    for p in patches:
        subprocess.run(["patch", "-p1", "-i", os.path.join(REPO, "scripts", "ab", p)], cwd=os.path.join(root, "pkg"), check=True)
    return os.path.join(root, "pkg", "csrc", "hydro_kernels.hip")


if sys.argv[1] == "build":
    from silver2_isaacsim_amd import build as hb
    for spec in sys.argv[2:]:
        name, _, rest = spec.partition("=")
        flags, _, patches = rest.partition("@")
        src = scratch_source(name, [p for p in patches.split("+") if p])
        hb.compile_library(src, so(name), [f for f in flags.split(",") if f])
        print("built", so(name), flags, patches)
    sys.exit(0)

import numpy as np, torch
from silver2_isaacsim_amd import _native as nat, scenes
from silver2_isaacsim_amd.engine import HydroEngine
import bench
names = sys.argv[2:]
dev = torch.device("cuda:0"); stream = torch.cuda.Stream(dev)
LOG = open(os.path.join(OUT, "ab.log"), "a")
def say(*a):
    s = " ".join(str(x) for x in a); print(s, flush=True); LOG.write(s + "\n"); LOG.flush()

CASES = [("c5", 1048576, "f16", 4), ("c5", 1048576, "f32", 4), ("c5", 4194304, "f32", 2)]
if os.environ.get("HYDRO_AB_CASES"):            # e.g. "2097152:f16:2,4194304:f16:2"
    CASES = [("c5", int(a), b, int(c)) for a, b, c in (x.split(":") for x in os.environ["HYDRO_AB_CASES"].split(","))]
for kind, n, coeff, sets in CASES:
    sc = bench.build_scene(kind, n, 11)
    reps = {}
    for nm in names:
        import ctypes
        libname, _, opt = nm.partition("@")          # "new@w5b128": lib "new", 5 waves/SIMD cap, 128-thread blocks
        raw = ctypes.CDLL(so(libname)); full = dict(nat.SIGNATURES)
        for k in [k for k in full if not hasattr(raw, k)]:       # older builds lack newer entry points
            del nat.SIGNATURES[k]
        nat._lib = nat.load(so(libname))      # HydroEngine binds whatever nat.load() returns
        nat.SIGNATURES.update(full)
        if not hasattr(nat._lib, "hydro_set_semantics"):
            HydroEngine.set_semantics = lambda self, *_a, **_k: None
        layout = os.environ.get("HYDRO_AB_LAYOUT", "tiled")          # tiled | soa | aos (the simulator-facing entry)
        from scripts.bench_extras import AosReplica
        cls = AosReplica if layout == "aos" else bench.Replica
        reps[nm] = [cls(sc, coeff, dev, roll=7919 * k, layout=layout) for k in range(sets)]
        if opt:
            import re
            w = re.search(r"w(\d)", opt); b = re.search(r"b(\d+)", opt)
            for R in reps[nm]:
                R.engine.set_tuning(0, int(b.group(1)) if b else 0, -1, int(w.group(1)) if w else -1)
    outs = {nm: reps[nm][0] for nm in names}
    with torch.cuda.stream(stream):
        for nm in names: reps[nm][0].step()
    stream.synchronize()
    ref = reps[names[0]][0].out
    for nm in names[1:]:
        d = (reps[nm][0].out - ref).abs().max().item()
        say(f"{kind} n={n} {coeff}: max |{nm} - {names[0]}| = {d:.3e}")
    bench.spin_up(reps[names[0]], stream, 1.0)
    K, ROUNDS = 400 if n <= 2 ** 20 else 120, int(os.environ.get("HYDRO_AB_ROUNDS", "9"))
    res = {nm: [] for nm in names}
    with torch.cuda.stream(stream):
        for r in range(ROUNDS):
            for nm in names:
                R = reps[nm]
                for k in range(40): R[k % sets].step()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for k in range(K): R[k % sets].step()
                e1.record(stream); stream.synchronize()
                res[nm].append(e0.elapsed_time(e1) * 1e3 / K)
    for nm in names:
        v = res[nm]
        say(f"{kind} n={n} {coeff} {nm:12s}: median {statistics.median(v):7.2f} us  min {min(v):7.2f}  max {max(v):7.2f}   rounds: " + " ".join(f"{x:.1f}" for x in v))
    for nm in names:
        for R in reps[nm]: R.engine.close()
    del reps
    torch.cuda.empty_cache()
