#!/usr/bin/env python3
"""DEV-ONLY: interleaved A/B timing of kernel variants (scripts/tune_kernels.hip) on the GPU box.
Usage: python scripts/tune.py [n ...]   -> gpurun_out/tune.log"""
import ctypes
import os
import statistics
import subprocess
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from silver2_isaacsim_amd import scenes          # noqa: E402

OUT = os.path.join(REPO, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
SO = os.path.join(OUT, "libtune.so")
LOG = open(os.path.join(OUT, "tune.log"), "a")


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); LOG.write(s + "\n"); LOG.flush()


def build():
    src = os.path.join(REPO, "scripts", "tune_kernels.hip")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
                        "-o", SO, src], check=True)
    return ctypes.CDLL(SO)


class TArgs(ctypes.Structure):
    _fields_ = [("st", ctypes.c_void_p * 13), ("pv", ctypes.c_void_p * 6), ("dims", ctypes.c_void_p * 3),
                ("coef", ctypes.c_void_p * 7), ("mass", ctypes.c_void_p), ("out", ctypes.c_void_p * 6),
                ("rho", ctypes.c_float), ("g", ctypes.c_float), ("inv_dt", ctypes.c_float), ("n", ctypes.c_uint32)]


VARIANTS = {0: "baseline 256", 3: "non-temporal", 8: "block 128", 11: "memory-only nt", 12: "nt + block 128",
            13: "memory-only nt b128", 14: "nt + block 64"}
MEMV = {0: "tiled64 4buf nt b256", 1: "tiled64 4buf nt b128", 2: "tiled256 4buf nt", 3: "tiled64 1buf nt b256",
        4: "tiled64 1buf nt b128", 5: "tiled64 4buf plain", 6: "float4 copy 4:1", 7: "float4 copy 4:1 nt-load"}


class MArgs(ctypes.Structure):
    _fields_ = [("st", ctypes.c_void_p), ("pv", ctypes.c_void_p), ("pr", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("n", ctypes.c_uint32)]


def main():
    lib = build()
    lib.tune_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(TArgs), ctypes.c_int, ctypes.c_void_p]
    dev = torch.device("cuda:0")
    sizes = [int(x) for x in sys.argv[1:]] or [1048576, 4194304]
    base = scenes.scene_c4(n=262144, seed=9)
    for n in sizes:
        reps = (n + base.n - 1) // base.n
        st = np.tile(base.state, (reps, 1))[:n]; pv = np.tile(base.prev, (reps, 1))[:n]; pr = np.tile(base.params, (reps, 1))[:n]
        sets = 4 if n <= 2 ** 21 else 2
        bufs = []
        for k in range(sets):
            S = torch.from_numpy(scenes.to_soa(st)).to(dev); P = torch.from_numpy(scenes.to_soa(pv)).to(dev)
            Q = torch.from_numpy(scenes.to_soa(pr)).to(dev); H = Q[3:10].to(torch.float16).contiguous()
            O = torch.empty((6, n), device=dev)
            both = []
            for half in (0, 1):
                a = TArgs()
                for f in range(13): a.st[f] = S.data_ptr() + f * n * 4
                for f in range(6): a.pv[f] = P.data_ptr() + f * n * 4
                for f in range(3): a.dims[f] = Q.data_ptr() + f * n * 4
                for f in range(7): a.coef[f] = (H.data_ptr() + f * n * 2) if half else (Q.data_ptr() + (3 + f) * n * 4)
                a.mass = Q.data_ptr() + 10 * n * 4
                for f in range(6): a.out[f] = O.data_ptr() + f * n * 4
                a.rho, a.g, a.inv_dt, a.n = 1025.0, 9.81, 60.0, n
                both.append(a)
            bufs.append((S, P, Q, H, O, both))
        stream = torch.cuda.Stream(dev)
        K, ROUNDS = 40, 7
        for half in (0, 1):
            res = {v: [] for v in VARIANTS}
            persist_blocks = 256 * 6
            with torch.cuda.stream(stream):
                for rnd in range(ROUNDS):
                    for v in VARIANTS:
                        for k in range(4):
                            lib.tune_launch(v, half, ctypes.byref(bufs[k % sets][5][half]), persist_blocks, stream.cuda_stream)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(stream)
                        for k in range(K):
                            rc = lib.tune_launch(v, half, ctypes.byref(bufs[k % sets][5][half]), persist_blocks, stream.cuda_stream)
                            assert rc == 0, (v, rc)
                        e1.record(stream)
                        stream.synchronize()
                        res[v].append(e0.elapsed_time(e1) * 1e3 / K)
            bpb = 130 if half else 144
            for v, name in VARIANTS.items():
                med, mn = statistics.median(res[v]), min(res[v])
                say(f"n={n} coeff={'f16' if half else 'f32'} {name:22s} median {med:7.2f} us  min {mn:7.2f} us  "
                    f"{n * bpb / med / 1e3:7.1f} GB/s alg ({n * bpb / med / 1e3 / 80:5.1f}% of 8 TB/s)")
        # ---- memory-only layout probes (fp32 fields: 28 read + 6 written per body) ----
        lib.tune_mem.argtypes = [ctypes.c_int, ctypes.POINTER(MArgs), ctypes.c_void_p]
        mb = []
        for k in range(sets):
            big = torch.rand((30 * n,), device=dev)            # one 30-field record buffer (also serves st/pv/pr views)
            out = torch.empty((6 * n,), device=dev)
            m = MArgs(); m.st = big.data_ptr(); m.pv = big.data_ptr() + 13 * n * 4; m.pr = big.data_ptr() + 19 * n * 4
            m.out = out.data_ptr(); m.n = n
            mb.append((big, out, m))
        resm = {v: [] for v in MEMV}
        with torch.cuda.stream(stream):
            for rnd in range(ROUNDS):
                for v in MEMV:
                    for k in range(4):
                        lib.tune_mem(v, ctypes.byref(mb[k % sets][2]), stream.cuda_stream)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for k in range(K):
                        rc = lib.tune_mem(v, ctypes.byref(mb[k % sets][2]), stream.cuda_stream)
                        assert rc == 0, (v, rc)
                    e1.record(stream)
                    stream.synchronize()
                    resm[v].append(e0.elapsed_time(e1) * 1e3 / K)
        for v, name in MEMV.items():
            med = statistics.median(resm[v])
            actual = n * ((28 + 6) * 4 if v < 6 else (96 + 24))
            say(f"n={n} MEM {name:26s} median {med:7.2f} us  actual {actual / med / 1e3:7.1f} GB/s  "
                f"(as 144 B/body: {n * 144 / med / 1e3 / 80:5.1f}% of 8 TB/s)")
        del bufs, mb
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
