#!/usr/bin/env python3
"""DEV-ONLY: memory probes for the traffic shape of the tiled wrench kernel (scripts/mem_probes.hip), interleaved on
the GPU box.   python scripts/mem_probes.py [bodies ...]   -> gpurun_out/mem_probes.log"""
import ctypes, os, statistics, subprocess, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_out"); os.makedirs(OUT, exist_ok=True)
SO = os.path.join(REPO, "scripts", "_variants", "libmemprobes.so")


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-o", SO,
                    os.path.join(REPO, "scripts", "mem_probes.hip")], check=True)
    return SO


class PArgs(ctypes.Structure):
    _fields_ = [("st", ctypes.c_void_p), ("pv", ctypes.c_void_p), ("prm", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("tiles", ctypes.c_uint32)]


NAMES = {0: "product pattern (4-byte loads/stores)", 1: "same bytes, 16-byte loads/stores", 2: "same reads, 16-byte, no stores",
         3: "float4 copy 1:1, same total bytes"}

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        print(build()); sys.exit(0)
    lib = ctypes.CDLL(SO)
    lib.probe_launch.argtypes = [ctypes.c_int, ctypes.POINTER(PArgs), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    dev = torch.device("cuda:0"); stream = torch.cuda.Stream(dev)
    log = open(os.path.join(OUT, "mem_probes.log"), "a")
    for n in [int(x) for x in sys.argv[1:]] or [1048576, 4194304, 16777216]:
        tiles = n // 64
        sets = 4 if n <= 1048576 else 2
        bufs = []
        for _ in range(sets):
            st = torch.rand(tiles * 832, device=dev); pv = torch.rand(tiles * 384, device=dev)
            prm = torch.rand(tiles * 480, device=dev); out = torch.empty(tiles * 384, device=dev)
            bufs.append((st, pv, prm, out, PArgs(st.data_ptr(), pv.data_ptr(), prm.data_ptr(), out.data_ptr(), tiles)))
        read_b, write_b = tiles * (2816 + 1536 + 1920), tiles * 1536
        n4 = (read_b + write_b) // 2 // 16
        src = [torch.rand(n4 * 4, device=dev) for _ in range(sets)]; dst = [torch.empty(n4 * 4, device=dev) for _ in range(sets)]
        res = {k: [] for k in NAMES}
        sp = ctypes.c_void_p(stream.cuda_stream)
        with torch.cuda.stream(stream):
            for rnd in range(7):
                for k in NAMES:
                    reps = 300 if n <= 1048576 else (100 if n <= 4194304 else 30)
                    for r in range(20):
                        b = bufs[r % sets]; lib.probe_launch(k, ctypes.byref(b[4]), src[r % sets].data_ptr(), dst[r % sets].data_ptr(), n4, sp)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for r in range(reps):
                        b = bufs[r % sets]; lib.probe_launch(k, ctypes.byref(b[4]), src[r % sets].data_ptr(), dst[r % sets].data_ptr(), n4, sp)
                    e1.record(stream); stream.synchronize()
                    res[k].append(e0.elapsed_time(e1) * 1e3 / reps)
        for k, name in NAMES.items():
            us = statistics.median(res[k])
            nbytes = read_b + (write_b if k in (0, 1) else 0) if k != 3 else n4 * 32
            line = f"n={n:9d} {name:42s}: {us:8.2f} us  {nbytes / us / 1e6:7.2f} TB/s of real traffic ({nbytes / 1e6:.1f} MB)"
            print(line, flush=True); log.write(line + "\n")
        del bufs, src, dst
        torch.cuda.empty_cache()
