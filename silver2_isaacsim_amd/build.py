"""In-tree build of libhydro.so for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "hydro_kernels.hip")
DEPS = [SRC, os.path.join(_HERE, "csrc", "hydro_body.h"), os.path.join(os.path.dirname(_HERE), "include", "hydro.h")]
OUT = os.path.join(_HERE, "lib", "libhydro.so")
ARCH = "gfx950"


def hipcc_path() -> str:
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found")
    return p


def is_stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def device_flags() -> list[str]:
    """Code-generation flags of the product library (also used by tests/test_isa_budget.py to read the assembly)."""
    return ["-O3", f"--offload-arch={ARCH}", "-std=c++17",
            # the SLP vectoriser packs the scalar fp32 math into v_pk_* pairs at the price of ~115
            # v_mov and +24 VGPRs per lane (103 -> 79 without it): occupancy matters more here
            "-fno-slp-vectorize",
            # FMA contraction per source expression only (the HIP default, "fast", contracts across
            # statements and does so differently in each template instantiation): every kernel
            # variant, SoA or AoS, 1 or 2 bodies per lane, then returns the same bits for a body
            "-ffp-contract=on",
            # kernarg preload (gfx950): the first 16 dwords of a kernel's scalar arguments arrive in SGPRs with the wave
            # instead of behind scalar-memory loads; the hot kernels order their arguments for it (hydro_kernels.hip)
            "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def compile_library(src: str, out: str, extra_flags: list[str] | None = None, verbose: bool = False) -> str:
    """hipcc `src` -> shared library `out` with the product's flags (also used by scripts/ab_variants.py for lab builds)."""
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [hipcc_path()] + device_flags() + ["-fPIC", "-shared", "-Wall", "-Wno-unused-function"] \
        + (extra_flags or []) + ["-o", out + ".tmp", src]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(" ".join(cmd))
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed ({res.returncode})")
    os.replace(out + ".tmp", out)
    return out


def build(force: bool = False, verbose: bool = False, extra_flags: list[str] | None = None) -> str:
    """Compile csrc/hydro_kernels.hip -> lib/libhydro.so.  Returns the library path."""
    if not force and not is_stale():
        return OUT
    return compile_library(SRC, OUT, extra_flags, verbose)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
