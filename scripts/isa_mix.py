#!/usr/bin/env python3
"""Instruction mix of the compute-bound kernels, read from the gfx950 assembly of the product build (no GPU needed):
the per-step loop of step_fused_multi_tiled_kernel (the resident closed loop) and the whole body of the headline wrench
kernel, by issue class.  bench.py prices the mix with the per-class issue costs measured by scripts/ubench_valu.hip and
reports the VALU-issue fraction of the closed loop next to its us/step.

  python scripts/isa_mix.py            -> profiles/isa_mix.json      (tests/test_isa_budget.py checks it is current)
"""
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "profiles", "isa_mix.json")

# issue classes (scripts/ubench_valu.hip prices them)
def classify(op: str) -> str:
    if not op.startswith("v_"):
        return "not-valu"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)", op):
        return "transcendental"
    if op.startswith("v_cvt_"):
        return "conversion"
    if op.startswith("v_cmp"):
        return "compare"
    if re.search(r"_f64(_|$)", op):
        return "fp64 arithmetic"
    if re.search(r"_(f32|legacy_f32)(_|$)", op):
        return "fp32 arithmetic"
    return "integer / select / move"


KERNELS = {
    # name -> (mangled-name needle, take only the inner loop?)
    "resident closed loop, one step (step_fused_multi_tiled_kernel<f32 parameters, temporal, explicit, no KE, Numba>)":
        ("step_fused_multi_tiled_kernelILb0ELb0ELb0ELb0ELb0E", True),
    "resident closed loop, one step, implicit drag (step_fused_multi_tiled_kernel<f32, temporal, implicit, no KE, Numba>)":
        ("step_fused_multi_tiled_kernelILb0ELb0ELb1ELb0ELb0E", True),
    "headline wrench (wrench_tiled_kernel<256, fp16 coefficients, caller's previous velocity, streaming, no KE, Numba>)":
        ("wrench_tiled_kernelILi256ELb1ELb0ELb1ELb0ELb0E", False),
}


def assembly() -> str:
    from silver2_isaacsim_amd import build as hb
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "hydro.s")
        res = subprocess.run([hb.hipcc_path()] + hb.device_flags() + ["--cuda-device-only", "-S", "-o", out, hb.SRC],
                             capture_output=True, text=True, cwd=d)
        if res.returncode:
            raise RuntimeError(res.stderr[-2000:])
        return open(out).read()


def mix(asm: str) -> dict:
    result = {}
    for name, (needle, loop_only) in KERNELS.items():
        m = re.search(r"^(_Z\S*" + re.escape(needle) + r"[^\s:]*):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M)
        assert m, needle
        body = m.group(2)
        if loop_only:
            lm = re.search(r"^(\.LBB\d+_\d+):[^\n]*Inner Loop Header[^\n]*\n(.*?)^\s+s_branch \1$", body, re.S | re.M)
            assert lm, "inner loop not found in " + needle
            body = lm.group(2)
        ops = re.findall(r"^\s+([a-z][a-z0-9_]+)", body, re.M)
        counts = {}
        for op in ops:
            c = classify(op)
            counts[c] = counts.get(c, 0) + 1
        valu = {k: v for k, v in sorted(counts.items()) if k != "not-valu"}
        result[name] = {"valu_by_class": valu, "valu_total": sum(valu.values()), "other_instructions": counts.get("not-valu", 0)}
    return result


if __name__ == "__main__":
    data = {"source": "hipcc -S of silver2_isaacsim_amd/csrc/hydro_kernels.hip with build.device_flags()", "kernels": mix(assembly())}
    json.dump(data, open(OUT, "w"), indent=1, sort_keys=True)
    for k, v in data["kernels"].items():
        print(k, v)
