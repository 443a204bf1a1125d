"""What DESIGN.md section 5 says about the generated code, read back from the gfx950 assembly of the product build
(hipcc cross-compiles without a GPU; ~20 s): no kernel of the library spills, none uses MFMA, and the headline kernel is
the streaming kernel the measurements are quoted for - 28 non-temporal loads, 6 write-through stores, no LDS, registers
for 4 waves per SIMD, the instruction diet not undone."""
import os
import re
import subprocess

import pytest

from conftest import REPO
from silver2_isaacsim_amd import build as hb

HEADLINE = "wrench_tiled_kernelILi256ELb1ELb0ELb1ELb0ELb0E"     # <256, fp16 coefficients, caller's previous velocity, streaming, no KE, Numba semantics>


@pytest.fixture(scope="module")
def assembly(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("isa") / "hydro.s")
    cmd = [hb.hipcc_path()] + hb.device_flags() + ["--cuda-device-only", "-S", "-o", out, hb.SRC]
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(out))
    assert res.returncode == 0, res.stderr[-3000:]
    return open(out).read()


def _descriptors(asm):
    """kernel name -> {directive: value} from the .amdhsa_kernel blocks."""
    out = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", asm, re.S):
        out[m.group(1)] = dict(re.findall(r"\.amdhsa_(\w+) (\S+)", m.group(2)))
    return out


def _body(asm, needle):
    m = re.search(r"^(_Z\S*" + re.escape(needle) + r"[^\s:]*):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M)
    assert m, needle
    return m.group(2)


def test_no_kernel_spills_and_none_uses_mfma(assembly):
    desc = _descriptors(assembly)
    assert len(desc) > 100                                    # every template instantiation of the library
    spilled = {k: d["private_segment_fixed_size"] for k, d in desc.items() if int(d["private_segment_fixed_size"]) != 0}
    assert not spilled, spilled
    assert "v_mfma" not in assembly and "v_smfma" not in assembly
    # 3 waves per SIMD (512 / 3 = 170 registers) for the plain-SoA and the resident multi-step kernels, 4 or more for all
    # others; only the two-bodies-per-lane tuning variants of the plain-SoA kernel (hydro_set_tuning, never the default) need more
    default = {k: int(d["next_free_vgpr"]) for k, d in desc.items() if not re.search(r"wrench_soa_kernelILi\d+ELi2E", k)}
    assert max(default.values()) <= 168, max(default.items(), key=lambda kv: kv[1])
    tiled = {k: v for k, v in default.items() if "wrench_tiled_kernel" in k or "wrench_aos_direct_kernel" in k or "step_fused_tiled_kernel" in k}
    assert max(tiled.values()) <= 128, max(tiled.items(), key=lambda kv: kv[1])        # the step kernels proper: 4 waves per SIMD


def test_headline_kernel_is_what_the_measurements_describe(assembly):
    desc = {k: d for k, d in _descriptors(assembly).items() if HEADLINE in k}
    assert len(desc) == 1
    d = next(iter(desc.values()))
    assert int(d["next_free_vgpr"]) <= 112                    # 107 under amdgpu_waves_per_eu(1, 4); built for 5 waves it needs 95
    assert int(d.get("group_segment_fixed_size", 0)) == 0      # no LDS
    body = _body(assembly, HEADLINE)
    ins = re.findall(r"^\s+([a-z][a-z0-9_]+)", body, re.M)
    loads = [l for l in body.splitlines() if re.match(r"\s+global_load_", l)]
    stores = [l for l in body.splitlines() if re.match(r"\s+global_store_", l)]
    assert len(loads) == 28 and all(l.rstrip().endswith(" nt") for l in loads)           # 21 dword + 7 ushort, p_x / p_y never loaded
    assert sum("global_load_ushort" in l for l in loads) == 7
    assert len(stores) == 6 and all(l.rstrip().endswith("sc0 sc1") for l in stores)       # write-through, not nt
    assert not any(i.startswith(("ds_", "scratch_", "buffer_")) for i in ins)
    valu = [i for i in ins if i.startswith("v_")]
    assert len(valu) <= 465, len(valu)                         # round 2: 520; after the round-3 diet: 460
    fp64 = [i for i in valu if i.endswith("_f64") or "_f64_" in i]
    assert len(fp64) <= 340
    # every field offset sits in the instruction's immediate: one address register per record, no 64-bit address arithmetic
    assert sum(i.startswith(("v_lshl_add_u64", "v_add_co", "v_addc_co")) for i in valu) == 0


def test_array_of_structs_kernel_keeps_nt_stores_and_row_accesses(assembly):
    body = _body(assembly, "wrench_aos_direct_kernelILb0ELb1ELb0E")
    stores = [l for l in body.splitlines() if re.match(r"\s+global_store_", l)]
    assert sum("global_store_dwordx3" in l for l in stores) == 2 and all(l.rstrip().endswith(" nt") for l in stores)
    loads = [l for l in body.splitlines() if re.match(r"\s+global_load_", l)]
    assert any("global_load_dwordx3" in l for l in loads) and any("global_load_dwordx4" in l for l in loads)


def test_committed_instruction_mix_is_current(assembly):
    """profiles/isa_mix.json (scripts/isa_mix.py) is what bench.py prices the compute-bound kernels with (the VALU-issue
    roofline of the resident closed loop): it must describe the code that is built."""
    import json
    from scripts import isa_mix
    committed = json.load(open(os.path.join(REPO, "profiles", "isa_mix.json")))["kernels"]
    assert committed == isa_mix.mix(assembly), "run `python scripts/isa_mix.py` and commit profiles/isa_mix.json"
    loop = next(v for k, v in committed.items() if k.startswith("resident closed loop, one step ("))
    assert loop["valu_total"] <= 530                          # the per-step loop of the resident kernel (round 3: 527 + 3)
