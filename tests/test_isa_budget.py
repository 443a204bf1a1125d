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
    assert len(valu) <= 446, len(valu)                         # round 2: 520; round-3 diet: 462; round 6 (scalar tile addressing, bit flags): 444
    fp64 = [i for i in valu if i.endswith("_f64") or "_f64_" in i]
    assert len(fp64) <= 340
    # The tile index is wave-uniform (round 6): record bases are SCALAR adds, every field offset an instruction immediate, and
    # the per-lane addressing of a body is lane * 4, lane * 2 and ONE 64-bit add for the six write-through stores (atomic
    # stores take no scalar base) - no 24-bit multiplies, no add-shifts per record
    addressing = [i for i in valu if i.startswith(("v_lshl_add_u64", "v_add_co", "v_addc_co", "v_mul_u32_u24", "v_mad_u32_u24", "v_add_lshl", "v_lshl_or"))]
    assert addressing == ["v_lshl_add_u64"], addressing
    assert sum(i == "v_readfirstlane_b32" for i in valu) == 1 and "s_mul_i32" in ins


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


# ---- the kinetic-energy reduction's memory protocol (hydro_kernels.hip "kinetic-energy reduction") -------------------------
# Relaxed device-scope atomics + s_waitcnt stand in for release / acquire (a release fence is a buffer_wbl2: 30 us per launch
# at 1 M bodies).  That is only correct while the compiler keeps THIS order, so the order is pinned here, per kernel that
# carries the reduction: a compiler bump that sinks a published value below its ticket fails the CPU suite.
KE_KERNELS = {
    "stand-alone, rotational": "ke_kernelILb1E",
    "stand-alone, translational": "ke_kernelILb0E",
    "tiled wrench, sampling <256, f32, caller prev, streaming, KE, Numba>": "wrench_tiled_kernelILi256ELb0ELb0ELb1ELb1ELb0E",
    "tiled wrench, sampling <256, f16, engine prev, temporal, KE, Numba>": "wrench_tiled_kernelILi256ELb1ELb1ELb0ELb1ELb0E",
    "fused step, sampling <f32, streaming, explicit, KE, Numba>": "step_fused_tiled_kernelILb0ELb1ELb0ELb1ELb0E",
    "resident multi-step, sampling <f32, temporal, explicit, KE, Numba>": "step_fused_multi_tiled_kernelILb0ELb0ELb0ELb1ELb0E",
}


def _protocol_events(body):
    """The instructions the protocol is made of, in program order: ('publish' | 'wait' | 'ticket' | 'inv' | 'fetch' | ...)."""
    ev = []
    for line in body.splitlines():
        l = line.strip()
        if re.match(r"global_store_dwordx2 .* sc1$", l):
            ev.append("publish")                                     # write-through fp64 store (partial, class sum, poison, result)
        elif re.match(r"global_store_dword .* sc1$", l):
            ev.append("reset")                                       # a finisher zeroes its counter
        elif l.startswith("global_atomic_add"):
            ev.append("ticket")
        elif l.startswith("s_waitcnt") and "vmcnt(0)" in l:
            ev.append("wait")
        elif l.startswith("buffer_inv"):
            assert l == "buffer_inv sc1", l
            ev.append("inv")
        elif re.match(r"global_load_dwordx2 .* sc1$", l):
            ev.append("fetch")
        elif l.startswith(("buffer_wbl2", "global_wb", "buffer_gl")):
            ev.append("writeback:" + l)
        elif l.startswith("global_atomic"):
            ev.append("other-atomic:" + l)
    return ev


@pytest.mark.parametrize("which", sorted(KE_KERNELS))
def test_kinetic_energy_reduction_keeps_its_memory_order(assembly, which):
    body = _body(assembly, KE_KERNELS[which])
    ev = _protocol_events(body)
    assert not [e for e in ev if e.startswith(("writeback", "other-atomic"))], ev      # no L2 write-back fence, integer tickets only
    assert "v_mfma" not in body
    tickets = [i for i, e in enumerate(ev) if e == "ticket"]
    assert len(tickets) == 2, ev                                     # the class counter, then the top counter
    for t in tickets:
        # producer side: published values -> s_waitcnt vmcnt(0) -> the ticket, nothing published in between
        before = ev[:t]
        assert before[-1] == "wait", (which, before[-4:])
        assert "publish" in before[:-1] and before[:-1][-1] in ("publish", "reset"), (which, before[-4:])
        # consumer side: the ticket's value is waited for, then the non-coherent caches are invalidated, THEN others' values are read
        after = ev[t + 1:]
        assert after[0] == "wait" and after[1] == "inv", (which, after[:4])
        nxt = [e for e in after[2:] if e != "wait"]
        assert nxt and nxt[0] == "fetch", (which, after[:6])
    # no value of another block is fetched before the first ticket has been drawn and its invalidate issued
    assert "fetch" not in ev[:tickets[0] + 3]
    # the poison of `out` (block 0) precedes the first ticket; the result is published, the top counter zeroed, and the LAST thing a
    # class finisher does (round 6) is to leave NaNs in the partials it has consumed - one loop body, a pair, behind everything
    # this launch waits for
    assert ev[:tickets[0]].count("publish") == 4                     # block 0's two NaNs, then the block's pair
    assert ev[-5:] == ["publish", "publish", "reset", "publish", "publish"], ev[-7:]
