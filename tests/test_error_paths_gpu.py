"""Argument errors of the tiled / fused / array-of-structs / layout entry points (SURVEY.md 8b "Errors" row): each
must come back as HYDRO_E_ARG with a message in hydro_last_error, launch nothing, and leave the engine usable.
Called through the raw C ABI (ctypes), the way a foreign host would."""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from silver2_isaacsim_amd import _native as nat
from silver2_isaacsim_amd import scenes
from silver2_isaacsim_amd.engine import HydroEngine

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
E_ARG = -1
N = 1000                      # 16 tiles
ST, PV, WR = 13 * 64, 6 * 64, 6 * 64


@pytest.fixture()
def rig(native_built):
    fx = load_golden("c4")
    eng = HydroEngine(8192, DEV, float(fx["rho"]), float(fx["g"]))
    eng.set_params(fx["params"][:N])
    state = torch.from_numpy(scenes.to_tiled(fx["state"][:N])).to(DEV)
    prev = torch.from_numpy(scenes.to_tiled(fx["prev"][:N])).to(DEV)
    out = eng.alloc_tiled(6, N)
    sentinel = torch.full_like(out, 7.0)
    out.copy_(sentinel)
    yield eng, state, prev, out, float(fx["dt"])
    # nothing was launched by any failing call: the output buffer still holds the sentinel ...
    torch.cuda.synchronize()
    assert torch.equal(out, sentinel)
    # ... and the engine still works
    good = eng.step_wrench_tiled(state, N, float(fx["dt"]), prev=prev)
    torch.cuda.synchronize()
    assert torch.isfinite(good).all()
    eng.close()


def _expect_arg(eng, rc, needle):
    assert rc == E_ARG, nat.STATUS_NAMES.get(rc, rc)
    msg = eng._lib.hydro_last_error(eng._h).decode()
    assert needle in msg, msg


def p(t, byte_offset=0):
    return ctypes.c_void_p(t.data_ptr() + byte_offset)


def test_tiled_stride_and_alignment_errors(rig):
    eng, state, prev, out, dt = rig
    L, h = eng._lib, eng._h
    step = lambda st=p(state), ss=ST, pv=p(prev), ps=PV, o=p(out), os_=WR: L.hydro_step_wrench_tiled(h, N, st, ss, pv, ps, dt, o, os_, None)   # noqa: E731
    _expect_arg(eng, step(ss=ST - 64), "tile stride")                 # stride < fields * 64
    _expect_arg(eng, step(ps=PV - 4), "tile stride")
    _expect_arg(eng, step(os_=WR - 64), "tile stride")
    _expect_arg(eng, step(ss=1 << 24), "tile stride")                 # stride >= 2^24 (24-bit multiplies in the kernel)
    _expect_arg(eng, step(ss=ST + 2), "tile stride")                  # not a multiple of 4 floats
    _expect_arg(eng, step(st=p(state, 4)), "16-byte aligned")         # misaligned base pointers
    _expect_arg(eng, step(pv=p(prev, 8)), "16-byte aligned")
    _expect_arg(eng, step(o=p(out, 4)), "16-byte aligned")
    _expect_arg(eng, step(st=None), "null state")
    _expect_arg(eng, step(o=None), "null wrench")
    # tiled buffer of 4 GiB or more: 32-bit byte offsets in the kernels.  Argument check only - the pointer is never
    # dereferenced: 4 160 bodies = 65 tiles x (2^24 - 4) floats x 4 B >= 2^32
    eng.set_params(np.tile(load_golden("c4")["params"], (2, 1))[:4160])
    rc = L.hydro_step_wrench_tiled(h, 4160, p(state), (1 << 24) - 4, p(prev), PV, dt, p(out), WR, None)
    _expect_arg(eng, rc, "4 GiB")
    rc = L.hydro_step_wrench_tiled(h, N, p(state), ST, p(prev), PV, 0.0, p(out), WR, None)
    _expect_arg(eng, rc, "dt must be > 0")


def test_fused_step_errors(rig):
    eng, state, prev, out, dt = rig
    L, h = eng._lib, eng._h
    old = torch.zeros_like(state)
    pv = p(old, 7 * 64 * 4)

    def fused(st=p(state), ss=ST, pvp=pv, ps=ST, so=p(old), sos=ST, w=p(out), ws=WR):
        return L.hydro_step_fused_tiled(h, N, st, ss, pvp, ps, dt, so, sos, w, ws, 0, None)
    _expect_arg(eng, fused(so=p(state)), "must not alias state")      # state_out == state
    _expect_arg(eng, fused(ss=ST - 4), "tile stride")
    _expect_arg(eng, fused(ps=PV - 4), "tile stride")
    _expect_arg(eng, fused(sos=ST - 64), "tile stride")
    _expect_arg(eng, fused(ws=WR + 1), "tile stride")
    _expect_arg(eng, fused(sos=1 << 24), "tile stride")
    _expect_arg(eng, fused(pvp=None), "null prev")
    _expect_arg(eng, fused(so=None), "null state_out")
    _expect_arg(eng, fused(st=p(state, 4)), "16-byte aligned")
    _expect_arg(eng, fused(so=p(old, 8)), "16-byte aligned")
    rc = L.hydro_integrate_tiled(h, N, p(state), ST - 64, p(out), WR, dt, p(old), ST, None)
    _expect_arg(eng, rc, "tile stride")
    rc = L.hydro_integrate_tiled(h, N, p(state), ST, p(out), WR, dt, p(old, 4), ST, None)
    _expect_arg(eng, rc, "16-byte aligned")
    torch.cuda.synchronize()
    assert not old.any()                                              # nothing was written


def test_array_of_structs_and_layout_errors(rig):
    eng, state, prev, out, dt = rig
    L, h = eng._lib, eng._h
    pos = torch.zeros((N + 4, 3), device=DEV); quat = torch.zeros((N + 4, 4), device=DEV); quat[:, 0] = 1
    vel = torch.zeros((N + 4, 6), device=DEV)
    f = torch.full((N + 4, 3), 7.0, device=DEV); t = torch.full((N + 4, 3), 7.0, device=DEV)

    def aos(n=N, a=p(pos), b=p(quat), c=p(vel), d=p(f), e=p(t), dt_=dt):
        return L.hydro_step_wrench_aos(h, n, a, b, 0, c, dt_, d, e, None)
    _expect_arg(eng, aos(a=p(pos, 4)), "16-byte aligned")             # every tensor is read / written in 16-byte chunks
    _expect_arg(eng, aos(b=p(quat, 8)), "16-byte aligned")
    _expect_arg(eng, aos(c=p(vel, 4)), "16-byte aligned")
    _expect_arg(eng, aos(d=p(f, 12)), "16-byte aligned")
    _expect_arg(eng, aos(e=p(t, 4)), "16-byte aligned")
    _expect_arg(eng, aos(a=None), "null tensor")
    _expect_arg(eng, aos(dt_=-1.0), "dt must be > 0")
    _expect_arg(eng, aos(n=(1 << 26) + 1), "2^26")                    # checked before anything else about n
    _expect_arg(eng, aos(n=-1), "n out of range")
    torch.cuda.synchronize()
    assert (f == 7.0).all() and (t == 7.0).all()
    # layout edges
    tiled_state = eng.alloc_tiled(13, N)
    _expect_arg(eng, L.hydro_pack_state_aos(h, N, p(pos, 4), p(quat), 0, p(vel), p(tiled_state), ST, None), "16-byte aligned")
    _expect_arg(eng, L.hydro_pack_state_aos(h, N, p(pos), p(quat), 0, p(vel), p(tiled_state), ST - 64, None), "tile stride")
    _expect_arg(eng, L.hydro_pack_state_aos(h, N, p(pos), None, 0, p(vel), p(tiled_state), ST, None), "null tensor")
    _expect_arg(eng, L.hydro_pack_state_aos(h, 8193, p(pos), p(quat), 0, p(vel), p(tiled_state), ST, None), "n out of range")
    _expect_arg(eng, L.hydro_unpack_wrench_aos(h, N, p(out), WR, p(f, 4), p(t), None), "16-byte aligned")
    _expect_arg(eng, L.hydro_unpack_wrench_aos(h, N, p(out), WR - 4, p(f), p(t), None), "tile stride")
    _expect_arg(eng, L.hydro_unpack_wrench_aos(h, N, p(out), WR, None, p(t), None), "null tensor")
    soa = torch.zeros((13, N), device=DEV)
    tab = nat.pointer_table([soa.data_ptr() + k * N * 4 for k in range(13)])
    _expect_arg(eng, L.hydro_repack(h, N, 13, tab, p(tiled_state), ST - 64, 1, None), "tile stride")
    _expect_arg(eng, L.hydro_repack(h, N, 25, tab, p(tiled_state), 25 * 64, 1, None), "bad arguments")      # > 24 fields
    _expect_arg(eng, L.hydro_repack(h, N, 13, tab, p(tiled_state, 4), ST, 1, None), "16-byte aligned")
    bad = nat.pointer_table([soa.data_ptr() + k * N * 4 if k != 5 else 0 for k in range(13)])
    _expect_arg(eng, L.hydro_repack(h, N, 13, bad, p(tiled_state), ST, 1, None), "null field pointer")
    _expect_arg(eng, L.hydro_kinetic_energy_tiled(h, N, p(state), ST - 64, 0, p(torch.zeros(2, dtype=torch.float64, device=DEV)), None), "tile stride")
    torch.cuda.synchronize()
    assert not tiled_state.any() and (f == 7.0).all()


# ---- the kinetic-energy reduction after a launch that did not finish -----------------------
def _ke_rig(n=100003, seed=31):
    sc = scenes.scene_c4(n=n, seed=seed)
    eng = HydroEngine(sc.n, DEV, sc.rho, sc.g)
    eng.set_params(sc.params)
    st = torch.from_numpy(scenes.to_tiled(sc.state)).to(DEV)
    return sc, eng, st


def test_kinetic_energy_never_returns_a_stale_pair_and_rearms():
    """The reduction's ticket counters are zero between launches that finish.  A launch that was cut short (device reset,
    aborted graph) leaves some of them non-zero, and no later launch then draws the "last" ticket.  hydro_debug_ke_fault puts
    a counter into that state.  What must hold: (1) the next result is NaN - block 0 poisons the output before any ticket
    is drawn - never the pair of an earlier launch; (2) hydro_ke_rearm restores the reduction, same bits as before;
    (3) a fault that the library has SEEN (a failed HIP call marks the handle) is repaired by the next launch itself."""
    sc, eng, st = _ke_rig()
    L, h = eng._lib, eng._h
    good = eng.kinetic_energy(st, rotational=True).clone()
    lin, rot = scenes.kinetic_energy_fp64(sc.state, sc.params)
    assert good[0].item() == pytest.approx(lin, rel=1e-12) and good[1].item() == pytest.approx(rot, rel=1e-12)
    out = good.clone()                                                # holds a perfectly plausible pair from an earlier launch
    # (1) the top counter at 64: the class finishers draw 64 .. 127, none of them the last ticket (63) - no final sum
    assert L.hydro_debug_ke_fault(h, 0, 64, 0) == 0
    eng.kinetic_energy(st, rotational=True, out=out)
    torch.cuda.synchronize()
    assert torch.isnan(out).all()                                     # visible, not stale
    # (2) the documented recovery
    eng.ke_rearm()
    eng.kinetic_energy(st, rotational=True, out=out)
    torch.cuda.synchronize()
    assert torch.equal(out, good)
    assert torch.equal(eng.kinetic_energy(st, rotational=True), good)     # and the launch left the counters at zero again
    # (3) class counter 5 left at 3 by a launch the library saw fail: the next launch zeroes the counters first
    assert L.hydro_debug_ke_fault(h, 1 + 5, 3, 1) == 0
    eng.kinetic_energy(st, rotational=True, out=out)
    torch.cuda.synchronize()
    assert torch.equal(out, good)
    # the same through the sampling step kernels (they share the scratch and the counters)
    prev = torch.from_numpy(scenes.to_tiled(sc.prev)).to(DEV)
    ke = torch.zeros(2, dtype=torch.float64, device=DEV)
    assert L.hydro_debug_ke_fault(h, 0, 64, 0) == 0
    eng.step_wrench_tiled(st, sc.n, sc.dt, prev=prev, ke_out=ke, rotational=True)
    torch.cuda.synchronize()
    assert torch.isnan(ke).all()
    assert L.hydro_debug_ke_fault(h, 0, 64, 1) == 0                   # (seen by the library this time)
    eng.step_wrench_tiled(st, sc.n, sc.dt, prev=prev, ke_out=ke, rotational=True)
    torch.cuda.synchronize()
    assert torch.equal(ke, good)
    assert L.hydro_debug_ke_fault(h, 65, 1, 0) == E_ARG and L.hydro_debug_ke_fault(h, -1, 1, 0) == E_ARG
    # (4) ADVICE r5: a class counter at 0 < k < members that the library did NOT see, and a scene that changed since the
    # last launch that finished: class 5 is added three blocks early, and what it finds in the three unwritten slots are
    # the NaNs its own finisher left there after the last good launch - NaN (or, if those blocks happened to arrive in time,
    # the correct bits of the NEW state), never a finite pair mixing old and new bodies
    faster = sc.state.copy(); faster[:, 7:13] *= 1.5
    st2 = torch.from_numpy(scenes.to_tiled(faster)).to(DEV)
    want2 = eng.kinetic_energy(st2, rotational=True).clone()
    assert not torch.equal(want2, good)
    eng.kinetic_energy(st, rotational=True, out=out)                  # the last launch that finished saw the OLD state
    assert L.hydro_debug_ke_fault(h, 1 + 5, 3, 0) == 0
    eng.kinetic_energy(st2, rotational=True, out=out)
    torch.cuda.synchronize()
    assert torch.isnan(out).all() or torch.equal(out, want2), out
    eng.ke_rearm()
    assert torch.equal(eng.kinetic_energy(st2, rotational=True), want2)
    eng.close()


def test_the_fault_injector_is_refused_without_the_environment_switch():
    """hydro_debug_ke_fault is exported (one build), but does nothing - HYDRO_E_STATE and a message - unless
    HYDRO_ENABLE_TEST_HOOKS=1 was in the environment when the library was loaded.  This process has it (conftest.py); a child
    without it is refused and its engine keeps working."""
    import subprocess
    import sys
    from conftest import REPO
    child = r'''
import os, sys, json, torch
sys.path.insert(0, os.environ["HYDRO_REPO"])
from silver2_isaacsim_amd import scenes
from silver2_isaacsim_amd.engine import HydroEngine
sc = scenes.scene_c4(n=4096, seed=1)
eng = HydroEngine(sc.n, "cuda:0", sc.rho, sc.g); eng.set_params(sc.params)
st = torch.from_numpy(scenes.to_tiled(sc.state)).to("cuda:0")
good = eng.kinetic_energy(st, rotational=True).clone()
rc = eng._lib.hydro_debug_ke_fault(eng._h, 0, 64, 0)
msg = eng._lib.hydro_last_error(eng._h).decode()
after = eng.kinetic_energy(st, rotational=True)
print(json.dumps({"rc": rc, "msg": msg, "same": bool(torch.equal(after, good))}))
'''
    import json
    env = {k: v for k, v in os.environ.items() if k != "HYDRO_ENABLE_TEST_HOOKS"}
    env["HYDRO_REPO"] = REPO
    res = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["rc"] == -5 and "HYDRO_ENABLE_TEST_HOOKS" in d["msg"] and d["same"] is True


def test_kinetic_energy_launches_on_two_streams_are_ordered_by_the_library():
    """Two reductions of one engine in flight at once would draw each other's tickets.  The library orders a launch on a
    stream other than the previous one's behind it (an event wait on the device): alternating two streams without any
    synchronisation by the caller gives the bits of the single-stream result every time."""
    sc, eng, st = _ke_rig(n=1048576 // 4, seed=32)
    good = eng.kinetic_energy(st, rotational=True).clone()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)]
    outs = [torch.zeros(2, dtype=torch.float64, device=DEV) for _ in range(40)]
    for k, o in enumerate(outs):
        eng.kinetic_energy(st, rotational=True, out=o, stream=streams[k % 2])
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, good)
    eng.close()


def test_plain_soa_step_validates_before_it_allocates():
    """hydro_step_wrench keeps the previous velocity in an engine-owned plain-SoA copy that is allocated on first use:
    a call that is going to be refused must be refused BEFORE that (n beyond the parameters, bad dt, null tables)."""
    sc = scenes.scene_c4(n=4096, seed=33)
    eng = HydroEngine(1 << 20, DEV, sc.rho, sc.g)                     # 82 B per body of plain-SoA copies = 86 MB if they came
    eng.set_params(sc.params)
    L, h = eng._lib, eng._h
    S = torch.from_numpy(scenes.to_soa(sc.state)).to(DEV)
    W = torch.empty((6, sc.n), device=DEV)
    tab_s, tab_w = eng._table(S, 13), eng._table(W, 6)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(DEV)[0]
    assert L.hydro_step_wrench(h, sc.n + 1, tab_s, sc.dt, tab_w, None) == -5          # HYDRO_E_STATE: parameters not set for n bodies
    assert L.hydro_step_wrench(h, sc.n, tab_s, 0.0, tab_w, None) == E_ARG
    assert L.hydro_step_wrench(h, sc.n, None, sc.dt, tab_w, None) == E_ARG
    assert L.hydro_step_wrench(h, (1 << 20) + 1, tab_s, sc.dt, tab_w, None) == E_ARG
    holes = nat.pointer_table([S[f].data_ptr() if f != 4 else 0 for f in range(13)])
    assert L.hydro_step_wrench(h, sc.n, holes, sc.dt, tab_w, None) == E_ARG
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(DEV)[0] == free0                   # nothing was allocated by any of them
    assert L.hydro_step_wrench(h, sc.n, tab_s, sc.dt, tab_w, None) == 0              # the good call does allocate, once
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(DEV)[0] < free0 and torch.isfinite(W).all()
    eng.close()
