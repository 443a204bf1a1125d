#!/usr/bin/env python3
"""Developer GPU check: parity of every C-ABI entry point against the oracle on the golden
fixtures, then a timing sweep over kernel variants.  Writes gpurun_out/dev_check.log."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import hydro_oracle as ho                       # noqa: E402
from silver2_isaacsim_amd import scenes                     # noqa: E402
from silver2_isaacsim_amd.engine import HydroEngine         # noqa: E402

os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(REPO, "gpurun_out", "dev_check.log"), "w")


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + "\n"); LOG.flush()


dev = torch.device("cuda:0")
say("device", torch.cuda.get_device_name(0))


def soa(x):
    return torch.from_numpy(scenes.to_soa(x)).to(dev)


def parity(name):
    z = np.load(os.path.join(REPO, "tests", "golden", f"{name}.npz"))
    st, pv, pr = z["state"], z["prev"], z["params"]
    rho, g, dt = float(z["rho"]), float(z["g"]), float(z["dt"])
    n = len(st)
    rf, rt, aux = ho.step_wrench(st, pv, pr, rho, g, dt)
    half = name == "c5"
    eng = HydroEngine(n, 0, rho, g)
    eng.set_params(pr, "f16" if half else "f32")
    S, P = soa(st), soa(pv)
    for vec in (1, 2, 4):
        eng.set_tuning(vec)
        out = eng.step_wrench(S, dt, prev=P)
        torch.cuda.synchronize()
        o = out.cpu().numpy().T
        e = ho.wrench_error(o[:, :3], o[:, 3:], rf, rt, pr, rho, g)
        say(f"{name} ext  vec={vec} max_err {e.max():.3e} n>1e-5 {(e > 1e-5).sum()} nan {np.isnan(o).sum()}")
    eng.set_tuning(0)
    eng.set_prev_velocity(pv)
    out = eng.step_wrench(S, dt)
    torch.cuda.synchronize()
    o = out.cpu().numpy().T
    e = ho.wrench_error(o[:, :3], o[:, 3:], rf, rt, pr, rho, g)
    pnow = eng.get_prev_velocity().cpu().numpy().T
    say(f"{name} own-prev max_err {e.max():.3e}; prev updated exactly: {np.array_equal(pnow, st[:, 7:13])}")
    # AoS entry
    eng.set_prev_velocity(pv)
    pos = torch.from_numpy(np.ascontiguousarray(st[:, 0:3])).to(dev)
    quat = torch.from_numpy(np.ascontiguousarray(st[:, [6, 3, 4, 5]])).to(dev)
    vel = torch.from_numpy(np.ascontiguousarray(st[:, 7:13])).to(dev)
    F, T = eng.step_wrench_aos(pos, quat, vel, dt)
    torch.cuda.synchronize()
    e = ho.wrench_error(F.cpu().numpy(), T.cpu().numpy(), rf, rt, pr, rho, g)
    say(f"{name} aos  max_err {e.max():.3e} bit-equal to SoA: {np.array_equal(F.cpu().numpy(), o[:, :3]) and np.array_equal(T.cpu().numpy(), o[:, 3:])}")
    # components
    acc = ((st[:, 7:13].astype(np.float64) - pv.astype(np.float64)) / dt).astype(np.float32)
    comps, ratio = eng.step_components(S, soa(acc))
    torch.cuda.synchronize()
    c = comps.cpu().numpy().T.reshape(n, 8, 3)
    ref = z["components"]
    scale = np.maximum(np.linalg.norm(ref, axis=2), 1e-3 * rho * g * pr[:, :3].prod(1)[:, None])
    ce = (np.linalg.norm(c - ref, axis=2) / scale)
    say(f"{name} comps max rel per field {np.round(ce[:, :6].max(0), 9)} cob/cop abs {np.abs(c[:, 6:] - ref[:, 6:]).max():.3e} ratio abs {np.abs(ratio.cpu().numpy() - z['ratio']).max():.3e}")
    # KE
    ke = eng.kinetic_energy(S, rotational=True)
    torch.cuda.synchronize()
    ref_lin = ho.kinetic_energy(st, pr, False)[0]
    ref_tot = ho.kinetic_energy(st, pr, True)[0]
    k = ke.cpu().numpy()
    say(f"{name} KE lin rel {abs(k[0] - ref_lin) / ref_lin:.3e} tot rel {abs(k.sum() - ref_tot) / ref_tot:.3e}")
    eng.close()


for nm in ("c2", "c5"):
    parity(nm)


def timing(n, half, sets=4, steps=40, warm=5):
    sc = scenes.scene_c4(n=min(n, 262144), seed=9)
    reps = (n + sc.n - 1) // sc.n
    st = np.tile(sc.state, (reps, 1))[:n]; pv = np.tile(sc.prev, (reps, 1))[:n]; pr = np.tile(sc.params, (reps, 1))[:n]
    engs, S, P, O = [], [], [], []
    for k in range(sets):
        e = HydroEngine(n, 0, sc.rho, sc.g); e.set_params(pr, "f16" if half else "f32"); engs.append(e)
        S.append(soa(st)); P.append(soa(pv)); O.append(torch.empty((6, n), device=dev))
    bytes_per = 130 if half else 144
    stream = torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        _timing_body(n, half, sets, steps, warm, sc, engs, S, P, O, bytes_per)
    for e in engs:
        e.close()


def _timing_body(n, half, sets, steps, warm, sc, engs, S, P, O, bytes_per):
    for vec in (1, 2, 4):
        for e in engs:
            e.set_tuning(vec)
        for k in range(warm):
            engs[k % sets].step_wrench(S[k % sets], sc.dt, out=O[k % sets], prev=P[k % sets])
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for k in range(steps):
            engs[k % sets].step_wrench(S[k % sets], sc.dt, out=O[k % sets], prev=P[k % sets])
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / steps
        say(f"n={n} half={half} vec={vec}: {ms * 1e3:.1f} us/step  {n / ms * 1e3:.3e} body-steps/s  "
            f"{n * bytes_per / ms / 1e6:.1f} GB/s  ({n * bytes_per / ms / 1e6 / 8000 * 100:.1f}% of 8 TB/s)")


for n in (4096, 19456, 262144, 1048576, 4194304):
    timing(n, False)
timing(1048576, True)
timing(4194304, True)
say("done")
