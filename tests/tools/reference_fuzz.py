#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference): fuzz the oracle AND the kernel arithmetic (host instantiation of
csrc/hydro_body.h) against the REFERENCE ITSELF on populations in which special values are injected field by field - exact
zeros, thresholds of the model (speeds of 1e-6 and 0.2, angular likewise), axis-aligned and quantised vectors, cube
rotations, non-unit quaternions, tiny and zero dimensions, p_z on exact ties - on top of continuous random bodies.
    python3 -B tests/tools/reference_fuzz.py [bodies] [seed]
Prints how many bodies disagree (oracle vs reference at 1e-9 of the body's scale; kernel arithmetic vs reference at the
1e-5 gate, centres at half an fp32 ulp + 1e-6) and the worst cases."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
import numpy as np  # noqa: E402

import make_golden as mg  # noqa: E402  (imports the reference)
import populations  # noqa: E402
from oracle import hydro_oracle as ho  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
S = populations._S
quats = np.concatenate([populations.TIE_QUATS, [(0, 0, 0, 0), (0, 0, 0, 2), (0, 0, 0, -1), (S, 0, S, 0)]])


def special(col, values, p):
    m = rng.uniform(size=len(col)) < p
    col[m] = rng.choice(values, m.sum())


state = np.zeros((n, 13)); params = np.zeros((n, 11))
dims = np.exp(rng.uniform(np.log(0.05), np.log(2.0), (n, 3)))
for k in range(3):
    special(dims[:, k], [0.25, 0.5, 1.0, 2.0, 1e-7, 0.0, 3e-7], 0.35)
q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
m = rng.uniform(size=n) < 0.5
q[m] = quats[rng.integers(0, len(quats), m.sum())]
v = rng.normal(0, 1.0, (n, 3)) * np.exp(rng.uniform(-8, 1, (n, 1)))
w = rng.normal(0, 0.5, (n, 3)) * np.exp(rng.uniform(-8, 1, (n, 1)))
for arr in (v, w):
    for k in range(3):
        special(arr[:, k], [0.0, -0.0, 0.2, -0.2, 1e-6, 0.125, -0.5, 1.0, 5e-7, 2e-6], 0.4)
state[:, 3:7] = q; state[:, 7:10] = v; state[:, 10:13] = w
state[:, 0:2] = rng.uniform(-50, 50, (n, 2))
state = state.astype(np.float32).astype(np.float64); dims = dims.astype(np.float32).astype(np.float64)
x, y, z, ww = state[:, 3:7].T
row2 = np.stack([2 * (x * z - ww * y), 2 * (y * z + ww * x), 1.0 - 2 * (x * x + y * y)], axis=1)
e = 0.5 * dims * row2
extent = np.abs(e).sum(axis=1)
kind = rng.integers(0, 7, n)
pz = np.select([kind == 0, kind == 1, kind == 2, kind == 3, kind == 4], [-extent, extent, np.zeros(n), -e[:, 0], e[:, 2]],
               extent * rng.uniform(-1.5, 1.5, n))
state[:, 2] = pz
coeffs = np.array([1.2, 0.8, 300.0, 150.0, 1.0, 0.05, 0.02]) * np.exp(rng.uniform(np.log(0.5), np.log(2.0), (n, 7)))
for k in range(7):
    special(coeffs[:, k], [0.0], 0.15)
mass = np.where(rng.uniform(size=n) < 0.1, 1.0, 0.5 * 1025.0 * np.maximum(dims.prod(axis=1), 1e-3))
params = np.concatenate([dims, coeffs, mass[:, None]], axis=1)
state, params = state.astype(np.float32), params.astype(np.float32)
prev = (state[:, 7:13].astype(np.float64) - rng.integers(-4, 5, (n, 6)) / 64.0).astype(np.float32)
prev[rng.uniform(size=n) < 0.3] = 0.0
rho, g, dt = populations.RHO, populations.G, populations.DT
accel = (state[:, 7:13].astype(np.float64) - prev.astype(np.float64)) / dt

with np.errstate(all="ignore"):
    comps, ratio, rest = mg.reference_batch(state, accel, params, rho, g)
    net_f, net_t = mg.reference_behavior_batch(state, prev, params, rho, g, dt)
    o = ho.solve_components(state, accel, params.astype(np.float64), rho, g)
    rf, rt, aux = ho.step_wrench(state, prev, params, rho, g, dt)
finite = np.isfinite(comps).all(axis=(1, 2)) & np.isfinite(net_f).all(axis=1) & np.isfinite(net_t).all(axis=1)
print(f"{n} bodies (seed {seed}): reference finite on {finite.sum()}, rest-completed {rest.sum()}, dry {(ratio == 0).sum()}, full {(ratio == 1).sum()}")
scale = np.maximum(1.0, np.abs(comps[:, :6]).max(axis=(1, 2)))
d_or = np.max([np.abs(o[f] - comps[:, k]).max(axis=1) / (scale if k < 6 else np.maximum(1.0, np.abs(comps[:, k]).max(axis=1))) for k, f in enumerate(ho.COMPONENT_FIELDS)], axis=0)
d_or = np.maximum(d_or, np.abs(o["ratio"] - ratio))
bad = np.where(finite & (d_or > 1e-9))[0]
print(f"oracle vs reference (components, ratio): {len(bad)} bodies above 1e-9, max {np.nanmax(np.where(finite, d_or, 0)):.3e}")
for i in bad[:5]:
    print("   ", i, state[i], params[i], d_or[i])
with np.errstate(all="ignore"):
    err_o = ho.wrench_error(rf, rt, net_f, net_t, params, rho, g)
_vol0 = params[:, :3].astype(np.float64).prod(axis=1) == 0.0
_big = np.maximum(np.abs(np.concatenate([net_f, net_t], axis=1)).max(axis=1), 1e-30)
err_o = np.where(_vol0, np.abs(np.concatenate([rf - net_f, rt - net_t], axis=1)).max(axis=1) / _big, err_o)
print(f"oracle vs reference (net wrench of _apply_behavior): max {np.nanmax(np.where(finite, err_o, 0)):.3e}")

lib = ctypes.CDLL(os.path.join(REPO, "tests", "host_emul", "libemul.so"))
fp = ctypes.POINTER(ctypes.c_float)
f = np.empty((n, 3), np.float32); t = np.empty((n, 3), np.float32); r = np.empty(n, np.float32)
lib.emul_wrench(ctypes.c_int64(n), state.ctypes.data_as(fp), prev.ctypes.data_as(fp), params.ctypes.data_as(fp), ctypes.c_double(rho),
                ctypes.c_double(g), ctypes.c_double(dt), f.ctypes.data_as(fp), t.ctypes.data_as(fp), r.ctypes.data_as(fp))
with np.errstate(all="ignore"):
    err = ho.wrench_error(f, t, net_f, net_t, params, rho, g)
# bodies of zero volume: the metric's floor (1e-3 rho g V) is 0 and a zero torque gives 0 / 0 - for those, the difference
# relative to the body's largest wrench component
vol0 = params[:, :3].astype(np.float64).prod(axis=1) == 0.0
big = np.maximum(np.abs(np.concatenate([net_f, net_t], axis=1)).max(axis=1), 1e-30)
alt = np.abs(np.concatenate([f - net_f, t - net_t], axis=1)).max(axis=1) / big
err = np.where(vol0, alt, err)
badk = np.where(finite & ~(err <= 1e-5))[0]
print(f"kernel arithmetic vs reference (net wrench): {len(badk)} bodies above the 1e-5 gate, max {np.nanmax(np.where(finite, err, 0)):.3e}")
for i in badk[:8]:
    print("   ", i, "state", state[i], "params", params[i], "got", f[i], t[i], "ref", net_f[i], net_t[i], err[i])
if len(badk):
    # is it the kernels or the reference?  The reference forms WORLD-space centres and subtracts the position again
    # (hydrodynamics_behavior.py:212-214): at |p_xy| ~ 50 m that costs 1e-14 m of a lever arm - nothing, unless the body is
    # 1e-7 m small and its torque a 300-fold cancellation.  Re-run the reference with p_x = p_y = 0 (the wrench does not
    # depend on them): bodies that then agree were the reference's own rounding.
    st0 = state[badk].copy(); st0[:, 0:2] = 0.0
    with np.errstate(all="ignore"):
        nf0, nt0 = mg.reference_behavior_batch(st0, prev[badk], params[badk], rho, g, dt)
        e0 = ho.wrench_error(f[badk], t[badk], nf0, nt0, params[badk], rho, g)
    print(f"    of those, against the reference re-run at p_x = p_y = 0: {int((~(e0 <= 1e-5)).sum())} above the gate, max {np.nanmax(e0):.3e}; "
          f"smallest dimension among them {params[badk, :3].min(axis=1).max():.1e} m")
acc32 = accel.astype(np.float32)
out = np.empty((n, 8, 3), np.float32); rr = np.empty(n, np.float32)
lib.emul_components(ctypes.c_int64(n), state.ctypes.data_as(fp), acc32.ctypes.data_as(fp), params.ctypes.data_as(fp), ctypes.c_double(rho),
                    ctypes.c_double(g), out.ctypes.data_as(fp), rr.ctypes.data_as(fp))
worst = 0; nb = 0
for k in (6, 7):
    want = comps[:, k]
    tol = 0.5 * np.spacing(np.abs(want).astype(np.float32)).astype(np.float64) * (1 + 1e-6) + 1e-6
    b = finite & (np.abs(out[:, k] - want) > tol).any(axis=1)
    nb += b.sum()
    for i in np.where(b)[0][:4]:
        print("    centre", k, i, state[i], params[i], out[i, k], want[i])
print(f"kernel arithmetic vs reference (cob / cop): {nb} bodies outside half an fp32 ulp + 1e-6")
nf = 0
for k in (0, 1, 2, 3):                                            # buoyancy, drag force, lift, drag torque: independent of the accelerations
    b = finite & (np.abs(out[:, k] - comps[:, k]).max(axis=1) > 1e-6 * scale)
    nf += b.sum()
    for i in np.where(b)[0][:4]:
        print("    component", k, i, state[i], params[i], out[i, k], comps[i, k])
print(f"kernel arithmetic vs reference (buoyancy, drag force, lift, drag torque): {nf} bodies beyond 1e-6 of the body's largest term")
print(f"ratio: max |diff| {np.nanmax(np.where(finite, np.abs(rr - ratio), 0)):.3e}")
