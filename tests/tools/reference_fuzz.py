#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference): fuzz the oracle AND the kernel arithmetic (host instantiation of
csrc/hydro_body.h) against the REFERENCE ITSELF on populations in which special values are injected field by field - exact
zeros, thresholds of the model (speeds of 1e-6 and 0.2, angular likewise), axis-aligned and quantised vectors, cube
rotations, non-unit quaternions, tiny and zero dimensions, p_z on exact ties - on top of continuous random bodies.
    python3 -B tests/tools/reference_fuzz.py [bodies] [seed]
Prints how many bodies disagree (oracle vs reference at 1e-9 of the body's scale; kernel arithmetic vs reference at the
1e-5 gate, centres at half an fp32 ulp + 1e-6) and the worst cases.  `run(n, seed)` returns the same as a dict:
tests/test_reference_fuzz.py asserts it with the CPU suite (skipped where /root/reference is absent)."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for _p in (REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "golden")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import numpy as np  # noqa: E402

GATE = 1e-5
TINY_DIMENSION = 1e-6        # bodies thinner than this (the injected 0, 1e-7 and 3e-7 m) are classified separately, see run()


def population(n: int, seed: int):
    """(state, prev, params, rho, g, dt, accel): n bodies, fp32-exact, special values injected field by field."""
    import populations
    rng = np.random.default_rng(seed)
    S = populations._S
    quats = np.concatenate([populations.TIE_QUATS, [(0, 0, 0, 0), (0, 0, 0, 2), (0, 0, 0, -1), (S, 0, S, 0)]])

    def special(col, values, p):
        m = rng.uniform(size=len(col)) < p
        col[m] = rng.choice(values, m.sum())

    state = np.zeros((n, 13))
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
    return state, prev, params, rho, g, dt, accel


def _wrench_error(ho, f, t, ref_f, ref_t, params, rho, g):
    """SURVEY.md 8d's metric; for bodies of zero volume (floor 1e-3 rho g V = 0, a zero torque gives 0 / 0) the difference
    relative to the body's largest wrench component."""
    with np.errstate(all="ignore"):
        err = ho.wrench_error(f, t, ref_f, ref_t, params, rho, g)
    vol0 = params[:, :3].astype(np.float64).prod(axis=1) == 0.0
    big = np.maximum(np.abs(np.concatenate([ref_f, ref_t], axis=1)).max(axis=1), 1e-30)
    alt = np.abs(np.concatenate([f - ref_f, t - ref_t], axis=1)).max(axis=1) / big
    return np.where(vol0, alt, err)


def run(n: int = 20000, seed: int = 1, verbose: bool = False) -> dict:
    """Execute the reference, the oracle and the host instantiation of the kernel arithmetic on population(n, seed)."""
    import make_golden as mg                      # (imports the reference through the identity-njit stub)
    from oracle import hydro_oracle as ho
    say = print if verbose else (lambda *a, **k: None)
    state, prev, params, rho, g, dt, accel = population(n, seed)
    with np.errstate(all="ignore"):
        comps, ratio, rest = mg.reference_batch(state, accel, params, rho, g)
        net_f, net_t = mg.reference_behavior_batch(state, prev, params, rho, g, dt)
        o = ho.solve_components(state, accel, params.astype(np.float64), rho, g)
        rf, rt, aux = ho.step_wrench(state, prev, params, rho, g, dt)
    finite = np.isfinite(comps).all(axis=(1, 2)) & np.isfinite(net_f).all(axis=1) & np.isfinite(net_t).all(axis=1)
    res = {"n": n, "seed": seed, "reference_finite": int(finite.sum()), "rest_completed": int(rest.sum()),
           "dry": int((ratio == 0).sum()), "full": int((ratio == 1).sum())}
    say(f"{n} bodies (seed {seed}): reference finite on {finite.sum()}, rest-completed {rest.sum()}, dry {(ratio == 0).sum()}, full {(ratio == 1).sum()}")
    # ---- oracle vs reference: the nine outputs, then the behaviour-level net wrench ----
    scale = np.maximum(1.0, np.abs(comps[:, :6]).max(axis=(1, 2)))
    d_or = np.max([np.abs(o[f] - comps[:, k]).max(axis=1) / (scale if k < 6 else np.maximum(1.0, np.abs(comps[:, k]).max(axis=1)))
                   for k, f in enumerate(ho.COMPONENT_FIELDS)], axis=0)
    d_or = np.maximum(d_or, np.abs(o["ratio"] - ratio))
    bad = np.where(finite & (d_or > 1e-9))[0]
    res["oracle_components_over_1e-9"] = len(bad)
    res["oracle_components_max"] = float(np.nanmax(np.where(finite, d_or, 0)))
    say(f"oracle vs reference (components, ratio): {len(bad)} bodies above 1e-9, max {res['oracle_components_max']:.3e}")
    for i in bad[:5]:
        say("   ", i, state[i], params[i], d_or[i])
    err_o = _wrench_error(ho, rf, rt, net_f, net_t, params, rho, g)
    res["oracle_wrench_max"] = float(np.nanmax(np.where(finite, err_o, 0)))
    say(f"oracle vs reference (net wrench of _apply_behavior): max {res['oracle_wrench_max']:.3e}")
    # ---- kernel arithmetic (host instantiation of csrc/hydro_body.h) vs reference ----
    lib = ctypes.CDLL(os.path.join(REPO, "tests", "host_emul", "libemul.so"))
    fp = ctypes.POINTER(ctypes.c_float)
    f = np.empty((n, 3), np.float32); t = np.empty((n, 3), np.float32); r = np.empty(n, np.float32)
    lib.emul_wrench(ctypes.c_int64(n), state.ctypes.data_as(fp), prev.ctypes.data_as(fp), params.ctypes.data_as(fp), ctypes.c_double(rho),
                    ctypes.c_double(g), ctypes.c_double(dt), f.ctypes.data_as(fp), t.ctypes.data_as(fp), r.ctypes.data_as(fp))
    err = _wrench_error(ho, f, t, net_f, net_t, params, rho, g)
    over = finite & ~(err <= GATE)
    tiny = params[:, :3].min(axis=1) < TINY_DIMENSION
    badk = np.where(over)[0]
    res["kernel_wrench_over_gate"] = len(badk)
    res["kernel_wrench_over_gate_ordinary_bodies"] = int((over & ~tiny).sum())        # smallest dimension >= 1e-6 m
    res["kernel_wrench_max_ordinary_bodies"] = float(np.nanmax(np.where(finite & ~tiny, err, 0)))
    res["kernel_wrench_max"] = float(np.nanmax(np.where(finite, err, 0)))
    res["tiny_bodies"] = int((finite & tiny).sum())
    say(f"kernel arithmetic vs reference (net wrench): {len(badk)} bodies above the 1e-5 gate, max {res['kernel_wrench_max']:.3e}")
    for i in badk[:8]:
        say("   ", i, "state", state[i], "params", params[i], "got", f[i], t[i], "ref", net_f[i], net_t[i], err[i])
    # Is it the kernels or the reference?  The reference forms WORLD-space centres and subtracts the position again
    # (hydrodynamics_behavior.py:212-214): at |p_xy| ~ 50 m that costs 1e-14 m of a lever arm - nothing, unless the body is
    # 1e-7 m thin and its torque a 300-fold cancellation.  Re-run the reference with p_x = p_y = 0 (the wrench does not
    # depend on them) on EVERY tiny body: those must agree with the kernels inside the gate.
    idx = np.where(finite & tiny)[0]
    res["tiny_over_gate_at_pxy0"], res["tiny_max_at_pxy0"] = 0, 0.0
    if len(idx):
        st0 = state[idx].copy(); st0[:, 0:2] = 0.0
        with np.errstate(all="ignore"):
            nf0, nt0 = mg.reference_behavior_batch(st0, prev[idx], params[idx], rho, g, dt)
        e0 = _wrench_error(ho, f[idx], t[idx], nf0, nt0, params[idx], rho, g)
        ok0 = np.isfinite(nf0).all(axis=1) & np.isfinite(nt0).all(axis=1)
        res["tiny_over_gate_at_pxy0"] = int((ok0 & ~(e0 <= GATE)).sum())
        res["tiny_max_at_pxy0"] = float(np.nanmax(np.where(ok0, e0, 0)))
        say(f"    the {len(idx)} bodies with a dimension below {TINY_DIMENSION:g} m against the reference re-run at p_x = p_y = 0: "
            f"{res['tiny_over_gate_at_pxy0']} above the gate, max {res['tiny_max_at_pxy0']:.3e}")
    # ---- components: centres, the acceleration-independent forces, the ratio ----
    acc32 = accel.astype(np.float32)
    out = np.empty((n, 8, 3), np.float32); rr = np.empty(n, np.float32)
    lib.emul_components(ctypes.c_int64(n), state.ctypes.data_as(fp), acc32.ctypes.data_as(fp), params.ctypes.data_as(fp), ctypes.c_double(rho),
                        ctypes.c_double(g), out.ctypes.data_as(fp), rr.ctypes.data_as(fp))
    nb = 0
    for k in (6, 7):
        want = comps[:, k]
        tol = 0.5 * np.spacing(np.abs(want).astype(np.float32)).astype(np.float64) * (1 + 1e-6) + 1e-6
        b = finite & (np.abs(out[:, k] - want) > tol).any(axis=1)
        nb += int(b.sum())
        for i in np.where(b)[0][:4]:
            say("    centre", k, i, state[i], params[i], out[i, k], want[i])
    res["centres_outside_half_ulp"] = nb
    say(f"kernel arithmetic vs reference (cob / cop): {nb} bodies outside half an fp32 ulp + 1e-6")
    nf = 0
    for k in (0, 1, 2, 3):                                            # buoyancy, drag force, lift, drag torque: independent of the accelerations
        b = finite & (np.abs(out[:, k] - comps[:, k]).max(axis=1) > 1e-6 * scale)
        nf += int(b.sum())
        for i in np.where(b)[0][:4]:
            say("    component", k, i, state[i], params[i], out[i, k], comps[i, k])
    res["components_beyond_1e-6"] = nf
    say(f"kernel arithmetic vs reference (buoyancy, drag force, lift, drag torque): {nf} bodies beyond 1e-6 of the body's largest term")
    res["ratio_max_diff"] = float(np.nanmax(np.where(finite, np.abs(rr - ratio), 0)))
    say(f"ratio: max |diff| {res['ratio_max_diff']:.3e}")
    return res


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 20000, int(sys.argv[2]) if len(sys.argv) > 2 else 1, verbose=True)
