"""Designed (not random) populations for the conditioning tests: the situations in which terms of the wrench
cancel each other, which random scenes only hit by accident (about one body in 10^5 at 100x)."""
import numpy as np

from oracle import hydro_oracle as ho
from silver2_isaacsim_amd import scenes

RHO, G, DT = 1025.0, 9.81, float(np.float32(1.0 / 60.0))


def _f32(x):
    return np.asarray(x, dtype=np.float32)


def terminal_rise(n=4096, seed=11, cancel=100.0):
    """Fully submerged bodies rising at (1 +- 1/cancel) times the speed at which drag along z equals
    buoyancy: F_z is a `cancel`-fold cancellation between the two largest terms of the wrench."""
    rng = np.random.default_rng(seed)
    dims = np.exp(rng.uniform(np.log(0.05), np.log(1.0), (n, 3)))
    q = scenes.random_unit_quats(rng, n)
    dims32, q32 = _f32(dims), _f32(q)
    ext = scenes.vertical_extent(q32, dims32)
    pz = -ext * rng.uniform(1.5, 10.0, n)
    coeffs = np.array([1.2, 0.8, 300.0, 150.0, 1.0, 0.0, 0.0]) * np.exp(rng.uniform(np.log(0.5), np.log(2.0), (n, 7)))
    mass = 0.5 * RHO * dims.prod(axis=1) * 100.0                     # clamp far away
    params = _f32(np.concatenate([dims32, coeffs, mass[:, None]], axis=1))
    direction = np.stack([rng.normal(0, 0.05, n), rng.normal(0, 0.05, n), np.ones(n)], axis=1)
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)

    def fz_over_b(speed):
        st = np.zeros((n, 13)); st[:, 2] = pz; st[:, 3:7] = q32; st[:, 7:10] = direction * speed[:, None]
        c = ho.solve_components(st, np.zeros((n, 6)), params.astype(np.float64), RHO, G)
        return -c["drag_force"][:, 2] / c["buoyancy_force"][:, 2]
    lo, hi = np.full(n, 1e-3), np.full(n, 1e3)
    for _ in range(60):                                                # drag_z / B is monotone in the speed
        mid = np.sqrt(lo * hi)
        below = fz_over_b(mid) < 1.0
        lo, hi = np.where(below, mid, lo), np.where(below, hi, mid)
    speed = np.sqrt(lo * hi) * (1.0 + rng.choice([-1.0, 1.0], n) / cancel)
    state = np.zeros((n, 13)); state[:, 0:2] = rng.uniform(-50, 50, (n, 2)); state[:, 2] = pz; state[:, 3:7] = q32
    state[:, 7:10] = direction * speed[:, None]
    state[:, 10:13] = rng.normal(0, 0.3, (n, 3))
    state = _f32(state)
    prev = state[:, 7:13].copy()                                       # no acceleration: added mass stays out of it
    return state, prev, params


def near_upright_floaters(n=4096, seed=12, tilt_deg=0.5):
    """Partially submerged boxes tilted by ~tilt_deg, at rest: the whole torque is the buoyancy lever arm, whose
    horizontal components are ~tan(tilt) of its length."""
    rng = np.random.default_rng(seed)
    dims = np.stack([rng.uniform(0.5, 1.5, n), rng.uniform(0.5, 1.5, n), rng.uniform(1.0, 3.0, n)], axis=1)
    axis = np.stack([rng.normal(size=n), rng.normal(size=n), np.zeros(n)], axis=1)
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    ang = np.deg2rad(tilt_deg) * rng.uniform(0.5, 1.5, n)
    q = np.concatenate([axis * np.sin(ang / 2)[:, None], np.cos(ang / 2)[:, None]], axis=1)
    dims32, q32 = _f32(dims), _f32(q)
    ext = scenes.vertical_extent(q32, dims32)
    pz = ext * rng.uniform(-0.6, 0.6, n)
    coeffs = np.tile(np.array([1.2, 0.8, 300.0, 150.0, 1.0, 0.05, 0.02]), (n, 1))
    mass = 0.5 * RHO * dims.prod(axis=1)
    params = _f32(np.concatenate([dims32, coeffs, mass[:, None]], axis=1))
    state = np.zeros((n, 13)); state[:, 0:2] = rng.uniform(-50, 50, (n, 2)); state[:, 2] = pz; state[:, 3:7] = q32
    state[:, 7:13] = rng.normal(0, 1e-4, (n, 6))                      # essentially at rest
    state = _f32(state)
    keep = scenes.branch_margins(state, params) > 1e-4
    return state[keep], state[keep, 7:13].copy(), params[keep]


def torque_balance(n=4096, seed=15, cancel=300.0):
    """Submerged and part-submerged bodies spinning at (1 +- 1/cancel) times the rate at which the angular drag
    torque equals the sum of the other torque terms (buoyancy arm, drag arm, lift arm): the net torque is a
    `cancel`-fold cancellation of terms that an fp32 evaluation delivers to 1-2e-7 each: the population that no
    fp32 formulation of the model can pass, and the reason the kernels evaluate it in fp64."""
    rng = np.random.default_rng(seed)
    dims = np.exp(rng.uniform(np.log(0.1), np.log(2.0), (n, 3)))
    q = scenes.random_unit_quats(rng, n)
    dims32, q32 = _f32(dims), _f32(q)
    ext = scenes.vertical_extent(q32, dims32)
    pz = np.where(rng.uniform(0, 1, n) < 0.5, ext * rng.uniform(-0.8, 0.8, n), -ext * rng.uniform(1.2, 5.0, n))
    coeffs = np.array([1.2, 0.8, 300.0, 150.0, 1.0, 0.0, 0.0]) * np.exp(rng.uniform(np.log(0.5), np.log(2.0), (n, 7)))
    mass = 0.5 * RHO * dims.prod(axis=1) * 100.0
    params = _f32(np.concatenate([dims32, coeffs, mass[:, None]], axis=1))
    v = _f32(rng.normal(0.0, 1.0, (n, 3)))
    state = np.zeros((n, 13)); state[:, 0:2] = rng.uniform(-50, 50, (n, 2)); state[:, 2] = pz; state[:, 3:7] = q32
    state[:, 7:10] = v
    state = _f32(state)

    def terms(w):
        st = state.astype(np.float64); st[:, 10:13] = w
        c = ho.solve_components(st, np.zeros((n, 6)), params.astype(np.float64), RHO, G)
        p = st[:, 0:3]
        rest = (np.cross(c["center_of_buoyancy"] - p, c["buoyancy_force"])
                + np.cross(c["center_of_pressure"] - p, c["drag_force"] + c["lift_force"]))
        return rest, c["drag_torque"]
    rest, _ = terms(np.zeros((n, 3)))                                   # independent of the spin
    rest_n = np.linalg.norm(rest, axis=1)
    direction = rest / np.maximum(rest_n, 1e-300)[:, None]             # drag torque = ang_k w with ang_k < 0
    lo, hi = np.full(n, 1e-6), np.full(n, 1e4)
    for _ in range(70):                                                 # |drag torque| is monotone in the spin
        mid = np.sqrt(lo * hi)
        below = np.linalg.norm(terms(direction * mid[:, None])[1], axis=1) < rest_n
        lo, hi = np.where(below, mid, lo), np.where(below, hi, mid)
    spin = np.sqrt(lo * hi) * (1.0 + rng.choice([-1.0, 1.0], n) / cancel)
    state[:, 10:13] = _f32(direction * spin[:, None])
    keep = (scenes.branch_margins(state, params) > 1e-4) & (rest_n > 0) & (spin < 200.0)
    return state[keep], state[keep, 7:13].copy(), params[keep]


# --------------------------------------------------------------------------
# exact surface ties (round 4): QUANTISED bodies, on which the comparisons of the model are decided exactly
# --------------------------------------------------------------------------
_S = float(np.float32(np.sqrt(0.5)))
# quaternions xyzw, used as given (N7): the identity, half turns, quarter turns with the fp32 sqrt(1/2) (R then holds
# 2 s^2 = 1 - 3e-8 and 1 - 2 s^2 = 3e-8, exactly), thirds of a turn about the diagonals (exact permutation matrices),
# and three NON-unit ones (sheared / scaled "rotations": what the reference does with them is part of the contract)
TIE_QUATS = np.array([
    (0, 0, 0, 1), (1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0),
    (_S, 0, 0, _S), (-_S, 0, 0, _S), (0, _S, 0, _S), (0, -_S, 0, _S), (0, 0, _S, _S), (0, 0, -_S, _S),
    (0.5, 0.5, 0.5, 0.5), (-0.5, 0.5, 0.5, 0.5), (0.5, -0.5, 0.5, 0.5), (0.5, 0.5, -0.5, 0.5),
    (_S, _S, 0, 0), (0, _S, _S, 0),
    (1, 0, 0, 1), (0, 0.75, 0, 0.75), (0.25, 0.25, 0.25, 0.75),
], dtype=np.float64)
TIE_DIMS = np.array([0.25, 0.5, 1.0, 2.0])
TIE_KINDS = ("grid", "top", "bottom", "centre", "face")


def surface_ties(n=4096, seed=21):
    """Bodies whose keypoints / face centres sit EXACTLY on the water surface: TIE_QUATS x dims in {1/4,1/2,1,2}^3 x
    p_z placed so that the top keypoint, the bottom keypoint, the centre or a face centre has z = 0 where that height is
    an fp32 number (the rest of the population: p_z on a 1/8 grid, which produces every kind of tie by itself), with
    velocities that are zero, axis-aligned or generic.  Returns (state, prev, params, kind) with fp32-exact values;
    `kind` indexes TIE_KINDS (how p_z was chosen - what actually ties is a property of the numbers, see tie_census)."""
    rng = np.random.default_rng(seed)
    qi = rng.integers(0, len(TIE_QUATS), n)
    q = TIE_QUATS[qi]
    dims = TIE_DIMS[rng.integers(0, 4, (n, 3))]
    x, y, z, w = q.T
    row2 = np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1.0 - 2 * (x * x + y * y)], axis=1)
    e = 0.5 * dims * row2                                              # half-extent heights e_a = h_a R[2][a]
    extent = np.abs(e).sum(axis=1)
    kind = rng.integers(0, len(TIE_KINDS), n)
    grid = rng.integers(-24, 25, n) / 8.0
    face = e[np.arange(n), rng.integers(0, 3, n)] * rng.choice([-1.0, 1.0], n)
    pz = np.select([kind == 1, kind == 2, kind == 3, kind == 4], [-extent, extent, np.zeros(n), -face], grid)
    state = np.zeros((n, 13))
    state[:, 0:2] = rng.integers(-400, 401, (n, 2)) / 8.0
    state[:, 2] = pz
    state[:, 3:7] = q
    vkind = rng.integers(0, 4, n)
    axis_v = np.eye(3)[rng.integers(0, 3, n)] * (rng.choice([-1.0, 1.0], n) * rng.choice([0.125, 0.5, 2.0], n))[:, None]
    diag_v = rng.choice([-0.5, 0.5], (n, 3))
    gen_v = rng.integers(-16, 17, (n, 3)) / 16.0
    state[:, 7:10] = np.select([(vkind == 0)[:, None], (vkind == 1)[:, None], (vkind == 2)[:, None]], [np.zeros((n, 3)), axis_v, diag_v], gen_v)
    state[:, 10:13] = np.where((rng.integers(0, 4, n) == 0)[:, None], 0.0, rng.integers(-8, 9, (n, 3)) / 16.0)
    state = _f32(state)
    prev = _f32(state[:, 7:13].astype(np.float64) - rng.integers(-4, 5, (n, 6)) / 64.0)
    prev[rng.integers(0, 3, n) == 0] = 0.0                             # first step after reset: v_last = 0
    coeffs = np.tile(np.array([1.2, 0.8, 300.0, 150.0, 1.0, 0.05, 0.02]), (n, 1))
    coeffs[rng.integers(0, 4, n) == 0] = (1.0, 0.5, 20.0, 2.0, 0.5, 0.25, 0.125)
    mass = np.where(rng.integers(0, 8, n) == 0, 1.0, 0.5 * RHO * dims.prod(axis=1))     # 1 kg: clamp active
    params = _f32(np.concatenate([dims, coeffs, mass[:, None]], axis=1))
    return state, prev, params, kind


def tie_census(state, params):
    """How many bodies have which exact tie (evaluated in fp64 as the oracle does: R @ k + p)."""
    q = state[:, 3:7].astype(np.float64)
    x, y, z, w = q.T
    row2 = np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1.0 - 2 * (x * x + y * y)], axis=1)
    e = 0.5 * params[:, 0:3].astype(np.float64) * row2
    pz = state[:, 2].astype(np.float64)
    sgn = np.array([(i, j, k) for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1)], dtype=np.float64)
    zk = (e[:, None, :] * sgn[None, :, :]).sum(axis=2) + pz[:, None]
    zmax, zmin = zk.max(axis=1), zk.min(axis=1)
    faces = np.concatenate([pz[:, None] + e, pz[:, None] - e], axis=1)
    return {"top keypoint on the surface": int((zmax == 0).sum()),
            "top on the surface, not all keypoints on it": int(((zmax == 0) & (zmin < 0)).sum()),
            "bottom keypoint on the surface": int((zmin == 0).sum()),
            "centre on the surface": int((pz == 0).sum()),
            "a face centre on the surface": int((faces == 0).any(axis=1).sum()),
            "some keypoint on the surface": int((zk == 0).any(axis=1).sum()),
            "at rest": int((np.abs(state[:, 7:10]).max(axis=1) == 0).sum())}
