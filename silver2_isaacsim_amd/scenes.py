"""Synthetic N-body scenes C1-C5 (input laws of SURVEY.md section 8d).

Host-side NumPy only: builds fp32-exact state / previous-velocity / parameter
arrays for tests, `bench.py` and the golden-fixture generator.  There is no
hydrodynamics arithmetic here apart from the geometric predicates needed to
place bodies (dry / partial / submerged) and to apply the branch-margin rule.

Field orders (same as include/hydro.h):
  state  (N,13): px py pz | qx qy qz qw | vx vy vz | wx wy wz   (quat xyzw)
  prev   (N, 6): velocity [lin | ang] at the previous physics step
  params (N,11): dimx dimy dimz | cd_lin cd_ang | damp_lin damp_ang | lift |
                 am_lin am_ang | mass

Parameter values for the SILVER2 link classes follow the reference's
hydrodynamics_config.json:7-54 and the `physics:mass` attributes recovered from
silver2_isaac_sim.usd; README defaults follow
hydrodynamics_behavior.py:30-45.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from .config import PART_TABLE, PART_MASS, SCHEMA_DEFAULTS

RHO = 1025.0
G = 9.81

# README / schema defaults in PARAM order [cd_lin cd_ang damp_lin damp_ang lift am_lin am_ang]
_DEFAULT_COEFFS = np.array([
    SCHEMA_DEFAULTS["linearDragCoefficient"], SCHEMA_DEFAULTS["angularDragCoefficient"],
    SCHEMA_DEFAULTS["linearDamping"], SCHEMA_DEFAULTS["angularDamping"],
    SCHEMA_DEFAULTS["liftCoefficient"],
    SCHEMA_DEFAULTS["linearAddedMassCoefficient"], SCHEMA_DEFAULTS["angularAddedMassCoefficient"],
], dtype=np.float64)


def part_params(part: str) -> np.ndarray:
    """11-vector PARAM row for one SILVER2 link class ('body','coxa','femur','tibia')."""
    t = PART_TABLE[part]
    return np.array([t["xDimension"], t["yDimension"], t["zDimension"],
                     t["linearDragCoefficient"], t["angularDragCoefficient"],
                     t["linearDamping"], t["angularDamping"], t["liftCoefficient"],
                     t["linearAddedMassCoefficient"], t["angularAddedMassCoefficient"],
                     PART_MASS[part]], dtype=np.float64)


@dataclass
class Scene:
    name: str
    state: np.ndarray          # (N,13) float32
    prev: np.ndarray           # (N,6)  float32
    params: np.ndarray         # (N,11) float32
    rho: float = RHO
    g: float = G
    dt: float = float(np.float32(1.0 / 60.0))
    coeff_dtype: str = "f32"   # 'f16': the 7 coefficient columns are fp16-representable
    info: dict = field(default_factory=dict)

    @property
    def n(self) -> int:
        return int(self.state.shape[0])

    def shard(self, rank: int, world: int) -> "Scene":
        from .distributed import shard_range
        lo, hi = shard_range(self.n, rank, world)
        return Scene(self.name, self.state[lo:hi], self.prev[lo:hi], self.params[lo:hi],
                     self.rho, self.g, self.dt, self.coeff_dtype,
                     dict(self.info, shard=(rank, world, lo, hi)))


# --------------------------------------------------------------------------
# geometric helpers (float64, evaluated on the fp32-rounded inputs)
# --------------------------------------------------------------------------
def _row2(q):
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1.0 - 2 * (x * x + y * y)], axis=1)


def _rot(q):
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    r = np.empty((q.shape[0], 3, 3))
    r[:, 0, 0] = 1 - 2 * (y * y + z * z); r[:, 0, 1] = 2 * (x * y - w * z); r[:, 0, 2] = 2 * (x * z + w * y)
    r[:, 1, 0] = 2 * (x * y + w * z); r[:, 1, 1] = 1 - 2 * (x * x + z * z); r[:, 1, 2] = 2 * (y * z - w * x)
    r[:, 2, 0] = 2 * (x * z - w * y); r[:, 2, 1] = 2 * (y * z + w * x); r[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return r


def vertical_extent(q, dims):
    """Half-height of the rotated box along world z: sum |h_a R2a|."""
    return (np.abs(_row2(q.astype(np.float64))) * (0.5 * dims.astype(np.float64))).sum(axis=1)


def random_unit_quats(rng, n):
    q = rng.standard_normal((n, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


_IJK = np.array([(i, j, k) for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1)], dtype=np.float64)


def branch_margins(state, params):
    """Smallest relative distance of each body from any discontinuous branch of
    the model (SURVEY 8d 'branch-margin rule').  All quantities dimensionless."""
    s = state.astype(np.float64)
    pr = params.astype(np.float64)
    p, q, v, w = s[:, 0:3], s[:, 3:7], s[:, 7:10], s[:, 10:13]
    dims = pr[:, 0:3]
    half = 0.5 * dims
    L = dims.max(axis=1)
    r2 = _row2(q)
    e = r2 * half                                        # (N,3) z-contribution per axis
    zk = p[:, 2, None] + e @ _IJK.T                      # (N,27)
    m = np.abs(zk).min(axis=1) / L
    zf = np.concatenate([p[:, 2, None] + e, p[:, 2, None] - e], axis=1)
    m = np.minimum(m, np.abs(zf).min(axis=1) / L)
    speed = np.linalg.norm(v, axis=1)
    wsp = np.linalg.norm(w, axis=1)
    for sp in (speed, wsp):
        m = np.minimum(m, np.abs(sp - 0.2) / 0.2)
        m = np.minimum(m, np.abs(sp - 1e-6) / 1e-6)
    moving = speed > 1e-6
    vdir = v / np.where(moving, speed, 1.0)[:, None]
    rot = _rot(q)
    u = np.einsum("nba,nb->na", rot, vdir)               # R^T vhat : face alignments are -/+ u_a
    m = np.where(moving, np.minimum(m, np.abs(u).min(axis=1)), m)
    axis_n = np.linalg.norm(np.cross(vdir, rot[:, :, 2]), axis=1)
    m = np.where(moving, np.minimum(m, np.abs(axis_n - 1e-6)), m)
    # projected area threshold
    face = np.stack([dims[:, 1] * dims[:, 2], dims[:, 0] * dims[:, 2], dims[:, 0] * dims[:, 1]], axis=1)
    area = np.zeros(len(s))
    for a in range(3):
        for sign in (1.0, -1.0):
            al = -sign * u[:, a]
            cz = p[:, 2] + sign * e[:, a]
            area += np.where(moving & (al > 0) & (cz < 0), al * face[:, a], 0.0)
    m = np.minimum(m, np.abs(area - 1e-6) / 1e-6)
    return m


def _gated(rng, n, draw, margin, max_rounds=64):
    """Draw n bodies with `draw(rng, k)->(state,prev,params)`, resampling any body
    whose branch margin is below `margin`.  Returns arrays + resample count."""
    state, prev, params = draw(rng, n)
    resampled = 0
    if margin:
        # margins of the whole population once (in chunks: the (N,27) keypoint table of a million
        # bodies is 200 MB), afterwards only of the bodies that were redrawn
        bad = np.concatenate([lo + np.nonzero(branch_margins(state[lo:lo + 131072], params[lo:lo + 131072]) < margin)[0]
                              for lo in range(0, n, 131072)]) if n else np.empty(0, dtype=np.int64)
        for _ in range(max_rounds):
            if bad.size == 0:
                break
            resampled += int(bad.size)
            s2, p2, pa2 = draw(rng, bad.size, bad)
            state[bad], prev[bad], params[bad] = s2, p2, pa2
            bad = bad[branch_margins(s2, pa2) < margin]
        else:
            raise RuntimeError("branch-margin resampling did not converge")
    return state, prev, params, resampled


def _f32(*arrs):
    return tuple(np.ascontiguousarray(a, dtype=np.float32) for a in arrs)


# --------------------------------------------------------------------------
# C1: one buoy, closed-loop (CPU plumbing config)
# --------------------------------------------------------------------------
def scene_c1() -> Scene:
    """Unit cube, README defaults, m=500 kg, dt=1/60, z0=0.3, v0=(0,0,-1e-3)."""
    state = np.zeros((1, 13)); state[0, 2] = 0.3; state[0, 6] = 1.0; state[0, 9] = -1e-3
    params = np.concatenate([[1.0, 1.0, 1.0], _DEFAULT_COEFFS, [500.0]])[None, :]
    state, prev, params = _f32(state, np.zeros((1, 6)), params)
    return Scene("C1", state, prev, params, info={"steps": 10000})


# --------------------------------------------------------------------------
# C2: 4096 partially submerged buoys
# --------------------------------------------------------------------------
def scene_c2(n: int = 4096, seed: int = 2, margin: float | None = 1e-4) -> Scene:
    dt = float(np.float32(1.0 / 60.0))

    def draw(rng, k, idx=None):
        dims = np.stack([rng.uniform(0.5, 1.5, k), rng.uniform(0.5, 1.5, k), rng.uniform(1.0, 3.0, k)], axis=1)
        pxy = rng.uniform(-50.0, 50.0, (k, 2))
        pz = rng.uniform(-0.45, 0.45, k) * dims[:, 2]
        q = np.concatenate([rng.normal(0.0, 0.1, (k, 3)), np.ones((k, 1))], axis=1)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        v = rng.normal(0.0, 0.5, (k, 3)); w = rng.normal(0.0, 0.2, (k, 3))
        a = rng.normal(0.0, 1.0, (k, 3)); al = rng.normal(0.0, 1.0, (k, 3))
        coeffs = _DEFAULT_COEFFS[None, :] * rng.uniform(0.8, 1.2, (k, 7))
        mass = 0.5 * RHO * dims.prod(axis=1)
        state = np.concatenate([pxy, pz[:, None], q, v, w], axis=1)
        prev = np.concatenate([v - a * dt, w - al * dt], axis=1)
        params = np.concatenate([dims, coeffs, mass[:, None]], axis=1)
        return _f32(state, prev, params)

    rng = np.random.default_rng(seed)
    state, prev, params, res = _gated(rng, n, draw, margin)
    return Scene("C2", state, prev, params, dt=dt, info={"seed": seed, "resampled": res, "margin": margin})


# --------------------------------------------------------------------------
# C3: SILVER2 hexapod (body + 6 coxa + 6 femur + 6 tibia) x envs
# --------------------------------------------------------------------------
C3_LINKS = ["body"] + ["coxa"] * 6 + ["femur"] * 6 + ["tibia"] * 6


def scene_c3(envs: int = 1024, seed: int = 3, margin: float | None = 1e-4) -> Scene:
    dt = float(np.float32(1.0 / 120.0))
    n = envs * len(C3_LINKS)
    link_rows = np.stack([part_params(p) for p in C3_LINKS])        # (19,11)
    side = int(np.ceil(np.sqrt(envs)))
    env_id = np.repeat(np.arange(envs), len(C3_LINKS))
    origin = np.stack([(env_id % side) * 4.0, (env_id // side) * 4.0, np.full(n, -18.44)], axis=1)
    all_params = np.tile(link_rows, (envs, 1))

    def draw(rng, k, idx=None):
        sel = np.arange(n) if idx is None else idx
        off = rng.uniform(-1.0, 1.0, (k, 3))
        off *= (0.6 * rng.uniform(0.0, 1.0, k) ** (1 / 3) / np.maximum(np.linalg.norm(off, axis=1), 1e-9))[:, None]
        q = random_unit_quats(rng, k)
        v = rng.normal(0.0, 0.1, (k, 3)); w = rng.normal(0.0, 0.3, (k, 3))
        a = rng.normal(0.0, 0.5, (k, 3)); al = rng.normal(0.0, 2.0, (k, 3))
        state = np.concatenate([origin[sel] + off, q, v, w], axis=1)
        prev = np.concatenate([v - a * dt, w - al * dt], axis=1)
        return _f32(state, prev, all_params[sel])

    rng = np.random.default_rng(seed)
    state, prev, params, res = _gated(rng, n, draw, margin)
    return Scene("C3", state, prev, params, dt=dt,
                 info={"seed": seed, "envs": envs, "links": len(C3_LINKS), "resampled": res, "margin": margin})


# --------------------------------------------------------------------------
# C4 / C5: large mixed population (dry / partial / submerged)
# --------------------------------------------------------------------------
_CLASS_ROWS = np.stack([part_params(p)[3:10] for p in ("body", "coxa", "femur", "tibia")])


def _draw_mixed(dt, fp16_coeffs):
    def draw(rng, k, idx=None):
        dims = np.exp(rng.uniform(np.log(0.05), np.log(2.0), (k, 3)))
        q = random_unit_quats(rng, k)
        dims32, q32 = _f32(dims, q)
        ext = vertical_extent(q32, dims32)
        kind = rng.uniform(0.0, 1.0, k)
        pz = np.where(kind < 0.25, ext * rng.uniform(1.05, 3.0, k),                 # dry
             np.where(kind < 0.50, ext * rng.uniform(-0.95, 0.95, k),               # partial
                      -ext * rng.uniform(1.05, 20.0, k)))                           # submerged
        pxy = rng.uniform(-100.0, 100.0, (k, 2))

        def vel(sigma):
            x = rng.normal(0.0, sigma, (k, 3))
            sel = rng.uniform(0.0, 1.0, k)
            slow = sel < 0.05
            tgt = rng.uniform(1e-3, 0.2, k)
            x = np.where(slow[:, None], x * (tgt / np.maximum(np.linalg.norm(x, axis=1), 1e-12))[:, None], x)
            at_rest = (sel >= 0.05) & (sel < 0.06)                # exactly 0: the N1 rest branch
            return np.where(at_rest[:, None], 0.0, x)

        v = vel(1.0); w = vel(1.0)
        a = rng.normal(0.0, 2.0, (k, 3)); al = rng.normal(0.0, 2.0, (k, 3))
        cls = rng.integers(0, 4, k)
        coeffs = _CLASS_ROWS[cls] * np.exp(rng.uniform(np.log(0.5), np.log(2.0), (k, 7)))
        if fp16_coeffs:
            coeffs = coeffs.astype(np.float16).astype(np.float64)
        mass = 0.5 * RHO * dims.prod(axis=1)
        mass = np.where(rng.uniform(0.0, 1.0, k) < 0.02, mass * 1e-3, mass)          # clamp-active minority
        state = np.concatenate([pxy, pz[:, None], q32, v, w], axis=1)
        prev = np.concatenate([v - a * dt, w - al * dt], axis=1)
        params = np.concatenate([dims32, coeffs, mass[:, None]], axis=1)
        return _f32(state, prev, params)
    return draw


def scene_c4(n: int = 262144, seed: int = 4, margin: float | None = 1e-4) -> Scene:
    dt = float(np.float32(1.0 / 60.0))
    rng = np.random.default_rng(seed)
    state, prev, params, res = _gated(rng, n, _draw_mixed(dt, False), margin)
    return Scene("C4", state, prev, params, dt=dt, info={"seed": seed, "resampled": res, "margin": margin})


def scene_c5(n: int = 1048576, seed: int = 5, margin: float | None = 1e-4) -> Scene:
    dt = float(np.float32(1.0 / 60.0))
    rng = np.random.default_rng(seed)
    state, prev, params, res = _gated(rng, n, _draw_mixed(dt, True), margin)
    return Scene("C5", state, prev, params, dt=dt, coeff_dtype="f16",
                 info={"seed": seed, "resampled": res, "margin": margin})


SCENES = {"c1": scene_c1, "c2": scene_c2, "c3": scene_c3, "c4": scene_c4, "c5": scene_c5}


def make_scene(name: str, **kw) -> Scene:
    return SCENES[name.lower()](**kw)


def kinetic_energy_fp64(state: np.ndarray, params: np.ndarray, rotational: bool = True) -> tuple[float, float]:
    """[translational, rotational] kinetic energy of a scene as a float64 host sum: sum 1/2 m |v|^2 and
    sum 1/2 w_b^T diag(m/12 (d^2+h^2), m/12 (w^2+h^2), m/12 (w^2+d^2)) w_b with w_b = R^T w (the quaternion used as given,
    SURVEY.md 8e).  The quantity does not exist in the reference; this sum is what the device reduction and the
    all-reduce over the ranks are checked against (bench.py `rel_err_vs_host_fp64`, tests)."""
    st = np.asarray(state, dtype=np.float64)
    pr = np.asarray(params, dtype=np.float64)
    m = pr[:, 10]
    lin = math.fsum(0.5 * m * (st[:, 7:10] ** 2).sum(axis=1))
    if not rotational:
        return lin, 0.0
    wb = np.einsum("nba,nb->na", _rot(st[:, 3:7]), st[:, 10:13])
    d2 = pr[:, 0:3] ** 2
    inertia = (m / 12.0)[:, None] * np.stack([d2[:, 1] + d2[:, 2], d2[:, 0] + d2[:, 2], d2[:, 0] + d2[:, 1]], axis=1)
    return lin, math.fsum(0.5 * (inertia * wb * wb).sum(axis=1))


def to_soa(arr: np.ndarray) -> np.ndarray:
    """(N,F) -> contiguous (F,N) float32: one row per SoA field."""
    return np.ascontiguousarray(np.asarray(arr, dtype=np.float32).T)


TILE = 64


def to_tiled(arr: np.ndarray) -> np.ndarray:
    """(N,F) -> (ceil(N/64), F, 64) float32, the engine's tiled-SoA layout (last tile zero-padded):
    body i, field f at [i // 64, f, i % 64]."""
    a = np.asarray(arr, dtype=np.float32)
    n, f = a.shape
    tiles = (n + TILE - 1) // TILE
    pad = np.zeros((tiles * TILE, f), dtype=np.float32)
    pad[:n] = a
    return np.ascontiguousarray(pad.reshape(tiles, TILE, f).transpose(0, 2, 1))


def from_tiled(t: np.ndarray, n: int) -> np.ndarray:
    """Inverse of `to_tiled`: (tiles,F,64) -> (N,F)."""
    tiles, f, _ = t.shape
    return np.ascontiguousarray(t.transpose(0, 2, 1).reshape(tiles * TILE, f)[:n])
