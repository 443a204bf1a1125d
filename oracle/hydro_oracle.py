"""CPU oracle (NumPy, float64) for the per-body hydrodynamic wrench path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`silver2_isaacsim_amd/`) may import this module; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it, and
only as the checker.

This is a restatement, in this repo's own code, of the reference algorithm:

  * kernels      /root/reference/src/scripts/physics/numba_hydrodynamics.py:9-314
  * geometry     /root/reference/src/scripts/physics/numba_hydrodynamics_wrapper.py:55-112
  * epilogue     /root/reference/src/scripts/physics/hydrodynamics_behavior.py:194-238

Parity pinning: `tests/golden/make_golden.py` executes the reference's own
functions (identity-`njit` stub; Numba itself is not installed in the build
container) on seeded inputs and commits inputs+outputs as `.npz` fixtures;
`tests/test_oracle_golden.py` checks this module against them to <=1e-12.
The behaviour-level epilogue (A13-A16) is pinned the same way: the generator
executes the reference's own `_apply_behavior` (hydrodynamics_behavior.py:176-238,
float64 torch tensors, Kit-only imports replaced by empty shells) and stores the
net force / torque it hands to the simulator; the survey's K1-K5 known-answer
vectors are checked too.  Only the C1 trajectory's
integrator (PhysX in the reference) is this repo's own.

Documented completion: the reference's
`calculate_pressure_and_area` has no return value when `speed <= 1e-6`
(numba_hydrodynamics.py:118,143).  The evident intent - the "Defaults" at
:113-115 - is `(center_of_buoyancy, 0.0)`; this oracle, the golden generator
and the HIP kernels all apply that completion.

Warp twin (`semantics="warp"`): the two places where warp_hydrodynamics.py:233-335 differs from the
Numba path (added-mass rotation, centres of a dry body) are restated from its source text.  PARITY
UNPINNED for that mode: `warp` is not importable in the build container and the reference holds no
outputs of it, so nothing executes or pins the Warp path; the Numba mode is unaffected.

Field orders used everywhere in this repo
  state  (13): px py pz | qx qy qz qw | vx vy vz | wx wy wz     (quat xyzw)
  prev    (6): vx vy vz | wx wy wz      (velocity at the previous step)
  params (11): dimx dimy dimz | cd_lin cd_ang | damp_lin damp_ang | lift |
               am_lin am_ang | mass
"""
from __future__ import annotations

import numpy as np

STATE_FIELDS = ("px", "py", "pz", "qx", "qy", "qz", "qw",
                "vx", "vy", "vz", "wx", "wy", "wz")
PREV_FIELDS = ("pvx", "pvy", "pvz", "pwx", "pwy", "pwz")
PARAM_FIELDS = ("dimx", "dimy", "dimz", "cd_lin", "cd_ang", "damp_lin",
                "damp_ang", "lift", "am_lin", "am_ang", "mass")
COMPONENT_FIELDS = ("buoyancy_force", "drag_force", "lift_force", "drag_torque",
                    "added_mass_force", "added_mass_torque",
                    "center_of_buoyancy", "center_of_pressure")

LOW_SPEED_THRESHOLD = 0.2     # numba_hydrodynamics.py:154
SPEED_EPS = 1e-6              # :118,156,170,192,286
DRY_EPS = 1e-9                # :192,225,277
AREA_EPS = 1e-6               # :140
HEIGHT_EPS = 1e-6             # :92
AXIS_EPS = 1e-6               # :210
MAX_ACCEL = 500.0             # hydrodynamics_behavior.py:221
CLAMP_EPS = 1e-6              # hydrodynamics_behavior.py:224


# --------------------------------------------------------------------------
# geometry (numba_hydrodynamics_wrapper.py:55-112)
# --------------------------------------------------------------------------
def lattice_keypoints(dims):
    """27 body-frame keypoints {-hx,0,hx} x {-hy,0,hy} x {-hz,0,hz} (wrapper :55-73)."""
    hx, hy, hz = (0.5 * float(d) for d in dims)
    pts = [(i * hx, j * hy, k * hz)
           for k in (1, 0, -1) for j in (1, 0, -1) for i in (-1, 0, 1)]
    return np.asarray(pts, dtype=np.float64)


def box_faces(dims):
    """Face centres, outward normals and areas of the box (wrapper :75-99).

    Face order: +X, -X, +Y, -Y, +Z, -Z.
    """
    w, d, h = (float(x) for x in dims)
    normals = np.zeros((6, 3))
    centers = np.zeros((6, 3))
    half = (0.5 * w, 0.5 * d, 0.5 * h)
    for axis in range(3):
        for s, sign in enumerate((1.0, -1.0)):
            normals[2 * axis + s, axis] = sign
            centers[2 * axis + s, axis] = sign * half[axis]
    areas = np.array([d * h, d * h, w * h, w * h, w * d, w * d], dtype=np.float64)
    return centers, normals, areas


def added_mass_diagonal(dims, rho, am_lin, am_ang):
    """Diagonal of the 6x6 added-mass matrix (wrapper :101-112)."""
    w, d, h = (float(x) for x in dims)
    vol = w * d * h
    lin = vol * am_lin * rho
    return np.array([lin, lin, lin,
                     vol * (d * d + h * h) * am_ang * rho,
                     vol * (w * w + h * h) * am_ang * rho,
                     vol * (w * w + d * d) * am_ang * rho], dtype=np.float64)


# --------------------------------------------------------------------------
# scalar (one body) restatement - follows the reference step by step
# --------------------------------------------------------------------------
def rotation_from_quat_xyzw(q):
    """Unit quaternion [x,y,z,w] -> row-major 3x3, no normalisation
    (numba_hydrodynamics.py:9-51)."""
    x, y, z, w = (float(c) for c in q)
    x2, y2, z2 = x + x, y + y, z + z
    xx, xy, xz = x * x2, x * y2, x * z2
    yy, yz, zz = y * y2, y * z2, z * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    return np.array([[1.0 - (yy + zz), xy - wz, xz + wy],
                     [xy + wz, 1.0 - (xx + zz), yz - wx],
                     [xz - wy, yz + wx, 1.0 - (xx + yy)]], dtype=np.float64)


def submersion_and_cob(world_pts, position):
    """Submersion ratio from the z-extent, CoB = mean of points with z<0
    (numba_hydrodynamics.py:54-105)."""
    z = world_pts[:, 2]
    z_lo, z_hi = float(z.min()), float(z.max())
    if z_lo >= 0.0:
        return 0.0, position.copy()
    if z_hi <= 0.0:
        return 1.0, position.copy()
    height = z_hi - z_lo
    if height < HEIGHT_EPS:
        ratio = 1.0 if z_lo < 0.0 else 0.0
    else:
        ratio = min(1.0, -z_lo / height)
    wet = z < 0.0
    n_wet = int(wet.sum())
    cob = position.copy() if n_wet == 0 else world_pts[wet].sum(axis=0) / n_wet
    return ratio, cob


def pressure_centre_and_area(speed, vel_dir, cob, rot, position, dims):
    """Projected area of the wet faces whose normal opposes the velocity and the
    area-weighted centre of those faces (numba_hydrodynamics.py:108-143).
    `speed <= 1e-6` -> (cob, 0.0): the N1 completion, see module docstring."""
    if not speed > SPEED_EPS:
        return cob.copy(), 0.0
    centers, normals, areas = box_faces(dims)
    area = 0.0
    weighted = np.zeros(3)
    for f in range(6):
        n_w = rot @ normals[f]
        c_w = rot @ centers[f] + position
        alignment = -float(n_w @ vel_dir)
        if alignment > 0.0 and c_w[2] < 0.0:
            a = alignment * areas[f]
            area += a
            weighted += c_w * a
    cop = weighted / area if area > AREA_EPS else cob.copy()
    return cop, area


def hybrid_drag(speed, vel_dir, ratio, rho, area, volume,
                cd_lin, damp_lin, v, cd_ang, damp_ang, w):
    """Quadratic + linear drag force and torque (numba_hydrodynamics.py:146-182)."""
    quad_f = np.zeros(3)
    if speed > SPEED_EPS:
        quad_f = -(0.5 * rho * speed ** 2 * cd_lin * area) * vel_dir
    s_lin = speed / LOW_SPEED_THRESHOLD if speed < LOW_SPEED_THRESHOLD else 1.0
    force = (quad_f - damp_lin * v * s_lin) * ratio

    w_speed = float(np.linalg.norm(w))
    quad_t = np.zeros(3)
    if w_speed > SPEED_EPS:
        quad_t = -(0.5 * rho * w_speed ** 2 * cd_ang * volume) * (w / w_speed)
    s_ang = w_speed / LOW_SPEED_THRESHOLD if w_speed < LOW_SPEED_THRESHOLD else 1.0
    torque = (quad_t - damp_ang * w * s_ang) * ratio
    return force, torque


def lift_force(speed, vel_dir, rot, area, rho, lift_coeff, ratio):
    """Flat-plate angle-of-attack lift (numba_hydrodynamics.py:185-217)."""
    if speed < SPEED_EPS or ratio <= DRY_EPS:
        return np.zeros(3)
    up = rot[:, 2]
    d = -float(up @ vel_dir)
    d = max(-1.0, min(1.0, d))
    c_l = np.sin(2.0 * np.arcsin(d))
    mag = 0.5 * rho * speed ** 2 * c_l * area * lift_coeff
    axis = np.cross(vel_dir, up)
    n_axis = float(np.linalg.norm(axis))
    if n_axis < AXIS_EPS:
        return np.zeros(3)
    direction = np.cross(axis / n_axis, vel_dir)
    return mag * direction * ratio


def added_mass(ratio, a, alpha, rot, diag, semantics="numba"):
    """-M * body-frame acceleration, rotated back (numba_hydrodynamics.py:220-253).
    semantics="warp": the Warp twin rotates the world accelerations with quat_rotate(q, .) = R
    instead of R^T (warp_hydrodynamics.py:216-217); everything else is the same."""
    if ratio <= DRY_EPS:
        return np.zeros(3), np.zeros(3)
    to_local = rot if semantics == "warp" else rot.T
    a_b = to_local @ a
    al_b = to_local @ alpha
    f_b = -diag[:3] * a_b
    t_b = -diag[3:] * al_b
    return (rot @ f_b) * ratio, (rot @ t_b) * ratio


def solve_components_one(p, q, v, w, a, alpha, params, rho, g, semantics="numba"):
    """One body, 9 outputs in the reference's order
    (numba_hydrodynamics.py:256-314).  `params` = the 11 PARAM_FIELDS.
    semantics="warp": warp_hydrodynamics.py:233-335 where it differs (added mass, N3; a dry body
    reports cob = cop = centre of its wet keypoints or its position, N6)."""
    p, q, v, w, a, alpha = (np.asarray(x, dtype=np.float64) for x in (p, q, v, w, a, alpha))
    dims = np.asarray(params[:3], dtype=np.float64)
    cd_lin, cd_ang, damp_lin, damp_ang, lift_c, am_lin, am_ang = (float(x) for x in params[3:10])
    volume = float(dims[0] * dims[1] * dims[2])

    rot = rotation_from_quat_xyzw(q)
    world = lattice_keypoints(dims) @ rot.T + p
    ratio, cob = submersion_and_cob(world, p)
    if ratio <= DRY_EPS:
        z3 = np.zeros(3)
        if semantics == "warp":                # warp_hydrodynamics.py:58-61,283-290: outputs initialised, cop = cob
            return tuple(z3.copy() for _ in range(6)) + (cob.copy(), cob.copy(), 0.0)
        return tuple(z3.copy() for _ in range(8)) + (0.0,)

    buoy = np.array([0.0, 0.0, rho * (ratio * volume) * g])
    speed = float(np.linalg.norm(v))
    vel_dir = v / speed if speed > SPEED_EPS else np.zeros(3)
    cop, area = pressure_centre_and_area(speed, vel_dir, cob, rot, p, dims)
    drag_f, drag_t = hybrid_drag(speed, vel_dir, ratio, rho, area, volume,
                                 cd_lin, damp_lin, v, cd_ang, damp_ang, w)
    lift_f = lift_force(speed, vel_dir, rot, area, rho, lift_c, ratio)
    am_f, am_t = added_mass(ratio, a, alpha, rot,
                            added_mass_diagonal(dims, rho, am_lin, am_ang), semantics)
    return buoy, drag_f, lift_f, drag_t, am_f, am_t, cob, cop, ratio


def behavior_epilogue_one(p, comps, mass):
    """Lever-arm torques, sum and the 500 m/s^2 * mass clamp
    (hydrodynamics_behavior.py:212-226), float64."""
    buoy, drag_f, lift_f, drag_t, am_f, am_t, cob, cop = comps[:8]
    p = np.asarray(p, dtype=np.float64)
    t_b = np.cross(cob - p, buoy)
    t_d = np.cross(cop - p, drag_f)
    t_l = np.cross(cop - p, lift_f)
    net_f = buoy + drag_f + lift_f + am_f
    net_t = t_b + t_d + t_l + drag_t + am_t
    scale = min(1.0, float(mass) * MAX_ACCEL / (float(np.linalg.norm(net_f)) + CLAMP_EPS))
    return net_f * scale, net_t * scale, scale


# --------------------------------------------------------------------------
# vectorised batch versions (same arithmetic, N bodies at once)
# --------------------------------------------------------------------------
def _rot_batch(q):
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    x2, y2, z2 = x + x, y + y, z + z
    xx, xy, xz = x * x2, x * y2, x * z2
    yy, yz, zz = y * y2, y * z2, z * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    r = np.empty((q.shape[0], 3, 3))
    r[:, 0, 0] = 1.0 - (yy + zz); r[:, 0, 1] = xy - wz;         r[:, 0, 2] = xz + wy
    r[:, 1, 0] = xy + wz;         r[:, 1, 1] = 1.0 - (xx + zz); r[:, 1, 2] = yz - wx
    r[:, 2, 0] = xz - wy;         r[:, 2, 1] = yz + wx;         r[:, 2, 2] = 1.0 - (xx + yy)
    return r


_LATTICE_IJK = np.array([(i, j, k) for k in (1, 0, -1) for j in (1, 0, -1) for i in (-1, 0, 1)],
                        dtype=np.float64)  # (27,3)


def solve_components(state, accel, params, rho, g, semantics="numba"):
    """Vectorised A1-A11 (semantics as in `solve_components_one`).

    state  (N,13) float64, accel (N,6) [a | alpha], params (N,11).
    Returns dict with the eight (N,3) COMPONENT_FIELDS plus 'ratio' (N,) and
    diagnostic 'area' (N,), 'rest' (N,) bool = N1 completion fired.
    """
    state = np.asarray(state, dtype=np.float64)
    accel = np.asarray(accel, dtype=np.float64)
    params = np.asarray(params, dtype=np.float64)
    n = state.shape[0]
    p, q, v, w = state[:, 0:3], state[:, 3:7], state[:, 7:10], state[:, 10:13]
    a, alpha = accel[:, 0:3], accel[:, 3:6]
    dims = params[:, 0:3]
    cd_lin, cd_ang, damp_lin, damp_ang, lift_c, am_lin, am_ang = (params[:, i] for i in range(3, 10))
    half = 0.5 * dims
    volume = dims[:, 0] * dims[:, 1] * dims[:, 2]
    rot = _rot_batch(q)

    # A2/A3: world keypoints, extent ratio, CoB
    local = _LATTICE_IJK[None, :, :] * half[:, None, :]              # (N,27,3)
    world = np.einsum("nab,nkb->nka", rot, local) + p[:, None, :]     # (N,27,3)
    z = world[:, :, 2]
    z_lo, z_hi = z.min(axis=1), z.max(axis=1)
    height = z_hi - z_lo
    with np.errstate(divide="ignore", invalid="ignore"):
        frac = np.minimum(1.0, -z_lo / height)
    frac = np.where(height < HEIGHT_EPS, np.where(z_lo < 0.0, 1.0, 0.0), frac)
    ratio = np.where(z_lo >= 0.0, 0.0, np.where(z_hi <= 0.0, 1.0, frac))
    wet = z < 0.0
    n_wet = wet.sum(axis=1)
    partial = (z_lo < 0.0) & (z_hi > 0.0) & (n_wet > 0)
    mean_wet = (world * wet[:, :, None]).sum(axis=1) / np.maximum(n_wet, 1)[:, None]
    cob = np.where(partial[:, None], mean_wet, p)

    live = ratio > DRY_EPS

    # A5
    buoy = np.zeros((n, 3))
    buoy[:, 2] = rho * (ratio * volume) * g

    # A6
    speed = np.linalg.norm(v, axis=1)
    moving = speed > SPEED_EPS
    vel_dir = np.where(moving[:, None], v / np.where(moving, speed, 1.0)[:, None], 0.0)

    # A7 (N1 completion when not moving)
    area = np.zeros(n)
    weighted = np.zeros((n, 3))
    face_area = np.stack([dims[:, 1] * dims[:, 2], dims[:, 0] * dims[:, 2], dims[:, 0] * dims[:, 1]], axis=1)
    for axis in range(3):
        for sign in (1.0, -1.0):
            n_w = sign * rot[:, :, axis]
            c_w = sign * half[:, axis, None] * rot[:, :, axis] + p
            alignment = -np.einsum("na,na->n", n_w, vel_dir)
            take = moving & (alignment > 0.0) & (c_w[:, 2] < 0.0)
            a_f = np.where(take, alignment * face_area[:, axis], 0.0)
            area += a_f
            weighted += c_w * a_f[:, None]
    has_area = area > AREA_EPS
    cop = np.where(has_area[:, None], weighted / np.where(has_area, area, 1.0)[:, None], cob)

    # A8
    quad_f = -(0.5 * rho * speed ** 2 * cd_lin * area)[:, None] * vel_dir
    s_lin = np.where(speed < LOW_SPEED_THRESHOLD, speed / LOW_SPEED_THRESHOLD, 1.0)
    drag_f = (quad_f - (damp_lin * s_lin)[:, None] * v) * ratio[:, None]
    w_speed = np.linalg.norm(w, axis=1)
    spinning = w_speed > SPEED_EPS
    w_dir = np.where(spinning[:, None], w / np.where(spinning, w_speed, 1.0)[:, None], 0.0)
    quad_t = -(0.5 * rho * w_speed ** 2 * cd_ang * volume)[:, None] * w_dir
    s_ang = np.where(w_speed < LOW_SPEED_THRESHOLD, w_speed / LOW_SPEED_THRESHOLD, 1.0)
    drag_t = (quad_t - (damp_ang * s_ang)[:, None] * w) * ratio[:, None]

    # A9
    up = rot[:, :, 2]
    d = np.clip(-np.einsum("na,na->n", up, vel_dir), -1.0, 1.0)
    c_l = np.sin(2.0 * np.arcsin(d))
    mag = 0.5 * rho * speed ** 2 * c_l * area * lift_c
    axis_v = np.cross(vel_dir, up)
    n_axis = np.linalg.norm(axis_v, axis=1)
    lift_ok = (~(speed < SPEED_EPS)) & (~(n_axis < AXIS_EPS))
    direction = np.cross(axis_v / np.where(lift_ok, n_axis, 1.0)[:, None], vel_dir)
    lift_f = np.where(lift_ok[:, None], (mag * ratio)[:, None] * direction, 0.0)

    # A10
    lin = volume * am_lin * rho
    m_ang = np.stack([volume * (dims[:, 1] ** 2 + dims[:, 2] ** 2) * am_ang * rho,
                      volume * (dims[:, 0] ** 2 + dims[:, 2] ** 2) * am_ang * rho,
                      volume * (dims[:, 0] ** 2 + dims[:, 1] ** 2) * am_ang * rho], axis=1)
    to_local = "nab,nb->na" if semantics == "warp" else "nba,nb->na"     # N3: R (Warp) / R^T (Numba)
    a_b = np.einsum(to_local, rot, a)
    al_b = np.einsum(to_local, rot, alpha)
    am_f = np.einsum("nab,nb->na", rot, -lin[:, None] * a_b) * ratio[:, None]
    am_t = np.einsum("nab,nb->na", rot, -m_ang * al_b) * ratio[:, None]

    out = {
        "buoyancy_force": buoy, "drag_force": drag_f, "lift_force": lift_f,
        "drag_torque": drag_t, "added_mass_force": am_f, "added_mass_torque": am_t,
        "center_of_buoyancy": cob, "center_of_pressure": cop,
    }
    for k in out:                       # A4: dry bodies return zeros for everything
        if semantics == "warp" and k in ("center_of_buoyancy", "center_of_pressure"):
            # N6: cob as computed (position, or mean of the wet keypoints), cop = cob
            out[k] = np.where(live[:, None], out[k], cob)
            continue
        out[k] = np.where(live[:, None], out[k], 0.0)
    out["ratio"] = np.where(live, ratio, 0.0)
    out["area"] = np.where(live, area, 0.0)
    out["rest"] = live & ~moving
    return out


def finite_difference_accel(state, prev, dt):
    """A13: a = (v - v_last)/dt, alpha likewise (hydrodynamics_behavior.py:196-202)."""
    state = np.asarray(state, dtype=np.float64)
    prev = np.asarray(prev, dtype=np.float64)
    return (state[:, 7:13] - prev) / float(dt)


def behavior_epilogue(position, comps, mass):
    """Vectorised A14-A15 (hydrodynamics_behavior.py:212-226)."""
    p = np.asarray(position, dtype=np.float64)
    arm_b = comps["center_of_buoyancy"] - p
    arm_p = comps["center_of_pressure"] - p
    net_f = (comps["buoyancy_force"] + comps["drag_force"] + comps["lift_force"]
             + comps["added_mass_force"])
    net_t = (np.cross(arm_b, comps["buoyancy_force"]) + np.cross(arm_p, comps["drag_force"])
             + np.cross(arm_p, comps["lift_force"]) + comps["drag_torque"]
             + comps["added_mass_torque"])
    f_mag = np.linalg.norm(net_f, axis=1)
    scale = np.minimum(1.0, np.asarray(mass, dtype=np.float64) * MAX_ACCEL / (f_mag + CLAMP_EPS))
    return net_f * scale[:, None], net_t * scale[:, None], scale


def step_wrench(state, prev, params, rho, g, dt, semantics="numba"):
    """The fused entry point the HIP path implements: A13 + A1-A11 + A14 + A15.

    Returns (net_force (N,3), net_torque (N,3), aux dict)."""
    state = np.asarray(state, dtype=np.float64)
    params = np.asarray(params, dtype=np.float64)
    accel = finite_difference_accel(state, prev, dt)
    comps = solve_components(state, accel, params, rho, g, semantics)
    net_f, net_t, scale = behavior_epilogue(state[:, 0:3], comps, params[:, 10])
    comps["scale"] = scale
    return net_f, net_t, comps


def kinetic_energy(state, params, rotational=False):
    """sum 1/2 m |v|^2 (+ optional box-inertia rotational term).  New functionality
    named by BASELINE.json north_star; not present in the reference."""
    state = np.asarray(state, dtype=np.float64)
    params = np.asarray(params, dtype=np.float64)
    m = params[:, 10]
    ke = 0.5 * m * np.einsum("na,na->n", state[:, 7:10], state[:, 7:10])
    if rotational:
        rot = _rot_batch(state[:, 3:7])
        d = params[:, 0:3]
        inertia = (m / 12.0)[:, None] * np.stack([d[:, 1] ** 2 + d[:, 2] ** 2,
                                                  d[:, 0] ** 2 + d[:, 2] ** 2,
                                                  d[:, 0] ** 2 + d[:, 1] ** 2], axis=1)
        w_b = np.einsum("nba,nb->na", rot, state[:, 10:13])
        ke = ke + 0.5 * np.einsum("na,na->n", inertia * w_b, w_b)
    return float(ke.sum()), ke


def wrench_error(net_f, net_t, ref_f, ref_t, params, rho, g):
    """Per-body error metric of SURVEY 8d:
    max(|dF| / max(|F_ref|, 1e-3 rho g V), |dT| / max(|T_ref|, 1e-3 rho g V L))."""
    params = np.asarray(params, dtype=np.float64)
    dims = params[:, 0:3]
    vol = dims.prod(axis=1)
    floor_f = 1e-3 * rho * g * vol
    floor_t = floor_f * dims.max(axis=1)
    e_f = np.linalg.norm(np.asarray(net_f, np.float64) - ref_f, axis=1) / np.maximum(np.linalg.norm(ref_f, axis=1), floor_f)
    e_t = np.linalg.norm(np.asarray(net_t, np.float64) - ref_t, axis=1) / np.maximum(np.linalg.norm(ref_t, axis=1), floor_t)
    return np.maximum(e_f, e_t)
