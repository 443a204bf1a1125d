"""Degenerate inputs shared by the CPU (host-emulated arithmetic) and GPU edge-case tests.

Two parts: the hand-written cases of rounds 1-3 (identity quaternion mostly), and - round 4 - every exact tie the model
decides (numba_hydrodynamics.py:80 `pz < 0`, :86-87 `z_min >= 0` / `z_max <= 0`, :132-134 `alignment > 0` and face centre
below the surface, :210 `norm(axis) < 1e-6`) under FOUR orientations: the identity, quarter turns about x and about y
written with the fp32 sqrt(1/2), and a non-unit quaternion (used as given, N7).  check() compares the net wrench AND,
when given, the calculator surface (eight vectors + ratio, hydro_step_components*) with the oracle: a wrench-only check
cannot see a wrong centre whose lever arm happens to be parallel to the force.
"""
import numpy as np

STD = [1.2, 0.8, 300.0, 150.0, 1.0, 0.05, 0.02]


def _body(p=(0, 0, -1), q=(0, 0, 0, 1), v=(0.3, -0.1, 0.5), w=(0.2, 0.1, -0.4), dims=(1, 1, 1), co=STD, mass=500.0,
          pv=(0, 0, 0, 0, 0, 0)):
    return list(p) + list(q) + list(v) + list(w), list(pv), list(dims) + list(co) + [mass]


CASES = {
    "zero dims, submerged": _body(dims=(0, 0, 0)),
    "zero dims, above water": _body(p=(0, 0, 1), dims=(0, 0, 0)),
    "zero-height plate lying on the surface": _body(p=(0, 0, 0), dims=(1, 1, 0)),
    "zero mass (clamp scale 0)": _body(mass=0.0),
    "at rest (N1 completion)": _body(v=(0, 0, 0), w=(0, 0, 0)),
    "speed at the 1e-6 threshold": _body(v=(1e-6, 0, 0)),
    "speed 1e-9": _body(v=(1e-9, 0, 0)),
    "huge speed (clamp active)": _body(v=(3e3, -2e3, 1e3), w=(50, 10, -20)),
    "bottom face exactly on the surface": _body(p=(0, 0, 0.5)),
    "top face exactly on the surface": _body(p=(0, 0, -0.5)),
    "centre exactly on the surface": _body(p=(0, 0, 0.0)),
    "p_z = -0.0": _body(p=(0, 0, -0.0)),
    "velocity along +up (d = -1, lift axis degenerate)": _body(v=(0, 0, 1.0)),
    "velocity along -up (d = +1)": _body(v=(0, 0, -2.0), p=(0, 0, -0.2)),
    "velocity along body x (two zero alignments)": _body(v=(1.5, 0, 0), p=(0, 0, -0.2)),
    "upside down": _body(q=(1, 0, 0, 0), p=(0, 0, -0.3)),
    "all coefficients zero": _body(co=[0.0] * 7),
    "10 km deep and away": _body(p=(1e4, -1e4, -1e4)),
    "10 km up": _body(p=(0, 0, 1e4)),
}

# ---- round 4: every surface tie x four orientations ------------------------------------------------------------------
_S = float(np.float32(np.sqrt(0.5)))
# name -> (quaternion xyzw, box dimensions).  With the fp32 sqrt(1/2) a quarter turn has R entries 2 s^2 and 1 - 2 s^2
# (exact in fp64); the two box dimensions it mixes are equal so that the tie heights are fp32 numbers.
ORIENTATIONS = {
    "identity": ((0.0, 0.0, 0.0, 1.0), (0.5, 1.0, 2.0)),
    "quarter turn about x": ((_S, 0.0, 0.0, _S), (2.0, 1.0, 1.0)),
    "quarter turn about y": ((0.0, _S, 0.0, _S), (1.0, 0.5, 1.0)),
    "non-unit (1,0,0,1)": ((1.0, 0.0, 0.0, 1.0), (0.5, 1.0, 2.0)),      # row 2 of "R" = (0, 2, -1): a sheared, scaled box
}


def _rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _add_tie_cases():
    for oname, (q, dims) in ORIENTATIONS.items():
        half = 0.5 * np.array(dims)
        rot = _rot(q)
        e = half * rot[2]                                  # heights of the three half-extents
        extent = float(np.abs(e).sum())
        up = rot[:, 2]
        tag = f" [{oname}]"
        assert float(np.float32(extent)) == extent, "tie heights must be fp32 numbers"
        CASES["top keypoint exactly on the surface" + tag] = _body(p=(0.25, -0.5, -extent), q=q, dims=dims)
        CASES["top keypoint on the surface, at rest" + tag] = _body(p=(0.25, -0.5, -extent), q=q, dims=dims, v=(0, 0, 0), w=(0, 0, 0))
        CASES["top keypoint one ulp above the surface" + tag] = _body(p=(0, 0, float(np.nextafter(np.float32(-extent), np.float32(0)))), q=q, dims=dims)
        CASES["top keypoint one ulp below the surface" + tag] = _body(p=(0, 0, float(np.nextafter(np.float32(-extent), np.float32(-9)))), q=q, dims=dims)
        CASES["bottom keypoint exactly on the surface" + tag] = _body(p=(0, 0, extent), q=q, dims=dims)
        CASES["bottom keypoint one ulp below the surface" + tag] = _body(p=(0, 0, float(np.nextafter(np.float32(extent), np.float32(0)))), q=q, dims=dims)
        CASES["centre exactly on the surface" + tag] = _body(p=(1.0, 2.0, 0.0), q=q, dims=dims)
        for a, an in enumerate("xyz"):
            if e[a] != 0.0 and float(np.float32(e[a])) == e[a]:
                CASES[f"+{an} face centre exactly on the surface" + tag] = _body(p=(0, 0, -e[a]), q=q, dims=dims)
                CASES[f"-{an} face centre exactly on the surface" + tag] = _body(p=(0, 0, e[a]), q=q, dims=dims, v=(-0.3, 0.1, -0.5))
        for a, an in enumerate("xyz"):                     # alignment == 0 for the four faces along the flow
            va = [0.0, 0.0, 0.0]; va[a] = 1.5
            CASES[f"velocity along world {an} (zero alignments)" + tag] = _body(p=(0, 0, -0.25), q=q, dims=dims, v=va)
        vup = tuple(float(np.float32(0.5 * c)) for c in up)
        CASES["velocity along +up (|axis| = 0)" + tag] = _body(p=(0, 0, -0.25), q=q, dims=dims, v=vup)
        CASES["velocity along -up (|axis| = 0), submerged" + tag] = _body(p=(0, 0, -8.0), q=q, dims=dims, v=tuple(-c for c in vup))
        CASES["submerged, top keypoint well below" + tag] = _body(p=(0, 0, -extent - 0.125), q=q, dims=dims)


_add_tie_cases()
NAMES = list(CASES)
STATE = np.array([CASES[k][0] for k in NAMES], dtype=np.float32)
PREV = np.array([CASES[k][1] for k in NAMES], dtype=np.float32)
PARAMS = np.array([CASES[k][2] for k in NAMES], dtype=np.float32)
RHO, G, DT = 1025.0, 9.81, float(np.float32(1.0 / 60.0))
ACCEL = (STATE[:, 7:13].astype(np.float64) - PREV.astype(np.float64)) / DT          # what the fused entries see (A13)
ACCEL32 = ACCEL.astype(np.float32)                                                  # what component mode is handed


def reference_outputs():
    """tests/golden/edge_cases.npz: what the REFERENCE ITSELF returns for this table (make_golden.py save_edge_cases);
    refuses a fixture that was made for another version of the table."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "edge_cases.npz"))
    assert [str(x) for x in z["names"]] == NAMES and np.array_equal(z["state"], STATE) and np.array_equal(z["params"], PARAMS) \
        and np.array_equal(z["prev"], PREV), "edge_cases.npz is stale: python3 -B tests/golden/make_golden.py --only-ties"
    return {k: z[k] for k in z.files}


def check(f, t, ratio=None, comps=None, comp_ratio=None):
    """f, t: (n,3) net wrench of the fused entries.  comps: (n,8,3) + comp_ratio (n,) of component mode evaluated with
    ACCEL32 as the accelerations (reference order: buoyancy F, drag F, lift F, drag T, added-mass F, added-mass T, cob, cop).
    Checked against the oracle AND against the reference's own outputs for this table (reference_outputs)."""
    from oracle import hydro_oracle as ho
    with np.errstate(all="ignore"):
        rf, rt, aux = ho.step_wrench(STATE, PREV, PARAMS, RHO, G, DT)
    fx = reference_outputs()
    for i, name in enumerate(NAMES):                            # the oracle is what the reference does, case by case
        scale = max(1.0, np.abs(fx["net_force"][i]).max(), np.abs(fx["net_torque"][i]).max())
        assert np.abs(rf[i] - fx["net_force"][i]).max() <= 1e-12 * scale and np.abs(rt[i] - fx["net_torque"][i]).max() <= 1e-12 * scale, name
        assert aux["ratio"][i] == fx["ratio"][i], name
    assert np.isfinite(f).all() and np.isfinite(t).all()
    for i, name in enumerate(NAMES):
        tol_f = 1e-6 * max(1.0, np.abs(rf[i]).max())
        tol_t = 1e-6 * max(1.0, np.abs(rt[i]).max(), np.abs(rf[i]).max())
        assert np.abs(f[i] - rf[i]).max() <= tol_f, (name, f[i], rf[i])
        assert np.abs(t[i] - rt[i]).max() <= tol_t, (name, t[i], rt[i])
        if aux["ratio"][i] == 0.0:
            assert np.all(f[i] == 0.0) and np.all(t[i] == 0.0), name
        if ratio is not None:
            assert abs(ratio[i] - aux["ratio"][i]) < 1e-6, name
    if comps is None:
        return
    with np.errstate(all="ignore"):
        c = ho.solve_components(STATE, ACCEL32.astype(np.float64), PARAMS.astype(np.float64), RHO, G)
    assert np.isfinite(comps).all()
    for i, name in enumerate(NAMES):
        scale = max(1.0, max(np.abs(c[fld][i]).max() for fld in ho.COMPONENT_FIELDS[:6]))
        for k, fld in enumerate(ho.COMPONENT_FIELDS[:6]):
            assert np.abs(comps[i, k] - c[fld][i]).max() <= 1e-6 * scale, (name, fld, comps[i, k], c[fld][i])
        for k, fld in ((6, "center_of_buoyancy"), (7, "center_of_pressure")):
            ref = c[fld][i]
            # a centre is p + arm evaluated in fp64 and rounded to fp32 once: half an fp32 ulp of the coordinate
            tol = 0.5 * np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64) * (1 + 1e-6) + 1e-12
            assert np.all(np.abs(comps[i, k] - ref) <= tol), (name, fld, comps[i, k], ref)
        for k in (0, 1, 2, 3, 6, 7):                            # everything but the added mass is independent of the accelerations:
            want = fx["components"][i, k]                        # straight against the reference's own numbers
            tol = 1e-6 * scale if k < 6 else 0.5 * np.spacing(np.abs(want).astype(np.float32)).astype(np.float64) * (1 + 1e-6) + 1e-12
            assert np.all(np.abs(comps[i, k] - want) <= tol), (name, k, comps[i, k], want)
        if c["ratio"][i] == 0.0:
            assert np.all(comps[i] == 0.0), name                      # Numba: zeros for everything, centres included (N6)
        if comp_ratio is not None:
            assert abs(comp_ratio[i] - c["ratio"][i]) < 1e-6, name
