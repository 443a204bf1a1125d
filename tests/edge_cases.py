"""Degenerate inputs shared by the CPU (host-emulated arithmetic) and GPU edge-case tests."""
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
NAMES = list(CASES)
STATE = np.array([CASES[k][0] for k in NAMES], dtype=np.float32)
PREV = np.array([CASES[k][1] for k in NAMES], dtype=np.float32)
PARAMS = np.array([CASES[k][2] for k in NAMES], dtype=np.float32)
RHO, G, DT = 1025.0, 9.81, float(np.float32(1.0 / 60.0))


def check(f, t, ratio=None):
    from oracle import hydro_oracle as ho
    with np.errstate(all="ignore"):
        rf, rt, aux = ho.step_wrench(STATE, PREV, PARAMS, RHO, G, DT)
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
