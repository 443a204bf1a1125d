"""CPU-only robustness sweep of the host instantiation of hydro_body.h far outside the bench scenes:
dims 1e-3..30 m, speeds 1e-5..50 m/s, spins 1e-5..50 rad/s, depths to 1e4 m, accelerations to 1e4 m/s^2.
python tests/tools/extreme_ranges.py [n] [seed]"""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO)
from oracle import c_oracle, hydro_oracle as ho
from silver2_isaacsim_amd import scenes
def population(n, seed):
    """(state, prev, params, dt) of the stress population, branch-margin gated"""
    rng = np.random.default_rng(seed)
    lu = lambda lo, hi, size: np.exp(rng.uniform(np.log(lo), np.log(hi), size))
    dims = lu(1e-3, 30.0, (n, 3))
    q = scenes.random_unit_quats(rng, n)
    dims32, q32 = dims.astype(np.float32), q.astype(np.float32)
    ext = scenes.vertical_extent(q32, dims32)
    kind = rng.uniform(0, 1, n)
    pz = np.where(kind < 0.2, ext * rng.uniform(1.01, 3.0, n), np.where(kind < 0.6, ext * rng.uniform(-0.999, 0.999, n), -ext - lu(1e-3, 1e4, n)))
    def vec(mag):
        d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True); return d * mag[:, None]
    v = vec(lu(1e-5, 50.0, n)); w = vec(lu(1e-5, 50.0, n))
    a = vec(lu(1e-3, 1e4, n)); al = vec(lu(1e-3, 1e4, n))
    dt = 1.0 / 60.0
    coeffs = np.stack([lu(0.1, 3, n), lu(0.01, 3, n), lu(0.1, 1e3, n), lu(0.1, 1e3, n), lu(0.01, 2, n), lu(1e-3, 1, n), lu(1e-3, 1, n)], 1)
    mass = lu(0.05, 5.0, n) * 1025.0 * dims.prod(1)
    state = np.concatenate([rng.uniform(-1e3, 1e3, (n, 2)), pz[:, None], q32, v, w], 1).astype(np.float32)
    prev = np.concatenate([v - a * dt, w - al * dt], 1).astype(np.float32)
    params = np.concatenate([dims32, coeffs, mass[:, None]], 1).astype(np.float32)
    keep = scenes.branch_margins(state, params) > 1e-4
    return state[keep], prev[keep], params[keep], dt


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    state, prev, params, dt = population(n, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    m = len(state)
    lib = ctypes.CDLL(os.path.join(REPO, "tests", "host_emul", "libemul.so")); fp = ctypes.POINTER(ctypes.c_float)
    f = np.empty((m, 3), np.float32); t = np.empty((m, 3), np.float32); r = np.empty(m, np.float32)
    lib.emul_wrench(ctypes.c_int64(m), state.ctypes.data_as(fp), prev.ctypes.data_as(fp), params.ctypes.data_as(fp), ctypes.c_double(1025.0),
                    ctypes.c_double(9.81), ctypes.c_double(dt), f.ctypes.data_as(fp), t.ctypes.data_as(fp), r.ctypes.data_as(fp))
    rf, rt = c_oracle.wrench(state, prev, params, 1025.0, 9.81, dt, threads=8)
    err = ho.wrench_error(f, t, rf, rt, params, 1025.0, 9.81)
    print("bodies", m, "finite", bool(np.isfinite(f).all() and np.isfinite(t).all()), "max", err.max(), "over 1e-5:", int((err > 1e-5).sum()),
          "p99.99", np.percentile(err, 99.99), "median", np.median(err))
    for i in np.argsort(-err)[:6]:
        print(f"  err {err[i]:.2e} dims {params[i, :3]} |v| {np.linalg.norm(state[i, 7:10]):.3g} |w| {np.linalg.norm(state[i, 10:13]):.3g} pz {state[i, 2]:.4g} "
              f"|F| {np.linalg.norm(rf[i]):.3g} dF {np.linalg.norm(f[i] - rf[i]):.3g} |T| {np.linalg.norm(rt[i]):.3g} dT {np.linalg.norm(t[i] - rt[i]):.3g}")
    # component breakdown of the worst body
    i = int(np.argmax(err))
    out = np.zeros(25, np.float32)
    lib.emul_body(state[i].ctypes.data_as(fp), prev[i].ctypes.data_as(fp), params[i].ctypes.data_as(fp), ctypes.c_double(1025.0), ctypes.c_double(9.81),
                  ctypes.c_double(dt), out.ctypes.data_as(fp))
    acc = ho.finite_difference_accel(state[i:i + 1].astype(np.float64), prev[i:i + 1].astype(np.float64), dt)
    comps, ratio = c_oracle.components(state[i:i + 1], acc, params[i:i + 1, :10], 1025.0, 9.81)
    c = comps[0]; v = out[:24].reshape(8, 3)
    print("worst body: q", state[i, 3:7], "v", state[i, 7:10], "w", state[i, 10:13], "ratio", ratio[0])
    for k, name in enumerate(("buoyF", "dragF", "liftF", "dragT", "amF", "amT", "cob", "cop")):
        a_, b_ = v[k].astype(np.float64), c[k]
        print(f"  {name:6s} emul {a_} ref {b_} rel {np.linalg.norm(a_ - b_) / max(np.linalg.norm(b_), 1e-300):.2e}")
    print("  netT ref", rt[i] , "dT", t[i] - rt[i])
