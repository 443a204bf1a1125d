#!/usr/bin/env python3
"""CPU-only numerics study of the error tail: csrc/hydro_body.h compiled for the host (tests/host_emul)
against the fp64 C oracle over many seeded scenes; prints the worst bodies with their force budget.
python tests/tools/diag_tail.py [seeds] [n] [threshold]"""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO)
from oracle import c_oracle, hydro_oracle as ho
from silver2_isaacsim_amd import scenes

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 5e-6
lib = ctypes.CDLL(os.path.join(REPO, "tests", "host_emul", "libemul.so"))
fp = ctypes.POINTER(ctypes.c_float)


def emul(sc):
    st = np.ascontiguousarray(sc.state, np.float32); pv = np.ascontiguousarray(sc.prev, np.float32)
    pr = np.ascontiguousarray(sc.params, np.float32)
    f = np.empty((sc.n, 3), np.float32); t = np.empty((sc.n, 3), np.float32); r = np.empty(sc.n, np.float32)
    lib.emul_wrench(ctypes.c_int64(sc.n), st.ctypes.data_as(fp), pv.ctypes.data_as(fp), pr.ctypes.data_as(fp),
                    ctypes.c_double(sc.rho), ctypes.c_double(sc.g), ctypes.c_double(sc.dt),
                    f.ctypes.data_as(fp), t.ctypes.data_as(fp), r.ctypes.data_as(fp))
    return f, t, r


edges = [0, 1e-7, 3e-7, 1e-6, 3e-6, 5e-6, 1e-5, 3e-5, 1e9]
hist = np.zeros(len(edges) - 1, int)
for seed in range(100, 100 + seeds):
    for law in ("c4", "c5"):
        sc = (scenes.scene_c4 if law == "c4" else scenes.scene_c5)(n=n, seed=seed, margin=None)
        f, t, r = emul(sc)
        rf, rt = c_oracle.wrench(sc.state, sc.prev, sc.params, sc.rho, sc.g, sc.dt, threads=8)
        err = ho.wrench_error(f, t, rf, rt, sc.params, sc.rho, sc.g)
        hist += np.histogram(err, bins=edges)[0]
        bad = np.nonzero(err > thr)[0]
        for i in bad:
            acc = ho.finite_difference_accel(sc.state[i:i + 1].astype(np.float64), sc.prev[i:i + 1].astype(np.float64), sc.dt)
            comps, ratio = c_oracle.components(sc.state[i:i + 1], acc, sc.params[i:i + 1, :10], sc.rho, sc.g)
            c = comps[0]
            ef = np.linalg.norm(f[i] - rf[i]) / max(np.linalg.norm(rf[i]), 1e-30)
            et = np.linalg.norm(t[i] - rt[i]) / max(np.linalg.norm(rt[i]), 1e-30)
            print(f"seed {seed} {law} body {i}: err {err[i]:.2e} (F {ef:.2e}, T {et:.2e}) ratio {ratio[0]:.4g}")
            print(f"   netF {rf[i]}  dF {f[i] - rf[i]}")
            print(f"   buoy {c[0]} drag {c[1]} lift {c[2]} amF {c[4]}")
            print(f"   netT {rt[i]}  dT {t[i] - rt[i]}")
            print(f"   dragT {c[3]} amT {c[5]} cob-p {c[6] - sc.state[i, :3]} cop-p {c[7] - sc.state[i, :3]}")
    print("seed", seed, "hist", dict(zip([f"{e:g}" for e in edges[1:]], hist.tolist())), flush=True)
