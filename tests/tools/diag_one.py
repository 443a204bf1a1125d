"""CPU-only: one body of a seeded scene, every component of the host instantiation of hydro_body.h against the
fp64 C oracle.   python tests/tools/diag_one.py SEED c4|c5 BODY"""
import ctypes, os, sys, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO)
from oracle import c_oracle, hydro_oracle as ho
from silver2_isaacsim_amd import scenes
seed, law, i = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
sc = (scenes.scene_c4 if law == "c4" else scenes.scene_c5)(n=262144, seed=seed, margin=1e-4 if os.environ.get("HYDRO_GATED") else None)
lib = ctypes.CDLL(os.path.join(REPO, "tests", "host_emul", "libemul.so")); fp = ctypes.POINTER(ctypes.c_float)
st = sc.state[i].copy(); pv = sc.prev[i].copy(); pr = sc.params[i].copy(); out = np.zeros(25, np.float32)
lib.emul_body(st.ctypes.data_as(fp), pv.ctypes.data_as(fp), pr.ctypes.data_as(fp), ctypes.c_double(sc.rho), ctypes.c_double(sc.g), ctypes.c_double(sc.dt), out.ctypes.data_as(fp))
acc = ho.finite_difference_accel(sc.state[i:i+1].astype(np.float64), sc.prev[i:i+1].astype(np.float64), sc.dt)
comps, ratio = c_oracle.components(sc.state[i:i+1], acc, sc.params[i:i+1,:10], sc.rho, sc.g)
c = comps[0]; v = out[:24].reshape(8, 3)
print("state", st); print("params", pr)
def cmp(name, a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    print(f"{name:10s} emul {a} ref {b} rel {np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-300):.2e}")
cmp("ratio", out[24], ratio[0])
for k, name in enumerate(("buoyF", "dragF", "liftF", "dragT", "amF", "amT", "cob", "cop")):
    cmp(name, v[k], c[k])
