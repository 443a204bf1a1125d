"""CPU-only: one body of a seeded scene, every component of the host instantiation of hydro_body.h against the
fp64 C oracle.   python tests/tools/diag_one.py SEED c4|c5 BODY"""
import ctypes, os, sys, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO)
from oracle import c_oracle, hydro_oracle as ho
from silver2_isaacsim_amd import scenes
seed, law, i = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
sc = (scenes.scene_c4 if law == "c4" else scenes.scene_c5)(n=262144, seed=seed, margin=1e-4 if os.environ.get("HYDRO_GATED") else None)
lib = ctypes.CDLL(os.path.join(REPO, "tests", "host_emul", "libemul.so")); fp = ctypes.POINTER(ctypes.c_float)
st = sc.state[i].copy(); pv = sc.prev[i].copy(); pr = sc.params[i].copy(); out = np.zeros(30, np.float32)
lib.emul_body(st.ctypes.data_as(fp), pv.ctypes.data_as(fp), pr.ctypes.data_as(fp), ctypes.c_double(sc.rho), ctypes.c_double(sc.g), ctypes.c_double(sc.dt), out.ctypes.data_as(fp))
acc = ho.finite_difference_accel(sc.state[i:i+1].astype(np.float64), sc.prev[i:i+1].astype(np.float64), sc.dt)
comps, ratio = c_oracle.components(sc.state[i:i+1], acc, sc.params[i:i+1,:10], sc.rho, sc.g)
c = comps[0]; p = sc.state[i,:3].astype(np.float64)
print("state", st); print("params", pr)
def cmp(name, a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    print(f"{name:10s} emul {a} ref {b} rel {np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-300):.2e}")
cmp("ratio", out[0], ratio[0]); cmp("buoy", out[1], c[0][2]); cmp("dragF", out[2:5], c[1]); cmp("liftF", out[5:8], c[2])
cmp("dragT", out[8:11], c[3]); cmp("amF", out[11:14], c[4]); cmp("amT", out[14:17], c[5])
cmp("armb", out[17:20], c[6]-p); cmp("armp", out[20:23], c[7]-p)
cmp("armb x B", [out[27], out[28], 0.0], np.cross(c[6]-p, c[0]))
cmp("fz_core", out[29], c[0][2] + c[1][2])
cmp("dragarmT", out[23:26], np.cross(c[7]-p, c[1]))
cmp("liftarmT", np.cross(out[20:23].astype(np.float64), out[5:8].astype(np.float64)), np.cross(c[7]-p, c[2]))
