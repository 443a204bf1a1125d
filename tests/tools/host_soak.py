#!/usr/bin/env python3
"""CPU-only parity soak of the kernel arithmetic: csrc/hydro_body.h compiled for the host (tests/host_emul)
against the fp64 C oracle, margin-gated and ungated C4 / C5 populations, many seeds, several processes.
    python tests/tools/host_soak.py FIRST_SEED N_SEEDS [n=262144] [workers=4] [gated|ungated|both]
    HYDRO_SOAK_LAWS=c2,c3 picks the buoy / SILVER2-link laws (up to 65 536 bodies each) instead of c4,c5
Prints every (seed, law, body) above 3e-6 and the error histogram; writes gpurun_out/host_soak.json."""
import ctypes, json, os, sys, time
from concurrent.futures import ProcessPoolExecutor
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO)

EDGES = [0, 1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 1e9]
fp = ctypes.POINTER(ctypes.c_float)


def one(args):
    seed, law, gated, n = args
    from oracle import c_oracle, hydro_oracle as ho
    from silver2_isaacsim_amd import scenes
    lib = ctypes.CDLL(os.environ.get("HYDRO_EMUL", os.path.join(REPO, "tests", "host_emul", "libemul.so")))
    if law == "c2":
        sc = scenes.scene_c2(n=min(n, 65536), seed=seed, margin=1e-4 if gated else None)
    elif law == "c3":
        sc = scenes.scene_c3(envs=min(n, 65536) // 19, seed=seed, margin=1e-4 if gated else None)
    else:
        sc = (scenes.scene_c4 if law == "c4" else scenes.scene_c5)(n=n, seed=seed, margin=1e-4 if gated else None)
    st = np.ascontiguousarray(sc.state, np.float32); pv = np.ascontiguousarray(sc.prev, np.float32)
    pr = np.ascontiguousarray(sc.params, np.float32)
    f = np.empty((sc.n, 3), np.float32); t = np.empty((sc.n, 3), np.float32); r = np.empty(sc.n, np.float32)
    lib.emul_wrench(ctypes.c_int64(sc.n), st.ctypes.data_as(fp), pv.ctypes.data_as(fp), pr.ctypes.data_as(fp),
                    ctypes.c_double(sc.rho), ctypes.c_double(sc.g), ctypes.c_double(sc.dt),
                    f.ctypes.data_as(fp), t.ctypes.data_as(fp), r.ctypes.data_as(fp))
    rf, rt = c_oracle.wrench(sc.state, sc.prev, sc.params, sc.rho, sc.g, sc.dt, threads=2)
    err = ho.wrench_error(f, t, rf, rt, sc.params, sc.rho, sc.g)
    bad = np.nonzero(err > 3e-6)[0]
    refined = int(getattr(lib, "emul_refined_count")()) if hasattr(lib, "emul_refined_count") else -1
    return {"seed": seed, "law": law, "gated": gated, "n": sc.n, "max": float(err.max()), "hist": np.histogram(err, bins=EDGES)[0].tolist(),
            "bad": [(int(i), float(err[i])) for i in bad], "refined": refined}


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
    workers = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    which = sys.argv[5] if len(sys.argv) > 5 else "gated"
    modes = {"gated": [True], "ungated": [False], "both": [True, False]}[which]
    seeds = [int(x) for x in os.environ["HYDRO_SOAK_SEEDS"].split(",")] if os.environ.get("HYDRO_SOAK_SEEDS") else range(first, first + count)
    laws = os.environ.get("HYDRO_SOAK_LAWS", "c4,c5").split(",")
    jobs = [(s, law, g, n) for s in seeds for law in laws for g in modes]
    hist = {True: np.zeros(len(EDGES) - 1, int), False: np.zeros(len(EDGES) - 1, int)}
    worst, over, runs, t0, refined = {True: 0.0, False: 0.0}, {True: [], False: []}, [], time.time(), 0
    with ProcessPoolExecutor(workers) as ex:
        for k, r in enumerate(ex.map(one, jobs)):
            hist[r["gated"]] += np.array(r["hist"]); worst[r["gated"]] = max(worst[r["gated"]], r["max"])
            refined += max(r["refined"], 0)
            for i, e in r["bad"]:
                if e > 1e-5:
                    over[r["gated"]].append((r["seed"], r["law"], i, e))
                print(f"  seed {r['seed']} {r['law']} {'gated' if r['gated'] else 'ungated'} body {i}: {e:.3e}", flush=True)
            runs.append({k2: v for k2, v in r.items() if k2 != "bad"})
            if (k + 1) % 40 == 0:
                print(f"[{k + 1}/{len(jobs)}] {time.time() - t0:.0f}s gated max {worst[True]:.3e} ungated max {worst[False]:.3e}", flush=True)
    out = {"evaluations": int(sum(r["n"] for r in runs)), "bin_edges": EDGES[:-1] + ["inf"],
           "gated": {"hist": hist[True].tolist(), "max": worst[True], "over_1e-5": over[True]},
           "ungated": {"hist": hist[False].tolist(), "max": worst[False], "over_1e-5": over[False]},
           "refined_bodies": refined, "seeds": [first, first + count - 1], "n": n}
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", f"host_soak_{first}_{count}_{which}.json"), "w"), indent=1)
    print(json.dumps(out))
