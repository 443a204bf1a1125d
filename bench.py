#!/usr/bin/env python3
"""Benchmark of the fused hydrodynamic-wrench step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one launch of `hydro_step_wrench_tiled` over every body of one scene replica on this rank's GPU:
finite-difference acceleration + nine-component model + lever arms + sum + clamp.  Inputs are resident
in HBM before the timed region.  Bodies shard embarrassingly: every rank owns its own replica(s) (weak scaling), no
data-path collective; the one RCCL call is the global kinetic-energy all-reduce after the timed loop.

Workload (default `c5` = BASELINE.json configs[4]): 1 048 576 synthetic bodies per GPU, fp32 state, 7 coefficients
stored fp16, fp64 arithmetic rounded to fp32 once, 130 algorithmic bytes per body-step; `--scenes` (default 4)
replicas are stepped round-robin so that the bytes between two uses of a line exceed the 256 MiB Infinity Cache.

Prints ONE JSON line on rank 0, at most LINE_LIMIT (8 192) bytes: the contract fields, `roofline`, `cpu_baseline`,
`roofline_4m`, `configs` (C2 / C3 / C4-shard / C4, eager and graph), `box` (which bound binds on this box, clock held)
and, for N > 1, `per_rank` and `c4_strong` (BASELINE configs[3] as stated, kinetic energy sampled inside the region).
Everything else (scripts/bench_extras.py) goes to the side file `--extras-out` (default bench_extras.json beside this
file) and to stderr.  `python bench.py --explain` says what every field means; no prose travels on the line.
Exit code: 0; 3 when the N > 1 leg raised or hung (the line, with "ok": false, is out before the exit).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.modules.setdefault("bench", sys.modules[__name__])      # `import bench` in scripts/ is THIS module, also when run as __main__

from silver2_isaacsim_amd import distributed as hd          # noqa: E402
from silver2_isaacsim_amd import scenes                     # noqa: E402
from silver2_isaacsim_amd.engine import HydroEngine         # noqa: E402

EXIT_LEG_FAILED = 3              # exit code when the N > 1 leg raised or hung (the line, "ok": false, is out before it)
LINE_LIMIT = 8192                # bytes of the one JSON line (r05: a 22.5 KB line was not parsed by the driver)
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
BYTES_PER_BODY = {"f32": 144, "f16": 130}    # SURVEY.md 8d / BASELINE.md 3 (the algorithmic figure `frac` uses)
TRAFFIC_BYTES_PER_BODY = {"f32": 136, "f16": 122}    # what the kernel moves (profiles/traffic.json): p_x, p_y never loaded
INFINITY_CACHE_BYTES = 256 << 20
# resident bytes of one scene replica per body: state 52 + previous velocity 24 + parameters + wrench 24
REPLICA_BYTES_PER_BODY = {"f32": 52 + 24 + 44 + 24, "f16": 52 + 24 + 30 + 24}

WORKLOADS = {
    # name: (scene kind, bodies, coefficient dtype, description)
    "c5": ("c5", 1048576, "f16", "C5: 1 048 576 bodies/GPU, fp32 state, fp16 coefficients, fp64 arithmetic rounded to fp32 once"),
    "c5-f32": ("c4", 1048576, "f32", "C5 population with fp32 coefficients (144 B/body-step)"),
    "c4": ("c4", 262144, "f32", "C4: 262 144 bodies (per GPU under weak scaling)"),
    "c3": ("c3", 19456, "f32", "C3: SILVER2 hexapod x 1024 envs"),
    "c2": ("c2", 4096, "f32", "C2: 4 096 buoys"),
}
BASELINE_CONFIG = {"c2": "configs[1]", "c3": "configs[2]", "c4": "configs[3]", "c5": "configs[4]"}

_SCENES: dict = {}
DISTINCT_MAX = 1048576


def build_scene(kind: str, n: int, seed: int):
    """Scene of n bodies drawn from the law of SURVEY.md 8d.  Up to 1 048 576 bodies every body is DISTINCT (the
    margin-gated generator takes ~2 s per million); larger scenes are permuted copies of the 1 048 576-body one."""
    key = (kind, n, seed)
    if key in _SCENES:
        return _SCENES[key]
    if kind == "c2":
        sc = scenes.scene_c2(n=n, seed=seed)
    elif kind == "c3":
        sc = scenes.scene_c3(envs=n // len(scenes.C3_LINKS), seed=seed)
    else:
        base_n = min(n, DISTINCT_MAX)
        sc = scenes.scene_c5(n=base_n, seed=seed) if kind == "c5" else scenes.scene_c4(n=base_n, seed=seed)
        if base_n < n:
            reps = (n + base_n - 1) // base_n
            rng = np.random.default_rng(seed + 1000)
            perms = [rng.permutation(base_n) for _ in range(reps)]
            sc = scenes.Scene(sc.name, np.concatenate([sc.state[p] for p in perms])[:n], np.concatenate([sc.prev[p] for p in perms])[:n],
                              np.concatenate([sc.params[p] for p in perms])[:n], sc.rho, sc.g, sc.dt, sc.coeff_dtype,
                              dict(sc.info, copies_of_distinct_population=reps))
    if len(_SCENES) > 6:
        _SCENES.clear()
    _SCENES[key] = sc
    return sc


class Replica:
    """One scene replica resident on the GPU: state, previous velocity, engine (params), output.
    layout 'tiled' = the engine's native tiled-SoA buffers (hydro_step_wrench_tiled),
    'soa' = plain struct-of-arrays field pointers (hydro_step_wrench_ext)."""

    def __init__(self, sc, coeff: str, dev, roll: int, layout: str = "tiled"):
        idx = np.roll(np.arange(sc.n), roll)
        self.n = sc.n
        self.dt = sc.dt
        self.layout = layout
        self.engine = HydroEngine(sc.n, dev, sc.rho, sc.g)
        self.engine.set_params(sc.params[idx], coeff)
        if layout == "tiled":
            self.state = torch.from_numpy(scenes.to_tiled(sc.state[idx])).to(dev)
            self.prev = torch.from_numpy(scenes.to_tiled(sc.prev[idx])).to(dev)
            self.out = self.engine.alloc_tiled(6, sc.n)
        else:
            self.state = torch.from_numpy(scenes.to_soa(sc.state[idx])).to(dev)
            self.prev = torch.from_numpy(scenes.to_soa(sc.prev[idx])).to(dev)
            self.out = torch.empty((6, sc.n), dtype=torch.float32, device=dev)
        self.index = idx
        self._prepared = None
        self._prepared_ke = None

    def step(self):
        if self.layout == "tiled":
            if self._prepared is None:                 # arguments validated once; bound to the stream current now
                own = os.environ.get("HYDRO_BENCH_OWN_PREV") == "1"     # dev: the engine-owned previous velocity (updated in place)
                self._prepared = self.engine.prepare_step_wrench_tiled(self.state, self.n, self.dt, out=self.out, prev=None if own else self.prev)
            self._prepared()
        else:
            self.engine.step_wrench(self.state, self.dt, out=self.out, prev=self.prev)

    def step_sampling(self, ke_out):
        """The same step through the kernel variant that also leaves the kinetic energy of this replica's state in
        `ke_out` (hydro_step_wrench_tiled_ke: the bodies are in registers anyway, no second pass)."""
        if self._prepared_ke is None or self._prepared_ke[0] is not ke_out:
            self._prepared_ke = (ke_out, self.engine.prepare_step_wrench_tiled(self.state, self.n, self.dt, out=self.out, prev=self.prev,
                                                                                ke_out=ke_out, rotational=True))
        self._prepared_ke[1]()

    def kinetic_energy(self):
        return self.engine.kinetic_energy(self.state, rotational=True)

    def wrench_rows(self, m: int) -> np.ndarray:
        """(m,6) host copy of the first m bodies' wrench."""
        if self.layout == "tiled":
            tiles = (m + 63) // 64
            return scenes.from_tiled(self.out[:tiles].cpu().numpy(), m)
        return self.out[:, :m].cpu().numpy().T


def spin_up(replicas, stream, seconds: float):
    """Untimed: run the same step loop for `seconds` so that the GPU has left its idle power state before anything is
    measured (the first ~50 ms after idle run ~9 % slower; W warm-up steps of a ~25 us kernel are far shorter than that).
    This happens BEFORE the W warm-up steps and the K timed steps."""
    if seconds <= 0:
        return
    dev = replicas[0].state.device
    t0 = time.perf_counter()
    k = 0
    with torch.cuda.stream(stream):
        while time.perf_counter() - t0 < seconds:
            for _ in range(64):
                replicas[k % len(replicas)].step()
                k += 1
            stream.synchronize()
    torch.cuda.synchronize(dev)


def timed_steps(replicas, steps: int, warmup: int, stream, collectives: bool = False, after_step=None):
    """W warm-up steps, then exactly K steps between barrier+synchronize pairs.  Returns
    (wall seconds max over ranks, HIP-event milliseconds on the launch stream).
    The launch stream is made current BEFORE the region (entering torch's stream context costs the host ~10 us);
    without a process group the barrier is a no-op and the synchronize after it is skipped."""
    dev = replicas[0].state.device
    grouped = hd._collectives_on()
    with torch.cuda.stream(stream):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream); ev1.record(stream)              # (a torch event is created by its first record: not inside the region)
        for k in range(warmup):
            replicas[k % len(replicas)].step()
        torch.cuda.synchronize(dev)
        hd.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ev0.record(stream)
        for k in range(steps):
            replicas[k % len(replicas)].step()
            if after_step is not None:
                after_step(k + 1, replicas[k % len(replicas)])
        ev1.record(stream)
        torch.cuda.synchronize(dev)
        local = time.perf_counter() - t0                    # this rank's own steps are done (before it waits for the others)
        if grouped:
            hd.barrier()
            torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
    t = torch.tensor([wall], dtype=torch.float64, device=hd.collective_device(dev) if collectives else "cpu")
    hd.all_reduce_max_(t)
    timed_steps.last_local_wall = local                      # (main() gathers them for the N > 1 line)
    return float(t.item()), float(ev0.elapsed_time(ev1))


def usable_cpus() -> int:
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.999)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, (quota + period - 1) // period))
            break
        except Exception:
            continue
    return max(1, n)


def cpu_baseline_leg(sc, replica, budget_s: float):
    """The C oracle (oracle/hydro_oracle.c, a port of the reference's Numba path) timed on this box's host cores on a
    bounded sample of the same workload, and used as the checker of the GPU result for that sample.  Only this function
    touches oracle/."""
    from oracle import c_oracle, hydro_oracle
    m = min(sc.n, 262144)
    idx = replica.index[:m]
    st, pv, pr = sc.state[idx], sc.prev[idx], sc.params[idx]
    if replica.layout == "aos" or os.environ.get("HYDRO_BENCH_OWN_PREV") == "1":     # this entry keeps the previous velocity IN THE ENGINE and updates it every step: the state
        pv = np.ascontiguousarray(st[:, 7:13])      # does not change between the bench's steps, so from the second step on it equals the velocity
    c_oracle.wrench(st[:1024], pv[:1024], pr[:1024], sc.rho, sc.g, sc.dt)          # warm
    reps, t0 = 0, time.perf_counter()
    while True:
        ref_f, ref_t = c_oracle.wrench(st, pv, pr, sc.rho, sc.g, sc.dt, threads=1)
        reps += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    single = m * reps / (time.perf_counter() - t0)
    threads = min(c_oracle.max_threads(), usable_cpus())
    c_oracle.wrench(st, pv, pr, sc.rho, sc.g, sc.dt, threads=threads)               # spin the pool up
    reps_mt, t0 = 0, time.perf_counter()
    while True:
        c_oracle.wrench(st, pv, pr, sc.rho, sc.g, sc.dt, threads=threads)
        reps_mt += 1
        if time.perf_counter() - t0 >= budget_s / 3:
            break
    multi = m * reps_mt / (time.perf_counter() - t0)
    gpu = replica.wrench_rows(m)
    err = hydro_oracle.wrench_error(gpu[:, :3], gpu[:, 3:], ref_f, ref_t, pr, sc.rho, sc.g)
    try:
        cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        cpu_model = "unknown"
    return {
        "value": single, "unit": "body-steps/s", "cores": 1, "kind": "port",
        "sample": f"{m} bodies of the bench scene x {reps} passes, oracle/hydro_oracle.c (C port of the Numba path), 1 thread",
        "all_core_value": multi, "all_cores": threads, "hardware_threads": os.cpu_count(), "cpu_model": cpu_model,
        "gpu_vs_oracle_max_rel_err": float(err.max()), "gpu_vs_oracle_n_over_1e-5": int((err > 1e-5).sum()),
        "gpu_vs_oracle_checked": int(m),
    }


def residency(n: int, coeff: str, sets: int, bytes_per_body: int | None = None) -> dict:
    """Where the rotating working set of a measurement lives.  Below 256 MiB it stays in the Infinity Cache between
    two uses: the GB/s of such an entry is a CACHE rate, never an HBM fraction (SURVEY.md 8d cache caveat)."""
    ws = sets * n * (bytes_per_body or REPLICA_BYTES_PER_BODY[coeff])
    return {"working_set_bytes": ws, "resident": "infinity-cache" if ws < INFINITY_CACHE_BYTES else "hbm"}


def quick_rate(kind: str, n: int, coeff: str, dev, stream, steps: int = 60, sets: int = 4, seed: int = 11,
               layout: str = "tiled"):
    """Small untimed-contract measurement (not the headline): HIP events over `steps` launches, rotating replicas."""
    sc = build_scene(kind, n, seed)
    reps = [Replica(sc, coeff, dev, roll=r * 97, layout=layout) for r in range(sets)]
    spin_up(reps, stream, 0.15)
    _, ms = timed_steps(reps, steps, 10, stream)
    for r in reps:
        r.engine.close()
    us = ms * 1e3 / steps
    return {"n": sc.n, "coeff": coeff, "layout": layout, "us_per_step": us, "body_steps_per_s": sc.n / (us * 1e-6),
            "algorithmic_gbs": sc.n * BYTES_PER_BODY[coeff] / (us * 1e-6) / 1e9, **residency(sc.n, coeff, sets)}


def graph_rate(kind: str, n: int, coeff: str, dev, stream, steps_per_graph: int = 64, replays: int = 40, seed: int = 11):
    """Launch-bound small scenes: `steps_per_graph` consecutive steps (round-robin over 4 scene replicas) captured into
    one HIP graph and replayed - one host call per 64 physics steps.  The C-ABI step functions are capture-safe."""
    sc = build_scene(kind, n, seed)
    reps = [Replica(sc, coeff, dev, roll=r * 97) for r in range(4)]
    spin_up(reps, stream, 0.1)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        for k in range(8):
            reps[k % 4].step()
        stream.synchronize()
        with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
            for k in range(steps_per_graph):
                reps[k % 4].step()
        for _ in range(5):
            g.replay()
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(replays):
            g.replay()
        e1.record(stream)
        stream.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (replays * steps_per_graph)
    del g
    for r in reps:
        r.engine.close()
    return {"n": sc.n, "coeff": coeff, "layout": "tiled", "graph_steps": steps_per_graph,
            "us_per_step": us, "body_steps_per_s": sc.n / (us * 1e-6),
            "algorithmic_gbs": sc.n * BYTES_PER_BODY[coeff] / (us * 1e-6) / 1e9, **residency(sc.n, coeff, 4)}


def configs_block(dev, stream) -> dict:
    """SURVEY.md 8d's per-config absolutes, driver-visible: every BASELINE config that is not the headline (C2, C3, the
    32 768-body shard C4 leaves on each of 8 GPUs, C4 whole), eager (one ctypes launch per step) and graph (64 steps per
    replay).  Cache-resident sizes: microseconds and body-steps/s, no HBM fraction."""
    out = {}
    for key, kind, n, steps in (("c2", "c2", 4096, 200), ("c3", "c3", 19456, 200), ("c4_shard", "c4", 32768, 200), ("c4", "c4", 262144, 100)):
        try:
            e = quick_rate(kind, n, "f32", dev, stream, steps=steps)
            g = graph_rate(kind, n, "f32", dev, stream)
            out[key] = {"n": e["n"], "us_per_step": e["us_per_step"], "body_steps_per_s": e["body_steps_per_s"],
                        "graph_us_per_step": g["us_per_step"], "graph_body_steps_per_s": g["body_steps_per_s"]}
        except Exception as ex:                             # noqa: BLE001 - report, never lose the line
            out[key] = {"error": repr(ex)[:120]}
    return out


def roofline_4m(dev, stream, coeff: str = "f16", n: int = 4194304, sets: int = 2, steps: int = 200):
    """Second roofline object: the headline kernel on 4 194 304 bodies, two rotating replicas (1.1 GB): nothing of it
    survives in the 256 MiB Infinity Cache between two uses.  HIP events on the launch stream over the timed steps."""
    r = quick_rate("c5" if coeff == "f16" else "c4", n, coeff, dev, stream, steps=steps, sets=sets, seed=5)
    ach = r["algorithmic_gbs"]
    tr = load_traffic(f"{coeff}_4m:tiled")
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": tr["hbm_bytes_per_launch"] if tr else None, "kernel": "wrench_tiled_kernel", "kernel_us": r["us_per_step"],
            "bodies": r["n"], "algorithmic_bytes_per_launch": r["n"] * BYTES_PER_BODY[coeff],
            "frac_traffic": r["n"] * TRAFFIC_BYTES_PER_BODY[coeff] / (r["us_per_step"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "working_set_bytes": r["working_set_bytes"], "resident": r["resident"], "steps": steps}


def load_traffic(workload: str):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE doubled per the gfx950
    correction).  None when not measured."""
    try:
        with open(os.path.join(REPO, "profiles", "traffic.json")) as f:
            return json.load(f).get(workload)
    except Exception:
        return None


# ---- the one line -----------------------------------------------------------------------------------------------------------

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline")
def _round(x, sig: int = 12):
    """Numbers as short as they are meaningful: floats to `sig` significant digits, recursively; everything else untouched."""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if math.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _round(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round(v, sig) for v in x]
    return x


def render_line(out: dict) -> str:
    """The JSON line for `out`: compact separators, floats to 12 significant digits (`value` and `ms_per_step` in full), and at
    most LINE_LIMIT bytes BY CONSTRUCTION: should a line ever exceed it (this file's own blocks fit twice over), whole
    non-contract blocks are dropped, the largest first, and named in `dropped_for_size`; the contract keys always stay."""
    keep_full = {k: out[k] for k in ("value", "ms_per_step") if k in out}
    cur = dict(_round(out), **keep_full)
    dumps = lambda x: json.dumps(x, separators=(",", ":"), allow_nan=False)      # noqa: E731
    line, dropped = dumps(cur), []
    while len(line.encode()) > LINE_LIMIT:
        rest = [k for k in cur if k not in CONTRACT_KEYS and k != "dropped_for_size"]
        if not rest:
            raise ValueError(f"bench line of {len(line)} bytes with the contract keys alone")
        big = max(rest, key=lambda k: len(dumps(cur[k])))
        del cur[big]
        dropped.append(big)
        cur["dropped_for_size"] = dropped
        line = dumps(cur)
    return line


def write_all(fd: int, data: bytes) -> None:
    """os.write may write less than it is given (a pipe): loop until every byte is out."""
    view = memoryview(data)
    while view:
        view = view[os.write(fd, view):]


def write_side_file(path: str, payload: dict) -> str | None:
    """The side file (full-precision line + extras) beside the script, or under the temp dir when that is not writable;
    echoed to stderr either way.  Returns the path written, None if nowhere."""
    import tempfile
    payload = _round(payload, 17)                           # (non-finite floats -> null: strict JSON)
    text = json.dumps(payload, indent=1, allow_nan=False, default=repr)
    sys.stderr.write("bench.py: side file (extras):\n" + json.dumps(payload, allow_nan=False, default=repr) + "\n")
    sys.stderr.flush()
    for p in (path, os.path.join(tempfile.gettempdir(), os.path.basename(path))):
        try:
            with open(p, "w") as f:
                f.write(text + "\n")
            return p
        except OSError:
            continue
    return None


def self_launch(n: int) -> int:
    """Run this script as `n` ranks of one node (one process per GPU, RCCL between them) and relay rank 0's JSON line AND
    the outcome: 0, or EXIT_LEG_FAILED when the line says "ok": false, or the child's code when there is no line.
    Nothing here initialises the GPU: `torch.cuda.device_count()` only counts."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    share = os.environ.get("HYDRO_BENCH_SHARE_GPU") == "1"       # rehearsal: several ranks on GPU 0 (with HYDRO_DIST_BACKEND=gloo)
    if ndev < n and not (share and ndev >= 1):
        sys.stderr.write(f"bench.py: --gpus {n} but only {ndev} GPU(s) are visible; refusing to benchmark fewer GPUs than asked "
                         f"(set HYDRO_BENCH_SHARE_GPU=1 HYDRO_DIST_BACKEND=gloo to rehearse the multi-rank path on one GPU)\n")
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    sys.stderr.write(f"bench.py: --gpus {n} without a torchrun environment: launching {' '.join(cmd[1:8])} ...\n")
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for cand in res.stdout.splitlines():
        cand = cand.strip()
        if cand.startswith("{") and '"metric"' in cand:
            line = cand
    if line is None:
        sys.stderr.write(res.stdout)
        sys.stderr.write(f"bench.py: the {n}-rank run failed (exit code {res.returncode}, no JSON line)\n")
        return res.returncode or EXIT_LEG_FAILED
    d = json.loads(line)
    if d.get("n_gpus") != n:
        sys.stderr.write(f"bench.py: the child reported n_gpus={d.get('n_gpus')}, expected {n}\n")
        return 4
    sys.stdout.write(line + "\n")                           # a line that is out is relayed whatever the exit code: it says "ok" itself
    sys.stdout.flush()
    if d.get("ok") is False or res.returncode != 0:
        sys.stderr.write(f"bench.py: the {n}-rank run exited with {res.returncode}, ok={d.get('ok')}: relaying its line and failing\n")
        return EXIT_LEG_FAILED
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="c5", choices=sorted(WORKLOADS))
    ap.add_argument("--bodies", type=int, default=0, help="bodies per GPU (default: the workload's)")
    ap.add_argument("--scenes", type=int, default=4, help="scene replicas stepped round-robin per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip scripts/bench_extras.py (side file) and the `box` block")
    ap.add_argument("--extras-out", default=os.path.join(REPO, "bench_extras.json"), help="side file: the full-precision line + the secondary measurements")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (C2 / C3 / C4, eager and graph)")
    ap.add_argument("--no-roofline-4m", action="store_true", help="skip the 4 194 304-body second roofline object (N=1 default run)")
    ap.add_argument("--no-strong-leg", action="store_true", help="N>1: skip the configs[3] strong-scaling leg")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 child runs (N=1 default workload); use profiles/traffic.json")
    ap.add_argument("--extras-budget-seconds", type=float, default=200.0,
                    help="secondary measurements are skipped once this much time has gone into them")
    ap.add_argument("--bodies-per-lane", type=int, default=0)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every GPU gets the workload's bodies (default); strong: the workload's bodies are block-partitioned over the GPUs")
    ap.add_argument("--spinup-seconds", type=float, default=1.0, help="untimed run of the step loop before the W warm-up steps (GPU clock ramp)")
    ap.add_argument("--layout", default="tiled", choices=["tiled", "soa", "aos"],
                    help="tiled = engine-native tiled SoA (hydro_step_wrench_tiled); soa = plain field pointers; "
                         "aos = the simulator's (N,3)/(N,4)/(N,6) tensors (hydro_step_wrench_aos, 168 B per body-step)")
    ap.add_argument("--explain", action="store_true", help="print what the fields of the line and of the side file mean, and exit")
    args = ap.parse_args()

    if args.explain:
        from scripts.bench_extras import FIELD_NOTES
        print(FIELD_NOTES)
        return
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) <= 1 and os.environ.get("HYDRO_BENCH_FORCE_GROUP") != "1":
        # `python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves, BEFORE anything in this
        # process touches the GPU.  Never fall through to a one-GPU run that would report n_gpus = 1.
        raise SystemExit(self_launch(args.gpus))

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner on stdout), so
    # file descriptor 1 is pointed at stderr for the whole run and the line goes to a saved copy of the real stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank, local_rank, world = hd.env_rank_world()
    if world != max(1, args.gpus) and not (world == 1 and os.environ.get("HYDRO_BENCH_FORCE_GROUP") == "1"):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or plain `python bench.py --gpus N`, "
                         f"which starts them itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # HYDRO_BENCH_FORCE_GROUP=1 (with HYDRO_DIST_ALWAYS=1): WORLD_SIZE=1 still builds a one-rank process group and takes
    # the N > 1 code path - how a single-GPU box runs the real RCCL calls (tests/test_rccl_single_rank_gpu.py)
    force_group = os.environ.get("HYDRO_BENCH_FORCE_GROUP") == "1"
    hd.init_process_group(force=force_group, node_barrier=True)
    multi = world > 1 or force_group
    ndev = torch.cuda.device_count()
    share = os.environ.get("HYDRO_BENCH_SHARE_GPU") == "1"
    if world > 1 and not share and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: local_rank {local_rank} but only {ndev} GPU(s) visible")
    dev = torch.device("cuda", (local_rank % ndev) if world > 1 else 0)
    torch.cuda.set_device(dev)
    # The ranks that REALLY take part in a collective of the live group: a run asked for --gpus N that joins fewer is refused.
    live_ranks = hd.live_ranks(dev) if multi else 1
    if live_ranks != max(1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but {live_ranks} rank(s) joined the process group's all-reduce: refusing to report")

    kind, n_default, coeff, desc = WORKLOADS[args.workload]
    n = args.bodies or n_default
    if n != n_default:
        desc = f"{desc} [overridden: {n} bodies/GPU]"
    if args.scaling == "strong" and world > 1:
        full = build_scene(kind, n, seed=5)                      # same scene on every rank ...
        sc = full.shard(rank, world)                             # ... each keeps its contiguous block
        desc = f"{desc} [strong scaling: {n} bodies over {world} GPUs]"
    else:
        sc = build_scene(kind, n, seed=5 + rank)
    if args.layout == "aos":
        from scripts.bench_extras import AosReplica
        cls = AosReplica
        if args.scenes * n * 52 < (410 << 20):
            args.scenes = max(args.scenes, -(-(410 << 20) // (n * 52)))       # (see bench_extras.aos_rate: no resident rows)
    else:
        cls = Replica
    replicas = [cls(sc, coeff, dev, roll=r * 131071, layout=args.layout) for r in range(args.scenes)]
    if args.bodies_per_lane:
        for r in replicas:
            r.engine.set_tuning(args.bodies_per_lane)
    stream = torch.cuda.Stream(dev)

    spin_up(replicas, stream, args.spinup_seconds)
    wall, ev_ms = timed_steps(replicas, args.steps, args.warmup, stream, multi)
    n_all = torch.tensor([float(sc.n)], dtype=torch.float64, device=hd.collective_device(dev) if multi else "cpu")
    hd.all_reduce_sum_(n_all)
    value = float(n_all.item()) * args.steps / wall
    kernel_us = ev_ms * 1e3 / args.steps              # HIP events on the launch stream around the K timed steps
    per_rank = hd.gather_rows([timed_steps.last_local_wall * 1e6 / args.steps, kernel_us], dev) if multi else None
    step_us = wall * 1e6 / args.steps                 # the interval `value` and `ms_per_step` are computed from
    bpb = BYTES_PER_BODY[coeff] + (24 if args.layout == "aos" else 0)        # the AoS entry also updates the engine's previous velocity
    tpb = bpb if args.layout == "aos" else TRAFFIC_BYTES_PER_BODY[coeff]
    # ONE clock for `value` and `roofline.frac`: algorithmic bytes per launch / (timed interval / K)
    achieved = sc.n * bpb / (step_us * 1e-6) / 1e9

    # the one collective of the path: global kinetic energy (every rank reduces its shard on device) ...
    with torch.cuda.stream(stream):
        ke = replicas[0].kinetic_energy()
    stream.synchronize()
    ke = ke.to(hd.collective_device(dev))
    t0 = time.perf_counter()
    hd.global_kinetic_energy(ke)
    torch.cuda.synchronize(dev)
    ke_us = (time.perf_counter() - t0) * 1e6
    # ... checked against float64 host sums gathered exactly (nothing of the reference value went through the all-reduce)
    host_ke = hd.gather_rows(scenes.kinetic_energy_fp64(sc.state, sc.params, rotational=True), dev)
    host_ke = [math.fsum(host_ke[:, k].tolist()) for k in range(2)]
    ke_rel_err = max(abs(float(ke[k]) - host_ke[k]) / host_ke[k] for k in range(2))

    out, traffic = None, None
    if rank == 0:
        traffic = load_traffic(f"{args.workload}:{args.layout}") if world == 1 and not args.bodies else None
        on_gpu = hd.collective_device(dev).type == "cuda"
        out = {
            "metric": "body-steps/sec", "value": value, "unit": "body-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic", "ok": True,
            "config": {"workload": desc, "baseline_config": BASELINE_CONFIG.get(args.workload, "variant"),
                       "bodies_per_gpu": sc.n, "coefficients": coeff,
                       "scene_replicas_per_gpu": args.scenes, "bytes_per_body_step": bpb,
                       "sharding": f"bodies x{world} (no data-path collective)",
                       "layout": {"tiled": "tiled SoA [tile][field][64]", "soa": "plain SoA", "aos": "array-of-structs tensors"}[args.layout],
                       "entry_point": {"tiled": "hydro_step_wrench_tiled", "soa": "hydro_step_wrench_ext", "aos": "hydro_step_wrench_aos"}[args.layout]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
                         "kernel": {"tiled": "wrench_tiled_kernel", "soa": "wrench_soa_kernel", "aos": "wrench_aos_direct_kernel"}[args.layout],
                         "step_us": step_us, "kernel_us": kernel_us,
                         "frac_contract_steps": sc.n * bpb / (kernel_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": sc.n * bpb,
                         "traffic_bytes_per_body": tpb,
                         "frac_traffic": sc.n * tpb / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "traffic_measured": "committed" if traffic else None,
                         **residency(sc.n, coeff, args.scenes)},
            "max_rel_err": None,
            "collective": {"backend": (f"{'nccl (RCCL)' if on_gpu else 'gloo'}" if multi else "none"), "ranks": live_ranks,
                           "rccl_ranks": live_ranks if multi and on_gpu else 0, "barrier": hd.barrier_kind(),
                           "global_ke_J": [float(x) for x in ke.cpu().tolist()], "host_fp64_ke_J": host_ke,
                           "ke_rel_err": ke_rel_err, "ke_allreduce_us": ke_us},
        }
        if per_rank is not None:
            out["per_rank"] = {"step_us": [float(x) for x in per_rank[:, 0]], "kernel_us": [float(x) for x in per_rank[:, 1]]}

    # N > 1: BASELINE configs[3] as stated (262 144 bodies over the N GPUs) on every rank.  The headline is complete at this
    # point; whatever happens to this leg, rank 0 still prints it (LegGuard) - and the exit code says so.
    strong, guard = None, None
    if multi and not args.no_strong_leg:
        from scripts import bench_strong
        guard = bench_strong.LegGuard(rank, out, json_fd, float(os.environ.get("HYDRO_BENCH_STRONG_TIMEOUT", bench_strong.STRONG_LEG_TIMEOUT_S)))
        if rank == 0 and world == 1 and args.cpu_seconds > 0:    # (a one-rank group: the CPU leg applies, and is measured BEFORE the leg)
            out["cpu_baseline"] = guard.extra["cpu_baseline"] = _cpu_leg(sc, replicas, args)
        strong = bench_strong.guarded_strong_leg(rank, world, dev, stream, args, multi, guard)

    side = {}
    if rank == 0:
        rf = out["roofline"]
        if traffic and not args.no_live_traffic and args.layout == "tiled" and world == 1:
            from scripts.bench_extras import measure_traffic_live
            live, why_not = measure_traffic_live()
            if live is None:
                side["traffic_live_skipped"] = why_not
            else:
                side["traffic_live"] = live
                rf.update(traffic_committed=traffic["hbm_bytes_per_launch"], traffic=live["hbm_bytes_per_launch"], traffic_measured="live",
                          traffic_bytes_per_body_measured=live["hbm_bytes_per_launch"] / sc.n,
                          frac_traffic=live["hbm_bytes_per_launch"] / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS)
        if "cpu_baseline" not in out:
            out["cpu_baseline"] = _cpu_leg(sc, replicas, args) if world == 1 and args.cpu_seconds > 0 else None
        if out["cpu_baseline"] and out["cpu_baseline"].get("value"):
            out["max_rel_err"] = out["cpu_baseline"]["gpu_vs_oracle_max_rel_err"]
        if strong is not None:
            out["c4_strong"] = strong
        if world == 1 and args.workload == "c5" and not args.bodies and not args.no_roofline_4m:
            try:
                out["roofline_4m"] = roofline_4m(dev, stream)
            except Exception as e:                          # noqa: BLE001 - report, never lose the line
                out["roofline_4m"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_extras:
            try:                                            # median of 5 runs (each 200 steps after 20 warm-up steps), same replicas
                runs = sorted(timed_steps(replicas, 200, 20, stream)[1] * 1e3 / 200 for _ in range(5))
                side["kernel_us_5x200_runs"] = runs
                rf["kernel_us_median_of_5"] = runs[2]
                rf["frac_median_of_5"] = sc.n * bpb / (runs[2] * 1e-6) / 1e9 / HBM_PEAK_GBS
            except Exception as e:                          # noqa: BLE001
                side["kernel_us_5x200_runs"] = repr(e)
        for r in replicas:
            r.engine.close()
        replicas = []
        torch.cuda.empty_cache()
        if world == 1 and not args.no_configs:
            out["configs"] = configs_block(dev, stream)
        if world == 1 and not args.no_extras:
            try:
                from scripts import bench_extras
                side["extras"] = bench_extras.run(dev, stream, args.extras_budget_seconds)
                box = bench_extras.box_summary(side["extras"])
                if box:
                    out["box"] = box
            except Exception as e:                          # noqa: BLE001 - the secondary measurements never cost the line
                side["extras"] = {"error": repr(e)}
            side["line"] = out
            out["extras_file"] = os.path.basename(write_side_file(args.extras_out, side) or "") or None
        sys.stdout.flush()
        write_all(json_fd, (render_line(out) + "\n").encode())
        if guard is not None:
            guard.line_is_out()

    # (the line is out) a rank that never arrives here must not look like a pass: the teardown barrier has a deadline, and a
    # rank that gives up on it leaves THROUGH the guard (rank 0's line first, then exit code 3 everywhere)
    try:
        hd.barrier(timeout_s=float(os.environ.get("HYDRO_BENCH_TEARDOWN_TIMEOUT", "60")))
        if guard is not None:
            guard.finish()
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
    except Exception as e:                                  # noqa: BLE001
        sys.stderr.write(f"bench.py: rank {rank}: teardown: {e!r}\n")
        sys.stderr.flush()
        if guard is not None:
            guard.leave(f"teardown: {e!r} on rank {rank}; the headline on this line is complete")
        os._exit(EXIT_LEG_FAILED)


def _cpu_leg(sc, replicas, args):
    try:
        return cpu_baseline_leg(sc, _last_stepped(replicas, args.steps), args.cpu_seconds)
    except Exception as e:                                  # noqa: BLE001 - report, never lose the line
        return {"value": None, "unit": "body-steps/s", "cores": 1, "kind": "port", "sample": "failed", "error": repr(e)[:200]}


def _last_stepped(replicas, steps):
    """Replica whose output buffer holds the result of a completed step."""
    return replicas[(steps - 1) % len(replicas)] if steps > 0 else replicas[0]


if __name__ == "__main__":
    main()
