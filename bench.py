#!/usr/bin/env python3
"""Benchmark of the fused hydrodynamic-wrench step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one launch of `hydro_step_wrench_ext` over every body of one scene
replica on this rank's GPU: finite-difference acceleration + nine-component
model + lever arms + sum + clamp.  Inputs are resident in HBM
before the timed region.  Bodies shard embarrassingly: every rank owns its own
replica(s) of the workload (weak scaling), there is no data-path collective;
the one RCCL call is the global kinetic-energy all-reduce after the timed loop.

Workload (default `c5` = BASELINE.json configs[4], the configuration the
"% of HBM roofline" part of the metric is quoted on): 1 048 576 synthetic bodies
per GPU, fp32 state, 7 coefficients stored fp16, fp64 arithmetic (the reference's
Numba path is float64; results are rounded to fp32 once), 130 algorithmic bytes per
body-step.  `--scenes` (default 4) independent scene replicas are stepped
round-robin so that the bytes touched between two uses of any line exceed the
256 MiB Infinity Cache (cache caveat): the rate is an HBM rate.
The same line carries `roofline_4m` - the same kernel on 4 194 304 bodies
(two rotating replicas, 1.1 GB), a size no cache can assist - and, for N > 1,
`c4_strong`: BASELINE.json configs[3] as stated (262 144 bodies block-partitioned
over the N GPUs) with the global kinetic energy sampled at least twice inside the
timed region (every min(256, K // 2) steps) through `simulate.KineticEnergyMonitor`
(device reduction + asynchronous all-reduce on a side stream).  That leg checks
itself: `kinetic_energy.rel_err_vs_host_fp64` (gate 1e-12), `shards_bit_identical`;
the line carries `rccl_ranks` (ranks that really joined an all-reduce of the live
group) and the run exits non-zero when that differs from --gpus.  Beside it,
`c4_strong.graph_resident_sampling`: the same leg with each sample's pipeline (RCCL
all-reduce and pinned copy) captured into the step graph - no host work per sample.
The headline is complete before that leg starts and is protected from it
(`guarded_strong_leg`: a watchdog prints it with `c4_strong.error` should the leg hang).

Prints ONE JSON line on rank 0.
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

from silver2_isaacsim_amd import distributed as hd          # noqa: E402
from silver2_isaacsim_amd import scenes                     # noqa: E402
from silver2_isaacsim_amd.engine import HydroEngine         # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_COPY_CEILING_GBS = 6290.0    # measured float4-copy ceiling, same guide
BYTES_PER_BODY = {"f32": 144, "f16": 130}    # SURVEY.md 8d / BASELINE.md 3 (the algorithmic figure `frac` uses)
# what the kernel really moves (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/traffic.json): p_x, p_y are provably
# unused by the wrench and never loaded, 8 B under the algorithmic figure
TRAFFIC_BYTES_PER_BODY = {"f32": 136, "f16": 122}
INFINITY_CACHE_BYTES = 256 << 20
# resident bytes of one scene replica per body: state 52 + previous velocity 24 + parameters + wrench 24
REPLICA_BYTES_PER_BODY = {"f32": 52 + 24 + 44 + 24, "f16": 52 + 24 + 30 + 24}

WORKLOADS = {
    # name: (scene builder kwargs, coefficient dtype, description)
    "c5": ("c5", 1048576, "f16", "C5: 1 048 576 bodies/GPU, fp32 state, fp16-stored coefficients, fp64 arithmetic rounded to fp32 once"),
    "c5-f32": ("c4", 1048576, "f32", "C5 population with fp32 coefficients (144 B/body-step)"),
    "c4": ("c4", 262144, "f32", "C4: 262 144 bodies (per GPU under weak scaling)"),
    "c3": ("c3", 19456, "f32", "C3: SILVER2 hexapod x 1024 envs"),
    "c2": ("c2", 4096, "f32", "C2: 4 096 buoys"),
}


# which entry of BASELINE.json "configs" a workload is (C5 = configs[4] is the one whose line asks for
# "achieved HBM GB/s vs peak", i.e. the roofline part of the metric; configs[1..3] are launch-bound
# parity cases and are reported in `extras`)
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
    """Untimed: run the same step loop for `seconds` so that the GPU has left its idle power
    state before anything is measured.  Measured on MI355X: the first ~50 ms after idle run
    ~9 % slower (25.1 vs 23.0 us per C5 step); W warm-up steps of a ~25 us kernel are far
    shorter than that.  This happens BEFORE the W warm-up steps and the K timed steps."""
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
    after_step(k, replica): called inside the timed region after step k (1-based).
    The launch stream is made current BEFORE the region (entering torch's stream context costs the host ~10 us, which is
    2 % of a 20-step region and none of the K steps); without a process group the barrier is a no-op and the
    synchronize after it is skipped."""
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
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota (a GPU box hands a
    share of its host to each job; running 128 OpenMP threads on a 16-CPU share only measures thrashing)."""
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
    """The C oracle (oracle/hydro_oracle.c, a port of the reference's Numba path) timed on this
    box's host cores on a bounded sample of the same workload, and used as the checker of the
    GPU result for that sample.  Only this function touches oracle/."""
    from oracle import c_oracle, hydro_oracle
    m = min(sc.n, 262144)
    idx = replica.index[:m]
    st, pv, pr = sc.state[idx], sc.prev[idx], sc.params[idx]
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
        "sample": f"{m} bodies of the bench scene x {reps} passes, fp64 C port of the Numba path "
                  f"(oracle/hydro_oracle.c, gcc -O3 -ffast-math), 1 thread",
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
    """Small untimed-contract measurement for the 'extras' block (not the headline)."""
    sc = build_scene(kind, n, seed)
    reps = [Replica(sc, coeff, dev, roll=r * 97, layout=layout) for r in range(sets)]
    spin_up(reps, stream, 0.15)
    _, ms = timed_steps(reps, steps, 10, stream)
    for r in reps:
        r.engine.close()
    us = ms * 1e3 / steps
    return {"n": sc.n, "coeff": coeff, "layout": layout, "us_per_step": us, "body_steps_per_s": sc.n / (us * 1e-6),
            "algorithmic_gbs": sc.n * BYTES_PER_BODY[coeff] / (us * 1e-6) / 1e9, **residency(sc.n, coeff, sets)}


def batch_rate(kind: str, n: int, coeff: str, dev, stream, scenes_per_launch: int = 4, sets: int = 2, steps: int = 100, seed: int = 11):
    """NOT the headline protocol: `scenes_per_launch` independent scenes of n bodies stepped by ONE launch
    (hydro_step_wrench_tiled_batch), `sets` such groups rotating; beside it the same scenes as single launches, one
    after the other on the same stream.  Same bits either way (tests/test_parity_gpu.py); the difference is the ramp
    and drain a launch pays once instead of `scenes_per_launch` times.  HIP events on the launch stream."""
    sc = build_scene(kind, n, seed)
    k = scenes_per_launch
    groups = [[Replica(sc, coeff, dev, roll=(g * k + j) * 97) for j in range(k)] for g in range(sets)]
    with torch.cuda.stream(stream):
        batched = [HydroEngine.prepare_step_wrench_tiled_batch([r.engine for r in grp], [r.state for r in grp], sc.dt,
                                                               outs=[r.out for r in grp], prevs=[r.prev for r in grp])[0] for grp in groups]
        for grp in groups:
            for r in grp:
                r.step()
        spin_up([r for grp in groups for r in grp], stream, 0.15)

        def timed(fn):
            for w in range(10):
                fn(w)
            samples = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for it in range(steps):
                    fn(it)
                e1.record(stream)
                stream.synchronize()
                samples.append(e0.elapsed_time(e1) * 1e3 / steps)
            return sorted(samples)[len(samples) // 2]

        def singles(it):
            for r in groups[it % sets]:
                r.step()
        us_single = timed(singles)
        us_batch = timed(lambda it: batched[it % sets]())
    for grp in groups:
        for r in grp:
            r.engine.close()
    per = k * sc.n * BYTES_PER_BODY[coeff]
    return {"n_per_scene": sc.n, "scenes_per_launch": k, "coeff": coeff, "rotating_groups": sets,
            "us_per_group_as_single_launches": us_single, "us_per_group_one_launch": us_batch,
            "frac_single_launches": per / (us_single * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "frac_one_launch": per / (us_batch * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "body_steps_per_s_one_launch": k * sc.n / (us_batch * 1e-6), **residency(sc.n, coeff, sets * k)}


def two_stream_rate(kind: str, n: int, coeff: str, dev, steps: int = 400, sets: int = 4, seed: int = 11):
    """NOT the headline protocol: the rotating replicas are independent scenes; stepped round-robin on TWO streams (even /
    odd replicas) the drain of one launch overlaps the ramp of the next.  The difference to the one-stream figure of the
    same run is what ramp and drain cost a launch; per-kernel durations of overlapping launches
    are meaningless, so this entry reports throughput only (HIP events from the first launch to the join of both streams)."""
    sc = build_scene(kind, n, seed)
    S = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    one = [Replica(sc, coeff, dev, roll=r * 97) for r in range(sets)]
    two = [Replica(sc, coeff, dev, roll=r * 97) for r in range(sets)]
    with torch.cuda.stream(S[0]):                       # (a prepared launch is bound to the stream current at its first call)
        for r in one:
            r.step()
    for k, r in enumerate(two):
        with torch.cuda.stream(S[k % 2]):
            r.step()
    spin_up(one, S[0], 0.15)
    us = {}
    for mode, reps in (("one_stream", one), ("two_streams", two)):
        samples = []
        for _ in range(5):
            torch.cuda.synchronize(dev)
            e0, e1, ej = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
            for timed in (False, True):
                if timed:
                    e0.record(S[0])
                    S[1].wait_event(e0)
                for k in range(steps if timed else steps // 4):
                    reps[k % sets].step()               # (bound to its stream above: no stream context per launch)
            ej.record(S[1]); S[0].wait_event(ej)
            e1.record(S[0]); e1.synchronize()
            samples.append(e0.elapsed_time(e1) * 1e3 / steps)
        us[mode] = float(np.median(samples))
    for r in one + two:
        r.engine.close()
    gbs = sc.n * BYTES_PER_BODY[coeff] / (us["two_streams"] * 1e-6) / 1e9
    return {"n": sc.n, "coeff": coeff, "us_per_step_one_stream": us["one_stream"], "us_per_step_two_streams": us["two_streams"],
            "ramp_and_drain_us_per_launch": us["one_stream"] - us["two_streams"],
            "body_steps_per_s_two_streams": sc.n / (us["two_streams"] * 1e-6), "algorithmic_gbs_two_streams": gbs,
            "frac_two_streams": gbs / HBM_PEAK_GBS,
            "note": "throughput of independent scenes on two streams; the headline, its roofline object and the profiles stay one stream, one launch at a time",
            **residency(sc.n, coeff, sets)}


def graph_rate(kind: str, n: int, coeff: str, dev, stream, steps_per_graph: int = 64, replays: int = 40, seed: int = 11):
    """Launch-bound small scenes: capture `steps_per_graph` consecutive steps (round-robin over 4
    scene replicas) into one HIP graph and replay it - one host call per 64 physics steps instead
    of one per step.  The C-ABI step functions are capture-safe (no allocation, no sync)."""
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
    return {"n": sc.n, "coeff": coeff, "layout": "tiled", "mode": f"hipGraph x{steps_per_graph} steps",
            "us_per_step": us, "body_steps_per_s": sc.n / (us * 1e-6),
            "algorithmic_gbs": sc.n * BYTES_PER_BODY[coeff] / (us * 1e-6) / 1e9, **residency(sc.n, coeff, 4)}


class AosReplica:
    """One scene replica as the simulator's tensor API hands it over - positions (N,3), orientations (N,4) wxyz,
    velocities (N,6) - with the previous velocity and the parameters (fp32) inside the engine: one
    hydro_step_wrench_aos launch per step (168 algorithmic bytes per body-step)."""

    def __init__(self, sc, coeff: str, dev, roll: int, layout: str = "aos"):
        idx = np.roll(np.arange(sc.n), roll)
        self.n, self.dt, self.index, self.layout = sc.n, sc.dt, idx, "aos"
        self.engine = HydroEngine(sc.n, dev, sc.rho, sc.g)
        self.engine.set_params(sc.params[idx], coeff)
        self.engine.set_prev_velocity(sc.prev[idx])
        st = sc.state[idx]
        self.pos = torch.from_numpy(np.ascontiguousarray(st[:, 0:3])).to(dev)
        self.quat = torch.from_numpy(np.ascontiguousarray(st[:, [6, 3, 4, 5]])).to(dev)
        self.vel = torch.from_numpy(np.ascontiguousarray(st[:, 7:13])).to(dev)
        self.force, self.torque = torch.empty((sc.n, 3), device=dev), torch.empty((sc.n, 3), device=dev)
        self.state = self.pos                                   # (spin_up / timed_steps only look at .state.device)
        self._prepared = None

    @property
    def out(self):
        return torch.cat([self.force, self.torque], dim=1)

    def wrench_rows(self, m: int) -> np.ndarray:
        return torch.cat([self.force[:m], self.torque[:m]], dim=1).cpu().numpy()

    def kinetic_energy(self):
        return self.engine.kinetic_energy(self.engine.pack_state_aos(self.pos, self.quat, self.vel), rotational=True)

    def step(self):
        if self._prepared is None:
            self._prepared = self.engine.prepare_step_wrench_aos(self.pos, self.quat, self.vel, forces=self.force, torques=self.torque)
        self._prepared(self.dt)


def aos_rate(n: int, dev, stream, steps: int = 100, sets: int = 8, seed: int = 13):
    """The simulator-facing entry (hydro_step_wrench_aos: (N,3)/(N,4)/(N,6) tensors in, forces/torques
    out, previous velocity kept in the engine): 168 algorithmic bytes per body-step, all of them real traffic.
    EIGHT rotating sets: the kernel reads the simulator's rows (52 B per body) with temporal loads, and four sets of them
    (218 MB) would sit in the 256 MiB Infinity Cache while everything else streams past - a cache rate (26.5 instead of
    29.8 us at 1 M bodies), not the HBM rate this entry is quoted at."""
    sc = build_scene("c4", n, seed)
    reps = [AosReplica(sc, "f32", dev, roll=r * 97) for r in range(sets)]
    spin_up(reps, stream, 0.15)
    _, ms = timed_steps(reps, steps, 10, stream)
    us = ms * 1e3 / steps
    for r in reps:
        r.engine.close()
    gbs = sc.n * 168 / (us * 1e-6) / 1e9
    return {"n": sc.n, "entry_point": "hydro_step_wrench_aos", "us_per_step": us, "body_steps_per_s": sc.n / (us * 1e-6),
            "algorithmic_gbs": gbs, "bytes_per_body_step": 168, "frac": gbs / HBM_PEAK_GBS,
            "rotating_sets": sets, "temporal_bytes_rotating": sets * sc.n * 52,
            **residency(sc.n, "f32", sets, 12 + 16 + 24 + 24 + 24 + 44)}


# VALU-issue roofline of the compute-bound path (the resident closed loop never touches HBM between steps).
# The PEAK is the hardware's issue rate, an upper bound by construction (MI355X_MICROARCH.md): a SIMD is 32 lanes wide, a
# wave64 VALU instruction issues over 2 cycles ("v_fma_f32 (wave64): 2 cyc"), fp64 arithmetic runs at half that rate (4 cycles:
# 78.6 TFLOP/s of vector fp64 against 157.3 of fp32) - and the clock is the HIGHEST the chip was ever read at in-kernel
# (2.55 GHz under arithmetic alone, `extras.clocks_1m.compute_only_ghz`; the spec's "max clock" of 2.4 GHz is not a bound,
# the chip boosts above it).  1 024 SIMDs x 2.55 GHz / 2 = 1 306 G wave-instructions/s for 2-cycle instructions.
# Round 4 priced the classes with scripts/ubench_valu.hip's own readings (fp64 4.2, fp32 2.7, the rest ~4 cycles) at 2.4 GHz:
# a MODEL of what the loop costs, not a bound - the driver's run read 1.03 of it.  It stays on the line as
# `model_measured_prices` (said to be a model), `frac` is against the hardware rate.
VALU_SPEC_CYCLES = {"fp64 arithmetic": 4.0, "fp32 arithmetic": 2.0, "conversion": 2.0, "compare": 2.0,
                    "integer / select / move": 2.0, "transcendental": 2.0}
VALU_MEASURED_CYCLES = {"fp64 arithmetic": 4.2, "fp32 arithmetic": 2.7, "conversion": 4.0, "compare": 4.0,
                        "integer / select / move": 4.0, "transcendental": 4.0}
SIMDS, SPEC_CLOCK_GHZ, BOOST_CLOCK_GHZ = 1024, 2.4, 2.55


def valu_roofline(kernel_prefix: str, n: int, us_per_step: float):
    """{"bound": "valu-issue", ...} for one step of a kernel whose instruction mix scripts/isa_mix.py recorded
    (profiles/isa_mix.json; tests/test_isa_budget.py keeps it current).  `frac` = the time the step's VALU instructions
    need at the hardware's issue rate and boost clock / the measured time: <= 1 on every box."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "isa_mix.json")
    try:
        kernels = json.load(open(path))["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    key = next((k for k in kernels if k.startswith(kernel_prefix)), None)
    if key is None:
        return None
    mix = kernels[key]["valu_by_class"]
    waves_per_simd = -(-n // 64) / SIMDS
    spec_cycles = sum(VALU_SPEC_CYCLES[c] * k for c, k in mix.items())
    floor_us = spec_cycles * waves_per_simd / (BOOST_CLOCK_GHZ * 1e3)
    model_cycles = sum(VALU_MEASURED_CYCLES[c] * k for c, k in mix.items())
    model_us = model_cycles * waves_per_simd / (SPEC_CLOCK_GHZ * 1e3)
    total = kernels[key]["valu_total"]
    return {"bound": "valu-issue", "kernel": key, "valu_instructions_per_body_step": total, "valu_by_class": mix,
            "waves_per_simd": waves_per_simd,
            "achieved": total * (n / 64) / (us_per_step * 1e-6) / 1e9,
            "peak": total / spec_cycles * SIMDS * BOOST_CLOCK_GHZ,
            "unit": "G wave-instructions/s of THIS instruction mix (peak: 2 cycles per wave64 instruction, 4 for fp64 arithmetic, "
                    "1 024 SIMDs at the 2.55 GHz boost clock)",
            "frac": floor_us / us_per_step,
            "issue_cycles_per_wave_step_at_hardware_rate": spec_cycles, "floor_us_per_step": floor_us,
            "frac_is": "time the step's VALU instructions need at the hardware issue rate (MI355X_MICROARCH.md: SIMD-32, wave64 in 2 "
                       "cycles, fp64 arithmetic in 4) and the highest clock ever read in-kernel (2.55 GHz) / measured time: an upper bound, <= 1",
            "model_measured_prices": {"issue_cycles_per_wave_step": model_cycles, "us_per_step_at_2.4GHz": model_us,
                                      "measured_over_model": us_per_step / model_us,
                                      "is": "a MODEL, not a bound: classes priced with scripts/ubench_valu.hip's readings (fp64 4.2, fp32 2.7, "
                                            "others ~4 cycles; they include that benchmark's own launch ramp) at the 2.4 GHz spec clock"}}


def closed_loop_rate(kind: str, n: int, steps: int = 4096, fused: bool = True, implicit_drag: bool = False, resident: bool = False):
    """Wrench + integrator ping-pong replayed from a HIP graph (simulate.ClosedLoopSim); RTF as
    benchmark_rtf.py defines it (sim time / wall time).  fused: one kernel per physics step
    (hydro_step_fused_tiled) instead of two.  resident: one launch per 64 physics steps, the bodies carried through them
    in registers (hydro_step_fused_tiled_multi; same bits) - no HBM traffic and no launch between the steps."""
    from silver2_isaacsim_amd.simulate import ClosedLoopSim
    sim = ClosedLoopSim(build_scene(kind, n, 17), fused=fused, implicit_drag=implicit_drag)
    r = sim.measure_rtf(steps, graph_steps=64, resident=resident, warm_seconds=0.25)      # sustained rate, as the headline's spin-up
    sim.close()
    mode = "hipGraph x64 (hydro_step_fused_tiled)" if fused else "hipGraph x64 (wrench_tiled + integrate_tiled)"
    if resident:
        mode = "64 steps per launch, bodies resident in registers (hydro_step_fused_tiled_multi)"
    if implicit_drag:
        mode += ", implicit drag"
    # ONE scene stepping on itself: state ping-pong (2 x 52 B) + parameters
    out = {"n": n, "mode": mode, **r, **residency(n, "f32", 1, 2 * 52 + 44)}
    if resident:          # compute-bound (no HBM traffic between the steps): its roofline is VALU issue, not bytes
        vr = valu_roofline("resident closed loop, one step, implicit drag" if implicit_drag else "resident closed loop, one step (", n, r["us_per_step"])
        if vr:
            out["roofline"] = vr
    return out


def roofline_4m(dev, stream, coeff: str = "f16", n: int = 4194304, sets: int = 2, steps: int = 200):
    """Second roofline object: the headline kernel on 4 194 304 bodies, two rotating replicas (1.1 GB): nothing of it
    survives in the 256 MiB Infinity Cache between two uses, and ramp and drain of a launch weigh a quarter of what
    they do at 1 048 576.  Same timing rule as the headline (HIP events on the launch stream over the timed steps)."""
    r = quick_rate("c5" if coeff == "f16" else "c4", n, coeff, dev, stream, steps=steps, sets=sets, seed=5)
    ach = r["algorithmic_gbs"]
    tr = load_traffic(f"{coeff}_4m:tiled")
    out = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
           "traffic": tr["hbm_bytes_per_launch"] if tr else None, "kernel": "wrench_tiled_kernel", "kernel_us": r["us_per_step"],
           "bodies": r["n"], "coefficients": coeff, "algorithmic_bytes_per_launch": r["n"] * BYTES_PER_BODY[coeff],
           "traffic_bytes_per_body": TRAFFIC_BYTES_PER_BODY[coeff],
           "frac_traffic": r["n"] * TRAFFIC_BYTES_PER_BODY[coeff] / (r["us_per_step"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
           "working_set_bytes": r["working_set_bytes"], "resident": r["resident"], "steps": steps,
           "scene": "4 permuted copies of the 1 048 576 distinct seed-5 bodies"}
    if tr:
        out["traffic_source"] = tr.get("source")
    return out


def strong_leg_cadence(steps: int) -> tuple[int, int]:
    """(ke_every, graph_steps) of the c4_strong leg for a timed region of `steps` steps: at least TWO kinetic-energy
    samples inside the region whatever `steps` is (the driver times 20), at most one per 256 steps, and a HIP graph of
    `graph_steps` <= 64 consecutive steps that divides `ke_every` (the sampling step is the last step of a replay).
    steps 20 -> (10, 10); 600 -> (256, 64); 2000 -> (256, 64); 300 -> (128, 64); 1 -> (1, 1)."""
    every = max(1, min(256, steps // 2))
    graph = min(64, every)
    return every // graph * graph, graph


def wrench_digest(rows: np.ndarray) -> list[int]:
    """32-byte digest of an (m,6) float32 wrench block, as 32 integers (what the ranks exchange to prove shard == unsharded)."""
    import hashlib
    return list(hashlib.blake2b(np.ascontiguousarray(rows, dtype=np.float32).tobytes(), digest_size=32).digest())


def c4_strong_leg(rank: int, world: int, dev, stream, steps: int, warmup: int, collectives: bool = True, progress: dict | None = None):
    """BASELINE.json configs[3] as it is stated: 262 144 bodies (seed 4) block-partitioned over the GPUs, every rank
    steps its contiguous shard (no data-path collective); the global kinetic energy is sampled every `ke_every`
    steps - at least twice inside the timed region, see strong_leg_cadence - by simulate.KineticEnergyMonitor: device
    reduction inside the step kernel, asynchronous all-reduce (RCCL under backend nccl) on a side stream, results picked
    up later by the host.  Same barrier / max-over-ranks timing as the headline.  After the region the leg PROVES itself:
    the last global sample against an fp64 host sum over all 262 144 bodies (`rel_err_vs_host_fp64`), and every rank's
    shard wrench against the unsharded scene stepped once on rank 0 (`shards_bit_identical`, digests exchanged through
    the same collective)."""
    from silver2_isaacsim_amd.simulate import KineticEnergyMonitor
    full = build_scene("c4", 262144, 4)                      # the same scene on every rank ...
    sc = full.shard(rank, world)                             # ... each keeps its contiguous block
    reps = [Replica(sc, "f32", dev, roll=0) for _ in range(2)]      # two buffer sets of the SAME shard (cache-resident sizes)
    ke_every, GRAPH_STEPS = strong_leg_cadence(steps)
    mon = KineticEnergyMonitor(reps[0].engine, every=ke_every)
    ke_dev = torch.zeros(2, dtype=torch.float64, device=dev)         # where the sampling step leaves the shard's pair
    with torch.cuda.stream(stream):
        mon.warm_up(stream)                                           # (its first pass costs 0.4 ms of one-time set-up: not in the region)
    spin_up(reps, stream, 0.3)
    # A shard of 32 768 bodies is one 2.7 us launch: issued one by one the loop is bound by the host call (3.4 us), so
    # GRAPH_STEPS consecutive steps are captured into one HIP graph (the step functions are capture-safe) and the K
    # timed steps are K // GRAPH_STEPS replays plus K % GRAPH_STEPS eager steps.  Where a sample is due at the end of a
    # replay, the replayed graph is the one whose LAST step is the kernel variant that also samples the kinetic energy of the
    # bodies it holds (no extra pass, no extra launch), and the monitor picks the pair up after it; the other replays are
    # of a graph of plain steps.
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        reps[1].step_sampling(ke_dev)                                 # (prepare outside the capture)
        reps[0].step_sampling(ke_dev)
        stream.synchronize()
        with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
            for k in range(GRAPH_STEPS):
                if k == GRAPH_STEPS - 1:
                    reps[k % 2].step_sampling(ke_dev)
                else:
                    reps[k % 2].step()
        g.replay()
        g_plain = None
        if ke_every > GRAPH_STEPS:                              # replays at whose end no sample is due: plain steps only
            g_plain = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_plain, stream=stream, capture_error_mode="thread_local"):
                for k in range(GRAPH_STEPS):
                    reps[k % 2].step()
            g_plain.replay()
        stream.synchronize()

    def run(k_steps, observe):
        done = 0
        for _ in range(k_steps // GRAPH_STEPS):
            due = (done + GRAPH_STEPS) % ke_every == 0
            (g if due or g_plain is None else g_plain).replay()
            done += GRAPH_STEPS
            if due and observe:
                mon.observe(done, stream=stream, sampled=ke_dev)
        for k in range(k_steps % GRAPH_STEPS):
            done += 1
            sample = observe and done % ke_every == 0
            if sample:
                mon.wait_before_overwrite(stream)
                reps[k % 2].step_sampling(ke_dev)
                mon.observe(done, stream=stream, sampled=ke_dev)
            else:
                reps[k % 2].step()
    with torch.cuda.stream(stream):                             # (made current before the region: see timed_steps)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream); ev1.record(stream)                  # (created by their first record: not inside the region)
        run(warmup, False)
        torch.cuda.synchronize(dev)
        hd.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ev0.record(stream)
        run(steps, True)
        ev1.record(stream)
        torch.cuda.synchronize(dev)
        hd.barrier()
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
    tmax = torch.tensor([wall], dtype=torch.float64, device=hd.collective_device(dev) if collectives else "cpu")
    hd.all_reduce_max_(tmax)
    wall, ev_ms = float(tmax.item()), float(ev0.elapsed_time(ev1))
    mon.collect(block=True)
    del g, g_plain
    last = mon.last()
    # ---- untimed: the leg checks itself ----
    # (1) the collective: the last global sample against the float64 host sum over ALL 262 144 bodies (the wrench
    #     step does not move the bodies, so every sample is the energy of the scene as built)
    host = scenes.kinetic_energy_fp64(full.state, full.params, rotational=True)
    rel = [abs(last[1][k] - host[k]) / host[k] for k in range(2)] if last else None
    # (2) the partition: this rank's shard wrench after the last step against the same bodies of the UNSHARDED scene
    with torch.cuda.stream(stream):
        reps[0].step()
    stream.synchronize()
    mine = gather_digests(wrench_digest(reps[0].wrench_rows(sc.n)), dev)
    identical = None
    if rank == 0:
        whole = Replica(full, "f32", dev, roll=0)
        with torch.cuda.stream(stream):
            whole.step()
        stream.synchronize()
        rows = whole.wrench_rows(full.n)
        whole.engine.close()
        identical = all(wrench_digest(rows[slice(*hd.shard_range(full.n, r, world))]) == mine[r] for r in range(world))
    result = {"value": full.n * steps / wall, "unit": "body-steps/s", "scaling": "strong", "baseline_config": "configs[3]",
            "bodies_total": full.n, "bodies_this_rank": sc.n, "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": wall * 1e3 / steps, "kernel_us_rank0": ev_ms * 1e3 / steps,
            "kinetic_energy": {"every_steps": ke_every, "samples": len(mon.samples), "host_waits": mon.waited_on_host,
                               "sampled_at_steps": [s for s, _ in mon.samples],
                               "last_step": last[0] if last else None, "global_J": last[1] if last else None,
                               "host_fp64_J": list(host), "rel_err_vs_host_fp64": max(rel) if rel else None,
                               "rel_err_gate": 1e-12,
                               "how": "sampled inside the step kernel (hydro_step_wrench_tiled_ke, last step of a graph replay), "
                                      "all_reduce(async_op=True) + pinned copy on a side stream; checked against a float64 "
                                      "host sum over all bodies of the scene"},
            "shards_bit_identical": identical,
            "shards_checked": "blake2b digests of every rank's (n_shard, 6) fp32 wrench == the same rows of the unsharded 262 144-body "
                              "scene stepped once on rank 0 (untimed)",
            "mode": f"hipGraph x{GRAPH_STEPS} steps per replay + eager remainder",
            **residency(sc.n, "f32", 2)}
    if progress is not None:
        progress["main"] = dict(result)                     # (the watchdog of guarded_strong_leg prints this much if the variant below hangs)
    # ---- the same leg with the sample's pipeline INSIDE the step graph ----
    try:
        result["graph_resident_sampling"] = strong_leg_graph_resident(reps, full, sc, dev, stream, steps, warmup, ke_every, GRAPH_STEPS, collectives, host)
    except Exception as e:                                  # noqa: BLE001 - a variant: it never costs the leg above its result
        result["graph_resident_sampling"] = {"error": repr(e)}
    for r in reps:
        r.engine.close()
    return result


STRONG_LEG_TIMEOUT_S = 240.0


def guarded_strong_leg(rank: int, world: int, dev, stream, args, multi: bool, headline: dict | None, json_fd: int):
    """c4_strong_leg, with the headline protected from it.  The leg is the one part of this file that no hardware with more
    than one GPU has ever run; it sits after the headline measurement and before the JSON line.  If a rank raises in it, or a
    collective in it never returns, the scaling run must still deliver its headline: a watchdog thread (the main thread may
    be blocked inside a collective, where no Python exception or signal handler runs) lets rank 0 print the line it has,
    with `c4_strong: {"error": ...}`, and every rank leave with exit code 0 after STRONG_LEG_TIMEOUT_S (the leg itself
    takes seconds).  A rank-local exception does the same at once."""
    import threading
    done = threading.Event()
    timeout_s = float(os.environ.get("HYDRO_BENCH_STRONG_TIMEOUT", STRONG_LEG_TIMEOUT_S))

    progress: dict = {}

    def leave(why: str):
        sys.stderr.write(f"bench.py: rank {rank}: configs[3] leg: {why}\n")
        sys.stderr.flush()
        if rank == 0 and headline is not None:
            if "main" in progress:                          # the host-driven leg had finished: only the captured variant is lost
                strong = dict(progress["main"], graph_resident_sampling={"error": why})
            else:
                strong = {"error": why, "baseline_config": "configs[3]"}
            line = dict(headline, cpu_baseline=None, c4_strong=strong)
            os.write(json_fd, (json.dumps(line) + "\n").encode())
        os._exit(0)

    def on_timeout():
        if not done.is_set():
            leave(f"no result after {timeout_s:.0f} s (a rank raised or a collective did not return); the headline on this line is complete")

    timer = threading.Timer(timeout_s, on_timeout)
    timer.daemon = True
    timer.start()
    try:
        if os.environ.get("HYDRO_BENCH_STRONG_FAULT") == f"raise:{rank}":       # test hook (tests/test_bench_gpu.py)
            raise RuntimeError("injected fault")
        if os.environ.get("HYDRO_BENCH_STRONG_FAULT") == f"hang:{rank}":
            time.sleep(3600)
        if os.environ.get("HYDRO_BENCH_STRONG_FAULT") == f"hang-resident:{rank}":
            globals()["strong_leg_graph_resident"] = lambda *a, **k: time.sleep(3600)
        return c4_strong_leg(rank, world, dev, stream, args.steps, args.warmup, collectives=multi, progress=progress)
    except Exception as e:                                  # noqa: BLE001 - the other ranks may be inside a collective: leave, do not wait
        done.set()
        leave(f"{e!r} on rank {rank}; the headline on this line is complete")
    finally:
        done.set()
        timer.cancel()


def strong_leg_graph_resident(reps, full, sc, dev, stream, steps: int, warmup: int, ke_every: int, graph_steps: int, collectives: bool, host_ke):
    """The configs[3] leg once more, with every sample's pipeline CAPTURED INTO THE STEP GRAPH (KineticEnergyMonitor.capture_sample):
    the replay that ends in a sampling step also carries the RCCL all-reduce of the pair and its copy to pinned host memory, so a
    sample costs the host nothing - in the host-driven form above the host spends 30-70 us per sample between two replays, which is
    what a 20-step region of 3-7 us steps is bound by.  Two sampling graphs (ring slots 0 / 1) alternate, a plain one runs where no
    sample is due.  Needs a device-side collective (backend nccl) or no group; under gloo it is skipped.  Same region protocol
    (barrier + synchronize pairs, max over ranks), same check against the float64 host sum."""
    from silver2_isaacsim_amd.simulate import KineticEnergyMonitor
    mon = KineticEnergyMonitor(reps[0].engine, every=ke_every)
    if not mon.graph_capturable:
        return {"skipped": "the collectives of this run are on the CPU (gloo): nothing to capture"}
    G = graph_steps
    graphs, plain = [], None
    with torch.cuda.stream(stream):
        mon.warm_up(stream)
        for j in (0, 1):
            reps[(G - 1) % 2].step_sampling(mon.slot_buffer(j))          # (prepare outside the captures)
        stream.synchronize()
        for j in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                for k in range(G):
                    if k == G - 1:
                        reps[k % 2].step_sampling(mon.slot_buffer(j))
                    else:
                        reps[k % 2].step()
                mon.capture_sample(j)
            graphs.append(g)
        if ke_every > G:
            plain = torch.cuda.CUDAGraph()
            with torch.cuda.graph(plain, stream=stream, capture_error_mode="thread_local"):
                for k in range(G):
                    reps[k % 2].step()
        for g in graphs + ([plain] if plain is not None else []):
            g.replay()
        stream.synchronize()

    def run(k_steps, observe):
        done = sampled = 0
        for _ in range(k_steps // G):
            if (done + G) % ke_every == 0:
                j = sampled % 2
                mon.reserve(j)
                graphs[j].replay()
                done += G
                if observe:
                    mon.submit_captured(done, j, stream)
                sampled += 1
            else:
                plain.replay()
                done += G
        for k in range(k_steps % G):
            reps[k % 2].step()
    with torch.cuda.stream(stream):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream); ev1.record(stream)
        run(warmup, False)
        torch.cuda.synchronize(dev)
        hd.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ev0.record(stream)
        run(steps, True)
        ev1.record(stream)
        torch.cuda.synchronize(dev)
        hd.barrier()
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
    tmax = torch.tensor([wall], dtype=torch.float64, device=hd.collective_device(dev) if collectives else "cpu")
    hd.all_reduce_max_(tmax)
    wall = float(tmax.item())
    mon.collect(block=True)
    last = mon.last()
    rel = max(abs(last[1][k] - host_ke[k]) / host_ke[k] for k in range(2)) if last else None
    return {"value": full.n * steps / wall, "ms_per_step": wall * 1e3 / steps, "kernel_us_rank0": float(ev0.elapsed_time(ev1)) * 1e3 / steps,
            "samples": len(mon.samples), "sampled_at_steps": [s_ for s_, _ in mon.samples], "rel_err_vs_host_fp64": rel,
            "mode": f"hipGraph x{G} steps per replay; a sampling replay carries the all-reduce and the pinned copy of its sample",
            "is": "the same leg with the sample pipeline captured into the step graph (KineticEnergyMonitor.capture_sample): no host work per sample"}


def gather_digests(digest: list[int], dev) -> list[list[int]]:
    """Every rank's 32-byte digest, by rank (distributed.gather_rows: exact, order-independent)."""
    return [[int(x) for x in row] for row in hd.gather_rows(digest, dev, dtype=torch.int64).tolist()]


def plugin_rate(batched: bool | str = True, steps: int = 2000, view_buffers: str = "stable"):
    """Host cost of the plugin surface: the 20 prims of the main scene, each with its own HydrodynamicsBehavior on the
    in-memory host of silver2_isaacsim_amd/testing.py; one physics step = 20 callbacks -> (batched) ONE
    hydro_step_wrench_aos launch + one apply.  Wall time per physics step, GPU drained at the end."""
    from silver2_isaacsim_amd import behavior as hb
    from silver2_isaacsim_amd.testing import build_main_scene
    hb.REGISTRY.clear()
    world, host, prims, behaviors = build_main_scene(batched, view_buffers=view_buffers)
    for b in behaviors:
        b.on_play()
    for _ in range(100):
        host.step(1.0 / 60.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        host.step(1.0 / 60.0)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / steps * 1e6
    for b in behaviors:
        b.on_stop()
    hb.REGISTRY.clear()
    label = {"stable": " (the same device tensors every step: the BEST case - the launch is prepared once)",
             "fresh": " (new tensors every step: the launch is re-prepared every step)",
             "static": " (the same device tensors every step, never refreshed, and an apply that only counts: the in-memory "
                       "simulator costs the host nothing here, what is left is the plugin's own work)"}.get(view_buffers, "")
    return {"prims": len(prims), "batched": batched, "us_per_physics_step": us, "rtf_at_60hz": 1e6 / us / 60.0,
            "apply_calls": world.apply_calls, "view_buffers": view_buffers + label,
            "host": "silver2_isaacsim_amd.testing.FakeHost (in-memory; Isaac Sim cannot run on this box)"}


def plugin_own_rate(steps: int = 4000):
    """What the PLUGIN costs the host per physics step, separated from the in-memory simulator's own stepping: the 20 prims
    of the main scene on a view that hands out the same tensors without refreshing them and whose apply only counts
    (testing.FakeRigidView buffers="static") - one group callback -> is_valid, two fetches, the key compare of the prepared
    launch, ONE hydro_step_wrench_aos through ctypes, one apply call.  Beside it: what the same loop costs with the
    kernel launch alone (the prepared callable), and with an empty Python callback (the loop itself)."""
    from silver2_isaacsim_amd import behavior as hb
    own = plugin_rate(True, steps=steps, view_buffers="static")
    out = {"prims": own["prims"], "plugin_own_us_per_step": own["us_per_physics_step"], "apply_calls": own["apply_calls"]}
    # the pieces: the prepared launch by itself, and the bare loop
    from silver2_isaacsim_amd.testing import build_main_scene
    hb.REGISTRY.clear()
    world, host, prims, behaviors = build_main_scene(True, view_buffers="static")
    for b in behaviors:
        b.on_play()
    host.step(1.0 / 60.0)
    grp = next(iter(hb.REGISTRY._groups.values()))
    launch = grp._stepper.launch
    for _ in range(100):
        launch(1.0 / 60.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        launch(1.0 / 60.0)
    torch.cuda.synchronize()
    out["prepared_launch_alone_us"] = (time.perf_counter() - t0) / steps * 1e6
    noop = lambda dt: None                                  # noqa: E731
    t0 = time.perf_counter()
    for _ in range(steps):
        noop(1.0 / 60.0)
    out["empty_python_callback_us"] = (time.perf_counter() - t0) / steps * 1e6
    out["plugin_bookkeeping_us_per_step"] = out["plugin_own_us_per_step"] - out["prepared_launch_alone_us"]
    for b in behaviors:
        b.on_stop()
    hb.REGISTRY.clear()
    out["is"] = ("host time per physics step of HydrodynamicsBehavior itself for the 20 prims of silver2_isaac_sim.usd (group "
                 "subscription, stable buffers); `plugin_20prims_us_per_step` beside it includes the in-memory simulator's five torch "
                 "launches per step.  The reference pays ~40 GPU launches per prim per step (warp_hydrodynamics_wrapper.py:85-120, "
                 "hydrodynamics_behavior.py:194-238)")
    return out


def bound_probes_leg(n: int, dev, stream):
    """{memory-only, compute-only, kernel} microseconds per launch at n bodies (scripts/probes.py: the product's own
    arithmetic on inputs that cost no HBM traffic; its traffic shape with a trivial combine; the kernel itself),
    interleaved in one process, plus the same pair for the array-of-structs entry and the kinetic-energy reduction."""
    from scripts import probes
    r = probes.bound_probes(n, dev, stream, rounds=3, reps=120 if n <= 1048576 else 40)
    us = r["us"]
    pick = lambda key: next(v for k, v in us.items() if key in k)           # noqa: E731
    out = {"n": n,
           "memory_only_us": pick("product pattern, write-through"), "memory_only_nt_stores_us": pick("product pattern (4-byte"),
           "compute_only_us": pick("lane-generated"),
           "compute_l2_resident_inputs_us": pick("L2-resident"), "kernel_us": pick("hydro_step_wrench_tiled"),
           "aos_memory_only_us": pick("AoS traffic, one row per lane"), "aos_memory_only_chunked_us": pick("AoS traffic, 16-byte"),
           "aos_kernel_us": pick("hydro_step_wrench_aos"),
           "ke_memory_only_us": pick("KE reads"), "ke_kernel_us": pick("hydro_kinetic_energy_tiled"),
           "how": "scripts/probes.py bound_probes: medians of 3 interleaved rounds, rotating buffer sets as the headline"}
    out["kernel_over_memory_only"] = out["kernel_us"] / out["memory_only_us"]
    out["compute_only_over_kernel"] = out["compute_only_us"] / out["kernel_us"]
    out["binding"] = "hbm" if out["memory_only_us"] >= out["compute_only_us"] else "valu"
    out["aos_kernel_over_memory_only"] = out["aos_kernel_us"] / out["aos_memory_only_us"]
    out["aos_frac"] = n * 168 / (out["aos_kernel_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
    out["ke_frac"] = n * 56 / (out["ke_kernel_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
    return out


def clock_probes_leg(n: int, dev, stream):
    """The shader clock this box holds under the kernel's whole body, under its memory traffic alone and under its
    arithmetic alone (scripts/probes.py clock_probes: s_memtime / s_memrealtime stamped by every wave, after 1 s of
    back-to-back launches of each kind).  The wrench kernels are co-limited at the combined-load clock; boxes differ in
    how far they throttle there, and that - not the code - is the spread of `ms_per_step` between runs."""
    from scripts import probes
    c = probes.clock_probes(n, dev, stream, seconds=1.0)
    return {"n": n, "whole_body_ghz": c["whole_body"]["ghz"], "memory_only_ghz": c["memory_only"]["ghz"],
            "compute_only_ghz": c["compute_only"]["ghz"],
            "sustained_arithmetic_ghz": c["sustained_arithmetic"]["ghz"],     # 64 passes of the body per wave: the resident loop's load
            "wave_lifetime_us": {k: v["wave_lifetime_us"] for k, v in c.items()},
            "how": "in-kernel: d(s_memtime) / d(s_memrealtime) x 100 MHz, median over the waves of 50 launches"}


def plugin_c3_rate(steps: int = 2000):
    """BASELINE config 3 through the PLUGIN surface: 19 456 prims (1 024 SILVER2 robots x 19 links), one
    HydrodynamicsBehavior instance each, scene mode (ONE physics-step subscription for the group, one
    hydro_step_wrench_aos launch, one apply).  Wall time per physics step on the in-memory host, GPU drained at the end."""
    from silver2_isaacsim_amd import behavior as hb
    from silver2_isaacsim_amd.testing import build_c3_scene
    hb.REGISTRY.clear()
    world, host, prims, behaviors, sc = build_c3_scene(1024)
    for b in behaviors:
        b.on_play()
    for _ in range(200):
        host.step(sc.dt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        host.step(sc.dt)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / steps * 1e6
    subs, fired = len(host._subs), host.callbacks_fired
    for b in behaviors:
        b.on_stop()
    hb.REGISTRY.clear()
    return {"prims": len(prims), "mode": "scene (one subscription per group)", "us_per_physics_step": us,
            "rtf_at_120hz": 1e6 / us / 120.0, "body_steps_per_s": len(prims) / (us * 1e-6),
            "physics_step_subscriptions": subs, "callbacks_per_step": fired / (steps + 200), "apply_calls": world.apply_calls,
            "host": "silver2_isaacsim_amd.testing.FakeHost, stable-buffer views (in-memory; Isaac Sim cannot run on this box)"}


def measure_traffic_live(timeout_s: float = 150.0):
    """HBM bytes per launch of the headline kernel measured NOW: two child runs of this script under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` (separate passes, as MI355X_MICROARCH.md prescribes;
    FETCH_SIZE is doubled per its gfx950 note; counters are in KB), median over the wrench kernel's dispatches.
    Returns (dict, None) or (None, reason) - the committed figure is used then and the reason goes on the line."""
    import csv
    import glob
    import shutil
    import signal
    import statistics
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    # this run is itself being profiled (rocprofv3 -- python bench.py): do not nest profilers
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "this run is itself under a profiler"
    short = [sys.executable, os.path.abspath(__file__), "--steps", "40", "--warmup", "8", "--spinup-seconds", "0.2",
             "--cpu-seconds", "0", "--no-extras", "--no-roofline-4m", "--no-live-traffic"]
    out = {}
    # the children are plain single-process runs: nothing of a process group or of the rehearsal knobs may leak into them
    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "HYDRO_BENCH_FORCE_GROUP", "HYDRO_DIST_ALWAYS",
            "HYDRO_BENCH_SHARE_GPU", "HYDRO_DIST_BACKEND")
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env["TMPDIR"] = "/tmp"
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix=f"hydro_pmc_{counter}_", dir="/tmp")
        proc = None
        try:
            # own session: on a timeout the WHOLE group goes (rocprofv3 and the bench.py under it, which holds the GPU)
            proc = subprocess.Popen([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + short,
                                    cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
            try:
                _, err = proc.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)
                proc.communicate()
                return None, f"{counter} pass timed out after {timeout_s:.0f} s (process group killed)"
            vals = []
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(path, newline="") as f:
                    for r in csv.DictReader(f):
                        if "wrench_tiled_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                            vals.append(float(r["Counter_Value"]))
            if proc.returncode != 0:
                return None, f"{counter} pass exited with {proc.returncode}: {(err or '').strip()[-300:]}"
            if len(vals) < 8:
                return None, f"{counter} pass: only {len(vals)} dispatches of the wrench kernel in the counter file"
            out[counter] = statistics.median(vals) * 1024.0
        except Exception as e:                              # noqa: BLE001 - never lose the headline over the profiler
            if proc is not None and proc.poll() is None:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                    proc.communicate()
                except Exception:                           # noqa: BLE001
                    pass
            return None, f"{counter} pass: {e!r}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"hbm_bytes_per_launch": 2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"], "fetch_size_bytes_raw": out["FETCH_SIZE"],
            "write_size_bytes": out["WRITE_SIZE"],
            "source": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE on two child runs of "
                      "bench.py (40 timed steps each), median over the wrench kernel's dispatches; FETCH_SIZE x2 (gfx950)"}, None


def load_traffic(workload: str):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (collected
    separately; FETCH_SIZE doubled per the gfx950 correction).  None when not measured."""
    path = os.path.join(REPO, "profiles", "traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f).get(workload)
        return rec
    except Exception:
        return None


def self_launch(n: int) -> int:
    """Run this script as `n` ranks of one node (one process per GPU, RCCL between them) and relay rank 0's JSON line.
    Returns the exit code for the parent.  Nothing here initialises the GPU: `torch.cuda.device_count()` only counts."""
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
    if res.returncode != 0 or line is None:
        sys.stderr.write(res.stdout)
        sys.stderr.write(f"bench.py: the {n}-rank run failed (exit code {res.returncode}, JSON line {'found' if line else 'missing'})\n")
        return res.returncode or 3
    if json.loads(line).get("n_gpus") != n:
        sys.stderr.write(f"bench.py: the child reported n_gpus={json.loads(line).get('n_gpus')}, expected {n}\n")
        return 4
    sys.stdout.write(line + "\n")
    sys.stdout.flush()
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
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-roofline-4m", action="store_true", help="skip the 4 194 304-body second roofline object (N=1 default run)")
    ap.add_argument("--no-strong-leg", action="store_true", help="N>1: skip the configs[3] strong-scaling leg")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 child runs (N=1 default workload); use profiles/traffic.json")
    ap.add_argument("--extras-budget-seconds", type=float, default=200.0,
                    help="secondary measurements are skipped once this much time has gone into them")
    ap.add_argument("--bodies-per-lane", type=int, default=0)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every GPU gets the workload's bodies (default); strong: the workload's bodies are "
                         "block-partitioned over the GPUs (BASELINE config 4: 262 144 bodies over 8 GPUs)")
    ap.add_argument("--spinup-seconds", type=float, default=1.0,
                    help="untimed run of the step loop before the W warm-up steps (GPU clock ramp)")
    ap.add_argument("--layout", default="tiled", choices=["tiled", "soa", "aos"],
                    help="tiled = engine-native tiled SoA (hydro_step_wrench_tiled); soa = plain field pointers; "
                         "aos = the simulator's (N,3)/(N,4)/(N,6) tensors (hydro_step_wrench_aos, 168 B per body-step)")
    args = ap.parse_args()

    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) <= 1 and os.environ.get("HYDRO_BENCH_FORCE_GROUP") != "1":
        # `python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves (a child
        # `python -m torch.distributed.run`), BEFORE anything in this process touches the GPU, and relay rank 0's JSON
        # line and the exit code.  Never fall through to a one-GPU run that would report n_gpus = 1.
        raise SystemExit(self_launch(args.gpus))

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner
    # on stdout when the first communicator is created), so file descriptor 1 is pointed at stderr for the whole
    # run and the line goes to a saved copy of the real stdout at the end.
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
    hd.init_process_group(force=force_group, node_barrier=True)       # (a measurement: its ranks may spin on a core for the microseconds a timed region opens in)
    multi = world > 1 or force_group
    # one rank per GPU; HYDRO_BENCH_SHARE_GPU=1 (with HYDRO_DIST_BACKEND=gloo) lets several ranks share
    # GPU 0 to rehearse the multi-rank path on a single-GPU box
    ndev = torch.cuda.device_count()
    share = os.environ.get("HYDRO_BENCH_SHARE_GPU") == "1"
    if world > 1 and not share and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: local_rank {local_rank} but only {ndev} GPU(s) visible")
    dev = torch.device("cuda", (local_rank % ndev) if world > 1 else 0)
    torch.cuda.set_device(dev)
    # The ranks that REALLY take part in a collective of the live group (a 1 from each, summed - by RCCL under backend nccl):
    # the number of GPUs on the line is this one, and a run that was asked for --gpus N but joins fewer is refused.
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
    cls = AosReplica if args.layout == "aos" else Replica
    if args.layout == "aos" and args.scenes * (args.bodies or n_default) * 52 < (410 << 20):
        args.scenes = max(args.scenes, -(-(410 << 20) // ((args.bodies or n_default) * 52)))     # (see aos_rate: no resident rows)
    replicas = [cls(sc, coeff, dev, roll=r * 131071, layout=args.layout) for r in range(args.scenes)]
    if args.bodies_per_lane:
        for r in replicas:
            r.engine.set_tuning(args.bodies_per_lane)
    stream = torch.cuda.Stream(dev)

    spin_up(replicas, stream, args.spinup_seconds)
    wall, ev_ms = timed_steps(replicas, args.steps, args.warmup, stream, multi)
    # bodies on all ranks (shards differ by at most one body under strong scaling)
    n_all = torch.tensor([float(sc.n)], dtype=torch.float64, device=hd.collective_device(dev) if multi else "cpu")
    hd.all_reduce_sum_(n_all)
    body_steps = float(n_all.item()) * args.steps
    value = body_steps / wall
    kernel_us = ev_ms * 1e3 / args.steps              # HIP events on the launch stream around the K timed steps
    # every rank's own figures, by rank (N > 1: which GPU set the max-over-ranks time, and how far apart the boxes' GPUs are)
    per_rank = hd.gather_rows([timed_steps.last_local_wall * 1e6 / args.steps, kernel_us], dev) if multi else None
    step_us = wall * 1e6 / args.steps                 # the interval `value` and `ms_per_step` are computed from
    bpb = BYTES_PER_BODY[coeff] + (24 if args.layout == "aos" else 0)        # the AoS entry also updates the engine's previous velocity
    # ONE clock for `value` and `roofline.frac`: algorithmic bytes per launch / (timed interval / K).  The event figure
    # of the same K steps (always a little shorter: it leaves out the host's synchronisation at both ends) is kept as
    # `frac_contract_steps`, the median of 5 x 200 steps as `frac_median_of_5`.
    achieved = sc.n * bpb / (step_us * 1e-6) / 1e9
    achieved_events = sc.n * bpb / (kernel_us * 1e-6) / 1e9

    # the one collective of the path: global kinetic energy (every rank reduces its shard on device)
    with torch.cuda.stream(stream):
        ke = replicas[0].kinetic_energy()
    stream.synchronize()
    ke = ke.to(hd.collective_device(dev))
    t0 = time.perf_counter()
    hd.global_kinetic_energy(ke)
    torch.cuda.synchronize(dev)
    ke_us = (time.perf_counter() - t0) * 1e6
    # ... checked against float64 host sums: every rank sums its OWN scene on the host, the per-rank sums are gathered
    # exactly (distributed.gather_rows) and added with fsum - nothing of the reference value went through the all-reduce
    host_ke = hd.gather_rows(scenes.kinetic_energy_fp64(sc.state, sc.params, rotational=True), dev)
    host_ke = [math.fsum(host_ke[:, k].tolist()) for k in range(2)]
    ke_rel_err = max(abs(float(ke[k]) - host_ke[k]) / host_ke[k] for k in range(2))

    out = None
    if rank == 0:
        traffic = load_traffic(f"{args.workload}:{args.layout}") if world == 1 and not args.bodies else None
        out = {
            "metric": "body-steps/sec", "value": value, "unit": "body-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
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
                         "clock": "the timed interval of `value` (wall time between the barrier + synchronize pairs, max over "
                                  "ranks) / steps; `kernel_us` / `frac_contract_steps` = HIP events around the same steps",
                         "step_us": step_us, "kernel_us": kernel_us,
                         "achieved_contract_steps": achieved_events, "frac_contract_steps": achieved_events / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": sc.n * bpb,
                         "frac_of_measured_copy_ceiling": achieved / HBM_COPY_CEILING_GBS,
                         # what the counters say: the kernel moves 122 B per body (fp16 coefficients), not the 130
                         # algorithmic ones - frac_traffic is the honest bandwidth fraction
                         "traffic_bytes_per_body": bpb if args.layout == "aos" else TRAFFIC_BYTES_PER_BODY[coeff],
                         "frac_traffic": sc.n * (bpb if args.layout == "aos" else TRAFFIC_BYTES_PER_BODY[coeff]) / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "traffic_measured": "rocprofv3 --pmc passes committed under profiles/ (not re-measured in this run)",
                         **residency(sc.n, coeff, args.scenes)},
            "spinup_seconds": args.spinup_seconds,
            "collectives": (f"{'nccl (RCCL)' if hd.collective_device(dev).type == 'cuda' else 'gloo'}, {world} rank(s)" if multi else "none (single process)"),
            "barrier": hd.barrier_kind(),
            "rccl_ranks": live_ranks if multi and hd.collective_device(dev).type == "cuda" else 0,
            "collective_ranks": live_ranks,
            "global_kinetic_energy_J": [float(x) for x in ke.cpu().tolist()],
            "global_kinetic_energy_host_fp64_J": host_ke,
            "global_kinetic_energy_rel_err_vs_host_fp64": ke_rel_err,
            "ke_allreduce_us": ke_us,
        }
        if per_rank is not None:
            out["per_rank"] = {"step_us": [float(x) for x in per_rank[:, 0]], "kernel_us": [float(x) for x in per_rank[:, 1]],
                               "is": "each rank's own wall interval / steps (up to the point where ITS steps are done, before the closing barrier) and HIP-event "
                                     "time / steps, by rank: `ms_per_step` is the slowest rank's plus the barrier; the spread is the spread of the node's "
                                     "GPUs (DVFS, DESIGN.md section 6), not of the software"}

    # N > 1: BASELINE configs[3] as stated (262 144 bodies over the N GPUs, strong scaling) on every rank.  The headline is
    # complete at this point: whatever happens to this secondary leg on hardware it has never met, rank 0 still prints it.
    strong = None
    if multi and not args.no_strong_leg:
        strong = guarded_strong_leg(rank, world, dev, stream, args, multi, out, json_fd)

    if rank == 0:
        if traffic:
            out["roofline"]["traffic_source"] = traffic.get("source")
        if traffic and not args.no_live_traffic and args.layout == "tiled" and world > 1:
            out["roofline"]["traffic_live_skipped"] = "multi-rank run: the counters are read in the N = 1 run (the committed passes are used here)"
        elif traffic and not args.no_live_traffic and args.layout == "tiled":
            live, why_not = measure_traffic_live()
            if live is None:
                out["roofline"]["traffic_live_skipped"] = why_not
            else:
                out["roofline"]["traffic_committed"] = traffic["hbm_bytes_per_launch"]
                out["roofline"]["traffic"] = live["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = live["source"]
                out["roofline"]["traffic_measured"] = "in this run"
                out["roofline"]["traffic_bytes_per_body_measured"] = live["hbm_bytes_per_launch"] / sc.n
                out["roofline"]["frac_traffic"] = live["hbm_bytes_per_launch"] / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS
        if world == 1 and args.cpu_seconds > 0:
            try:
                out["cpu_baseline"] = cpu_baseline_leg(sc, _last_stepped(replicas, args.steps), args.cpu_seconds)
            except Exception as e:                          # noqa: BLE001 - report, never lose the line
                out["cpu_baseline"] = {"value": None, "unit": "body-steps/s", "cores": 1, "kind": "port",
                                       "sample": "failed", "error": repr(e)}
        else:
            out["cpu_baseline"] = None
        if strong is not None:
            out["c4_strong"] = strong
        if world == 1 and args.workload == "c5" and not args.bodies and not args.no_roofline_4m:
            try:
                out["roofline_4m"] = roofline_4m(dev, stream)
            except Exception as e:                          # noqa: BLE001 - report, never lose the line
                out["roofline_4m"] = {"error": repr(e)}
        if world == 1 and not args.no_extras:
            # median of 5 runs (each 200 steps after 20 warm-up steps), same replicas
            try:
                runs = []
                for _ in range(5):
                    _, ms5 = timed_steps(replicas, 200, 20, stream)
                    runs.append(ms5 * 1e3 / 200)
                runs.sort()
                out["roofline"]["kernel_us_5x200_runs"] = runs
                out["roofline"]["kernel_us_median_of_5"] = runs[2]
                out["roofline"]["frac_median_of_5"] = sc.n * bpb / (runs[2] * 1e-6) / 1e9 / HBM_PEAK_GBS
            except Exception as e:                          # noqa: BLE001
                out["roofline"]["kernel_us_5x200_runs"] = repr(e)
            for r in replicas:
                r.engine.close()
            replicas = []
            torch.cuda.empty_cache()
            ex = {}

            t_extras = time.perf_counter()

            def guarded(key, fn, *fa, **fk):
                if time.perf_counter() - t_extras > args.extras_budget_seconds:
                    ex[key] = {"skipped": "extras time budget"}
                    return
                try:
                    ex[key] = fn(*fa, **fk)
                except Exception as e:                      # noqa: BLE001 - extras never break the headline
                    ex[key] = {"error": repr(e)}
            # which bound binds: memory-only / compute-only probes beside the kernels themselves (verdict r2 item 2)
            guarded("bound_probes_1m", bound_probes_leg, 1048576, dev, stream)
            guarded("bound_probes_4m", bound_probes_leg, 4194304, dev, stream)
            guarded("clocks_1m", clock_probes_leg, 1048576, dev, stream)
            guarded("c2_4096", quick_rate, "c2", 4096, "f32", dev, stream, steps=200)
            guarded("c3_19456", quick_rate, "c3", 19456, "f32", dev, stream, steps=200)
            guarded("c4_shard_32768", quick_rate, "c4", 32768, "f32", dev, stream, steps=200)
            guarded("c2_4096_graph", graph_rate, "c2", 4096, "f32", dev, stream)
            guarded("c3_19456_graph", graph_rate, "c3", 19456, "f32", dev, stream)
            guarded("c4_shard_32768_graph", graph_rate, "c4", 32768, "f32", dev, stream)
            guarded("c4_262144", quick_rate, "c4", 262144, "f32", dev, stream, steps=100)
            guarded("c5_f32_1048576", quick_rate, "c4", 1048576, "f32", dev, stream, steps=100)
            guarded("f32_4194304", quick_rate, "c4", 4194304, "f32", dev, stream, steps=50, sets=2)
            guarded("f16_4194304", quick_rate, "c5", 4194304, "f16", dev, stream, steps=50, sets=2)
            guarded("batch_4x_c5_1048576", batch_rate, "c5", 1048576, "f16", dev, stream)
            guarded("two_streams_c5_1048576", two_stream_rate, "c5", 1048576, "f16", dev)
            guarded("two_streams_f16_4194304", two_stream_rate, "c5", 4194304, "f16", dev, steps=100, sets=2)
            guarded("plain_soa_c5_1048576", quick_rate, "c5", 1048576, "f16", dev, stream, steps=100, layout="soa")
            guarded("plain_soa_f32_4194304", quick_rate, "c4", 4194304, "f32", dev, stream, steps=50, sets=2, layout="soa")
            guarded("aos_entry_1048576", aos_rate, 1048576, dev, stream)
            guarded("plugin_20prims_us_per_step", plugin_rate, True)
            guarded("plugin_20prims_own_host_cost", plugin_own_rate)
            guarded("plugin_20prims_fresh_tensors_every_step", plugin_rate, True, steps=1000, view_buffers="fresh")
            guarded("plugin_20prims_callbacks_mode", plugin_rate, "callbacks", steps=1000)
            guarded("plugin_20prims_per_prim_mode", plugin_rate, False, steps=500)
            guarded("plugin_c3_19456prims", plugin_c3_rate)
            guarded("closed_loop_c2_4096", closed_loop_rate, "c2", 4096)
            guarded("closed_loop_c2_4096_unfused", closed_loop_rate, "c2", 4096, fused=False)
            guarded("closed_loop_c3_1024envs_implicit", closed_loop_rate, "c3", 19456, implicit_drag=True)
            guarded("closed_loop_c2_262144", closed_loop_rate, "c2", 262144, steps=1024)
            guarded("closed_loop_c2_262144_unfused", closed_loop_rate, "c2", 262144, steps=1024, fused=False)
            guarded("closed_loop_c2_1048576", closed_loop_rate, "c2", 1048576, steps=512)
            guarded("closed_loop_c2_4096_resident", closed_loop_rate, "c2", 4096, resident=True)
            guarded("closed_loop_c3_1024envs_implicit_resident", closed_loop_rate, "c3", 19456, implicit_drag=True, resident=True)
            guarded("closed_loop_c2_262144_resident", closed_loop_rate, "c2", 262144, steps=1024, resident=True)
            guarded("closed_loop_c2_1048576_resident", closed_loop_rate, "c2", 1048576, steps=2560, resident=True)
            # the compute-bound entries also get the fraction at the clock this box held under SUSTAINED arithmetic (64 passes
            # per wave: scripts/probes.py) - how much of the issue rate the loop uses at the clock it is given
            held = ex.get("clocks_1m", {}).get("sustained_arithmetic_ghz") if isinstance(ex.get("clocks_1m"), dict) else None
            for v in ex.values():
                r = v.get("roofline") if isinstance(v, dict) else None
                if held and isinstance(r, dict) and r.get("bound") == "valu-issue":
                    r["clock_held_ghz"] = held
                    r["frac_of_issue_rate_at_clock_held"] = r["frac"] * BOOST_CLOCK_GHZ / held
            out["extras"] = ex
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    # (the line is out: a rank that left early - see guarded_strong_leg - must not turn the run into a failure here)
    try:
        hd.barrier(timeout_s=float(os.environ.get("HYDRO_BENCH_TEARDOWN_TIMEOUT", "60")))
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
    except Exception as e:                                  # noqa: BLE001
        sys.stderr.write(f"bench.py: rank {rank}: teardown: {e!r}\n")


def _last_stepped(replicas, steps):
    """Replica whose output buffer holds the result of a completed step."""
    return replicas[(steps - 1) % len(replicas)] if steps > 0 else replicas[0]


if __name__ == "__main__":
    main()
