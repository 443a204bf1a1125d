#!/usr/bin/env python3
"""Secondary measurements of bench.py - everything that is NOT the headline protocol.

bench.py prints ONE compact JSON line (<= 8 192 bytes: the driver must be able to read it).  What used to ride on that
line as `extras` - bound probes, clocks, two streams, several scenes per launch, the array-of-structs entry, the plugin
surface, the closed loop with its VALU-issue rooflines - is measured here and written to a side file
(`bench.py --extras-out PATH`, default bench_extras.json beside bench.py, echoed to stderr).  `run()` is called by
bench.py on rank 0 of an N = 1 run under --extras-budget-seconds; every leg is also callable by itself
(tests/test_bench_gpu.py, scripts/run_aux.py).  What the fields mean: FIELD_NOTES below (`python bench.py --explain`)
and DESIGN.md section 6 - no explanatory text travels in the JSON.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from bench import (BYTES_PER_BODY, HBM_PEAK_GBS, REPO, Replica, build_scene, graph_rate, quick_rate, residency,      # noqa: E402
                   spin_up, timed_steps)
from silver2_isaacsim_amd.engine import HydroEngine

THROTTLE_RATIO = 1.15

FIELD_NOTES = """\
bench.py - what the fields of the JSON line and of the side file mean
=====================================================================
value, ms_per_step      whole-job body-steps/s and the wall interval per step: wall time between the barrier + synchronize
                        pairs around EXACTLY --steps launches, max over ranks.  Inputs resident in HBM before the region.
roofline.frac           algorithmic bytes per launch (130 B x bodies, fp16 coefficients; 144 with fp32) / ms_per_step / 8 TB/s.
                        ONE clock: the same interval `value` comes from.
roofline.kernel_us      HIP events on the launch stream around the same K steps / K (no host synchronisation in it);
                        frac_contract_steps is the fraction on that clock, frac_median_of_5 the median of 5 x 200 steps.
roofline.traffic        HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes:
                        traffic_measured "live" = two child runs of bench.py in this run (40 timed steps each, median over
                        the wrench kernel's dispatches), "committed" = profiles/traffic.json.  122 B per body: p_x and p_y
                        are provably unused by the wrench and never loaded (8 B under the algorithmic 130).
roofline.resident       "hbm" when the rotating working set (scene replicas x bodies x resident bytes) exceeds the 256 MiB
                        Infinity Cache, else "infinity-cache" (then a GB/s figure is a cache rate, never an HBM fraction).
roofline_4m             the same kernel on 4 194 304 bodies, two rotating replicas (1.1 GB): no cache can assist.
cpu_baseline            the C port of the reference's Numba path (oracle/hydro_oracle.c, gcc -O3 -ffast-math) on a bounded
                        sample of the bench scene: `value` 1 thread, `all_core_value` on the cores this job may use; the same
                        sample checks the GPU result (max_rel_err = wrench_error vs the oracle, n_over = bodies above 1e-5).
configs                 per BASELINE config: us per step and body-steps/s, eager (one ctypes launch per step) and `graph`
                        (64 consecutive steps replayed from one HIP graph); four rotating replicas; cache-resident sizes.
box                     which kind of box this is: memory_only_us / compute_only_us / kernel_us (scripts/probes.py, medians of
                        3 interleaved rounds) and the shader clock held under the kernel's whole body, its memory traffic
                        alone, its arithmetic alone (in-kernel s_memtime / s_memrealtime).  kernel_over_memory_only >= 1.15
                        = a box that throttles under the combined load; the spread of `frac` between boxes is this, not the code.
c4_strong (N > 1)       BASELINE configs[3] as stated: 262 144 bodies block-partitioned over the N GPUs, kinetic energy
                        sampled >= 2 times inside the region (device reduction in the step kernel, asynchronous all-reduce on a
                        side stream).  Self-checks: ke.rel_err (last global sample vs float64 host sum, gate 1e-12),
                        shards_bit_identical (blake2b of every rank's wrench == the same rows of the unsharded scene).
                        captured = the same leg with each sample's pipeline (RCCL all-reduce + pinned copy) inside the step graph.
ok                      false when the N > 1 leg raised or hung: the headline on the line is complete, the run exits 3.
side file: extras.*     bound_probes_*: probes beside the kernels (AoS entry, kinetic energy too); two_streams_*: independent
                        scenes on two streams (ramp/drain of a launch overlapped - throughput only); batch_4x_*: four scenes per
                        launch; plain_soa_*: hydro_step_wrench_ext; aos_entry_*: hydro_step_wrench_aos (168 B per body-step,
                        eight rotating sets so that no rows stay in the Infinity Cache); plugin_*: host cost per physics step of
                        the HydrodynamicsBehavior surface on testing.FakeHost (own = view that costs nothing, so what is left
                        is the plugin); closed_loop_*: wrench + integrator, RTF as benchmark_rtf.py defines it; *_resident: 64
                        steps per launch with the bodies in registers - compute-bound, so its roofline is VALU issue:
                        frac = time the step's VALU instructions (profiles/isa_mix.json) need at 2 cycles per wave64
                        instruction (4 for fp64 arithmetic), 1 024 SIMDs, 2.55 GHz boost / measured time: an upper bound <= 1;
                        model_measured_prices is a MODEL (classes priced with scripts/ubench_valu.hip's readings at 2.4 GHz).
"""

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
            **residency(sc.n, coeff, sets)}


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
    path = os.path.join(REPO, "profiles", "isa_mix.json")
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
            "unit": "G wave-instructions/s",
            "frac": floor_us / us_per_step,
            "issue_cycles_per_wave_step_at_hardware_rate": spec_cycles, "floor_us_per_step": floor_us,
            "model_measured_prices": {"issue_cycles_per_wave_step": model_cycles, "us_per_step_at_2.4GHz": model_us,
                                      "measured_over_model": us_per_step / model_us, "kind": "model, not a bound"}}


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
    return {"prims": len(prims), "batched": batched, "us_per_physics_step": us, "rtf_at_60hz": 1e6 / us / 60.0,
            "apply_calls": world.apply_calls, "view_buffers": view_buffers, "host": "testing.FakeHost"}


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
           "ke_memory_only_us": pick("KE reads"), "ke_kernel_us": pick("hydro_kinetic_energy_tiled")}
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
            "wave_lifetime_us": {k: v["wave_lifetime_us"] for k, v in c.items()}}


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
            "physics_step_subscriptions": subs, "callbacks_per_step": fired / (steps + 200), "apply_calls": world.apply_calls, "host": "testing.FakeHost"}



def measure_traffic_live(timeout_s: float = 150.0):
    """HBM bytes per launch of the headline kernel measured NOW: two child runs of this script under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` (separate passes, as MI355X_MICROARCH.md prescribes;
    FETCH_SIZE is doubled per its gfx950 note; counters are in KB), median over the wrench kernel's dispatches.
    Returns (dict, None) or (None, reason) - the committed figure is used then."""
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
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "this run is itself under a profiler"
    short = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "40", "--warmup", "8", "--spinup-seconds", "0.2",
             "--cpu-seconds", "0", "--no-extras", "--no-configs", "--no-roofline-4m", "--no-live-traffic"]
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
                return None, f"{counter} pass exited with {proc.returncode}: {(err or '').strip()[-200:]}"
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
            "write_size_bytes": out["WRITE_SIZE"]}, None


def run(dev, stream, budget_s: float) -> dict:
    """Every secondary leg, each guarded (an exception or the time budget never costs the caller anything)."""
    ex: dict = {}
    t0 = time.perf_counter()

    def guarded(key, fn, *fa, **fk):
        if time.perf_counter() - t0 > budget_s:
            ex[key] = {"skipped": "extras time budget"}
            return
        try:
            ex[key] = fn(*fa, **fk)
        except Exception as e:                              # noqa: BLE001
            ex[key] = {"error": repr(e)}
    guarded("bound_probes_1m", bound_probes_leg, 1048576, dev, stream)
    guarded("clocks_1m", clock_probes_leg, 1048576, dev, stream)
    guarded("bound_probes_4m", bound_probes_leg, 4194304, dev, stream)
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
    # the compute-bound entries also get the fraction at the clock this box held under SUSTAINED arithmetic
    held = ex.get("clocks_1m", {}).get("sustained_arithmetic_ghz") if isinstance(ex.get("clocks_1m"), dict) else None
    for v in ex.values():
        r = v.get("roofline") if isinstance(v, dict) else None
        if held and isinstance(r, dict) and r.get("bound") == "valu-issue":
            r["clock_held_ghz"] = held
            r["frac_of_issue_rate_at_clock_held"] = r["frac"] * BOOST_CLOCK_GHZ / held
    ex["seconds"] = time.perf_counter() - t0
    return ex


def box_summary(ex: dict) -> dict | None:
    """The few numbers of the probes that go on bench.py's compact line (`box`): which bound binds on THIS box and the
    clock it holds under the whole-body kernel - so that a low `roofline.frac` on a throttling box explains itself."""
    p, c = ex.get("bound_probes_1m"), ex.get("clocks_1m")
    if not (isinstance(p, dict) and "kernel_us" in p):
        return None
    out = {k: p[k] for k in ("memory_only_us", "compute_only_us", "kernel_us", "kernel_over_memory_only", "binding")}
    if isinstance(c, dict) and "whole_body_ghz" in c:
        out.update(clock_held_ghz=c["whole_body_ghz"], memory_only_ghz=c["memory_only_ghz"], compute_only_ghz=c["compute_only_ghz"])
    # ordinary boxes read 1.03-1.06, boxes that throttle under the combined load 1.20-1.32 (profiles/README.md): the line between them
    out["throttles_under_combined_load"] = bool(p["kernel_over_memory_only"] >= THROTTLE_RATIO)
    return out
