#!/usr/bin/env python3
"""The N > 1 leg of bench.py: BASELINE.json configs[3] as stated - 262 144 bodies block-partitioned over the N GPUs
(strong scaling), the global kinetic energy sampled at least twice inside the timed region through
`simulate.KineticEnergyMonitor` - and the guard that keeps the headline, and the exit code, honest whatever this leg does
on hardware it has never met (`LegGuard`).  bench.py imports this module only when WORLD_SIZE > 1 (or a forced one-rank
group); tests/test_bench_gpu.py rehearses it with 2 and 4 ranks on one GPU, tests/test_rccl_single_rank_gpu.py with a
one-rank RCCL communicator.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

from bench import EXIT_LEG_FAILED, Replica, build_scene, render_line, residency, spin_up, write_all      # noqa: E402
from silver2_isaacsim_amd import distributed as hd
from silver2_isaacsim_amd import scenes

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


def gather_digests(digest: list[int], dev) -> list[list[int]]:
    """Every rank's 32-byte digest, by rank (distributed.gather_rows: exact, order-independent)."""
    return [[int(x) for x in row] for row in hd.gather_rows(digest, dev, dtype=torch.int64).tolist()]


def _region(run, steps, warmup, dev, stream, collectives):
    """The timed-region protocol of the headline around run(k_steps, observe): (wall seconds max over ranks, event ms)."""
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
    return float(tmax.item()), float(ev0.elapsed_time(ev1))


def c4_strong_leg(rank: int, world: int, dev, stream, steps: int, warmup: int, collectives: bool = True, progress: dict | None = None):
    """BASELINE.json configs[3] as it is stated: 262 144 bodies (seed 4) block-partitioned over the GPUs, every rank
    steps its contiguous shard (no data-path collective); the global kinetic energy is sampled every `ke_every`
    steps - at least twice inside the timed region - by simulate.KineticEnergyMonitor: device reduction inside the step
    kernel, asynchronous all-reduce (RCCL under backend nccl) on a side stream, picked up later by the host.  A shard of
    32 768 bodies is one 2.7 us launch, so `graph_steps` consecutive steps are one HIP graph (the K timed steps are
    K // graph_steps replays plus an eager remainder); a replay at whose end a sample is due is of the graph whose LAST step
    is the kernel variant that also samples.  After the region the leg PROVES itself: the last global sample against an fp64
    host sum over all 262 144 bodies, and every rank's shard wrench against the unsharded scene stepped once on rank 0."""
    from silver2_isaacsim_amd.simulate import KineticEnergyMonitor
    full = build_scene("c4", 262144, 4)                      # the same scene on every rank ...
    sc = full.shard(rank, world)                             # ... each keeps its contiguous block
    reps = [Replica(sc, "f32", dev, roll=0) for _ in range(2)]      # two buffer sets of the SAME shard (cache-resident sizes)
    ke_every, G = strong_leg_cadence(steps)
    mon = KineticEnergyMonitor(reps[0].engine, every=ke_every)
    ke_dev = torch.zeros(2, dtype=torch.float64, device=dev)         # where the sampling step leaves the shard's pair
    with torch.cuda.stream(stream):
        mon.warm_up(stream)                                           # (its first pass costs 0.4 ms of one-time set-up: not in the region)
    spin_up(reps, stream, 0.3)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        reps[1].step_sampling(ke_dev)                                 # (prepare outside the capture)
        reps[0].step_sampling(ke_dev)
        stream.synchronize()
        with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
            for k in range(G):
                if k == G - 1:
                    reps[k % 2].step_sampling(ke_dev)
                else:
                    reps[k % 2].step()
        g.replay()
        g_plain = None
        if ke_every > G:                                        # replays at whose end no sample is due: plain steps only
            g_plain = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_plain, stream=stream, capture_error_mode="thread_local"):
                for k in range(G):
                    reps[k % 2].step()
            g_plain.replay()
        stream.synchronize()

    def run(k_steps, observe):
        done = 0
        for _ in range(k_steps // G):
            due = (done + G) % ke_every == 0
            (g if due or g_plain is None else g_plain).replay()
            done += G
            if due and observe:
                mon.observe(done, stream=stream, sampled=ke_dev)
        for k in range(k_steps % G):
            done += 1
            if observe and done % ke_every == 0:
                reps[k % 2].step_sampling(ke_dev)
                mon.observe(done, stream=stream, sampled=ke_dev)
            else:
                reps[k % 2].step()
    wall, ev_ms = _region(run, steps, warmup, dev, stream, collectives)
    mon.collect(block=True, timeout_s=60.0)
    del g, g_plain
    last = mon.last()
    # ---- untimed: the leg checks itself ----
    host = scenes.kinetic_energy_fp64(full.state, full.params, rotational=True)
    rel = [abs(last[1][k] - host[k]) / host[k] for k in range(2)] if last else None
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
              "ms_per_step": wall * 1e3 / steps, "kernel_us_rank0": ev_ms * 1e3 / steps, "graph_steps": G,
              "ke": {"every_steps": ke_every, "samples": len(mon.samples), "host_waits": mon.waited_on_host,
                     "sampled_at_steps": [s for s, _ in mon.samples][-8:],
                     "last_step": last[0] if last else None, "global_J": last[1] if last else None,
                     "host_fp64_J": list(host), "rel_err": max(rel) if rel else None, "rel_err_gate": 1e-12},
              "shards_bit_identical": identical, "resident": residency(sc.n, "f32", 2)["resident"]}
    if progress is not None:
        progress["main"] = dict(result)                     # (the watchdog of guarded_strong_leg prints this much if the variant below hangs)
    try:                                                    # the same leg with the sample's pipeline INSIDE the step graph
        result["captured"] = strong_leg_graph_resident(reps, full, sc, dev, stream, steps, warmup, ke_every, G, collectives, host)
    except Exception as e:                                  # noqa: BLE001 - a variant: it never costs the leg above its result
        result["captured"] = {"error": repr(e)[:200]}
    for r in reps:
        r.engine.close()
    return result


def strong_leg_graph_resident(reps, full, sc, dev, stream, steps: int, warmup: int, ke_every: int, G: int, collectives: bool, host_ke):
    """The configs[3] leg once more, with every sample's pipeline CAPTURED INTO THE STEP GRAPH
    (KineticEnergyMonitor.capture_sample): the replay that ends in a sampling step also carries the RCCL all-reduce of the
    pair and its copy to pinned host memory, so a sample costs the host nothing.  Two sampling graphs (ring slots 0 / 1)
    alternate, a plain one runs where no sample is due.  Needs a device-side collective (backend nccl) or no group; under
    gloo it is skipped.  Same region protocol, same check against the float64 host sum."""
    from silver2_isaacsim_amd.simulate import KineticEnergyMonitor
    mon = KineticEnergyMonitor(reps[0].engine, every=ke_every)
    if not mon.graph_capturable:
        return {"skipped": "gloo: a CPU collective cannot live in a HIP graph"}
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
    wall, ev_ms = _region(run, steps, warmup, dev, stream, collectives)
    mon.collect(block=True, timeout_s=60.0)
    last = mon.last()
    rel = max(abs(last[1][k] - host_ke[k]) / host_ke[k] for k in range(2)) if last else None
    return {"value": full.n * steps / wall, "ms_per_step": wall * 1e3 / steps, "kernel_us_rank0": ev_ms * 1e3 / steps,
            "samples": len(mon.samples), "sampled_at_steps": [s_ for s_, _ in mon.samples][-8:], "rel_err": rel}


STRONG_LEG_TIMEOUT_S = 240.0


class LegGuard:
    """Keeps the headline safe from the N > 1 leg, and the exit code honest.  The leg is the one part of the bench no hardware
    with more than one GPU has run; it sits after the headline measurement.  If a rank raises in it, or a collective in it (or
    the teardown barrier after it) never returns, rank 0 still prints the line it has - with "ok": false and `c4_strong.error` -
    and THEN every rank leaves with exit code 3 (os._exit: the main thread may be blocked inside a collective, where no Python
    exception or signal handler runs; a process that holds the GPU is never re-exec'ed, nothing is retried).
    The ORDER matters under torchrun, which tears the job down as soon as one rank has failed: a failing rank k > 0 first drops
    a marker file and waits (bounded) until rank 0 has its line out; every rank's watchdog thread polls for markers four times
    a second, so rank 0 reacts at once.  Marker files (temp dir, name drawn by rank 0 and broadcast before the leg):
    <base>.fault.<k> = rank k is leaving, with the reason; <base>.out = rank 0's line is on stdout (a complete one or a
    failure one); <base>.failed = rank 0 has left with a failure.  The guard stays armed until the teardown barrier is behind
    every rank (finish()): a rank that waits there for one that hangs must not leave before rank 0 has printed."""

    def __init__(self, rank: int, headline: dict | None, json_fd: int, timeout_s: float):
        import tempfile
        import threading
        self.rank, self.headline, self.fd, self.timeout_s = rank, headline, json_fd, timeout_s
        name = [f"hydro_bench_{os.getuid()}_{os.urandom(8).hex()}" if rank == 0 else None]
        if hd._collectives_on():                            # (a collective - but the headline's own collectives have just worked)
            torch.distributed.broadcast_object_list(name, src=0)
        self.base = os.path.join(tempfile.gettempdir(), name[0] or f"hydro_bench_{os.getuid()}_{os.getpid()}")
        self.progress: dict = {}
        self.extra: dict = {}                               # fields rank 0 already has for the line (cpu_baseline)
        self._done = False
        self._main_seen = False
        self._deadline = time.monotonic() + timeout_s
        self._lock = threading.Lock()
        self._thread = threading.Thread(target=self._watch, daemon=True)
        self._thread.start()

    def _touch(self, suffix: str, text: str = "") -> None:
        try:
            with open(self.base + suffix, "w") as f:
                f.write(text)
        except OSError:
            pass

    def _watch(self):
        import glob
        while not self._done:
            if not self._main_seen and "main" in self.progress:
                # the host-driven leg is done and safe in `progress`: what still runs is the captured-sampling VARIANT, which
                # takes a second or two - it gets 90 s, not the rest of the leg's allowance, before the line goes out without it
                self._main_seen = True
                self._deadline = min(self._deadline, time.monotonic() + 90.0)
            if self.rank != 0 and os.path.exists(self.base + ".failed"):
                os._exit(EXIT_LEG_FAILED)                   # rank 0 has printed its failure line and left
            hits = sorted(glob.glob(self.base + ".fault.*"))
            if hits:
                try:
                    why = open(hits[0]).read().strip() or "a rank failed"
                except OSError:
                    why = "a rank failed"
                self.leave(why, mark=False)
            if time.monotonic() > self._deadline:
                self.leave(f"no result after {self.timeout_s:.0f} s (a rank raised or a collective did not return); the headline on this line is complete")
            time.sleep(0.25)

    def leg_done(self, allowance_s: float = 180.0) -> None:
        """The leg returned on this rank: what is left is rank 0's line and the teardown barrier - a fresh deadline for those."""
        self._deadline = time.monotonic() + allowance_s

    def line_is_out(self) -> None:
        """Rank 0 has written its (complete) line: nothing is ever printed again, and nobody needs to wait for it."""
        self.headline = None
        self._touch(".out")

    def finish(self) -> None:
        """Every rank is through the teardown barrier: disarm; rank 0 removes the markers."""
        self._done = True
        if self.rank == 0:
            for suffix in (".out",):
                try:
                    os.unlink(self.base + suffix)
                except OSError:
                    pass

    def leave(self, why: str, mark: bool = True):
        with self._lock:                                    # (watchdog thread and main thread: only one of them leaves)
            if self._done:
                return
            sys.stderr.write(f"bench.py: rank {self.rank}: configs[3] leg: {why}\n")
            sys.stderr.flush()
            if self.rank == 0:
                if self.headline is not None:               # (None: the complete line is out already - the failure is the exit code)
                    if "main" in self.progress:             # the host-driven leg had finished: only the captured variant is lost
                        strong = dict(self.progress["main"], captured={"error": why})
                    else:
                        strong = {"error": why, "baseline_config": "configs[3]"}
                    line = dict(self.headline, cpu_baseline=None)
                    line.update(self.extra, c4_strong=strong, ok=False)
                    write_all(self.fd, (render_line(line) + "\n").encode())
                self._touch(".out")
                self._touch(".failed")
            else:
                if mark:
                    self._touch(f".fault.{self.rank}", why)
                t_end = time.monotonic() + 20.0             # rank 0 prints first (torchrun ends the job at the first failed rank)
                while not os.path.exists(self.base + ".out") and time.monotonic() < t_end:
                    time.sleep(0.05)
            os._exit(EXIT_LEG_FAILED)


def guarded_strong_leg(rank: int, world: int, dev, stream, args, multi: bool, guard: LegGuard):
    """c4_strong_leg under a LegGuard (fault injection for tests/test_bench_gpu.py: HYDRO_BENCH_STRONG_FAULT)."""
    fault = os.environ.get("HYDRO_BENCH_STRONG_FAULT")
    try:
        if fault == f"raise:{rank}":
            raise RuntimeError("injected fault")
        if fault == f"hang:{rank}":
            time.sleep(3600)
        if fault == f"hang-resident:{rank}":
            globals()["strong_leg_graph_resident"] = lambda *a, **k: time.sleep(3600)
        res = c4_strong_leg(rank, world, dev, stream, args.steps, args.warmup, collectives=multi, progress=guard.progress)
        guard.leg_done()
        return res
    except Exception as e:                                  # noqa: BLE001 - the other ranks may be inside a collective: leave, do not wait
        guard.leave(f"{e!r} on rank {rank}; the headline on this line is complete")
