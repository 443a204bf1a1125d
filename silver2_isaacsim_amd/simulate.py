"""Closed-loop, device-resident stepping: wrench kernel + explicit integrator, ping-pong state
buffers in the tiled layout, optionally captured into a HIP graph.

The reference delegates integration to PhysX; this stand-in exists so that configs 1-5 can run for
thousands of steps without a host round-trip and so that a real-time factor can be reported the way
the reference's `benchmark_rtf.py` does (sim time / wall time).  One physics step =
`hydro_step_fused_tiled` (wrench + integrator in one kernel; `fused=False` runs
`hydro_step_wrench_tiled` followed by `hydro_integrate_tiled`, same bits).  The previous velocity is
read in place from the other state buffer.  All entry points are capture-safe, so K consecutive steps become ONE host
call (`graph_steps`), which is what makes small scenes (launch-bound at ~8 us per ctypes launch) run
at the kernels' own pace.  With a kinetic-energy monitor (`ke_every`) a replay that ends on a sampling point is of a graph
that also carries that sample's whole pipeline - sampling step, copy to pinned memory (`KineticEnergyMonitor.capture_sample`) -
where no other rank is involved; with more than one rank the sample's all-reduce is host-driven on a side stream by default
 and capturing it into the graph is opt-in (`graph_resident_sampling=True`).  Waiting for samples or for the
stream takes a deadline (`collect(timeout_s=)`, `synchronize(timeout_s=)`): a collective a rank never joins raises, it does not hang.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from . import distributed as hd
from . import scenes
from .engine import HydroEngine


class KineticEnergyMonitor:
    """The one collective of the path: global kinetic energy, off the step path.

    Every `every` steps the rank's shard is reduced ON DEVICE to one float64 pair [translational, rotational]
    (wave64 shuffles -> LDS -> one partial per block -> fixed-order second stage), on the stream the steps run on -
    either INSIDE the step kernel, for the bodies it holds in registers anyway (`hydro_step_*_tiled_ke`: the caller
    passes the pair as `sampled=`; no second pass over the state), or by the stand-alone `hydro_kinetic_energy*` on
    the state passed to `observe`.  The pair is then summed over the ranks on a SIDE stream -
    `all_reduce(async_op=True)`, RCCL over xGMI under backend "nccl" (16 bytes: latency, not bandwidth), gloo on a
    pinned host copy otherwise - and copied to pinned host memory.  The step stream never waits for any of it (all it
    does for a sample is a 16-byte device copy into the sample's slot); the host picks a sample up `every` steps later
    (`collect()`), when it has long arrived.  The reference has no
    counterpart (single process, no reduction of any kind); its oracle is an fp64 NumPy sum.

    `reduce_local(out)` writes the rank's float64 pair into `out` (a tensor on `device`) using the current stream;
    the default is `engine.kinetic_energy(state, rotational, out=out)` on the state passed to `observe`."""

    def __init__(self, engine: HydroEngine | None = None, every: int = 64, rotational: bool = True, slots: int = 4,
                 device: torch.device | str | None = None, reduce_local=None, timeout_s: float | None = 300.0):
        if every <= 0 or slots < 2:
            raise ValueError("every must be positive, slots at least 2")
        if engine is None and device is None:
            raise ValueError("KineticEnergyMonitor needs an engine, or a device (with reduce_local=, or with observe(..., sampled=))")
        self.engine, self.every, self.rotational = engine, int(every), bool(rotational)
        self.device = torch.device(device) if device is not None else engine.device
        self._reduce_local = reduce_local
        self._gpu = self.device.type == "cuda"
        self._nccl = self._gpu and hd.collective_device(self.device).type == "cuda"
        self._dev = [torch.zeros(2, dtype=torch.float64, device=self.device) for _ in range(slots)]
        self._host = [torch.zeros(2, dtype=torch.float64, pin_memory=self._gpu) for _ in range(slots)]
        self._side = torch.cuda.Stream(self.device) if self._gpu else None
        # one set of events per slot, reused (creating a HIP event costs more than recording one)
        self._ev = [tuple(torch.cuda.Event() for _ in range(3)) for _ in range(slots)] if self._gpu else None
        self._pending: list = []              # (step, slot, work handle or None, completion event or None)
        self._next_slot = 0
        self.samples: list = []               # (step, [translational, rotational]) in submission order
        self.submitted = 0
        self.waited_on_host = 0               # samples the host had to wait for (0 when `every` covers the latency)
        # deadline of every wait the monitor does on its own (warm_up, a full ring, reserve); collect(timeout_s=) overrides it.
        # None = wait without limit.  A collective that a rank never joins raises TimeoutError instead of hanging the host.
        self.timeout_s = timeout_s

    def warm_up(self, stream=None) -> None:
        """One full pass of the sampling pipeline per slot with a dummy pair, discarded.  The FIRST pass through it creates
        the side stream's hardware queue, maps the pinned host buffers and (under nccl) takes RCCL's first-call path -
        measured 0.4 ms of host time, which belongs in no timed region (bench.py's configs[3] leg times 20 steps).
        COLLECTIVE under a process group: every rank calls it at the same point."""
        dummy = torch.zeros(2, dtype=torch.float64, device=self.device)
        for k in range(len(self._dev)):
            self.observe(self.every * (k + 1), stream=stream, sampled=dummy)
            try:
                self.collect(block=True)
            except TimeoutError as e:
                raise TimeoutError(f"warm-up pass {k} of the monitor: {e}") from None
        self.samples.clear()
        self.submitted = self.waited_on_host = self._next_slot = 0

    # ---- graph-resident sampling: the whole pipeline of a sample captured into the caller's HIP graph -------------------------
    @property
    def graph_capturable(self) -> bool:
        """True when a sample's pipeline can live inside a HIP graph: on a GPU, with the collective on the device (RCCL under
        backend nccl - RCCL kernels are capturable) or without a process group.  False under gloo (a CPU collective)."""
        return self._gpu and (self._nccl or not hd._collectives_on())

    def slot_buffer(self, slot: int) -> torch.Tensor:
        """The float64 device pair of ring slot `slot`: pass it as `ke_out=` to the sampling step that is being captured."""
        return self._dev[slot]

    def capture_sample(self, slot: int) -> None:
        """Call INSIDE a graph capture, on the capturing stream, right after the sampling step that wrote `slot_buffer(slot)`:
        records the rest of the sample's pipeline into the graph - the all-reduce over the ranks (RCCL; the capturing stream
        joins it) and the 16-byte copy to pinned host memory.  A replay of that graph then takes the sample with NO host
        work at all (the host-driven `observe` costs 30-70 us of host time per sample, which is what bounds a short region
        of small steps); `submit_captured` tells the monitor after each replay.  COLLECTIVE under a process group: every
        rank captures and replays the same graphs in the same order."""
        if not self.graph_capturable:
            raise RuntimeError("capture_sample needs a device-side collective (backend nccl) or no process group")
        dev_buf = self._dev[slot]
        if self._nccl:
            hd.all_reduce_sum_(dev_buf)                     # (not async: the capturing stream is ordered after RCCL's)
        self._host[slot].copy_(dev_buf, non_blocking=True)

    def reserve(self, slot: int) -> None:
        """Before replaying a graph that samples into `slot`: make sure the previous sample of that slot has been picked up
        (it has, long ago, unless replays that sample come back to back - then this waits for it)."""
        while any(p[1] == slot for p in self._pending):
            self.collect(block_oldest=True)                 # (under self.timeout_s)

    def submit_captured(self, step: int, slot: int, stream=None) -> None:
        """After a replay of a graph that carries `capture_sample(slot)`: the sample of physics step `step` is on its way."""
        stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        done = self._ev[slot][2]
        done.record(stream)
        self._pending.append((step, slot, None, done))
        self.submitted += 1

    def wait_before_overwrite(self, stream=None) -> None:
        """Kept for callers of rounds 3-4: nothing to wait for any more - `observe(sampled=...)` takes its copy of the
        caller's buffer on the step stream itself, so a later step on that stream may overwrite the buffer at once."""
        return None

    def observe(self, step: int, state: torch.Tensor | None = None, stream=None, sampled: torch.Tensor | None = None) -> bool:
        """Call after physics step `step` (1-based count of completed steps) with the state that step produced - or
        with `sampled`, the float64 pair a sampling step kernel (ke_out=) has already written on `stream`.
        Submits a sample when `step` is a multiple of `every`; returns True if it did."""
        if step % self.every:
            return False
        if sampled is None and state is None and self._reduce_local is None:
            raise ValueError("observe() needs the state, a sampled pair, or a reduce_local callback")
        self.collect(block_oldest=len(self._pending) >= len(self._dev) - 1)      # free a slot if the ring is full
        slot = self._next_slot
        self._next_slot = (slot + 1) % len(self._dev)
        self.reserve(slot)                                  # (graph-resident samples use slots 0 / 1 of the same ring)
        dev_buf, host_buf = self._dev[slot], self._host[slot]
        if self._gpu:
            stream = stream if stream is not None else torch.cuda.current_stream(self.device)
            ready, _, done = self._ev[slot]                 # (the slot is free: its previous sample has been collected)
            # The pair goes into this sample's slot ON THE STEP STREAM: the stand-alone reduction writes it there, a pair that a
            # sampling step kernel left in the caller's buffer is copied there (16 bytes, device to device, ~3 us of the step
            # stream's time).  The caller's buffer is then free at once - the next sampling step may overwrite it without
            # waiting for anything.  (Rounds 3-4 made that copy on the SIDE stream and had the next sampling step wait for
            # it, a round trip step stream -> side stream -> step stream: 12.3 against 10.7 us per step in bench.py's 20-step
            # configs[3] leg on one GPU, profiles/r05_monitor_copy_ab.log.)
            same = torch.cuda.current_stream(self.device) == stream
            if sampled is None:
                if same:
                    self._local(state, dev_buf)
                else:
                    with torch.cuda.stream(stream):
                        self._local(state, dev_buf)
            else:
                src = sampled if sampled.numel() == 2 else sampled[:2]
                if same:
                    dev_buf.copy_(src, non_blocking=True)
                else:
                    with torch.cuda.stream(stream):
                        dev_buf.copy_(src, non_blocking=True)
            ready.record(stream)
            self._side.wait_event(ready)                    # the side stream, not the host, waits for the pair
            work = None
            with torch.cuda.stream(self._side):
                if self._nccl:
                    work = hd.all_reduce_sum_(dev_buf, async_op=True)
                    if work is not None:
                        work.wait()                         # orders the SIDE stream after RCCL's; the host does not block
                host_buf.copy_(dev_buf, non_blocking=True)
                done.record(self._side)
            self._pending.append((step, slot, None if self._nccl else "gloo", done))
        else:
            if sampled is not None:
                dev_buf.copy_(sampled[:2])
            else:
                self._local(state, dev_buf)
            host_buf.copy_(dev_buf)
            self._pending.append((step, slot, hd.all_reduce_sum_(host_buf, async_op=True), None))
        self.submitted += 1
        return True

    def _local(self, state, out) -> None:
        if self._reduce_local is not None:
            self._reduce_local(out)
        else:
            self.engine.kinetic_energy(state, self.rotational, out=out)

    def collect(self, block: bool = False, block_oldest: bool = False, timeout_s: float | None = None) -> list:
        """Move finished samples to `samples` (all of them, waiting if need be, with block=True).
        timeout_s: the longest the host waits for ONE sample (a monotonic-clock poll of its completion event / work handle);
        past it a TimeoutError names the sample's step and this rank - a collective some rank never joined must not hang
        the host for ever.  The sample stays pending (nothing is retried, nothing re-executed); None = the monitor's own
        `timeout_s` (300 s unless constructed otherwise; None there = wait without limit)."""
        if timeout_s is None:
            timeout_s = self.timeout_s
        out = []
        while self._pending:
            step, slot, work, done = self._pending[0]
            must = block or block_oldest or work == "gloo"      # (gloo ranks must reach their all-reduce in step)
            if done is not None:                            # GPU: the pinned copy is complete when `done` has fired
                if not done.query():
                    if not must:
                        break
                    self.waited_on_host += 1
                    self._wait(done.query, done.synchronize, timeout_s, step, "its device pipeline (reduction, all-reduce, pinned copy)")
                if work == "gloo":                          # ranks share nothing but the host here (tests, rehearsals)
                    handle = hd.all_reduce_sum_(self._host[slot], async_op=timeout_s is not None)
                    if handle is not None:
                        self._wait(handle.is_completed, handle.wait, timeout_s, step, "the gloo all-reduce of its host pair")
                        handle.wait()
            elif work is not None:                          # CPU + gloo: asynchronous handle
                if not work.is_completed():
                    if not must:
                        break
                    self.waited_on_host += 1
                    self._wait(work.is_completed, work.wait, timeout_s, step, "its all-reduce")
                work.wait()
            self._pending.pop(0)
            block_oldest = False
            sample = (step, [float(x) for x in self._host[slot].tolist()])
            self.samples.append(sample)
            out.append(sample)
        return out

    @staticmethod
    def _wait(is_done, wait, timeout_s: float | None, step: int, what: str) -> None:
        if timeout_s is None:
            wait()
            return
        t0 = time.monotonic()
        while not is_done():
            waited = time.monotonic() - t0
            if waited > timeout_s:
                rank = hd.env_rank_world()[0]
                raise TimeoutError(f"kinetic-energy sample of step {step} on rank {rank}: {what} did not finish within "
                                   f"{timeout_s:g} s (a rank that never joined the collective, or a stalled device)")
            if waited > 2e-4:                               # a sample that is nearly there is spun for; a late one is slept for
                time.sleep(5e-5)

    def last(self):
        """The newest sample the host has PICKED UP (collect()); a sample in flight is not in it.  ClosedLoopSim.run polls
        collect() (non-blocking) before every sampling replay, so this lags by at most one sampling period."""
        return self.samples[-1] if self.samples else None


class ClosedLoopSim:
    def __init__(self, scene: "scenes.Scene", device: int | str = 0, coeff_dtype: str | None = None,
                 fused: bool = True, implicit_drag: bool = False, ke_every: int = 0, graph_resident_sampling: bool | None = None,
                 sample_timeout_s: float | None = 300.0):
        if implicit_drag and not fused:
            raise ValueError("implicit drag needs the fused step (the drag coefficients never leave the kernel)")
        self.implicit_drag = implicit_drag
        self.scene = scene
        self.fused = fused                                      # one kernel per step (hydro_step_fused_tiled)
        self.n = scene.n
        self.dt = scene.dt
        self.engine = HydroEngine(scene.n, device, scene.rho, scene.g)
        self.engine.set_params(scene.params, coeff_dtype or scene.coeff_dtype)
        dev = self.engine.device
        self.cur = torch.from_numpy(scenes.to_tiled(scene.state)).to(dev)
        prev_state = np.zeros_like(scene.state)
        prev_state[:, 7:13] = scene.prev                       # only the velocity fields of "previous" matter
        self.old = torch.from_numpy(scenes.to_tiled(prev_state)).to(dev)
        self.wrench = self.engine.alloc_tiled(6, scene.n)
        self.stream = torch.cuda.Stream(dev)
        self.steps_done = 0
        self._graph = None
        self._graph_steps = 0
        self._graph_bufs = None
        self._graph_sampling: list = []
        self._captured_samples = 0
        # optional global kinetic energy every `ke_every` steps (asynchronous, see KineticEnergyMonitor).  The fused
        # step SAMPLES it for the bodies it has in registers (ke_out=): no extra pass over the state.  With HIP-graph
        # replays the sampling step is the last step of the graph, so `ke_every` must be a multiple of graph_steps.
        self.monitor = KineticEnergyMonitor(self.engine, every=ke_every, timeout_s=sample_timeout_s) if ke_every else None
        # graph replays: may a sample's pipeline (all-reduce included) be captured into the step graph?  By default ONLY where no
        # other rank is involved (no process group, or a one-rank group): there a replay that samples costs the host nothing.
        # With more than one rank the default is the host-driven pipeline (`observe`: asynchronous all-reduce on a side stream,
        # SURVEY.md 8e - 5 us of host time per sampled step); a collective captured into a graph is OPT-IN there
        # (graph_resident_sampling=True or HYDRO_GRAPH_SAMPLING=1) until multi-GPU hardware has run it.  False / =0: never.
        env = os.environ.get("HYDRO_GRAPH_SAMPLING")
        if graph_resident_sampling is not None:
            want = bool(graph_resident_sampling)
        elif env is not None:
            want = env != "0"
        else:
            want = not hd._collectives_on() or torch.distributed.get_world_size() == 1
        self._wants_graph_sampling = want
        self._graph_sampling_ok = want and self.monitor is not None and self.monitor.graph_capturable
        self.ke_dev = torch.zeros(2, dtype=torch.float64, device=dev) if ke_every else None
        self._monitor_warm = self.monitor is None

    def _warm_monitor(self) -> None:
        """First run*() with a monitor: one discarded pass of the sampling pipeline per slot (side-stream queue, pinned
        mappings, RCCL's first call).  COLLECTIVE under a process group - like every run that samples, and unlike the
        constructor, which is local: ranks may build their sims in any order."""
        if not self._monitor_warm:
            self._monitor_warm = True
            with torch.cuda.stream(self.stream):
                self.monitor.warm_up(self.stream)

    # one physics step on the current stream context; sample=True: the step also leaves the kinetic energy of the state it
    # produces in self.ke_dev (fused step: inside the kernel; two-kernel path: the stand-alone reduction afterwards)
    def _step_once(self, sample: bool = False, ke_out: torch.Tensor | None = None) -> None:
        e = self.engine
        ke_out = (ke_out if ke_out is not None else self.ke_dev) if sample else None
        if self.fused:
            e.step_fused_tiled(self.cur, self.old, self.n, self.dt, implicit_drag=self.implicit_drag, ke_out=ke_out)   # new state -> old buffer
        else:
            e.step_wrench_tiled(self.cur, self.n, self.dt, out=self.wrench, prev=self.old)
            e.integrate_tiled(self.cur, self.wrench, self.n, self.dt, state_out=self.old)   # overwrite the old buffer
            if sample:
                e.kinetic_energy(self.old, True, out=ke_out)
        self.cur, self.old = self.old, self.cur

    def run_eager(self, steps: int) -> None:
        self._warm_monitor()
        with torch.cuda.stream(self.stream):
            for _ in range(steps):
                sample = self.monitor is not None and (self.steps_done + 1) % self.monitor.every == 0
                if sample:
                    self.monitor.wait_before_overwrite(self.stream)
                self._step_once(sample)
                self.steps_done += 1
                if sample:
                    self.monitor.observe(self.steps_done, stream=self.stream, sampled=self.ke_dev)

    def _capture(self, graph_steps: int) -> None:
        if graph_steps % 2:
            raise ValueError("graph_steps must be even (the ping-pong must return to the same buffers)")
        if self.monitor is not None and self.monitor.every % graph_steps:
            raise ValueError(f"ke_every ({self.monitor.every}) must be a multiple of graph_steps ({graph_steps}): with graph "
                             f"replays the kinetic energy is sampled by the last step of a replay")
        # One graph of PLAIN steps, replayed wherever no sample is due at its end; with a monitor, sampling graphs for the
        # replays that end on a sampling point.  Where the collective can live in a graph (backend nccl, or no group:
        # KineticEnergyMonitor.graph_capturable) there are two of them, one per ring slot 0 / 1, and each carries the WHOLE
        # pipeline of its sample - the sampling step writes the slot's device pair, the all-reduce over the ranks and the
        # copy to pinned host memory follow inside the capture (capture_sample): a replay takes the sample, the host only
        # records an event.  Otherwise (gloo) one sampling graph leaves the pair in self.ke_dev and the host drives the rest
        # (observe).  COLLECTIVE under a process group, like every replay of a sampling graph.
        def capture(sample_into=None, slot=None):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local"):
                for k in range(graph_steps):
                    self._step_once(sample=sample_into is not None and k == graph_steps - 1, ke_out=sample_into)
                if slot is not None:
                    self.monitor.capture_sample(slot)
            return g
        with torch.cuda.stream(self.stream):
            self.stream.synchronize()
            plain = capture()
            sampling = []
            if self.monitor is not None:
                if self._graph_sampling_ok:
                    sampling = [capture(self.monitor.slot_buffer(j), j) for j in (0, 1)]
                else:
                    sampling = [capture(self.ke_dev)]
        # the graphs hard-code which physical buffer is "current": valid only while the ping-pong is in this phase
        self._graph, self._graph_steps, self._graph_bufs = plain, graph_steps, (self.cur.data_ptr(), self.old.data_ptr())
        self._graph_sampling = sampling

    def run(self, steps: int, graph_steps: int = 64) -> None:
        """Advance `steps` physics steps; full groups of `graph_steps` are graph replays.  With a monitor under a process
        group this is COLLECTIVE (every rank runs the same number of steps with the same cadence)."""
        self._warm_monitor()
        if graph_steps and steps >= graph_steps:
            if self.steps_done % graph_steps and self.monitor is not None:
                raise ValueError("with a kinetic-energy monitor, graph replays must start at a multiple of graph_steps")
            # (an odd number of eager steps or resident launches since the capture leaves the buffers swapped: a replay
            # would step the stale one - recapture for the phase the ping-pong is in now)
            if self._graph is None or self._graph_steps != graph_steps or self._graph_bufs != (self.cur.data_ptr(), self.old.data_ptr()):
                self._capture(graph_steps)              # capturing records, it does not execute
            mon = self.monitor
            with torch.cuda.stream(self.stream):
                for _ in range(steps // graph_steps):
                    due = mon is not None and (self.steps_done + graph_steps) % mon.every == 0
                    if not due:
                        self._graph.replay()
                        self.steps_done += graph_steps
                    elif self._graph_sampling_ok:                   # the sample rides in the graph: no host work
                        j = self._captured_samples % 2
                        mon.collect()                               # (non-blocking, ~1 us: finished samples become visible to last())
                        mon.reserve(j)
                        self._graph_sampling[j].replay()
                        self.steps_done += graph_steps
                        mon.submit_captured(self.steps_done, j, self.stream)
                        self._captured_samples += 1
                    else:
                        self._graph_sampling[0].replay()
                        self.steps_done += graph_steps
                        mon.observe(self.steps_done, stream=self.stream, sampled=self.ke_dev)
            steps %= graph_steps
        if steps:
            self.run_eager(steps)

    def run_resident(self, steps: int, chunk: int = 64) -> None:
        """Advance `steps` physics steps with the bodies RESIDENT IN REGISTERS: one hydro_step_fused_tiled_multi launch
        per `chunk` steps (and one for the remainder).  The bodies are independent, so a launch reads every body once,
        carries it through `chunk` steps and writes it once - no HBM traffic and no launch between the steps, same bits
        as run_eager.  States between the ends of chunks never exist in memory: a kinetic-energy monitor samples at the
        end of a chunk, so `ke_every` must be a multiple of `chunk` (and the run must start on such a boundary)."""
        if not self.fused:
            raise ValueError("the resident loop is the fused step")
        if chunk < 1:
            raise ValueError("chunk must be >= 1")
        if self.monitor is not None and (self.monitor.every % chunk or self.steps_done % chunk):
            raise ValueError(f"ke_every ({self.monitor.every}) must be a multiple of chunk ({chunk}) and the run must start "
                             f"at one: the kinetic energy is sampled by the last step of a launch")
        self._warm_monitor()
        with torch.cuda.stream(self.stream):
            while steps > 0:
                k = min(chunk, steps)
                sample = self.monitor is not None and k == chunk and (self.steps_done + k) % self.monitor.every == 0
                if sample:
                    self.monitor.wait_before_overwrite(self.stream)
                self.engine.step_fused_tiled_multi(self.cur, self.old, self.n, self.dt, k, implicit_drag=self.implicit_drag,
                                                   ke_out=self.ke_dev if sample else None)
                self.cur, self.old = self.old, self.cur
                self.steps_done += k
                steps -= k
                if sample:
                    self.monitor.observe(self.steps_done, stream=self.stream, sampled=self.ke_dev)

    def synchronize(self, timeout_s: float | None = None) -> None:
        """Wait for everything submitted to the step stream.  timeout_s: poll an event against a monotonic clock instead of
        blocking in the driver, and raise TimeoutError (naming the step count and this rank) past it - a captured collective
        that some rank never replays would otherwise hang the host with no deadline."""
        if timeout_s is None:
            self.stream.synchronize()
            return
        ev = torch.cuda.Event()
        ev.record(self.stream)
        t_end = time.monotonic() + timeout_s
        while not ev.query():
            if time.monotonic() > t_end:
                raise TimeoutError(f"closed loop on rank {hd.env_rank_world()[0]}: the step stream had not drained {timeout_s:g} s after "
                                   f"step {self.steps_done} was submitted (a collective some rank never joined, or a stalled device)")
            time.sleep(2e-4)

    def state(self) -> np.ndarray:
        """(N,13) host copy of the current state."""
        self.synchronize()
        return scenes.from_tiled(self.cur.cpu().numpy(), self.n)

    def kinetic_energy(self, rotational: bool = True) -> np.ndarray:
        with torch.cuda.stream(self.stream):
            ke = self.engine.kinetic_energy(self.cur, rotational)
        self.synchronize()
        return ke.cpu().numpy()

    def measure_rtf(self, steps: int, graph_steps: int = 64, resident: bool = False, warm_seconds: float = 0.0) -> dict:
        """Real-time factor the way benchmark_rtf.py:48-71 defines it: sim time / wall time.
        resident=True: run_resident with chunk = graph_steps instead of graph replays of single steps.
        warm_seconds: keep stepping untimed for that long first (a GPU coming out of idle needs ~50 ms to reach the clock
        it then holds; a timed region of a few milliseconds right after one warm launch measures the ramp)."""
        go = (lambda k: self.run_resident(k, graph_steps or 64)) if resident else (lambda k: self.run(k, graph_steps))
        go(graph_steps or 2)                            # capture + warm
        self.synchronize()
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < warm_seconds:
            go(graph_steps or 2)
            self.synchronize()
        t0 = time.perf_counter()
        go(steps)
        self.synchronize()
        wall = time.perf_counter() - t0
        return {"physics_steps": steps, "wall_time_s": wall, "sim_time_s": steps * self.dt,
                "rtf": steps * self.dt / wall, "fps": steps / wall,
                "body_steps_per_s": steps * self.n / wall, "us_per_step": wall / steps * 1e6}

    def close(self) -> None:
        self._graph = None
        self._graph_sampling = []
        self.engine.close()
