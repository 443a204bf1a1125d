"""Closed-loop, device-resident stepping: wrench kernel + explicit integrator, ping-pong state
buffers in the tiled layout, optionally captured into a HIP graph.

The reference delegates integration to PhysX; this stand-in exists so that configs 1-5 can run for
thousands of steps without a host round-trip and so that a real-time factor can be reported the way
the reference's `benchmark_rtf.py` does (sim time / wall time).  One physics step =
`hydro_step_fused_tiled` (wrench + integrator in one kernel; `fused=False` runs
`hydro_step_wrench_tiled` followed by `hydro_integrate_tiled`, same bits).  The previous velocity is
read in place from the other state buffer.  All entry points are capture-safe, so K consecutive steps become ONE host
call (`graph_steps`), which is what makes small scenes (launch-bound at ~8 us per ctypes launch) run
at the kernels' own pace.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from . import scenes
from .engine import HydroEngine


class ClosedLoopSim:
    def __init__(self, scene: "scenes.Scene", device: int | str = 0, coeff_dtype: str | None = None,
                 fused: bool = True, implicit_drag: bool = False):
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

    # one physics step on the current stream context
    def _step_once(self) -> None:
        e = self.engine
        if self.fused:
            e.step_fused_tiled(self.cur, self.old, self.n, self.dt, implicit_drag=self.implicit_drag)   # new state -> old buffer
        else:
            e.step_wrench_tiled(self.cur, self.n, self.dt, out=self.wrench, prev=self.old)
            e.integrate_tiled(self.cur, self.wrench, self.n, self.dt, state_out=self.old)   # overwrite the old buffer
        self.cur, self.old = self.old, self.cur

    def run_eager(self, steps: int) -> None:
        with torch.cuda.stream(self.stream):
            for _ in range(steps):
                self._step_once()
        self.steps_done += steps

    def _capture(self, graph_steps: int) -> None:
        if graph_steps % 2:
            raise ValueError("graph_steps must be even (the ping-pong must return to the same buffers)")
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(self.stream):
            self.stream.synchronize()
            with torch.cuda.graph(g, stream=self.stream):
                for _ in range(graph_steps):
                    self._step_once()
        self._graph, self._graph_steps = g, graph_steps

    def run(self, steps: int, graph_steps: int = 64) -> None:
        """Advance `steps` physics steps; full groups of `graph_steps` are graph replays."""
        if graph_steps and steps >= graph_steps:
            if self._graph is None or self._graph_steps != graph_steps:
                self._capture(graph_steps)              # capturing records, it does not execute
            with torch.cuda.stream(self.stream):
                for _ in range(steps // graph_steps):
                    self._graph.replay()
            self.steps_done += (steps // graph_steps) * graph_steps
            steps %= graph_steps
        if steps:
            self.run_eager(steps)

    def synchronize(self) -> None:
        self.stream.synchronize()

    def state(self) -> np.ndarray:
        """(N,13) host copy of the current state."""
        self.synchronize()
        return scenes.from_tiled(self.cur.cpu().numpy(), self.n)

    def kinetic_energy(self, rotational: bool = True) -> np.ndarray:
        with torch.cuda.stream(self.stream):
            ke = self.engine.kinetic_energy(self.cur, rotational)
        self.synchronize()
        return ke.cpu().numpy()

    def measure_rtf(self, steps: int, graph_steps: int = 64) -> dict:
        """Real-time factor the way benchmark_rtf.py:48-71 defines it: sim time / wall time."""
        self.run(graph_steps or 2, graph_steps)         # capture + warm
        self.synchronize()
        t0 = time.perf_counter()
        self.run(steps, graph_steps)
        self.synchronize()
        wall = time.perf_counter() - t0
        return {"physics_steps": steps, "wall_time_s": wall, "sim_time_s": steps * self.dt,
                "rtf": steps * self.dt / wall, "fps": steps / wall,
                "body_steps_per_s": steps * self.n / wall, "us_per_step": wall / steps * 1e6}

    def close(self) -> None:
        self._graph = None
        self.engine.close()
