"""Telemetry parity: the reference's two validation / performance artefacts.

* `BenchmarkRtf`  - real-time-factor meter, counterpart of `benchmark_rtf.py:7-75`:
  accumulates the physics dt handed to the step callback, counts steps, prints a live line every
  600 steps and a final block with wall time, sim time, steps, `RTF = sim / wall`, `FPS = steps / wall`.
* `LogVelocity`   - per-frame CSV logger, counterpart of `log_velocity.py:9-63`: same file name,
  same ten columns in the same (z, x, y) order.

Both talk to the simulator through the `SimHost` / `BodyView` protocols of `behavior.py`, so they run
under Kit and under `silver2_isaacsim_amd.testing.FakeHost` alike.
"""
from __future__ import annotations

import csv
import datetime
import logging
import os
import time
from typing import Callable

log = logging.getLogger("silver2_isaacsim_amd")

LIVE_EVERY_STEPS = 600                                   # benchmark_rtf.py:45
CSV_FILE_NAME = "velocity_log.csv"                       # log_velocity.py:13
CSV_HEADER = ["timestamp",                               # log_velocity.py:17-20
              "z_position", "linear_velocity_z", "angular_velocity_z",
              "x_position", "linear_velocity_x", "angular_velocity_x",
              "y_position", "linear_velocity_y", "angular_velocity_y"]


class BenchmarkRtf:
    """Standalone real-time-factor meter (whole-simulator metric, not a kernel benchmark)."""

    def __init__(self, host=None, clock: Callable[[], float] = time.time, out: Callable[[str], None] = print):
        self._host = host
        self._clock = clock
        self._out = out
        self._physx_subscription = None
        self._running = False

    def on_init(self):
        self._physx_subscription = None
        self._running = False
        self._out("[RTF Benchmark] Initialized. Ready to run.")

    def on_destroy(self):
        self._physx_subscription = None

    def on_play(self):
        self._start_wall_time = self._clock()
        self._total_sim_time = 0.0
        self._frame_count = 0
        self._running = True
        if self._host is not None:
            self._physx_subscription = self._host.subscribe_physics_step(self._on_physics_step)
        log.info("[RTF Benchmark] Benchmarking started...")

    def on_stop(self):
        stats = self._report_final_stats()
        self._running = False
        self._physx_subscription = None
        return stats

    def _on_physics_step(self, delta_time: float):
        if not self._running:
            return
        self._total_sim_time += delta_time
        self._frame_count += 1
        if self._frame_count % LIVE_EVERY_STEPS == 0:
            self._print_live_stats()

    def stats(self) -> dict | None:
        if not hasattr(self, "_start_wall_time"):
            return None
        wall = self._clock() - self._start_wall_time
        if wall < 0.001:                                 # benchmark_rtf.py:55-57
            return None
        return {"wall_time_s": wall, "sim_time_s": self._total_sim_time, "physics_steps": self._frame_count,
                "rtf": self._total_sim_time / wall, "fps": self._frame_count / wall}

    def _report_final_stats(self):
        s = self.stats()
        if s is None:
            return None
        bar, rule = "=" * 40, "-" * 40
        self._out(f"\n{bar}")
        self._out("BENCHMARK RESULTS")
        self._out(rule)
        self._out(f"Total Wall Time:  {s['wall_time_s']:.4f} s")
        self._out(f"Total Sim Time:   {s['sim_time_s']:.4f} s")
        self._out(f"Physics Steps:    {s['physics_steps']}")
        self._out(rule)
        self._out(f"AVERAGE RTF:      {s['rtf']:.4f} x")
        self._out(f"AVERAGE FPS:      {s['fps']:.2f}")
        self._out(f"{bar}\n")
        return s

    def _print_live_stats(self):
        current_wall = self._clock() - self._start_wall_time
        rtf = self._total_sim_time / current_wall if current_wall > 0 else float("inf")
        self._out(f"[RTF Live] Sim: {self._total_sim_time:.2f}s | RTF: {rtf:.3f}")


class LogVelocity:
    """CSV logger of one prim's pose / velocity, one row per render frame (`on_update`)."""

    def __init__(self, prim=None, host=None, directory: str | None = None,
                 now: Callable[[], datetime.datetime] = datetime.datetime.now):
        self.prim = prim
        self._host = host
        self._now = now
        self._directory = directory or os.path.dirname(os.path.abspath(__file__))
        self._rigid_prim = None

    def on_init(self):
        self._log_file_path = os.path.join(self._directory, CSV_FILE_NAME)

    def on_play(self):
        self._setup()
        try:
            with open(self._log_file_path, "w", newline="") as f:
                csv.writer(f).writerow(CSV_HEADER)
            log.info("Log file created at %s", self._log_file_path)
        except Exception as e:                           # noqa: BLE001 - reference logs and carries on (:26-27)
            log.error("Failed to create log file: %s", e)

    def on_stop(self):
        self._rigid_prim = None

    def on_update(self, current_time: float, delta_time: float):
        if self._rigid_prim is None:
            return
        positions, _ = self._rigid_prim.get_world_poses()
        vel = self._rigid_prim.get_velocities()
        p, v = positions[0], vel[0]
        f = lambda x: float(x)                           # noqa: E731
        row = [self._now().isoformat(),
               f(p[2]), f(v[2]), f(v[5]),
               f(p[0]), f(v[0]), f(v[3]),
               f(p[1]), f(v[1]), f(v[4])]
        try:
            with open(self._log_file_path, "a", newline="") as fh:
                csv.writer(fh).writerow(row)
        except Exception as e:                           # noqa: BLE001
            log.warning("Failed to write to log file: %s", e)

    def _setup(self):
        if not self._host.has_rigid_body_api(self.prim):
            log.warning("HydrodynamicsComponent on prim %s requires a RigidBody component.", self._host.prim_path(self.prim))
            return
        self._rigid_prim = self._host.make_rigid_view([self._host.prim_path(self.prim)], "log_velocity_view")
        self._rigid_prim.initialize()
