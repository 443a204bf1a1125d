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
from dataclasses import dataclass
from typing import Callable

log = logging.getLogger("silver2_isaacsim_amd")

LIVE_EVERY_STEPS = 600                                   # benchmark_rtf.py:45
CSV_FILE_NAME = "velocity_log.csv"                       # log_velocity.py:13
CSV_HEADER = ["timestamp",                               # log_velocity.py:17-20
              "z_position", "linear_velocity_z", "angular_velocity_z",
              "x_position", "linear_velocity_x", "angular_velocity_x",
              "y_position", "linear_velocity_y", "angular_velocity_y"]


# The artefact IS the text (SURVEY.md 8f row 3 asks for the reference's RTF printer): every line the meter can emit, in one
# table, so that tests/test_telemetry.py pins the format and the class below holds no format strings of its own.
RTF_LINES = {
    "ready": "[RTF Benchmark] Initialized. Ready to run.",
    "live": "[RTF Live] Sim: {sim:.2f}s | RTF: {rtf:.3f}",
    "report": ("\n" + "=" * 40,
               "BENCHMARK RESULTS",
               "-" * 40,
               "Total Wall Time:  {wall_time_s:.4f} s",
               "Total Sim Time:   {sim_time_s:.4f} s",
               "Physics Steps:    {physics_steps}",
               "-" * 40,
               "AVERAGE RTF:      {rtf:.4f} x",
               "AVERAGE FPS:      {fps:.2f}",
               "=" * 40 + "\n"),
}
MIN_WALL_S = 0.001                                       # shorter runs report nothing (benchmark_rtf.py:55-57)


@dataclass
class RtfRun:
    """One play -> stop interval of the meter."""
    started: float                                       # wall clock at play
    sim_time: float = 0.0                                # sum of the physics dt handed to the step callback
    steps: int = 0

    def summary(self, now: float) -> dict | None:
        wall = now - self.started
        if wall < MIN_WALL_S:
            return None
        return {"wall_time_s": wall, "sim_time_s": self.sim_time, "physics_steps": self.steps,
                "rtf": self.sim_time / wall, "fps": self.steps / wall}


class BenchmarkRtf:
    """Standalone real-time-factor meter (whole-simulator metric, not a kernel benchmark): RTF = simulated time / wall
    time between play and stop, a live line every 600 physics steps, a final block at stop.  `clock` and `out` are
    injectable (tests; a host that wants the lines in its own log)."""

    def __init__(self, host=None, clock: Callable[[], float] = time.time, out: Callable[[str], None] = print):
        self._host, self._clock, self._out = host, clock, out
        self._token = None                               # the physics-step subscription while playing
        self.run: RtfRun | None = None                   # the current (or last) interval
        self._live = False

    # lifecycle, as Kit calls a behavior script
    def on_init(self):
        self._token, self._live = None, False
        self._out(RTF_LINES["ready"])

    def on_destroy(self):
        self._token = None

    def on_play(self):
        self.run = RtfRun(started=self._clock())
        self._live = True
        if self._host is not None:
            self._token = self._host.subscribe_physics_step(self._on_physics_step)
        log.info("[RTF Benchmark] Benchmarking started...")

    def on_stop(self):
        summary = self.stats()
        if summary is not None:
            for line in RTF_LINES["report"]:
                self._out(line.format(**summary))
        self._live, self._token = False, None
        return summary

    def _on_physics_step(self, delta_time: float):
        run = self.run
        if not self._live or run is None:
            return
        run.sim_time += delta_time
        run.steps += 1
        if run.steps % LIVE_EVERY_STEPS == 0:
            wall = self._clock() - run.started
            self._out(RTF_LINES["live"].format(sim=run.sim_time, rtf=run.sim_time / wall if wall > 0 else float("inf")))

    def stats(self) -> dict | None:
        """{"wall_time_s", "sim_time_s", "physics_steps", "rtf", "fps"} of the current interval; None before the first play
        and for intervals shorter than a millisecond."""
        return None if self.run is None else self.run.summary(self._clock())


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
