"""Telemetry parity (benchmark_rtf.py:7-75, log_velocity.py:9-63) on the in-memory host (CPU)."""
import csv
import datetime

import pytest

from silver2_isaacsim_amd import config as cfg
from silver2_isaacsim_amd.telemetry import CSV_HEADER, BenchmarkRtf, LogVelocity
from silver2_isaacsim_amd.testing import FakeHost, FakeWorld


def test_rtf_meter_counts_sim_time_and_reports():
    world = FakeWorld("cpu"); host = FakeHost(world)
    t = {"now": 100.0}
    lines = []
    rtf = BenchmarkRtf(host, clock=lambda: t["now"], out=lines.append)
    rtf.on_init()
    assert lines == ["[RTF Benchmark] Initialized. Ready to run."]
    host.step(1 / 60)                              # before play: not subscribed, nothing counted
    rtf.on_play()
    for _ in range(1200):
        t["now"] += 1 / 120                        # the simulator runs 2x faster than real time
        host.step(1 / 60)
    live = [l for l in lines if l.startswith("[RTF Live]")]
    assert len(live) == 2 and live[0] == "[RTF Live] Sim: 10.00s | RTF: 2.000"      # every 600 steps
    stats = rtf.on_stop()
    assert stats["physics_steps"] == 1200
    assert stats["sim_time_s"] == pytest.approx(20.0) and stats["wall_time_s"] == pytest.approx(10.0)
    assert stats["rtf"] == pytest.approx(2.0) and stats["fps"] == pytest.approx(120.0)
    text = "\n".join(lines)
    for needle in ("BENCHMARK RESULTS", "Total Wall Time:  10.0000 s", "Total Sim Time:   20.0000 s",
                   "Physics Steps:    1200", "AVERAGE RTF:      2.0000 x", "AVERAGE FPS:      120.00"):
        assert needle in text
    # the final block, line for line (the printed artefact of benchmark_rtf.py:59-71)
    assert lines[-10:] == ["\n" + "=" * 40, "BENCHMARK RESULTS", "-" * 40, "Total Wall Time:  10.0000 s", "Total Sim Time:   20.0000 s",
                           "Physics Steps:    1200", "-" * 40, "AVERAGE RTF:      2.0000 x", "AVERAGE FPS:      120.00", "=" * 40 + "\n"]
    host.step(1 / 60)                              # stopped: ignored
    assert rtf.run.steps == 1200 and rtf.stats()["physics_steps"] == 1200
    assert BenchmarkRtf(host).on_stop() is None    # never played: no report


def test_velocity_csv_columns_and_order(tmp_path):
    world = FakeWorld("cpu"); host = FakeHost(world)
    prim = cfg.AttributeStore("Obsea_Buoy")
    world.add_body(prim.path, (1.0, 2.0, 3.0), (1, 0, 0, 0), [0.1, 0.2, 0.3, 0.4, 0.5, 0.6], 700.0)
    stamp = datetime.datetime(2026, 1, 2, 3, 4, 5)
    lg = LogVelocity(prim, host, directory=str(tmp_path), now=lambda: stamp)
    lg.on_init(); lg.on_play()
    lg.on_update(0.0, 1 / 60)
    world.positions[0, 2] = 2.5
    lg.on_update(1 / 60, 1 / 60)
    lg.on_stop()
    lg.on_update(2 / 60, 1 / 60)                   # after stop: no row
    rows = list(csv.reader(open(tmp_path / "velocity_log.csv")))
    assert rows[0] == CSV_HEADER == ["timestamp", "z_position", "linear_velocity_z", "angular_velocity_z",
                                     "x_position", "linear_velocity_x", "angular_velocity_x",
                                     "y_position", "linear_velocity_y", "angular_velocity_y"]
    assert len(rows) == 3
    assert rows[1][0] == stamp.isoformat()
    assert [float(x) for x in rows[1][1:]] == pytest.approx([3.0, 0.3, 0.6, 1.0, 0.1, 0.4, 2.0, 0.2, 0.5])
    assert float(rows[2][1]) == 2.5


def test_velocity_logger_needs_a_rigid_body(tmp_path):
    world = FakeWorld("cpu"); host = FakeHost(world)
    prim = cfg.AttributeStore("Decor", rigid_body=False)
    lg = LogVelocity(prim, host, directory=str(tmp_path)); lg.on_init(); lg.on_play()
    lg.on_update(0.0, 1 / 60)                      # warned at play, logs nothing, does not raise
    assert len(list(csv.reader(open(tmp_path / "velocity_log.csv")))) == 1
