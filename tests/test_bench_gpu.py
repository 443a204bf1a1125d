"""bench.py contract: one JSON line with the fields the driver reads."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_bench_json_contract(native_built):
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
           "--bodies", "65536", "--cpu-seconds", "1", "--no-extras"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{")          # exactly one line on stdout, whatever libraries print
    assert len(lines[0].encode()) <= 8192                        # the driver must be able to read it (r05: 22.5 KB was not parsed)
    d = json.loads(lines[0])
    assert d["ok"] is True and "extras" not in d and "configs" in d and set(d["configs"]) == {"c2", "c3", "c4_shard", "c4"}
    for c in d["configs"].values():                              # SURVEY 8d's per-config absolutes are driver-visible
        assert c["us_per_step"] > 0 and c["graph_us_per_step"] > 0 and c["body_steps_per_s"] == pytest.approx(c["n"] / (c["us_per_step"] * 1e-6), rel=1e-6)
    assert not any(isinstance(v, str) and len(v) > 160 for v in _leaves(d))      # numbers and names, no prose (bench.py --explain has it)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "body-steps/s" and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    # ONE clock: `frac` is on the interval `value` and `ms_per_step` come from (bytes per launch / ms_per_step) ...
    assert r["frac"] * r["peak"] * 1e9 * d["ms_per_step"] * 1e-3 == pytest.approx(r["algorithmic_bytes_per_launch"], rel=1e-7)
    assert d["value"] == pytest.approx(65536 / (d["ms_per_step"] * 1e-3), rel=1e-9)
    # ... and the HIP-event figure of the same steps is kept beside it (shorter: no host synchronisation in it)
    assert r["frac_contract_steps"] >= r["frac"] and r["kernel_us"] <= r["step_us"] * 1.0001
    assert r["traffic_bytes_per_body"] == 122 and r["frac_traffic"] == pytest.approx(r["frac"] * 122 / 130, rel=1e-6)
    assert r["resident"] in ("hbm", "infinity-cache") and "roofline_4m" not in d      # 65 536 bodies x 4: a cache-resident test size
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    assert c["gpu_vs_oracle_max_rel_err"] <= 1e-5 and c["gpu_vs_oracle_n_over_1e-5"] == 0     # the CPU leg is also the checker
    assert d["max_rel_err"] == c["gpu_vs_oracle_max_rel_err"]    # the metric's third part ("max rel-err vs Numba") at the top level
    assert d["value"] > 1e8


def _leaves(x):
    if isinstance(x, dict):
        for v in x.values():
            yield from _leaves(v)
    elif isinstance(x, list):
        for v in x:
            yield from _leaves(v)
    else:
        yield x


def test_default_run_writes_the_side_file_and_a_short_line(native_built, tmp_path):
    """The driver's own command shape (`--gpus 1 --steps 20 --warmup 5`, nothing else), with the secondary measurements on a
    short budget: ONE line of at most 8 192 bytes on stdout carrying `roofline`, `cpu_baseline`, `roofline_4m`, `configs` and
    `box`; the extras in the side file and on stderr, not on the line."""
    side = tmp_path / "extras.json"
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "2",
           "--extras-out", str(side), "--extras-budget-seconds", "25", "--no-live-traffic"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0].encode()) <= 8192
    d = json.loads(lines[0])
    assert d["config"]["baseline_config"] == "configs[4]" and d["config"]["bodies_per_gpu"] == 1048576 and d["roofline"]["resident"] == "hbm"
    assert 0.3 < d["roofline"]["frac"] <= 1.0 and 0.3 < d["roofline_4m"]["frac"] <= 1.0 and d["cpu_baseline"]["value"] > 0
    assert "frac_median_of_5" in d["roofline"] and d["extras_file"] == "extras.json" and "extras" not in d
    box = d["box"]                                               # which kind of box this is, on the line
    assert box["kernel_over_memory_only"] == pytest.approx(box["kernel_us"] / box["memory_only_us"], rel=1e-6)
    assert 1.0 < box["clock_held_ghz"] < 3.0 and box["throttles_under_combined_load"] == (box["kernel_over_memory_only"] >= 1.15)
    payload = json.loads(side.read_text())
    assert payload["line"]["value"] == d["value"] and "bound_probes_1m" in payload["extras"] and "clocks_1m" in payload["extras"]
    assert "bench.py: side file (extras):" in res.stderr


def _two_rank_run(steps, warmup, extra=()):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, HYDRO_BENCH_SHARE_GPU="1", HYDRO_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warmup),
           "--bodies", "65536", "--spinup-seconds", "0.2", *extra]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{")          # the gloo / RCCL banners go to stderr
    assert len(lines[0].encode()) <= 8192
    return json.loads(lines[0])


def test_bench_two_ranks_share_the_gpu(native_built):
    """The driver's N>1 launch (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N --steps 20 --warmup 5`)
    rehearsed with two ranks on this box's one GPU (gloo for the collectives, HYDRO_BENCH_SHARE_GPU=1): one JSON line from
    rank 0, whole-job aggregate over both ranks, no CPU leg - and the run PROVES SURVEY.md 8e by itself: the kinetic energy
    is sampled at least twice inside the 20 timed steps through the asynchronous monitor, the global value equals a
    float64 host sum over all bodies to 1e-12, every shard's wrench has the bits of the unsharded scene, and the number of
    ranks is the one the live group's all-reduce counted."""
    d = _two_rank_run(20, 5)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None and d["steps"] == 20 and d["warmup"] == 5
    assert d["config"]["bodies_per_gpu"] == 65536 and "x2" in d["config"]["sharding"]
    assert d["value"] == pytest.approx(2 * 65536 * 20 / (d["ms_per_step"] * 1e-3 * 20), rel=1e-6)
    col = d["collective"]
    assert col["ranks"] == 2 and col["rccl_ranks"] == 0 and col["backend"] == "gloo"      # gloo rehearsal: no RCCL rank in it
    assert len(d["per_rank"]["step_us"]) == 2 and max(d["per_rank"]["step_us"]) <= d["ms_per_step"] * 1e3 * 1.0001      # (+ the closing barrier)
    assert all(k <= s * 1.0001 for k, s in zip(d["per_rank"]["kernel_us"], d["per_rank"]["step_us"]))
    assert len(col["global_ke_J"]) == 2 and col["global_ke_J"][0] > 0
    assert col["ke_rel_err"] <= 1e-12                                             # two different 65 536-body scenes, summed over the ranks
    # N > 1 also runs BASELINE configs[3] as stated: 262 144 bodies block-partitioned over the ranks (strong scaling)
    cs = d["c4_strong"]
    assert cs["scaling"] == "strong" and cs["baseline_config"] == "configs[3]" and cs["n_gpus"] == 2
    assert cs["bodies_total"] == 262144 and cs["bodies_this_rank"] == 131072
    assert cs["value"] == pytest.approx(262144 * 20 / (cs["ms_per_step"] * 1e-3 * 20), rel=1e-6)
    ke = cs["ke"]
    assert ke["every_steps"] == 10 and ke["samples"] >= 2 and ke["sampled_at_steps"] == [10, 20] and ke["last_step"] == 20
    assert ke["rel_err"] <= 1e-12 and ke["rel_err_gate"] == 1e-12
    assert cs["shards_bit_identical"] is True and d["ok"] is True
    assert "skipped" in cs["captured"]                                             # gloo: a CPU collective cannot live in a HIP graph
    # the host sum on the line is the one this test computes itself
    import numpy as np
    sys.path.insert(0, REPO)
    from silver2_isaacsim_amd import scenes
    sc = scenes.scene_c4(n=262144, seed=4)
    m = sc.params[:, 10].astype(np.float64)
    lin = float((0.5 * m * (sc.state[:, 7:10].astype(np.float64) ** 2).sum(1)).sum())
    assert ke["host_fp64_J"][0] == pytest.approx(lin, rel=1e-11) and ke["global_J"][0] == pytest.approx(lin, rel=1e-11)      # (12 digits on the line)


def test_bench_two_ranks_long_region_keeps_the_256_step_cadence(native_built):
    """600 timed steps: one sample per 256 steps (two in the region), 64-step graph replays, same self-checks."""
    cs = _two_rank_run(600, 8)["c4_strong"]
    ke = cs["ke"]
    assert ke["every_steps"] == 256 and ke["samples"] == 2 and ke["last_step"] == 512 and cs["graph_steps"] == 64
    assert ke["rel_err"] <= 1e-12 and cs["shards_bit_identical"] is True


def test_bench_gpus_2_launches_its_own_ranks(native_built):
    """`python bench.py --gpus 2` with NO torchrun environment must not fall through to a one-GPU run: it starts the two
    ranks itself (before anything touches the GPU) and relays rank 0's line.  Rehearsed on this box's one GPU with both
    ranks sharing it and gloo for the collectives."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HYDRO_BENCH_SHARE_GPU="1", HYDRO_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "8", "--bodies", "65536",
           "--spinup-seconds", "0.2"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{")
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["collective"]["backend"] == "gloo" and d["collective"]["ranks"] == 2 and "x2" in d["config"]["sharding"]
    assert d["collective"]["barrier"] == "node-local shared-memory epoch barrier"      # the ranks of one host time their region with it
    assert d["c4_strong"]["n_gpus"] == 2 and d["c4_strong"]["bodies_this_rank"] == 131072
    assert d["c4_strong"]["ke"]["samples"] >= 2 and d["c4_strong"]["shards_bit_identical"] is True


def test_bench_refuses_more_gpus_than_visible(native_built):
    """--gpus N with fewer than N devices visible (and no rehearsal knob): non-zero exit, a message, no JSON line."""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "HYDRO_BENCH_SHARE_GPU")}
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode != 0 and res.stdout.strip() == ""
    assert f"--gpus {n} but only {n - 1} GPU(s) are visible" in res.stderr


def test_resident_loop_roofline_is_an_upper_bound(native_built):
    """The VALU-issue roofline of the compute-bound resident loop: its peak is the hardware's issue rate
    at the boost clock, so `frac` <= 1 on every box - the round-4 form priced the instruction classes with a microbenchmark's
    own readings at the 2.4 GHz spec clock and read 1.03 on the driver's box."""
    sys.path.insert(0, REPO)
    from scripts import bench_extras as bench
    for n, steps in ((1048576, 640), (262144, 1024)):
        r = bench.closed_loop_rate("c2", n, steps=steps, resident=True)
        roof = r["roofline"]
        assert roof["bound"] == "valu-issue" and 0.35 < roof["frac"] <= 1.0, roof
        assert roof["frac"] == pytest.approx(roof["floor_us_per_step"] / r["us_per_step"])
        assert roof["achieved"] <= roof["peak"] and roof["achieved"] / roof["peak"] == pytest.approx(roof["frac"], rel=1e-9)
        assert roof["model_measured_prices"]["kind"] == "model, not a bound"
    # what the bound is made of: 2 cycles per wave64 instruction, 4 for fp64 arithmetic, 2.55 GHz, 1 024 SIMDs
    mix = roof["valu_by_class"]
    cycles = 4.0 * mix["fp64 arithmetic"] + 2.0 * (roof["valu_instructions_per_body_step"] - mix["fp64 arithmetic"])
    assert roof["issue_cycles_per_wave_step_at_hardware_rate"] == cycles
    assert roof["floor_us_per_step"] == pytest.approx(cycles * (262144 / 64 / 1024) / 2.55e3)


def test_plugin_own_host_cost_is_reported_separately(native_built):
    """the plugin's own host time per physics step, next to - and below - the figure that includes the
    in-memory simulator's stepping."""
    sys.path.insert(0, REPO)
    from scripts import bench_extras as bench
    own = bench.plugin_own_rate(steps=1500)
    full = bench.plugin_rate(True, steps=1500)
    assert own["prims"] == 20 and own["apply_calls"] >= 1500
    # (host timings on a shared box: the two pieces of the plugin's own cost are 2-3 us apart, so only the wide gaps are asserted)
    assert 0 < own["prepared_launch_alone_us"] < full["us_per_physics_step"]
    assert own["plugin_own_us_per_step"] < full["us_per_physics_step"]
    assert own["plugin_own_us_per_step"] < 25.0               # a regression guard, not a target: measured 5.7-6 us


def test_bench_four_ranks_share_the_gpu(native_built):
    """The N = 4 shape of the driver's scaling run, rehearsed on one GPU (gloo collectives; the box allows six GPU processes):
    four shards of 65 536 bodies of configs[3], four different weak-scaling scenes, one JSON line - and every self-check of the
    N > 1 line green with more than two ranks (rank count, both kinetic-energy errors, shard bit-identity)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, HYDRO_BENCH_SHARE_GPU="1", HYDRO_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "4", "--steps", "20", "--warmup", "5",
           "--bodies", "32768", "--spinup-seconds", "0.1"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.strip()][0])
    assert d["n_gpus"] == 4 and d["collective"]["ranks"] == 4 and "x4" in d["config"]["sharding"]
    assert d["collective"]["ke_rel_err"] <= 1e-12 and len(res.stdout.strip().encode()) <= 8192
    cs = d["c4_strong"]
    assert cs["bodies_this_rank"] == 65536 and cs["ke"]["samples"] >= 2
    assert cs["ke"]["rel_err"] <= 1e-12 and cs["shards_bit_identical"] is True


def _faulty_run(fault, via_torchrun=True, strong_timeout="8", teardown_timeout="10"):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HYDRO_BENCH_SHARE_GPU="1", HYDRO_DIST_BACKEND="gloo", HYDRO_BENCH_STRONG_FAULT=fault, HYDRO_BENCH_STRONG_TIMEOUT=strong_timeout,
               HYDRO_BENCH_TEARDOWN_TIMEOUT=teardown_timeout)
    tail = [os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--bodies", "65536", "--spinup-seconds", "0.1"]
    if via_torchrun:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + tail
    else:
        cmd = [sys.executable] + tail
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)


@pytest.mark.parametrize("fault", ["raise:1", "hang:1", "raise:0", "hang-resident:0"])
def test_headline_survives_a_failing_strong_leg(native_built, fault):
    """The configs[3] leg runs after the headline measurement and before the JSON line, on hardware nobody has tried it on.
    Whatever happens in it - a rank raises, a rank never arrives at a collective - rank 0 still prints the headline it has, with
    "ok": false and `c4_strong.error`, and THEN the job exits non-zero (scripts/bench_strong.py LegGuard: a failing rank k > 0 waits
    for rank 0's line before it leaves, because torchrun ends the job at the first failed rank).  A hang must not look like a pass
    to a driver that gates on the exit code."""
    res = _faulty_run(fault)
    assert res.returncode != 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1 and len(lines[0].encode()) <= 8192
    d = json.loads(lines[0])
    assert d["ok"] is False and d["n_gpus"] == 2 and d["value"] > 1e8 and d["collective"]["ranks"] == 2 and d["cpu_baseline"] is None
    assert len(d["per_rank"]["step_us"]) == 2 and d["collective"]["ke_rel_err"] <= 1e-12
    if fault.startswith("hang-resident"):          # the host-driven leg had finished: its results are on the line, only the variant is lost
        cs = d["c4_strong"]
        assert cs["ke"]["samples"] >= 2 and cs["shards_bit_identical"] is True and cs["value"] > 0
        assert "the headline on this line is complete" in cs["captured"]["error"]
    else:
        assert "error" in d["c4_strong"] and "the headline on this line is complete" in d["c4_strong"]["error"]
    assert "configs[3] leg" in res.stderr


def test_self_launch_relays_the_line_and_exit_code_3(native_built):
    """`python bench.py --gpus 2` (no torchrun environment) with a failing leg: the parent relays the child's line AND fails
    with exit code 3 - it used to drop the line whenever the child's code was non-zero."""
    res = _faulty_run("raise:1", via_torchrun=False)
    assert res.returncode == 3, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["ok"] is False and d["n_gpus"] == 2 and d["value"] > 1e8 and "error" in d["c4_strong"]


def test_a_rank_that_gives_up_at_the_teardown_barrier_lets_rank_0_print_first(native_built):
    """The production ordering of the deadlines: the teardown barrier (60 s) is SHORTER than the leg's watchdog (240 s).  Rank 0 hangs
    in the captured variant; rank 1 is through its leg and gives up on the teardown barrier long before rank 0's watchdog would
    fire.  It must not simply exit (torchrun would end the job and the headline with it): it leaves through the guard - marker,
    wait for rank 0's line - and rank 0, polling for markers, prints at once.  Here: watchdog 90 s, teardown 5 s, done well under 60 s."""
    import time
    t0 = time.monotonic()
    res = _faulty_run("hang-resident:0", strong_timeout="90", teardown_timeout="5")
    took = time.monotonic() - t0
    assert res.returncode != 0 and took < 75, (took, res.stderr[-2000:])
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["ok"] is False and d["value"] > 1e8 and d["c4_strong"]["shards_bit_identical"] is True
    assert "teardown" in d["c4_strong"]["captured"]["error"] and "node barrier" in d["c4_strong"]["captured"]["error"]


def test_bench_through_the_array_of_structs_entry(native_built):
    """`--layout aos` (what scripts/profile_aos.sh profiles): the simulator-facing entry as the bench workload - 168 algorithmic bytes
    per body-step, all of them real traffic, eight or more rotating sets so that no rows stay in the Infinity Cache."""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--layout", "aos", "--workload", "c5-f32",
           "--bodies", "65536", "--cpu-seconds", "1", "--no-extras", "--no-configs"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.strip()][0])
    assert d["config"]["entry_point"] == "hydro_step_wrench_aos" and d["config"]["bytes_per_body_step"] == 168
    assert d["roofline"]["kernel"] == "wrench_aos_direct_kernel" and d["roofline"]["traffic_bytes_per_body"] == 168
    assert d["config"]["scene_replicas_per_gpu"] * 65536 * 52 >= (410 << 20) and "configs" not in d
    assert d["cpu_baseline"]["gpu_vs_oracle_max_rel_err"] <= 1e-5 and d["max_rel_err"] <= 1e-5      # the CPU leg checks this entry's result too


@pytest.mark.parametrize("flags,entry,n", [
    (["--workload", "c2"], "hydro_step_wrench_tiled", 4096),
    (["--workload", "c3"], "hydro_step_wrench_tiled", 19456),
    (["--workload", "c4", "--layout", "soa"], "hydro_step_wrench_ext", 262144),
    (["--workload", "c5-f32", "--bodies", "100001"], "hydro_step_wrench_tiled", 100001),          # a ragged last tile
])
def test_bench_other_workloads_and_layouts(native_built, flags, entry, n):
    """Every workload / layout the bench offers still produces a checked line (the CPU leg is the checker: gate 1e-5)."""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "0.5",
           "--no-extras", "--no-configs", "--spinup-seconds", "0.1"] + flags
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.strip()][0])
    assert d["config"]["entry_point"] == entry and d["config"]["bodies_per_gpu"] == n and d["ok"] is True
    assert d["cpu_baseline"]["gpu_vs_oracle_n_over_1e-5"] == 0 and d["max_rel_err"] <= 1e-5
    assert d["collective"]["ke_rel_err"] <= 1e-12 and d["value"] == pytest.approx(n / (d["ms_per_step"] * 1e-3), rel=1e-9)


def test_bench_strong_scaling_flag_two_ranks(native_built):
    """`--scaling strong`: the workload's bodies block-partitioned over the ranks (two, sharing the GPU over gloo)."""
    d = _two_rank_run(20, 5, extra=("--scaling", "strong", "--no-strong-leg"))
    assert d["scaling"] == "strong" and d["config"]["bodies_per_gpu"] == 32768 and "strong scaling: 65536 bodies over 2 GPUs" in d["config"]["workload"]
    assert d["value"] == pytest.approx(65536 / (d["ms_per_step"] * 1e-3), rel=1e-6) and "c4_strong" not in d
    assert d["collective"]["ke_rel_err"] <= 1e-12
