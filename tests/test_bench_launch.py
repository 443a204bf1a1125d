"""bench.py --gpus N can not be mis-launched: without a torchrun environment it starts its own ranks, and it refuses
to benchmark fewer GPUs than asked.  CPU-only half (no device is touched before the check)."""
import os
import subprocess
import sys

from conftest import REPO


def test_gpus_n_without_devices_exits_non_zero():
    import torch
    n = torch.cuda.device_count() + 2
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "HYDRO_BENCH_SHARE_GPU")}
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n)], capture_output=True, text=True,
                         timeout=300, env=env)
    assert res.returncode != 0
    assert res.stdout.strip() == ""                                  # no JSON line that could be mistaken for a result
    assert f"--gpus {n} but only {n - 2} GPU(s) are visible" in res.stderr


def test_world_size_mismatch_is_an_error():
    """A torchrun environment whose WORLD_SIZE differs from --gpus is refused before any device work."""
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert res.returncode != 0 and "--gpus 2 but WORLD_SIZE=3" in res.stderr and res.stdout.strip() == ""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert res.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in res.stderr


def _canned(n_gpus: int) -> dict:
    """A line as bench.py builds it, every block present, every float at full width - the widest shape the file can produce."""
    w = 1234567.890123456789
    line = {"metric": "body-steps/sec", "value": 4.7e10 + w, "unit": "body-steps/s", "n_gpus": n_gpus, "steps": 20, "warmup": 5,
            "ms_per_step": 0.0221234567890123, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "ok": True,
            "config": {"workload": "C5: 1 048 576 bodies/GPU, fp32 state, fp16 coefficients, fp64 arithmetic rounded to fp32 once",
                       "baseline_config": "configs[4]", "bodies_per_gpu": 1048576, "coefficients": "f16", "scene_replicas_per_gpu": 4,
                       "bytes_per_body_step": 130, "sharding": f"bodies x{n_gpus} (no data-path collective)",
                       "layout": "tiled SoA [tile][field][64]", "entry_point": "hydro_step_wrench_tiled"},
            "roofline": {k: w for k in ("achieved", "peak", "frac", "traffic", "step_us", "kernel_us", "frac_contract_steps",
                                        "algorithmic_bytes_per_launch", "traffic_bytes_per_body", "frac_traffic", "traffic_committed",
                                        "traffic_bytes_per_body_measured", "kernel_us_median_of_5", "frac_median_of_5", "working_set_bytes")}
            | {"bound": "hbm", "unit": "GB/s", "kernel": "wrench_tiled_kernel", "traffic_measured": "live", "resident": "hbm"},
            "max_rel_err": 2.3688437725e-07,
            "collective": {"backend": "nccl (RCCL)", "ranks": n_gpus, "rccl_ranks": n_gpus, "barrier": "node-local shared-memory epoch barrier",
                           "global_ke_J": [w, w], "host_fp64_ke_J": [w, w], "ke_rel_err": 1.5e-16, "ke_allreduce_us": w},
            "cpu_baseline": {"value": w, "unit": "body-steps/s", "cores": 1, "kind": "port",
                             "sample": "262144 bodies of the bench scene x 382 passes, oracle/hydro_oracle.c (fp64 C port of the Numba path), 1 thread",
                             "all_core_value": w, "all_cores": 16, "hardware_threads": 256, "cpu_model": "AMD EPYC 9575F 64-Core Processor",
                             "gpu_vs_oracle_max_rel_err": 2.36e-7, "gpu_vs_oracle_n_over_1e-5": 0, "gpu_vs_oracle_checked": 262144},
            "roofline_4m": {k: w for k in ("achieved", "peak", "frac", "traffic", "kernel_us", "bodies", "algorithmic_bytes_per_launch",
                                           "frac_traffic", "working_set_bytes", "steps")} | {"bound": "hbm", "unit": "GB/s", "kernel": "wrench_tiled_kernel", "resident": "hbm"},
            "configs": {c: {"n": 262144, "us_per_step": w, "body_steps_per_s": w, "graph_us_per_step": w, "graph_body_steps_per_s": w}
                        for c in ("c2", "c3", "c4_shard", "c4")},
            "box": {k: w for k in ("memory_only_us", "compute_only_us", "kernel_us", "kernel_over_memory_only", "clock_held_ghz", "memory_only_ghz",
                                   "compute_only_ghz")} | {"binding": "hbm", "throttles_under_combined_load": False},
            "extras_file": "bench_extras.json"}
    if n_gpus > 1:
        line["cpu_baseline"] = None
        line["per_rank"] = {"step_us": [w] * n_gpus, "kernel_us": [w] * n_gpus}
        line["c4_strong"] = {"value": w, "unit": "body-steps/s", "scaling": "strong", "baseline_config": "configs[3]", "bodies_total": 262144,
                             "bodies_this_rank": 32768, "n_gpus": n_gpus, "steps": 2000, "warmup": 100, "ms_per_step": w, "kernel_us_rank0": w,
                             "graph_steps": 64,
                             "ke": {"every_steps": 256, "samples": 7, "host_waits": 0, "sampled_at_steps": [256 * k for k in range(1, 8)],
                                    "last_step": 1792, "global_J": [w, w], "host_fp64_J": [w, w], "rel_err": 1.5e-16, "rel_err_gate": 1e-12},
                             "shards_bit_identical": True, "resident": "infinity-cache",
                             "captured": {"value": w, "ms_per_step": w, "kernel_us_rank0": w, "samples": 7,
                                          "sampled_at_steps": [256 * k for k in range(1, 8)], "rel_err": 1.5e-16}}
    return line


def test_the_line_fits_8192_bytes_without_a_gpu():
    """the bound the driver needs, checked on the CPU from canned result dicts - the N = 1 shape with every
    optional block, the N = 8 shape with `per_rank` and `c4_strong`, and a line whose leg failed with a long error text."""
    import json
    import sys
    sys.path.insert(0, REPO)
    import bench
    assert bench.LINE_LIMIT == 8192
    for n in (1, 2, 8):
        line = bench.render_line(_canned(n))
        assert len(line.encode()) <= 8192 and "\n" not in line, (n, len(line))
        d = json.loads(line)
        assert "dropped_for_size" not in d and d["n_gpus"] == n          # nothing had to go: the shapes fit by a wide margin
        assert len(line.encode()) <= 7168, len(line)
        assert d["value"] == _canned(n)["value"] and d["ms_per_step"] == _canned(n)["ms_per_step"]      # these two in full
        assert d["roofline"]["frac"] == float(f"{1234567.890123456789:.12g}")                            # the rest to 12 digits
    failed = dict(_canned(8), ok=False, c4_strong={"error": "RuntimeError('x')" + " y" * 150, "baseline_config": "configs[3]"})
    assert len(bench.render_line(failed).encode()) <= 8192


def test_a_line_that_would_not_fit_sheds_whole_blocks_never_the_contract():
    """`render_line` holds the limit by construction: non-contract blocks go, the largest first, and are named; the contract keys stay."""
    import json
    import sys
    sys.path.insert(0, REPO)
    import bench
    fat = dict(_canned(8))
    fat["box"] = {f"k{i}": 1.23456789 for i in range(400)}
    fat["per_rank"] = {"step_us": [1.23456789] * 400, "kernel_us": [1.23456789] * 400}
    fat["somebody_added_this"] = "z" * 9000
    d = json.loads(bench.render_line(fat))
    assert d["dropped_for_size"] == ["somebody_added_this", "per_rank", "box"]
    assert all(k in d for k in bench.CONTRACT_KEYS) and "c4_strong" in d and "configs" in d
    assert len(bench.render_line(fat).encode()) <= 8192
    # non-finite numbers never reach the line (strict JSON): they become null
    assert json.loads(bench.render_line(dict(_canned(1), max_rel_err=float("nan"))))["max_rel_err"] is None


def test_write_all_loops_over_short_writes(monkeypatch):
    import sys
    sys.path.insert(0, REPO)
    import bench
    got = []

    def short_write(fd, data):
        got.append(bytes(data[:7]))
        return len(got[-1])
    monkeypatch.setattr(os, "write", short_write)
    bench.write_all(1, b"0123456789abcdefghij")
    assert b"".join(got) == b"0123456789abcdefghij" and len(got) == 3


def test_explain_prints_the_field_notes():
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--explain"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "roofline.frac" in res.stdout and "kernel_over_memory_only" in res.stdout


def test_box_summary_from_canned_probes():
    """`box` on the line (scripts/bench_extras.box_summary): which bound binds on this box, the clock it holds, and the flag -
    ordinary boxes read 1.03-1.06, boxes that throttle under the combined load 1.20-1.32; the line between them is 1.15."""
    import sys
    sys.path.insert(0, REPO)
    from scripts import bench_extras as be
    probes = {"memory_only_us": 20.5, "compute_only_us": 16.1, "kernel_us": 21.4, "kernel_over_memory_only": 21.4 / 20.5, "binding": "hbm"}
    clocks = {"whole_body_ghz": 2.09, "memory_only_ghz": 2.37, "compute_only_ghz": 2.56}
    box = be.box_summary({"bound_probes_1m": probes, "clocks_1m": clocks})
    assert box["throttles_under_combined_load"] is False and box["clock_held_ghz"] == 2.09 and box["binding"] == "hbm"
    slow = dict(probes, kernel_us=24.53, kernel_over_memory_only=24.53 / 20.46)
    assert be.box_summary({"bound_probes_1m": slow, "clocks_1m": clocks})["throttles_under_combined_load"] is True
    assert be.box_summary({"bound_probes_1m": {"error": "x"}}) is None and be.box_summary({}) is None
    only_probes = be.box_summary({"bound_probes_1m": probes, "clocks_1m": {"skipped": "extras time budget"}})
    assert "clock_held_ghz" not in only_probes and only_probes["kernel_us"] == 21.4
    assert be.THROTTLE_RATIO == 1.15 and "kernel_over_memory_only >= 1.15" in be.FIELD_NOTES
