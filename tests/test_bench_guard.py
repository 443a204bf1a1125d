"""scripts/bench_strong.LegGuard on the CPU (world-size-2 gloo, no GPU): the ORDER in which a failing N > 1 bench run leaves.
torchrun ends a job at the first failed rank, so a rank k > 0 that fails must wait until rank 0 has its line out; rank 0 reacts to
the marker within a fraction of a second, prints the headline it has with "ok": false, and only then do both exit with code 3."""
import json
import os
import socket

import torch.multiprocessing as mp

from conftest import REPO

HEADLINE = {"metric": "body-steps/sec", "value": 1.0e10, "unit": "body-steps/s", "n_gpus": 2, "steps": 20, "warmup": 5, "ms_per_step": 0.02,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "ok": True,
            "config": {"workload": "canned"}, "roofline": {"bound": "hbm", "frac": 0.5}}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, scenario, out_path, stamp_path):
    import sys
    import time
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from silver2_isaacsim_amd import distributed as hd
    assert hd.init_process_group(backend="gloo")
    from scripts import bench_strong
    fd = os.open(out_path, os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o600) if rank == 0 else -1
    guard = bench_strong.LegGuard(rank, dict(HEADLINE) if rank == 0 else None, fd, timeout_s=30.0)
    if scenario == "rank1_raises":
        if rank == 1:
            time.sleep(1.0)
            guard.leave("RuntimeError('boom') on rank 1; the headline on this line is complete")
        time.sleep(60)                                        # rank 0: "blocked in a collective" - only its watchdog thread can act
    elif scenario == "leg_done_then_rank1_gives_up_at_teardown":
        guard.progress["main"] = {"value": 5.0e9, "shards_bit_identical": True}
        if rank == 1:
            guard.leg_done()
            time.sleep(1.0)
            guard.leave("teardown: TimeoutError('node barrier') on rank 1")
        time.sleep(60)                                        # rank 0 hangs in the captured variant
    elif scenario == "line_already_out":
        if rank == 0:
            os.write(fd, (json.dumps(HEADLINE) + "\n").encode())
            guard.line_is_out()
        else:
            time.sleep(1.0)
            guard.leave("teardown: rank 1 gave up")
        time.sleep(60)
    elif scenario == "all_good":
        guard.leg_done()
        if rank == 0:
            os.write(fd, (json.dumps(HEADLINE) + "\n").encode())
            guard.line_is_out()
        hd.barrier()
        guard.finish()
        with open(stamp_path + f".{rank}", "w") as f:
            f.write(guard.base)
        os._exit(0)


def _run(scenario, tmp_path):
    world, port = 2, _free_port()
    out, stamp = str(tmp_path / "stdout.txt"), str(tmp_path / "stamp")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, scenario, out, stamp)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
    lines = [json.loads(l) for l in open(out).read().splitlines() if l.strip()] if os.path.exists(out) else []
    return [p.exitcode for p in procs], lines, stamp


def test_rank_0_prints_before_a_failing_rank_leaves(tmp_path):
    codes, lines, _ = _run("rank1_raises", tmp_path)
    assert codes == [3, 3]
    assert len(lines) == 1 and lines[0]["ok"] is False and lines[0]["value"] == 1.0e10 and lines[0]["cpu_baseline"] is None
    assert "boom" in lines[0]["c4_strong"]["error"] and lines[0]["c4_strong"]["baseline_config"] == "configs[3]"


def test_a_finished_leg_stays_on_the_line_when_only_the_variant_is_lost(tmp_path):
    codes, lines, _ = _run("leg_done_then_rank1_gives_up_at_teardown", tmp_path)
    assert codes == [3, 3] and len(lines) == 1
    cs = lines[0]["c4_strong"]
    assert lines[0]["ok"] is False and cs["value"] == 5.0e9 and cs["shards_bit_identical"] is True and "teardown" in cs["captured"]["error"]


def test_a_line_that_is_out_is_never_printed_twice(tmp_path):
    codes, lines, _ = _run("line_already_out", tmp_path)
    assert codes == [3, 3]                                    # the failure is the exit code ...
    assert len(lines) == 1 and lines[0]["ok"] is True        # ... the complete line stays the only one


def test_a_clean_run_leaves_no_markers_and_exits_0(tmp_path):
    import glob
    codes, lines, stamp = _run("all_good", tmp_path)
    assert codes == [0, 0] and len(lines) == 1
    base = open(stamp + ".0").read()
    assert base == open(stamp + ".1").read() and not glob.glob(base + "*")
