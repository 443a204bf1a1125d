#!/usr/bin/env python3
"""More than one GPU as a USER of the package would write it: one process per GPU, the bodies of one scene
block-partitioned over the ranks, every rank steps its shard in closed loop on its own device (no collective on the step
path), and the global kinetic energy - the one collective - is sampled inside the step kernel every `--every` steps and summed
over the ranks asynchronously (simulate.ClosedLoopSim(ke_every=...) -> KineticEnergyMonitor: RCCL over xGMI under backend nccl).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/sharded_closed_loop.py
    HYDRO_DIST_BACKEND=gloo HYDRO_EXAMPLE_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node 2 ...   (one-GPU rehearsal)

Waiting for the stream or for a sample has a deadline (`--timeout`): a rank that never joins a collective makes the others
exit 4 with a TimeoutError naming the step and the rank, not hang.  Rank 0 prints one JSON line: the global kinetic energy at
every sampling point, its value at the end against a float64 host sum over the gathered final state of ALL shards, and
whether the sharded run reproduces the unsharded one bit for bit."""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np                                                   # noqa: E402
import torch                                                         # noqa: E402

from silver2_isaacsim_amd import distributed as hd, scenes          # noqa: E402
from silver2_isaacsim_amd.simulate import ClosedLoopSim             # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--bodies", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--every", type=int, default=64)
    ap.add_argument("--timeout", type=float, default=120.0, help="deadline (s) for the step stream and for a kinetic-energy sample")
    args = ap.parse_args(argv)
    rank, local_rank, world = hd.env_rank_world()
    hd.init_process_group()
    ndev = torch.cuda.device_count()
    share = os.environ.get("HYDRO_EXAMPLE_SHARE_GPU") == "1"          # rehearsal: several ranks on one GPU (with gloo)
    dev = local_rank % ndev if share else local_rank
    torch.cuda.set_device(dev)

    full = scenes.scene_c2(n=args.bodies, seed=9)                     # buoys bobbing at the surface; the same scene on every rank ...
    mine = full.shard(rank, world)                                    # ... each keeps its contiguous block
    sim = ClosedLoopSim(mine, device=dev, ke_every=args.every, sample_timeout_s=args.timeout)    # (local; the first run warms the monitor up, collectively)
    try:
        sim.run(args.steps, graph_steps=args.every)                   # HIP-graph replays; the last step of a replay samples
        sim.synchronize(timeout_s=args.timeout)                       # a deadline, not a hang, should a rank never join a collective
        sim.monitor.collect(block=True, timeout_s=args.timeout)
    except TimeoutError as e:
        sys.stderr.write(f"sharded_closed_loop: {e}\n")
        sys.stderr.flush()
        os._exit(4)                                                   # (other ranks may sit in a collective: leave, do not wait, do not retry)
    samples = sim.monitor.samples                                     # [(step, [translational, rotational])], identical on every rank

    # checks: gather every shard's final state on rank 0 (exact: distributed.gather_rows)
    lo, hi = hd.shard_range(full.n, rank, world)
    flat = np.zeros((full.n, 13), dtype=np.float64)
    flat[lo:hi] = sim.state()
    rows = hd.gather_rows(flat.reshape(-1), torch.device("cuda", dev)).sum(dim=0).reshape(full.n, 13).numpy()
    out = None
    if rank == 0:
        final = rows.astype(np.float32)
        host = scenes.kinetic_energy_fp64(final, full.params)
        last = samples[-1][1]
        whole = ClosedLoopSim(full, device=dev)
        whole.run(args.steps, graph_steps=args.every)
        same = bool(np.array_equal(whole.state(), final))
        whole.close()
        out = {"ranks": world, "bodies": full.n, "bodies_per_rank": hi - lo, "steps": args.steps,
               "kinetic_energy_J": [{"step": s, "translational": v[0], "rotational": v[1]} for s, v in samples],
               "last_sample_rel_err_vs_host_fp64": max(abs(last[k] - host[k]) / host[k] for k in range(2)),
               "sharded_equals_unsharded_bit_for_bit": same, "host_waits": sim.monitor.waited_on_host}
        print(json.dumps(out))
    sim.close()
    hd.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
